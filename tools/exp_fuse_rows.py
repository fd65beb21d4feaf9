"""A/B: generate() time at small row counts against GDR_DECODE_FUSE_MIN_ROWS (fused split-K reduce + residual + norm)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
names = synth.make_cluster_ids(320000, cluster_size=12, V=30)[0]
model = GDRModel(cfg, synth.make_state_dict(cfg, seed=1234), dev, ragged=True, prefix_trie=codec.Trie.from_docids(names, 30))
for B, R in ((1, 10), (4, 10), (1, 100), (16, 10), (32, 10), (64, 10)):
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    f = lambda: model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8, num_return_sequences=R, output_scores=True)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): f()
    torch.cuda.synchronize()
    print(f"min_rows={os.environ.get('GDR_DECODE_FUSE_MIN_ROWS','1')} B={B} beams={R}: {(time.perf_counter()-t0)/8*1e3:.2f} ms")
