"""generate() throughput against the query batch size: where does the decode chain stop being launch-bound?"""
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
for B, R in ((64, 10), (64, 30), (128, 10), (256, 10), (512, 10)):
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    f = lambda: model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8, num_return_sequences=R, output_scores=True)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): f()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 4
    print(f"MID={os.environ.get('GDR_GEMM_STREAMK_MID','512')} SK={os.environ.get('GDR_GEMM_STREAMK','6')} B={B} beams={R}: {t*1e3:.2f} ms  {B/t:.0f} q/s")
