"""Is generate() launch-bound on the host?  Time the enqueue (call returns) against the synchronised total."""
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
for B, R in ((1, 100), (64, 10)):
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    enc_h, _ = model.enc.forward(ids, mask, want_pooled=False, ragged=True)
    for _ in range(3):
        model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table)
    torch.cuda.synchronize()
    te, tt = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        te.append(t1 - t0); tt.append(t2 - t0)
    print(f"B={B} R={R}: enqueue {sorted(te)[5]*1e3:.2f} ms, total {sorted(tt)[5]*1e3:.2f} ms (decode only)")
