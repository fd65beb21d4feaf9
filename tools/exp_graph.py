"""HIP-graph replay of gdr_t5_generate vs eager launches: same outputs, decode time at B=1 x 100 beams and B=64 x 10."""
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
    a = model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table)
    b = model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table, graph=True)
    c = model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table, graph=True)
    torch.cuda.synchronize()
    print("identical:", all(torch.equal(x, y) for x, y in zip(a, b)), all(torch.equal(x, y) for x, y in zip(a, c)))
    for g in (False, True):
        for _ in range(3):
            model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table, graph=g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table, graph=g)
        torch.cuda.synchronize()
        print(f"B={B} R={R} graph={g}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per decode")
