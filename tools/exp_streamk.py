import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
enc = ops.T5EncoderHandle(cfg, sd, dev)
ids, mask = synth.make_tokens(512, L=40, seed=11)
it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("ragged pooled-only ms", t(lambda: enc.forward(it, mt, want_hidden=False, ragged=True, live_rows_hint=int(mask.sum()))))
print("padded ms", t(lambda: enc.forward(it, mt)))
h0, p0 = enc.forward(it, mt); _, p1 = enc.forward(it, mt, want_hidden=False, ragged=True)
print("pooled equal", torch.equal(p0, p1))
