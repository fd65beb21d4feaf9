"""A/B: padded encoder time at mid-size batches (256 < tiles <= 512 for the wide linears) — GDR_GEMM_STREAMK_MID."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
enc = ops.T5EncoderHandle(cfg, sd, dev)
for B in (48, 56, 64, 72, 80, 96, 112):
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    f = lambda: enc.forward(it, mt)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    tm = (B * 40 + 127) // 128
    print(f"MID={os.environ.get('GDR_GEMM_STREAMK_MID','512')} B={B} rows={B*40} T(qkv)={tm*18} T(wi)={tm*24}: {(time.perf_counter()-t0)/10*1e3:.3f} ms")
