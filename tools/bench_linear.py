#!/usr/bin/env python3
"""Per-shape timing of gdr_linear_f32 at the encoder's call shapes (B=512, L=40 -> M=20480), with/without the
residual epilogue and in a back-to-back sequence like a layer.  Feeds DESIGN.md's GEMM discussion."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, _ffi
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 20480))
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.05).to(dev)
shapes = [("qkv", 2304, 768, False, _ffi.EPI_NONE), ("o", 768, 768, True, _ffi.EPI_NONE), ("o-nores", 768, 768, False, _ffi.EPI_NONE),
          ("ffn1", 3072, 768, False, _ffi.EPI_RELU), ("ffn2", 768, 3072, True, _ffi.EPI_NONE), ("ffn2-nores", 768, 3072, False, _ffi.EPI_NONE)]
for name, N, K, res, epi in shapes:
    a, w = rnd(M, K), rnd(N, K)
    r = rnd(M, N) if res else None
    out = torch.empty(M, N, device=dev)
    for _ in range(5): ops.linear(a, w, epi, residual=r, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): ops.linear(a, w, epi, residual=r, out=out)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / n * 1e6
    print(f"{name:11s} M={M} N={N} K={K} res={res}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF")

# cold-cache variant: a 1 GiB fill between calls evicts L2 + Infinity Cache; hipEvent timing of the GEMM alone
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
for name, N, K, res, epi in shapes[:5]:
    a, w = rnd(M, K), rnd(N, K)
    r = rnd(M, N) if res else None
    out = torch.empty(M, N, device=dev)
    ts = []
    for it in range(8):
        big.fill_(float(it))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.linear(a, w, epi, residual=r, out=out); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts[2:])[len(ts[2:]) // 2]
    print(f"cold {name:11s}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF")
