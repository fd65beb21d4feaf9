#!/usr/bin/env python3
"""The encoder's four linears at a packed batch's row count under the forms gemm_f32.hip can take (whole tiles only, stream-K
tail, one workgroup per CU): one process per form, selected by GDR_GEMM_STREAMK / GDR_GEMM_STREAMK_MID."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, _ffi
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.05).to(dev)
ws = torch.empty(40 << 20, dtype=torch.uint8, device=dev)
shapes = [("qkv", 2304, 768, False, _ffi.EPI_NONE), ("o", 768, 768, True, _ffi.EPI_NONE), ("wi", 3072, 768, False, _ffi.EPI_RELU),
          ("wo", 768, 3072, True, _ffi.EPI_NONE)]
only = os.environ.get("ONLY")
for M in [int(x) for x in os.environ.get("MS", "12308").split(",")]:
    for name, N, K, res, epi in shapes:
        if only and name not in only.split(","):
            continue
        a, w = rnd(M, K), rnd(N, K)
        r = rnd(M, N) if res else None
        out = torch.empty(M, N, device=dev)
        for _ in range(5):
            ops.linear(a, w, epi, residual=r, out=out, splitk_ws=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for _ in range(n):
            ops.linear(a, w, epi, residual=r, out=out, splitk_ws=ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"  M={M} {name:4s} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
