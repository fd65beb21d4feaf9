#!/usr/bin/env python3
"""time(K) at fixed M, N for gdr_linear_f32: slope = steady-state k-loop, intercept = per-tile prologue/epilogue + launch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, _ffi
dev = torch.device("cuda:0"); M = 20480
for N in (768, 2304):
    xs, ys = [], []
    for K in (256, 512, 768, 1536, 3072):
        a = (torch.randn(M, K) * 0.05).to(dev); w = (torch.randn(N, K) * 0.05).to(dev); out = torch.empty(M, N, device=dev)
        for _ in range(5): ops.linear(a, w, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ops.linear(a, w, out=out)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 30 * 1e6
        xs.append(K); ys.append(us)
        print(f"N={N} K={K}: {us:.1f} us {2.0*M*N*K/us/1e6:.1f} TF")
    s, i = np.polyfit(xs, ys, 1)
    print(f"N={N}: slope {s*32:.3f} us per k-step (-> {2.0*M*N*32/(s*32)/1e6:.1f} TF asymptotic), intercept {i:.1f} us")
