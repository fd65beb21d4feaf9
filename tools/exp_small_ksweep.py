#!/usr/bin/env python3
"""time(K) of the 64x64-tile decode linear at M = 640 rows, un-split (no scratch given), HIP-graph replay: the slope is the
steady-state K-loop rate, the intercept is what a launch costs whatever K is (dispatch, prologue, epilogue, drain)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "640"))
for N in [int(x) for x in os.environ.get("NS", "3072,2304,1536,768").split(",")]:
    pts = []
    for K in [int(x) for x in os.environ.get("KS", "128,256,512,768,1024,1536,3072").split(",")]:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.03
        C = torch.empty(M, N, device=dev)
        s = torch.cuda.Stream()
        reps = 64
        with torch.cuda.stream(s):
            for i in range(3):
                ops.linear(A, W, out=C)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(reps):
                    ops.linear(A, W, out=C)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
        pts.append((K, us))
        print(f"M={M} N={N} K={K:5d} wgs={((M+63)//64)*((N+63)//64):4d} {us:7.2f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
    (k0, t0), (k1, t1) = pts[2], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print(f"  slope {slope * 32:.3f} us per K-step of 32 -> {2.0 * M * N * 32 / (slope * 32) / 1e6:.1f} TFLOP/s steady; intercept {t0 - slope * k0:.2f} us")
