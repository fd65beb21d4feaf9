#!/usr/bin/env python3
"""Decode-time linears with HOT weights (one buffer replayed: it stays in the XCDs' L2) against COLD weights (a rotation of
buffers larger than the Infinity Cache, as in the decode chain where 226 MB of decoder weights pass between two uses of a
matrix).  A HIP graph of the rotation is replayed so the host launch rate is out of the picture."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
shapes = [("o/q_c N=768 K=768", 768, 768), ("qkv N=2304 K=768", 2304, 768), ("wi N=3072 K=768", 3072, 768), ("wo N=768 K=3072", 768, 3072)]
M = int(os.environ.get("M", "640"))
for name, N, K in shapes:
    for label, total_mb in (("hot", 0), ("cold", 600)):
        nw = max(1, int(total_mb * 1e6 / (N * K * 4)))
        nw = min(nw, 256)
        Ws = [torch.randn(N, K, device=dev) * 0.03 for _ in range(nw)]
        A = torch.randn(M, K, device=dev)
        C = torch.empty(M, N, device=dev)
        reps = max(nw, 64)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(3):
                ops.linear(A, Ws[i % nw], out=C, splitk_ws=ws)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(reps):
                    ops.linear(A, Ws[i % nw], out=C, splitk_ws=ws)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
        print(f"M={M} {name:22s} {label:5s} buffers={nw:3d}  {us:7.2f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
        del Ws
