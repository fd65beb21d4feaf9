#!/usr/bin/env python3
"""The decode-time linears in isolation (split-K scratch given, as gdr_t5_generate runs them): time per call, TFLOP/s and
the weight-stream rate, for the row counts of C3 (640 = 64 queries x 10 beams), infer.sh (100) and C5 (1920)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
shapes = [("self/cross o, q_c (N=768,K=768)", 768, 768), ("qkv (N=2304,K=768)", 2304, 768), ("wi (N=3072,K=768)", 3072, 768),
          ("wo (N=768,K=3072)", 768, 3072), ("adaptor lin1 (N=2048,K=768)", 2048, 768), ("adaptor lin2 (N=768,K=2048)", 768, 2048),
          ("head (N=23808,K=768)", 23808, 768)]
out = []
for M in [int(x) for x in os.environ.get("MS", "100,640,1920").split(",")]:
    for name, N, K in shapes:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.03
        C = torch.empty(M, N, device=dev)
        for _ in range(5):
            ops.linear(A, W, out=C, splitk_ws=ws)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            ops.linear(A, W, out=C, splitk_ws=ws)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        r = dict(M=M, shape=name, us=round(t * 1e6, 1), tflops=round(2.0 * M * N * K / t / 1e12, 1),
                 weight_gbs=round(N * K * 4 / t / 1e9))
        out.append(r)
        print(r)
