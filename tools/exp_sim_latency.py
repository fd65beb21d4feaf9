#!/usr/bin/env python3
"""Latency-mode similarity + top-100 (fp32, 320k x 768) at B = 1, 8, 16, 32: ms per ops.sim_topk call, and the result checked
against a second call with the direct-append filter epilogue (GDR_SIM_LOCAL_LIST=0 in a child process is the A/B)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
N, d, k = 320000, 768, 100
D = torch.from_numpy(synth.make_corpus(N, d)).to(dev)
for B in (1, 8, 16, 32):
    Qn, _ = synth.make_queries(D[:50000].cpu().numpy(), B)
    Q = torch.from_numpy(Qn).to(dev)
    ws = ops.Workspace(dev)
    for _ in range(5):
        r = ops.sim_topk(Q, D, k, workspace=ws, exact_on_overflow=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        r = ops.sim_topk(Q, D, k, workspace=ws, exact_on_overflow=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ms = sorted(ts)[len(ts) // 2] * 1e3
    vals, idx = r[0], r[1]
    print(f"B={B:2d}: {ms:.4f} ms  checksum idx {int(idx.sum())} val {float(vals.double().sum()):.6f}", flush=True)
