#!/usr/bin/env python3
"""Latency-mode similarity (B <= 32) with and without the bf16 pre-filter: ms per ops.sim_topk call (perf_counter + synchronize, median
of 30), 320k x 768 fp32, top-100.  The pre-filter reads the bf16 image (491 MB) instead of the fp32 corpus (983 MB)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
N, d, k = 320000, 768, 100
Dn = synth.make_corpus(N, d)
D = torch.from_numpy(Dn).to(dev)
P = ops.PrefilteredCorpus(D)
for B in (1, 8, 32, 64):
    Qn, _ = synth.make_queries(Dn[:50000], B)
    Q = torch.from_numpy(Qn).to(dev)
    ws = ops.Workspace(dev)
    out = {}
    for name, corpus in (("fp32", D), ("prefilter", P)):
        for _ in range(5):
            r = ops.sim_topk(Q, corpus, k, workspace=ws, exact_on_overflow=False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            r = ops.sim_topk(Q, corpus, k, workspace=ws, exact_on_overflow=False)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[name] = (sorted(ts)[15] * 1e3, r)
    same = float((out["fp32"][1][1] == out["prefilter"][1][1]).all(dim=1).float().mean())
    print(f"B={B:2d}: fp32 {out['fp32'][0]:.4f} ms   prefilter {out['prefilter'][0]:.4f} ms   rows with identical ids {same:.3f}", flush=True)
