#!/usr/bin/env python3
"""Similarity + top-k alone: fp32 vs bf16 corpus, several batch sizes (incl. the HBM-bound latency-mode point B=32).
Prints ms, TFLOP/s, corpus-stream GB/s and the fraction of the binding roof (fp32 MFMA 157.3 TF / bf16 MFMA 2500 TF /
HBM 8000 GB/s)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
N, d, k = int(os.environ.get("N", 320000)), 768, 100
D = torch.from_numpy(synth.make_corpus(N, d)).to(dev)
Db = ops.to_bf16(D)
out = []
for B in (1, 8, 32, 128, 512, 4096):
    Qn, _ = synth.make_queries(D[:50000].cpu().numpy(), B)
    Q = torch.from_numpy(Qn).to(dev)
    variants = [("f32", D, 157.3, 0), ("bf16", Db, 2500.0, 0)]
    if B <= 32:
        variants.insert(1, ("f32-tiled", D, 157.3, 2))            # SIM_NO_STREAM: A/B of the latency-mode kernel
    for name, Dm, peak, fl in variants:
        ws = ops.Workspace(dev)
        for _ in range(3): ops.sim_topk(Q, Dm, k, workspace=ws, flags=fl, exact_on_overflow=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): ops.sim_topk(Q, Dm, k, workspace=ws, flags=fl, exact_on_overflow=False)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        tf = 2.0 * B * N * d / (ms * 1e-3) / 1e12
        gbs = N * d * Dm.element_size() / (ms * 1e-3) / 1e9
        out.append(dict(B=B, dtype=name, ms=round(ms, 3), tflops=round(tf, 1), frac_mfma=round(tf / peak, 3), corpus_gbs=round(gbs, 1),
                        frac_hbm=round(gbs / 8000, 3), qps=round(B / (ms * 1e-3))))
        print(out[-1])
