#!/usr/bin/env python3
"""Evidence tool: per-shape duration of the encoder's big linears from a rocprofv3 kernel trace of tools/exp_ragged_only.py.
usage: sk_by_shape.py <kernel_trace.csv> [launches_per_forward=45]"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_nt_f32_persistent" in r["Kernel_Name"] or "gemm_nt_f32_streamk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 45
names = ["qkv N2304 K768", "o N768 K768", "wi N3072 K768", "wo N768 K3072"]
acc = collections.defaultdict(list)
for i, r in enumerate(rows):
    j = i % per
    if i < 3 * per:
        continue
    kind = "streamk" if "streamk" in r["Kernel_Name"] else "whole"
    acc[(names[j % 4], kind)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:18s} {k[1]:8s} n={len(v):4d} avg {sum(v)/len(v):7.1f} us  min {min(v):7.1f}")
