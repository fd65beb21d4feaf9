#!/usr/bin/env python3
"""kernel_trace.csv -> per-burst GPU busy time vs wall span.  A burst = kernels separated by < GAP_US (default 2000).
Usage: python tools/trace_busy.py kernel_trace.csv"""
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2e6
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-8:]) > gap:
        bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
for b in bursts:
    if len(b) < 20: continue
    span = max(x[1] for x in b) - b[0][0]
    ev = sorted([(s, 1) for s, e, _ in b] + [(e, -1) for s, e, _ in b])
    busy, depth, last = 0, 0, None
    for t, d in ev:
        if depth > 0: busy += t - last
        depth += d; last = t
    per = collections.Counter()
    for s, e, n in b: per[n.split("(")[0][-40:]] += e - s
    top = ", ".join(f"{k}={v/1e3:.0f}us" for k, v in per.most_common(6))
    print(f"kernels={len(b)} span={span/1e6:.3f}ms busy={busy/1e6:.3f}ms ({100*busy/span:.0f}%)  top: {top}")
