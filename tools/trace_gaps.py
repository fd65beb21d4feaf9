#!/usr/bin/env python3
"""Per-queue timeline summary of a rocprofv3 kernel trace: busy time, gaps between consecutive kernels, and the kernels
ranked by total time — for the launch-bound decode chain.  usage: trace_gaps.py KERNEL_TRACE.csv [last_fraction]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]                     # the last call(s)
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
print(f"{len(rows)} kernels over {(t1 - t0) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    gaps = [g for g in gaps if g > 0]
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us, gaps {sum(gaps) / 1e3:.1f} us "
          f"(median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us)")
# union busy time of all queues
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in rows] + [(int(r["End_Timestamp"]), -1) for r in rows])
depth, last, busy = 0, None, 0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print(f"GPU busy (any queue): {busy / 1e3:.1f} us = {busy / (t1 - t0) * 100:.1f} % of the span")
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    tot[k][0] += 1
    tot[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {k[:60]:60s} {n:5d} x {t / n / 1e3:7.2f} us = {t / 1e3:8.1f} us")
