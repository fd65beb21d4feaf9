#!/usr/bin/env python3
"""Decode-chain breakdown of a rocprofv3 kernel trace of tools/prof_generate.py: the LAST generate() call only, kernels grouped
by (name, grid size) — the grid tells the GEMM shapes apart — with count, mean duration, total, and per-queue busy / gap sums.
usage: trace_steps.py KERNEL_TRACE.csv [n_calls=4]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into calls at beam_init_kernel
starts = [i for i, r in enumerate(rows) if "beam_init_kernel" in r["Kernel_Name"]]
fins = [i for i, r in enumerate(rows) if "beam_finalize_kernel" in r["Kernel_Name"]]
lo, hi = starts[-1], fins[-1]
rs = rows[lo:hi + 1]
t0, t1 = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
print(f"last decode: {len(rs)} kernels over {(t1 - t0) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for r in rs:
    byq[r["Queue_Id"]].append(r)
for q, qs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in qs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(qs, qs[1:])]
    pos = [g for g in gaps if g > 0]
    print(f"queue {q}: {len(qs)} kernels, busy {busy / 1e3:.1f} us, gaps {sum(pos) / 1e3:.1f} us (median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f})")
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in rs] + [(int(r["End_Timestamp"]), -1) for r in rs])
depth, last, busy = 0, None, 0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print(f"GPU busy (any queue): {busy / 1e3:.1f} us = {busy / (t1 - t0) * 100:.1f} % of the span")
tot = collections.defaultdict(lambda: [0, 0, set()])
for r in rs:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gdr::", "")
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
    key = (k[:44], g)
    tot[key][0] += 1
    tot[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[key][2].add(r["Queue_Id"])
print(f"  {'kernel':44s} {'wgs':>6s} {'n':>5s} {'avg us':>8s} {'total us':>9s}  queue")
for (k, g), (n, t, qs) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k:44s} {g:6d} {n:5d} {t / n / 1e3:8.2f} {t / 1e3:9.1f}  {','.join(sorted(qs))}")
# ---- per decode step (delimited by beam_topk_kernel): wall span and busy time per queue
marks = [i for i, r in enumerate(rs) if "beam_topk_kernel" in r["Kernel_Name"]]
prev = 0
print("  step   span us   " + "   ".join(f"q{q} busy" for q in sorted(byq)))
for si, mi in enumerate(marks):
    seg = rs[prev:mi + 1]
    a, b = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    busy = {q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if r["Queue_Id"] == q) for q in sorted(byq)}
    print(f"  {si:4d} {(b - a) / 1e3:9.1f}   " + "   ".join(f"{busy[q] / 1e3:8.1f}" for q in sorted(byq)))
    prev = mi + 1
# ---- optional: kernel table restricted to a step range (argv[3] = "a-b")
if len(sys.argv) > 3:
    a_, b_ = [int(x) for x in sys.argv[3].split("-")]
    lo_i = marks[a_ - 1] + 1 if a_ > 0 else 0
    hi_i = marks[b_]
    sub = rs[lo_i:hi_i + 1]
    tot2 = collections.defaultdict(lambda: [0, 0, set()])
    for r in sub:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gdr::", "")
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
        key = (k[:44], g)
        tot2[key][0] += 1
        tot2[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot2[key][2].add(r["Queue_Id"])
    print(f"steps {a_}..{b_}:")
    for (k, g), (n, t, qs) in sorted(tot2.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {k:44s} {g:6d} {n:5d} {t / n / 1e3:8.2f} {t / 1e3:9.1f}  {','.join(sorted(qs))}")
