"""Per-kernel summary of the rocprofv3 --pmc passes tools/pmc_generate.sh collects (one counter group per pass directory):
share of the chip-busy cycles, vector-ALU issue fraction (SQ_INSTS_VALU x 4 cycles / 1 024 SIMDs / busy cycles), texture-addresser
busy fraction (TA_TA_BUSY_sum / 256 CUs / busy), VALU instructions per wave, HBM read rate (FETCH_SIZE KB x 2 on gfx950 for 16-byte
lanes, at a nominal 2.1 GHz) and the fraction of wave time parked at s_waitcnt / barriers."""
import collections, csv, glob, sys
root = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for p in sorted(glob.glob(f"{root}/p*/")):
    seen = set()
    for f in glob.glob(f"{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gdr::", "")[:58]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CYCLES" and (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"]))
                cnt[k] += 1
allbusy = sum(v["SQ_BUSY_CYCLES"] for v in tot.values()) or 1.0
rows = []
for k, v in tot.items():
    busy = v["SQ_BUSY_CYCLES"] / 32
    if busy <= 0 or only not in k:
        continue
    rows.append((busy, k, cnt[k], v["SQ_INSTS_VALU"] * 4 / 1024 / busy, v["TA_TA_BUSY_sum"] / 256 / busy,
                 v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), v["FETCH_SIZE"] * 2 * 1024 / (busy / 2.1e9) / 1e12,
                 v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), busy / max(cnt[k], 1)))
rows.sort(reverse=True)
print(f"{'kernel':58s} {'n':>4s} {'busy%':>6s} {'kcyc/launch':>11s} {'valu':>5s} {'ta':>5s} {'valu/wave':>9s} {'HBM TB/s':>8s} {'parked':>6s}")
for busy, k, n, vf, tf, vw, bw, w, per in rows[:40]:
    print(f"{k:58s} {n:4d} {100 * busy * 32 / allbusy:6.1f} {per / 1e3:11.1f} {vf:5.2f} {tf:5.2f} {vw:9.0f} {bw:8.2f} {w:6.2f}")
