#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/TAG/...) into the committed summaries under profiles/.
usage: summarize_profiles.py TAG [gpurun_out/TAG]

Writes  profiles/TAG_bench_n1.json, TAG_bench_n1_padded.json, TAG_bench_bf16.json      the bench lines
        profiles/TAG_bench_kernel_stats.csv, TAG_bench_padded_kernel_stats.csv,
        profiles/TAG_generate_kernel_stats.csv                                           rocprofv3 --stats summaries
        profiles/TAG_bench_pmc_hbm.md       HBM-side bytes per launch (FETCH_SIZE x2 / WRITE_SIZE, separate passes)
        profiles/TAG_mfma_busy.md           SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE of the GEMM kernels
        profiles/traffic.json               what bench.py reports as roofline.traffic
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/{tag}"


def one(pattern):
    m = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)   # the newest run (gpurun merges, never deletes)
    return m[-1] if m else None


def copy(pattern, dst):
    f = one(pattern)
    if f:
        shutil.copy(f, f"profiles/{dst}")
    return f


copy("bench_n1.json", f"{tag}_bench_n1.json")
copy("bench_n1_padded.json", f"{tag}_bench_n1_padded.json")
copy("bench_bf16.json", f"{tag}_bench_bf16.json")
copy("stats/*/*kernel_stats.csv", f"{tag}_bench_kernel_stats.csv")
copy("stats_padded/*/*kernel_stats.csv", f"{tag}_bench_padded_kernel_stats.csv")
copy("stats_generate/*/*kernel_stats.csv", f"{tag}_generate_kernel_stats.csv")


def counters(pattern):
    """{(kernel, grid): {counter: [values]}, 'dur': [ns]} per dispatch of a counter-collection CSV."""
    f = one(pattern)
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return d
    seen = set()
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"], r["Grid_Size"])
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if (r["Dispatch_Id"], "dur") not in seen:
            seen.add((r["Dispatch_Id"], "dur"))
            d[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d


def avg(x):
    return sum(x) / len(x) if x else 0.0


def short(name):
    return name.split("(")[0].replace("void ", "")


CMD = ("    cd /tmp && export TMPDIR=/tmp\n"
       "    B=\"python3 bench.py --no-cpu-baseline --no-recall --no-stages\"\n"
       "    rocprofv3 --kernel-trace --stats --output-format csv -d OUT/stats -- $B --steps 5 --warmup 2\n"
       "    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/pmc_fetch -- $B --steps 2 --warmup 1\n"
       "    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/pmc_write -- $B --steps 2 --warmup 1\n"
       "    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d OUT/pmc_mfma -- $B --steps 2 --warmup 1\n"
       "    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d OUT/pmc_clk -- $B --steps 2 --warmup 1\n"
       "    (the *_padded passes add `--encoder padded`; tools/collect_profiles.sh is the script)\n")

# ---------------------------------------------------------------------------------------------- HBM-side traffic
traffic = {}
with open(f"profiles/{tag}_bench_pmc_hbm.md", "w") as o:
    o.write(f"# {tag} — HBM-side traffic per launch (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes)\n\n"
            "Commands (each counter group in its own run, kernel-trace only, the program directly after `--`):\n\n" + CMD +
            "\nCorrection per MI355X_MICROARCH.md §HBM: FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on "
            "gfx950 -> bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 is exact for 16-B/lane stores.  FETCH counts L2 misses "
            "(Infinity-Cache hits included), so it bounds HBM reads from above.\n")
    for mode, fp, wp in (("ragged encoder (the default bench)", "pmc_fetch/*/*counter_collection.csv", "pmc_write/*/*counter_collection.csv"),
                         ("padded encoder (`--encoder padded`)", "pmc_fetch_padded/*/*counter_collection.csv",
                          "pmc_write_padded/*/*counter_collection.csv")):
        f, w = counters(fp), counters(wp)
        o.write(f"\n## {mode}\n\n| kernel | grid (threads) | launches | read MB/launch (FETCH x2) | write MB/launch |\n|---|---|---|---|---|\n")
        tot = collections.defaultdict(lambda: [0, 0.0])
        for k in sorted(f, key=lambda k: -sum(f[k]["FETCH_SIZE"])):
            n = len(f[k]["FETCH_SIZE"])
            rb = avg(f[k]["FETCH_SIZE"]) * 1024 * 2
            wb = avg(w[k]["WRITE_SIZE"]) * 1024 if k in w else 0.0
            if rb + wb < 1e5:
                continue
            o.write(f"| `{short(k[0])[:60]}` | {k[1]} | {n} | {rb / 1e6:.1f} | {wb / 1e6:.1f} |\n")
            tot[short(k[0])][0] += n
            tot[short(k[0])][1] += (rb + wb) * n
        lin = [(k, v) for k, v in tot.items() if "persistent" in k or "streamk" in k]
        if lin:
            n = sum(v[0] for _, v in lin)
            b = sum(v[1] for _, v in lin) / n
            key = "linear_gemm_bytes_per_launch_ragged" if "ragged" in mode else "linear_gemm_bytes_per_launch"
            traffic[key] = b
            o.write(f"\nThe linear GEMM (whole-tile + stream-K forms together, {n} launches): **{b / 1e6:.1f} MB per launch** "
                    "(read + write) — `roofline.traffic` of this mode.\n")
    o.write("\nAlgorithmic bytes of the linear GEMM, averaged over its four call shapes per layer (A + W + C (+ residual)): "
            "290 MB/launch padded (20 480 rows), 176 MB ragged (12 308 rows).  The excess is operand panels re-read through the "
            "Infinity Cache when an XCD's 4 MiB L2 cannot hold the panels its tiles touch.  In the whole-tile rounds the 64 "
            "workgroups of an XCD move through k in lockstep and share 16 panels (8 row x 8 column); in the stream-K tail (the "
            "last 1-2 rounds' worth of tiles of a launch, DESIGN.md §4) they sit at different k of different tiles and every "
            "tile fetches its two panels for itself — 8x the fetch per tile.  The tail made every launch faster (the kernel is "
            "MFMA-bound: 26.9 k q/s against 25.3 k, padded 17.1 k against 16.6 k), so this is energy, not time.\n")
traffic["source"] = f"profiles/{tag}_bench_pmc_hbm.md"
# the kernels these bytes were counted on: bench.py compares the hash with the source it runs and says "traffic_stale" when they differ
import hashlib
traffic["gemm_f32_sha16"] = hashlib.sha256(open("gdr_amd/csrc/gemm_f32.hip", "rb").read()).hexdigest()[:16]
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)

# ---------------------------------------------------------------------------------------------- MFMA utilisation
NORM = ("Per launch averages.  `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x busy cycles), busy cycles = SQ_BUSY_CYCLES / 32 "
        "(the counter sums the 32 shader engines' sequencers; it only advances while waves are resident).  Through round 3 the "
        "denominator was duration x GRBM_GUI_ACTIVE / 8 / duration; GRBM_GUI_ACTIVE also counts cycles OUTSIDE a short kernel's own "
        "span, which read as impossible clocks (2.9-5.2 GHz) for kernels under ~50 us and under-stated their MFMA share — the last "
        "column keeps those old figures for comparison.  `busy cycles / duration` must stay below the chip's 2.4 GHz; for launches "
        "of >= 300 us the two normalisations agree within ~5 % (the SQ is idle only in the ramp and the tail).  Durations are those "
        "of the profiled pass.\n\n")
m, c = counters("pmc_mfma/*/*counter_collection.csv"), counters("pmc_clk/*/*counter_collection.csv")
with open(f"profiles/{tag}_mfma_busy.md", "w") as o:
    o.write(f"# {tag} — MFMA utilisation of the GEMM kernels in the bench (rocprofv3 --pmc)\n\n"
            "Two counter passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-stages` (ragged encoder):\n\n" + CMD +
            "\n" + NORM +
            "| kernel | grid | launches | avg us | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE | busy cycles / duration (GHz) | MFMA busy | (old: GRBM clock, MFMA busy) |\n"
            "|---|---|---|---|---|---|---|---|---|---|\n")
    for k in sorted(m, key=lambda k: -sum(m[k]["dur_ns"])):
        if not any(s in k[0] for s in ("gemm_nt", "attention_mfma")):
            continue
        dur = avg(m[k]["dur_ns"])
        gui = avg(c[k]["GRBM_GUI_ACTIVE"]) if k in c else 0.0
        dur_c = avg(c[k]["dur_ns"]) if k in c else 0.0
        clk = gui / 8 / dur_c if dur_c else 0.0                      # cycles per ns = GHz
        busy = avg(m[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
        frac_old = busy / (1024 * dur * clk) if clk else 0.0
        sqc = avg(m[k]["SQ_BUSY_CYCLES"]) / 32.0                       # busy cycles of one shader engine (32 SQ instances are summed)
        frac = busy / (1024 * sqc) if sqc else 0.0
        o.write(f"| `{short(k[0])[:48]}` | {k[1]} | {len(m[k]['dur_ns'])} | {dur / 1e3:.1f} | {busy:.3e} | "
                f"{avg(m[k]['SQ_BUSY_CYCLES']):.3e} | {gui:.3e} | {sqc / dur:.2f} | {frac:.3f} | {clk:.2f}, {frac_old:.3f} |\n")
    o.write("\nReading: for the fp32 GEMMs MFMA busy x clock / 2.4 GHz is the fraction of the 157.3 TFLOP/s peak that the issue "
            "stream could deliver; what bench.py reports as `roofline.frac` is lower by the tile-edge waste (rows past the live "
            "count are computed and discarded) and by the epilogue / prologue phases in which no MFMA is issued.\n")
# ---------------------------------------------------------------------------------------------- the decode chain
copy("generate_steps.txt", f"{tag}_generate_steps.txt")
m, c = counters("pmc_mfma_generate/*/*counter_collection.csv"), counters("pmc_clk_generate/*/*counter_collection.csv")
if m:
    with open(f"profiles/{tag}_generate_mfma_busy.md", "w") as o:
        o.write(f"# {tag} — MFMA utilisation of the decode chain's GEMM kernels (rocprofv3 --pmc on tools/prof_generate.py: 4 x generate(), "
                "64 queries x 10 beams, prefix table)\n\nSame counters and formulae as the bench table (`{tag}_mfma_busy.md`): per launch "
                "averages over all four calls; `grid` = threads.  The 64x64-tile split-K kernel of the decode linears is listed by grid: "
                "122 880 threads = 480 workgroups (the N = 768 projections split 4 ways, wi un-split, wo split 4 ways), 92 160 = 360 (qkv / "
                "in_proj), 163 840 = 640 (adaptor lin1).\n\n"
                "\n\n".replace("{tag}", tag) + NORM +
                "| kernel | grid | launches | avg us | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | busy cycles / duration (GHz) | MFMA busy | (old: GRBM clock, MFMA busy) |\n|---|---|---|---|---|---|---|---|---|\n")
        for k in sorted(m, key=lambda k: -sum(m[k]["dur_ns"])):
            if "gemm_nt" not in k[0]:
                continue
            dur = avg(m[k]["dur_ns"])
            gui = avg(c[k]["GRBM_GUI_ACTIVE"]) if k in c else 0.0
            dur_c = avg(c[k]["dur_ns"]) if k in c else 0.0
            clk = gui / 8 / dur_c if dur_c else 0.0
            busy = avg(m[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
            frac_old = busy / (1024 * dur * clk) if clk else 0.0
            sqc = avg(m[k]["SQ_BUSY_CYCLES"]) / 32.0
            frac = busy / (1024 * sqc) if sqc else 0.0
            o.write(f"| `{short(k[0])[:48]}` | {k[1]} | {len(m[k]['dur_ns'])} | {dur / 1e3:.1f} | {busy:.3e} | {avg(m[k]['SQ_BUSY_CYCLES']):.3e} | "
                    f"{sqc / dur:.2f} | {frac:.3f} | {clk:.2f}, {frac_old:.3f} |\n")
    print(open(f"profiles/{tag}_generate_mfma_busy.md").read())
print(open(f"profiles/{tag}_mfma_busy.md").read())
print(open(f"profiles/{tag}_bench_pmc_hbm.md").read()[-3000:])

# ---------------------------------------------------------------------------------------------- the bf16 precision mode's kernels
copy("stats_bf16/*/*kernel_stats.csv", f"{tag}_bench_bf16_kernel_stats.csv")
m = counters("pmc_mfma_bf16/*/*counter_collection.csv")
if m:
    with open(f"profiles/{tag}_bf16_mfma_busy.md", "w") as o:
        o.write(f"# {tag} — MFMA utilisation of the bf16 precision mode's kernels (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES on "
                "`bench.py --dtype bf16 --no-cpu-baseline --no-recall --no-stages --steps 2 --warmup 1`)\n\n" + NORM +
                "bf16 MFMA peak: 2.5 PFLOP/s dense.  `grid` = threads.\n\n"
                "| kernel | grid | launches | avg us | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | busy cycles / duration (GHz) | MFMA busy |\n|---|---|---|---|---|---|---|---|\n")
        for k in sorted(m, key=lambda k: -sum(m[k]["dur_ns"])):
            if not any(s_ in k[0] for s_ in ("gemm_nt", "attention_mfma")):
                continue
            dur = avg(m[k]["dur_ns"])
            busy = avg(m[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
            sqc = avg(m[k]["SQ_BUSY_CYCLES"]) / 32.0
            o.write(f"| `{short(k[0])[:56]}` | {k[1]} | {len(m[k]['dur_ns'])} | {dur / 1e3:.1f} | {busy:.3e} | {avg(m[k]['SQ_BUSY_CYCLES']):.3e} | "
                    f"{sqc / dur:.2f} | {busy / (1024 * sqc) if sqc else 0.0:.3f} |\n")
    print(open(f"profiles/{tag}_bf16_mfma_busy.md").read())
