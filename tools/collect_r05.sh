#!/usr/bin/env bash
# Round-5 evidence (run on the GPU box through gpurun): kernel stats of the headline command, of C5 at depth 1 and of generate() at 64 x 10.
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- python3 $ROOT/bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-recall --no-stages > "$OUT/bench.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_depth1" -- python3 $ROOT/bench.py --workload c5 --depth 1 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c5_depth1.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gen_64x10" -- python3 $ROOT/tools/prof_generate.py > "$OUT/gen_64x10.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
for d in bench c5_depth1 gen_64x10; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv"; done
ls -la "$OUT"
