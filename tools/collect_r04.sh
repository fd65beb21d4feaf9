#!/usr/bin/env bash
# Round-4 evidence (run on the GPU box through gpurun): clean single-stream kernel stats of the C5 and the large-batch C3 decode.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_depth1" -- python3 $ROOT/bench.py --workload c5 --depth 1 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c5_depth1.log" 2>&1
B=2048 timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gen_b2048" -- python3 $ROOT/tools/prof_generate.py > "$OUT/gen_b2048.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*.db" -delete
# MFMA-busy of the bf16 precision mode's kernels (the C2 step with bf16 linears / bf16-MFMA attention / bf16 similarity)
cd /tmp
B16="python3 $ROOT/bench.py --dtype bf16 --no-cpu-baseline --no-recall --no-stages --steps 2 --warmup 1"
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_mfma_bf16" -- $B16 > "$OUT/pmc_mfma_bf16.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_bf16" -- $B16 > "$OUT/stats_bf16.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*.db" -delete
