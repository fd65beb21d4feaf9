#!/usr/bin/env bash
# Collects the rocprofv3 evidence behind bench.py's numbers into gpurun_out/$TAG (run on the GPU box through gpurun):
#   kernel-trace stats of the bench command, HBM-side traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass), MFMA-busy /
#   busy / clock counters, the bench line itself, and the kernel stats of the decode chain (tools/prof_generate.py).
# rocprofv3 gets the program directly after `--` (no env / shell hop), counters in passes of their own.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-recall --no-stages"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B --steps 5 --warmup 2 > "$OUT/stats.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $B --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $B --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_mfma" -- $B --steps 2 --warmup 1 > "$OUT/pmc_mfma.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$OUT/pmc_clk" -- $B --steps 2 --warmup 1 > "$OUT/pmc_clk.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_padded" -- $B --steps 5 --warmup 2 --encoder padded > "$OUT/stats_padded.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_padded" -- $B --steps 2 --warmup 1 --encoder padded > "$OUT/pmc_fetch_padded.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_padded" -- $B --steps 2 --warmup 1 --encoder padded > "$OUT/pmc_write_padded.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_generate" -- python3 $ROOT/tools/prof_generate.py > "$OUT/stats_generate.log" 2>&1
python3 $ROOT/tools/trace_steps.py $(ls $OUT/stats_generate/*/*kernel_trace.csv | head -1) 4 1-3 > "$OUT/generate_steps.txt" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_mfma_generate" -- python3 $ROOT/tools/prof_generate.py > "$OUT/pmc_mfma_generate.log" 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$OUT/pmc_clk_generate" -- python3 $ROOT/tools/prof_generate.py > "$OUT/pmc_clk_generate.log" 2>&1
cd "$ROOT"
python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
python3 bench.py --encoder padded --no-stages > "$OUT/bench_n1_padded.json" 2> "$OUT/bench_n1_padded.err"
python3 bench.py --dtype bf16 --no-cpu-baseline > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"
# keep what travels back small: the per-dispatch traces are large
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*.db" -delete
ls -la "$OUT"/* | head -60
