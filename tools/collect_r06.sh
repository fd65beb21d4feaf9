#!/usr/bin/env bash
# Round-6 evidence beside tools/collect_profiles.sh (run on the GPU box through gpurun): the two-stage bench lines (C3, C5), the latency-mode
# similarity kernel stats (B = 32 and 1), the decode chain per (kernel, grid) with and without the fused sub-blocks.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --workload c5 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
python3 bench.py --workload c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
cd /tmp && export TMPDIR=/tmp
for b in 32 1; do
  rm -rf /tmp/sp
  B=$b DT=f32 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 $ROOT/tools/prof_sim.py > /dev/null 2>&1
  cp $(ls /tmp/sp/*/*kernel_stats.csv | head -1) "$OUT/sim_latency_B${b}_kernel_stats.csv"
done
rm -rf /tmp/sp
SPLIT=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 $ROOT/tools/prof_split_encoder.py > /dev/null 2>&1
cp $(ls /tmp/sp/*/*kernel_stats.csv | head -1) "$OUT/split_encoder_kernel_stats.csv"
cd "$ROOT"
ls -la "$OUT"
