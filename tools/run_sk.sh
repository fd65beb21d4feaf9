#!/usr/bin/env bash
# Evidence behind gemm_f32.hip's streamk_wanted(): per-shape duration of the encoder's four linears, whole-tile persistent
# kernel (GDR_GEMM_STREAMK=0) against the stream-K-tail kernel on every launch (=1), over ragged batch sizes and the padded
# C2 batch.  Run on the GPU box (gpurun); the table goes to stdout (profiles/r02_streamk_sweep.txt is a formatted copy).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for B in 320 416 512 608 704 832 1024; do
for mode in 0 1; do
export GDR_GEMM_STREAMK=$mode GDR_GEMM_STREAMK_MID=0 EXP_B=$B
rm -rf /tmp/skt; rocprofv3 --kernel-trace --output-format csv -d /tmp/skt -- python3 $R/tools/exp_ragged_only.py 2>/dev/null | grep "live rows"
echo "== B=$B STREAMK=$mode"; python3 $R/tools/sk_by_shape.py $(ls /tmp/skt/*/*kernel_trace.csv | head -1) 45
done; done
export EXP_PADDED=1 EXP_B=512
for mode in 0 1; do
export GDR_GEMM_STREAMK=$mode
rm -rf /tmp/skt; rocprofv3 --kernel-trace --output-format csv -d /tmp/skt -- python3 $R/tools/exp_ragged_only.py 2>/dev/null | grep "live rows"
echo "== PADDED B=512 STREAMK=$mode"; python3 $R/tools/sk_by_shape.py $(ls /tmp/skt/*/*kernel_trace.csv | head -1) 48
done
