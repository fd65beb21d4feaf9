for pr in 0 1; do
echo "== PRIO=$pr"; GDR_GEMM_STREAMK_PRIO=$pr python tools/exp_streamk.py 2>&1 | tail -3
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pr in 0 1; do
export GDR_GEMM_STREAMK_PRIO=$pr EXP_B=512
rm -rf /tmp/skt; rocprofv3 --kernel-trace --output-format csv -d /tmp/skt -- python3 $R/tools/exp_ragged_only.py 2>/dev/null | grep "live rows"
echo "== B=512 PRIO=$pr"; python3 $R/tools/sk_by_shape.py $(ls /tmp/skt/*/*kernel_trace.csv | head -1) 45
done
