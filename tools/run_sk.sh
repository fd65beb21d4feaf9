cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 256 320 416 512 608 704 832 1024; do
for cfg in "0 0" "1 1"; do set -- $cfg
export GDR_GEMM_STREAMK_DP=$1 GDR_GEMM_STREAMK=$2 EXP_B=$B
rm -rf /tmp/skt; rocprofv3 --kernel-trace --output-format csv -d /tmp/skt -- python3 $R/tools/exp_ragged_only.py 2>/dev/null | grep "live rows"
echo "== B=$B DP=$1 TH=$2"; python3 $R/tools/sk_by_shape.py $(ls /tmp/skt/*/*kernel_trace.csv | head -1) 45
done; done
export EXP_PADDED=1 EXP_B=512
for cfg in "0 0" "1 1"; do set -- $cfg
export GDR_GEMM_STREAMK_DP=$1 GDR_GEMM_STREAMK=$2
rm -rf /tmp/skt; rocprofv3 --kernel-trace --output-format csv -d /tmp/skt -- python3 $R/tools/exp_ragged_only.py 2>/dev/null | grep "live rows"
echo "== PADDED B=512 DP=$1 TH=$2"; python3 $R/tools/sk_by_shape.py $(ls /tmp/skt/*/*kernel_trace.csv | head -1) 48
done
