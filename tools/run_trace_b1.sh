cd /tmp && export TMPDIR=/tmp
export B=1 BEAMS=100 CALLS=4
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_b1 -- python3 $GRAFT_REPO_ROOT/tools/prof_generate.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(find gpurun_out/trace_b1 -name "*kernel_trace.csv" | head -1) 0.2
find gpurun_out/trace_b1 -name "*kernel_trace.csv" -size +8M -delete
