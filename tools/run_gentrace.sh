# rocprofv3 kernel trace of the decode chain, summarised per (kernel, grid): B / BEAMS from the environment.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export B=${B:-64} BEAMS=${BEAMS:-10} CALLS=4
rm -rf /tmp/gp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -- python3 $R/tools/prof_generate.py > /dev/null 2>&1
python3 $R/tools/trace_steps.py $(ls /tmp/gp/*/*kernel_trace.csv | head -1) 4 $STEPS
if [ -n "$KEEP" ]; then cp $(ls /tmp/gp/*/*kernel_stats.csv | head -1) $R/gpurun_out/$KEEP; fi
