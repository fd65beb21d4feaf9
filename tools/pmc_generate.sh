# rocprofv3 --pmc passes over one generate() call (tools/prof_generate.py; B / BEAMS / DTYPE from the environment) or, with
# TARGET="bench.py --steps 2 ...", over any other python target, summarised
# per kernel by tools/summarize_pmc.py.  One counter group per pass, each under its own timeout: a counter set the hardware
# cannot collect in one pass makes rocprofv3 abort and then hang until it is killed (FETCH_SIZE alone takes 3 of the 4 TCC slots).
cd /tmp && export TMPDIR=/tmp
export B=${B:-512} BEAMS=${BEAMS:-30} CALLS=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-pmc_generate}
mkdir -p $O
cd $R
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p1 -- python3 ${TARGET:-tools/prof_generate.py} > $O/p1.log 2>&1; echo p1 rc=$?
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/p2 -- python3 ${TARGET:-tools/prof_generate.py} > $O/p2.log 2>&1; echo p2 rc=$?
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS TA_TA_BUSY_sum --output-format csv -d $O/p3 -- python3 ${TARGET:-tools/prof_generate.py} > $O/p3.log 2>&1; echo p3 rc=$?
python3 tools/summarize_pmc.py $O | tee $O/summary.txt
find $O -name "*kernel_trace.csv" -size +8M -delete
