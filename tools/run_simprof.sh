cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 1 32; do
export B DT=f32
rm -rf /tmp/sp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 $R/tools/prof_sim.py > /dev/null 2>&1
echo "== B=$B"; python3 - <<PY
import csv,glob
f=glob.glob('/tmp/sp/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
python3 $R/tools/trace_gaps.py $(ls /tmp/sp/*/*kernel_trace.csv | head -1) 2>/dev/null | tail -12
done
