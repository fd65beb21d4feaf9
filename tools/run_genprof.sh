cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export B=${B:-512} BEAMS=${BEAMS:-10} CALLS=3
rm -rf /tmp/gp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -- python3 $R/tools/prof_generate.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/gp/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:22]: print(r['Name'][:84].ljust(84), r['Calls'].rjust(6), f"{int(r['TotalDurationNs'])/1e6:8.2f}ms", f"{float(r['AverageNs'])/1e3:8.1f}us", r['Percentage'])
print('total', tot/1e6)
PY
python3 $R/tools/trace_gaps.py $(ls /tmp/gp/*/*kernel_trace.csv | head -1) 2>/dev/null | head -8
