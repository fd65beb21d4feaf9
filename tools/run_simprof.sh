# rocprofv3 kernel stats of the latency-mode similarity (B from the environment, default 32; fp32)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export B=${B:-32} DT=f32
rm -rf /tmp/sp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 $R/tools/prof_sim.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/sp/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:8.1f}us", f"min {int(r['MinNs'])/1e3:7.1f}")
PY
python3 $R/tools/trace_gaps.py $(ls /tmp/sp/*/*kernel_trace.csv | head -1) 0.2 2>/dev/null | head -6
