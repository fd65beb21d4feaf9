# rocprofv3 kernel stats of the bf16-mode C2 bench step; output -> gpurun_out/bf16prof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/bf16prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bf16prof -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-stages > $R/gpurun_out/bf16prof.log 2>&1
f=$(ls $R/gpurun_out/bf16prof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
t=$(ls $R/gpurun_out/bf16prof/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
seq = [(r['Kernel_Name'].split('(')[0][:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp'])) for r in rows]
i = [j for j, s in enumerate(seq) if 'embed' in s[0]][-1]
prev_end = None
for s in seq[i:i + 14]:
    print(f"{s[0]:60s} {s[1]:8.1f} us")
PY
