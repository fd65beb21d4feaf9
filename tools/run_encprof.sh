# rocprofv3 kernel stats of the small-batch encoder pass (tools/prof_encoder_small.py); output -> gpurun_out/encprof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/prof_encoder_small.py 2>&1 | grep -v "Warning\|amdgpu.ids"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/encprof -- python3 $R/tools/prof_encoder_small.py > $R/gpurun_out/encprof.log 2>&1
f=$(ls $R/gpurun_out/encprof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
