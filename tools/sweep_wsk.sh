# wave-split-K linear: the decode shapes under each (BK, stages, waves) form, one process per form (the override is read once)
cd $GRAFT_REPO_ROOT
for cfg in 32,3,4 32,2,4 16,2,4 16,3,4 16,4,4 16,2,8 16,3,8 32,2,8; do
  echo "== GDR_WSK_CFG=$cfg"
  GDR_WSK_CFG=$cfg MS=${MS:-640,100} python3 tools/bench_wsk.py 2>&1 | grep -v amdgpu.ids
done
