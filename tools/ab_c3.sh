# A/B of the decode chain settings on the C3 shapes (each setting in its own process: the switches are read once)
cd $GRAFT_REPO_ROOT
for env in "GDR_DECODE_WSK=0" "GDR_WSK_CFG=16,2,4" "GDR_WSK_CFG=16,3,4" "GDR_WSK_CFG=32,3,4" ; do
  echo "== $env"
  env $env SHAPES=${SHAPES:-64x10} python3 tools/bench_c3.py 2>&1 | grep -E "decode_ms|generate_ms|two_stage_ms\"" 
done
