# run one command under several environment settings (one process each, same box): ab_env.sh "<cmd>" "A=1" "B=2 C=3" ...
cd $GRAFT_REPO_ROOT
cmd="$1"; shift
for e in "$@"; do
  echo "== $e"
  env $e $cmd 2>&1 | grep -v "Warning\|amdgpu.ids"
done
