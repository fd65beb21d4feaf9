#!/bin/bash
# A/B builds for the lab: tools/build_variant.sh <out.so> [extra hipcc flags...]  — compiles every csrc/*.hip with the extra flags
# (e.g. -DGDR_ATTN_NO_XCD_REMAP) into a scratch directory and links <out.so>; run a tool with GDR_HIP_LIB=<out.so> to use it.
set -e
out=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
for f in common gemm_f32 sim_topk layers encoder rerank decode bert sim_stream gemm_bf16 gemm_small; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc "$@" \
      -I"$here/gdr_amd/csrc" -c "$here/gdr_amd/csrc/$f.hip" -o "$tmp/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$tmp"/*.o
rm -rf "$tmp"
echo "built $out"
