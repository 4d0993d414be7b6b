#!/bin/bash
# A / B of remap kernel variants (livevisionkit_amd/variants/liblvk_hip_<name>.so, scripts/variant_build.sh) on scripts/bench_remap.py: the kernels
# alone on the GPU, back-to-back launches.  usage: bash scripts/remap_variants.sh name1 name2 ...   ("base" = the committed library)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/remap_variants.txt
mkdir -p $R/gpurun_out
: > $OUT
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset LVK_HIP_LIB; else export LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_$v.so; fi
    echo "== $v (pass $round)" >> $OUT
    python $R/scripts/bench_remap.py >> $OUT 2>&1
  done
done
unset LVK_HIP_LIB
cat $OUT
