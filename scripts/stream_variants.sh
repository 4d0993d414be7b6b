#!/bin/bash
# A / B of library variants (livevisionkit_amd/variants/liblvk_hip_<name>.so, scripts/variant_build.sh) on the free-running 4K stream of scripts/stream_ab.py,
# two passes each, one box.  usage: bash scripts/stream_variants.sh name1 name2 ...   ("base" = the committed library); STREAM_ARGS="1080 1920" for other streams
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/stream_variants.txt
mkdir -p $R/gpurun_out
: > $OUT
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset LVK_HIP_LIB; else export LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_$v.so; fi
    echo "== $v (pass $round)" >> $OUT
    python $R/scripts/stream_ab.py $STREAM_ARGS 2>&1 | grep -v amdgpu.ids >> $OUT
  done
done
unset LVK_HIP_LIB
cat $OUT
