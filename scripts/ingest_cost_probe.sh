#!/bin/bash
# What would a cheaper 4:2:0 -> 4:4:4 conversion buy?  Builds a variant of the library whose conversion skips the vertical pass (WRONG
# pixels, ~45 % fewer VALU instructions) and runs the free-running bench with both, alternating.  build: where hipcc is; run: where a GPU is.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/livevisionkit_amd/variants/liblvk_hip_ingest_cheap.so
if [ "$1" = "build" ]; then
  T=$(mktemp -d); mkdir -p $T/livevisionkit_amd $T/include $R/livevisionkit_amd/variants
  cp -r $R/livevisionkit_amd/csrc $T/livevisionkit_amd/; cp -r $R/include/* $T/include/; rm -f $T/livevisionkit_amd/csrc/*.o
  sed -i "s/^HIPFLAGS *=/HIPFLAGS = -DLVK_INGEST_CHEAP_PROBE /" $T/livevisionkit_amd/csrc/Makefile
  make -j8 -C $T/livevisionkit_amd/csrc > /dev/null 2>&1
  cp $T/livevisionkit_amd/liblvk_hip.so $V; rm -rf $T; echo built $V; exit 0
fi
cd $R; mkdir -p gpurun_out/probe
for i in 1 2 3; do for lib in "" $V; do
LVK_HIP_LIB=$lib python bench.py --steps 2000 --warmup 50 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead > gpurun_out/probe/ic.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/probe/ic.json').read().strip().splitlines()[-1])
print('cheap ' if '$lib' else 'normal', round(d['value']), round(d['sustained']['frames_per_s']), d.get('stage_us', {}).get('ingest'))"
done; done | tee gpurun_out/probe/ingest_cost.txt
