#!/bin/bash
mkdir -p gpurun_out/la
for a in "" 1; do
echo "== field preset, announce=$a"
LVK_BENCH_ANNOUNCE=$a LVK_HIP_HOST_TRACE=1 python bench.py --preset field --steps 1500 --warmup 50 --no-configs --no-multi-stream --no-pcie --no-reference-kernel --no-lookahead --no-cpu-baseline 2> gpurun_out/la/f$a.err > gpurun_out/la/f$a.json
grep -A12 "2[0-9][0-9][0-9] frames" gpurun_out/la/f$a.err | head -14
python -c "
import json; d=json.loads(open('gpurun_out/la/f$a.json').read().strip().splitlines()[-1])
print('value', round(d['value']), d.get('stage_us'))"
done
