#!/bin/bash
# where the time goes with every frame announced: host trace + per-stage GPU times, against the plain loop
mkdir -p gpurun_out/la
for a in "" 1; do
echo "== LVK_BENCH_ANNOUNCE=$a"
LVK_BENCH_ANNOUNCE=$a LVK_HIP_HOST_TRACE=1 python bench.py --steps 1500 --warmup 50 --no-configs --no-multi-stream --no-pcie --no-reference-kernel --no-lookahead --no-cpu-baseline 2> gpurun_out/la/trace$a.err > gpurun_out/la/trace$a.json
grep -A12 "2[0-9][0-9][0-9] frames" gpurun_out/la/trace$a.err | head -14
python -c "
import json; d=json.loads(open('gpurun_out/la/trace$a.json').read().strip().splitlines()[-1])
print('value', d['value'], 'sustained', d['sustained']['frames_per_s']); print(d.get('stage_us')); print(d['roofline']['avg_launch_us'])"
done
