#!/bin/bash
# vector-field preset (latency-bound chain): plain against every frame announced one push ahead
mkdir -p gpurun_out/la
for a in "" 1 "" 1; do
LVK_BENCH_ANNOUNCE=$a python bench.py --preset field --steps 1500 --warmup 50 --no-configs --no-multi-stream --no-pcie --no-reference-kernel --no-lookahead --no-cpu-baseline 2> gpurun_out/la/f.err > gpurun_out/la/f.json
python -c "
import json; d=json.loads(open('gpurun_out/la/f.json').read().strip().splitlines()[-1])
print('announce=$a', 'value', round(d['value']), 'sustained', round(d['sustained']['frames_per_s']), d['latency_ms'])"
done
