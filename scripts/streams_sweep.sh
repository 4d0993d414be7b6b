#!/bin/bash
# K streams on ONE GPU (bench.py --streams-per-gpu K): aggregate frames/s and per-stream latency, K = 1 .. 8.  Run on the GPU box.
mkdir -p gpurun_out/sweep
for K in 1 2 3 4 6 8; do
python bench.py --streams-per-gpu $K --frames-per-step 1 --steps 1000 --warmup 50 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead > gpurun_out/sweep/k$K.json 2> gpurun_out/sweep/k$K.err
python - <<P
import json
d = json.loads(open('gpurun_out/sweep/k$K.json').read().strip().splitlines()[-1])
lat = d.get('latency_ms', {})
print('K=$K value %.0f sustained %.0f per-frame us %.1f p50 %.3f p99 %.3f' % (d['value'], d['sustained']['frames_per_s'], 1e6 / d['sustained']['frames_per_s'], lat.get('p50', 0), lat.get('p99', 0)), 'remap us', round(d['roofline']['avg_launch_us'], 1))
P
done | tee gpurun_out/sweep/summary.txt
