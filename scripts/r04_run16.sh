#!/bin/bash
# does the runtime's limit of hardware queues per process (GPU_MAX_HW_QUEUES, default 4) cost the stream anything?
mkdir -p gpurun_out/hwq
for i in 1 2; do for q in 4 8 2; do
GPU_MAX_HW_QUEUES=$q LVK_HIP_HOST_TRACE=1 python bench.py --steps 2000 --warmup 50 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead > gpurun_out/hwq/b.json 2> gpurun_out/hwq/b.err
python -c "
import json; d=json.loads(open('gpurun_out/hwq/b.json').read().strip().splitlines()[-1])
print('queues $q', round(d['value']), round(d['sustained']['frames_per_s']), d['latency_ms'])"
grep -E "remap kernel launch|conversion wait|lk sync|downscale\+pyramid" gpurun_out/hwq/b.err | tail -4
done; done
