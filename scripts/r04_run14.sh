#!/bin/bash
# where the 4:2:0 conversion goes for a free-running caller: behind the chain on the tracking stream (auto) or behind the previous remap on the bulk stream
mkdir -p gpurun_out/place
for i in 1 2 3; do for v in auto bulk; do
LVK_HIP_INGEST_PLACEMENT=$v python bench.py --steps 2000 --warmup 50 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead > gpurun_out/place/b.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/place/b.json').read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['sustained']['frames_per_s']), d['latency_ms'], {k: round(v,1) for k,v in d.get('stage_us',{}).items()})"
done; done
LVK_HIP_INGEST_PLACEMENT=bulk LVK_HIP_LIB=$PWD/livevisionkit_amd/variants/liblvk_hip_timeline.so python scripts/timeline_free.py 2>&1 | tail -16
