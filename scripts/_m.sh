#!/bin/bash
python -m pytest tests/test_stabilizer_gpu.py -m gpu -x -q -k "fused_downscale or overlap_yuv420_full or yuv420_in_out" 2>&1 | tail -5
run() { python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-pcie --quality-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], 'p50 %.3f p99 %.3f' % (d['latency_ms']['p50'], d['latency_ms']['p99']), {k: round(v,1) for k,v in d['stage_us'].items()})
"; }
for i in 1 2; do
echo fused; run
echo two-kernel; LVK_HIP_FUSE_AREA_PYRAMID=0 run
done
echo field-fused; run --preset field; echo field-two; LVK_HIP_FUSE_AREA_PYRAMID=0 run --preset field
echo driver-style; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --quality-frames 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], d['timed_region_ms'])
"
