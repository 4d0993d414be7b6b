#!/bin/bash
run() { python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-pcie --quality-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], 'p50 %.3f p99 %.3f' % (d['latency_ms']['p50'], d['latency_ms']['p99']), {k: round(v,1) for k,v in d['stage_us'].items()})
"; }
for i in 1 2; do
echo default; run
echo side; LVK_HIP_INGEST_PLACEMENT=s run
done
echo side-trace; LVK_HIP_INGEST_PLACEMENT=s LVK_HIP_HOST_TRACE=1 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-pcie --quality-frames 0 2>&1 | grep -a -A12 "host trace" | head -14
echo field; run --preset field; echo field-side; LVK_HIP_INGEST_PLACEMENT=s run --preset field
