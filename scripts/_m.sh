#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-pcie --quality-frames 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], 'p50 %.3f p99 %.3f' % (d['latency_ms']['p50'], d['latency_ms']['p99']))
"; }
run A=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=1
run A=1
