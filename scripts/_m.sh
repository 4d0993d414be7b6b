#!/bin/bash
python -m pytest tests/test_tracker_ops_gpu.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-pcie --quality-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], 'p50 %.3f p99 %.3f' % (d['latency_ms']['p50'], d['latency_ms']['p99']), {k: round(v,1) for k,v in d['stage_us'].items()})
"; }
run --rows 1440 --cols 2560 --pool 200
run --rows 720 --cols 1280 --pool 200
run --rows 1200 --cols 1920 --pool 200
