#!/bin/bash
python -m pytest tests/test_stabilizer_gpu.py tests/test_long_run_gpu.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env $E python bench.py --no-cpu-baseline --no-pcie --quality-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], 'p50 %.3f p99 %.3f' % (d['latency_ms']['p50'], d['latency_ms']['p99']))
"; }
for i in 1 2; do
E="A=1" run --steps 2000 --warmup 200
E="LVK_HIP_DETECT_FIRST=0" run --steps 2000 --warmup 200
E="A=1" run --steps 2000 --warmup 200 --preset field
E="LVK_HIP_DETECT_FIRST=0" run --steps 2000 --warmup 200 --preset field
done
E="A=1" run --steps 20 --warmup 5
E="LVK_HIP_DETECT_FIRST=0" run --steps 20 --warmup 5
