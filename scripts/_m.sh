#!/bin/bash
for w in 5 5; do
  python bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], d['timed_region_ms'])
"
done
