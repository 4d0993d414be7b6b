#!/bin/bash
run() { echo "== $*"; env "$@" python scripts/host_feed_probe.py 600 2>&1 | grep -v amdgpu.ids | tail -${T:-1}; }
run LOOKAHEAD=1
run LOOKAHEAD=1
run LOOKAHEAD=0
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
FUZZ_SEED=77 FUZZ_TRIALS=24 timeout 700 python scripts/fuzz_overlap.py 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], 'p50 %.3f p99 %.3f' % (d['latency_ms']['p50'], d['latency_ms']['p99'])); p=d['pcie_inclusive']; print('pcie %.0f' % p['value'], p['GBps_each_way'], p['latency_ms']['p50'], p['latency_ms']['p99']); print(d['cpu_baseline']['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
"
