#!/bin/bash
run() { echo "== $*"; env "$@" python scripts/host_feed_probe.py 300 2>&1 | grep -v amdgpu.ids | tail -1; }
run LOOKAHEAD=1
run LOOKAHEAD=0
python -m pytest tests/test_host_frames_gpu.py tests/test_remap_gpu.py -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --quality-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['sustained']['frames_per_s']); p=d['pcie_inclusive']; print(p.get('value'), p.get('latency_ms'), p.get('error'))"
