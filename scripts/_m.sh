#!/bin/bash
python -m pytest tests/test_host_frames_gpu.py tests/test_tracker_ops_gpu.py -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" python scripts/host_feed_probe.py 300 2>&1 | grep -v amdgpu.ids | tail -1; }
run LOOKAHEAD=1
run LOOKAHEAD=0
run LOOKAHEAD=0 LVK_HIP_HOST_NO_MIRROR=1
for e in X=1 LVK_HIP_HOST_NO_MIRROR=1; do env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --quality-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pcie_inclusive']; print('$e', p.get('value'), p.get('latency_ms', {}).get('p50'), p.get('latency_ms', {}).get('p99'), p.get('error'))"; done
