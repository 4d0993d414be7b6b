#!/bin/bash
python -m pytest tests/test_host_frames_gpu.py tests/test_tracker_ops_gpu.py tests/test_facade_cpp.py tests/test_lifetime_gpu.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" python scripts/host_feed_probe.py 400 2>&1 | grep -v amdgpu.ids | tail -${T:-1}; }
run LOOKAHEAD=1
run LOOKAHEAD=1 LVK_HIP_HOST_UP2=1
run LOOKAHEAD=0
run LOOKAHEAD=1
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --quality-frames 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f' % d['value'], 'sustained %.0f' % d['sustained']['frames_per_s'], 'p50 %.3f p99 %.3f' % (d['latency_ms']['p50'], d['latency_ms']['p99'])); p=d['pcie_inclusive']; print('pcie %.0f' % p['value'], p['latency_ms'])
"
export TMPDIR=/tmp; R=$PWD; cd /tmp; LOOKAHEAD=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/hf -- python $R/scripts/host_feed_probe.py 120 > /dev/null 2>&1; cd $R; python scripts/timeline_window.py /tmp/hf 800 40
