#!/bin/bash
python -m pytest tests/test_mesh_gpu.py tests/test_stabilizer_gpu.py tests/test_config5_gpu.py tests/test_golden.py -x -q 2>&1 | tail -4
python bench.py --preset field --steps 600 --warmup 50 --no-cpu-baseline --no-pcie --quality-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['sustained']['frames_per_s'], d['latency_ms'], d['stage_us'])"
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/field_tl3 -- python $R/bench.py --preset field --steps 200 --warmup 50 --no-cpu-baseline --no-pcie --quality-frames 0 > /dev/null 2>&1
