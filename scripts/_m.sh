#!/bin/bash
python -m pytest tests/test_mesh_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do python bench.py --preset field --steps 600 --warmup 50 --no-cpu-baseline --no-pcie --quality-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('field', d['value'], d['sustained']['frames_per_s'], d['latency_ms'], d['stage_us']['motion'])"; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/field_tl6 -- python $R/bench.py --preset field --steps 200 --warmup 50 --no-cpu-baseline --no-pcie --quality-frames 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/homog_tl -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-pcie --quality-frames 0 > /dev/null 2>&1
cd $R; f=$(find gpurun_out/field_tl6 -name "*kernel_stats.csv"); grep "k_nd" $f | cut -d, -f1-4 | cut -c1-120
