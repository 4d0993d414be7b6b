#!/bin/bash
mkdir -p gpurun_out/la
python -m pytest tests/test_stabilizer_gpu.py -m gpu -x -q -k "lookahead" 2>&1 | tail -3
for v in "LVK_HIP_LATE_POST=1" "LVK_X=1" "LVK_HIP_LATE_POST=1" "LVK_X=1"; do
echo "== $v (announce)"
env $v LVK_BENCH_ANNOUNCE=1 LVK_HIP_HOST_TRACE=1 python bench.py --steps 1500 --warmup 50 --no-configs --no-multi-stream --no-pcie --no-reference-kernel --no-lookahead --no-cpu-baseline 2> gpurun_out/la/t.err > gpurun_out/la/t.json
grep -A12 "2[0-9][0-9][0-9] frames" gpurun_out/la/t.err | head -14
python -c "
import json; d=json.loads(open('gpurun_out/la/t.json').read().strip().splitlines()[-1])
print('value', d['value'], 'sustained', d['sustained']['frames_per_s'], d['latency_ms'])"
done
