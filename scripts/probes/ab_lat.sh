#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_stabilizer_gpu.py tests/test_long_run_gpu.py::test_free_running_pushes_bit_exact tests/test_schedule_fuzz_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2 3; do for which in base new; do
  if [ $which = base ]; then export LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_base.so; else unset LVK_HIP_LIB; fi
  python $R/bench.py --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead --quality-frames 0 --steps 1000 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$which', round(j['value']), 'fps  sustained', round(j['sustained']['frames_per_s']), ' p50', round(j['latency_ms']['p50'], 4), 'p99', round(j['latency_ms']['p99'], 4), 'crc', j['tracking']['last_output_crc32'])"
done; done
