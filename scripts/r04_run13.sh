#!/bin/bash
# completion ticket from the chain's last kernel (default) against the stream's completion signal (LVK_HIP_NO_TICKET=1)
mkdir -p gpurun_out/ticket
python -m pytest tests/test_stabilizer_gpu.py tests/test_host_frames_gpu.py tests/test_long_run_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do for v in LVK_HIP_NO_TICKET=1 LVK_X=1; do
env $v python bench.py --steps 2000 --warmup 50 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel > gpurun_out/ticket/b.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/ticket/b.json').read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['sustained']['frames_per_s']), 'lookahead', round(d['lookahead']['frames_per_s']), d['latency_ms'], d['lookahead']['latency_ms']['p50'])"
done; done
python bench.py --steps 20 --warmup 5 --no-configs --no-multi-stream --no-pcie --no-reference-kernel --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style', round(d['value']), d['timed_region_ms'])"
LVK_HIP_HOST_TRACE=1 python bench.py --steps 2000 --warmup 50 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead 2>&1 >/dev/null | grep -A20 "[23][0-9][0-9][0-9] frames" | head -16
