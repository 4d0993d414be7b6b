#!/bin/bash
# device-frame look-ahead: parity tests, then the bench line with its `lookahead` block (twice: boxes differ)
mkdir -p gpurun_out/la
python -m pytest tests/test_stabilizer_gpu.py -m gpu -x -q -k "lookahead" > gpurun_out/la/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/la/pytest.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-configs --no-multi-stream --no-pcie --no-reference-kernel > gpurun_out/la/bench$i.json 2> gpurun_out/la/bench$i.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/la/bench$i.json').read().strip().splitlines()[-1])
print('value', d['value'], 'sustained', d['sustained']['frames_per_s'], 'p50', d['latency_ms']); print(d['lookahead'])"
done
LVK_HIP_HOST_TRACE=1 python bench.py --steps 1500 --warmup 50 --no-configs --no-multi-stream --no-pcie --no-reference-kernel --no-lookahead 2>&1 >/dev/null | grep -A16 "host trace" | head -20
