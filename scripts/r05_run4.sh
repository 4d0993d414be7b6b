#!/bin/bash
# Round 5, run 4: the remap's byte -> float table in LDS (parity, A/B against the build without it, rocprof kernel time + SQ counters), and
# K vector-field streams with more hardware queues.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_run4; mkdir -p $O
timeout 900 python -m pytest tests/test_remap_gpu.py tests/test_ref_pin_gpu.py tests/test_config5_gpu.py tests/test_lens_gpu.py tests/test_scaling_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== A/B: base = conversion + multiply per channel, new = LDS table"
bash $R/scripts/ab_bench.sh livevisionkit_amd/variants/liblvk_hip_nolut.so --no-configs --no-multi-stream --no-reference-kernel --no-lookahead --quality-frames 0 2>&1 | tee $O/ab_lut.txt
python $R/scripts/bench_remap.py 2>&1 | tail -12 | tee $O/bench_remap_new.txt
LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_nolut.so python $R/scripts/bench_remap.py 2>&1 | tail -12 | tee $O/bench_remap_base.txt
echo "== field preset, K streams, hardware queues"
for Q in 4 8 16; do for K in 4 8; do
  GPU_MAX_HW_QUEUES=$Q python $R/bench.py --preset field --streams-per-gpu $K --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('field K=$K GPU_MAX_HW_QUEUES=$Q', round(j['value']), 'frames/s, sustained', round(j['sustained']['frames_per_s']), 'p50/p99 ms', round(j['latency_ms']['p50'],3), round(j['latency_ms']['p99'],3))" | tee -a $O/field_k_queues.txt
done; done
