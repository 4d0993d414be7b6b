#!/bin/bash
# Round 5, run 3: k_ransac_finalize on 512 threads (parity, then A/B against the 256-thread build on the same box) and what K concurrent
# vector-field streams deliver on one GPU.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_run3; mkdir -p $O
timeout 900 python -m pytest tests/test_stabilizer_gpu.py tests/test_golden.py tests/test_usac_semantics.py tests/test_ransac_third_party.py tests/test_oracle_frozen.py tests/test_schedule_fuzz_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== A/B: base = finalize on 256 threads, new = 512 threads"
bash $R/scripts/ab_bench.sh livevisionkit_amd/variants/liblvk_hip_finalize256.so --no-configs --no-multi-stream --no-reference-kernel --no-lookahead --quality-frames 0 2>&1 | tee $O/ab_finalize.txt
echo "== field preset, K streams on one GPU"
for K in 1 2 4 6 8; do
  python $R/bench.py --preset field --streams-per-gpu $K --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('field K=$K', round(j['value']), 'frames/s, sustained', round(j['sustained']['frames_per_s']), 'p50/p99 ms', round(j['latency_ms']['p50'],3), round(j['latency_ms']['p99'],3))" | tee -a $O/field_k_sweep.txt
done
for K in 4 8; do
  python $R/bench.py --preset homography --streams-per-gpu $K --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('homography K=$K', round(j['value']), 'frames/s, sustained', round(j['sustained']['frames_per_s']), 'p50/p99 ms', round(j['latency_ms']['p50'],3), round(j['latency_ms']['p99'],3))" | tee -a $O/field_k_sweep.txt
done
