#!/bin/bash
# Round 5, run 7: the warp mesh in LDS for the mesh remap kernels -- parity, then A/B against the global-memory path (kernel alone, field preset
# single stream and 4 streams), and the kernel stats of 4 concurrent field streams.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_run7; mkdir -p $O
timeout 900 python -m pytest tests/test_remap_gpu.py tests/test_ref_pin_gpu.py tests/test_config5_gpu.py tests/test_lens_gpu.py tests/test_mesh_gpu.py "tests/test_stabilizer_gpu.py" -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for which in base new base new; do
  if [ $which = base ]; then export LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_nomeshlds.so; else unset LVK_HIP_LIB; fi
  python $R/scripts/bench_remap.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$which: /" | tee -a $O/bench_remap.txt
done
for which in base new base new; do
  if [ $which = base ]; then export LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_nomeshlds.so; else unset LVK_HIP_LIB; fi
  for K in 1 4; do
  python $R/bench.py --preset field --streams-per-gpu $K --steps 800 --warmup 100 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead --quality-frames 0 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$which field K=$K', round(j['value']), 'frames/s, sustained', round(j['sustained']['frames_per_s']), 'p50/p99 ms', round(j['latency_ms']['p50'],3), round(j['latency_ms']['p99'],3), 'remap us', round(j['roofline']['avg_launch_us'],1))" | tee -a $O/field_ab.txt
  done
done
unset LVK_HIP_LIB
P=$R/gpurun_out/prof; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python $R/bench.py --preset field --streams-per-gpu 4 --steps 600 --warmup 100 --pool 64 --no-cpu-baseline > $P/stats.log 2>&1
mkdir -p $O/profiles
PROF_DST=$O/profiles python $R/scripts/summarize_prof.py r05field_k4 2>&1 | head -30
tail -1 $P/stats.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('under rocprofv3: field K=4', round(j['value']), 'frames/s, sustained', round(j['sustained']['frames_per_s']), 'frames', j['sustained']['frames'])"
