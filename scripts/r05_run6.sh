#!/bin/bash
# Round 5, run 6: the output remap enqueued ahead of the chain's synchronisation behind a stream wait-value -- parity, then A/B against the build
# without it (free-running rate, synchronised latency, host trace), then the in-kernel timeline.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_run6; mkdir -p $O
timeout 1200 python -m pytest tests/test_stabilizer_gpu.py tests/test_config5_gpu.py tests/test_long_run_gpu.py tests/test_host_frames_gpu.py tests/test_schedule_fuzz_gpu.py tests/test_lifetime_gpu.py tests/test_facade_cpp.py tests/test_lens_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo "== A/B: base = remap launched after the smoother (-DLVK_NO_PRELAUNCH), new = pre-launched behind a wait-value"
bash $R/scripts/ab_bench.sh livevisionkit_amd/variants/liblvk_hip_noprelaunch.so --no-configs --no-multi-stream --no-reference-kernel --no-lookahead --quality-frames 0 2>&1 | tee $O/ab_prelaunch.txt
for which in base new; do
  if [ $which = base ]; then export LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_noprelaunch.so; else unset LVK_HIP_LIB; fi
  LVK_HIP_HOST_TRACE=1 python $R/bench.py --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead --quality-frames 0 --steps 2000 2>&1 | grep -A22 "lvk host trace" | head -24 | sed "s/^/$which: /" | tee -a $O/host_trace.txt
done
unset LVK_HIP_LIB
LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_timeline.so python scripts/timeline_free.py > $O/timeline.txt 2>&1; tail -22 $O/timeline.txt
