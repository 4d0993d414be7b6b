#!/bin/bash
# Round 4, third GPU visit: the -m gpu suite with the device-side suppression grid, then A/B against the host loop (LVK_HIP_HOST_GRID=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_3
rm -rf $OUT; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_stabilizer_gpu.py tests/test_long_run_gpu.py tests/test_config5_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
for i in 1 2; do
  for which in host dev; do
    if [ $which = host ]; then export LVK_HIP_HOST_GRID=1; else unset LVK_HIP_HOST_GRID; fi
    for preset in homography field; do
      python bench.py --steps 1500 --warmup 100 --pool 600 --preset $preset --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --quality-frames 0 2>$OUT/err_${which}_$preset.txt | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); s = j['stage_us']
print('$which $preset', round(j['value']), 'fps sustained', round(j['sustained']['frames_per_s']), ' p50', round(j['latency_ms']['p50'], 4), 'p99', round(j['latency_ms']['p99'], 4), ' '.join(f'{k}={v:.1f}' for k, v in s.items() if v), 'fr p90', round(j['free_running_ms']['p90'],4))"
    done
  done
done
unset LVK_HIP_HOST_GRID
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"
python - <<'PY'
import json, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
j=json.loads(open(f"{R}/gpurun_out/r04_3/bench_driver.json").read().strip().splitlines()[-1])
for k in ("value","sustained","latency_ms","stage_us","timed_region_ms","reference_kernel","configs","multi_stream"):
    print(k, json.dumps(j.get(k))[:600])
print("roofline", {k:j["roofline"][k] for k in ("frac","avg_launch_us","standalone_us","valu_instr_per_px")})
PY
