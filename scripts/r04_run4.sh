#!/bin/bash
# Round 4: full -m gpu suite, final rocprofv3 profiles (homography + field presets), the driver-style bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_4
rm -rf $OUT $R/gpurun_out/prof; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
bash scripts/profile_gpu.sh > $OUT/profile.log 2>&1
PROF_DST=$OUT/prof python scripts/summarize_prof.py r04 > $OUT/summary.txt 2>&1
head -24 $OUT/summary.txt
grep -h "^{" $R/gpurun_out/prof/stats.log | tail -1 > $OUT/bench_under_rocprof.json
rm -rf $R/gpurun_out/prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- python $R/bench.py --steps 120 --warmup 10 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --preset field > $OUT/field_stats.log 2>&1)
BENCH_ARGS="--preset field" PROF_DST=$OUT/prof python scripts/summarize_prof.py r04field > $OUT/summary_field.txt 2>&1
head -22 $OUT/summary_field.txt
rm -rf $R/gpurun_out/prof
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_2ranks_refused.json 2> $OUT/bench_2ranks_refused.err; echo "bench --gpus 2 on one GPU rc $? (expected non-zero)"; tail -2 $OUT/bench_2ranks_refused.err
python bench.py --streams-per-gpu 4 --steps 200 --warmup 20 --pool 128 > $OUT/bench_4streams.json 2> $OUT/bench_4streams.err; echo "bench 4 streams rc $?"
python - <<'PY'
import json, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
j=json.loads(open(f"{R}/gpurun_out/r04_4/bench_driver.json").read().strip().splitlines()[-1])
for k in ("value","sustained","latency_ms","stage_us","timed_region_ms","configs","multi_stream","cpu_baseline","quality"):
    print(k, json.dumps(j.get(k))[:900])
print("reference_kernel", {k:j["reference_kernel"].get(k) for k in ("avg_launch_us","product_avg_launch_us","speedup","outputs_bit_equal")})
print("roofline", {k:j["roofline"][k] for k in ("frac","avg_launch_us","standalone_us","valu_instr_per_px")})
print("pcie", j["pcie_inclusive"]["value"], j["pcie_inclusive"]["latency_ms"])
j=json.loads(open(f"{R}/gpurun_out/r04_4/bench_4streams.json").read().strip().splitlines()[-1])
print("4 streams: value", j["value"], "sustained", j["sustained"]["frames_per_s"], [ (round(x["p50"],3), round(x["p99"],3)) for x in j["ranks"][0]["stream_latency_ms"]])
PY
