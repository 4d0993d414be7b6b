#!/bin/bash
# Round-closing validation on the GPU box (through gpurun): the whole GPU suite, smoke(), the driver's bench command, then the rocprofv3 passes of
# scripts/profile_gpu.sh (kernel stats + FETCH / WRITE / SQ counters, separate runs) for the homography and the vector-field preset, and the
# in-kernel timeline when the timeline variant of the library is present (scripts/variant_build.sh timeline -DLVK_TIMELINE).
# usage: bash scripts/gpu_round_check.sh <tag>      e.g. r06   -> gpurun_out/<tag>_check/ (summaries to copy into profiles/)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_check; mkdir -p $O/profiles
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err; echo "bench (driver style) rc=$?"
python bench.py > $O/bench_default.json 2>> $O/bench.err; echo "bench (defaults) rc=$?"
python - <<PY
import json
for name in ("bench_driver_style", "bench_default"):
    d = json.loads(open("$O/%s.json" % name).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(name, "value %.0f  ms/step %.4f  sustained %.0f  p50 %.4f p99 %.4f" % (d["value"], d["ms_per_step"], d["sustained"]["frames_per_s"], d["latency_ms"]["p50"], d["latency_ms"]["p99"]))
    print("  roofline: launch %.1f us  frac %.4f  binding_frac %.3f  standalone %.1f us;  cpu %s;  pcie %s" % (r["avg_launch_us"], r["frac"], r["binding_frac"], r["standalone_us"] or 0,
          (d.get("cpu_baseline") or {}).get("value"), (d.get("pcie_inclusive") or {}).get("value")))
    print("  K=4 %s  K=4 field %s  configs %s" % ((d.get("multi_stream") or {}).get("value"), (d.get("multi_stream_field") or {}).get("value"),
          [(c.get("value") and round(c["value"])) for c in (d.get("configs") or [])]))
    print("  reference kernel", {k: (d.get("reference_kernel") or {}).get(k) for k in ("avg_launch_us", "product_avg_launch_us", "speedup", "outputs_bit_equal")})
PY
rm -rf $R/gpurun_out/prof; bash $R/scripts/profile_gpu.sh > $O/profile_homography.log 2>&1
PROF_DST=$O/profiles python $R/scripts/summarize_prof.py ${TAG} > $O/summary_homography.txt 2>&1; head -16 $O/summary_homography.txt; cat $O/profiles/${TAG}_sq_counters_per_kernel.txt | grep "remap\|finalize"
rm -rf $R/gpurun_out/prof; BENCH_ARGS="--preset field" bash $R/scripts/profile_gpu.sh > $O/profile_field.log 2>&1
PROF_DST=$O/profiles BENCH_ARGS="--preset field" python $R/scripts/summarize_prof.py ${TAG}field > $O/summary_field.txt 2>&1; head -22 $O/summary_field.txt
rm -rf $R/gpurun_out/prof
# where the remap's issue slots go (SQ stall counters + effective clock of the shipped kernels): profiles/<tag>_remap_stalls.txt, remap_stalls.json
bash $R/scripts/pmc_stalls.sh > $O/stalls.log 2>&1
cp $R/gpurun_out/r06_stalls/summary.txt $O/profiles/${TAG}_remap_stalls.txt; cp $R/gpurun_out/r06_stalls/remap_stalls.json $O/profiles/remap_stalls.json
if [ -f $R/livevisionkit_amd/variants/liblvk_hip_timeline.so ]; then
  LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_timeline.so python $R/scripts/timeline_free.py > $O/profiles/${TAG}_timeline_free_running.txt 2>&1; tail -14 $O/profiles/${TAG}_timeline_free_running.txt
fi
