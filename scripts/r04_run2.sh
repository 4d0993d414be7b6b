#!/bin/bash
# Round 4, second GPU visit: in-kernel timing of the RANSAC finalize, rocprofv3 stats / traffic / SQ counters of the bench (homography + field presets)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_2
rm -rf $OUT $R/gpurun_out/prof; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_rt.so python scripts/ransac_timing.py > $OUT/ransac_timing.txt 2>&1
tail -30 $OUT/ransac_timing.txt
bash scripts/profile_gpu.sh > $OUT/profile.log 2>&1
PROF_DST=$OUT/prof python scripts/summarize_prof.py r04 > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
grep -h "^{" $R/gpurun_out/prof/stats.log | tail -1 > $OUT/bench_under_rocprof.json
rm -rf $R/gpurun_out/prof
(cd /tmp && BENCH_ARGS="--preset field" && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- python $R/bench.py --steps 120 --warmup 10 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --preset field > $OUT/field_stats.log 2>&1)
BENCH_ARGS="--preset field" PROF_DST=$OUT/prof python scripts/summarize_prof.py r04field > $OUT/summary_field.txt 2>&1
head -30 $OUT/summary_field.txt
rm -rf $R/gpurun_out/prof
