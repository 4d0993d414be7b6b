#!/bin/bash
# round-3 profiles: kernel stats + HBM traffic + SQ counters of the bench, default and vector-field preset; summaries only travel back
R=$PWD
mkdir -p gpurun_out/r03prof
bash scripts/profile_gpu.sh > gpurun_out/r03prof/default.log 2>&1
PROF_DST=$R/gpurun_out/r03prof python scripts/summarize_prof.py r03 > gpurun_out/r03prof/summ_default.log 2>&1
cp gpurun_out/r03prof/remap_pmc_traffic.json gpurun_out/r03prof/remap_pmc_traffic_default.json
rm -rf gpurun_out/prof
BENCH_ARGS="--preset field" bash scripts/profile_gpu.sh > gpurun_out/r03prof/field.log 2>&1
BENCH_ARGS="--preset field" PROF_DST=$R/gpurun_out/r03prof python scripts/summarize_prof.py r03field > gpurun_out/r03prof/summ_field.log 2>&1
rm -rf gpurun_out/prof
ls -la gpurun_out/r03prof
