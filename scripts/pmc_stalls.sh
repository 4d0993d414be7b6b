#!/bin/bash
# Round 6: where the remap's issue slots go.  SQ stall / busy counters + GRBM_GUI_ACTIVE (effective clock) of the SHIPPED remap kernels,
#   alone  = scripts/bench_remap.py (back-to-back launches, nothing else on the GPU)
#   live   = bench.py's timed stream (tracker chain + remap; under counter collection rocprofv3 serialises the kernels, so "live" here
#            means "the launches the pipeline makes", not "concurrently with the tracker")
# Counter passes are their own runs (--pmc + --kernel-trace only), 8 SQ slots + 2 GRBM slots per pass (MI355X_MICROARCH.md, PMC slots).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_stalls
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
# second pass: whatever of these this rocprofv3 knows (an unknown name fails the whole pass)
P2=""
for c in SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU; do
  if grep -qw "$c" $OUT/counters_available.txt; then n=$(echo $P2 | wc -w); [ $n -lt 8 ] && P2="$P2 $c"; fi
done
P2="$P2 GRBM_GUI_ACTIVE"
echo "pass 2 counters: $P2" > $OUT/passes.txt
ALONE="python $R/scripts/bench_remap.py"
LIVE="python $R/bench.py --frames-per-step 1 --steps 120 --warmup 10 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-lookahead ${BENCH_ARGS:-}"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/alone_p1 -- $ALONE > $OUT/alone_p1.log 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/alone_p2 -- $ALONE > $OUT/alone_p2.log 2>&1
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/live_p1 -- $LIVE > $OUT/live_p1.log 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/live_p2 -- $LIVE > $OUT/live_p2.log 2>&1
# field preset live (k_remap_mesh_420)
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/livefield_p1 -- $LIVE --preset field > $OUT/livefield_p1.log 2>&1
# un-instrumented durations of the same commands, for the denominators
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone_stats -- $ALONE > $OUT/alone_stats.log 2>&1
python $R/scripts/summarize_stalls.py $OUT $OUT/remap_stalls.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the raw per-dispatch CSVs are large: keep the summary, the logs and the counter list
find $OUT -name "*.csv" -size +2M -delete
