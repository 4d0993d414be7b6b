#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + HBM traffic counters of the bench command.
# Counter passes are separate runs (one counter each), as MI355X_MICROARCH.md prescribes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --frames-per-step 1 --steps 120 --warmup 10 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-lookahead ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/stats.log
