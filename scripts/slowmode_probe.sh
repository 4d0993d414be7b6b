#!/bin/bash
# Some boxes of the pool run the free-running 4K stream at ~7 000 instead of ~9 000 frames/s with every kernel at its usual duration: the two streams
# settle into another phase relationship (round 6: remap launched 55 us later relative to the chain, the 4:2:0 conversion under it at 33 instead of
# 8.5 us).  On such a box: the in-kernel timeline, the host trace, the schedule counters, clocks and the host CPU.  No-op elsewhere.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/slowmode; mkdir -p $OUT
V=$(python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-lookahead --no-reference-kernel 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(int(j['value_sustained']))")
echo "sustained frames/s: $V" | tee $OUT/value.txt
if [ "$V" -lt 8000 ]; then
  echo "slow mode: probing"
  LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_timeline.so python $R/scripts/timeline_free.py > $OUT/timeline.txt 2>&1
  tail -24 $OUT/timeline.txt
  LVK_HIP_HOST_TRACE=1 python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-lookahead --no-reference-kernel > $OUT/bench_trace.json 2> $OUT/host_trace.txt
  tail -30 $OUT/host_trace.txt
  python -c "
import json
d=json.loads(open('$OUT/bench_trace.json').read().strip().splitlines()[-1])
print(d['value'], d.get('schedule'), d.get('stage_us'))"
  rocm-smi --showclocks --showpower 2>&1 | grep -v "^=\|^$" | head -12
  lscpu | grep -i "model name\|mhz\|socket\|numa" | head
  for k in LVK_HIP_INGEST_PLACEMENT=bulk LVK_HIP_INGEST_PLACEMENT=tracker; do
    echo "== $k"; env $k python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-lookahead --no-reference-kernel 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(int(j['value']), int(j['value_sustained']))"
  done
fi
