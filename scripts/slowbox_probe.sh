#!/bin/bash
# Some boxes of the pool run the tracker's small kernels ~1.8x slower (stage `motion` 76 instead of 43 us).  On such a box: record
# the in-kernel timeline and rocprofv3's kernel durations for comparison with a normal box.  No-op elsewhere.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/slowbox; mkdir -p $OUT
M=$(python $R/bench.py --frames-per-step 1 --steps 300 --warmup 100 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(int(j['stage_us']['motion']), int(j['value']))")
echo "motion_us fps: $M"
set -- $M
if [ "$1" -gt 60 ]; then
  echo "slow box: probing"
  LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_timeline.so python $R/scripts/timeline_free.py > $OUT/timeline.txt 2>&1
  tail -20 $OUT/timeline.txt
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --frames-per-step 1 --steps 120 --warmup 10 --no-cpu-baseline --no-pcie --no-overlap > $OUT/stats.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_" in r["Name"]: print(r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"])
PY
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^=\|^$" | head -20
  lscpu | head -20
fi
