#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_pf
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for preset in homography field; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$preset -- python $R/bench.py --steps 300 --warmup 20 --pool 128 --preset $preset --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --quality-frames 0 > $OUT/$preset.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/$preset/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("k_fast","k_pyrlk","k_ransac","k_area","k_pyr_f","k_match","k_mesh")):
            print("$preset", n[:60], r["Calls"], "avg", round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2), "max", round(float(r["MaxNs"])/1e3,2))
PY
done
