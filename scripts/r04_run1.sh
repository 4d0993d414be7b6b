#!/bin/bash
# Round 4, first GPU visit: the whole -m gpu suite, the remap variants (instructions per pixel + time), the driver-style bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_1
rm -rf $OUT; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for v in base a cur; do
  if [ $v = cur ]; then unset LVK_HIP_LIB; else export LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_$v.so; fi
  python scripts/bench_remap.py > $OUT/remap_$v.txt 2>&1
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_$v -- python $R/scripts/bench_remap.py > $OUT/pmc_$v.log 2>&1)
done
unset LVK_HIP_LIB
python - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for v in ("base","a","cur"):
    print(open(f"{R}/gpurun_out/r04_1/remap_{v}.txt").read().strip())
    for f in glob.glob(f"{R}/gpurun_out/r04_1/pmc_{v}/*/*_counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_remap" not in k: continue
            name=k.split("(")[0][-60:]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name,d in acc.items():
            m={c: sum(x)/len(x) for c,x in d.items()}
            print(v, name, {c: round(x) for c,x in m.items()}, "VALU/px", round(m.get("SQ_INSTS_VALU",0)*64/(3840*2160),1))
PY
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"
python - <<'PY'
import json, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
try:
    j=json.loads(open(f"{R}/gpurun_out/r04_1/bench_driver.json").read().strip().splitlines()[-1])
    for k in ("value","sustained","latency_ms","stage_us","roofline_secondary","reference_kernel","configs","multi_stream","cpu_baseline"):
        print(k, json.dumps(j.get(k))[:700])
    print("roofline", {k:j["roofline"][k] for k in ("frac","avg_launch_us","standalone_us")})
except Exception as e:
    print("bench parse failed", e); print(open(f"{R}/gpurun_out/r04_1/bench_driver.err").read()[-3000:])
PY
