#!/bin/bash
# SQ counter passes over the whole bench (all kernels), summarised per kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --frames-per-step 1 --steps 100 --warmup 10 --no-cpu-baseline --no-pcie ${BENCH_ARGS---no-overlap}"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
python - <<'PY'
import csv, glob, collections, os, re
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1","p2"):
    for f in glob.glob(f"{R}/gpurun_out/pmc_bench/{p}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m=re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
            if not m: continue
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{R}/gpurun_out/pmc_bench/summary.txt","w") as out:
    for k,d in sorted(acc.items()):
        line = k + " " + " ".join(f"{c}={sum(v)/len(v):.0f}" for c,v in sorted(d.items()))
        print(line); out.write(line+"\n")
PY
