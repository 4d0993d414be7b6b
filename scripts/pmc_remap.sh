#!/bin/bash
# PMC pass over the remap micro-benchmark (SQ counters only; 8 slots per pass).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_remap
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/p1 -- python $R/scripts/bench_remap.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/p2 -- python $R/scripts/bench_remap.py > $OUT/p2.log 2>&1
python - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in ("p1","p2"):
    for f in glob.glob(f"{R}/gpurun_out/pmc_remap/{p}/*/*_counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_remap" not in k: continue
            name="homography" if "homography" in k else "mesh"
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name,d in acc.items():
            print(p,name,{c: round(sum(v)/len(v)) for c,v in d.items()})
PY
