#!/bin/bash
# Kernel durations + SQ counters of the ScalingFilter micro-benchmark (scripts/bench_scaling.py), summarised per kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_scaling
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/scripts/bench_scaling.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.log 2>&1
python - <<'PY'
import csv, glob, collections, os, re
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
O=f"{R}/gpurun_out/pmc_scaling"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1","p2","p3","p4"):
    for f in glob.glob(f"{O}/{p}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m=re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
            if not m: continue
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{O}/summary.txt","w") as out:
    for f in glob.glob(f"{O}/stats/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "k_rcas" in r["Name"] or "k_easu" in r["Name"]:
                line=f'{r["Name"][:60]} calls={r["Calls"]} avg_ns={r["AverageNs"]} min_ns={r["MinNs"]} max_ns={r["MaxNs"]}'
                print(line); out.write(line+"\n")
    for k,d in sorted(acc.items()):
        line = k + " " + " ".join(f"{c}={sum(v)/len(v):.0f}" for c,v in sorted(d.items()))
        print(line); out.write(line+"\n")
PY
