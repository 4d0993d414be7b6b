#!/bin/bash
# Instruction-supply side of the remap kernels (SQ_IFETCH*, SQC_ICACHE_*) and the VALU opcode-class counters: is the 2.9-cycle VALU cadence
# (profiles/r06_remap_stalls.txt) an instruction-fetch problem?  Counter passes are their own runs (--pmc + --kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_ifetch
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ GRBM_GUI_ACTIVE"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU GRBM_GUI_ACTIVE"
ALONE="python $R/scripts/bench_remap.py"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- $ALONE > $OUT/p$i.log 2>&1
done
python - <<'PY' > $OUT/summary.txt 2>&1
import csv, glob, collections, os, re
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r06_ifetch"
for p in sorted(glob.glob(out + "/p*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_remap_\w+(<\w+>)?", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(0)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); n[k] += 1
    for k in acc:
        print(os.path.basename(p.rstrip("/")), k[:60], "launches", n[k], " ".join(f"{c}={v / n[k]:.0f}" for c, v in sorted(acc[k].items())))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
