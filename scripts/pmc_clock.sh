#!/bin/bash
# Effective shader clock of the remap kernel under its own load: GRBM_GUI_ACTIVE (cycles) / kernel duration (MI355X_MICROARCH.md, DVFS).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_clock
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -- python $R/scripts/bench_remap.py > $OUT/p.log 2>&1
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dur = collections.defaultdict(list); act = collections.defaultdict(list)
for f in glob.glob(f"{R}/gpurun_out/pmc_clock/p/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_remap" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob(f"{R}/gpurun_out/pmc_clock/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_remap" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
with open(f"{R}/gpurun_out/pmc_clock/summary.txt", "w") as o:
    for k in dur:
        d = sum(dur[k]) / len(dur[k]); a = sum(act[k]) / max(1, len(act[k]))
        line = f"{k}: {len(dur[k])} launches, mean {d / 1e3:.1f} us, GRBM_GUI_ACTIVE {a:.0f} cycles -> effective clock {a / d:.3f} GHz (raw counter; divide by the number of XCDs if it is summed over them)"
        print(line); o.write(line + "\n")
PY
