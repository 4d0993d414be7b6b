#!/bin/bash
# Runs on the GPU box (through gpurun): kernel + memory-copy timeline of a short bench run (no counters).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --frames-per-step 1 --steps 40 --warmup 10 --no-cpu-baseline --no-pcie ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- $CMD > $OUT/log.txt 2>&1
find $OUT -name "*.csv" -size +0 | head
python - <<PY
import csv, glob, sys
ev = []
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:34], "q" + r.get("Queue_Id", "?")))
for f in glob.glob("$OUT/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "copy"))
ev.sort()
# ~4 steady-state frames from the middle of the free-running timed region (the tail of the run is the per-step synchronised latency pass)
rem = [i for i, e in enumerate(ev) if "k_remap" in e[2]]
mid = rem[len(rem) // 2 - 10]
tail = ev[mid:mid + 50]
t0 = tail[0][0]
with open("$OUT/frames.txt", "w") as o:
    for s, e, n, q in tail:
        o.write(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  {q:6s} {n}\n")
print(open("$OUT/frames.txt").read())
PY
