"""Per-kernel stall picture from the counter passes of scripts/pmc_stalls.sh.  Usage: python scripts/summarize_stalls.py <out-dir>
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over
the 8 XCDs.  Prints, per remap kernel and leg: per-launch averages, the effective shader clock, and the three disjoint wave-time buckets
(ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY ~ WAVE_CYCLES) as fractions."""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]
XCDS, SIMDS = 8, 256 * 4


def short(name):
    m = re.search(r"(k_remap[a-z_0-9]*(?:<[a-z0-9, ]+>)?)", name)
    return m.group(1) if m else None


def load(leg):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in sorted(glob.glob(os.path.join(out, leg + "_p*"))):
        for f in glob.glob(os.path.join(p, "*", "*_counter_collection.csv")):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k:
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (r.get("Dispatch_Id"), k)
                if key not in seen and "Start_Timestamp" in r and r.get("Start_Timestamp"):
                    seen.add(key); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for f in glob.glob(os.path.join(p, "*", "*_kernel_trace.csv")):
            if p.endswith("_p1"):
                for r in csv.DictReader(open(f)):
                    k = short(r["Kernel_Name"])
                    if k:
                        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc, dur


for leg in ("alone", "live", "livefield"):
    acc, dur = load(leg)
    if not acc:
        continue
    print(f"==== {leg}")
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        n_launch = max(len(v) for v in acc[k].values())
        d_ns = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
        print(f"{k}: {n_launch} launches, mean duration under counter collection {d_ns / 1e3:.1f} us")
        print("   " + " ".join(f"{n}={v:.0f}" for n, v in sorted(c.items())))
        if "GRBM_GUI_ACTIVE" in c and d_ns == d_ns:
            print(f"   effective shader clock = GRBM_GUI_ACTIVE / {XCDS} XCDs / duration = {c['GRBM_GUI_ACTIVE'] / XCDS / d_ns:.3f} GHz")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            parts = [(n, c[n] / wc) for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if n in c]
            print("   of SQ_WAVE_CYCLES: " + ", ".join(f"{n} {v:.3f}" for n, v in parts) + f"  (sum {sum(v for _, v in parts):.3f})")
            if "SQ_ACTIVE_INST_VALU" in c:
                print(f"   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {c['SQ_ACTIVE_INST_VALU'] / wc:.3f}")
        if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
            # SQ_BUSY_CYCLES is per SQ (summed over them); ACTIVE_INST_VALU per wave-issue: the per-SIMD VALU duty is ACTIVE_INST_VALU (quad-cycles,
            # summed over waves) over the SIMD-time of the launch
            if "GRBM_GUI_ACTIVE" in c:
                simd_quads = c["GRBM_GUI_ACTIVE"] / XCDS / 4.0 * SIMDS
                print(f"   VALU busy: SQ_ACTIVE_INST_VALU / (GUI_ACTIVE/8/4 x {SIMDS} SIMDs) = {c['SQ_ACTIVE_INST_VALU'] / simd_quads:.3f}"
                      f";  all waves' time / SIMD time = mean {wc / simd_quads:.2f} waves per SIMD resident" if wc else "")
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / XCDS
            print(f"   VALU wave-instructions per SIMD-cycle = {c['SQ_INSTS_VALU'] / (cyc * SIMDS):.3f}  (1 per 4 cycles = 0.25 is the fast-op ceiling)")
