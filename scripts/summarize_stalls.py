"""Per-kernel stall picture from the counter passes of scripts/pmc_stalls.sh.
Usage: python scripts/summarize_stalls.py <out-dir> [json-out]      (text to stdout; json-out = profiles/remap_stalls.json, read by bench.py)

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles; GRBM_GUI_ACTIVE is summed over
the 8 XCDs.  Per remap kernel and leg: per-launch averages, the effective shader clock (GRBM_GUI_ACTIVE / 8 / duration), the three disjoint
wave-time buckets (ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY = WAVE_CYCLES), and the VALU issue slots used: a SIMD issues one wave64 fp32
instruction of the fast class per 2 cycles (profiles/r01_valu_issue_classes.txt: v_fma / v_add / v_mov 2.0 cycles, v_cvt / v_max / v_perm / v_cmp
4, v_rcp 8), so slots = cycles x 1024 SIMDs / 2 and used = SQ_INSTS_VALU."""
import collections
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
json_out = sys.argv[2] if len(sys.argv) > 2 else None
XCDS, SIMDS = 8, 256 * 4


def short(name):
    m = re.search(r"(k_remap[a-z_0-9]*(?:<[a-z0-9, ]+>)?)", name)
    return m.group(1) if m else None


def load(leg):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in sorted(glob.glob(os.path.join(out, leg + "_p*"))):
        for f in glob.glob(os.path.join(p, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if p.endswith("_p1"):
            for f in glob.glob(os.path.join(p, "*", "*_kernel_trace.csv")):
                for r in csv.DictReader(open(f)):
                    k = short(r["Kernel_Name"])
                    if k:
                        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc, dur


def plain_durations():
    """Durations of the same kernels WITHOUT counter collection (rocprofv3 --kernel-trace --stats of the alone command)."""
    res = {}
    for f in glob.glob(os.path.join(out, "alone_stats", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                res[k] = float(r["AverageNs"]) / 1e3
    return res


machine = {}
plain = plain_durations()
for leg in ("alone", "live", "livefield"):
    acc, dur = load(leg)
    if not acc:
        continue
    print(f"==== {leg}" + {"alone": "  (scripts/bench_remap.py: back-to-back launches, nothing else on the GPU)",
                          "live": "  (bench.py's stream, OBS 'homography' preset: the launches the pipeline makes; counter collection serialises kernels)",
                          "livefield": "  (bench.py --preset field)"}[leg])
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        n_launch = max(len(v) for v in acc[k].values())
        d_ns = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
        print(f"{k}: {n_launch} launches, mean duration under counter collection {d_ns / 1e3:.1f} us"
              + (f" (without counters, same command: {plain[k]:.1f} us)" if leg == "alone" and k in plain else ""))
        print("   " + " ".join(f"{n}={v:.0f}" for n, v in sorted(c.items())))
        rec = {"launches": n_launch, "duration_us_under_counters": d_ns / 1e3, "counters": {n: round(v) for n, v in c.items()}}
        cyc = None
        if "GRBM_GUI_ACTIVE" in c and d_ns == d_ns:
            cyc = c["GRBM_GUI_ACTIVE"] / XCDS
            rec["clock_mhz_under_kernel"] = cyc / d_ns * 1e3
            rec["cycles_per_launch"] = cyc
            print(f"   effective shader clock = GRBM_GUI_ACTIVE / {XCDS} XCDs / duration = {cyc / d_ns:.3f} GHz  ({cyc:.0f} cycles per launch)")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc and all(n in c for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")):
            a, wi, wa = c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc
            rec.update(active_inst_any_frac=a, wait_inst_any_frac=wi, wait_any_frac=wa)
            print(f"   of the waves' time (SQ_WAVE_CYCLES): issuing {a:.3f}, ready but not issued (SQ_WAIT_INST_ANY: issue arbitration / pipe) {wi:.3f}, "
                  f"parked on s_waitcnt (SQ_WAIT_ANY: memory) {wa:.3f}   (sum {a + wi + wa:.3f})")
            if cyc:
                simd_quads = cyc / 4.0 * SIMDS
                rec["waves_per_simd"] = wc / simd_quads
                print(f"   waves resident per SIMD (waves' time / SIMD time) {wc / simd_quads:.2f}; of them ready-but-not-issued at any time {wi * wc / simd_quads:.2f}, "
                      f"waiting for memory {wa * wc / simd_quads:.2f}")
        if "SQ_INSTS_VALU" in c and cyc:
            per_simd = c["SQ_INSTS_VALU"] / SIMDS
            rec["valu_cycles_per_instr"] = cyc / per_simd
            rec["valu_issue_slot_frac"] = c["SQ_INSTS_VALU"] * 2.0 / (cyc * SIMDS)
            print(f"   VALU: {per_simd:.0f} wave-instructions per SIMD in {cyc:.0f} cycles = one per {cyc / per_simd:.2f} cycles; full-rate issue is one per 2 "
                  f"-> {rec['valu_issue_slot_frac']:.3f} of the issue slots")
        ident = k.split("<")[0]
        machine.setdefault("live" if leg.startswith("live") else leg, {}).setdefault(ident, rec)

if json_out:
    machine["source"] = "scripts/pmc_stalls.sh on MI355X (rocprofv3 --pmc, SQ_* + GRBM_GUI_ACTIVE; per-launch averages); text: profiles/r06_remap_stalls.txt"
    json.dump(machine, open(json_out, "w"), indent=1)
