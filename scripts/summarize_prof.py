"""Condense rocprofv3 output (gpurun_out/prof/{stats,fetch,write}) into the tracked summaries under profiles/.
Usage: python scripts/summarize_prof.py <round-tag> [rows cols]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
rows, cols = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.environ.get("PROF_DST") or os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(?:<[a-z0-9, ]+>)?)", name)
    if m:
        return m.group(1)
    if "easu_remap" in name:                                   # the REFERENCE's kernel (oracle/_ref/fsr_yuv.hsaco), launched by bench.py's reference_kernel leg
        return "reference:" + name.split("(")[0].strip().replace(".kd", "")
    return None


# ---- kernel-trace --stats
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
lines = []
if stats:
    with open(stats[0]) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            if k or "copyBuffer" in r["Name"]:
                lines.append((k or "__amd_rocclr_copyBuffer", int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                              float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --frames-per-step 1 --steps 120 --warmup 10 --no-cpu-baseline --no-pcie %s (MI355X, %dx%d)\n" % (os.environ.get("BENCH_ARGS", ""), cols, rows))
        f.write("kernel,calls,avg_us,min_us,max_us,total_ms\n")
        for l in sorted(lines, key=lambda x: -x[5]):
            f.write("%s,%d,%.2f,%.2f,%.2f,%.3f\n" % l)

# ---- PMC passes (one counter per run)
def pmc(kind, counter):
    files = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))
    acc = {}
    if not files:
        return acc
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            if not k:
                continue
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
sq = {c: pmc("sq", c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVES")}
if any(sq.values()):
    with open(os.path.join(dst, f"{tag}_sq_counters_per_kernel.txt"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES (one run), wave-instructions per launch (averages)\n")
        for k in sorted(set().union(*[set(v) for v in sq.values()])):
            f.write(k + " " + " ".join("%s=%.0f" % (c, v[k][1] / v[k][0]) for c, v in sq.items() if k in v) + "\n")
if fetch or write:
    with open(os.path.join(dst, f"{tag}_pmc_traffic.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), per-launch averages. Units: KiB as reported;\n")
        f.write("# MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> fetch_x2 column.\n")
        f.write("kernel,launches,fetch_KiB,fetch_x2_MB,write_KiB,write_MB\n")
        for k in sorted(set(fetch) | set(write)):
            fn, fv = fetch.get(k, [0, 0.0]); wn, wv = write.get(k, [0, 0.0])
            fk = fv / fn if fn else 0.0; wk = wv / wn if wn else 0.0
            f.write("%s,%d,%.1f,%.3f,%.1f,%.3f\n" % (k, max(fn, wn), fk, 2 * fk * 1024 / 1e6, wk, wk * 1024 / 1e6))
    key = max((k for k in fetch if k.startswith("k_remap")), key=lambda k: fetch[k][0], default=None)      # the remap the pipeline runs
    if key:
        fk = fetch[key][1] / fetch[key][0]; wk = write.get(key, [1, 0.0]); wk = wk[1] / max(1, wk[0])
        valu = sq["SQ_INSTS_VALU"].get(key)
        # the packed-output kernel of the same family (bench.py's standalone leg / reference_kernel leg launch it in the same run)
        family = key.split("_420")[0].split("<")[0]
        packed = next((v for k, v in sq["SQ_INSTS_VALU"].items() if k.split("<")[0] == family), None)
        name = "remap_pmc_traffic_field.json" if "mesh" in key else "remap_pmc_traffic.json"
        json.dump({"rows": rows, "cols": cols, "kernel": key, "fetch_size_KiB": fk, "write_size_KiB": wk,
                   "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
                   "valu_per_px": (valu[1] / valu[0]) * 64.0 / (rows * cols) if valu else None,
                   "valu_per_px_packed": (packed[1] / packed[0]) * 64.0 / (rows * cols) if packed else None,
                   "round": tag,
                   "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B request); WRITE_SIZE as reported; valu_per_px = SQ_INSTS_VALU x 64 / "
                           "pixels of the pipeline's kernel, valu_per_px_packed = the same for the packed-output kernel of the family (standalone leg)"},
                  open(os.path.join(dst, name), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
p = os.path.join(dst, f"{tag}_pmc_traffic.csv")
if os.path.exists(p):
    print(open(p).read())
