#!/bin/bash
# The bench over its configurations on one box (boxes of the pool differ by +-10 %: compare within one run): one summary line each.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { name=$1; shift; python bench.py "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/bench_{n}.json"))
    r = d["roofline"]; q = d.get("quality") or {}; c = d.get("cpu_baseline") or {}; p = d.get("pcie_inclusive") or {}
    print(f"{n:14s} {d['value']:8.0f} frames/s  p50 {d['latency_ms']['p50']:.3f} p99 {d['latency_ms']['p99']:.3f} ms  remap {r['avg_launch_us']:.1f} us frac {r['frac']:.4f} valu_spec {r['valu_frac_spec']:.3f}"
          + (f"  standalone {r['standalone_us']:.1f} us" if r.get('standalone_us') else "")
          + (f"  cpu {c['value']:.1f} fps/{c['cores']}thr" if c else "") + (f"  pcie {p['value']:.0f} fps p99 {p['latency_ms']['p99']:.2f} ms" if p and 'value' in p else "")
          + (f"  psnr gpu {q['psnr_gpu']:.2f} oracle {q['psnr_oracle']:.2f} in {q['psnr_unstabilized']:.2f} equal {q['gpu_equals_oracle']}" if q and 'psnr_gpu' in q else ""))
except Exception as e:
    print(n, "FAILED", e)
PY
}
run default
run nv12 --format nv12 --no-cpu-baseline --no-pcie
run packed --format packed --no-cpu-baseline --no-pcie
run lensfused --lens fused --no-cpu-baseline --no-pcie
run field --preset field --no-cpu-baseline --no-pcie
run 1080p --rows 1080 --cols 1920 --no-cpu-baseline --no-pcie
run noov --no-overlap --no-cpu-baseline --no-pcie
