#!/bin/bash
# Round 5, run 5: the whole GPU suite on the restructured library (split translation units, device guards everywhere, pruned experiments),
# smoke, the driver's bench command, and a free-running timeline of the chain.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_run5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "sustained", d["sustained"]["frames_per_s"], "lat", d["latency_ms"]["p50"], d["latency_ms"]["p99"])
print("roofline", {k: d["roofline"][k] for k in ("avg_launch_us","frac","binding_frac","standalone_us")})
print("cpu", d.get("cpu_baseline",{}).get("value"), "quality", d.get("quality"))
print("ms4", (d.get("multi_stream") or {}).get("value"), "ms4 field", (d.get("multi_stream_field") or {}).get("value"))
PY
LVK_HIP_LIB=$R/livevisionkit_amd/variants/liblvk_hip_timeline.so python scripts/timeline_free.py > $O/timeline.txt 2>&1; tail -30 $O/timeline.txt
