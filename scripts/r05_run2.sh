#!/bin/bash
# Round 5, run 2: the remap with 32-bit tap addressing -- parity (three-way pins, config 5, fuzz), then kernel stats + SQ counters of the bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_remap_gpu.py tests/test_ref_pin_gpu.py tests/test_config5_gpu.py tests/test_lens_gpu.py tests/test_scaling_gpu.py tests/test_schedule_fuzz_gpu.py tests/test_bench_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "sustained", d["sustained"]["frames_per_s"], "lat", d["latency_ms"])
print("roofline", {k: d["roofline"][k] for k in ("avg_launch_us","frac","binding_frac","standalone_us")})
print("ms4", d["multi_stream"]["value"] if d.get("multi_stream") else None, "ms4 field", d.get("multi_stream_field"))
print("configs", [(c.get("workload","")[:30], c.get("value"), c.get("remap_us")) for c in d.get("configs") or []])
print("refk", d.get("reference_kernel"))
PY
P=$R/gpurun_out/prof; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 120 --warmup 10 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-lookahead --no-reference-kernel"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $CMD > $P/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- $CMD > $P/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- $CMD > $P/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $P/sq -- $CMD > $P/sq.log 2>&1
mkdir -p $O/profiles
PROF_DST=$O/profiles python $R/scripts/summarize_prof.py r05 2>&1 | head -60
cat $O/profiles/r05_sq_counters_per_kernel.txt
