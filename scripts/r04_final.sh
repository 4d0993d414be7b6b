#!/bin/bash
# Round-4 closing validation: whole GPU suite, smoke, the driver's bench command.
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/final/pytest.log
tail -3 gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"
tail -2 gpurun_out/final/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d.get('sustained'), d.get('reference_kernel'))"
