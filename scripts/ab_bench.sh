#!/bin/bash
# Same-box A/B of two builds of the library: bash scripts/ab_bench.sh <baseline .so> [bench args]; alternates base / new three times.
R=${GRAFT_REPO_ROOT:-/root/repo}
BASE=$1; shift
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export LVK_HIP_LIB=$R/$BASE; else unset LVK_HIP_LIB; fi
    python $R/bench.py --no-cpu-baseline --no-pcie --frames-per-step 1 --steps 800 "$@" 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); s = j['stage_us']
print('$which', round(j['value']), 'fps  p50', round(j['latency_ms']['p50'], 4), ' '.join(f'{k}={v:.1f}' for k, v in s.items() if v))"
  done
done
