#!/bin/bash
# Same-box A/B of an environment switch: bash scripts/ab_env.sh VAR=value [bench args]; alternates off / on three times.
R=${GRAFT_REPO_ROOT:-/root/repo}
KV=$1; shift
for i in 1 2 3; do
  for which in off on; do
    if [ $which = on ]; then export "$KV"; else unset "${KV%%=*}"; fi
    python $R/bench.py --no-cpu-baseline --no-pcie --quality-frames 0 --frames-per-step 1 --steps 800 "$@" 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); s = j['stage_us']
print('$which', round(j['value']), 'fps  p50', round(j['latency_ms']['p50'], 4), ' '.join(f'{k}={v:.1f}' for k, v in s.items() if v))"
  done
done
