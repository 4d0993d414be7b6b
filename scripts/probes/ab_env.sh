#!/bin/bash
# A/B of one environment switch on one box: bash scripts/probes/ab_env.sh VAR=value [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
KV=$1; shift
for i in 1 2 3; do for which in base new; do
  if [ $which = new ]; then export $KV; else unset ${KV%%=*}; fi
  python $R/bench.py --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead --quality-frames 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$which', round(j['value']), 'fps  sustained', round(j['sustained']['frames_per_s']), ' p50', round(j['latency_ms']['p50'], 4), 'p99', round(j['latency_ms']['p99'], 4), ' '.join('%s=%.1f' % (k, v) for k, v in j['stage_us'].items() if v), 'crc', j['tracking']['last_output_crc32'])"
done; done
