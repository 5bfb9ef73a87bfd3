#!/bin/bash
# Same-session A/B of library builds on the headline step and the time to converge (n = 8192): alternating processes.
# usage: bash tools/ab_quick.sh rounds name=path.so|default ...
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
rounds=$1; shift
B="--steps 20 --warmup 5 --no-cpu-baseline --no-c5 --no-secondary --no-plugin-boundary --no-live-pmc"
for r in $(seq $rounds); do
  for a in "$@"; do
    name=${a%%=*}; path=${a#*=}
    if [ "$path" = default ]; then unset CIPKKT_LIB; else export CIPKKT_LIB=$R/$path; fi
    python3 bench.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'KKT/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'noprof', round(d['profiling_cost']['ms_per_step_without'],4), 'converge_s', round(d['converge']['wall_s'],5), d['converge']['iters'], 'solve4x4', round(d['breakdown_ms']['solve4x4'],4))"
  done
done
