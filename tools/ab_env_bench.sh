#!/bin/bash
# Same-box A/B of one environment switch on the default bench command: tools/ab_env_bench.sh VAR valueA valueB [rounds] [extra bench args]
V=$1; A=$2; B=$3; R=${4:-3}; shift 4 2>/dev/null
for i in $(seq $R); do
  for val in $A $B; do
    env $V=$val python bench.py --no-secondary --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$V=$val', r['kernel'].split('<')[0], r['kernel_avg_us'], round(d['ms_per_step']*1000,2), round(d['value']), d['parity_timed_kernel_max_rel_err'], d['results_sha1'])"
  done
done
