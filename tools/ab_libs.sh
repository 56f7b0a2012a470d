#!/bin/bash
# Same-box A/B of engine builds on the default bench command: tools/ab_libs.sh libA.so libB.so [rounds] [extra bench args]
# (alternating runs; prints kernel time, step time, evals/s, parity of the timed results and their digest per run)
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq $R); do
  for lib in $A $B; do
    SBAYES_AMD_LIB=$PWD/$lib python bench.py --no-secondary --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', r['kernel'].split('<')[0], r['kernel_avg_us'], round(d['ms_per_step']*1000,2), round(d['value']), d['parity_timed_kernel_max_rel_err'], d['results_sha1'])"
  done
done
