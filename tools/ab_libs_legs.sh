#!/bin/bash
# like tools/ab_libs.sh, with the hbm_regime leg: tools/ab_libs_legs.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for lib in $A $B; do
    SBAYES_AMD_LIB=$PWD/$lib python bench.py --no-secondary --no-cpu-baseline --steps 100 --legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; h=d['hbm_regime']
print('$lib', 'B=4096 kernel', r['kernel_avg_us'], 'us', round(d['value']/1e6,2), 'M/s | hbm_regime 8192:', h['kernel_avg_us'], 'us', round(h['evals_per_s']/1e6,2), 'M/s | parity', d['parity_timed_kernel_max_rel_err'], h['parity_max_rel_err'], d['results_sha1'])"
  done
done
