#!/bin/bash
# GPU box: A/B of engine library builds on the stress bench, alternating, same box.  usage: tools/ab.sh B libA.so libB.so ...
B=$1; shift
for rep in 1 2 3; do for lib in "$@"; do
SBAYES_AMD_LIB=$PWD/$lib python bench.py --workload ${WL:-stress} --batch $B --steps 30 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib B $B', d['roofline']['kernel'], d['roofline']['kernel_avg_us'], round(d['ms_per_step']*1000,1), d['roofline']['frac'], d['parity_timed_kernel_max_rel_err'])"
done; done
