#!/bin/bash
# GPU box: same-box A/B of an ENVIRONMENT switch on the sampler replay (engine microseconds per recorded MCMC step),
# alternating settings.  usage: tools/ab_env.sh VAR "v1 v2 ..." [tag ...]      e.g. tools/ab_env.sh SBE_FUSE_TABLES "0 1" headline
VAR=$1; VALS=$2; shift 2
TAGS=${@:-headline south_america headline_gibbs south_america_gibbs}
for rep in 1 2 3; do for v in $VALS; do
env $VAR=$v python tools/replay_bench.py $TAGS 2>/dev/null | python -c "
import json,sys
print('$VAR=$v', ' '.join('%s %.1f' % (d['tag'], d['gpu_us_per_step']) for d in map(json.loads, sys.stdin.read().strip().splitlines())))"
done; done
