#!/bin/bash
# GPU box: same-box A/B of two RECORDINGS of the sampler (engine microseconds per recorded MCMC step / engine calls per step):
# tests/golden/<tag>_before_calls.npz (e.g. `git show HEAD:tests/golden/headline_calls.npz > tests/golden/headline_before_calls.npz`
# before regenerating the logs after a host-layer change) against tests/golden/<tag>_calls.npz, alternating.
# usage: tools/ab_logs.sh [tag ...]
TAGS=${@:-cfg1 south_america headline}
PAIRS=""; for t in $TAGS; do PAIRS="$PAIRS ${t}_before $t"; done
for rep in 1 2 3; do
python tools/replay_bench.py $PAIRS 2>/dev/null | python -c "
import json,sys
print(' '.join('%s %.1f/%.1f' % (d['tag'], d['gpu_us_per_step'], d['calls_per_step']) for d in map(json.loads, sys.stdin.read().strip().splitlines())))"
done
