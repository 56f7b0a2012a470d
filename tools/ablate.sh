#!/bin/bash
# Runs on the GPU box: rebuild the engine with one phase of k_mixture_tuple64 disabled at a time and
# time the default bench launch (diagnostic builds; results are wrong by construction).
set -u
mkdir -p gpurun_out
for abl in NONE SBE_ABL_NOLOG SBE_ABL_NOGATHER "SBE_ABL_NOLOG -DSBE_ABL_NOGATHER"; do
  if [ "$abl" = NONE ]; then ./build.sh; else ./build.sh -D$abl; fi
  echo "== $abl" >> gpurun_out/ablate.log
  python bench.py --no-cpu-baseline --no-secondary --steps 100 ${BENCH_ARGS:-} 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"]*1e3, d["roofline"]["kernel_avg_us"])' >> gpurun_out/ablate.log
done
./build.sh
cat gpurun_out/ablate.log
