#!/bin/bash
# GPU box: rocprofv3 average of one kernel over a command, alternating engine builds.  usage: tools/ab_kernel_avg.sh KERNEL "cmd" libA.so libB.so ...
export TMPDIR=/tmp
K=$1; CMD=$2; shift 2
for rep in 1 2; do for lib in "$@"; do
  rm -rf gpurun_out/prof_abk
  SBAYES_AMD_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_abk -- $CMD > gpurun_out/prof_abk.log 2>&1
  f=$(find gpurun_out/prof_abk -name "*kernel_stats.csv" | head -1)
  echo "$lib $(grep "$K" $f | cut -d, -f2,4,6-7) | $(grep -h "chains\|us/step" gpurun_out/prof_abk.log | cut -c1-64 | head -3 | tr '\n' ' ')"
done; done
rm -rf gpurun_out/prof_abk
