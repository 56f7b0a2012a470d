#!/bin/bash
# GPU box: A/B of engine-library builds (e.g. with experiment -D flags) at the stress shape on ONE box; boxes differ by
# several per cent, so only same-box comparisons mean anything.  usage: tools/ablate_rows.sh "<flags A>" "<flags B>" ...
set -u
i=0
libs=()
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-function $flags \
      sbayes_amd/csrc/sbe_engine.hip sbayes_amd/csrc/sbe_mixture.hip -o sbayes_amd/ab_$i.so 2>/dev/null || { echo "build failed: $flags"; exit 1; }
  echo "ab_$i.so = [$flags]"; libs+=("sbayes_amd/ab_$i.so"); i=$((i+1))
done
bash tools/ab.sh ${B:-64} "${libs[@]}"
rm -f sbayes_amd/ab_*.so
