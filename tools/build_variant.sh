#!/bin/bash
# Builds a VARIANT of the engine library for same-box experiments: one unit recompiled with extra flags, the other objects taken
# from build/obj (run ./build.sh or __graft_entry__.build() first).   tools/build_variant.sh TAG UNIT [-DFLAG ...]
#   -> sbayes_amd/libsbe_var_TAG.so     (git-ignored; use with SBAYES_AMD_LIB=$PWD/sbayes_amd/libsbe_var_TAG.so)
set -e
cd "$(dirname "$0")/.."
TAG=$1; UNIT=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function"
mkdir -p build/obj_var
/opt/rocm/bin/hipcc $FLAGS "$@" -c sbayes_amd/csrc/$UNIT.hip -o build/obj_var/${UNIT}_$TAG.o
OBJS=""
for u in sbe_engine sbe_engine_steps sbe_engine_resident sbe_engine_stateless sbe_mixture sbe_mixture_tuple sbe_mixture_rows sbe_mixture_mfma sbe_mixture_mfma_ws; do
  if [ $u = $UNIT ]; then OBJS="$OBJS build/obj_var/${UNIT}_$TAG.o"; else OBJS="$OBJS build/obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o sbayes_amd/libsbe_var_$TAG.so
echo "built sbayes_amd/libsbe_var_$TAG.so"
