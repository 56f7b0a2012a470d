#!/bin/bash
# Build the gfx950 engine library in-tree (also done by __graft_entry__.build()): the translation units are compiled
# in parallel and linked into sbayes_amd/libsbe_engine.so.  Extra arguments are passed to every compile (experiments).
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function"
mkdir -p build/obj
pids=""
for unit in sbe_engine_steps sbe_engine sbe_engine_resident sbe_engine_stateless sbe_mixture sbe_mixture_tuple sbe_mixture_rows sbe_mixture_mfma sbe_mixture_mfma_ws; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c sbayes_amd/csrc/$unit.hip -o build/obj/$unit.o &
  pids="$pids $!"
done
for pid in $pids; do wait $pid; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/sbe_engine.o build/obj/sbe_engine_steps.o build/obj/sbe_engine_resident.o \
    build/obj/sbe_engine_stateless.o build/obj/sbe_mixture.o build/obj/sbe_mixture_tuple.o build/obj/sbe_mixture_rows.o build/obj/sbe_mixture_mfma.o build/obj/sbe_mixture_mfma_ws.o \
    -o sbayes_amd/libsbe_engine.so
# the host layer's CPython extension (plain C, no device code): sbayes_amd/_fast.py uses it when present
# (optional: without it sbayes_amd/_fast.py takes the ctypes route to the same helpers)
gcc -O3 -fPIC -shared -Wall $(python3 -c "import sysconfig; print('-I' + sysconfig.get_paths()['include'])") sbayes_amd/csrc/sbe_pyhost.c \
    -o sbayes_amd/_sbe_pyhost$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))") \
    || echo "[build.sh] warning: sbayes_amd._sbe_pyhost not built; the host layer takes the ctypes / Python route"
