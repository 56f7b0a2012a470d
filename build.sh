#!/bin/bash
# Build the gfx950 engine library in-tree (also done by __graft_entry__.build()).
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 \
  -Wall -Wno-unused-function "$@" sbayes_amd/csrc/sbe_engine.hip -o sbayes_amd/libsbe_engine.so
