#!/bin/bash
# AddressSanitizer + UBSan run of the host layer's CPython extension (CPU build only: GPU sanitizers are not available on the pool).
# Builds sbayes_amd/_sbe_pyhost with -fsanitize=address,undefined, runs the host-flow tests (incl. the real reference sampler where
# /root/reference exists) under it, and restores the release build.     tools/asan_pyhost.sh
set -e
cd "$(dirname "$0")/.."
EXT=sbayes_amd/_sbe_pyhost$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')
INC=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
restore() { gcc -O3 -fPIC -shared -Wall -I"$INC" sbayes_amd/csrc/sbe_pyhost.c -o "$EXT"; }
trap restore EXIT
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -Wall -I"$INC" sbayes_amd/csrc/sbe_pyhost.c -o "$EXT"
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
python3 -m pytest tests/test_native_host_flow_cpu.py tests/test_fast_host_cpu.py tests/test_host_logic_cpu.py tests/test_state_cpu.py \
    tests/test_patch_cpu.py tests/test_reference_sampler_cpu.py tests/test_overlap_cpu.py tests/test_dynamic_priors_cpu.py -x -q "$@"
