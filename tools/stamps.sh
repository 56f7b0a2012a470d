#!/bin/bash
set -u
./build.sh -DSBE_STAMPS
SBE_STAMPS_FILE=gpurun_out/stamps.bin python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > /dev/null 2> gpurun_out/stamps.err
python tools/stamps.py gpurun_out/stamps.bin | tee gpurun_out/stamps.txt
./build.sh
