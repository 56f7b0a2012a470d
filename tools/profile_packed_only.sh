#!/bin/bash
set -u
TAG=${1:-r1b}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary"
kern=packed
rm -rf $OUT/trace_$kern $OUT/pmc_fetch_$kern $OUT/pmc_write_$kern $OUT/pmc_sq1_$kern $OUT/pmc_sq2_$kern
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$kern -- $BENCH --kernel $kern > $OUT/bench_$kern.json 2> $OUT/bench_$kern.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq1_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
for f in $(find $OUT -name '*_counter_collection.csv'); do (head -1 $f; grep -E 'k_mixture|read_dword' $f) > $f.tmp && mv $f.tmp $f; done
ls $OUT
