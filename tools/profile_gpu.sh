#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace + separate PMC passes of the default
# bench command (dominant kernel: every counter set; the other kernel forms: kernel trace + FETCH_SIZE),
# the stress shape, plus the FETCH_SIZE calibration.
# Outputs under gpurun_out/prof_$TAG/ ; tools/summarize_profiles.py turns them into profiles/.
set -u
TAG=${1:-r2}
ONLY=${2:-all}            # "headline": the headline passes only (the stress shape and the calibration are skipped)
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary"
for kern in packed packed_tuple packed_tuple_lds packed_general packed_v2 onehot onehot_general; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$kern -- $BENCH --kernel $kern > $OUT/bench_$kern.json 2> $OUT/bench_$kern.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
  echo "done $kern trace+fetch"
  if [ $kern = packed ]; then
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq1_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq3_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
    echo "done $kern write+sq"
  fi
  if [ $kern = packed_tuple ]; then      # the vector-pipe form the matrix-pipe form replaced (same-box comparison of the two)
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_$kern -- $BENCH --kernel $kern > /dev/null 2>&1
  fi
done
if [ "$ONLY" = headline ]; then
  find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
  for f in $(find $OUT -name '*_counter_collection.csv'); do (head -1 $f; grep -E 'k_mixture|read_dword' $f) > $f.tmp && mv $f.tmp $f; done
  ls $OUT; exit 0
fi
# stress shape (HBM/MALL streaming regime): rows kernel (default at B >= 2), the older general kernel, one-hot stream
for kern in packed packed_v2 onehot; do
  for B in 8 64; do
    SB="python3 bench.py --workload stress --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --kernel $kern"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_stress_${kern}_b$B -- $SB > $OUT/bench_stress_${kern}_b$B.json 2> $OUT/bench_stress_${kern}_b$B.err
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_stress_${kern}_b$B -- $SB > /dev/null 2>&1
    echo "done stress $kern B=$B"
  done
done
SB="python3 bench.py --workload stress --batch 64 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --kernel packed"
# (control: the rows kernel with the objects in their own order -- what ran before round 5)
SBE_ROWS_SORTED=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_stress_packed_unsorted_b64 -- $SB > $OUT/bench_stress_packed_unsorted_b64.json 2> $OUT/bench_stress_packed_unsorted_b64.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq1_stress_packed_b64 -- $SB > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_stress_packed_b64 -- $SB > /dev/null 2>&1
if [ -x tools/fetch_calib ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_calib -- ./tools/fetch_calib > $OUT/fetch_calib.log 2>&1
fi
# keep the merged-back volume small: the per-dispatch traces are not needed once stats exist
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
for f in $(find $OUT -name '*_counter_collection.csv'); do (head -1 $f; grep -E 'k_mixture|read_dword' $f) > $f.tmp && mv $f.tmp $f; done
ls $OUT
