#!/bin/bash
# GPU box: stall / instruction counters of the headline launch for one setting of an environment switch (separate --pmc passes)
#   tools/pmc_ws.sh TAG [VAR=value ...]
TAG=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
BENCH="python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/sq1 -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/sq3 -- $BENCH > /dev/null 2>&1
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
for f in $(find $OUT -name '*_counter_collection.csv'); do (head -1 $f; grep -E 'k_mixture' $f) > $f.tmp && mv $f.tmp $f; done
python3 - <<PY
import csv, glob, collections
for d in ("sq1","sq2","sq3"):
    for f in glob.glob("$OUT/%s/**/*_counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("$TAG", k, c, round(sum(v) / len(v)), len(v))
for f in glob.glob("$OUT/trace/**/*_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_mixture" in row["Name"]: print("$TAG", row["Name"][:50], "calls", row["Calls"], "avg_ns", row["AverageNs"])
PY
