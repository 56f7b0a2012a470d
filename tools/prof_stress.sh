#!/bin/bash
# GPU box: rocprofv3 kernel trace + two SQ counter passes of the stress-shape bench (dominant kernel only).
# usage: tools/prof_stress.sh TAG [BATCH] ; outputs gpurun_out/prof_stress_TAG/
set -u
TAG=${1:-x}; B=${2:-64}
OUT=gpurun_out/prof_stress_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="python3 bench.py --workload stress --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -- $BENCH > /dev/null 2>&1
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
for f in $(find $OUT -name '*_counter_collection.csv'); do (head -1 $f; grep -E 'k_mixture' $f) > $f.tmp && mv $f.tmp $f; done
python3 - <<PY
import csv, glob, collections
for part in ("pmc1","pmc2","pmc3"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%part):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(part, k, "mean %.4g n=%d"%(sum(v)/len(v), len(v)))
for f in glob.glob("$OUT/trace/*/*_kernel_stats.csv"):
    print(open(f).read()[:1500])
PY
