#!/bin/bash
# SQ / LDS counters for the bench workload (own pass, no tracing flags beyond what rocprofv3 needs for PMC).
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_sq_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_ATOMIC_RETURN --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ('a','b'):
    for f in glob.glob('$OUT/'+sub+'/*/*_counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = 'full_bp' if 'full_bp_kernel' in r['Kernel_Name'] else 'sample' if 'sample_philox' in r['Kernel_Name'] else None
            if k: agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(agg.items()):
            print(f"{k:8s} {c:24s} {sum(v)/len(v):16.0f}")
PY
