#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root.  Collects, for the bench workload:
#   1. rocprofv3 --kernel-trace --stats            → per-kernel time
#   2. rocprofv3 --pmc FETCH_SIZE  (own pass)       → HBM read bytes per dispatch  (×2 on gfx950, see guide)
#   3. rocprofv3 --pmc WRITE_SIZE  (own pass)       → HBM write bytes per dispatch
# Raw CSVs land in gpurun_out/prof_$TAG/; tools/summarize_prof.py condenses them into profiles/.
set -e
TAG=${1:-r03}
CFG=${2:-C2}            # bench.py --config; the other configs land in gpurun_out/prof_${TAG}_${CFG}/
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
[ "$CFG" != C2 ] && OUT=${OUT}_$CFG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --config $CFG"   # default steps/warmup: the same kernel work as the plain `python bench.py`
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/bench_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/bench_write.log 2>&1
echo "write done"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_rdreq -- python3 $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/bench_rdreq.log 2>&1 || echo "rdreq pass failed"
echo "rdreq done"
[ "${SQ:-1}" = 0 ] && { echo "sq skipped"; exit 0; }
# instruction mix / pipe occupancy of the two kernels (SQ counters are in units of 4 cycles per SIMD where they count time)
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_sq$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/bench_sq$i.log 2>&1 || echo "sq pass $i failed"
done
echo "sq done"
find $OUT -name "*.csv" | head -20
