#!/bin/bash
# Run on the GPU box from the repo root: SQ counters (VALU / LDS / wait cycles) of the two throughput kernels,
# one rocprofv3 --pmc pass per group, on tools/ab_v2.py.  Output: gpurun_out/pmc_sq/*.csv + a per-kernel summary.
TAG=${1:-sq}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_WAVE32_LDS" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/ab_v2.py 4096 1 > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
  echo "group $i done"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        if "philox_v2" not in k and "full_bp_small" not in k: continue
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
print(open("$OUT/summary.txt").read())
PY
