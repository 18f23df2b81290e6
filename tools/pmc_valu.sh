#!/bin/bash
# Run on the GPU box from the repo root: VALU / SALU / LDS instruction counts per wave for every kernel of
# tools/ab_v2.py (one rocprofv3 --pmc pass).  The sampler is VALU-issue bound: this is the number to watch.
OUT=$(pwd)/gpurun_out/pmc_valu
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/g -- python3 $(pwd)/tools/ab_v2.py 4096 3 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print(k)
    print("    per wave: " + "  ".join(f"{c[9:]}={sum(v)/len(v)/w:.0f}" for c, v in sorted(d.items()) if c != "SQ_WAVES") + f"  waves={w:.0f}")
PY
grep "trials/s" $OUT/run.log
