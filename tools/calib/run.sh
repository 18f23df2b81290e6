#!/bin/bash
# On the GPU box (via gpurun), from the repo root: build the calibration program and read its FETCH_SIZE / WRITE_SIZE.
set -e
TAG=${1:-r02}
OUT=$(pwd)/gpurun_out/calib_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $OUT/calib tools/calib/calib.hip
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $OUT/calib > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $OUT/calib > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $OUT/calib > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_rdreq -- $OUT/calib > $OUT/rdreq.log 2>&1 || echo "rdreq counters unavailable"
find $OUT -name "*counter_collection.csv" | while read f; do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"][:40], r["Counter_Name"])] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print(k, v)
PY
done
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat
