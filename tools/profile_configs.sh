#!/bin/bash
# rocprofv3 kernel-trace stats of tools/config_runs.py (every kernel family on the BASELINE configurations).
# Run on the GPU box from the repo root; the summary CSV is copied to profiles/ by the caller.
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_cfg_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/config_runs.py > $OUT/config_runs.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/${TAG}_config_kernel_stats.csv
echo done
