#!/bin/bash
# rocprofv3 kernel stats of a short C3 (Subleq) bench run: bash tools/prof_c3.sh <tag>  -> gpurun_out/<tag>_c3_kernel_stats.csv
set -e
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c3 -- python3 $R/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_c3.log 2>&1
cp $OUT/${TAG}_c3/*/*_kernel_stats.csv $OUT/${TAG}_c3_kernel_stats.csv
cut -c1-150 $OUT/${TAG}_c3_kernel_stats.csv | sed -n 1,8p
grep '^{"metric"' $OUT/${TAG}_c3.log | cut -c1-200
