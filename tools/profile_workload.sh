#!/bin/bash
# rocprofv3 kernel statistics of one bench workload (c3 / c4): bash tools/profile_workload.sh r01_f c3
set -e
TAG=${1:-r00}; WL=${2:-c3}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${WL}_stats -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_${WL}_stats.log 2>&1
cp $OUT/${TAG}_${WL}_stats/*/*_kernel_stats.csv $OUT/${TAG}_${WL}_bench_kernel_stats.csv
head -7 $OUT/${TAG}_${WL}_bench_kernel_stats.csv | cut -c1-70,140-260
