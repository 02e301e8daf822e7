#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the default bench run + two PMC passes (FETCH_SIZE, WRITE_SIZE).
# usage (through gpurun): bash tools/profile_round.sh r01_d     -> gpurun_out/<tag>_*; copy the summaries into profiles/
set -e
TAG=${1:-r00}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_pmc_$c.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc_traffic.json "DeepSea-32, B=4096, 64 sims" $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
cp $OUT/${TAG}_stats/*/*_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
head -12 $OUT/${TAG}_bench_kernel_stats.csv
