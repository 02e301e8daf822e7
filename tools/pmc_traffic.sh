#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE: two separate --pmc passes, --kernel-trace only) of one bench workload:
#   bash tools/pmc_traffic.sh r02_j c3 "Subleq-16, B=4096, 64 sims"   -> gpurun_out/<tag>_<wl>_pmc_traffic.json
set -e
TAG=${1:-r00}; WL=${2:-c3}; DESC=${3:-$WL}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_${WL}_pmc_$c -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_${WL}_pmc_$c.log 2>&1
  echo "pass $c done"
done
python3 $R/tools/pmc_summary.py $OUT/${TAG}_${WL}_pmc_traffic.json "$DESC" $OUT/${TAG}_${WL}_pmc_FETCH_SIZE $OUT/${TAG}_${WL}_pmc_WRITE_SIZE | grep -i "search_kernel\|selfplay_kernel\|resnet_forward" | cut -c1-300
