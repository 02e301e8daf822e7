#!/bin/bash
# SQ counters (MFMA pipe busy, waits, LDS) of the dominant kernel of a workload, one rocprofv3 --pmc pass per counter:
#   bash tools/pmc_sq.sh r02_f c2|c3   -> gpurun_out/<tag>_<wl>_pmc_sq.json
set -e
TAG=${1:-r00}; WL=${2:-c2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
DIRS=""
for c in MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_${WL}_sq_$c -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_${WL}_sq_$c.log 2>&1 || echo "FAILED $c"
  DIRS="$DIRS $OUT/${TAG}_${WL}_sq_$c"
  echo "pass $c done"
done
python3 $R/tools/pmc_summary.py $OUT/${TAG}_${WL}_pmc_sq.json "$WL" $DIRS | grep -A12 "search_kernel\|selfplay_kernel\|search_persist" | head -40
