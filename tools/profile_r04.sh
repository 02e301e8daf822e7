#!/bin/bash
# Round-4 profile set on the GPU box (through gpurun): default bench line, rocprofv3 kernel stats of C2 / C3 / C4 / Subleq-256, HBM-side traffic
# (two --pmc passes each) of C2 and C3, SQ counters of the C2 and C3 kernels, in-kernel stamps of the one-launch kernels.  Every JSON that
# tools/pmc_summary.py writes carries the source hash of the tree (= eaz_build_hash() of the library that ran).
#   bash tools/profile_r04.sh r04_a [parts]   -> gpurun_out/<tag>_*; the summaries are then copied into profiles/
TAG=${1:-r04_a}; PARTS=${2:-all}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
want() { [ "$PARTS" = all ] || [[ ",$PARTS," == *",$1,"* ]]; }
if want bench; then timeout -k 10 900 python3 bench.py > $OUT/${TAG}_bench_default.json.log 2>&1; echo "default bench rc=$?"; fi
if want c2; then timeout -k 10 300 bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1; echo "c2 profile rc=$?"; fi
if want c3; then
  timeout -k 10 200 bash tools/profile_workload.sh $TAG c3 > /dev/null 2>&1; echo "c3 stats rc=$?"
  timeout -k 10 200 bash tools/pmc_traffic.sh $TAG c3 "Subleq-16, B=4096, 64 sims" > /dev/null 2>&1; echo "c3 traffic rc=$?"
  timeout -k 10 500 bash tools/pmc_sq.sh $TAG c3 > /dev/null 2>&1; echo "c3 sq rc=$?"
fi
if want c2sq; then timeout -k 10 500 bash tools/pmc_sq.sh $TAG c2 > /dev/null 2>&1; echo "c2 sq rc=$?"; fi
if want stamps; then
  timeout -k 10 200 python3 tools/sqp_stamps.py > $OUT/${TAG}_sqp_stamps.txt 2>&1; echo "sqp stamps rc=$?"
  ORDER=ascending timeout -k 10 200 python3 tools/sqp_stamps.py > $OUT/${TAG}_sqp_stamps_ascending.txt 2>&1; echo "sqp stamps (ascending) rc=$?"
  timeout -k 10 200 python3 tools/persist_stamps.py > $OUT/${TAG}_persist_stamps.txt 2>&1; echo "persist stamps rc=$?"
fi
if want c4; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c4_stats -- python3 $R/bench.py --workload c4 --selfplay-steps 32 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_c4_T32.json.log 2>&1; echo "c4 T=32 rc=$?"
  cp $OUT/${TAG}_c4_stats/*/*_kernel_stats.csv $OUT/${TAG}_c4_bench_kernel_stats.csv
  cd $R
fi
if want w256; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_w256_stats -- python3 $R/bench.py --workload c3 --word-size 256 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_bench_subleq256.json.log 2>&1; echo "w256 rc=$?"
  cp $OUT/${TAG}_w256_stats/*/*_kernel_stats.csv $OUT/${TAG}_subleq256_kernel_stats.csv
  cd $R
fi
if want w256t; then
  for t in 2 32; do timeout -k 10 300 python3 bench.py --workload c3 --word-size 256 --selfplay-steps $t --steps 2 --no-extra --no-cpu-baseline > $OUT/${TAG}_bench_subleq256_T$t.json.log 2>&1; echo "w256 T=$t rc=$?"; done
  timeout -k 10 300 bash tools/w256_sweep.sh > $OUT/${TAG}_subleq256_knob_sweep.txt 2>&1; echo "w256 sweep rc=$?"
fi
