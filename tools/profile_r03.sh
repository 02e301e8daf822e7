#!/bin/bash
# Round-3 profile set on the GPU box (through gpurun): default bench line, rocprofv3 kernel stats of C2 / C3 / C4, HBM-side traffic (two
# --pmc passes each) of C2 and C3, SQ counters of the C3 kernel, in-kernel stamps of the three one-launch kernels.
#   bash tools/profile_r03.sh r03_a    -> gpurun_out/<tag>_*; the summaries are then copied into profiles/
TAG=${1:-r03_a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout -k 10 600 python3 bench.py > $OUT/${TAG}_bench_default.json.log 2>&1; echo "default bench rc=$?"
timeout -k 10 300 bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1; echo "c2 profile rc=$?"
timeout -k 10 200 bash tools/profile_workload.sh $TAG c3 > /dev/null 2>&1; echo "c3 stats rc=$?"
timeout -k 10 300 bash tools/profile_workload.sh $TAG c4 > /dev/null 2>&1; echo "c4 stats rc=$?"
timeout -k 10 200 bash tools/pmc_traffic.sh $TAG c3 "Subleq-16, B=4096, 64 sims" > /dev/null 2>&1; echo "c3 traffic rc=$?"
timeout -k 10 400 bash tools/pmc_sq.sh $TAG c3 > /dev/null 2>&1; echo "c3 sq rc=$?"
timeout -k 10 200 python3 tools/sqp_stamps.py > $OUT/${TAG}_sqp_stamps.txt 2>&1; echo "sqp stamps rc=$?"
BETA=1 timeout -k 10 200 python3 tools/sqp_stamps.py > $OUT/${TAG}_sqp_stamps_beta1.txt 2>&1; echo "sqp stamps beta1 rc=$?"
timeout -k 10 200 python3 tools/persist_stamps.py > $OUT/${TAG}_persist_stamps.txt 2>&1; echo "persist stamps rc=$?"
timeout -k 10 200 python3 tools/subleq_stamps.py > $OUT/${TAG}_subleq_stamps.txt 2>&1; echo "subleq stamps rc=$?"
