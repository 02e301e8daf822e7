set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for c in MfmaUtil SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcx_$c -- python3 $R/bench.py --steps 1 --warmup 0 --selfplay-steps 2 --no-cpu-baseline > $OUT/pmcx_$c.log 2>&1 || echo "FAILED $c"
done
python3 $R/tools/pmc_summary.py $OUT/pmcx.json "x" $OUT/pmcx_* | grep "fc16_forward_kernel<2>\|tree_step"
