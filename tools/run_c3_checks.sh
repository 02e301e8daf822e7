#!/bin/bash
# GPU-box helper (through gpurun): the Subleq-16 / golden / driver tests, then the C3 bench line under both layer-1 input orders -- what is re-run
# after a change to subleq16_selfplay_kernel
python -m pytest tests -m gpu -x -q -k "subleq16 or c3 or golden or evaluate or reanalyze" > gpurun_out/c3_checks_tests.log 2>&1; tail -3 gpurun_out/c3_checks_tests.log
for o in constants-first ascending; do
  python bench.py --workload c3 --no-extra --no-cpu-baseline --l1-order $o > gpurun_out/c3_checks_bench_$o.log 2>&1
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/c3_checks_bench_$o.log').read().splitlines() if l.startswith('{')][-1]); print('$o', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
