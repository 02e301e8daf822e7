#!/bin/bash
# GPU-box helper (through gpurun): the GPU suite, then the Subleq-256 and connect_four bench lines -- what is re-run after a change to the per-simulation path
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t16.log 2>&1; tail -4 gpurun_out/r4_t16.log; for t in 2 32; do python bench.py --workload c3 --word-size 256 --selfplay-steps $t --steps 2 --no-extra --no-cpu-baseline > gpurun_out/r4_b13_w256_T$t.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r4_b13_w256_T$t.log').read().splitlines() if l.startswith('{')][-1]); print($t, d['value'], d.get('kernel_ms_instrumented_pass'))"; done; python bench.py --workload c4 --selfplay-steps 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4_b13_c4.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r4_b13_c4.log').read().splitlines() if l.startswith('{')][-1]); print('c4', d['value'], d.get('kernel_ms_instrumented_pass'))"
