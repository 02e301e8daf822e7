#!/bin/bash
# GPU-box helper (through gpurun): what is re-run after a change to search_persist_kernel -- the DeepSea parity tests of every variant, the C2 bench
# line and the in-kernel stamps of the diagnostic build
python -m pytest tests -m gpu -x -q -k "deepsea or selfplay_matches or search_ or evaluate or facade or fullsize" > gpurun_out/c2_tests.log 2>&1; tail -3 gpurun_out/c2_tests.log
python bench.py --no-extra --no-cpu-baseline > gpurun_out/c2_bench.log 2>&1
python -c "
import json
d=json.loads([l for l in open('gpurun_out/c2_bench.log').read().splitlines() if l.startswith('{')][-1]); print('c2', round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
python tools/persist_stamps.py > gpurun_out/c2_stamps.txt 2>&1; head -16 gpurun_out/c2_stamps.txt | cut -c1-160
