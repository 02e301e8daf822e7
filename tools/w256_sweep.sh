#!/bin/bash
# GPU-box helper: Subleq-256 self-play (the first two steps of the episodes, round 3's measurement) under the sub-batch / FC-tile knobs
for cfg in "0 0" "2 0" "4 0" "1 0" "2 64" "4 16"; do set -- $cfg
  EAZ_SHARDS=$1 EAZ_FC_TILE=$2 python bench.py --workload c3 --word-size 256 --selfplay-steps 2 --steps 3 --no-extra --no-cpu-baseline > gpurun_out/w256_s$1_t$2.log 2>&1
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/w256_s$1_t$2.log').read().splitlines() if l.startswith('{')][-1]); print('shards $1 fc_tile $2:', round(d['value']), d.get('kernel_ms_instrumented_pass'))"
done
