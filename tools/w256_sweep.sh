set -e
for sh in 0 1 2; do for ft in 0 64; do
echo "== EAZ_SHARDS=$sh EAZ_FC_TILE=$ft"
EAZ_SHARDS=$sh EAZ_FC_TILE=$ft timeout -k 10 200 python bench.py --workload c3 --word-size 256 --steps 2 --warmup 1 --selfplay-steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']), j['ms_per_step'], j['roofline']['kernel'], round(j['roofline']['avg_launch_us'],1), j['kernel_ms_instrumented_pass'])
"
done; done
