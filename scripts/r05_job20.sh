set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job20; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_half_steps.py -q -x 2>&1 | tail -12 > $O/tests.txt
cat $O/tests.txt
python bench.py --workload gravity3 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_gravity3.json 2> $O/err.txt
SHC_ROT_SPLIT=0 python bench.py --workload gravity3 --steps 300 --warmup 30 --no-cpu-baseline --no-fused-probe > $O/bench_gravity3_onekernel.json 2>> $O/err.txt
python - <<'PY'
import json
for f in ('bench_gravity3','bench_gravity3_onekernel'):
    t=[l for l in open(f'gpurun_out/r05_job20/{f}.json').read().splitlines() if l.startswith('{')]
    d=json.loads(t[-1]); c=d['config']
    fk=c.get('fused_K_with_per_cycle_inputs') or {}
    print(f, '%.3e'%d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'parity', d['parity'] and d['parity']['max_abs_dq'], 'fused16', c.get('fused_16_cycles_per_launch_value'), 'fusedK', fk.get('value'), fk.get('error'))
PY
tail -3 $O/err.txt
