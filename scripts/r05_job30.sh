set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_step_k.py -q -x 2>&1 | tail -4
python bench.py > $O/bench_default_long.json 2> $O/bench_long.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_drv.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_like_b.json 2>> $O/bench_drv.err
python - <<'PY'
import json
for f in ('bench_default_long','bench_driver_like','bench_driver_like_b'):
    t=[l for l in open(f'gpurun_out/r05_job30/{f}.json').read().splitlines() if l.startswith('{')]
    d=json.loads(t[-1])
    print(f, 'headline %.4e'%d['value'], d['ms_per_step'], d['roofline'].get('bound'), d['roofline'].get('valu_issue_frac'), d['roofline'].get('frac'), 'launch traffic', d['roofline']['one_launch_per_cycle'].get('traffic'), d['roofline']['one_launch_per_cycle'].get('frac'))
    for a in d['config'].get('also', []):
        fk=a.get('fused_K_with_per_cycle_inputs') or {}
        r=a.get('roofline') or {}
        print('  ', (a.get('workload') or '')[:50], '| %.3e'%a['value'], '| hbm %.3f'%r.get('frac',0), 'traffic', r.get('traffic'), 'valu', r.get('valu_issue_frac'), r.get('bound'), '| fused16', a.get('fused_16_cycles_per_launch_value'), '| fusedK', fk.get('value'))
PY
