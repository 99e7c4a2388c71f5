set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job5; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench.err
tail -5 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_job5/bench_driver_like.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline'].get('bound'), d['roofline'].get('valu_issue_frac'))
for a in d['config']['also']:
    c=a
    fk=c.get('fused_K_with_per_cycle_inputs') or {}
    print((a.get('workload') or c.get('workload'))[:58], '| %.3e'%a['value'], '| frac %.3f'%a['roofline']['frac'], '| fused16', c.get('fused_16_cycles_per_launch_value'), '| fusedK', fk.get('value'), fk.get('error'), (fk.get('parity') or {}).get('max_abs_dq'), (fk.get('roofline') or {}).get('frac'), '| parity', (a.get('parity') or {}).get('max_abs_dq'))
PY
