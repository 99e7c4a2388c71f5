set -u
cd $GRAFT_REPO_ROOT
bash scripts/profile_all.sh > /dev/null 2>&1
ls gpurun_out/summaries | head -30
cp gpurun_out/summaries/traffic.json profiles/traffic.json   # (on the box: the long run below reports the fractions of these very kernels)
O=$PWD/gpurun_out/r05_job17; mkdir -p $O
python bench.py > $O/bench_default_long.json 2> $O/bench_long.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_drv.err
python bench.py --steps 20 --warmup 5 --force-dist --workload config4 --no-cpu-baseline > $O/bench_forcedist_config4_rccl.json 2>> $O/bench_drv.err
python bench.py --steps 20 --warmup 5 --force-dist --workload config4 --gather peer --no-cpu-baseline > $O/bench_forcedist_config4_peer.json 2>> $O/bench_drv.err
python - <<'PY'
import json
for f in ('bench_default_long','bench_driver_like'):
    t=[l for l in open(f'gpurun_out/r05_job17/{f}.json').read().splitlines() if l.startswith('{')]
    d=json.loads(t[-1])
    print(f, 'headline %.4e'%d['value'], d['ms_per_step'], d['roofline'].get('bound'), d['roofline'].get('valu_issue_frac'), d['roofline'].get('frac'))
    for a in d['config'].get('also', []):
        fk=a.get('fused_K_with_per_cycle_inputs') or {}
        r=a.get('roofline') or {}
        print('  ', (a.get('workload') or '')[:50], '| %.3e'%a['value'], '| hbm %.3f'%r.get('frac',0), 'valu', r.get('valu_issue_frac'), r.get('bound'), '| fused16', a.get('fused_16_cycles_per_launch_value'), '| fusedK', fk.get('value'))
PY
