set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job12; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/tests_gpu.txt 2>&1
cat $O/tests_gpu.txt
for g in rccl peer; do
  timeout 600 python bench.py --steps 20 --warmup 5 --force-dist --gather $g --no-cpu-baseline > $O/bench_forcedist_$g.json 2> $O/bench_forcedist_$g.err
  tail -3 $O/bench_forcedist_$g.err
  python - <<PY
import json
d=json.loads(open('$O/bench_forcedist_$g.json').read().strip().splitlines()[-1])
c=d['config']
print('$g', d['value'], d['ms_per_step'], c.get('gather_form'), c.get('gather_ms'), c.get('weak_scaling_efficiency'), (c.get('scale_reference') or {}).get('value'))
PY
done
bash scripts/r05_job5.sh
