set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job23; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_half_steps.py tests/test_gpu_mixed_dof.py tests/test_gpu_step_k.py -q -x 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
for w in gravity gravity3; do
python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-fused-probe > $O/bench_$w.json 2> $O/err.txt
python - <<PY
import json
t=[l for l in open('gpurun_out/r05_job23/bench_$w.json').read().splitlines() if l.startswith('{')]
d=json.loads(t[-1])
print('$w', '%.3e'%d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'parity', d['parity'] and d['parity']['max_abs_dq'])
PY
done
