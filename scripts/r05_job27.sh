set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job27; mkdir -p $O
for rep in 1 2; do
for v in noburst2 burst2; do
  echo "== $v rep $rep"
  SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so python scripts/resident_latency.py 2>&1 | grep RESULT
  SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench20 value %.4e ms_per_step %.5f kernel_ms %.5f'%(d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
  SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so python scripts/resident_cycle_time.py 4096 4000 config2 2>&1 | tail -1
done; done 2>&1 | tee $O/ab.txt
