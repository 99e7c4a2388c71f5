set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job18; mkdir -p $O
for rep in 1 2; do
for v in prev new; do
  if [ $v = new ]; then unset SHC_LIB; else export SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so; fi
  python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/ab.txt
  python scripts/resident_latency.py 2>&1 | grep "RESULT burst    1\|RESULT burst   20\|warm-up   5" | sed "s/^/$v: /" >> $O/ab.txt
done
done
unset SHC_LIB
cat $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_resident.py -q -x 2>&1 | tail -2
