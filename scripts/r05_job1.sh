set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job1; mkdir -p $O
for v in head new; do
  for rep in 1 2; do
    if [ $v = head ]; then export SHC_LIB=$PWD/gpurun_variants/head/libshc_batch.so; else unset SHC_LIB; fi
    python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/cycle_time.txt
    SHC_NO_EFFORTS=1 python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/$v noeff: /" >> $O/cycle_time.txt
  done
  python scripts/resident_cycle_time.py 4000 4000 octopod 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/cycle_time.txt
  python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/cycle_time.txt
done
unset SHC_LIB
cat $O/cycle_time.txt
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | tail -5 > $O/tests_resident.txt
cat $O/tests_resident.txt
python bench.py --steps 20 --warmup 5 --no-also > $O/bench_driver_like.json 2> $O/bench.err
python -c "
import json;d=json.loads(open('$O/bench_driver_like.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d.get('roofline'))"
