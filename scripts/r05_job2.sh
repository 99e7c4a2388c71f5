set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job2; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/doorbell_paths scripts/ubench/doorbell_paths.hip && timeout 120 /tmp/doorbell_paths > $O/doorbell_paths.txt 2>&1
cat $O/doorbell_paths.txt
for v in base uf vearly pose1 cin all base all; do
  if [ $v = base ]; then unset SHC_LIB; else export SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so; fi
  python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/cycle_time.txt
  SHC_NO_EFFORTS=1 python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/$v noeff: /" >> $O/cycle_time.txt
  python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/cycle_time.txt
done
unset SHC_LIB
cat $O/cycle_time.txt
timeout 900 python -m pytest tests/test_gpu_sequences.py -x -q -k "mixed" 2>&1 | tail -5 > $O/tests_seq.txt
cat $O/tests_seq.txt
timeout 600 python -m pytest tests/test_gpu_resident.py -x -q -k "direct or byte" 2>&1 | tail -3 > $O/tests_res.txt
cat $O/tests_res.txt
