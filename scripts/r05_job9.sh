set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job9; mkdir -p $O
SHC_LIB=$PWD/gpurun_variants/busy/libshc_batch.so python scripts/resident_cycle_time.py 2>&1 | grep "res2\|resident" | tail -3 > $O/busy.txt
python scripts/resident_cycle_time.py 2>&1 | tail -1 >> $O/busy.txt
SHC_NO_EFFORTS=1 python scripts/resident_cycle_time.py 2>&1 | tail -1 >> $O/busy.txt
python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 >> $O/busy.txt
python scripts/resident_cycle_time.py 4000 4000 octopod 2>&1 | tail -1 >> $O/busy.txt
cat $O/busy.txt
timeout 1200 python -m pytest tests/test_gpu_resident.py tests/test_gpu_step_k.py tests/test_gpu_sharding.py -q -x 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
python scripts/resident_latency.py 2>&1 | grep RESULT > $O/latency.txt; cat $O/latency.txt
