set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job8; mkdir -p $O
SHC_LIB=$PWD/gpurun_variants/busy/libshc_batch.so python scripts/resident_cycle_time.py 2>&1 | grep "res2\|resident" | tail -3 > $O/busy.txt
cat $O/busy.txt
python scripts/step_k_probe.py config3 > $O/step_k_probe.txt 2>&1
python scripts/step_k_probe.py config4 >> $O/step_k_probe.txt 2>&1
cat $O/step_k_probe.txt
