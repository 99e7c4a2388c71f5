set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job28; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $O/tests_gpu.txt 2>&1
cat $O/tests_gpu.txt
for c in tipalign config2; do python scripts/resident_cycle_time.py 4096 4000 $c 2>&1 | tail -1; SHC_NO_EFFORTS=1 python scripts/resident_cycle_time.py 4096 4000 $c 2>&1 | tail -1; done | tee $O/tipalign_resident.txt
bash scripts/r05_job17.sh
