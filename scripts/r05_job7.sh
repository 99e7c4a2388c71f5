set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job7; mkdir -p $O
python scripts/resident_cycle_time.py 2>&1 | tail -1 > $O/cycle_time.txt
SHC_NO_EFFORTS=1 python scripts/resident_cycle_time.py 2>&1 | tail -1 >> $O/cycle_time.txt
python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 >> $O/cycle_time.txt
python scripts/resident_cycle_time.py 4000 4000 octopod 2>&1 | tail -1 >> $O/cycle_time.txt
cat $O/cycle_time.txt
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/tests_gpu.txt 2>&1
cat $O/tests_gpu.txt
bash scripts/r05_job5.sh
