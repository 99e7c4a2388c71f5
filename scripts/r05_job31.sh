set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job31; mkdir -p $O
( time SHC_SOAK_CYCLES=20000 timeout 2400 python -m pytest tests/test_gpu_teacher_forced.py -q -x -k soak 2>&1 | tail -8 ) > $O/soak.txt 2>&1
cat $O/soak.txt
