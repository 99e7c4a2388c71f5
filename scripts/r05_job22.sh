set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job22; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $O/tests_gpu.txt 2>&1
cat $O/tests_gpu.txt
bash scripts/r05_job17.sh
