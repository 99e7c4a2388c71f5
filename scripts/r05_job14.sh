set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job14; mkdir -p $O
python scripts/step_k_probe.py config3 2>&1 | grep -v amdgpu.ids | grep "held\|no inputs\|everything" > $O/step_k_probe.txt
python scripts/step_k_probe.py config4 2>&1 | grep -v amdgpu.ids | grep "held\|no inputs\|change" >> $O/step_k_probe.txt
cat $O/step_k_probe.txt
timeout 900 python -m pytest tests/test_gpu_step_k.py tests/test_gpu_split_streams.py -q -x 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
