set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job13; mkdir -p $O
python scripts/step_k_probe.py config3 > $O/step_k_probe.txt 2>&1
python scripts/step_k_probe.py config4 >> $O/step_k_probe.txt 2>&1
grep -v amdgpu.ids $O/step_k_probe.txt
