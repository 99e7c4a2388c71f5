cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_half_steps.py -q -x -k "against_the_oracle and admittance" 2>&1 | grep -v "^\[" | tail -30
