#!/bin/bash
# One workload of scripts/profile_all.sh (same passes, same summariser): bash scripts/profile_one.sh c2_resident
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
want=$1
grep -E "^([A-Z_]+=[0-9]+ )?run $want " scripts/profile_all.sh > /tmp/profile_one_line.sh
sed -n '/^R=/,/^}/p' scripts/profile_all.sh > /tmp/profile_one_head.sh
cat /tmp/profile_one_head.sh /tmp/profile_one_line.sh > /tmp/profile_one_run.sh
bash /tmp/profile_one_run.sh
ls -la $R/gpurun_out/summaries
