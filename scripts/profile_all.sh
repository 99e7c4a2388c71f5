#!/bin/bash
# rocprofv3 evidence for the three single-GPU BASELINE.json workloads (run through gpurun from the repo root).
R=${GRAFT_REPO_ROOT:-$PWD}
PROF_DIR=prof_c2 bash $R/scripts/profile_round.sh --workload config2
PROF_DIR=prof_c3 PROF_STEPS=200 bash $R/scripts/profile_round.sh --workload config3
PROF_DIR=prof_c4 PROF_STEPS=200 bash $R/scripts/profile_round.sh --workload config4
# keep the merged output small: the per-dispatch CSVs are what scripts/summarize_prof.py reads
find $R/gpurun_out/prof_c* -name '*.db' -delete 2>/dev/null
du -sh $R/gpurun_out/prof_c*
