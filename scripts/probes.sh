#!/bin/bash
# The micro-benchmarks and probes DESIGN.md quotes, run through gpurun from the repo root; their output lands in gpurun_out/summaries/
# and is copied to profiles/r03_probe_*.txt.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
S=$R/gpurun_out/summaries; mkdir -p $S
T=$(mktemp -d)
for u in mapping_probe wave_placement; do
  hipcc --offload-arch=gfx950 -O3 -w -I syropod_highlevel_controller_amd/csrc -o $T/$u scripts/ubench/$u.hip > $S/r03_probe_$u.txt 2>&1 && $T/$u >> $S/r03_probe_$u.txt 2>&1
done
python scripts/resident_bench.py 4096 2000 > $S/r03_probe_resident_modes.txt 2>&1
python scripts/resident_latency.py > $S/r03_probe_resident_latency.txt 2>&1
python scripts/split_probe.py > $S/r03_probe_split_streams.txt 2>&1
python scripts/generic_probe.py > $S/r03_probe_generic_vs_feature_exact.txt 2>&1
python scripts/loop_calls_probe.py > $S/r03_probe_loop_level_calls.txt 2>&1
# walker / model wavefront clocks of the two-wavefront resident kernel (development build with phase stamps; the box's copy only)
SHC_EXTRA_FLAGS="-DSHC_RES2_TIMING" python scripts/resident_bench.py 4096 500 > $S/r03_probe_resident_phase_clocks.txt 2>&1
rm -rf $T
ls -la $S
