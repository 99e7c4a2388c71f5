#!/bin/bash
# Round profiles (r06) (run through gpurun from the repo root): the BASELINE.json configurations + the section 8(f) workloads, each its own
# rocprofv3 passes (kernel trace + stats; FETCH_SIZE, WRITE_SIZE, SQ, LDS counters each in a run of their own - scripts/profile_round.sh),
# summarised on the box (the raw CSVs exceed what gpurun copies back) into gpurun_out/summaries/, which is then copied to profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
S=$R/gpurun_out/summaries; mkdir -p $S
run() { # name, summary file, traffic key, resident K or "", bench arguments...
  local name=$1 out=$2 key=$3 res=$4; shift 4
  if [ -n "${ONLY:-}" ] && [[ " $ONLY " != *" $name "* ]]; then return; fi   # ONLY="c2_launch c4e": a subset, merged into the summaries' traffic.json
  PROF_DIR=prof_$name PROF_STEPS=${PROF_STEPS:-300} bash scripts/profile_round.sh "$@" > $S/$name.log 2>&1
  python scripts/summarize_prof.py gpurun_out/prof_$name $S/$out $key $res > /dev/null 2>> $S/$name.log
  rm -rf gpurun_out/prof_$name
}
SHC_BENCH_NO_POSTED_PROBE=1 run c2_resident r06_config2_resident_rocprofv3.txt config2:resident:4096:1 resident:4000 --workload config2
SHC_BENCH_NO_POSTED_PROBE=1 run c4_resident r06_config4_resident_8x5_rocprofv3.txt config4:resident:4000:1 resident:4000 --workload config4 --instances 4000 --no-joint-efforts
run c2_launch r06_config2_launch_rocprofv3.txt config2+efforts:4096:1 launch --workload config2 --mode launch
run c3 r06_config3_rocprofv3.txt config3:65536:1 split --workload config3 --no-joint-efforts
run c4 r06_config4_rocprofv3.txt config4:131072:1 split --workload config4 --no-joint-efforts
run c4e r06_config4_joint_torques_rocprofv3.txt config4+efforts:131072:1 split --workload config4 --joint-efforts
run rough r06_rough_terrain_rocprofv3.txt rough:65536:1 split --workload rough --no-joint-efforts
run gravity r06_gravity_aligned_rocprofv3.txt gravity:65536:1 pairs --workload gravity --no-joint-efforts
run gravity3 r06_gravity_aligned_admittance_imu_rocprofv3.txt gravity3:65536:1 pairs --workload gravity3 --no-joint-efforts
PROF_STEPS=100 run c5 r06_config5_rocprofv3.txt config5:1048576:1 fleet --workload config5
PROF_STEPS=60 run c4full r06_config4_full_size_rocprofv3.txt config4full:1048576:1 shards:8 --workload config4full
# shc_engine_step_k as the timed form (K = 16 cycles per launch, a new input row every cycle): the batch kernels behind the `also` rows
PROF_STEPS=640 run c3k r06_config3_step_k_rocprofv3.txt config3:stepk:65536:16 batch-split --workload config3 --mode step_k --no-joint-efforts
PROF_STEPS=640 run c4k r06_config4_step_k_rocprofv3.txt config4:stepk:131072:16 batch-split --workload config4 --mode step_k --no-joint-efforts
PROF_STEPS=640 run roughk r06_rough_terrain_step_k_rocprofv3.txt rough:stepk:65536:16 batch-split --workload rough --mode step_k --no-joint-efforts
PROF_STEPS=320 run gravityk r06_gravity_aligned_step_k_rocprofv3.txt gravity:stepk:65536:16 batch-split --workload gravity --mode step_k --no-joint-efforts
PROF_STEPS=320 run gravity3k r06_gravity_aligned_admittance_imu_step_k_rocprofv3.txt gravity3:stepk:65536:16 batch-split --workload gravity3 --mode step_k --no-joint-efforts
ls -la $S
if [ -n "${ONLY:-}" ]; then exit 0; fi
# the driver-shaped default run (every kernel of the bench line incl. the batch form of shc_engine_step_k), kernel trace + stats only
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $S/default.log 2>&1
python - <<PY > $S/r06_bench_default_rocprofv3.txt
import glob, os
f = sorted(glob.glob("$R/gpurun_out/prof_default/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
print("== rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline : kernel_stats.csv (the driver's N = 1 command)")
print(open(f[-1]).read() if f else "no kernel_stats.csv")
PY
grep -h '^{' $S/default.log | tail -1 > $S/r06_bench_default_line.json
rm -rf $R/gpurun_out/prof_default
ls -la $S
