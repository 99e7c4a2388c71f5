#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench command, FETCH_SIZE / WRITE_SIZE / SQ PMC passes (each in its own run),
#   and the same PMC passes on a plane copy of known size (calibration of the byte counters).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${PROF_DIR:-prof}; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps ${PROF_STEPS:-400} --warmup 40 --no-cpu-baseline --no-fused-probe --no-also --no-parity $*"   # (--no-parity: the parity block reads the output ring while the loop is alive - a kernel launch the PMC passes, which serialise dispatches, would hold back until the loop has idled out)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $BENCH > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $BENCH > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -- $BENCH > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_lds -- $BENCH > $O/pmc_lds.log 2>&1
CAL="python -c \"import sys; sys.path.insert(0,'$R'); from syropod_highlevel_controller_amd import engine; engine.lib().shc_debug_plane_copy(0, 64*1024*1024, 20)\""
eval rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_cal_fetch -- $CAL > $O/cal_fetch.log 2>&1
eval rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_cal_write -- $CAL > $O/cal_write.log 2>&1
grep -h '^{' $O/trace.log | tail -1 > $O/bench_line.json
ls $O
