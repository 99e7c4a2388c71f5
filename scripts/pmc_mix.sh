#!/bin/bash
# Dynamic instruction mix of the cycle kernel for one bench workload (development aid): rocprofv3 PMC passes of the
# per-type SQ_INSTS_VALU_* counters, medians per dispatch divided by the wave count.  usage: bash scripts/pmc_mix.sh [--workload configN]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_mix; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-fused-probe --no-also $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $O/a -- $BENCH > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/b -- $BENCH > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/c -- $BENCH > $O/c.log 2>&1
python - <<PY
import csv, glob, collections, statistics as st
for d in "abc":
    f = sorted(glob.glob("$O/%s/**/*_counter_collection.csv" % d, recursive=True))
    if not f: print("no output for pass", d); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[-1])):
        if "shc_cycle_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = st.median(agg["SQ_WAVES"])
    print("pass", d, "waves", w, {k: round(st.median(v) / w, 1) for k, v in sorted(agg.items()) if k != "SQ_WAVES"})
PY
