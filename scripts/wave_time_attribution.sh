#!/bin/bash
# Where a wavefront's time goes, from the SQ counters (quad-cycles; MI355X_MICROARCH.md: SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall,
# SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to SQ_WAVE_CYCLES).  One rocprofv3 --pmc pass per counter group and workload.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/wave_time; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
G2="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH"
G3="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_SALU"
run() { # name, bench args
  local name=$1; shift
  local B="python $R/bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-fused-probe --no-also --no-parity $*"
  for g in 1 2 3; do
    eval "C=\$G$g"
    SHC_BENCH_NO_POSTED_PROBE=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${name}_g$g -- $B > $O/${name}_g$g.log 2>&1
  done
}
run c2_resident --workload config2
run c3 --workload config3 --no-joint-efforts
run c4 --workload config4 --no-joint-efforts
run gravity --workload gravity --no-joint-efforts
python - <<PY > $O/summary.txt
import csv, glob, os, collections
for name in ("c2_resident", "c3", "c4", "gravity"):
    print("==", name)
    for g in (1, 2, 3):
        files = glob.glob("$O/%s_g%d/**/*counter_collection.csv" % (name, g), recursive=True)
        if not files:
            print("  group", g, ": no counter file (a counter name this build of rocprofv3 does not know?)", open("$O/%s_g%d.log" % (name, g)).read()[-300:].replace("\n", " | "))
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in files:
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "shc_" not in k or "debug" in k: continue
                acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            n = max(len(v) for v in cs.values())
            big = {c: max(v) for c, v in cs.items()}
            mean = {c: sum(v) / len(v) for c, v in cs.items()}
            use = big if "resident" in k else mean
            print("  group %d %-62s dispatches %d (%s)" % (g, k, n, "largest launch" if "resident" in k else "mean per launch"))
            w = use.get("SQ_WAVES", 0) or 1
            for c, v in sorted(use.items()):
                extra = ""
                if "SQ_WAVE_CYCLES" in use and c.startswith(("SQ_ACTIVE", "SQ_WAIT", "SQ_INST_CYCLES")):
                    extra = "  = %.3f of SQ_WAVE_CYCLES" % (v / use["SQ_WAVE_CYCLES"])
                print("      %-24s %16.0f   per wave %12.1f%s" % (c, v, v / w, extra))
PY
cat $O/summary.txt
for d in $O/*_g?; do rm -rf $d; done   # (the raw CSVs exceed what gpurun copies back)
