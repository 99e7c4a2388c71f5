#!/bin/bash
# config 4's two step-time modes against the address-translation counters.  Each line = one PROCESS: the octopod engine of 131 072 robots created
# fresh ("none") or after a small engine has lived in the process ("small", "small-wave": round 4's recipe for the slow mode), 900 steps timed by the host, the same
# process under rocprofv3 --pmc TCP_UTCL1_* (+ kernel trace): per launch of shc_cycle_kernel<8,5,65> the mean duration and the UTCL1 requests / misses.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/config4_modes; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
for rep in 1 2 3; do
  for v in none small small-wave; do
    python $R/scripts/history_probe.py $v 2>/dev/null | tail -1 | sed "s/^/untraced process $rep: /" >> $O/modes.txt
    rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --kernel-trace --output-format csv -d $O/p_${v}_$rep -- python $R/scripts/history_probe.py $v > $O/p_${v}_$rep.log 2>&1
    python - <<PY >> $O/modes.txt
import csv, glob, collections, statistics as st
f = glob.glob("$O/p_${v}_$rep/**/*_counter_collection.csv", recursive=True)
k = glob.glob("$O/p_${v}_$rep/**/*_kernel_trace.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "shc_cycle_kernel<8, 5, 65" in r["Kernel_Name"] and int(r["Grid_Size"] if "Grid_Size" in r else 1 << 30) > 100000:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(k[0])) if "shc_cycle_kernel<8, 5, 65" in r["Kernel_Name"]] if k else []
dur = [d for d in dur if d > 40000]
req, miss = st.mean(agg.get("TCP_UTCL1_REQUEST_sum", [0])), st.mean(agg.get("TCP_UTCL1_TRANSLATION_MISS_sum", [0]))
print(f"  traced process $rep, $v: half-launch mean {st.mean(dur) if dur else 0:.0f} ns, median {st.median(dur) if dur else 0:.0f} ns over {len(dur)} launches (dispatches serialised by the PMC pass); UTCL1 requests {req:.3e}, translation misses {miss:.3e} per launch = {miss / max(req, 1):.4f}")
PY
    rm -rf $O/p_${v}_$rep
  done
done
cat $O/modes.txt
