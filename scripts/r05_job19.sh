set -u
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05_job19; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
summ() { python - "$1" "$2" <<'PY'
import csv, glob, os, sys, collections, statistics as st
d, title = sys.argv[1], sys.argv[2]
print("== " + title)
f = sorted(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
if f:
    rows = list(csv.reader(open(f[-1])))
    for r in rows[:6]:
        print(",".join(r)[:260])
k = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
if k:
    by = collections.defaultdict(list); meta = {}
    for r in csv.DictReader(open(k[-1])):
        if "shc_" in r["Kernel_Name"] and ("resident" in r["Kernel_Name"] or "batch" in r["Kernel_Name"] or "cycle_kernel" in r["Kernel_Name"]):
            by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["Workgroup_Size_X"], r["Grid_Size_X"])
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:4]:
        print(f"{n[:100]}: {len(v)} launches, mean {st.mean(v):.0f} ns, median {st.median(v):.0f}, max {max(v)} | VGPR AGPR scratch wg grid = {' '.join(meta[n])}")
PY
}
for c in rough gravity; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$c -- python $R/scripts/resident_cycle_time.py 4096 3000 $c > $O/$c.log 2>&1
  summ $O/p_$c "rocprofv3 --kernel-trace --stats -- python scripts/resident_cycle_time.py 4096 3000 $c   (resident loop, one wavefront per robot group; 3 launches of 3 000 cycles)" >> $O/summary.txt
  grep "resident" $O/$c.log | tail -1 >> $O/summary.txt
  rm -rf $O/p_$c
done
for c in config3 config4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$c -- python $R/scripts/step_k_probe.py $c > $O/k_$c.log 2>&1
  summ $O/k_$c "rocprofv3 --kernel-trace --stats -- python scripts/step_k_probe.py $c   (shc_engine_step_k, K = 16: shc_batch_kernel halves on two streams; shc_cycle_kernel = shc_engine_step(16) with the inputs held)" >> $O/summary.txt
  grep "cycles/s" $O/k_$c.log >> $O/summary.txt
  rm -rf $O/k_$c
done
cat $O/summary.txt
