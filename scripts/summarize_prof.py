"""Turn rocprofv3 CSV output (kernel trace / stats / PMC passes) into the small text summaries committed under profiles/."""
import collections
import csv
import glob
import statistics as st
import sys


def kernel_stats(dirn, out):
    f = glob.glob(f"{dirn}/**/*_kernel_stats.csv", recursive=True)
    if f:
        out.write("== rocprofv3 --kernel-trace --stats : kernel_stats.csv\n")
        out.write(open(f[0]).read())
    f = glob.glob(f"{dirn}/**/*_kernel_trace.csv", recursive=True)
    if f:
        rows = list(csv.DictReader(open(f[0])))
        by = collections.defaultdict(list)
        meta = {}
        for r in rows:
            by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                                      r["Workgroup_Size_X"], r["Grid_Size_X"])
        out.write("\n== per-kernel duration from kernel_trace.csv (ns): calls, mean, median, min, p90, max | VGPR AGPR SGPR LDS scratch wg grid\n")
        for k, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            d = sorted(d)
            out.write(f"{k[:90]:90s} {len(d):6d} {st.mean(d):10.0f} {st.median(d):10.0f} {d[0]:9d} {d[int(.9 * (len(d) - 1))]:9d} {d[-1]:10d} | "
                      + " ".join(meta[k]) + "\n")


def pmc(dirn, out):
    for f in glob.glob(f"{dirn}/**/*_counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        for r in rows:
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        out.write(f"\n== rocprofv3 --pmc ({dirn}): per-dispatch counter values: kernel, counter, dispatches, mean, median\n")
        for (k, c), v in sorted(agg.items()):
            if "shc_cycle_kernel" in k:
                out.write(f"{k[:70]:70s} {c:22s} {len(v):5d} {st.mean(v):16.1f} {st.median(v):16.1f}\n")


if __name__ == "__main__":
    out = open(sys.argv[1], "w")
    for d in sys.argv[2:]:
        if "pmc" in d:
            pmc(d, out)
        else:
            kernel_stats(d, out)
    out.close()
    print(open(sys.argv[1]).read())
