"""Development aid: from a rocprofv3 kernel_trace.csv, the timeline of the last N launches of the cycle kernel per queue - duration, period, and how much
of each launch overlaps a launch on the other queue."""
import csv
import glob
import sys
import collections
import statistics as st

f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
pat = sys.argv[2] if len(sys.argv) > 2 else "shc_cycle_kernel<8, 5"
last = int(sys.argv[3]) if len(sys.argv) > 3 else 600
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 4 and sys.argv[4] == "region":   # the timed region of bench.run_workload: the `last` launches that follow the last fused (16-cycle, long) launch + its 5 warm-up steps
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    k = max(i for i, d in enumerate(dur) if d > 500000)
    rows = rows[k + 1 + 12:k + 1 + 12 + last]
    print("region: launches", k + 13, "...", k + 13 + last, "wall", (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 / (last / 2), "us per step")
else:
    rows = rows[-last:]
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"]))
print("queues:", {q: len(v) for q, v in byq.items()})
for q, v in byq.items():
    d = [b - a for a, b, _ in v]
    per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
    gap = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print(f"queue {q}: launches {len(v)} grid {v[-1][2]} duration mean {st.mean(d)/1e3:.1f} median {st.median(d)/1e3:.1f} us; period median {st.median(per)/1e3:.1f} us; gap to next launch median {st.median(gap)/1e3:.1f} us")
qs = list(byq)
if len(qs) == 2:
    a, b = byq[qs[0]], byq[qs[1]]
    j = 0
    ov = []
    for s0, e0, _ in a:
        tot = 0
        for s1, e1, _ in b:
            if e1 < s0:
                continue
            if s1 > e0:
                break
            tot += max(0, min(e0, e1) - max(s0, s1))
        ov.append(tot / (e0 - s0))
    print(f"share of a launch on queue {qs[0]} that overlaps launches on queue {qs[1]}: mean {st.mean(ov):.2f}")
    # phase of queue-b starts within queue-a's launch
    ph = []
    for s1, e1, _ in b:
        for s0, e0, _ in a:
            if s0 <= s1 < e0:
                ph.append((s1 - s0) / (e0 - s0))
                break
    if ph:
        print(f"phase at which the other queue's launches start within a launch: median {st.median(ph):.2f}, p10 {sorted(ph)[len(ph)//10]:.2f}, p90 {sorted(ph)[9*len(ph)//10]:.2f}")
