"""Turn the rocprofv3 CSV output of scripts/profile_round.sh (kernel trace / stats / PMC passes) into the text summary
committed under profiles/ and the per-launch HBM traffic figure bench.py reports (profiles/traffic.json).

usage: python scripts/summarize_prof.py gpurun_out/prof profiles/r01_config2_rocprofv3.txt [traffic key, default config2:4096:1]
"""
import collections
import csv
import glob
import json
import os
import statistics as st
import sys

KERNEL = "shc_cycle_kernel"
# resident mode: ONE long launch runs K cycles; every per-launch figure below is divided by K (argv[4] = "resident:<K>", the
# longest launch of the trace is the K-cycle one bench.py times)
RESIDENT_K = None
CAL_KERNEL = "shc_plane_copy_kernel"
CAL_KIB = 64 * 1024 * 1024 * 8 / 1024.0  # scripts/profile_round.sh copies 64 Mi doubles per launch


def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def kernel_stats(dirn, out):
    f = newest(f"{dirn}/**/*_kernel_stats.csv")
    if f:
        out.write("== rocprofv3 --kernel-trace --stats : kernel_stats.csv\n")
        out.write(open(f).read())
    f = newest(f"{dirn}/**/*_kernel_trace.csv")
    med = None
    best_total = 0
    if f:
        rows = list(csv.DictReader(open(f)))
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
            if PAIRS and KERNEL in k:  # the two launches of a cycle follow each other on one stream: their durations add up
                med = (st.mean(d) + (med[0] if med else 0.0), st.median(d) + (med[1] if med else 0.0), len(d))
            elif KERNEL in k and (med is None or sum(d) > best_total):  # the specialisation that did the work (largest total time)
                best_total = sum(d)
                med = (st.mean(d), st.median(d), len(d))
                if RESIDENT_K:
                    out.write(f"  resident: the longest launch ran K = {RESIDENT_K} cycles: {d[-1]} ns / K = {d[-1] / RESIDENT_K:.1f} ns per cycle\n")
                    med = (d[-1] / RESIDENT_K, d[-1] / RESIDENT_K, 1)
    return med


PAIRS = False  # argv[4] == "pairs": a cycle = the walker-half launch + the model-half launch (shc_cycle_half_kernel<..., 1 / 2>), on each of the two
               # halves of the batch: durations and counters are summed over the two kernels, a step is two such pairs
SHARDS = 1     # argv[4] == "shards:N": see __main__
FLEET = False  # argv[4] == "fleet": a step = one launch of EACH morphology bin's kernel on concurrent streams; counters are summed over the bins


def pmc_fleet(dirn, kernel, out, label):
    """fleet step: per kernel specialisation (= morphology bin) the median per-dispatch value (a launch moves the same state whether it
    runs 1 or 16 fused cycles), summed over the specialisations"""
    f = newest(f"{dirn}/**/*_counter_collection.csv")
    res = {}
    if not f:
        return res
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out.write(f"\n== rocprofv3 --pmc ({label}): counter, kernel specialisation, dispatches, median - and the sum over the specialisations (one fleet step)\n")
    for c, names in sorted(per.items()):
        total = 0.0
        fewest = min(len(v) for v in names.values())
        for k, v in sorted(names.items()):
            bins = max(1, round(len(v) / fewest))   # two bins of the same (legs, longest DOF) run the same kernel: twice the dispatches
            out.write(f"{c:22s} {k[k.find('<'):k.find('>') + 1]:28s} {len(v):6d} {st.median(v):16.1f}" + (f"  x {bins} bins" if bins > 1 else "") + "\n")
            total += st.median(v) * bins * SHARDS
        out.write(f"{c:22s} {'sum over the bins' + (f' x {SHARDS} shards' if SHARDS > 1 else ''):28s} {'':6s} {total:16.1f}\n")
        res[c] = total
    return res


def pmc(dirn, kernel, out, label):
    """median per-dispatch value of every counter collected for `kernel` in the newest CSV under dirn"""
    if (FLEET or PAIRS) and kernel == KERNEL:
        return pmc_fleet(dirn, kernel, out, label)
    f = newest(f"{dirn}/**/*_counter_collection.csv")
    res = {}
    if not f:
        return res
    agg = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    names = collections.Counter(r["Kernel_Name"] for r in rows)
    main = names.most_common(1)[0][0] if names else None   # the specialisation that did the work
    for r in rows:
        if r["Kernel_Name"] == main:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.write(f"\n== rocprofv3 --pmc ({label}): kernel, counter, dispatches, mean, median\n")
    for c, v in sorted(agg.items()):
        out.write(f"{kernel:24s} {c:22s} {len(v):5d} {st.mean(v):16.1f} {st.median(v):16.1f}\n")
        res[c] = st.median(v)
        if RESIDENT_K and kernel == KERNEL:  # the K-cycle launch is the one with the largest counts; per cycle
            res[c] = max(v) / (1 if c == "SQ_WAVES" else RESIDENT_K)
            out.write(f"{'':24s} {c:22s} largest launch / K = {res[c]:.2f} per cycle\n")
    return res


if __name__ == "__main__":
    prof, dest = sys.argv[1], sys.argv[2]
    key = sys.argv[3] if len(sys.argv) > 3 else "config2:4096:1"
    if len(sys.argv) > 4 and sys.argv[4].startswith("resident:"):
        RESIDENT_K = int(sys.argv[4].split(":")[1])
        KERNEL = "shc_resident"
    # large batches: a step is TWO launches (the halves of the batch on two streams, shc_engine_step): per-step bytes = 2 x per launch
    FLEET = len(sys.argv) > 4 and (sys.argv[4] == "fleet" or sys.argv[4].startswith("shards:"))   # (bins that share a device run as single launches: SHC_FEAT_SINGLE_STREAM, shc_fleet.hpp)
    if len(sys.argv) > 4 and sys.argv[4].startswith("shards:"):   # N shards of ONE morphology on one device (bench.py --workload config4full): a step = N launches of the same kernel
        SHARDS = int(sys.argv[4].split(":")[1])
    if len(sys.argv) > 4 and sys.argv[4] in ("batch", "batch-split"):   # shc_engine_step_k: one launch (or one per half of a large batch) runs K cycles with their own inputs
        KERNEL = "shc_batch_kernel"
    PAIRS = len(sys.argv) > 4 and sys.argv[4] == "pairs"
    if PAIRS:
        KERNEL = "shc_cycle_half_kernel"
    per_step = 2 if len(sys.argv) > 4 and sys.argv[4] in ("split", "pairs", "batch-split") else 1
    out = open(dest, "w")
    dur = kernel_stats(f"{prof}/trace", out)
    fetch = pmc(f"{prof}/pmc_fetch", KERNEL, out, "FETCH_SIZE pass").get("FETCH_SIZE")
    write = pmc(f"{prof}/pmc_write", KERNEL, out, "WRITE_SIZE pass").get("WRITE_SIZE")
    sq = pmc(f"{prof}/pmc_sq", KERNEL, out, "SQ pass")
    lds = pmc(f"{prof}/pmc_lds", KERNEL, out, "LDS pass")
    cf = pmc(f"{prof}/pmc_cal_fetch", CAL_KERNEL, out, "calibration, FETCH_SIZE").get("FETCH_SIZE")
    cw = pmc(f"{prof}/pmc_cal_write", CAL_KERNEL, out, "calibration, WRITE_SIZE").get("WRITE_SIZE")
    out.write("\n== PMC calibration on a plane copy of known size (shc_debug_plane_copy: 64 Mi doubles = %d KiB read + %d KiB written per "
              "launch, 16 B per lane)\n" % (CAL_KIB, CAL_KIB))
    kf = CAL_KIB / cf if cf else float("nan")
    kw = CAL_KIB / cw if cw else float("nan")
    out.write(f"FETCH_SIZE median {cf} KiB -> correction factor x{kf:.3f}\nWRITE_SIZE median {cw} KiB -> correction factor x{kw:.3f}\n")
    traffic = None
    if fetch and write and cf and cw:
        kib = kf * fetch + kw * write
        traffic = kib * 1024.0 * per_step
        out.write(f"\n== derived ({key})\nHBM traffic per {'cycle' if RESIDENT_K else 'launch'} = {kf:.3f} x FETCH_SIZE + {kw:.3f} x WRITE_SIZE = {kf:.3f} x {fetch:.1f} KiB + "
                  f"{kw:.3f} x {write:.1f} KiB = {kib:.1f} KiB = {kib * 1024.0 / 1e6:.2f} MB\n")
        if per_step == 2:
            out.write(f"a step of this batch = two launches (halves of the batch on two streams, overlapping): HBM traffic per step = {traffic / 1e6:.2f} MB; the step period "
                      f"is about one launch's duration\n")
    if sq.get("SQ_WAVES"):
        w = sq["SQ_WAVES"]
        out.write(f"VALU instructions per wave per {'cycle' if RESIDENT_K else 'launch'} = SQ_INSTS_VALU / SQ_WAVES = {sq.get('SQ_INSTS_VALU', 0) / w:.0f}; SALU {sq.get('SQ_INSTS_SALU', 0) / w:.0f}; "
                  f"LDS {sq.get('SQ_INSTS_LDS', 0) / w:.0f}\n")
        if sq.get("SQ_WAVE_CYCLES"):
            out.write(f"VALU issue share of wave lifetime = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {sq.get('SQ_ACTIVE_INST_VALU', 0) / sq['SQ_WAVE_CYCLES']:.3f}; "
                      f"wave lifetime = 4 x SQ_WAVE_CYCLES / SQ_WAVES = {4 * sq['SQ_WAVE_CYCLES'] / w:.0f} cycles\n")
    if lds.get("SQ_LDS_IDX_ACTIVE"):
        out.write(f"LDS bank-conflict share of LDS-active cycles = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = "
                  f"{lds.get('SQ_LDS_BANK_CONFLICT', 0) / lds['SQ_LDS_IDX_ACTIVE']:.3f}; LDS instructions per wave = "
                  f"{lds.get('SQ_INSTS_LDS', 0) / max(lds.get('SQ_WAVES', 1), 1):.0f}\n")
    valu = None
    if dur:
        out.write(f"kernel duration (kernel_trace): mean {dur[0]:.0f} ns, median {dur[1]:.0f} ns over {dur[2]} launches\n")
        if sq.get("SQ_ACTIVE_INST_VALU") and sq.get("SQ_WAVE_CYCLES"):
            # the issue-side roofline (SURVEY.md section 8d "report VALUBusy alongside GB/s"): SQ_ACTIVE_INST_VALU counts, in 4-clock units, the time a
            # SIMD's vector ALU is issuing for some wave; the chip has 1 024 SIMDs (256 CUs x 4), each with one such slot per 4 clocks at 2.4 GHz
            # (per STEP: a split step is two such launches side by side, a rotation-constrained step two walker + model pairs, a fleet step one launch per bin -
            #  the counters are summed accordingly, and the step lasts about as long as one launch / pair / the longest bin; bench.py divides the same sum by its own clock)
            quads = sq["SQ_ACTIVE_INST_VALU"] * per_step
            slots = 1024.0 * dur[0] * 2.4 / 4.0
            valu = {"valu_issue_share_per_wave": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"], "valu_active_quads_per_step": quads,
                    "valu_issue_frac_by_kernel_trace": quads / slots, "valu_insts_per_wave": sq.get("SQ_INSTS_VALU", 0) / max(sq.get("SQ_WAVES", 1), 1)}
            out.write(f"VALU issue share of the chip = {per_step} x SQ_ACTIVE_INST_VALU / (1 024 SIMDs x kernel duration x 2.4 GHz / 4) = {valu['valu_issue_frac_by_kernel_trace']:.3f} "
                      f"(per wave: {valu['valu_issue_share_per_wave']:.3f} of its lifetime)\n")
    bl = f"{prof}/bench_line.json"
    if os.path.exists(bl):
        out.write("\n== bench.py line of the traced run (rocprofv3 --kernel-trace --stats -- python bench.py --steps 400 --warmup 40 "
                  "--no-cpu-baseline --no-fused-probe)\n" + open(bl).read())
    out.close()
    if traffic or valu:
        tj = os.path.join(os.path.dirname(dest), "traffic.json")
        d = json.load(open(tj)) if os.path.exists(tj) else {}
        if traffic:
            d[key] = round(traffic)
        if valu:
            d[key + "#valu"] = {k: round(v, 4) for k, v in valu.items()}
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from syropod_highlevel_controller_amd.engine import _source_hash
        if d.get("_kernel_source_hash") != _source_hash():  # figures of an older kernel build are dropped, not mixed in
            d = {k: v for k, v in d.items() if k in (key, key + "#valu")}
        d["_kernel_source_hash"] = _source_hash()
        d["_note"] = ("HBM bytes per launch of shc_cycle_kernel from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in KiB, each scaled by the "
                      "factor calibrated on shc_debug_plane_copy in the same run); written by scripts/summarize_prof.py from " + os.path.basename(dest))
        json.dump(d, open(tj, "w"), indent=1)
    print(open(dest).read()[-2500:])
