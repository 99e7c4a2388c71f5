#!/usr/bin/env python
"""Which cycle-kernel specialisations a run launched (SHC_KERNEL_LOG=<file> in its environment: one line "form legs joints features" per
specialisation, written by shc_cycle_inst.hip the first time it is launched), grouped by feature word; with a resource report of the build
(engine.build_library(resource_report=...)) also which compiled specialisations the run never touched.

    python scripts/kernels_used.py gpurun_out/kernels_tests.txt [gpurun_out/kernels_bench.txt ...] [--report /tmp/res_report.txt]
"""
import collections
import re
import subprocess
import sys

KIND = {"shc_cycle_kernel": "cycle", "shc_batch_kernel": "batch", "shc_resident_kernel": "resident", "shc_resident2_kernel": "resident2",
        "shc_cycle_half_kernel": "half"}
BITS = [(1 << 31, "DYN"), (1 << 30, "ROT"), (1 << 29, "ROUGH"), (1 << 28, "TALIGN"), (1 << 27, "MLEGS"), (1, "MANUAL"), (2, "AUTO"), (4, "INCL"),
        (8, "IMU"), (16, "ADM"), (32, "TIPF"), (64, "ODOM")]


def feature_names(f):
    return "|".join(n for b, n in BITS if f & b) or "0"


def main(argv):
    report = None
    if "--report" in argv:
        i = argv.index("--report")
        report, argv = argv[i + 1], argv[:i] + argv[i + 2:]
    used = set()
    for path in argv:
        for line in open(path):
            k, legs, joints, f = line.split()
            used.add((k, int(legs), int(joints), int(f)))
    by = collections.defaultdict(list)
    for k, legs, joints, f in sorted(used):
        by[(f, k)].append(f"{legs}x{joints}")
    print(f"{len(used)} specialisations launched")
    for (f, k), v in sorted(by.items()):
        print(f"  {k:10s} {f:#010x} {feature_names(f):32s} {' '.join(v)}")
    if report:
        names = re.findall(r"remark: Function Name: (\S+)", open(report).read())
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        total, unused = collections.Counter(), []
        for d in dem:
            m = re.match(r"void shc::(shc_\w+)<(\d+), (\d+), (\d+)u(?:, (\d))?>", d)
            if not m or m.group(5) == "2":
                continue
            key = (KIND[m.group(1)], int(m.group(2)), int(m.group(3)), int(m.group(4)))
            total[key[0]] += 1
            if key not in used:
                unused.append(key)
        print(f"compiled: {dict(total)}; never launched by this run: {len(unused)}")
        for k, legs, joints, f in sorted(unused, key=lambda u: (u[3], u[0], u[1], u[2])):
            print(f"  {k:10s} {f:#010x} {feature_names(f):32s} {legs}x{joints}")


if __name__ == "__main__":
    main(sys.argv[1:])
