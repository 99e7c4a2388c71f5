"""Development aid: dynamic VALU / SALU / LDS instruction counts per phase of the cycle kernel.
Run under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES with the -DSHC_ABLATE library; then
`python scripts/ablate_counts.py parse <counter_collection.csv>` differences consecutive ablation levels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LEVELS = [(0, 3, "full"), (0, 2, "no tipforce"), (1, 2, "-pose"), (3, 2, "-limits"), (7, 2, "-stepper"), (15, 2, "-ik"), (31, 2, "-fk sincos"),
          (31 + 512, 2, "-2nd chain + tip"), (31 + 512 + 256, 2, "-stance"), (31 + 512 + 256 + 1024, 2, "-odometry"),
          (31 + 512 + 256 + 1024 + 32, 2, "-velocity"), (31 + 512 + 256 + 1024 + 32 + 128, 2, "-predicates (floor)")]
SINGLES = 20

if len(sys.argv) > 1 and sys.argv[1] == "parse":
    import csv, collections
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "shc_cycle_kernel" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)
    per = 2 + SINGLES
    prev = None
    for k, (_, _, name) in enumerate(LEVELS):
        chunk = ids[k * per + 2:(k + 1) * per]
        w = by[chunk[0]]["SQ_WAVES"]
        v = sum(by[i]["SQ_INSTS_VALU"] for i in chunk) / len(chunk) / w
        s = sum(by[i]["SQ_INSTS_SALU"] for i in chunk) / len(chunk) / w
        l = sum(by[i]["SQ_INSTS_LDS"] for i in chunk) / len(chunk) / w
        d = "" if prev is None else f"   delta VALU {prev[0] - v:7.0f} SALU {prev[1] - s:6.0f} LDS {prev[2] - l:5.0f}"
        print(f"{name:22s} VALU {v:7.0f} SALU {s:6.0f} LDS {l:5.0f}{d}")
        prev = (v, s, l)
    sys.exit(0)

os.environ["SHC_LIB"] = os.path.join(ROOT, "syropod_highlevel_controller_amd", "libshc_ablate.so")
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
n = 4096
p = default_hexapod_params("tripod")
for skip, feat, name in LEVELS:
    os.environ["SHC_DEBUG_SKIP"] = str(skip)
    rng = np.random.default_rng(0)
    eng = BatchEngine(p, n)
    eng.set_features(feat)
    eng.set_velocity(rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n))
    eng.set_joint_effort(rng.normal(0, .5, size=(n, 18)))
    eng.step(300)
    for _ in range(SINGLES): eng.step(1)
    eng.synchronize()
    del eng
