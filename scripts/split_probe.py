"""Development aid: steps of a large batch as one launch vs the engine's two-stream split.  usage: split_probe.py config3|config4 [steps]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import bench  # noqa: E402
from syropod_highlevel_controller_amd.engine import BatchEngine  # noqa: E402
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_SINGLE_STREAM  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = bench.DEFAULT_INSTANCES[name]
p, lin, ang, extra, key, desc = bench.make_workload(name, n, 0xC0FFEE)
eng = BatchEngine(p, n)
bench.apply_inputs(eng, lin, ang, extra)
for _ in range(30):
    eng.step(16)
for feat, label in ((FEAT_DEFAULT | FEAT_SINGLE_STREAM, "single launch"), (FEAT_DEFAULT, "two-stream split"), (FEAT_DEFAULT | FEAT_SINGLE_STREAM, "single launch"),
                    (FEAT_DEFAULT, "two-stream split")):
    eng.set_features(feat)
    for _ in range(30):
        eng.step(1)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(1)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name} {label:18s}: {dt * 1e6:8.2f} us/step  {n / dt:.3e} cycles/s  frac {bench.ALG_BYTES_PER_CYCLE[key] * n / dt / 8e12:.3f}")
