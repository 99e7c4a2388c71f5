"""Development aid: time the fused cycle kernel with phases ablated (SHC_DEBUG_SKIP) and features toggled."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ.setdefault("SHC_LIB", os.path.join(ROOT, "syropod_highlevel_controller_amd", "libshc_ablate.so"))  # built with -DSHC_ABLATE

def one(n, cps, skip, feat):
    os.environ["SHC_DEBUG_SKIP"] = str(skip)
    pass
    from syropod_highlevel_controller_amd import default_hexapod_params
    from syropod_highlevel_controller_amd.engine import BatchEngine
    p = default_hexapod_params("tripod")
    rng = np.random.default_rng(0)
    lin = rng.uniform(-0.7, 0.7, size=(n, 2)); ang = rng.uniform(-1, 1, size=n)
    eng = BatchEngine(p, n)
    eng.set_features(feat)
    eng.set_velocity(lin, ang)
    eng.step(300); eng.synchronize()
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps): eng.step(cps)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt

if __name__ == "__main__":
    n = int(sys.argv[1]); cps = int(sys.argv[2])
    for skip, feat, name in [(0, 1, "full"), (0, 0, "no tipforce"), (1, 0, "-pose"), (3, 0, "-pose-limits"), (7, 0, "-pose-limits-stepper"),
                             (15, 0, "-pose-limits-stepper-ik"), (31, 0, "-all (load/store + fsm only)")]:
        dt = one(n, cps, skip, feat)
        print(f"n={n} cps={cps} {name:32s} {dt*1e6:9.1f} us/launch  {dt*1e6/cps:8.2f} us/cycle", flush=True)
