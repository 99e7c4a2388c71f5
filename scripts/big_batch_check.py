import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
for name, p in (("hexapod", default_hexapod_params("tripod")), ("octopod", synthetic_octopod_params("ripple", 5, 8))):
    n = 1 << 20
    rng = np.random.default_rng(1)
    t0 = time.perf_counter()
    eng = BatchEngine(p, n)
    lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    lin[-1000:], ang[-1000:] = lin[:1000], ang[:1000]
    eng.set_velocity(lin, ang)
    eng.step(200); eng.synchronize()
    t1 = time.perf_counter()
    for _ in range(20): eng.step(1)
    eng.synchronize()
    t2 = time.perf_counter()
    q, _ = eng.joints()
    ws = eng.body_state()[2]
    print(name, n, "setup+200 cycles %.2f s" % (t1 - t0), "single-cycle launch %.1f us -> %.3e cycles/s" % ((t2 - t1) / 20 * 1e6, n * 20 / (t2 - t1)),
          "finite", bool(np.isfinite(q).all()), "moving", float((ws == 1).mean()), "dup equal", bool(np.array_equal(q[:1000], q[-1000:])))
    eng.close()
