import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
p = default_hexapod_params("tripod")
for n in (10, 4096):
    eng = BatchEngine(p, n)
    rng = np.random.default_rng(0)
    eng.set_velocity(rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n))
    eng.step(300); eng.synchronize()
    for cps in (1, 2, 4, 16):
        reps = 1000
        t0 = time.perf_counter()
        for _ in range(reps): eng.step(cps)
        eng.synchronize()
        print(f"n={n} cps={cps}: {1e6*(time.perf_counter()-t0)/reps:.2f} us/launch")
