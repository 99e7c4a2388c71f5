import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from syropod_highlevel_controller_amd import synthetic_octopod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
for ga in (0, 1):
    p = synthetic_octopod_params("ripple", 5, 8); p.gravity_aligned_tips = ga
    n = 131072
    rng = np.random.default_rng(1)
    eng = BatchEngine(p, n)
    eng.set_velocity(rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n))
    eng.step(200); eng.synchronize()
    for cps in (1, 16):
        t1 = time.perf_counter()
        for _ in range(20): eng.step(cps)
        eng.synchronize()
        dt = (time.perf_counter() - t1) / 20
        print(f"gravity_aligned={ga} cps={cps}: {dt*1e6:.1f} us/launch {dt*1e6/cps:.1f} us/cycle -> {n*cps/dt:.3e} cycles/s")
