"""Development aid: launch time of the config-2 cycle kernel against the batch size (waves per CU / load balance)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.parallel import velocity_inputs
p = default_hexapod_params("tripod")
sizes = [int(a) for a in sys.argv[1:]] or [640, 1280, 2560, 3840, 4096, 5120, 7680, 10230, 10240, 20480]
for n in sizes:
    lin, ang = velocity_inputs(1, 0, n)
    eng = BatchEngine(p, n)
    eng.set_velocity(lin, ang)
    eng.set_joint_effort(np.random.default_rng(0).normal(0, .5, (n, 18)))
    eng.step(400); eng.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(2000): eng.step(1)
        eng.synchronize()
        best = min(best, (time.perf_counter() - t0) / 2000)
    print(f"n {n:6d} waves {(n + 9) // 10:5d}: {best * 1e6:6.2f} us/launch  {n / best / 1e6:8.1f} M cycles/s", flush=True)
    eng.close()
