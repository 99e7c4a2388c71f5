"""Development aid: resident mode vs one launch per cycle on BASELINE.json config 2 (4 096 hexapods) and friends.
usage: python scripts/resident_bench.py [instances] [steps]"""
import sys
import time

import numpy as np
import torch  # (before the engine library: torch brings its own HIP runtime)

sys.path.insert(0, ".")
from syropod_highlevel_controller_amd import default_hexapod_params  # noqa: E402
from syropod_highlevel_controller_amd.engine import BatchEngine  # noqa: E402
from syropod_highlevel_controller_amd.parallel import velocity_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
p = default_hexapod_params("tripod")
lin, ang = velocity_inputs(0xC0FFEE, 0, n)
eng = BatchEngine(p, n)
eng.set_velocity(lin, ang)
eng.step(300)
eng.synchronize()
for _ in range(200):
    eng.step(1)
eng.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.step(1)
eng.synchronize()
dt = time.perf_counter() - t0
print(f"launch per cycle : {dt / steps * 1e6:7.2f} us/cycle  {n * steps / dt:.3e} cycles/s")
t0 = time.perf_counter()
for _ in range(steps):
    eng.set_velocity(lin, ang)   # host arrays every cycle (the velocity callback of every loop iteration)
    eng.step(1)
eng.synchronize()
dt = time.perf_counter() - t0
print(f"launch per cycle, velocities set from host arrays every cycle : {dt / steps * 1e6:7.2f} us/cycle  {n * steps / dt:.3e} cycles/s")
d_lin, d_ang = torch.from_numpy(lin).cuda(), torch.from_numpy(ang).cuda()
torch.cuda.synchronize()
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_RESIDENT_ONE_WAVE  # noqa: E402
for mode in ("publish_all", "publish_each", "post_each", "post_each_device", "post_and_publish", "post_and_publish_device", "one_wave publish_each"):
    eng.set_features(FEAT_DEFAULT | (FEAT_RESIDENT_ONE_WAVE if mode.startswith("one_wave") else 0))
    eng.resident_begin(ring_depth=16, max_cycles=steps + 300)
    eng.resident_publish(200)
    eng.resident_wait(200)
    t0 = time.perf_counter()
    if mode == "publish_all":
        eng.resident_publish(steps)
    elif mode.endswith("publish_each"):
        for _ in range(steps):
            eng.resident_publish(1)
    elif mode == "post_and_publish":   # ... post and doorbell in ONE kernel launch per cycle (shc_cycle_inputs.publish)
        for i in range(steps):
            eng.resident_post(velocity=(lin, ang), publish=True)
    elif mode == "post_and_publish_device":
        for i in range(steps):
            eng.resident_post(velocity=(d_lin.data_ptr(), d_ang.data_ptr()), on_device=True, publish=True)
    elif mode == "post_each_device":   # a new velocity set for every cycle, from arrays resident in HBM
        for i in range(steps):
            eng.resident_post(velocity=(d_lin.data_ptr(), d_ang.data_ptr()), on_device=True)
            eng.resident_publish(1)
    else:                              # ... from host arrays
        for i in range(steps):
            eng.resident_post(velocity=(lin, ang))
            eng.resident_publish(1)
    eng.resident_wait(200 + steps, 60000)
    dt = time.perf_counter() - t0
    ran = eng.resident_end()
    print(f"resident {mode:24s}: {dt / steps * 1e6:7.2f} us/cycle  {n * steps / dt:.3e} cycles/s  ({ran} cycles)")
q, _ = eng.joints()
print("finite", bool(np.isfinite(q).all()))
