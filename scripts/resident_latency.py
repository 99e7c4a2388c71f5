"""Development aid: round trip of ONE resident cycle (publish 1 tick, wait for it) and of short bursts - the fixed latency of the
doorbell -> relay -> gate -> workers -> progress -> relay -> host path."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.parallel import velocity_inputs
n = 4096
p = default_hexapod_params("tripod")
lin, ang = velocity_inputs(0xC0FFEE, 0, n)
eng = BatchEngine(p, n)
eng.set_velocity(lin, ang)
eng.step(300)
eng.resident_begin(ring_depth=16, max_cycles=200000)
done = 0
for burst in (1, 2, 5, 20, 100):
    ts = []
    for rep in range(200):
        t0 = time.perf_counter()
        for _ in range(burst):
            eng.resident_publish(1)
        done += burst
        eng.resident_wait(done)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"RESULT burst {burst:4d}: median {np.median(ts):8.2f} us  min {ts.min():8.2f}  p90 {np.percentile(ts, 90):8.2f}   per cycle {np.median(ts) / burst:6.2f}")
eng.resident_end()

# the driver's shape: a fresh launch, 5 warm-up cycles, 20 timed ticks
for warm in (5, 50, 500):
    ts = []
    for rep in range(30):
        eng.resident_begin(ring_depth=16, max_cycles=warm + 28)
        eng.resident_publish(warm)
        eng.resident_wait(warm)
        t0 = time.perf_counter()
        for _ in range(20):
            eng.resident_publish(1)
        eng.resident_wait(warm + 20)
        ts.append(time.perf_counter() - t0)
        eng.resident_end()
    ts = np.array(ts) * 1e6
    print(f"RESULT fresh launch, warm-up {warm:3d}, 20 ticks: median {np.median(ts):7.2f} us  min {ts.min():7.2f}  max {ts.max():7.2f}  first {ts[0]:7.2f}")
