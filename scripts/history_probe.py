"""Development probe: does an engine that existed earlier in the process change the two-stream step time of 131 072 octopods?
usage: python scripts/history_probe.py <variant>   (none | small | small-nostep | small-kept | small-octo | small-wave | small-resident)"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401  (before the HIP library)
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.parallel import velocity_inputs

variant = sys.argv[1] if len(sys.argv) > 1 else "none"
kept = []
if variant != "none":
    p = synthetic_octopod_params("ripple", 5, 8) if variant == "small-octo" else default_hexapod_params("wave" if variant == "small-wave" else "tripod")
    n = 64
    lin, ang = velocity_inputs(1, 0, n)
    e = BatchEngine(p, n, device=0, stream=0)
    e.set_velocity(lin, ang)
    if variant != "small-nostep":
        e.step(40)
    e.synchronize()
    if variant == "small-kept":
        kept.append(e)
    else:
        e.close()

p = synthetic_octopod_params("ripple", 5, 8)
n = 131072
lin, ang = velocity_inputs(2, 0, n)
e = BatchEngine(p, n, device=0, stream=0)
e.set_velocity(lin, ang)
for _ in range(20):
    e.step(16)
for _ in range(5):
    e.step(1)
e.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300):
        e.step(1)
    e.synchronize()
    dt = (time.perf_counter() - t0) / 300
    print(variant, "rep", rep, "%.2f us per step" % (dt * 1e6), flush=True)
e.close()
