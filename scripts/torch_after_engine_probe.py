"""Development probe: does torch's lazy HIP initialisation work after the engine has used the device (plain launches / a resident loop)?"""
import sys, os
sys.path.insert(0, ".")
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
mode = sys.argv[1]
p = default_hexapod_params("tripod")
eng = BatchEngine(p, 64)
eng.set_velocity(np.zeros((64, 2)) + 0.3, np.zeros(64))
eng.step(5); eng.synchronize()
if mode == "resident":
    eng.resident_begin(ring_depth=8, max_cycles=100)
    eng.resident_publish(10); eng.resident_wait(10); eng.resident_end()
print({k: v for k, v in os.environ.items() if "VISIBLE" in k or k.startswith("HSA") or k.startswith("HIP")})
import torch
try:
    torch.cuda.init(); print(mode, "torch after engine: OK", torch.cuda.device_count())
except Exception as e:
    print(mode, "torch after engine: FAILED", e)
