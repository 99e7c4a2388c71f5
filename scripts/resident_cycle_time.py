"""Development aid: time per cycle of the resident loop on bench.py's headline configuration (4 096 hexapods, joint efforts live), one
launch of K cycles released at once, plus a checksum of the final joints (variants of the library must agree byte for byte).
usage: [SHC_LIB=path] python scripts/resident_cycle_time.py [instances] [cycles] [config2|config3|octopod|rough|gravity]"""
import hashlib
import sys
import time

import numpy as np
import torch  # noqa: F401  (before the engine library)

sys.path.insert(0, ".")
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params  # noqa: E402
from syropod_highlevel_controller_amd.engine import BatchEngine  # noqa: E402
from syropod_highlevel_controller_amd.parallel import velocity_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
case = sys.argv[3] if len(sys.argv) > 3 else "config2"
if case == "octopod":
    p = synthetic_octopod_params("ripple", 5, 8)
elif case == "gravity":       # gravity-aligned tips on 5-joint legs: tip rotations + the rotation-constrained IK, one wavefront per robot group
    p = synthetic_octopod_params("ripple", 5, 8)
    p.gravity_aligned_tips = 1
elif case == "tipalign":      # gravity-aligned tips on 3-joint legs: the tip-align pose (one wavefront per robot group)
    p = default_hexapod_params("tripod")
    p.gravity_aligned_tips = 1
elif case == "rough":         # rough terrain mode: touchdown detection inside the loop, step-plane targets (one wavefront per robot group)
    p = default_hexapod_params("tripod")
    p.rough_terrain_mode, p.step_depth = 1, 0.012
else:
    p = default_hexapod_params("tripod" if case == "config2" else "wave")
    if case == "config3":
        p.admittance_control, p.imu_posing = 1, 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
lin, ang = velocity_inputs(0xC0FFEE, 0, n)
rng = np.random.default_rng(1)
eng = BatchEngine(p, n)
if case == "rough":   # contact forces as bench.py's rough workload draws them (held while the loop runs)
    f = rng.normal(0, 0.25, (n, 6, 3))
    f[..., 2] += rng.choice([0.0, 0.05, 0.6, 1.5], size=(n, 6), p=[0.3, 0.2, 0.2, 0.3])
    eng.set_tip_force(f)
import os  # noqa: E402
if not os.environ.get("SHC_NO_EFFORTS"):
    eng.set_joint_effort(rng.normal(0, 0.5, (n, p.leg_count * p.leg_dof[0])))
period = eng.tables().step.period
for gk in range(8):   # de-phase as bench.py does
    sel = (np.arange(n) % 8) <= gk
    eng.set_velocity(lin * sel[:, None], ang * sel)
    eng.step(max(1, period // 8))
eng.set_velocity(lin, ang)
eng.step(2 * period + 64)
eng.synchronize()
best = 1e9
for rep in range(3):
    eng.resident_begin(ring_depth=16, max_cycles=K)
    t0 = time.perf_counter()
    eng.resident_publish(K)
    eng.resident_wait(K, 60000)
    dt = time.perf_counter() - t0
    eng.resident_end()
    best = min(best, dt / K)
q, qd = eng.joints()
h = hashlib.sha256(q.tobytes() + qd.tobytes()).hexdigest()[:16]
print(f"{case} n={n}: resident {best * 1e6:.3f} us/cycle ({n / best:.3e} cycles/s), joints sha {h}, finite {bool(np.isfinite(q).all())}")
