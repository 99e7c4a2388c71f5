"""Development aid: what each input group costs shc_engine_step_k (K cycles per launch, batch form of the loop kernel) against shc_engine_step(K) with the
inputs held.  usage: python scripts/step_k_probe.py [config3|config4] [instances] [K]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bench  # noqa: E402
from syropod_highlevel_controller_amd.engine import BatchEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_INSTANCES[name]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
p, lin, ang, extra, key, desc = bench.make_workload(name, n, 0xC0FFEE, 0, False)
eng = BatchEngine(p, n)
bench.apply_inputs(eng, lin * 0.0, ang * 0.0, extra)
period = eng.tables().step.period
for gk in range(8):   # de-phase as bench.py does
    sel = (np.arange(n) % 8) <= gk
    eng.set_velocity(lin * sel[:, None], ang * sel)
    eng.step(max(1, period // 8))
eng.set_velocity(lin, ang)
eng.step(2 * period + 64)
eng.synchronize()
L, D = p.leg_count, p.leg_dof[0]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rows = {"lin": dev(np.repeat(lin[None], K, 0)), "ang": dev(np.repeat(ang[None], K, 0))}
if "imu_q" in extra:
    rows["imu_q"], rows["gyro"] = dev(np.repeat(extra["imu_q"][None], K, 0)), dev(np.repeat(extra["gyro"][None], K, 0))
if "force" in extra:
    rows["force"] = dev(np.repeat(extra["force"][None], K, 0))
rows["effort"] = dev(np.zeros((K, n, L * D)))
ks = np.arange(K)
rows["lin_var"] = dev(lin[None] * (1.0 + 5e-4 * np.cos(0.4 * ks))[:, None, None])
rows["ang_var"] = dev(ang[None] * (1.0 + 5e-4 * np.sin(0.3 * ks))[:, None])
if "force_sets" in extra:
    rows["force_var"] = dev(np.stack([extra["force_sets"][k % len(extra["force_sets"])] for k in range(K)]))
if "gyro" in extra:
    rows["gyro_var"] = dev(extra["gyro"][None] * (1.0 + 0.05 * ks)[:, None, None])
torch.cuda.synchronize()


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize()
    return n * K * reps / (time.perf_counter() - t0)


P = lambda k: rows[k].data_ptr() if k in rows else None
cases = [("shc_engine_step(K), inputs held", lambda: eng.step(K)),
         ("step_k, no inputs (held) - the loop form + K-deep output ring", lambda: eng.step_k(K)),
         ("step_k, velocity rows", lambda: eng.step_k(K, velocity=(P("lin"), P("ang"))))]
if "imu_q" in rows:
    cases.append(("step_k, velocity + IMU rows", lambda: eng.step_k(K, velocity=(P("lin"), P("ang")), imu=(P("imu_q"), P("gyro")))))
if "force" in rows:
    cases.append(("step_k, velocity + IMU + tip-force rows", lambda: eng.step_k(K, velocity=(P("lin"), P("ang")), imu=(P("imu_q"), P("gyro")), tip_force=P("force"))))
    cases.append(("step_k, tip-force rows only", lambda: eng.step_k(K, tip_force=P("force"))))
cases.append(("step_k, velocity rows that change by 5e-4 per cycle", lambda: eng.step_k(K, velocity=(P("lin_var"), P("ang_var")))))
if "force_var" in rows:
    cases.append(("step_k, tip-force rows only, a different random set every cycle", lambda: eng.step_k(K, tip_force=P("force_var"))))
    cases.append(("step_k, IMU rows only, gyro changing", lambda: eng.step_k(K, imu=(P("imu_q"), P("gyro_var")))))
    cases.append(("step_k, everything changing (bench.py's fused-K probe)", lambda: eng.step_k(K, velocity=(P("lin_var"), P("ang_var")), imu=(P("imu_q"), P("gyro_var")), tip_force=P("force_var"))))
cases.append(("shc_engine_step(K), inputs held (again)", lambda: eng.step(K)))
for label, fn in cases:
    print(f"{name} n={n} K={K}: {label:70s} {timed(fn):.3e} cycles/s", flush=True)
eng.close()
