"""Development aid: replay the soak schedule of tests/test_gpu_parity.py one cycle at a time and report the first divergence."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
gait, seed = sys.argv[1], int(sys.argv[2])
auto = int(sys.argv[3]) if len(sys.argv) > 3 else (1 if seed == 104 else 0)
use_pose = int(sys.argv[4]) if len(sys.argv) > 4 else 1
p = default_hexapod_params(gait)
p.auto_posing = auto
n = 64
rng = np.random.default_rng(seed)
eng, ob = BatchEngine(p, n), OracleBatch(p, n)
effort = rng.normal(0, 0.5, size=(n, 18))
for o in (eng, ob): o.set_joint_effort(effort)
done = 0
while done < 1500:
    lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    stop = rng.random(n) < 0.33
    lin[stop], ang[stop] = 0.0, 0.0
    tv, rv = rng.uniform(-1, 1, size=(n, 3)) * (rng.random((n, 1)) < 0.3), rng.uniform(-1, 1, size=(n, 3)) * (rng.random((n, 1)) < 0.3)
    reset = rng.choice([0, 0, 0, 1, 2, 3, 4, 5], size=n).astype(np.int32)
    if not use_pose:
        tv[:] = 0; rv[:] = 0; reset[:] = 0
    for o in (eng, ob):
        o.set_velocity(lin, ang); o.set_pose_input(tv, rv); o.set_pose_reset_mode(reset)
    ks = (1, int(rng.integers(2, 40)), int(rng.integers(20, 90)))
    for k in ks:
        for _ in range(k):
            eng.step(1); eng.synchronize(); ob.step(1, 8)
            done += 1
            qg, qo = eng.joints()[0], ob.joints()[0]
            pg, vg, wg = eng.body_state(); po, vo, wo = ob.body_state()
            lg, lo = eng.leg_state(), ob.leg_state()
            dq = np.abs(qg - qo).max(axis=1)
            dp = np.abs(pg - po).max(axis=1)
            dt = np.abs(lg["walker_tip"] - lo["walker_tip"]).reshape(n, -1).max(axis=1)
            dpt = np.abs(lg["poser_tip"] - lo["poser_tip"]).reshape(n, -1).max(axis=1)
            st = (lg["leg_status"] != lo["leg_status"]).any(axis=1)
            bad = (dq > 1e-7) | (dp > 1e-9) | (dt > 1e-9) | (wg != wo) | st | (dpt > 1e-8)
            if bad.any():
                i = int(np.nonzero(bad)[0][0])
                print(f"cycle {done}: {bad.sum()} instances differ; first {i}: dq {dq[i]:.3e} dpose {dp[i]:.3e} dtip {dt[i]:.3e} dposer {dpt[i]:.3e} ws {wg[i]}/{wo[i]} status_diff {st[i]}")
                print(" inputs: lin", lin[i], "ang", ang[i], "tv", tv[i], "rv", rv[i], "reset", reset[i])
                print(" pose gpu", pg[i], "\n pose orc", po[i])
                print(" leg_status gpu", lg["leg_status"][i] & 7, (lg["leg_status"][i] >> 8), "\n leg_status orc", lo["leg_status"][i] & 7, (lo["leg_status"][i] >> 8))
                print(" vel", vg[i], vo[i])
                sys.exit(0)
print("no divergence")
