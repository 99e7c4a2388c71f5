"""Quick GPU-vs-oracle parity probe (development aid; the graded checks live in tests/)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params
from syropod_highlevel_controller_amd.engine import BatchEngine

def run(name, p, n, cycles, seed=0, imu=False, force=False, check_every=50):
    rng = np.random.default_rng(seed)
    lin = rng.uniform(-1, 1, size=(n, 2)); ang = rng.uniform(-1, 1, size=n)
    ang[::7] = 0.0; lin[::11] = 0.0
    ob = OracleBatch(p, n); eng = BatchEngine(p, n)
    ob.set_velocity(lin, ang); eng.set_velocity(lin, ang)
    L = p.leg_count; D = p.leg_dof[0]
    if imu:
        e = rng.uniform(-0.15, 0.15, size=(n, 3)); e[:, 2] = rng.uniform(-3, 3, size=n)
        from scipy.spatial.transform import Rotation as R
        qx = R.from_euler('xyz', e).as_quat()  # x y z w
        quat = np.stack([qx[:, 3], qx[:, 0], qx[:, 1], qx[:, 2]], axis=1)
        gyro = rng.normal(0, 0.05, size=(n, 3))
        ob.set_imu(quat, gyro); eng.set_imu(quat, gyro)
    if force:
        f = np.stack([rng.normal(0, 1, size=(n, L)), rng.normal(0, 1, size=(n, L)), rng.uniform(0, 20, size=(n, L))], axis=2)
        ob.set_tip_force(f); eng.set_tip_force(f)
    eff = rng.normal(0, 0.5, size=(n, L * D))
    ob.set_joint_effort(eff); eng.set_joint_effort(eff)
    done = 0
    worst = 0.0
    while done < cycles:
        k = min(check_every, cycles - done)
        ob.step(k, threads=8); eng.step(k); eng.synchronize(); done += k
        qo, qdo = ob.joints(); qg, qdg = eng.joints()
        lo, lg = ob.leg_state(), eng.leg_state()
        po, vo, wo = ob.body_state(); pg, vg, wg = eng.body_state()
        dq = np.abs(qo - qg).max(); worst = max(worst, dq)
        print(f"{name} cyc {done:5d} dq {dq:.3e} dqd {np.abs(qdo-qdg).max():.3e} tip {np.abs(lo['walker_tip']-lg['walker_tip']).max():.3e} "
              f"poser {np.abs(lo['poser_tip']-lg['poser_tip']).max():.3e} model {np.abs(lo['model_tip']-lg['model_tip']).max():.3e} "
              f"tf {np.abs(lo['tip_force']-lg['tip_force']).max():.3e} adm {np.abs(lo['admittance']-lg['admittance']).max():.3e} "
              f"status_eq {np.array_equal(lo['leg_status'], lg['leg_status'])} ws_eq {np.array_equal(wo, wg)} "
              f"pose {np.abs(po-pg).max():.3e} vel {np.abs(vo-vg).max():.3e}", flush=True)
    return worst

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "c2"):
        run("hex-tripod", default_hexapod_params("tripod"), 250, 400)
    if which in ("all", "c3"):
        p = default_hexapod_params("wave"); p.admittance_control = 1; p.imu_posing = 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
        run("hex-wave-adm-imu", p, 123, 500, imu=True, force=True)
    if which in ("all", "auto"):
        p = default_hexapod_params("ripple"); p.auto_posing = 1
        run("hex-ripple-auto", p, 77, 500)
    if which in ("all", "c4"):
        run("octo-ripple", synthetic_octopod_params("ripple", 5, 8), 100, 400)
