import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import OracleBatch, OracleRobot
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
p = default_hexapod_params("tripod"); p.auto_posing = 1; p.pose_frequency = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
n = 4
rng = np.random.default_rng(3)
lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
t = OracleRobot(p).tables()
print("tables: pose_phase_length", t.pose_phase_length, "normaliser", t.pose_normaliser, "ref leg", t.auto_pose_reference_leg, "period", t.step.period)
eng, ob = BatchEngine(p, n, tables=t), OracleBatch(p, n)
for o in (eng, ob): o.set_velocity(lin, ang)
for c in range(1, 300):
    eng.step(1); eng.synchronize(); ob.step(1, 4)
    pg, _, wg = eng.body_state(); po, _, wo = ob.body_state()
    dq = np.abs(eng.joints()[0] - ob.joints()[0]).max()
    dp = np.abs(pg - po).max()
    if dp > 1e-9 or dq > 1e-7:
        print(f"cycle {c}: pose diff {dp:.3e} dq {dq:.3e}; walk state {wg} {wo}")
        print(" gpu pose", pg[0]); print(" orc pose", po[0])
        break
else:
    print("no divergence")
