import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import OracleBatch
from syropod_highlevel_controller_amd import synthetic_octopod_params, engine
from syropod_highlevel_controller_amd.engine import BatchEngine
dof, legs, gait = 5, 8, "ripple"
p = synthetic_octopod_params(gait, dof, legs); p.gravity_aligned_tips = 1
p0 = synthetic_octopod_params(gait, dof, legs); p0.gravity_aligned_tips = 0
n = 4
rng = np.random.default_rng(3)
lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
t1 = engine.generate_tables(p)
engA = BatchEngine(p, n)                       # rotation path
engB = BatchEngine(p0, n, tables=t1)           # position-only cycle, same start-up configuration
ob = OracleBatch(p, n)
for o in (engA, engB, ob): o.set_velocity(lin, ang)
prev = ob.joints()[0].copy()
for c in range(1, 6):
    engA.step(1); engB.step(1); engA.synchronize(); engB.synchronize(); ob.step(1, 4)
    qa, qb, qo = engA.joints()[0], engB.joints()[0], ob.joints()[0]
    print(f"cycle {c}: |A-orc| {np.abs(qa-qo).max():.3e}  |B-orc| {np.abs(qb-qo).max():.3e}  |orc step| {np.abs(qo-prev).max():.3e}  |A-B| {np.abs(qa-qb).max():.3e}")
    prev = qo.copy()
