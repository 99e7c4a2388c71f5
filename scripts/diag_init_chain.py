import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params, synthetic_octopod_params, engine
rng = np.random.default_rng(91)
plist = []
for k in range(48):
    if k % 4 == 3:
        p = synthetic_octopod_params(["ripple", "wave", "tripod"][k % 3], 3 + k % 3, [4, 6, 8][(k // 4) % 3])
    else:
        p = default_hexapod_params(["tripod", "wave", "ripple", "amble"][k % 4])
    for l in range(p.leg_count):
        for j in range(1, p.leg_dof[l] + 1):
            p.link[l][j].r *= 1.0 + rng.uniform(-0.08, 0.08)
        p.stance_position[l][0] *= 1.0 + rng.uniform(-0.05, 0.05)
        p.stance_position[l][1] *= 1.0 + rng.uniform(-0.05, 0.05)
    p.body_clearance *= 1.0 + rng.uniform(-0.1, 0.1)
    p.step_frequency = [1.0, 0.8, 1.25][k % 3]
    plist.append(p)
import time
engine.generate_tables_batch(plist[:2])  # warm-up: code-object load
t0 = time.perf_counter(); tables, status = engine.generate_tables_batch(plist); t1 = time.perf_counter()
hs = [engine.generate_tables(p) for p in plist]; t2 = time.perf_counter()
print(f"device batch {1e3*(t1-t0):.1f} ms, host {1e3*(t2-t1):.1f} ms for {len(plist)} morphologies")
big = plist * 20
t3 = time.perf_counter(); engine.generate_tables_batch(big); t4 = time.perf_counter()
print(f"device batch of {len(big)}: {1e3*(t4-t3):.1f} ms ({1e3*(t4-t3)/len(big):.3f} ms each; host {1e3*(t2-t1)/len(plist):.3f} ms each)")
for k, (p, t, h) in enumerate(zip(plist, tables, hs)):
    L, D = p.leg_count, p.leg_dof[0]
    dq = np.abs(np.array(t.default_joint_position)[:L, :D] - np.array(h.default_joint_position)[:L, :D]).max()
    dw = np.abs(np.array(t.workspace_radius)[:L] - np.array(h.workspace_radius)[:L]).max()
    dl = np.abs(np.array(t.max_linear_speed) - np.array(h.max_linear_speed)).max()
    print(k, L, D, f"dq {dq:.2e} dwork {dw:.2e} dlim {dl:.2e}")
