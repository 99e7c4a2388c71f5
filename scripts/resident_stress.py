"""Development aid: many short resident sessions with randomly timed posts and ticks against the same cycles through shc_engine_step - byte for byte
after every session (the handshakes of begin / post / publish / wait / end are what is exercised, not the arithmetic)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.parallel import velocity_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sessions = int(sys.argv[2]) if len(sys.argv) > 2 else 300
p = default_hexapod_params("tripod")
p.admittance_control, p.imu_posing = (1, 1) if len(sys.argv) > 3 else (0, 0)
p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
rng = np.random.default_rng(11)
lin, ang = velocity_inputs(0xC0FFEE, 0, n)
a, b = BatchEngine(p, n), BatchEngine(p, n)
for e in (a, b):
    e.set_velocity(lin, ang)
    e.step(60)
t0 = time.time()
total = 0
for s in range(sessions):
    k = int(rng.integers(1, 120))
    plan = []
    for c in range(k):
        plan.append((lin * rng.uniform(0, 1), ang * rng.uniform(-1, 1)) if rng.random() < 0.3 else None)
    for v in plan:                      # launch mode
        if v is not None:
            a.set_velocity(*v)
        a.step(1)
    b.resident_begin(ring_depth=int(rng.choice([2, 4, 16])), max_cycles=k + 4)
    done = 0
    for c, v in enumerate(plan):
        if v is not None:
            b.resident_post(velocity=v)
        b.resident_publish(1)
        if rng.random() < 0.2:
            b.resident_wait(c + 1, 5000)
        if c > 0 and rng.random() < 0.05:   # publishDesiredJointState of the cycle before: wait for it, read it from the output ring
            b.resident_wait(c, 5000)
            b.resident_joints(c - 1)
    b.resident_wait(k, 10000)
    assert b.resident_end() == k
    sa, sb = bytes(memoryview(a.get_state()).cast("B")), bytes(memoryview(b.get_state()).cast("B"))
    assert sa == sb, f"session {s}: states differ after {k} cycles"
    total += k
print(f"RESULT {sessions} sessions, {total} cycles, {n} robots: byte-identical after every session ({time.time() - t0:.1f} s)")
