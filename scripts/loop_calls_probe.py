"""Development aid: wall time per call of the loop-level entry points (the cold paths either side of the control cycle) on 4 096
hexapods - executeSequence START_UP (one thread per robot walks its legs), stepToNewStance, legStateToggle, executePlan - next to one
control cycle.  Each call includes its result read-back (a synchronisation), as a node would use it."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from syropod_highlevel_controller_amd import default_hexapod_params  # noqa: E402
from syropod_highlevel_controller_amd.engine import BatchEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def timed(label, fn, calls):
    fn()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    dt = (time.perf_counter() - t0) / calls
    print(f"{label:58s}: {dt * 1e6:9.1f} us per call  ({n / dt:.3e} robot-loops/s)")


for posing in (False, True):
    p = default_hexapod_params("tripod")
    if posing:
        p.imu_posing = 1
        p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    tag = " [IMU posing: + the cycle kernel's pose pass]" if posing else ""
    eng = BatchEngine(p, n)
    eng.step(5)
    eng.synchronize()
    timed("control cycle (shc_engine_step + synchronize)" + tag, lambda: (eng.step(1), eng.synchronize()), 200)
    sel = np.zeros(n, dtype=np.int32)
    timed("legStateToggle loop (shc_engine_toggle_leg_state)" + tag, lambda: eng.toggle_leg_state(sel), 60)
    eng.close()
    eng = BatchEngine(p, n)
    eng.step(5)
    eng.set_planner_mode(True)
    timed("executePlan loop, waiting (shc_engine_execute_plan)" + tag, lambda: eng.execute_plan(), 100)
    cfg = np.tile(np.array([0.1, -0.2, 0.15]), (n, 6, 1))
    eng.set_target_configuration(cfg.reshape(n, -1) + eng.joints()[0])
    timed("executePlan loop, configuration step" + tag, lambda: eng.execute_plan(), 100)
    eng.close()
p = default_hexapod_params("tripod")
eng = BatchEngine(p, n)
eng.begin_sequence_startup()
timed("executeSequence(START_UP) loop (shc_engine_execute_sequence)", lambda: eng.execute_sequence(0), 200)
eng.close()
eng = BatchEngine(p, n)
eng.step(5)
timed("stepToNewStance loop (shc_engine_step_to_new_stance)", lambda: eng.step_to_new_stance(), 80)
eng.close()
