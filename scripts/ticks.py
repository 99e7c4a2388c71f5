"""Development aid: per-phase s_memtime stamps of wave 0 (library built with -DSHC_TIMING)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SHC_LIB"] = os.path.join(ROOT, "syropod_highlevel_controller_amd", "libshc_timing.so")
sys.path.insert(0, ROOT)
import numpy as np
from syropod_highlevel_controller_amd import default_hexapod_params, engine
from syropod_highlevel_controller_amd.engine import BatchEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = default_hexapod_params("tripod")
rng = np.random.default_rng(0)
eng = BatchEngine(p, n)
eng.set_velocity(rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n))
eng.set_joint_effort(rng.normal(0, .5, size=(n, 18)))
L = engine.lib()
buf = (C.c_longlong * 32)()
L.shc_debug_ticks(buf)           # installs the buffer
eng.step(300); eng.synchronize()
names = ["kernel entry", "prologue done (loads+staging+barrier+sincos)", "cycle start", "predicates+gather", "pose", "admittance+limits", "velocity", "fsm",
         "stepper+iterate+plane", "stance", "ik+joints", "sincos+chain", "tip check+tip force", "store_leg issued", "epilogue done"]
acc = np.zeros(15); accp = np.zeros(4)
reps = 20
for _ in range(reps):
    eng.step(1); eng.synchronize()
    L.shc_debug_ticks(buf)
    t = np.array(list(buf)[:15], dtype=np.float64)
    acc += t - t[0]
    accp += np.array(list(buf)[16:20], dtype=np.float64) - t[0]
acc /= reps
accp /= reps
print('prologue: load_leg issued+parked %.0f | consts staged %.0f | tile staged %.0f | after barrier %.0f' % tuple(accp))
prev = 0
for i, nm in enumerate(names):
    print(f"{i:2d} {nm:48s} t={acc[i]:9.0f}  d={acc[i]-prev:8.0f} ticks")
    prev = acc[i]
