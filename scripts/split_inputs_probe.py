import sys, time, os
sys.path.insert(0, ".")
import torch
torch.cuda.init()
import numpy as np, bench
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_SINGLE_STREAM
name = "config3"
n = bench.DEFAULT_INSTANCES[name]
p, lin, ang, extra, key, desc = bench.make_workload(name, n, 0xC0FFEE)
fs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in extra.pop("force_sets")]
for variant in ("plain", "force every 10", "dephased", "dephased + force"):
    eng = BatchEngine(p, n, stream=0)
    bench.apply_inputs(eng, lin, ang, extra)
    if "dephased" in variant:
        period = eng.tables().step.period
        for gk in range(8):
            sel = (np.arange(n) % 8) <= gk
            eng.set_velocity(lin * sel[:, None], ang * sel)
            for _ in range(max(1, period // 8)): eng.step(1)
        eng.set_velocity(lin, ang)
    for _ in range(40): eng.step(16)
    for feat, fl in ((FEAT_DEFAULT | FEAT_SINGLE_STREAM, "single"), (FEAT_DEFAULT, "split"), (FEAT_DEFAULT | FEAT_SINGLE_STREAM, "single"), (FEAT_DEFAULT, "split")):
        eng.set_features(feat)
        for _ in range(30): eng.step(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(300):
            if "force" in variant and c % 10 == 0:
                eng.L.shc_engine_set_tip_force(eng.h, fs[(c // 10) % 4].data_ptr(), 1)
            eng.step(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300
        print(f"RESULT {variant:18s} {fl:7s}: {dt*1e6:8.2f} us/step")
    eng.close()
