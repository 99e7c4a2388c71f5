"""Development aid: the rough-terrain / gravity-aligned bench workloads on the feature-exact and on the generic kernels."""
import sys, time
sys.path.insert(0, ".")
import torch
import numpy as np, bench
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_GENERIC_KERNEL, FEAT_SINGLE_STREAM
for name in sys.argv[1:] or ["rough"]:
    n = bench.DEFAULT_INSTANCES[name]
    p, lin, ang, extra, key, desc = bench.make_workload(name, n, 0xC0FFEE)
    fs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in extra.pop("force_sets", [])]
    eng = BatchEngine(p, n)
    bench.apply_inputs(eng, lin, ang, extra)
    for _ in range(30): eng.step(16)
    for feat, fl in ((FEAT_DEFAULT | FEAT_SINGLE_STREAM, "exact, single"), (FEAT_DEFAULT | FEAT_SINGLE_STREAM | FEAT_GENERIC_KERNEL, "generic, single"), (FEAT_DEFAULT, "exact, split"),
                     (FEAT_DEFAULT | FEAT_GENERIC_KERNEL, "generic, split")):
        eng.set_features(feat)
        for _ in range(30): eng.step(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(300):
            if fs and c % 10 == 0:
                eng.L.shc_engine_set_tip_force(eng.h, fs[(c // 10) % 4].data_ptr(), 1)
            eng.step(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300
        print(f"RESULT {name} {fl:16s}: {dt*1e6:8.2f} us/step  frac {bench.ALG_BYTES_PER_CYCLE[key] * n / dt / 8e12:.3f}")
    eng.close()
