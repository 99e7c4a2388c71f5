"""Experiment: one batch as ONE launch vs the same batch as K engines on K HIP streams (waves of different launches interleave on
the SIMDs with unrelated phases).  usage: python scripts/split_streams.py [config3|config4|config2] [K ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from syropod_highlevel_controller_amd.engine import BatchEngine

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
n = int(os.environ.get("SPLIT_N", 0)) or (bench.DEFAULT_INSTANCES[name] if name != "config2" else 65536)
for k in ks:
    part = n // k
    streams = [torch.cuda.Stream() for _ in range(k)]
    engs = []
    for i in range(k):
        p, lin, ang, extra, key, desc = bench.make_workload(name, part, 0xC0FFEE + i, 0)
        e = BatchEngine(p, part, device=0, stream=streams[i].cuda_stream)
        bench.apply_inputs(e, lin, ang, extra)
        engs.append(e)
    for _ in range(40):
        for e in engs:
            e.step(16)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(300):
            for e in engs:
                e.step(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300
    print(f"{name}: {k} engine(s) x {part} instances on {k} stream(s): {dt * 1e6:.1f} us per step of the whole batch, {n / dt:.3e} cycles/s", flush=True)
    for e in engs:
        e.close()
