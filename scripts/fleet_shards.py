"""Experiment: shc_fleet with several shards of one bin on the SAME device (each shard = one engine on its own HIP stream; no join
between steps): the tail of one shard's launch overlaps the head of the other's.  usage: python scripts/fleet_shards.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from syropod_highlevel_controller_amd import synthetic_octopod_params
from syropod_highlevel_controller_amd.fleet import MixedFleet
from syropod_highlevel_controller_amd.parallel import velocity_inputs

def run(label, morphs, mid, shards, steps=200):
    fleet = MixedFleet(morphs, mid, devices=(0,) * shards)
    lin, ang = velocity_inputs(0x5EED5, 0, len(mid))
    fleet.set_velocity(lin, ang)
    for _ in range(45):
        fleet.step(16)
    fleet.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            fleet.step(1)
        fleet.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(f"{label}: {shards} shard(s) per bin: {best * 1e6:.1f} us per fleet step, {len(mid) / best:.3e} cycles/s", flush=True)
    fleet.close()

n5 = 1 << 20
morphs5 = [synthetic_octopod_params(g, d, l) for l, d, g in bench.CONFIG5_BINS]
for shards in (1, 2, 3):
    run("config5 (5 morphologies)", morphs5, np.arange(n5) % 5, shards, steps=100)
p4 = synthetic_octopod_params("ripple", 5, 8)
for shards in (1, 2, 4):
    run("config4 share (131072 8x5 octopods)", [p4], np.zeros(131072, dtype=np.int32), shards)
