"""Mixed-morphology batches (BASELINE.json configs[4]): instances that differ in leg count, joints per leg or gait.

The cycle kernel keeps one morphology's DH / limit tables in LDS and maps one leg to one lane, so a batch is uniform per
engine.  A fleet bins its instances by morphology id (the "sorted / binned variant" of SURVEY.md section 8d), runs one
engine per bin, each on its own HIP stream so that small bins overlap on the GPU, and keeps the caller's instance order at
the boundary: inputs arrive and outputs leave indexed by the caller's instance id, whatever the interleaving pattern.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import engine as _engine
from .params import Params


class MixedFleet:
    def __init__(self, morphologies: Sequence[Params], morph_id, device: int = 0, own_streams: bool = True,
                 device_init: bool = False):
        """morphologies[k] describes bin k; morph_id[i] in [0, len(morphologies)) assigns instance i to a bin.
        device_init: run the init chain of all bins as one batch of HIP kernels (shc_generate_tables_batch) instead of
        ~1 ms of host time per bin; the start-up joint configuration then agrees with the host's to ~1e-6 rad only (the
        reference's start-up iteration amplifies rounding differences)."""
        self.morph_id = np.asarray(morph_id, dtype=np.int64)
        self.n = len(self.morph_id)
        if self.n == 0 or self.morph_id.min() < 0 or self.morph_id.max() >= len(morphologies):
            raise ValueError("morph_id out of range")
        self.device = device
        self.L = _engine.lib()
        self.params = list(morphologies)
        self.index = [np.nonzero(self.morph_id == k)[0] for k in range(len(morphologies))]  # instance ids of bin k, ascending
        self.streams, self.engines = [], []
        # init chain (start-up solve, workspace search, walkspace, limits) of every bin at once on the GPU
        tables, status = _engine.generate_tables_batch(self.params, device) if device_init else (None, None)
        for k, idx in enumerate(self.index):
            if len(idx) == 0:
                self.streams.append(None)
                self.engines.append(None)
                continue
            s = C.c_void_p(0)
            if own_streams:
                _engine._check(self.L.shc_stream_create(device, C.byref(s)), "shc_stream_create")
            self.streams.append(s)
            if status is not None and status[k] != 0:
                raise _engine.ShcError(f"morphology {k} rejected by the init chain (code {status[k]})")
            self.engines.append(_engine.BatchEngine(morphologies[k], len(idx), device, s.value or 0,
                                                    tables=None if tables is None else tables[k]))
        self.max_legs = max(p.leg_count for p in self.params)
        self.max_dof = max(p.leg_dof[0] for p in self.params)

    def close(self):
        for e in self.engines:
            if e is not None:
                e.close()
        for s in self.streams:
            if s is not None and s.value:
                self.L.shc_stream_destroy(self.device, s)
        self.engines, self.streams = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _bins(self):
        return [(k, e, self.index[k]) for k, e in enumerate(self.engines) if e is not None]

    # ---- inputs in the caller's instance order
    def set_velocity(self, linear_xy, angular):
        for _, e, idx in self._bins():
            e.set_velocity(np.ascontiguousarray(linear_xy[idx]), np.ascontiguousarray(angular[idx]))

    def set_joint_effort(self, effort_padded):
        """effort_padded [n][max_legs][max_dof]; entries beyond a bin's (legs, dof) are ignored."""
        for k, e, idx in self._bins():
            p = self.params[k]
            e.set_joint_effort(np.ascontiguousarray(effort_padded[idx][:, :p.leg_count, :p.leg_dof[0]].reshape(len(idx), -1)))

    # ---- stepping: every bin advances n_cycles on its own stream; nothing orders one bin against another
    def step(self, n_cycles: int = 1):
        for _, e, _ in self._bins():
            e.step(n_cycles)

    def synchronize(self):
        for _, e, _ in self._bins():
            e.synchronize()

    # ---- outputs in the caller's instance order, NaN-padded to [n][max_legs][max_dof]
    def joints(self):
        q = np.full((self.n, self.max_legs, self.max_dof), np.nan)
        qd = np.full_like(q, np.nan)
        for k, e, idx in self._bins():
            p = self.params[k]
            a, b = e.joints()
            q[idx, :p.leg_count, :p.leg_dof[0]] = a.reshape(len(idx), p.leg_count, p.leg_dof[0])
            qd[idx, :p.leg_count, :p.leg_dof[0]] = b.reshape(len(idx), p.leg_count, p.leg_dof[0])
        return q, qd

    def walk_state(self):
        ws = np.zeros(self.n, dtype=np.int32)
        for _, e, idx in self._bins():
            ws[idx] = e.body_state()[2]
        return ws
