"""Mixed-morphology / multi-device batches: thin ctypes view of the C ABI's shc_fleet_* entry points (include/shc_batch.h).

The binning (one engine and one HIP stream per (morphology bin, device)), the contiguous sharding over devices and the
device-to-device all-gather of the joint buffer live in the library (csrc/shc_fleet.hpp); this class only converts numpy
arrays.  Inputs arrive and outputs leave indexed by the caller's instance id, whatever the interleaving pattern.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import engine as _engine
from .params import Params


class MixedFleet:
    def __init__(self, morphologies: Sequence[Params], morph_id, devices: Sequence[int] = (0,)):
        """morphologies[k] describes bin k; morph_id[i] in [0, len(morphologies)) assigns instance i to a bin; every bin is
        sharded over `devices` (repeating a device id gives several shards on that device)."""
        self.L = _engine.lib()
        if self.L.shc_device_count() < 1:
            raise _engine.ShcError("no HIP device visible: the batched engine has no CPU fallback")
        self.morph_id = np.ascontiguousarray(morph_id, dtype=np.int32)
        self.n = len(self.morph_id)
        self.params = list(morphologies)
        arr = (Params * len(self.params))(*self.params)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        _engine._check(self.L.shc_fleet_create(arr, len(self.params), self.morph_id.ctypes.data_as(C.c_void_p), self.n,
                                               dev.ctypes.data_as(C.c_void_p), len(dev), C.byref(h)), "shc_fleet_create")
        self.h, self.n_devices = h, len(dev)
        a, b = C.c_int(), C.c_int()
        _engine._check(self.L.shc_fleet_shape(self.h, C.byref(a), C.byref(b)), "shc_fleet_shape")
        self.max_legs, self.max_dof = a.value, b.value

    def close(self):
        if getattr(self, "h", None):
            self.L.shc_fleet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parts(self):
        """[(engine handle, morphology, device, instance ids)] of every (bin, device) part."""
        out = []
        for k in range(self.L.shc_fleet_part_count(self.h)):
            e, m, d, n = C.c_void_p(), C.c_int(), C.c_int(), C.c_int64()
            _engine._check(self.L.shc_fleet_part(self.h, k, C.byref(e), C.byref(m), C.byref(d), C.byref(n)), "shc_fleet_part")
            ids = np.zeros(n.value, dtype=np.int64)
            _engine._check(self.L.shc_fleet_part_instances(self.h, k, ids.ctypes.data_as(C.c_void_p)), "shc_fleet_part_instances")
            out.append((e.value, m.value, d.value, ids))
        return out

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def set_velocity(self, linear_xy, angular):
        a, b = np.ascontiguousarray(linear_xy, dtype=np.float64), np.ascontiguousarray(angular, dtype=np.float64)
        _engine._check(self.L.shc_fleet_set_velocity(self.h, self._p(a), self._p(b)), "shc_fleet_set_velocity")

    def set_joint_effort(self, effort_padded):
        """effort_padded [n][max_legs][max_dof]; entries beyond a bin's (legs, dof) are ignored."""
        a = np.ascontiguousarray(effort_padded, dtype=np.float64)
        assert a.shape == (self.n, self.max_legs, self.max_dof)
        _engine._check(self.L.shc_fleet_set_joint_effort(self.h, self._p(a)), "shc_fleet_set_joint_effort")

    def step(self, n_cycles: int = 1):
        _engine._check(self.L.shc_fleet_step(self.h, int(n_cycles)), "shc_fleet_step")

    def synchronize(self):
        _engine._check(self.L.shc_fleet_synchronize(self.h), "shc_fleet_synchronize")

    def joints(self):
        q = np.zeros((self.n, self.max_legs, self.max_dof))
        qd = np.zeros_like(q)
        _engine._check(self.L.shc_fleet_get_joint_state(self.h, self._p(q), self._p(qd)), "shc_fleet_get_joint_state")
        return q, qd

    def walk_state(self):
        ws = np.zeros(self.n, dtype=np.int32)
        _engine._check(self.L.shc_fleet_get_walk_state(self.h, self._p(ws)), "shc_fleet_get_walk_state")
        return ws

    def all_gather_joints(self):
        """Device pointers (one per device slot) of the gathered [n][max_legs][max_dof] joint buffers."""
        bufs = (C.c_void_p * self.n_devices)()
        _engine._check(self.L.shc_fleet_all_gather_joints(self.h, bufs), "shc_fleet_all_gather_joints")
        return [b for b in bufs]
