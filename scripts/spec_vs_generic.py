"""Development check: compile-time specialisations of the cycle kernel against the runtime-flag kernel, teacher-forced (the same
state injected into both every cycle), every state field compared bit for bit.  Prints the first differing fields."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_parity import make_inputs, apply
from syropod_highlevel_controller_amd.engine import BatchEngine
from syropod_highlevel_controller_amd.params import default_hexapod_params, synthetic_octopod_params, InstanceState, LegSnapshot, FEAT_DEFAULT, FEAT_GENERIC_KERNEL


def fields(st, legs):
    out = {}
    raw = np.frombuffer(bytes(st), dtype=np.uint8).reshape(len(st), -1)
    off = 0
    for name, typ in InstanceState._fields_:
        sz = __import__("ctypes").sizeof(typ)
        if name != "leg":
            out[name] = raw[:, off:off + sz]
        else:
            lsz = __import__("ctypes").sizeof(LegSnapshot)
            o2 = 0
            for lname, ltyp in LegSnapshot._fields_:
                s2 = __import__("ctypes").sizeof(ltyp)
                out["leg." + lname] = np.stack([raw[:, off + l * lsz + o2: off + l * lsz + o2 + s2] for l in range(legs)], 1)
                o2 += s2
        off += sz
    return out


def check(label, p, n, cycles, **kw):
    inp = make_inputs(p, n, 41, **kw)
    a, b = BatchEngine(p, n), BatchEngine(p, n)
    b.set_features(FEAT_DEFAULT | FEAT_GENERIC_KERNEL)
    apply(a, inp); apply(b, inp)
    bad = {}
    for c in range(cycles):
        S = a.get_state()
        a.set_state(S); b.set_state(S)
        a.step(1); b.step(1); a.synchronize(); b.synchronize()
        fa, fb = fields(a.get_state(), p.leg_count), fields(b.get_state(), p.leg_count)
        for k in fa:
            if not np.array_equal(fa[k], fb[k]):
                bad.setdefault(k, c)
    print(label, "differing fields (first cycle):", bad or "none", flush=True)
    return not bad


if __name__ == "__main__":
    ok = check("config2 hexapod tripod", default_hexapod_params("tripod"), 40, 230)
    p = default_hexapod_params("wave"); p.admittance_control, p.imu_posing = 1, 1; p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    ok &= check("config3 hexapod wave adm imu", p, 40, 400, imu=True, force=20.0)
    ok &= check("config4 octopod ripple", synthetic_octopod_params("ripple"), 32, 260)
    sys.exit(0 if ok else 1)
