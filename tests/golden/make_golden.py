"""Generates tests/golden/math_golden.json — known answers for the CPU oracle from INDEPENDENT implementations
(scipy.spatial.transform, numpy.linalg, scipy.linalg.expm, hand-derived step-cycle integers), produced in the build
container.  The reference ships no golden vectors and cannot be compiled here; these are the pins.

Run:  python tests/golden/make_golden.py      (writes the JSON next to this script)
"""
import json
import os

import numpy as np
from scipy.linalg import expm
from scipy.spatial.transform import Rotation as R, Slerp

rng = np.random.default_rng(20260928)
out = {}


def wxyz(rot):
    q = rot.as_quat()
    return [q[3], q[0], q[1], q[2]]


# ---- Euler <-> quaternion (standard_includes.h:227-291).  Extrinsic roll/pitch/yaw == scipy 'xyz'; intrinsic == 'XYZ'.
cases = []
for _ in range(40):
    e = [rng.uniform(-3.0, 3.0), rng.uniform(-1.5, 1.5), rng.uniform(-3.0, 3.0)]
    cases.append({"euler": e, "quat_extrinsic": wxyz(R.from_euler("xyz", e)), "quat_intrinsic": wxyz(R.from_euler("XYZ", e))})
out["euler"] = cases

# ---- FromTwoVectors: shortest-arc rotation a -> b:  q = normalise([1 + a.b, a x b]) for unit a, b
cases = []
for _ in range(20):
    a, b = rng.normal(size=3), rng.normal(size=3)
    ua, ub = a / np.linalg.norm(a), b / np.linalg.norm(b)
    q = np.concatenate([[1.0 + ua @ ub], np.cross(ua, ub)])
    q /= np.linalg.norm(q)
    cases.append({"a": a.tolist(), "b": b.tolist(), "quat": q.tolist()})
out["from_two_vectors"] = cases

# ---- slerp
cases = []
for _ in range(20):
    r0, r1 = R.random(random_state=int(rng.integers(1 << 30))), R.random(random_state=int(rng.integers(1 << 30)))
    t = float(rng.uniform(0, 1))
    q0, q1 = np.array(wxyz(r0)), np.array(wxyz(r1))
    if q0 @ q1 < 0:  # Eigen's slerp takes the short way by flipping the sign of the second weight
        ref = Slerp([0, 1], R.concatenate([r0, r1]))(t)
    else:
        ref = Slerp([0, 1], R.concatenate([r0, r1]))(t)
    cases.append({"a": q0.tolist(), "b": q1.tolist(), "t": t, "rotmat": ref.as_matrix().tolist()})
out["slerp"] = cases

# ---- quaternion from rotation matrix
cases = []
for _ in range(20):
    r = R.random(random_state=int(rng.integers(1 << 30)))
    cases.append({"m": r.as_matrix().reshape(-1).tolist(), "quat": wxyz(r)})
out["quat_from_matrix"] = cases

# ---- general inverse (partial-pivot LU in the reference) vs numpy.linalg.inv
cases = []
for n in (3, 4, 5, 6):
    for _ in range(5):
        j = rng.normal(size=(n, 3)) * 0.1
        a = j @ j.T + 0.02 ** 2 * np.eye(n)
        cases.append({"n": n, "a": a.reshape(-1).tolist(), "inv": np.linalg.inv(a).reshape(-1).tolist()})
out["inverse"] = cases


# ---- DH matrix + FK of the default.yaml hexapod legs (numpy chain product)
def dh(d, th, r, al):
    c, s, ca, sa = np.cos(th), np.sin(th), np.cos(al), np.sin(al)
    return np.array([[c, -s * ca, s * sa, r * c], [s, c * ca, -c * sa, r * s], [0, sa, ca, d], [0, 0, 0, 1]])


HEX_BASE_THETA = [-0.523, -1.571, -2.617, 2.617, 1.571, 0.523]
HEX_LINKS = [(0.0, 0.0, 0.050, 1.571), (0.0, 0.0, 0.050, 0.0), (0.0, -0.100, 0.100, 0.0)]  # d theta r alpha (coxa, femur, tibia)
HEX_JOINTS = [(-0.55, 0.55, 5.0), (-1.5, 1.5, 5.0), (-2.355, -0.1, 5.0)]
cases = []
for _ in range(24):
    leg = int(rng.integers(6))
    q = [rng.uniform(-0.5, 0.5), rng.uniform(-1.0, 1.0), rng.uniform(-2.2, -0.2)]
    t = dh(0.0, HEX_BASE_THETA[leg], 0.05, 0.0)
    for k, (d, th, r, al) in enumerate(HEX_LINKS):
        t = t @ dh(d, th + q[k], r, al)
    cases.append({"leg": leg, "q": q, "tip": t[:3, 3].tolist(), "quat": wxyz(R.from_matrix(t[:3, :3]))})
out["hexapod_fk"] = cases


# ---- one DLS IK step (model.cpp:726-857) in numpy, literally the reference's 6x6 formulation
def ik_step(leg, q, qd, desired, dt=0.02, clamp_vel=False):
    t1 = dh(0.0, HEX_BASE_THETA[leg], 0.05, 0.0)
    ts = [dh(d, th + q[k], r, al) for k, (d, th, r, al) in enumerate(HEX_LINKS)]
    c1 = ts[0]
    c2 = c1 @ ts[1]
    c3 = c2 @ ts[2]
    pe = c3[:3, 3]
    z = [np.array([0, 0, 1.0]), c1[:3, 2], c2[:3, 2]]
    p = [np.zeros(3), c1[:3, 3], c2[:3, 3]]
    jac = np.zeros((6, 3))
    for i in range(3):
        jac[:3, i] = np.cross(z[i], pe - p[i])
    cur = (t1 @ c3)[:3, 3]
    t1i = np.linalg.inv(t1)
    delta = np.zeros(6)
    delta[:3] = (t1i @ np.append(desired, 1))[:3] - (t1i @ np.append(cur, 1))[:3]
    jinv = jac.T @ np.linalg.inv(jac @ jac.T + 0.02 ** 2 * np.eye(6))
    w = 0.1
    pg, vg, pc, vc = np.zeros(3), np.zeros(3), 0.0, 0.0
    for i, (mn, mx, mv) in enumerate(HEX_JOINTS):
        rg, cen = mx - mn, mn + (mx - mn) / 2
        pc += (w * (q[i] - cen) / rg) ** 2
        pg[i] = -w * w * (q[i] - cen) / rg ** 2
        vc += (w * qd[i] / (2 * mv)) ** 2
        vg[i] = -w * w * qd[i] / (2 * mv) ** 2
    pg *= 0 if pc == 0 else 1 / np.sqrt(pc)
    vg *= 0 if vc == 0 else 1 / np.sqrt(vc)
    g = 0.25 * pg + 0.75 * vg
    dq = jinv @ delta + (np.eye(3) - jinv @ jac) @ g
    v = dq / dt
    qn = np.array(q) + v * dt
    for i, (mn, mx, mv) in enumerate(HEX_JOINTS):
        qn[i] = min(max(qn[i], mn), mx)
    return qn, v


cases = []
for _ in range(24):
    leg = int(rng.integers(6))
    q = [rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5), rng.uniform(-2.0, -1.0)]
    qd = (rng.normal(size=3) * 0.5).tolist()
    t = dh(0.0, HEX_BASE_THETA[leg], 0.05, 0.0)
    for k, (d, th, r, al) in enumerate(HEX_LINKS):
        t = t @ dh(d, th + q[k], r, al)
    desired = (t[:3, 3] + rng.normal(size=3) * 0.003).tolist()
    qn, v = ik_step(leg, q, qd, np.array(desired))
    cases.append({"leg": leg, "q": q, "qd": qd, "desired": desired, "q_out": qn.tolist(), "qd_out": v.tolist()})
out["hexapod_ik_step"] = cases

# ---- admittance: 30 RK4 steps (numpy, literal) and the exact matrix-exponential solution (RK4 truncation ~1e-9)
m, k, zeta, T, gain = 10.0, 12.0, 0.8, 0.5, 0.1
c = zeta * 2 * np.sqrt(m * k)
A = np.array([[0, 1], [-k / m, -c / m]])
cases = []
for _ in range(10):
    x0 = rng.normal(size=2) * 0.02
    f = [rng.normal(), rng.normal(), rng.uniform(0, 20)]
    x = x0.copy()
    xe = x0.copy()
    deltas, deltas_exact = [], []
    for fi in f:
        u = max(fi * gain, 0.0)
        b = np.array([0.0, -u / m])
        h = T / 30
        for _s in range(30):
            k1 = A @ x + b
            k2 = A @ (x + 0.5 * h * k1) + b
            k3 = A @ (x + 0.5 * h * k2) + b
            k4 = A @ (x + h * k3) + b
            x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        deltas.append(float(np.clip(-x[0], -0.2, 0.2)))
        # exact: x(T) = e^{AT} x0 + A^-1 (e^{AT} - I) b
        eat = expm(A * T)
        xe = eat @ xe + np.linalg.solve(A, (eat - np.eye(2)) @ b)
        deltas_exact.append(float(np.clip(-xe[0], -0.2, 0.2)))
    cases.append({"x0": x0.tolist(), "force": f, "state": x.tolist(), "delta": deltas, "state_exact": xe.tolist(), "delta_exact": deltas_exact})
out["admittance"] = cases

# ---- quartic Bezier and derivative via the Bernstein basis
from scipy.special import comb
cases = []
for _ in range(10):
    nodes = rng.normal(size=(5, 3))
    t = float(rng.uniform(0, 1))
    b = sum(comb(4, i) * (1 - t) ** (4 - i) * t ** i * nodes[i] for i in range(5))
    db = sum(4 * comb(3, i) * (1 - t) ** (3 - i) * t ** i * (nodes[i + 1] - nodes[i]) for i in range(4))
    cases.append({"nodes": nodes.reshape(-1).tolist(), "t": t, "b": b.tolist(), "db": db.tolist()})
out["bezier"] = cases

# ---- hand-derived step-cycle integers for default.yaml (step_frequency 1.0, time_delta 0.02)
out["step_cycle"] = {
    "tripod": {"period": 104, "stance_end": 26, "swing_start": 26, "swing_end": 78, "stance_start": 78, "stance_period": 52,
               "swing_period": 52, "frequency": 1.0 / (104 * 0.02), "phase_offset": [0, 52, 0, 52, 0, 52]},
    "wave": {"period": 312, "stance_end": 130, "swing_start": 130, "swing_end": 182, "stance_start": 182, "stance_period": 260,
             "swing_period": 52, "frequency": 1.0 / (312 * 0.02), "phase_offset": [104, 156, 208, 52, 0, 260]},
    "ripple": {"period": 156, "stance_end": 52, "swing_start": 52, "swing_end": 104, "stance_start": 104, "stance_period": 104,
               "swing_period": 52, "frequency": 1.0 / (156 * 0.02), "phase_offset": [52, 0, 104, 26, 78, 130]},
    "amble": {"period": 150, "stance_end": 50, "swing_start": 50, "swing_end": 100, "stance_start": 100, "stance_period": 100,
              "swing_period": 50, "frequency": 1.0 / (150 * 0.02), "phase_offset": [50, 100, 0, 50, 100, 0]},
}

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "math_golden.json")
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote", path, os.path.getsize(path), "bytes")
