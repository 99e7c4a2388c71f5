// shc_math.hpp — small fixed-size FP64 math for the batched leg-control engine.
//
// Product code (NOT the oracle): compiled by hipcc for gfx950 device code and for the host
// init chain.  Everything is scalar double in registers: the contractions on this path are
// <= 6x6, so there is no MFMA here; one leg lives in one lane.
//
// Reference semantics restated (csiro-robotics/syropod_highlevel_controller v0.5.11):
//   include/syropod_highlevel_controller/standard_includes.h:61-474 (helpers, Bezier, Euler<->quaternion)
//   include/syropod_highlevel_controller/pose.h:112-195 (Pose algebra)
// plus the Eigen 3.3 Quaterniond conventions those rely on (Hamilton product, slerp, FromTwoVectors,
// eulerAngles branch folding).
#pragma once

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#define SHC_HD __host__ __device__ __forceinline__

namespace shc {

constexpr double kPi = 3.14159265358979323846;
constexpr double kUnassigned = 2147483647.0;  // standard_includes.h:52 UNASSIGNED_VALUE
constexpr double kIkTolerance = 0.005;        // model.h:17
constexpr double kDls = 0.02;                 // model.h:19
constexpr double kJointLimitCostWeight = 0.1; // model.h:20
constexpr double kTipTolerance = 0.01;        // pose_controller.h:19
constexpr double kJointTolerance = 0.01;      // pose_controller.h:20
constexpr double kGravity = -9.81;            // standard_includes.h:59

struct V3 {
  double x, y, z;
};
struct Quat {
  double w, x, y, z;
};
struct Pose {
  V3 p;
  Quat r;
};

SHC_HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
SHC_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SHC_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SHC_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
SHC_HD V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
SHC_HD V3 operator*(double s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
// a * s rounded on its own: never contracted into a neighbouring addition (HIP honours the pragma under its default
// -ffp-contract=fast-honor-pragmas).  Used where the compiler's choice of what to fuse would otherwise depend on the surrounding
// code, i.e. differ between two instantiations of the same template (LegStepper control nodes).
SHC_HD V3 scaled(V3 a, double s) {
#pragma clang fp contract(off)
  return V3{a.x * s, a.y * s, a.z * s};
}
SHC_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// per-component select (a ternary on two V3 lvalues selects between their ADDRESSES and keeps whole structs out of registers)
SHC_HD V3 sel3(bool c, V3 a, V3 b) { return V3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
SHC_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
SHC_HD double norm(V3 a) { return sqrt(dot(a, a)); }
// Eigen 3.3 normalized(): unchanged when the squared norm is 0
SHC_HD V3 normalized(V3 a) {
  double z = dot(a, a);
  if (z > 0.0) {
    double s = sqrt(z);
    return V3{a.x / s, a.y / s, a.z / s};
  }
  return a;
}
// standard_includes.h:173 getProjection / :190 getRejection
SHC_HD V3 projection(V3 a, V3 b) {
  double bb = dot(b, b);
  if (dot(a, a) == 0.0 || bb == 0.0) return V3{0, 0, 0};
  return b * (dot(a, b) / bb);
}
SHC_HD V3 rejection(V3 a, V3 b) { return a - projection(a, b); }

SHC_HD double clampd(double v, double lo, double hi) { return fmax(lo, fmin(v, hi)); } // :106
SHC_HD double signd(double v) { return v > 0 ? 1.0 : -1.0; }                             // :88 (sign(0) = -1)
SHC_HD int mod_i(int a, int b) { return (a % b + b) % b; }                               // :76
SHC_HD int round_to_int(double x) { return x >= 0 ? int(x + 0.5) : -int(0.5 - x); }      // :93
SHC_HD int round_to_even_int(double x) { return (int(x) % 2 == 0) ? int(x) : int(x) + 1; } // :98
SHC_HD double smooth_step(double c) {                                                    // :163
  double c3 = c * c * c;
  return c3 * (10.0 + c * (-15.0 + 6.0 * c));
}
SHC_HD double rad2deg(double r) { return (r / (2.0 * kPi)) * 360.0; } // :69
SHC_HD double deg2rad(double d) { return d / 360.0 * 2.0 * kPi; }     // :64

// sin and cos of a joint angle (|x| well below 2^20 * pi/2: DH offsets + joint limits are a few radians).
// Cody-Waite reduction by pi/2 in two parts + the fdlibm kernel polynomials: < 1 ulp, ~45 FP64 instructions, no
// large-argument path (ocml's sincos carries a Payne-Hanek branch and costs ~2x as many issue slots).
template <bool REDUCE = true>
SHC_HD void sincos_joint(double x, double *sn, double *cs) {
  if (!REDUCE) { // caller guarantees |x| <= pi/4: the reduction below would return n = 0, y = x, tail 0
    const double z = x * x, v = z * x;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    *sn = x - ((z * (-(v * rs))) - v * S1);
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z, w1 = 1.0 - hz;
    *cs = w1 + (((1.0 - w1) - hz) + z * rc);
    return;
  }
  const double inv_pio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  const double pio2_1t = 6.07710050650619224932e-11; // pi/2 - pio2_1
  double fn = rint(x * inv_pio2);
  double r = fma(-fn, pio2_1, x);
  double w = fn * pio2_1t;
  double y = r - w;
  double yt = (r - y) - w; // tail of the reduced argument
  int n = int(fn);
  double z = y * y;
  // __kernel_sin
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double v = z * y;
  double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  double s = y - ((z * (0.5 * yt - v * rs) - yt) - v * S1);
  // __kernel_cos
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w1 = 1.0 - hz;
  double c = w1 + (((1.0 - w1) - hz) + (z * rc - y * yt));
  double ss = (n & 1) ? c : s, cc = (n & 1) ? s : c;
  if (n & 2) ss = -ss;
  if ((n + 1) & 2) cc = -cc;
  *sn = ss;
  *cs = cc;
}

// ---------------------------------------------------------------- quaternions (w, x, y, z)
SHC_HD Quat quat(double w, double x, double y, double z) { return Quat{w, x, y, z}; }
SHC_HD Quat quat_identity() { return Quat{1, 0, 0, 0}; }
SHC_HD Quat operator*(Quat a, Quat b) {
  return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
SHC_HD Quat conj(Quat q) { return Quat{q.w, -q.x, -q.y, -q.z}; }
SHC_HD double dot(Quat a, Quat b) { return a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z; }
SHC_HD Quat inverse(Quat q) { // Eigen: conj / |q|^2, zero quaternion stays zero
  double n2 = dot(q, q);
  if (n2 > 0.0) return Quat{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
  return Quat{0, 0, 0, 0};
}
SHC_HD Quat normalized(Quat q) {
  double z = dot(q, q);
  if (z > 0.0) {
    double s = sqrt(z);
    return Quat{q.w / s, q.x / s, q.y / s, q.z / s};
  }
  return q;
}
SHC_HD V3 rotate(Quat q, V3 v) { // QuaternionBase::_transformVector
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + uv * q.w + cross(qv, uv);
}
SHC_HD Quat correct_rotation(Quat t, Quat ref) { // standard_includes.h:211
  return dot(t, ref) < 0.0 ? Quat{-t.w, -t.x, -t.y, -t.z} : t;
}
SHC_HD Quat angle_axis_x(double a) { return Quat{cos(0.5 * a), sin(0.5 * a), 0, 0}; }
SHC_HD Quat angle_axis_y(double a) { return Quat{cos(0.5 * a), 0, sin(0.5 * a), 0}; }
SHC_HD Quat angle_axis_z(double a) { return Quat{cos(0.5 * a), 0, 0, sin(0.5 * a)}; }
// eulerAnglesToQuaternion (standard_includes.h:227): e = (roll, pitch, yaw)
SHC_HD Quat euler_to_quat(V3 e, bool intrinsic) {
  double sx, cx, sy, cy, sz, cz;
  sincos_joint(0.5 * e.x, &sx, &cx);
  sincos_joint(0.5 * e.y, &sy, &cy);
  sincos_joint(0.5 * e.z, &sz, &cz);
  Quat qx{cx, sx, 0, 0}, qy{cy, 0, sy, 0}, qz{cz, 0, 0, sz};
  return intrinsic ? (qx * qy) * qz : (qz * qy) * qx;
}
// Eigen 3.3 eulerAngles(a0,a1,a2) on the rotation matrix of q + the reference's flip fix-up
// (standard_includes.h:248-291).  Returns (roll, pitch, yaw).
SHC_HD V3 quat_to_euler(Quat q, bool intrinsic) {
  double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  double m00 = 1.0 - (tyy + tzz), m01 = txy - twz, m02 = txz + twy;
  double m10 = txy + twz, m11 = 1.0 - (txx + tzz), m12 = tyz - twx;
  double m20 = txz - twy, m21 = tyz + twx, m22 = 1.0 - (txx + tyy);
  double r0, r1, r2;
  if (intrinsic) { // eulerAngles(0,1,2): i=0 j=1 k=2, even permutation
    r0 = atan2(m12, m22);
    double c2 = sqrt(m00 * m00 + m01 * m01);
    if (r0 > 0.0) {
      r0 -= kPi;
      r1 = atan2(-m02, -c2);
    } else {
      r1 = atan2(-m02, c2);
    }
    double s1, c1;
#ifdef SHC_POSE_R5 // (development A/B: the round-5 form)
    sincos(r0, &s1, &c1);
#else
    sincos_joint(r0, &s1, &c1); // |r0| <= pi: the joint-angle sin / cos (< 1 ulp, a third of ocml's sincos with its large-argument path)
#endif
    r2 = atan2(s1 * m20 - c1 * m10, c1 * m11 - s1 * m21);
    r0 = -r0;
    r1 = -r1;
    r2 = -r2;
  } else { // eulerAngles(2,1,0): i=2 j=1 k=0, odd permutation
    r0 = atan2(m10, m00);
    double c2 = sqrt(m22 * m22 + m21 * m21);
    if (r0 < 0.0) {
      r0 += kPi;
      r1 = atan2(-m20, -c2);
    } else {
      r1 = atan2(-m20, c2);
    }
    double s1, c1;
#ifdef SHC_POSE_R5 // (development A/B: the round-5 form)
    sincos(r0, &s1, &c1);
#else
    sincos_joint(r0, &s1, &c1); // |r0| <= pi: the joint-angle sin / cos (< 1 ulp, a third of ocml's sincos with its large-argument path)
#endif
    r2 = atan2(s1 * m02 - c1 * m12, c1 * m11 - s1 * m01);
  }
  if (fabs(r1) > kPi / 2 || fabs(r2) > kPi / 2) {
    r0 -= kPi;
    if (r1 > kPi / 2.0) r1 = -r1 + kPi;
    else if (r1 < kPi / 2.0) r1 = -r1 - kPi;
    if (r2 > kPi / 2.0) r2 -= kPi;
    else if (r2 < kPi / 2.0) r2 += kPi;
  }
  return intrinsic ? V3{r0, r1, r2} : V3{r2, r1, r0};
}
// Quaterniond::FromTwoVectors (Eigen 3.3).  The anti-parallel branch (c < -1 + 1e-12) uses Eigen's SVD null
// vector there; it cannot occur on this path (walk-plane normals stay near +z) and maps to a fixed orthogonal axis.
SHC_HD Quat from_two_vectors(V3 a, V3 b) {
  V3 v0 = normalized(a), v1 = normalized(b);
  double c = dot(v1, v0);
  if (c < -1.0 + 1e-12) {
    c = fmax(c, -1.0);
    V3 o = fabs(v0.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
    V3 ax = normalized(cross(v0, o));
    double w2 = (1.0 + c) * 0.5;
    double s = sqrt(1.0 - w2);
    return Quat{sqrt(w2), ax.x * s, ax.y * s, ax.z * s};
  }
  V3 ax = cross(v0, v1);
  double s = sqrt((1.0 + c) * 2.0);
  double invs = 1.0 / s;
  return Quat{s * 0.5, ax.x * invs, ax.y * invs, ax.z * invs};
}
// Quaterniond(Matrix3d) (Eigen 3.3: Shoemake's trace method), m row-major
SHC_HD Quat quat_from_matrix(const double (&m)[9]) {
  double q[3], w;
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    w = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
  return Quat{w, q[0], q[1], q[2]};
}
// Eigen::AngleAxisd(q).axis() * angle() (AngleAxis = QuaternionBase; model.cpp:892-893)
SHC_HD V3 angle_axis_vector(Quat q) {
  double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (n != 0.0) {
    double angle = 2.0 * atan2(n, fabs(q.w));
    if (q.w < 0.0) n = -n;
    return V3{q.x / n * angle, q.y / n * angle, q.z / n * angle};
  }
  return V3{0, 0, 0}; // angle 0 about (1, 0, 0)
}
SHC_HD V3 lerp3(V3 a, V3 b, double c) { return b * c + a * (1.0 - c); } // interpolate() (standard_includes.h:143)
SHC_HD Quat slerp(Quat a, double t, Quat b) { // QuaternionBase::slerp
  const double one = 1.0 - 2.220446049250313e-16;
  double d = dot(a, b);
  double ad = fabs(d);
  double s0, s1;
  if (ad >= one) {
    s0 = 1.0 - t;
    s1 = t;
  } else {
    double th = acos(ad);
    double st = sin(th);
    s0 = sin((1.0 - t) * th) / st;
    s1 = sin(t * th) / st;
  }
  if (d < 0.0) s1 = -s1;
  return Quat{s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// ---------------------------------------------------------------- Pose (pose.h)
SHC_HD Pose pose_identity() { return Pose{V3{0, 0, 0}, Quat{1, 0, 0, 0}}; }
SHC_HD V3 transform_vector(const Pose &a, V3 v) { return a.p + rotate(a.r, v); }             // :151
SHC_HD V3 inverse_transform_vector(const Pose &a, V3 v) { return rotate(conj(a.r), v - a.p); } // :159 (R^T (v - p))
SHC_HD Pose add_pose(const Pose &a, const Pose &b) { return Pose{transform_vector(a, b.p), a.r * b.r}; }            // :167
SHC_HD Pose remove_pose(const Pose &a, const Pose &b) { return Pose{transform_vector(a, -b.p), a.r * inverse(b.r)}; } // :178
SHC_HD Pose interpolate_pose(const Pose &a, double c, const Pose &t) {                       // :190
  return Pose{t.p * c + a.p * (1.0 - c), slerp(a.r, c, t.r)};
}

// quartic Bezier and its derivative (standard_includes.h:402-420)
SHC_HD V3 quartic_bezier(const V3 *p, double t) {
  double s = 1.0 - t;
  double b0 = s * s * s * s, b1 = 4.0 * t * s * s * s, b2 = 6.0 * t * t * s * s, b3 = 4.0 * t * t * t * s, b4 = t * t * t * t;
  return p[0] * b0 + p[1] * b1 + p[2] * b2 + p[3] * b3 + p[4] * b4;
}
// The four-term sum is written as an explicit fma chain: with the compiler left to contract `a*b + c*d + ...` on its own, the
// choice of which product is rounded depends on the surrounding code, and two instantiations of the cycle kernel (compile-time
// and runtime feature flags) would differ in the last bit of the tip velocity.
SHC_HD V3 fma3(V3 a, double s, V3 c) { return V3{fma(a.x, s, c.x), fma(a.y, s, c.y), fma(a.z, s, c.z)}; }
SHC_HD V3 quartic_bezier_dot(V3 p0, V3 p1, V3 p2, V3 p3, V3 p4, double t) {
  double s = 1.0 - t;
  V3 r = (p1 - p0) * (4.0 * s * s * s);
  r = fma3(p2 - p1, 12.0 * s * s * t, r);
  r = fma3(p3 - p2, 12.0 * s * t * t, r);
  return fma3(p4 - p3, 4.0 * t * t * t, r);
}

// ---------------------------------------------------------------- small SPD solve  A x = b,  A = A^T > 0  (N <= 6)
// Reciprocal / reciprocal square root for the DLS solve, whose arithmetic form is this engine's own (shc_leg.hpp): the
// hardware estimate refined by two Newton steps (<= 1 ulp for the well-scaled positive arguments that occur: pivots
// >= lambda^2, limit costs) - 5-6 dependent instructions instead of the ~12-20 of an IEEE division / sqrt + division.
// The host build of the same templates (init chain) uses plain division.
template <bool EXACT = false>
SHC_HD double fast_rcp(double x) {
  if (EXACT) return 1.0 / x; // the init chain keeps IEEE division on both sides so that host and device tables agree
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
#else
  return 1.0 / x;
#endif
}
template <bool EXACT = false>
SHC_HD double fast_rsqrt(double x) {
  if (EXACT) return 1.0 / sqrt(x);
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
  y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

// A is the damped normal matrix J^T J + lambda^2 I of the DLS step (always SPD), so an unpivoted
// LDL^T in registers is exact enough and branch-free; fully unrolled for compile-time N.
template <int N, bool EXACT = false>
SHC_HD void spd_solve(double (&a)[N][N], double (&b)[N]) {
  double dinv[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    dinv[k] = fast_rcp<EXACT>(a[k][k]);
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      double l = a[i][k] * dinv[k];
#pragma unroll
      for (int j = k + 1; j <= i; ++j) a[i][j] -= l * a[j][k]; // a[j][k] still un-normalised (only the lower triangle is used)
    }
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      double l = a[i][k] * dinv[k];
      b[i] -= l * b[k];
      a[i][k] = l;
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) b[k] *= dinv[k];
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
#pragma unroll
    for (int i = k + 1; i < N; ++i) b[k] -= a[i][k] * b[i];
  }
}

} // namespace shc
