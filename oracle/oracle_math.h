/*
 * oracle_math.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Scalar double-precision restatement of the math layer the reference's hot path uses:
 *   - include/syropod_highlevel_controller/standard_includes.h:61-474 (helpers, Bezier, DH matrix, Euler<->quat)
 *   - include/syropod_highlevel_controller/pose.h:17-216 (Pose algebra)
 *   - the Eigen 3.3.x semantics those call (Quaterniond product/_transformVector/inverse/normalized/slerp/
 *     FromTwoVectors, Quaterniond(Matrix3d), toRotationMatrix, eulerAngles, AngleAxisd, isApprox,
 *     dynamic PartialPivLU inverse).  Eigen is NOT vendored in /root/reference and not installed in the
 *     build image (reference pin: distro Eigen 3.3.4 / 3.3.7, CMakeLists.txt:33), so these follow the
 *     published Eigen 3.3 algorithms.  PARITY UNPINNED: the reference has no tests / golden vectors
 *      and cannot be compiled here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <limits.h>
#include <math.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define ORC_UNASSIGNED_VALUE ((double)INT_MAX) /* standard_includes.h:52 */
#define ORC_GRAVITY_ACCELERATION (-9.81)       /* standard_includes.h:59 */

typedef struct { double x, y, z; } orc_v3;
typedef struct { double w, x, y, z; } orc_quat;
typedef struct { orc_v3 p; orc_quat r; } orc_pose;
typedef struct { double m[4][4]; } orc_m4;
typedef struct { double m[3][3]; } orc_m3;

/* ---------------------------------------------------------------- scalar helpers */
static inline double orc_deg2rad(double d) { return d / 360.0 * 2.0 * M_PI; }   /* standard_includes.h:64 */
static inline double orc_rad2deg(double r) { return (r / (2.0 * M_PI)) * 360.0; } /* :69 */
static inline int orc_mod(int a, int b) { return (a % b + b) % b; }              /* :76 */
static inline double orc_sqr(double v) { return v * v; }                         /* :82 */
static inline double orc_sign(double v) { return (v > 0 ? 1 : -1); }             /* :88 — sign(0) = -1 */
static inline int orc_round_to_int(double x) { return (x >= 0 ? (int)(x + 0.5) : -(int)(0.5 - x)); } /* :93 */
static inline int orc_round_to_even_int(double x) { return ((int)x % 2 == 0 ? (int)x : (int)x + 1); } /* :98 */
static inline double orc_clamped(double v, double lo, double hi) { return fmax(lo, fmin(v, hi)); }    /* :106 */
static inline double orc_set_precision(double v, int prec)                                           /* :143 */
{
  return orc_round_to_int(v * pow(10, prec)) / pow(10, prec);
}
static inline double orc_smooth_step(double c)                                                       /* :163 */
{
  return (6.0 * pow(c, 5) - 15.0 * pow(c, 4) + 10.0 * pow(c, 3));
}
static inline double orc_interpolate(double o, double t, double c) { return (1.0 - c) * o + c * t; } /* :201 */

/* ---------------------------------------------------------------- vec3 */
static inline orc_v3 orc_v3_make(double x, double y, double z) { orc_v3 v = { x, y, z }; return v; }
static inline orc_v3 orc_v3_add(orc_v3 a, orc_v3 b) { return orc_v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline orc_v3 orc_v3_sub(orc_v3 a, orc_v3 b) { return orc_v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_v3 orc_v3_scale(orc_v3 a, double s) { return orc_v3_make(a.x * s, a.y * s, a.z * s); }
static inline orc_v3 orc_v3_neg(orc_v3 a) { return orc_v3_make(-a.x, -a.y, -a.z); }
static inline double orc_v3_dot(orc_v3 a, orc_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline orc_v3 orc_v3_cross(orc_v3 a, orc_v3 b)
{
  return orc_v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double orc_v3_sqnorm(orc_v3 a) { return orc_v3_dot(a, a); }
static inline double orc_v3_norm(orc_v3 a) { return sqrt(orc_v3_sqnorm(a)); }
/* Eigen 3.3 MatrixBase::normalized(): z = squaredNorm; z > 0 ? n / sqrt(z) : n */
static inline orc_v3 orc_v3_normalized(orc_v3 a)
{
  double z = orc_v3_sqnorm(a);
  if (z > 0.0) { double s = sqrt(z); return orc_v3_make(a.x / s, a.y / s, a.z / s); }
  return a;
}
static inline orc_v3 orc_v3_lerp(orc_v3 o, orc_v3 t, double c) /* standard_includes.h:201 on Vector3d */
{
  return orc_v3_add(orc_v3_scale(o, 1.0 - c), orc_v3_scale(t, c));
}
/* Eigen isApprox: ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2), prec = 1e-12 */
static inline int orc_v3_is_approx(orc_v3 a, orc_v3 b)
{
  double p2 = 1e-12 * 1e-12;
  return orc_v3_sqnorm(orc_v3_sub(a, b)) <= p2 * fmin(orc_v3_sqnorm(a), orc_v3_sqnorm(b));
}
static inline orc_v3 orc_get_projection(orc_v3 a, orc_v3 b) /* standard_includes.h:173 */
{
  if (orc_v3_norm(a) == 0.0 || orc_v3_norm(b) == 0.0) return orc_v3_make(0, 0, 0);
  return orc_v3_scale(b, orc_v3_dot(a, b) / orc_v3_dot(b, b));
}
static inline orc_v3 orc_get_rejection(orc_v3 a, orc_v3 b) { return orc_v3_sub(a, orc_get_projection(a, b)); } /* :190 */
/* clamped(vector, magnitude) standard_includes.h:117 */
static inline orc_v3 orc_v3_clamp_mag(orc_v3 v, double mag)
{
  return orc_v3_norm(v) > mag ? orc_v3_scale(v, mag / orc_v3_norm(v)) : v;
}
static inline orc_v3 orc_v3_set_precision(orc_v3 v, int prec) /* :152 */
{
  return orc_v3_make(orc_round_to_int(v.x * pow(10, prec)) / pow(10, prec),
                     orc_round_to_int(v.y * pow(10, prec)) / pow(10, prec),
                     orc_round_to_int(v.z * pow(10, prec)) / pow(10, prec));
}
#define ORC_UNDEFINED_POSITION orc_v3_make((double)INT_MAX, (double)INT_MAX, (double)INT_MAX) /* :57 */

/* ---------------------------------------------------------------- quaternion (Eigen::Quaterniond, ctor order w,x,y,z) */
static inline orc_quat orc_quat_make(double w, double x, double y, double z) { orc_quat q = { w, x, y, z }; return q; }
static inline orc_quat orc_quat_identity(void) { return orc_quat_make(1, 0, 0, 0); }
#define ORC_UNDEFINED_ROTATION orc_quat_make(0, 0, 0, 0) /* standard_includes.h:56 */
static inline orc_quat orc_quat_mul(orc_quat a, orc_quat b) /* Eigen quat_product<Scalar> */
{
  return orc_quat_make(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                       a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                       a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                       a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
static inline orc_quat orc_quat_conj(orc_quat q) { return orc_quat_make(q.w, -q.x, -q.y, -q.z); }
static inline double orc_quat_sqnorm(orc_quat q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
static inline double orc_quat_dot(orc_quat a, orc_quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
/* QuaternionBase::inverse(): n2 > 0 ? conj / n2 : zero quaternion */
static inline orc_quat orc_quat_inverse(orc_quat q)
{
  double n2 = orc_quat_sqnorm(q);
  if (n2 > 0.0) return orc_quat_make(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
  return orc_quat_make(0, 0, 0, 0);
}
static inline orc_quat orc_quat_normalized(orc_quat q) /* coeffs().normalized(), Eigen 3.3 */
{
  double z = orc_quat_sqnorm(q);
  if (z > 0.0) { double s = sqrt(z); return orc_quat_make(q.w / s, q.x / s, q.y / s, q.z / s); }
  return q;
}
/* QuaternionBase::_transformVector */
static inline orc_v3 orc_quat_rotate(orc_quat q, orc_v3 v)
{
  orc_v3 qv = orc_v3_make(q.x, q.y, q.z);
  orc_v3 uv = orc_v3_cross(qv, v);
  uv = orc_v3_add(uv, uv);
  return orc_v3_add(orc_v3_add(v, orc_v3_scale(uv, q.w)), orc_v3_cross(qv, uv));
}
static inline int orc_quat_is_approx(orc_quat a, orc_quat b)
{
  double p2 = 1e-12 * 1e-12;
  double dw = a.w - b.w, dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return (dw * dw + dx * dx + dy * dy + dz * dz) <= p2 * fmin(orc_quat_sqnorm(a), orc_quat_sqnorm(b));
}
static inline int orc_quat_is_undefined(orc_quat q) { return orc_quat_is_approx(q, ORC_UNDEFINED_ROTATION); }
static inline orc_quat orc_correct_rotation(orc_quat test, orc_quat ref) /* standard_includes.h:211 */
{
  if (orc_quat_dot(test, ref) < 0.0) return orc_quat_make(-test.w, -test.x, -test.y, -test.z);
  return test;
}
/* QuaternionBase::toRotationMatrix */
static inline orc_m3 orc_quat_to_matrix(orc_quat q)
{
  orc_m3 r;
  double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r.m[0][0] = 1.0 - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz; r.m[1][1] = 1.0 - (txx + tzz); r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = 1.0 - (txx + tyy);
  return r;
}
/* Quaterniond(Matrix3d): quaternionbase_assign_impl<Other,3,3> (Shoemake) */
static inline orc_quat orc_quat_from_matrix(const orc_m3 *mat)
{
  double q[3];
  double w;
  double t = mat->m[0][0] + mat->m[1][1] + mat->m[2][2];
  if (t > 0.0)
  {
    t = sqrt(t + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    q[0] = (mat->m[2][1] - mat->m[1][2]) * t;
    q[1] = (mat->m[0][2] - mat->m[2][0]) * t;
    q[2] = (mat->m[1][0] - mat->m[0][1]) * t;
  }
  else
  {
    int i = 0;
    if (mat->m[1][1] > mat->m[0][0]) i = 1;
    if (mat->m[2][2] > mat->m[i][i]) i = 2;
    int j = (i + 1) % 3;
    int k = (j + 1) % 3;
    t = sqrt(mat->m[i][i] - mat->m[j][j] - mat->m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    w = (mat->m[k][j] - mat->m[j][k]) * t;
    q[j] = (mat->m[j][i] + mat->m[i][j]) * t;
    q[k] = (mat->m[k][i] + mat->m[i][k]) * t;
  }
  return orc_quat_make(w, q[0], q[1], q[2]);
}
/* Quaterniond(AngleAxisd) */
static inline orc_quat orc_quat_from_angle_axis(double angle, orc_v3 axis)
{
  double ha = 0.5 * angle;
  double s = sin(ha);
  return orc_quat_make(cos(ha), s * axis.x, s * axis.y, s * axis.z);
}
/* AngleAxisd(Quaterniond): returns angle, writes axis */
static inline double orc_angle_axis_from_quat(orc_quat q, orc_v3 *axis)
{
  double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  /* (n < epsilon -> stableNorm) : same value up to rounding */
  if (n != 0.0)
  {
    double angle = 2.0 * atan2(n, fabs(q.w));
    if (q.w < 0.0) n = -n;
    *axis = orc_v3_make(q.x / n, q.y / n, q.z / n);
    return angle;
  }
  *axis = orc_v3_make(1, 0, 0);
  return 0.0;
}
/* AngleAxisd * Vector3d == toRotationMatrix() * v */
static inline orc_v3 orc_angle_axis_rotate(double angle, orc_v3 axis, orc_v3 v)
{
  double s = sin(angle), c = cos(angle);
  orc_v3 sin_axis = orc_v3_scale(axis, s);
  orc_v3 cos1_axis = orc_v3_scale(axis, 1.0 - c);
  double r[3][3];
  double tmp;
  tmp = cos1_axis.x * axis.y; r[0][1] = tmp - sin_axis.z; r[1][0] = tmp + sin_axis.z;
  tmp = cos1_axis.x * axis.z; r[0][2] = tmp + sin_axis.y; r[2][0] = tmp - sin_axis.y;
  tmp = cos1_axis.y * axis.z; r[1][2] = tmp - sin_axis.x; r[2][1] = tmp + sin_axis.x;
  r[0][0] = cos1_axis.x * axis.x + c; r[1][1] = cos1_axis.y * axis.y + c; r[2][2] = cos1_axis.z * axis.z + c;
  return orc_v3_make(r[0][0] * v.x + r[0][1] * v.y + r[0][2] * v.z,
                     r[1][0] * v.x + r[1][1] * v.y + r[1][2] * v.z,
                     r[2][0] * v.x + r[2][1] * v.y + r[2][2] * v.z);
}
/*
 * Quaterniond::FromTwoVectors(a, b) (Eigen 3.3 setFromTwoVectors).  The anti-parallel branch of Eigen
 * takes the last right-singular vector of [v0; v1] (sign unspecified); here: a deterministic unit
 * vector orthogonal to v0.  That branch is unreachable on the accelerated path's inputs.
 */
static inline orc_quat orc_quat_from_two_vectors(orc_v3 a, orc_v3 b)
{
  orc_v3 v0 = orc_v3_normalized(a);
  orc_v3 v1 = orc_v3_normalized(b);
  double c = orc_v3_dot(v1, v0);
  if (c < -1.0 + 1e-12)
  {
    c = fmax(c, -1.0);
    orc_v3 ortho = fabs(v0.x) < 0.9 ? orc_v3_make(1, 0, 0) : orc_v3_make(0, 1, 0);
    orc_v3 axis = orc_v3_normalized(orc_v3_cross(v0, ortho));
    double w2 = (1.0 + c) * 0.5;
    double s = sqrt(1.0 - w2);
    return orc_quat_make(sqrt(w2), axis.x * s, axis.y * s, axis.z * s);
  }
  orc_v3 axis = orc_v3_cross(v0, v1);
  double s = sqrt((1.0 + c) * 2.0);
  double invs = 1.0 / s;
  return orc_quat_make(s * 0.5, axis.x * invs, axis.y * invs, axis.z * invs);
}
/* QuaternionBase::slerp(t, other) */
static inline orc_quat orc_quat_slerp(orc_quat a, double t, orc_quat b)
{
  const double one = 1.0 - 2.220446049250313e-16;
  double d = orc_quat_dot(a, b);
  double abs_d = fabs(d);
  double scale0, scale1;
  if (abs_d >= one)
  {
    scale0 = 1.0 - t;
    scale1 = t;
  }
  else
  {
    double theta = acos(abs_d);
    double sin_theta = sin(theta);
    scale0 = sin((1.0 - t) * theta) / sin_theta;
    scale1 = sin((t * theta)) / sin_theta;
  }
  if (d < 0.0) scale1 = -scale1;
  return orc_quat_make(scale0 * a.w + scale1 * b.w, scale0 * a.x + scale1 * b.x,
                       scale0 * a.y + scale1 * b.y, scale0 * a.z + scale1 * b.z);
}
/* MatrixBase::eulerAngles(a0,a1,a2), Eigen 3.3 Geometry/EulerAngles.h, Tait-Bryan branch (a0 != a2) */
static inline void orc_m3_euler_angles(const orc_m3 *mat, int a0, int a1, int a2, double res[3])
{
  (void)a2;
  const int odd = ((a0 + 1) % 3 == a1) ? 0 : 1;
  const int i = a0;
  const int j = (a0 + 1 + odd) % 3;
  const int k = (a0 + 2 - odd) % 3;
  res[0] = atan2(mat->m[j][k], mat->m[k][k]);
  double c2 = sqrt(mat->m[i][i] * mat->m[i][i] + mat->m[i][j] * mat->m[i][j]);
  if ((odd && res[0] < 0.0) || ((!odd) && res[0] > 0.0))
  {
    if (res[0] > 0.0) res[0] -= M_PI; else res[0] += M_PI;
    res[1] = atan2(-mat->m[i][k], -c2);
  }
  else
  {
    res[1] = atan2(-mat->m[i][k], c2);
  }
  double s1 = sin(res[0]);
  double c1 = cos(res[0]);
  res[2] = atan2(s1 * mat->m[k][i] - c1 * mat->m[j][i], c1 * mat->m[j][j] - s1 * mat->m[k][j]);
  if (!odd) { res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2]; }
}
/* eulerAnglesToQuaternion (standard_includes.h:227) */
static inline orc_quat orc_euler_to_quat(orc_v3 e, int intrinsic)
{
  orc_quat qx = orc_quat_from_angle_axis(e.x, orc_v3_make(1, 0, 0));
  orc_quat qy = orc_quat_from_angle_axis(e.y, orc_v3_make(0, 1, 0));
  orc_quat qz = orc_quat_from_angle_axis(e.z, orc_v3_make(0, 0, 1));
  if (intrinsic) return orc_quat_mul(orc_quat_mul(qx, qy), qz);
  return orc_quat_mul(orc_quat_mul(qz, qy), qx);
}
/* quaternionToEulerAngles (standard_includes.h:248-291), including the "flip" fix-up */
static inline orc_v3 orc_quat_to_euler(orc_quat q, int intrinsic)
{
  double r[3] = { 0, 0, 0 };
  orc_m3 m = orc_quat_to_matrix(q);
  if (intrinsic) orc_m3_euler_angles(&m, 0, 1, 2, r);
  else orc_m3_euler_angles(&m, 2, 1, 0, r);
  if (fabs(r[1]) > M_PI / 2 || fabs(r[2]) > M_PI / 2)
  {
    r[0] -= M_PI;
    if (r[1] > M_PI / 2.0) r[1] = -r[1] + M_PI;
    else if (r[1] < M_PI / 2.0) r[1] = -r[1] - M_PI;
    if (r[2] > M_PI / 2.0) r[2] -= M_PI;
    else if (r[2] < M_PI / 2.0) r[2] += M_PI;
  }
  return intrinsic ? orc_v3_make(r[0], r[1], r[2]) : orc_v3_make(r[2], r[1], r[0]);
}

/* ---------------------------------------------------------------- 4x4 */
static inline orc_m4 orc_m4_identity(void)
{
  orc_m4 r; memset(&r, 0, sizeof r);
  r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0;
  return r;
}
static inline orc_m4 orc_m4_mul(const orc_m4 *a, const orc_m4 *b)
{
  orc_m4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
    {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += a->m[i][k] * b->m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
/* createDHMatrix (standard_includes.h:466) */
static inline orc_m4 orc_create_dh_matrix(double d, double theta, double r, double alpha)
{
  orc_m4 m;
  m.m[0][0] = cos(theta); m.m[0][1] = -sin(theta) * cos(alpha); m.m[0][2] = sin(theta) * sin(alpha); m.m[0][3] = r * cos(theta);
  m.m[1][0] = sin(theta); m.m[1][1] = cos(theta) * cos(alpha); m.m[1][2] = -cos(theta) * sin(alpha); m.m[1][3] = r * sin(theta);
  m.m[2][0] = 0; m.m[2][1] = sin(alpha); m.m[2][2] = cos(alpha); m.m[2][3] = d;
  m.m[3][0] = 0; m.m[3][1] = 0; m.m[3][2] = 0; m.m[3][3] = 1;
  return m;
}

/*
 * Dynamic-size MatrixXd::inverse() == PartialPivLU(A).inverse() (Eigen 3.3: unblocked partial-pivot LU
 * for sizes <= 16, then solve against the row-permuted identity).  a: n x n row-major, lda = n.
 * Returns 0 on a zero pivot (Eigen would produce inf/nan; the callers' matrices are SPD + lambda^2 I).
 */
static inline int orc_lu_inverse(const double *a, int n, double *inv)
{
  double lu[100]; /* n <= 10 */
  int perm[10];
  if (n > 10) return 0;
  for (int i = 0; i < n * n; ++i) lu[i] = a[i];
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k)
  {
    int piv = k;
    double best = fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; ++i)
    {
      double v = fabs(lu[i * n + k]);
      if (v > best) { best = v; piv = i; }
    }
    if (best == 0.0) return 0;
    if (piv != k)
    {
      for (int j = 0; j < n; ++j) { double t = lu[k * n + j]; lu[k * n + j] = lu[piv * n + j]; lu[piv * n + j] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < n; ++i) lu[i * n + k] /= lu[k * n + k];
    for (int i = k + 1; i < n; ++i)
      for (int j = k + 1; j < n; ++j) lu[i * n + j] -= lu[i * n + k] * lu[k * n + j];
  }
  for (int c = 0; c < n; ++c)
  {
    double y[10];
    for (int i = 0; i < n; ++i) y[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) y[i] -= lu[i * n + j] * y[j];
    for (int i = n - 1; i >= 0; --i)
    {
      for (int j = i + 1; j < n; ++j) y[i] -= lu[i * n + j] * y[j];
      y[i] /= lu[i * n + i];
    }
    for (int i = 0; i < n; ++i) inv[i * n + c] = y[i];
  }
  return 1;
}

/* ---------------------------------------------------------------- Pose (pose.h) */
static inline orc_pose orc_pose_make(orc_v3 p, orc_quat r) { orc_pose o; o.p = p; o.r = r; return o; }
static inline orc_pose orc_pose_identity(void) { return orc_pose_make(orc_v3_make(0, 0, 0), orc_quat_identity()); } /* :199 */
static inline orc_pose orc_pose_undefined(void) { return orc_pose_make(ORC_UNDEFINED_POSITION, ORC_UNDEFINED_ROTATION); } /* :206 */
static inline int orc_pose_eq(orc_pose a, orc_pose b) /* pose.h:97 */
{
  return orc_v3_is_approx(a.p, b.p) && orc_quat_is_approx(a.r, b.r);
}
static inline int orc_pose_ne(orc_pose a, orc_pose b) /* pose.h:105 */
{
  return !orc_v3_is_approx(a.p, b.p) || !orc_quat_is_approx(a.r, b.r);
}
static inline int orc_pose_is_valid(orc_pose a) /* pose.h:42 */
{
  return fabs(a.p.x) < ORC_UNASSIGNED_VALUE && fabs(a.p.y) < ORC_UNASSIGNED_VALUE && fabs(a.p.z) < ORC_UNASSIGNED_VALUE &&
         fabs(a.r.w) < ORC_UNASSIGNED_VALUE && fabs(a.r.x) < ORC_UNASSIGNED_VALUE && fabs(a.r.y) < ORC_UNASSIGNED_VALUE &&
         fabs(a.r.z) < ORC_UNASSIGNED_VALUE;
}
static inline orc_pose orc_pose_inverse(orc_pose a) /* pose.h:112 operator~ */
{
  orc_quat c = orc_quat_conj(a.r);
  return orc_pose_make(orc_quat_rotate(c, orc_v3_neg(a.p)), c);
}
static inline orc_v3 orc_pose_transform_vector(orc_pose a, orc_v3 v) /* pose.h:151 */
{
  return orc_v3_add(a.p, orc_quat_rotate(a.r, v));
}
static inline orc_v3 orc_pose_inverse_transform_vector(orc_pose a, orc_v3 v) /* pose.h:159 */
{
  return orc_pose_transform_vector(orc_pose_inverse(a), v);
}
static inline orc_pose orc_pose_add(orc_pose a, orc_pose b) /* pose.h:167 */
{
  orc_pose r = a;
  r.p = orc_pose_transform_vector(a, b.p);
  r.r = orc_quat_mul(a.r, b.r);
  return r;
}
static inline orc_pose orc_pose_remove(orc_pose a, orc_pose b) /* pose.h:178 */
{
  orc_pose r = a;
  r.p = orc_pose_transform_vector(a, orc_v3_neg(b.p));
  r.r = orc_quat_mul(a.r, orc_quat_inverse(b.r));
  return r;
}
static inline orc_pose orc_pose_interpolate(orc_pose a, double c, orc_pose target) /* pose.h:190 */
{
  orc_v3 p = orc_v3_add(orc_v3_scale(target.p, c), orc_v3_scale(a.p, 1.0 - c));
  return orc_pose_make(p, orc_quat_slerp(a.r, c, target.r));
}
/* Pose::transform(Matrix4d) pose.h:135 */
static inline orc_pose orc_pose_transform_m4(orc_pose a, const orc_m4 *t)
{
  orc_pose r;
  r.p.x = t->m[0][0] * a.p.x + t->m[0][1] * a.p.y + t->m[0][2] * a.p.z + t->m[0][3] * 1.0;
  r.p.y = t->m[1][0] * a.p.x + t->m[1][1] * a.p.y + t->m[1][2] * a.p.z + t->m[1][3] * 1.0;
  r.p.z = t->m[2][0] * a.p.x + t->m[2][1] * a.p.y + t->m[2][2] * a.p.z + t->m[2][3] * 1.0;
  orc_m3 rm;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rm.m[i][j] = t->m[i][j];
  r.r = orc_quat_normalized(orc_quat_mul(orc_quat_from_matrix(&rm), a.r));
  return r;
}

/* ---------------------------------------------------------------- Bezier (standard_includes.h:347-420) */
static inline double orc_cubic_bezier_scalar(const double *p, double t) /* :347 on doubles */
{
  double s = 1.0 - t;
  return p[0] * (s * s * s) + p[1] * (3.0 * t * s * s) + p[2] * (3.0 * t * t * s) + p[3] * (t * t * t);
}
static inline orc_v3 orc_quartic_bezier(const orc_v3 *p, double t) /* :402 */
{
  double s = 1.0 - t;
  orc_v3 r = orc_v3_scale(p[0], s * s * s * s);
  r = orc_v3_add(r, orc_v3_scale(p[1], 4.0 * t * s * s * s));
  r = orc_v3_add(r, orc_v3_scale(p[2], 6.0 * t * t * s * s));
  r = orc_v3_add(r, orc_v3_scale(p[3], 4.0 * t * t * t * s));
  r = orc_v3_add(r, orc_v3_scale(p[4], t * t * t * t));
  return r;
}
static inline orc_v3 orc_quartic_bezier_dot(const orc_v3 *p, double t) /* :415 */
{
  double s = 1.0 - t;
  orc_v3 r = orc_v3_scale(orc_v3_sub(p[1], p[0]), 4.0 * s * s * s);
  r = orc_v3_add(r, orc_v3_scale(orc_v3_sub(p[2], p[1]), 12.0 * s * s * t));
  r = orc_v3_add(r, orc_v3_scale(orc_v3_sub(p[3], p[2]), 12.0 * s * t * t));
  r = orc_v3_add(r, orc_v3_scale(orc_v3_sub(p[4], p[3]), 4.0 * t * t * t));
  return r;
}

#endif /* ORACLE_MATH_H */
