// mapping_probe.hip — measured comparison of two lane mappings for the per-leg kinematic core of the cycle
// (applyFK -> Jacobian -> one DLS solveIK -> updateJointPositions -> applyFK), the part of the cycle whose arithmetic does not
// depend on how robots are assigned to waves.  BASELINE.json's north_star sketches "one robot per wavefront, one leg per lane
// group ... contractions with wavefront shuffles"; the engine maps one leg to ONE lane.  This probe
// runs the same arithmetic (shc_leg.hpp) both ways on 4 096 and 65 536 hexapods:
//   A  leg-per-lane      one lane = one leg (3 joints): everything in that lane's registers, no cross-lane traffic
//   B  joint-per-lane    one lane = one joint, a leg = 3 lanes of a DPP quad (the 4th idles): sin / cos, the leg's DH factor,
//                        one Jacobian column and one row of J^T J per lane in parallel; chain products, the 3 x 3 solve's
//                        inputs and the tip travel between the lanes by quad permutes (v_mov_dpp, no LDS)
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I syropod_highlevel_controller_amd/csrc -o mapping_probe scripts/ubench/mapping_probe.hip && ./mapping_probe
#include "shc_leg.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace shc;

static void ck_(hipError_t e, const char *what) {
  if (e != hipSuccess) {
    printf("%s: %s\n", what, hipGetErrorString(e));
    exit(1);
  }
}
#define CK(x) ck_((x), #x)

struct LegK { // one leg's constants, 3 joints
  LegConst<3> lc;
};

// ---------------------------------------------------------------- A: one leg per lane
__global__ void __launch_bounds__(64) core_leg_per_lane(const LegK *__restrict__ K, double *q, double *qd, const double *target, double *tip, int64_t n_legs,
                                                        int L, double dt) {
  const int64_t t = int64_t(blockIdx.x) * 64 + threadIdx.x;
  if (t >= n_legs) return;
  const LegConst<3> &lc = K[t % L].lc;
  double qq[3], qv[3], dq[3];
  for (int j = 0; j < 3; ++j) {
    qq[j] = q[j * n_legs + t];
    qv[j] = qd[j * n_legs + t];
  }
  Chain<3> ch;
  fk_chain<3>(lc, qq, ch);
  ik_step<3>(lc, ch, qq, qv, V3{target[t], target[n_legs + t], target[2 * n_legs + t]}, dq);
  update_joints<3>(lc, dq, dt, 1.0 / dt, true, true, qq, qv);
  fk_chain<3>(lc, qq, ch);
  const V3 p = tip_robot_frame(lc, ch.pe);
  for (int j = 0; j < 3; ++j) {
    q[j * n_legs + t] = qq[j];
    qd[j * n_legs + t] = qv[j];
  }
  tip[t] = p.x, tip[n_legs + t] = p.y, tip[2 * n_legs + t] = p.z;
}

// ---------------------------------------------------------------- B: one joint per lane, three lanes of a quad per leg
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) { // value of lane SRC of this lane's quad
  int lo = __double2loint(v), hi = __double2hiint(v);
  constexpr int perm = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
  lo = __builtin_amdgcn_mov_dpp(lo, perm, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, perm, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
struct M34 { // rotation columns X, Y, Z and translation P of a rigid transform
  double X[3], Y[3], Z[3], P[3];
};
__device__ __forceinline__ M34 mul(const M34 &a, const M34 &b) { // a * b
  M34 r;
  for (int i = 0; i < 3; ++i) {
    r.X[i] = a.X[i] * b.X[0] + a.Y[i] * b.X[1] + a.Z[i] * b.X[2];
    r.Y[i] = a.X[i] * b.Y[0] + a.Y[i] * b.Y[1] + a.Z[i] * b.Y[2];
    r.Z[i] = a.X[i] * b.Z[0] + a.Y[i] * b.Z[1] + a.Z[i] * b.Z[2];
    r.P[i] = a.X[i] * b.P[0] + a.Y[i] * b.P[1] + a.Z[i] * b.P[2] + a.P[i];
  }
  return r;
}
template <int SRC>
__device__ __forceinline__ M34 quad_bcast(const M34 &m) {
  M34 r;
  for (int i = 0; i < 3; ++i) {
    r.X[i] = quad_bcast<SRC>(m.X[i]);
    r.Y[i] = quad_bcast<SRC>(m.Y[i]);
    r.Z[i] = quad_bcast<SRC>(m.Z[i]);
    r.P[i] = quad_bcast<SRC>(m.P[i]);
  }
  return r;
}
// DH factor of joint j: Rz(theta + q) * Tz(d) * Tx(r) * Rx(alpha)
__device__ __forceinline__ M34 dh_factor(const LegConst<3> &lc, int j, double q) {
  double s, c;
  sincos_joint(lc.link_th[j] + q, &s, &c);
  const double sa = lc.link_sa[j], ca = lc.link_ca[j], r = lc.link_r[j], d = lc.link_d[j];
  M34 m;
  m.X[0] = c, m.X[1] = s, m.X[2] = 0;
  m.Y[0] = -s * ca, m.Y[1] = c * ca, m.Y[2] = sa;
  m.Z[0] = s * sa, m.Z[1] = -c * sa, m.Z[2] = ca;
  m.P[0] = r * c, m.P[1] = r * s, m.P[2] = d;
  return m;
}
// forward kinematics with one joint per lane: every lane ends up with the frames it needs
struct QuadFk {
  V3 z, p, pe; // this lane's joint axis / origin and the tip, joint-1 frame
};
__device__ __forceinline__ QuadFk quad_fk(const LegConst<3> &lc, int j, double q) {
  const M34 A = dh_factor(lc, j, q);       // each lane: its own factor (sin / cos of three joints evaluated at once)
  const M34 A0 = quad_bcast<0>(A), A1 = quad_bcast<1>(A), A2 = quad_bcast<2>(A);
  const M34 T01 = mul(A0, A1), T012 = mul(T01, A2); // (every lane repeats the two 3 x 4 products: cheaper than more permutes)
  QuadFk r;
  r.pe = V3{T012.P[0], T012.P[1], T012.P[2]};
  // joint j rotates about the z axis of the frame BEFORE its own factor: identity, A0, A0 A1
  const V3 z0{0, 0, 1}, z1{A0.Z[0], A0.Z[1], A0.Z[2]}, z2{T01.Z[0], T01.Z[1], T01.Z[2]};
  const V3 p0{0, 0, 0}, p1{A0.P[0], A0.P[1], A0.P[2]}, p2{T01.P[0], T01.P[1], T01.P[2]};
  r.z = j == 0 ? z0 : (j == 1 ? z1 : z2);
  r.p = j == 0 ? p0 : (j == 1 ? p1 : p2);
  return r;
}
__global__ void __launch_bounds__(64) core_joint_per_lane(const LegK *__restrict__ K, double *q, double *qd, const double *target, double *tip,
                                                          int64_t n_legs, int L, double dt) {
  const int64_t lane = int64_t(blockIdx.x) * 64 + threadIdx.x;
  const int64_t leg = lane >> 2;
  const int j = int(lane & 3);
  if (leg >= n_legs) return;
  const bool act = j < 3;
  const int jj = act ? j : 2;
  const LegConst<3> &lc = K[leg % L].lc;
  double qj = q[jj * n_legs + leg], vj = qd[jj * n_legs + leg];
  QuadFk f = quad_fk(lc, jj, qj);
  // this lane's Jacobian column and its row of J^T J + l^2 I
  const V3 d = f.pe - f.p;
  const V3 col = cross(f.z, d);
  const V3 c0{quad_bcast<0>(col.x), quad_bcast<0>(col.y), quad_bcast<0>(col.z)}, c1{quad_bcast<1>(col.x), quad_bcast<1>(col.y), quad_bcast<1>(col.z)},
      c2{quad_bcast<2>(col.x), quad_bcast<2>(col.y), quad_bcast<2>(col.z)};
  const V3 tgt{target[leg], target[n_legs + leg], target[2 * n_legs + leg]};
  const V3 delta = base_rotate_inv(lc, tgt - V3{lc.p1[0], lc.p1[1], lc.p1[2]}) - f.pe;
  // joint-limit gradient: this joint's terms, the two costs summed over the quad
  const double e = (qj - lc.jcentre[jj]) * lc.jw_range[jj], v = vj * lc.jw_vrange[jj];
  const double e2 = act ? e * e : 0.0, v2 = act ? v * v : 0.0;
  const double pcost = quad_bcast<0>(e2) + quad_bcast<1>(e2) + quad_bcast<2>(e2), vcost = quad_bcast<0>(v2) + quad_bcast<1>(v2) + quad_bcast<2>(v2);
  const double ps = pcost == 0.0 ? 0.0 : fast_rsqrt(pcost), vs = vcost == 0.0 ? 0.0 : fast_rsqrt(vcost);
  const double l2 = kDls * kDls;
  const double g = 0.25 * ((-e * lc.jw_range[jj]) * ps) + 0.75 * ((-v * lc.jw_vrange[jj]) * vs);
  const double rhs = dot(col, delta) + l2 * g;
  // every lane solves the same 3 x 3 system from the gathered columns / right-hand sides (replicated: ~40 instructions)
  double a[3][3], b[3] = {quad_bcast<0>(rhs), quad_bcast<1>(rhs), quad_bcast<2>(rhs)};
  a[0][0] = dot(c0, c0) + l2, a[1][0] = dot(c1, c0), a[1][1] = dot(c1, c1) + l2, a[2][0] = dot(c2, c0), a[2][1] = dot(c2, c1), a[2][2] = dot(c2, c2) + l2;
  spd_solve<3>(a, b);
  const double dq = jj == 0 ? b[0] : (jj == 1 ? b[1] : b[2]);
  // this joint's integration and clamps
  double nv = dq * (1.0 / dt);
  nv = fmin(fmax(nv, -lc.jvmax[jj]), lc.jvmax[jj]);
  double nq = fmin(fmax(qj + nv * dt, lc.jmin[jj]), lc.jmax[jj]);
  f = quad_fk(lc, jj, nq);
  const V3 p = tip_robot_frame(lc, f.pe);
  if (act) {
    q[j * n_legs + leg] = nq;
    qd[j * n_legs + leg] = nv;
    tip[j * n_legs + leg] = j == 0 ? p.x : (j == 1 ? p.y : p.z);
  }
}

static void fill_leg(LegK &k, int l) { // default.yaml-like 3-DOF leg at yaw l * 60 degrees
  LegConst<3> &lc = k.lc;
  memset(&lc, 0, sizeof lc);
  const double yaw = l * 1.0471975512;
  lc.r1[0] = cos(yaw), lc.r1[1] = -sin(yaw), lc.r1[3] = sin(yaw), lc.r1[4] = cos(yaw), lc.r1[8] = 1;
  lc.p1[0] = 0.1 * cos(yaw), lc.p1[1] = 0.1 * sin(yaw);
  const double r[3] = {0.05, 0.1, 0.15}, al[3] = {1.5707963268, 0, 0}, mn[3] = {-1, -1.5, -2.0}, mx[3] = {1, 1.5, -0.1};
  for (int j = 0; j < 3; ++j) {
    lc.link_r[j] = r[j], lc.link_sa[j] = sin(al[j]), lc.link_ca[j] = cos(al[j]);
    lc.jmin[j] = mn[j], lc.jmax[j] = mx[j], lc.jvmax[j] = 6.0;
    lc.jcentre[j] = mn[j] + (mx[j] - mn[j]) / 2, lc.jw_range[j] = 0.1 / (mx[j] - mn[j]), lc.jw_vrange[j] = 0.1 / 12.0;
  }
}

int main() {
  const int L = 6;
  std::vector<LegK> hk(L);
  for (int l = 0; l < L; ++l) fill_leg(hk[l], l);
  LegK *dk;
  CK(hipMalloc(&dk, L * sizeof(LegK)));
  CK(hipMemcpy(dk, hk.data(), L * sizeof(LegK), hipMemcpyHostToDevice));
  for (int64_t robots : {4096, 65536}) {
    const int64_t n = robots * L;
    std::vector<double> hq(3 * n), hqd(3 * n, 0.0), ht(3 * n);
    srand(1);
    for (int64_t i = 0; i < n; ++i) {
      hq[i] = 0.2 * (rand() / double(RAND_MAX) - 0.5), hq[n + i] = 0.4, hq[2 * n + i] = -1.2;
    }
    double *q[2], *qd[2], *tgt, *tip[2];
    for (int v = 0; v < 2; ++v) {
      CK(hipMalloc(&q[v], 3 * n * 8)), CK(hipMalloc(&qd[v], 3 * n * 8)), CK(hipMalloc(&tip[v], 3 * n * 8));
      CK(hipMemcpy(q[v], hq.data(), 3 * n * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(qd[v], hqd.data(), 3 * n * 8, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&tgt, 3 * n * 8));
    // targets: the current tips (computed by one pass of A on a scratch copy), nudged
    {
      double *sq, *sqd;
      CK(hipMalloc(&sq, 3 * n * 8)), CK(hipMalloc(&sqd, 3 * n * 8));
      CK(hipMemcpy(sq, hq.data(), 3 * n * 8, hipMemcpyHostToDevice)), CK(hipMemcpy(sqd, hqd.data(), 3 * n * 8, hipMemcpyHostToDevice));
      CK(hipMemset(tgt, 0, 3 * n * 8));
      core_leg_per_lane<<<dim3((unsigned)((n + 63) / 64)), dim3(64)>>>(dk, sq, sqd, tgt, tip[0], n, L, 0.02);
      CK(hipMemcpy(ht.data(), tip[0], 3 * n * 8, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < 3 * n; ++i) ht[i] += 0.002 * (rand() / double(RAND_MAX) - 0.5);
      CK(hipMemcpy(tgt, ht.data(), 3 * n * 8, hipMemcpyHostToDevice));
      CK(hipFree(sq)), CK(hipFree(sqd));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)), CK(hipEventCreate(&e1));
    float ms[2];
    std::vector<double> first_a(3 * n);
    const int reps = 2000;
    for (int v = 0; v < 2; ++v) {
      auto launch = [&]() {
        if (v == 0) core_leg_per_lane<<<dim3((unsigned)((n + 63) / 64)), dim3(64)>>>(dk, q[0], qd[0], tgt, tip[0], n, L, 0.02);
        else core_joint_per_lane<<<dim3((unsigned)((4 * n + 63) / 64)), dim3(64)>>>(dk, q[1], qd[1], tgt, tip[1], n, L, 0.02);
      };
      launch(); // first pass from identical state: results compared below
      CK(hipDeviceSynchronize());
      if (v == 0) CK(hipMemcpy(first_a.data(), q[0], 3 * n * 8, hipMemcpyDeviceToHost));
      if (v == 1) {
        std::vector<double> a(first_a), b(3 * n);
        CK(hipMemcpy(b.data(), q[1], 3 * n * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int64_t i = 0; i < 3 * n; ++i) worst = fmax(worst, fabs(a[i] - b[i]));
        printf("  first pass, max |q_A - q_B| = %.2e rad\n", worst);
      }
      for (int i = 0; i < 200; ++i) launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[v], e0, e1));
      ms[v] /= reps;
    }
    printf("%7lld hexapods: A leg-per-lane %4lld waves %7.2f us/launch | B joint-per-lane %5lld waves %7.2f us/launch\n", (long long)robots,
           (long long)((n + 63) / 64), ms[0] * 1e3, (long long)((4 * n + 63) / 64), ms[1] * 1e3);
    for (int v = 0; v < 2; ++v) CK(hipFree(q[v])), CK(hipFree(qd[v])), CK(hipFree(tip[v]));
    CK(hipFree(tgt));
  }
  return 0;
}
