// pmc_step.hpp -- one 50 Hz control step of the PMC tracking environment, fused:
//   10 x { PD torque (LR:119-148) + physics substep (what PyBullet's stepSimulation does at PLE:206) }
//   + mocap reference lookup (ML:65-115) + observation packing (PLE:276-317) + tracking reward (PLE:350-426)
//   + termination (PLE:337-348) + sampling-table bookkeeping (PLE:235-240) + optional in-kernel re-seed.
// PLE = primitive_level_env.py, ML = motion_lib.py, LR = legged_robot.py of the reference.
//
// Execution model: one environment = one quad of lanes, lane = leg (lanes.hpp).  State lives in registers
// across the 10 substeps; HBM is touched once per control step.
//
// Physics formulation (DESIGN.md "physics spec"): the robot is a star -- base + four 3-joint chains -- so the
// articulated-body elimination is done per leg lane in closed form.  All spatial quantities are expressed in
// the base frame F0 about the F0 origin:
//   leg lane:  FK, link inertias in F0, bias forces (recursive Newton-Euler), composite inertias,
//              3x3 joint-space inertia M_ll and its Cholesky factor Lm, Y = M_bl Lm^-T  (6x3)
//   quad sum:  S = I_base + sum_legs I^c_leg - sum_legs Y Y^T     == articulated-body inertia of the base
//   base:      Cholesky S = Lb Lb^T, base acceleration, back-substitution into the legs
// Contacts / joint limits are rows of a projected Gauss-Seidel solve carried in the whitened coordinates
//   gt = Lb^-1 (J_b - Y jt),  jt = Lm^-1 J_l,   A_rs = gt_r.gt_s + [same leg] jt_r.jt_s
// so a row update touches 6 shared numbers (quad broadcast) and 3 lane-private ones.
#pragma once
#include "../../include/llenv_model.h"
#include "pmc_math.hpp"
#include "pmc_params.hpp"
#include <type_traits>

#define LLS_DONE_FALL 1
#define LLS_DONE_CLIP_END 2
#define LLS_DONE_DIVERGED 4
#define LLS_DONE_COLLISION 8
#define LLS_DONE_NONFINITE 16

// PLE:235-240 for the batch, host statement (the CPU build of this source; the GPU folds with one wavefront, llenv.hip table_fold_wave): fold the
// statistics published by the episodes that finished in control step `sl` of a launch into the per-clip table and rebuild the sampling
// distribution  p ~ (1 - avg_reward_sum)^factor  as an inclusive CDF -- into `cdf` itself after the launch's last step, into version slot sl + 1
// otherwise (StepParams::cdf_ver: what a re-seed at step sl + 1 of the same launch samples from).
LL_HD void pmc_finalize_table(const StepParams& P, int sl, bool last_step) {
  const int n = P.n_clips;
  unsigned long long* pend_r = P.pending_reward + (long)sl * n;
  unsigned long long* pend_l = P.pending_len + (long)sl * n;
  const double* prev = sl == 0 ? P.cdf : P.cdf_ver + (long)sl * n;
  double* dst = last_step ? P.cdf : P.cdf_ver + (long)(sl + 1) * n;
  bool any = false;
  for (int c = 0; c < n; c++) {
    unsigned long long pr = pend_r[c], pl = pend_l[c];
    if (pr) {
      union { float f; uint32_t u; } a, b;
      a.u = (uint32_t)pr; b.u = (uint32_t)pl;
      P.avg_reward[c] = (double)a.f;
      P.avg_len[c] = (double)b.f;
      pend_r[c] = 0ull;
      pend_l[c] = 0ull;
      any = true;
    }
  }
  if (!any) {
    if (dst != prev) for (int c = 0; c < n; c++) dst[c] = prev[c];
    return;
  }
  double sum = 0.0;
  for (int c = 0; c < n; c++) {
    P.prob[c] = pow(1.0 - P.avg_reward[c], P.sample_factor);
    sum += P.prob[c];
  }
  double acc = 0.0;
  for (int c = 0; c < n; c++) {
    P.prob[c] /= sum;
    acc += P.prob[c];
    dst[c] = acc;
  }
  dst[n - 1] = 1.0;
}

template <class L>
struct Pmc {
  typedef typename L::F F;
  typedef typename L::B B;
  typedef typename L::I I;
  typedef typename L::D D;
  typedef typename L::F2 F2;
  typedef V3<float> V3u;
  typedef V3<F> V3l;

  // ---------------------------------------------------------------------------------------------------
  struct Base {   // quad-uniform dynamic state of the base
    V3u p;        // world position of the F0 origin
    Q4 q;         // world <- base
    V3u v, w;     // world linear / angular velocity
  };

  struct LegKin {
    M3<F> R1, R2, R3;     // link -> base rotations
    V3l p1, p2, p3;       // joint origins in F0
    V3l a2;               // common axis of joints 2 and 3 in F0 (joint 1 axis is +x)
    V3l s1, s2, s3;       // linear parts of the motion subspaces: p_j x a_j
  };

  static LL_HD V3l ld3c(const L& ln, const float* legc, int f) {
    return mk3<F>(ln.legc(legc, f), ln.legc(legc, f + 1), ln.legc(legc, f + 2));
  }

  // sin/cos for joint angles (|x| < ~1e3): two-constant Cody-Waite reduction to [-pi/4, pi/4] + minimax polynomials,
  // ~1 ulp, branch-free -- no large-argument path, which the library sincosf would inline six times per substep
  static LL_HD void sincos_joint(const L& ln, const F& x, F* s, F* c) {
    F kf = lm::rint_(x * 0.63661977236758134f);
    F r = x - kf * 1.5707962512969971f;            // pi/2 high part (float)
    r = r - kf * 7.5497894158615964e-08f;          // pi/2 low part
    F r2 = r * r;
    F sp = r + r * r2 * (-0.16666654611f + r2 * (0.0083321608736f + r2 * (-0.00019515295891f)));
    F cp = ln.lane_f(1.0f) + r2 * (-0.5f + r2 * (0.041666645683f + r2 * (-0.0013887316255f + r2 * 0.000024433157117f)));
    I k = L::f2i(kf);
    B swap = lm::odd_(k);
    B neg_s = lm::bit1_(k);                         // k mod 4 in {2,3}
    B neg_c = lm::bit1_(k + 1);                     // k mod 4 in {1,2}
    F ss = lm::sel(swap, cp, sp), cs = lm::sel(swap, sp, cp);
    *s = lm::sel(neg_s, ln.lane_f(0.0f) - ss, ss);
    *c = lm::sel(neg_c, ln.lane_f(0.0f) - cs, cs);
  }

  static LL_HD LegKin leg_fk(const L& ln, const float* legc, F q1, F q2, F q3) {
    LegKin k;
    // one sine / cosine per sub-lane (hip angle, thigh angle, thigh + shank angle), handed round the leg's sub-lanes
    F sk, ck, sv[3], cv[3];
    sincos_joint(ln, lm::sel(ln.is_sub(0), q1, lm::sel(ln.is_sub(1), q2, q2 + q3)), &sk, &ck);
    L::spread3(sk, sv); L::spread3(ck, cv);
    F c1 = cv[0], s1 = sv[0], c2 = cv[1], s2 = sv[1], c3 = cv[2], s3 = sv[2];
    F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f);
    // R1 = Rx(q1);  R2 = R1 * R(-y, q2);  R3 = R1 * R(-y, q2+q3)   (hip axis +x, thigh/shank axis -y)
    k.R1.m[0] = one; k.R1.m[1] = zero; k.R1.m[2] = zero;
    k.R1.m[3] = zero; k.R1.m[4] = c1; k.R1.m[5] = zero - s1;
    k.R1.m[6] = zero; k.R1.m[7] = s1; k.R1.m[8] = c1;
    k.R2.m[0] = c2; k.R2.m[1] = zero; k.R2.m[2] = zero - s2;
    k.R2.m[3] = zero - s1 * s2; k.R2.m[4] = c1; k.R2.m[5] = zero - s1 * c2;
    k.R2.m[6] = c1 * s2; k.R2.m[7] = s1; k.R2.m[8] = c1 * c2;
    k.R3.m[0] = c3; k.R3.m[1] = zero; k.R3.m[2] = zero - s3;
    k.R3.m[3] = zero - s1 * s3; k.R3.m[4] = c1; k.R3.m[5] = zero - s1 * c3;
    k.R3.m[6] = c1 * s3; k.R3.m[7] = s1; k.R3.m[8] = c1 * c3;
    k.p1 = ld3c(ln, legc, LC_R1);
    k.p2 = k.p1 + mul(k.R1, ld3c(ln, legc, LC_R2));
    k.p3 = k.p2 + mul(k.R2, ld3c(ln, legc, LC_R3));
    k.a2 = mk3<F>(zero, zero - c1, zero - s1);
    V3l a1 = mk3<F>(one, zero, zero);
    k.s1 = cross(k.p1, a1);
    k.s2 = cross(k.p2, k.a2);
    k.s3 = cross(k.p3, k.a2);
    return k;
  }

  // world position of this lane's foot (LR:199-205 compute_end_effector_info)
  static LL_HD V3l foot_world(const L& ln, const float* legc, const V3u& p, const M3<float>& R, F q1, F q2, F q3) {
    LegKin k = leg_fk(ln, legc, q1, q2, q3);
    V3l fb = k.p3 + mul(k.R3, ld3c(ln, legc, LC_FOOT));
    V3l fw = mul(R, fb);
    return mk3<F>(fw.x + p.x, fw.y + p.y, fw.z + p.z);
  }

  // ---- link-level work is split over the sub-lanes of a leg: sub-lane k < 3 owns link k + 1 (hip, thigh, shank), sub-lane 3 a link of
  // zero mass.  Its constants (mass, COM in the link frame, inertia about the COM) are ten words of the lane's own column of the
  // candidate table (pmc_params.hpp LK_BASE): registers in the occupancy-1 PMC build, LDS reads elsewhere.
  struct LinkC {
    F m;
    V3l com;
    S3<F> ic;
  };
  // the same constants picked from the per-leg table by sub-lane: for callers that hold them in registers across the substep loop.
  // Which of the two is faster is a matter of register pressure, measured per kernel (A/B on one box): the occupancy-1 PMC kernel and both
  // SEPMC kernels are 1 - 7 % faster holding them, the occupancy-2 PMC and EPMC kernels 1 - 10 % faster re-reading the table every substep.
  static LL_HD LinkC own_link_held(const L& ln, const float* legc) {
    const B s0 = ln.is_sub(0), s1 = ln.is_sub(1), s2 = ln.is_sub(2);
    const F zero = ln.lane_f(0.0f);
#define LL_PICK(F0, STRIDE) lm::sel(s0, ln.legc(legc, (F0)), lm::sel(s1, ln.legc(legc, (F0) + (STRIDE)), lm::sel(s2, ln.legc(legc, (F0) + 2 * (STRIDE)), zero)))
    LinkC c;
    c.m = LL_PICK(LC_M, 1);
    c.com = mk3<F>(LL_PICK(LC_COM, 3), LL_PICK(LC_COM + 1, 3), LL_PICK(LC_COM + 2, 3));
    c.ic.xx = LL_PICK(LC_IC, 6); c.ic.xy = LL_PICK(LC_IC + 1, 6); c.ic.xz = LL_PICK(LC_IC + 2, 6);
    c.ic.yy = LL_PICK(LC_IC + 3, 6); c.ic.yz = LL_PICK(LC_IC + 4, 6); c.ic.zz = LL_PICK(LC_IC + 5, 6);
#undef LL_PICK
    return c;
  }
  static LL_HD LinkC own_link(const L& ln) {
    LinkC c;
    c.m = ln.candc(LK_BASE);
    c.com = mk3<F>(ln.candc(LK_BASE + 1), ln.candc(LK_BASE + 2), ln.candc(LK_BASE + 3));
    c.ic.xx = ln.candc(LK_BASE + 4); c.ic.xy = ln.candc(LK_BASE + 5); c.ic.xz = ln.candc(LK_BASE + 6);
    c.ic.yy = ln.candc(LK_BASE + 7); c.ic.yz = ln.candc(LK_BASE + 8); c.ic.zz = ln.candc(LK_BASE + 9);
    return c;
  }
  // inertia of the lane's own link about the F0 origin, F0 axes (R, p: that link's frame)
  static LL_HD RI<F> own_link_inertia(const LinkC& lk, const M3<F>& R, const V3l& p, V3l* com_out, S3<F>* icom_out) {
    V3l c = p + mul(R, lk.com);
    S3<F> ib = rot_sym(R, lk.ic);
    RI<F> I;
    I.m = lk.m;
    I.h = scale(c, lk.m);
    F cc = dot(c, c);
    I.io.xx = ib.xx + lk.m * (cc - c.x * c.x); I.io.xy = ib.xy - lk.m * c.x * c.y; I.io.xz = ib.xz - lk.m * c.x * c.z;
    I.io.yy = ib.yy + lk.m * (cc - c.y * c.y); I.io.yz = ib.yz - lk.m * c.y * c.z; I.io.zz = ib.zz + lk.m * (cc - c.z * c.z);
    *com_out = c;
    *icom_out = ib;
    return I;
  }
  // link inertia about the F0 origin, F0 axes
  static LL_HD RI<F> link_inertia(const L& ln, const float* legc, int k, const M3<F>& R, const V3l& p, V3l* com_out, S3<F>* icom_out) {
    F m = ln.legc(legc, LC_M + k);
    V3l c = p + mul(R, ld3c(ln, legc, LC_COM + 3 * k));
    S3<F> ic;
    ic.xx = ln.legc(legc, LC_IC + 6 * k + 0); ic.xy = ln.legc(legc, LC_IC + 6 * k + 1); ic.xz = ln.legc(legc, LC_IC + 6 * k + 2);
    ic.yy = ln.legc(legc, LC_IC + 6 * k + 3); ic.yz = ln.legc(legc, LC_IC + 6 * k + 4); ic.zz = ln.legc(legc, LC_IC + 6 * k + 5);
    S3<F> ib = rot_sym(R, ic);
    RI<F> I;
    I.m = m;
    I.h = scale(c, m);
    F cc = dot(c, c);
    I.io.xx = ib.xx + m * (cc - c.x * c.x); I.io.xy = ib.xy - m * c.x * c.y; I.io.xz = ib.xz - m * c.x * c.z;
    I.io.yy = ib.yy + m * (cc - c.y * c.y); I.io.yz = ib.yz - m * c.y * c.z; I.io.zz = ib.zz + m * (cc - c.z * c.z);
    *com_out = c;
    *icom_out = ib;
    return I;
  }

  // external (non-gravity) force on a body: Bullet's per-link velocity damping, about the F0 origin
  template <class T>
  static LL_HD SV<T> damping_force(const SV<T>& v, const V3<T>& com, const S3<T>& icom, const T& m, float kd) {
    V3<T> vc = v.l + cross(v.a, com);
    T sv = lm::sqrt_(dot(vc, vc)), sw = lm::sqrt_(dot(v.a, v.a));
    V3<T> f = scale(vc, (kd + kd * sv) * m * (-1.0f));
    V3<T> n = scale(mul(icom, v.a), (kd + kd * sw) * (-1.0f));
    SV<T> r;
    r.a = n + cross(com, f);
    r.l = f;
    return r;
  }

  // 6x6 SPD Cholesky on the packed lower triangle a[i*(i+1)/2 + j], in place; d[i] = 1/L_ii
  static LL_HD void chol6(float* a, float* d) {
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j <= i; j++) {
        float s = a[i * (i + 1) / 2 + j];
        for (int k = 0; k < j; k++) s -= a[i * (i + 1) / 2 + k] * a[j * (j + 1) / 2 + k];
        if (i == j) {
          float r = lm::rsqrt_(s);
          d[i] = r;
          a[i * (i + 1) / 2 + i] = s * r;
        } else {
          a[i * (i + 1) / 2 + j] = s * d[j];
        }
      }
    }
  }
  template <class T>
  static LL_HD void fwd6(const float* a, const float* d, T* x) {   // x <- Lb^-1 x
    for (int i = 0; i < 6; i++) {
      T s = x[i];
      for (int k = 0; k < i; k++) s = s - x[k] * a[i * (i + 1) / 2 + k];
      x[i] = s * d[i];
    }
  }
  template <class T>
  static LL_HD void bwd6(const float* a, const float* d, T* x) {   // x <- Lb^-T x
    for (int i = 5; i >= 0; i--) {
      T s = x[i];
      for (int k = i + 1; k < 6; k++) s = s - x[k] * a[k * (k + 1) / 2 + i];
      x[i] = s * d[i];
    }
  }

  struct LegFactor {   // Cholesky of the lane's 3x3 joint-space inertia + the coupling block
    F l11, l21, l31, l22, l32, l33, i11, i22, i33;   // Lm and 1/diag
    SV<F> y1, y2, y3;                                // columns of Y = M_bl Lm^-T
  };
  static LL_HD void lm_fwd(const LegFactor& f, F* b) {   // b <- Lm^-1 b
    b[0] = b[0] * f.i11;
    b[1] = (b[1] - f.l21 * b[0]) * f.i22;
    b[2] = (b[2] - f.l31 * b[0] - f.l32 * b[1]) * f.i33;
  }
  static LL_HD void lm_bwd(const LegFactor& f, F* u) {   // u <- Lm^-T u
    u[2] = u[2] * f.i33;
    u[1] = (u[1] - f.l32 * u[2]) * f.i22;
    u[0] = (u[0] - f.l21 * u[1] - f.l31 * u[2]) * f.i11;
  }

  static LL_HD void sv_to6(const SV<F>& v, F* o) { o[0] = v.a.x; o[1] = v.a.y; o[2] = v.a.z; o[3] = v.l.x; o[4] = v.l.y; o[5] = v.l.z; }

  // ---------------------------------------------------------------------------------------------------
  // constraint rows: one per lane and round (limit row of joint `sub`, normal / t1 / t2 row of contact slot `sub`),
  // held in registers in whitened coordinates together with the Gram scalars against the 16 rows of the same round
  // ---------------------------------------------------------------------------------------------------
  struct Row {
    F2 ca01, ca23, cb01, cj01, cj23;   // the row's nine coefficients gt[6], jt[3], permuted for its lane (lanes.hpp "scattered velocity state"),
                                       // in register pairs: ca[k] = gt[sub ^ k], cb[k] = gt[4 + ((sub ^ k) & 1)], cj[k] = jt[sub ^ k]
    F c, inv, lam;
    F nk[16];                // -(gt . gt_L + [same leg] jt . jt_L) * inv   for L = lane of the env row
  };
  // x[s ^ k] for k = 0..3 in the lane of sub-lane s (x[3] may be a constant zero): two conditional swaps instead of four 4-way selects
  static LL_HD void permute4(const L& ln, const F& x0, const F& x1, const F& x2, const F& x3, F* c) {
    B odd = lm::odd_(ln.sub()), hi = lm::bit1_(ln.sub());
    F a01 = lm::sel(odd, x1, x0), a10 = lm::sel(odd, x0, x1), a23 = lm::sel(odd, x3, x2), a32 = lm::sel(odd, x2, x3);
    c[0] = lm::sel(hi, a23, a01); c[1] = lm::sel(hi, a32, a10); c[2] = lm::sel(hi, a01, a23); c[3] = lm::sel(hi, a10, a32);
  }
  // Gram scalars of the row against the 16 rows of its round, and the lane-permuted coefficients.  LIMIT: sub-lane 3 holds no row,
  // so the four turns of lanes 3, 7, 11, 15 do not exist and their scalars are not formed.
  template <bool LIMIT>
  static LL_HD void finish_row(const L& ln, Row& r, const F* gt, const F* jt) {
    F zero = ln.lane_f(0.0f);
    F ninv = zero - r.inv;
    F yg[6], yj[3];
    for (int i = 0; i < 6; i++) yg[i] = gt[i] * ninv;
    for (int i = 0; i < 3; i++) yj[i] = jt[i] * ninv;
    F nj[4];
    nj[0] = yj[0] * L::template subbcast<0>(jt[0]) + yj[1] * L::template subbcast<0>(jt[1]) + yj[2] * L::template subbcast<0>(jt[2]);
    nj[1] = yj[0] * L::template subbcast<1>(jt[0]) + yj[1] * L::template subbcast<1>(jt[1]) + yj[2] * L::template subbcast<1>(jt[2]);
    nj[2] = yj[0] * L::template subbcast<2>(jt[0]) + yj[1] * L::template subbcast<2>(jt[1]) + yj[2] * L::template subbcast<2>(jt[2]);
    if (!LIMIT) nj[3] = yj[0] * L::template subbcast<3>(jt[0]) + yj[1] * L::template subbcast<3>(jt[1]) + yj[2] * L::template subbcast<3>(jt[2]);
    if constexpr (L::kGram16) {
      // the base part of all 16 scalars as ONE 16 x 16 x 6 product on the matrix cores (lanes.hpp gram16); the rows of the own leg also meet in the joints
      F d[16];
      L::gram16(gt, yg, d);
      const F one = ln.lane_f(1.0f);
      const F lf[4] = {lm::sel(ln.is_leg(0), one, zero), lm::sel(ln.is_leg(1), one, zero), lm::sel(ln.is_leg(2), one, zero), lm::sel(ln.is_leg(3), one, zero)};
      for (int L_ = 0; L_ < 16; L_++)
        if (!(LIMIT && (L_ & 3) == 3)) r.nk[L_] = d[L_] + lf[L_ >> 2] * nj[L_ & 3];
    } else {
      for (int L_ = 0; L_ < 16; L_++)
        if (!(LIMIT && (L_ & 3) == 3)) r.nk[L_] = lm::sel(ln.is_leg(L_ >> 2), nj[L_ & 3], zero);      // the rows of the own leg also meet in the joints
      L::template gram4<0>(gt, yg, r.nk);                          // nk[L] += sum_i yg[i] * gt_L[i]
      L::template gram4<1>(gt, yg, r.nk);
      L::template gram4<2>(gt, yg, r.nk);
      if (!LIMIT) L::template gram4<3>(gt, yg, r.nk);
    }
    r.lam = zero;
    F t[4];
    permute4(ln, gt[0], gt[1], gt[2], gt[3], t);
    r.ca01 = L::pair(t[0], t[1]); r.ca23 = L::pair(t[2], t[3]);
    B odd = lm::odd_(ln.sub());
    r.cb01 = L::pair(lm::sel(odd, gt[5], gt[4]), lm::sel(odd, gt[4], gt[5]));
    permute4(ln, jt[0], jt[1], jt[2], zero, t);
    r.cj01 = L::pair(t[0], t[1]); r.cj23 = L::pair(t[2], t[3]);
  }
  // ---- cone-coupled friction (LLM_SPEC_FRICTION_MODE = 2: the published default of btMultiBodyConstraintSolver, resolveConeFrictionConstraintRows;
  //      the default of every step kernel since round 4, the pyramid of rounds 1 - 3 stays as mode 0 -- DESIGN.md 4).  The two friction rows of a contact are
  //      solved TOGETHER: both increments from the same velocity, the pair scaled back onto |(t1, t2)| <= mu N, both applied.  Besides the Gram
  //      scalars of each row kind (Row::nk) that takes the CROSS scalars between the t1 and t2 rows of different lanes:
  struct ConeX {
    F n12[16];               // -(gt1 . gt2_L + [same leg] jt1 . jt2_L) * inv1 : what lane L's t2 increment does to my t1 row's pending increment
    F n21[16];               // -(gt2 . gt1_L + [same leg] jt2 . jt1_L) * inv2
  };
  // out[L] = -(gt_mine . gt_oth(lane L) + [same leg] jt_mine . jt_oth(lane L)) * inv_mine   (finish_row's scalars, between two row kinds)
  static LL_HD void cross_gram(const L& ln, const F& inv_mine, const F* gt_mine, const F* jt_mine, const F* gt_oth, const F* jt_oth, F* out) {
    F zero = ln.lane_f(0.0f);
    F ninv = zero - inv_mine;
    F yg[6], yj[3];
    for (int i = 0; i < 6; i++) yg[i] = gt_mine[i] * ninv;
    for (int i = 0; i < 3; i++) yj[i] = jt_mine[i] * ninv;
    F nj[4];
    nj[0] = yj[0] * L::template subbcast<0>(jt_oth[0]) + yj[1] * L::template subbcast<0>(jt_oth[1]) + yj[2] * L::template subbcast<0>(jt_oth[2]);
    nj[1] = yj[0] * L::template subbcast<1>(jt_oth[0]) + yj[1] * L::template subbcast<1>(jt_oth[1]) + yj[2] * L::template subbcast<1>(jt_oth[2]);
    nj[2] = yj[0] * L::template subbcast<2>(jt_oth[0]) + yj[1] * L::template subbcast<2>(jt_oth[1]) + yj[2] * L::template subbcast<2>(jt_oth[2]);
    nj[3] = yj[0] * L::template subbcast<3>(jt_oth[0]) + yj[1] * L::template subbcast<3>(jt_oth[1]) + yj[2] * L::template subbcast<3>(jt_oth[2]);
    if constexpr (L::kGram16) {
      F d[16];
      L::gram16(gt_oth, yg, d);
      const F one = ln.lane_f(1.0f);
      const F lf[4] = {lm::sel(ln.is_leg(0), one, zero), lm::sel(ln.is_leg(1), one, zero), lm::sel(ln.is_leg(2), one, zero), lm::sel(ln.is_leg(3), one, zero)};
      for (int L_ = 0; L_ < 16; L_++) out[L_] = d[L_] + lf[L_ >> 2] * nj[L_ & 3];
    } else {
      for (int L_ = 0; L_ < 16; L_++) out[L_] = lm::sel(ln.is_leg(L_ >> 2), nj[L_ & 3], zero);
      L::template gram4<0>(gt_oth, yg, out);
      L::template gram4<1>(gt_oth, yg, out);
      L::template gram4<2>(gt_oth, yg, out);
      L::template gram4<3>(gt_oth, yg, out);
    }
  }
  // One cone-coupled round over the 16 friction pairs, turns slot-major like every round.  The state of a pair during the round is
  // S = lambda + pending increment (lambda itself only moves at the commit): lanes.hpp cone_turns4 has the turn.
  // (Skipping the four turns of a contact slot no env of the wave uses -- they commit exact zeros -- was measured slower here too, 13 issue slots
  // a turn notwithstanding: 0.1929 against 0.1887 ms per control step, profiles/r04_cone_ab.txt.)
  static LL_HD void gs_cone_round(const L& ln, Row& r1, Row& r2, const ConeX& cx, const F& lim, F& VA, F& VB, F& VJ) {
    F zero = ln.lane_f(0.0f);
    F w1 = L::vel_dot(r1.c, r1.ca01, r1.ca23, r1.cb01, r1.cj01, r1.cj23, VA, VB, VJ);
    F w2 = L::vel_dot(r2.c, r2.ca01, r2.ca23, r2.cb01, r2.cj01, r2.cj23, VA, VB, VJ);
    F S1 = lm::nfma_(w1, r1.inv, r1.lam), S2 = lm::nfma_(w2, r2.inv, r2.lam);
    F d1 = zero, d2 = zero;
    if (L::kConeInLds) {
      // the cross scalars of turn block S come from the row's LDS scratch (substep_impl put them there), one block ahead of their use
      F a[16], b[16];
      ln.cone_load(0, 0, a[0], a[4], a[8], a[12]); ln.cone_load(1, 0, b[0], b[4], b[8], b[12]);
      ln.cone_load(0, 1, a[1], a[5], a[9], a[13]); ln.cone_load(1, 1, b[1], b[5], b[9], b[13]);
      ln.template cone_turns4<0>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, a, b, r2.nk);
      ln.cone_load(0, 2, a[2], a[6], a[10], a[14]); ln.cone_load(1, 2, b[2], b[6], b[10], b[14]);
      ln.template cone_turns4<1>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, a, b, r2.nk);
      ln.cone_load(0, 3, a[3], a[7], a[11], a[15]); ln.cone_load(1, 3, b[3], b[7], b[11], b[15]);
      ln.template cone_turns4<2>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, a, b, r2.nk);
      ln.template cone_turns4<3>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, a, b, r2.nk);
    } else {
      ln.template cone_turns4<0>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, cx.n12, cx.n21, r2.nk);
      ln.template cone_turns4<1>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, cx.n12, cx.n21, r2.nk);
      ln.template cone_turns4<2>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, cx.n12, cx.n21, r2.nk);
      ln.template cone_turns4<3>(S1, S2, d1, d2, r1.lam, r2.lam, lim, r1.nk, cx.n12, cx.n21, r2.nk);
    }
    L::vel_commit2(d1, r1.lam, r1.ca01, r1.ca23, r1.cb01, r1.cj01, r1.cj23, d2, r2.lam, r2.ca01, r2.ca23, r2.cb01, r2.cj01, r2.cj23, VA, VB, VJ);   // (also lam += d)
  }

  template <int K_>
  static LL_HD void pick_rank(const L& ln, const F& rank, const F& me, const F& d, const F& sb, const F& jj, F& nd, F& ns, F& nj) {
    B take = lm::abs_(L::template subbcast<K_>(rank) - me) < 0.5f;
    nd = lm::sel(take, L::template subbcast<K_>(d), nd);
    ns = lm::sel(take, L::template subbcast<K_>(sb), ns);
    nj = lm::sel(take, L::template subbcast<K_>(jj), nj);
  }
  // One Gauss-Seidel round over the 16 rows of a kind.  The velocity the rows act on -- whitened base twist dx, whitened joint
  // rates dq -- is carried in the three scattered registers VA, VB, VJ (lanes.hpp): the round reads it with ten multiply-adds
  // (vel_dot), takes its sixteen turns, and folds the committed increments back with a 25-instruction transpose-reduce (vel_commit).
  // A turn: the lane whose turn it is commits clamp(u) -- every lane's pending increment then moves by nk[L] * d_L.  All turns run
  // unconditionally: a lane without a live row has inv = 0 and commits exactly zero.  (Skipping the 4-turn blocks no env of the wave
  // occupies was measured slower: each wave-uniform test costs more issue slots than the 16 instructions it occasionally saves.)
  // UNILATERAL (limit and normal rows: 0 <= lambda): the admissible increment is [-lambda, inf) -- the turns take the multiplier with
  // the instruction's source negation and `hi` (a huge constant) as it stands, no bound arithmetic; friction rows pass [-hi, hi].
  template <bool LIMIT, bool UNILATERAL>
  static LL_HD void gs_round(const L& ln, Row& r, const F& hi, F& VA, F& VB, F& VJ) {
    F zero = ln.lane_f(0.0f);
    F w = L::vel_dot(r.c, r.ca01, r.ca23, r.cb01, r.cj01, r.cj23, VA, VB, VJ);
    F u = (zero - w) * r.inv;                             // unclamped increment; admissible interval [lo - lam, hi - lam]
    F lo_d = UNILATERAL ? r.lam : (zero - hi) - r.lam, hi_d = UNILATERAL ? hi : hi - r.lam;
    F dl = zero;
    ln.template turns8<0, UNILATERAL>(u, dl, lo_d, hi_d, r.nk);
    if (LIMIT) ln.template turns4<2, UNILATERAL>(u, dl, lo_d, hi_d, r.nk[2], r.nk[6], r.nk[10], r.nk[14]);
    else ln.template turns8<1, UNILATERAL>(u, dl, lo_d, hi_d, r.nk);
    L::vel_commit(dl, r.lam, r.ca01, r.ca23, r.cb01, r.cj01, r.cj23, VA, VB, VJ);      // (also lam += dl)
  }

  // ---------------------------------------------------------------------------------------------------
  // one physics substep
  // ---------------------------------------------------------------------------------------------------
  // what an env other than PMC adds to a substep (EPMC, epmc_step.hpp): the episode's foot friction coefficient and the
  // push of randomizer/push_randomizer.py:72-77 -- applyExternalForce(linkIndex 0, LINK_FRAME): a force given in the FR hip
  // link's frame, acting at that link's centre of mass
  struct SubstepExtra {
    float mu_foot;
    bool has_push;
    float push[3];
    // terrain within reach of the robot (TERRAIN builds only): n_shapes records  x0 x1 y0 y1 | z0 z1 rod r  of axis-aligned
    // boxes; rod = +1 / -1 / 0: thin cylinders of radius r along y on the two top / bottom x-edges (BSE:43-104), or none
    const float* shapes;
    int n_shapes;
    float box_mu_scale;   // friction of a box relative to the plane's (default lateralFriction 0.5 vs plane.urdf 0.9)
    // SEPMC (sepmc_step.hpp): when want_touch is set the substep also reports whether a leg / wheel link -- any link below the
    // trunk except the foot sphere, CTG:427 -- has a contact point (within the margin) with the plane or a box other than
    // shapes[flag_shape], and with shapes[flag_shape] (the flag); what getContactPoints() would list after this substep
    bool want_touch;
    int flag_shape;       // index into shapes, -1: none
    mutable float touch_static, touch_flag;
    // PAIR builds (SEPMC): the other robot of the arena lives in the neighbouring row; pair_active = the two are close enough to
    // touch during this control step (the same value in both rows), pair_me = 0 / 1.  touch_robot (with want_touch): one of my leg /
    // wheel links -- a thigh capsule, or a shank capsule away from its foot end -- is within the margin of the other robot
    bool pair_active;
    int pair_me;
    mutable float touch_robot;
    // the PMC jump obstacle (PLE:182-193): ONE shape whose record is given in a frame turned by yaw about z around (ycx, ycy)
    bool yawed = false;
    float ycx = 0.0f, ycy = 0.0f, ycs = 1.0f, ysn = 0.0f;
  };
  // shape_sdf of record si at world point E, for terrain that may be yawed: distance, outward normal in WORLD coordinates
  template <class T>
  static LL_HD void terrain_sdf(const L& ln, const SubstepExtra* ex, int si, const V3<T>& E, T& d, V3<T>& n, T& is_box) {
    if (!ex->yawed) { shape_sdf<T>(ln, ex->shapes + si * 8, E, d, n, is_box); return; }
    const T dx = E.x - ex->ycx, dy = E.y - ex->ycy;
    V3<T> El = mk3<T>(dx * ex->ycs + dy * ex->ysn, dy * ex->ycs - dx * ex->ysn, E.z), nl;
    shape_sdf<T>(ln, ex->shapes + si * 8, El, d, nl, is_box);
    n = mk3<T>(nl.x * ex->ycs - nl.y * ex->ysn, nl.x * ex->ysn + nl.y * ex->ycs, nl.z);
  }
  template <class T>
  static LL_HD void terrain_sdf_rec(const L& ln, const SubstepExtra* ex, const BoxRec& rec, const V3<T>& E, T& d, V3<T>& n, T& is_box) {
    if (!ex->yawed) { shape_sdf_rec<T>(ln, rec, E, d, n, is_box); return; }
    const T dx = E.x - ex->ycx, dy = E.y - ex->ycy;
    V3<T> El = mk3<T>(dx * ex->ycs + dy * ex->ysn, dy * ex->ycs - dx * ex->ysn, E.z), nl;
    shape_sdf_rec<T>(ln, rec, El, d, nl, is_box);
    n = mk3<T>(nl.x * ex->ycs - nl.y * ex->ysn, nl.x * ex->ysn + nl.y * ex->ycs, nl.z);
  }
  // Signed distance of point E to record s (box united with its edge rods) and the outward normal there.  Inside a box the
  // face of least penetration gives both; outside, the nearest point of the box does.
  template <class T>
  static LL_HD void shape_sdf(const L& ln, const float* s, const V3<T>& E, T& d, V3<T>& n, T& is_box) {
    shape_sdf_rec<T>(ln, load_box(s), E, d, n, is_box);             // two 16-byte reads instead of eight scalar ones
  }
  // (the record already in registers: where the lane policy says so -- lanes.hpp WithShapePrefetch, chosen per kernel by A/B -- the candidate
  //  loops read it one shape ahead, so that the LDS round trip hides behind the previous shape's arithmetic or, for the first shape, behind
  //  the candidate's own kinematics)
  template <class T>
  static LL_HD void shape_sdf_rec(const L& ln, const BoxRec& rec, const V3<T>& E, T& d, V3<T>& n, T& is_box) {
    const T zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f), neg = ln.lane_f(-1.0f);
    T ax0 = ln.lane_f(rec.a.x) - E.x, ax1 = E.x - ln.lane_f(rec.a.y);
    T ay0 = ln.lane_f(rec.a.z) - E.y, ay1 = E.y - ln.lane_f(rec.a.w);
    T az0 = ln.lane_f(rec.c.x) - E.z, az1 = E.z - ln.lane_f(rec.c.y);
    T qx = lm::max_(ax0, ax1), qy = lm::max_(ay0, ay1), qz = lm::max_(az0, az1);
    T sx = lm::sel(ax1 > ax0, one, neg), sy = lm::sel(ay1 > ay0, one, neg), sz = lm::sel(az1 > az0, one, neg);
    d = qx; n = mk3<T>(sx, zero, zero);
    B by = qy > d;
    d = lm::sel(by, qy, d); n = mk3<T>(lm::sel(by, zero, n.x), lm::sel(by, sy, zero), zero);
    B bz = qz > d;
    d = lm::sel(bz, qz, d); n = mk3<T>(lm::sel(bz, zero, n.x), lm::sel(bz, zero, n.y), lm::sel(bz, sz, zero));
    is_box = one;
    {   // outside the box: Euclidean distance and the direction away from its nearest point (over a face the same as above; diagonally
        // outside an edge or corner the rounded distance -- a sphere or a link's mid-span meets an edge with the normal through its centre)
      T ox = lm::max_(qx, zero), oy = lm::max_(qy, zero), oz = lm::max_(qz, zero);
      T e2 = ox * ox + oy * oy + oz * oz;
      B outside = e2 > 0.0f;
      T e = lm::sqrt_(e2), ie = one / lm::max_(e, ln.lane_f(1e-30f));
      d = lm::sel(outside, e, d);
      n = mk3<T>(lm::sel(outside, sx * ox * ie, n.x), lm::sel(outside, sy * oy * ie, n.y), lm::sel(outside, sz * oz * ie, n.z));
    }
    if (rec.c.z != 0.0f) {
      const float ze = rec.c.z > 0.0f ? rec.c.y : rec.c.x, rr = rec.c.w;
      B iny = lm::and_(qy <= 0.0f, one > zero);
      for (int e = 0; e < 2; e++) {
        T dx = E.x - ln.lane_f(e == 0 ? rec.a.x : rec.a.y), dz = E.z - ln.lane_f(ze);
        T len = lm::sqrt_(dx * dx + dz * dz);
        T dr = len - rr;
        B better = lm::and_(iny, lm::and_(dr < d, len > 1e-6f));
        T il = one / lm::max_(len, ln.lane_f(1e-6f));
        d = lm::sel(better, dr, d);
        n = mk3<T>(lm::sel(better, dx * il, n.x), lm::sel(better, zero, n.y), lm::sel(better, dz * il, n.z));
        is_box = lm::sel(better, zero, is_box);
      }
    }
  }
  // Reverse candidates (DESIGN.md 8 "edges under the trunk"; the oracle's reverse_edge states the rule): the robot's own candidates are
  // vertices and spheres, blind to a step edge that crosses the flat of the body box between its corners.  Leg l tests top edge l of every
  // terrain box (0: x = x0, 1: x = x1, 2: y = y0, 3: y = y1, at z = z1 -- at z = z0 for a box that floats) against the body box in the box's frame: the edge is cut to the box
  // grown by the margin; the middle of what is left names the face it runs along; the edge cut to the exact extents of the other two
  // axes has two ends -- end `endf` (0 / 1) is this lane's candidate: depth = signed distance to the face's plane, point = the point of
  // the terrain edge (returned in F0), normal = the face's inward normal (returned in world coordinates).  Deepest over the boxes.
  static LL_HD void reverse_edge(const L& ln, const StepParams& P, const SubstepExtra* ex, const Base& bs, const M3<float>& R, const F& endf,
                                 F& depth, V3l& Pb, V3l& nw) {
    const F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f), far_ = ln.lane_f(1.0e30f);
    const float* bc = P.basec + BC_BOX;
    float h[3], W[9];                           // half extents; rows of W: the box's unit axes in world coordinates
    for (int a = 0; a < 3; a++) {
      const float ux = bc[3 + 3 * a], uy = bc[4 + 3 * a], uz = bc[5 + 3 * a];
      h[a] = sqrtf(ux * ux + uy * uy + uz * uz);
      const float ih = 1.0f / h[a];
      W[3 * a + 0] = (R.m[0] * ux + R.m[1] * uy + R.m[2] * uz) * ih;
      W[3 * a + 1] = (R.m[3] * ux + R.m[4] * uy + R.m[5] * uz) * ih;
      W[3 * a + 2] = (R.m[6] * ux + R.m[7] * uy + R.m[8] * uz) * ih;
    }
    const float cwx = bs.p.x + R.m[0] * bc[0] + R.m[1] * bc[1] + R.m[2] * bc[2], cwy = bs.p.y + R.m[3] * bc[0] + R.m[4] * bc[1] + R.m[5] * bc[2],
                cwz = bs.p.z + R.m[6] * bc[0] + R.m[7] * bc[1] + R.m[8] * bc[2];
    const float reach = sqrtf(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]) + P.margin_dist;
    const B l0 = ln.is_leg(0), l1 = ln.is_leg(1), l2 = ln.is_leg(2), l3 = ln.is_leg(3), end1 = endf > 0.5f;
    depth = far_;
    V3l Pw = mk3<F>(zero, zero, zero);
    nw = mk3<F>(zero, zero, one);
    for (int si = 0; si < ex->n_shapes; si++) {
      const BoxRec rec = load_box(ex->shapes + si * 8);
      // (skipping boxes no env of the wave has its trunk near is an optimisation only: an edge point within the margin of the body box
      //  is within `reach` of its centre)
      float lx = cwx, ly = cwy;
      if (ex->yawed) { const float dx = cwx - ex->ycx, dy = cwy - ex->ycy; lx = dx * ex->ycs + dy * ex->ysn; ly = dy * ex->ycs - dx * ex->ysn; }
      const float ze = rec.c.x > (float)LLM_FLOATING_MIN_Z ? rec.c.x : rec.c.y;      // a floating box (a hanging bar, BSE:366-412) offers its BOTTOM edges (round 5)
      const bool near = lx > rec.a.x - reach && lx < rec.a.y + reach && ly > rec.a.z - reach && ly < rec.a.w + reach && fabsf(cwz - ze) < reach;
      if (!L::any(ln.lane_f(near ? 1.0f : 0.0f) > 0.5f)) continue;
      F ax_ = lm::sel(l1, ln.lane_f(rec.a.y), ln.lane_f(rec.a.x)), ay_ = lm::sel(l3, ln.lane_f(rec.a.w), ln.lane_f(rec.a.z));
      F bx_ = lm::sel(l0, ln.lane_f(rec.a.x), ln.lane_f(rec.a.y)), by_ = lm::sel(l2, ln.lane_f(rec.a.z), ln.lane_f(rec.a.w));
      V3l aw = mk3<F>(ax_, ay_, ln.lane_f(ze)), bw = mk3<F>(bx_, by_, ln.lane_f(ze));
      if (ex->yawed) {
        aw = mk3<F>(ln.lane_f(ex->ycx) + ax_ * ex->ycs - ay_ * ex->ysn, ln.lane_f(ex->ycy) + ax_ * ex->ysn + ay_ * ex->ycs, aw.z);
        bw = mk3<F>(ln.lane_f(ex->ycx) + bx_ * ex->ycs - by_ * ex->ysn, ln.lane_f(ex->ycy) + bx_ * ex->ysn + by_ * ex->ycs, bw.z);
      }
      V3l ra = mk3<F>(aw.x - cwx, aw.y - cwy, aw.z - cwz), dw = bw - aw;
      F pa[3], dd[3], lo[3], hi[3];
      F t0 = zero, t1 = one;
      for (int i = 0; i < 3; i++) {
        pa[i] = ra.x * W[3 * i] + ra.y * W[3 * i + 1] + ra.z * W[3 * i + 2];
        dd[i] = dw.x * W[3 * i] + dw.y * W[3 * i + 1] + dw.z * W[3 * i + 2];
        F inv = one / lm::sel(lm::abs_(dd[i]) < 1e-9f, ln.lane_f(1e-9f), dd[i]);
        const float H = h[i] + P.margin_dist;
        F ta = (ln.lane_f(-H) - pa[i]) * inv, tb = (ln.lane_f(H) - pa[i]) * inv;
        t0 = lm::max_(t0, lm::min_(ta, tb)); t1 = lm::min_(t1, lm::max_(ta, tb));
        F ea = (ln.lane_f(-h[i]) - pa[i]) * inv, eb = (ln.lane_f(h[i]) - pa[i]) * inv;
        lo[i] = lm::min_(ea, eb); hi[i] = lm::max_(ea, eb);
      }
      F tm = (t0 + t1) * 0.5f;
      F pm0 = pa[0] + tm * dd[0], pm1 = pa[1] + tm * dd[1], pm2 = pa[2] + tm * dd[2];
      F q0 = lm::abs_(pm0) - h[0], q1 = lm::abs_(pm1) - h[1], q2 = lm::abs_(pm2) - h[2];
      B is1 = q1 > q0;
      F q = lm::sel(is1, q1, q0);
      B is2 = q2 > q;
      B a0 = lm::and_(lm::not_(is1), lm::not_(is2)), a1 = lm::and_(is1, lm::not_(is2));
      F pmx = lm::sel(is2, pm2, lm::sel(is1, pm1, pm0));
      F sg = lm::sel(pmx >= 0.0f, one, zero - one);
      F u0 = lm::max_(zero, lm::sel(a0, lm::max_(lo[1], lo[2]), lm::sel(a1, lm::max_(lo[0], lo[2]), lm::max_(lo[0], lo[1]))));
      F u1 = lm::min_(one, lm::sel(a0, lm::min_(hi[1], hi[2]), lm::sel(a1, lm::min_(hi[0], hi[2]), lm::min_(hi[0], hi[1]))));
      B valid = lm::and_(t0 <= t1, u0 <= u1);
      F u = lm::sel(end1, u1, u0);
      F pax = lm::sel(is2, pa[2], lm::sel(is1, pa[1], pa[0])), dax = lm::sel(is2, dd[2], lm::sel(is1, dd[1], dd[0]));
      F hax = lm::sel(is2, ln.lane_f(h[2]), lm::sel(is1, ln.lane_f(h[1]), ln.lane_f(h[0])));
      F dep = sg * (pax + u * dax) - hax;
      B better = lm::and_(valid, dep < depth);
      depth = lm::sel(better, dep, depth);
      Pw = mk3<F>(lm::sel(better, aw.x + u * dw.x, Pw.x), lm::sel(better, aw.y + u * dw.y, Pw.y), lm::sel(better, aw.z + u * dw.z, Pw.z));
      F nsg = zero - sg;
      nw = mk3<F>(lm::sel(better, nsg * lm::sel(is2, ln.lane_f(W[6]), lm::sel(is1, ln.lane_f(W[3]), ln.lane_f(W[0]))), nw.x),
                  lm::sel(better, nsg * lm::sel(is2, ln.lane_f(W[7]), lm::sel(is1, ln.lane_f(W[4]), ln.lane_f(W[1]))), nw.y),
                  lm::sel(better, nsg * lm::sel(is2, ln.lane_f(W[8]), lm::sel(is1, ln.lane_f(W[5]), ln.lane_f(W[2]))), nw.z));
    }
    Pb = mulT(R, mk3<F>(Pw.x - bs.p.x, Pw.y - bs.p.y, Pw.z - bs.p.z));
  }
  // Terrain edges against the flat faces of the LEG boxes (round 6: the mirror of reverse_edge for thigh and shank; the oracle's leg_reverse_edge states the rule).  The robot's own
  // candidates on a leg are vertices, rim points and two mid-span spheres per link: a shank laid ACROSS the edge of a hurdle between them sinks until one of those arrives.  So lane
  // (leg l, sub-lane s) tests edge s (0: x = x0, 1: x = x1, 2: y = y0, 3: y = y1; at the top z1, or at the bottom z0 of a box that floats) of every listed terrain box against the
  // leg's thigh box and shank box, each in the box's own frame exactly as reverse_edge does for the body box: the edge cut to the box grown by the margin, the middle of the rest
  // naming the face, the piece cut to the other two axes, BOTH its ends evaluated.  The lane's candidate (jj = 9) is the deepest of all of them: depth = signed distance to the
  // face's plane, point = the point of the terrain edge (returned in F0), normal = the face's inward normal (world), link = 2 (thigh) / 3 (shank).
  // edgef: which edge, as a float (the lane's own sub-lane when candidates are collected; the kept candidate's sub-lane when its row is built).
  static LL_HD void leg_edge(const L& ln, const StepParams& P, const SubstepExtra* ex, const Base& bs, const M3<float>& R, const LegKin& k, const float* legc, const F& edgef,
                             F& depth, V3l& Pb, V3l& nw, F& link, F& shape) {
    const F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f), far_ = ln.lane_f(1.0e30f);
    shape = ln.lane_f(-1.0f);
    const B e0 = edgef < 0.5f, e1 = lm::and_(edgef > 0.5f, edgef < 1.5f), e2 = lm::and_(edgef > 1.5f, edgef < 2.5f), e3 = edgef > 2.5f;
    depth = far_;
    link = ln.lane_f(3.0f);
    V3l Pw = mk3<F>(zero, zero, zero);
    nw = mk3<F>(zero, zero, one);
    const float cwx = bs.p.x, cwy = bs.p.y, cwz = bs.p.z;
    for (int lb = 0; lb < 2; lb++) {                        // 0: the thigh box, 1: the shank box
      const M3<F>& Rk = lb == 0 ? k.R2 : k.R3;
      const V3l& pk = lb == 0 ? k.p2 : k.p3;
      const int f0 = lb == 0 ? LC_THBOX : LC_SHBOX;
      // the box in world coordinates: centre, unit axes (rows of W), half extents
      const V3l cb = pk + mul(Rk, ld3c(ln, legc, f0));
      const V3l cwl = mul(R, cb);
      const V3l cw = mk3<F>(cwl.x + cwx, cwl.y + cwy, cwl.z + cwz);
      F h[3], W[9];
      for (int a = 0; a < 3; a++) {
        const V3l u = ld3c(ln, legc, f0 + 3 + 3 * a);
        h[a] = lm::sqrt_(dot(u, u));
        const V3l ub = mul(Rk, u), uw = mul(R, ub);
        const F ih = one / h[a];
        W[3 * a + 0] = uw.x * ih; W[3 * a + 1] = uw.y * ih; W[3 * a + 2] = uw.z * ih;
      }
      const F reach = lm::sqrt_(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]) + P.margin_dist;
      for (int si = 0; si < ex->n_shapes; si++) {
        const BoxRec rec = load_box(ex->shapes + si * 8);
        F lx = cw.x, ly = cw.y;
        if (ex->yawed) { const F dx = cw.x - ex->ycx, dy = cw.y - ex->ycy; lx = dx * ex->ycs + dy * ex->ysn; ly = dy * ex->ycs - dx * ex->ysn; }
        const float ze = rec.c.x > (float)LLM_FLOATING_MIN_Z ? rec.c.x : rec.c.y;
        const B near = lm::and_(lm::and_(lm::and_(lx > rec.a.x - reach, lx < rec.a.y + reach), lm::and_(ly > rec.a.z - reach, ly < rec.a.w + reach)), lm::abs_(cw.z - ze) < reach);
        if (!L::any(near)) continue;                        // (an optimisation only: an edge point within the margin of the box is within `reach` of its centre)
        F ax_ = lm::sel(e1, ln.lane_f(rec.a.y), ln.lane_f(rec.a.x)), ay_ = lm::sel(e3, ln.lane_f(rec.a.w), ln.lane_f(rec.a.z));
        F bx_ = lm::sel(e0, ln.lane_f(rec.a.x), ln.lane_f(rec.a.y)), by_ = lm::sel(e2, ln.lane_f(rec.a.z), ln.lane_f(rec.a.w));
        V3l aw = mk3<F>(ax_, ay_, ln.lane_f(ze)), bw = mk3<F>(bx_, by_, ln.lane_f(ze));
        if (ex->yawed) {
          aw = mk3<F>(ln.lane_f(ex->ycx) + ax_ * ex->ycs - ay_ * ex->ysn, ln.lane_f(ex->ycy) + ax_ * ex->ysn + ay_ * ex->ycs, aw.z);
          bw = mk3<F>(ln.lane_f(ex->ycx) + bx_ * ex->ycs - by_ * ex->ysn, ln.lane_f(ex->ycy) + bx_ * ex->ysn + by_ * ex->ycs, bw.z);
        }
        V3l ra = aw - cw, dw = bw - aw;
        F pa[3], dd[3], lo[3], hi[3];
        F t0 = zero, t1 = one;
        for (int i = 0; i < 3; i++) {
          pa[i] = ra.x * W[3 * i] + ra.y * W[3 * i + 1] + ra.z * W[3 * i + 2];
          dd[i] = dw.x * W[3 * i] + dw.y * W[3 * i + 1] + dw.z * W[3 * i + 2];
          F inv = one / lm::sel(lm::abs_(dd[i]) < 1e-9f, ln.lane_f(1e-9f), dd[i]);
          const F H = h[i] + P.margin_dist;
          F ta = ((zero - H) - pa[i]) * inv, tb = (H - pa[i]) * inv;
          t0 = lm::max_(t0, lm::min_(ta, tb)); t1 = lm::min_(t1, lm::max_(ta, tb));
          F ea = ((zero - h[i]) - pa[i]) * inv, eb = (h[i] - pa[i]) * inv;
          lo[i] = lm::min_(ea, eb); hi[i] = lm::max_(ea, eb);
        }
        F tm = (t0 + t1) * 0.5f;
        F pm0 = pa[0] + tm * dd[0], pm1 = pa[1] + tm * dd[1], pm2 = pa[2] + tm * dd[2];
        F q0 = lm::abs_(pm0) - h[0], q1 = lm::abs_(pm1) - h[1], q2 = lm::abs_(pm2) - h[2];
        {   // the face must be one the edge runs ACROSS: the box axis the edge is most parallel to is no candidate (a thin leg box is pierced lengthwise through its two small faces)
          const F d0 = lm::abs_(dd[0]), d1 = lm::abs_(dd[1]), d2 = lm::abs_(dd[2]);
          const B par0 = lm::and_(d0 >= d1, d0 >= d2), par1 = lm::and_(lm::not_(par0), d1 >= d2);
          const B par2 = lm::and_(lm::not_(par0), lm::not_(par1));
          const F never = ln.lane_f(-3.0e38f);
          q0 = lm::sel(par0, never, q0); q1 = lm::sel(par1, never, q1); q2 = lm::sel(par2, never, q2);
        }
        B is1 = q1 > q0;
        F q = lm::sel(is1, q1, q0);
        B is2 = q2 > q;
        B a0 = lm::and_(lm::not_(is1), lm::not_(is2)), a1 = lm::and_(is1, lm::not_(is2));
        F pmx = lm::sel(is2, pm2, lm::sel(is1, pm1, pm0));
        F sg = lm::sel(pmx >= 0.0f, one, zero - one);
        F u0 = lm::max_(zero, lm::sel(a0, lm::max_(lo[1], lo[2]), lm::sel(a1, lm::max_(lo[0], lo[2]), lm::max_(lo[0], lo[1]))));
        F u1 = lm::min_(one, lm::sel(a0, lm::min_(hi[1], hi[2]), lm::sel(a1, lm::min_(hi[0], hi[2]), lm::min_(hi[0], hi[1]))));
        B valid = lm::and_(lm::and_(t0 <= t1, u0 <= u1), near);
        F pax = lm::sel(is2, pa[2], lm::sel(is1, pa[1], pa[0])), dax = lm::sel(is2, dd[2], lm::sel(is1, dd[1], dd[0]));
        F hax = lm::sel(is2, h[2], lm::sel(is1, h[1], h[0]));
        F dep0 = sg * (pax + u0 * dax) - hax, dep1 = sg * (pax + u1 * dax) - hax;
        B second = dep1 < dep0;                              // the deeper of the piece's two ends (the first on a tie)
        F dep = lm::sel(second, dep1, dep0), u = lm::sel(second, u1, u0);
        B better = lm::and_(valid, dep < depth);
        depth = lm::sel(better, dep, depth);
        link = lm::sel(better, ln.lane_f(lb == 0 ? 2.0f : 3.0f), link);
        shape = lm::sel(better, ln.lane_f((float)si), shape);
        Pw = mk3<F>(lm::sel(better, aw.x + u * dw.x, Pw.x), lm::sel(better, aw.y + u * dw.y, Pw.y), lm::sel(better, aw.z + u * dw.z, Pw.z));
        F nsg = zero - sg;
        nw = mk3<F>(lm::sel(better, nsg * lm::sel(is2, W[6], lm::sel(is1, W[3], W[0])), nw.x),
                    lm::sel(better, nsg * lm::sel(is2, W[7], lm::sel(is1, W[4], W[1])), nw.y),
                    lm::sel(better, nsg * lm::sel(is2, W[8], lm::sel(is1, W[5], W[2])), nw.z));
      }
    }
    Pb = mulT(R, mk3<F>(Pw.x - bs.p.x, Pw.y - bs.p.y, Pw.z - bs.p.z));
  }
  // closest points of the segments p1-q1 and p2-q2 (Ericson 5.1.9; same branches as the oracle's seg_seg)
  static LL_HD void seg_seg(const L& ln, const V3l& p1, const V3l& q1, const V3l& p2, const V3l& q2, V3l& c1, V3l& c2) {
    F s, t;
    seg_seg_st(ln, p1, q1, p2, q2, c1, c2, s, t);
  }
  static LL_HD void seg_seg_st(const L& ln, const V3l& p1, const V3l& q1, const V3l& p2, const V3l& q2, V3l& c1, V3l& c2, F& s, F& t) {
    const F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f);
    V3l d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    F a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
    // nearly parallel axes: regularised toward the middle of the overlap (LLM_SEG_PARALLEL_REG; same statement as the oracle's seg_seg)
    F ia = one / a;
    F sa = lm::min_(lm::max_((zero - c) * ia, zero), one), sb = lm::min_(lm::max_((b - c) * ia, zero), one);
    F reg = a * e * (float)LLM_SEG_PARALLEL_REG;
    F den = a * e - b * b;
    F s0 = lm::min_(lm::max_((b * f - c * e + reg * (0.5f * (sa + sb))) / (den + reg), zero), one);
    F t0 = (b * s0 + f) / e;
    B lt = t0 < 0.0f, gt = t0 > 1.0f;
    t = lm::sel(lt, zero, lm::sel(gt, one, t0));
    F s_alt = lm::sel(lt, sa, sb);                         // clamp(-c / a) resp. clamp((b - c) / a): the projections computed above
    s = lm::sel(lm::or_(lt, gt), s_alt, s0);
    c1 = p1 + scale(d1, s);
    c2 = p2 + scale(d2, t);
  }
  // self-collision row: frictionless, between a point on leg `own` and the same point on leg `other`.  Its base part
  // gt[6] is env-uniform, its joint part jt[3] per leg; packed against the scattered velocity registers: a 16-lane sum of
  // sa * VA + sb * VB + sj * VJ is gt . dx + sum over legs jt . dq  (VA repeats dx[0..3] in four legs, VB dx[4..5] eight times)
  struct SelfRow {
    F sa, sb, sj;      // gt[sub] / 4,  gt[4 + (sub & 1)] / 8,  jt[sub] (sub 3: 0)
    float c, inv, lam;
  };
  static LL_HD void self_row_clear(const L& ln, SelfRow& rw) {
    rw.sa = rw.sb = rw.sj = ln.lane_f(0.0f);
    rw.c = rw.inv = rw.lam = 0.0f;
  }
  static LL_HD void self_row_pack(const L& ln, const F* jt, const float* gt, SelfRow& rw) {
    B s0 = ln.is_sub(0), s1 = ln.is_sub(1), s2 = ln.is_sub(2), odd = lm::odd_(ln.sub());
    rw.sa = lm::sel(s0, ln.lane_f(0.25f * gt[0]), lm::sel(s1, ln.lane_f(0.25f * gt[1]), lm::sel(s2, ln.lane_f(0.25f * gt[2]), ln.lane_f(0.25f * gt[3]))));
    rw.sb = lm::sel(odd, ln.lane_f(0.125f * gt[5]), ln.lane_f(0.125f * gt[4]));
    rw.sj = lm::sel(s0, jt[0], lm::sel(s1, jt[1], lm::sel(s2, jt[2], ln.lane_f(0.0f))));
  }
  static LL_HD float self_row_velocity(const SelfRow& rw, const F& VA, const F& VB, const F& VJ) {
    return L::qsum(L::subsum(rw.sa * VA + rw.sb * VB + rw.sj * VJ));
  }
  static LL_HD void self_row_apply(const L& ln, const SelfRow& rw, float d, F& VA, F& VB, F& VJ) {
    VA = VA + rw.sa * ln.lane_f(4.0f * d);
    VB = VB + rw.sb * ln.lane_f(8.0f * d);
    VJ = VJ + rw.sj * ln.lane_f(d);
  }
  static LL_HD void self_turn(const L& ln, SelfRow& rw, F& VA, F& VB, F& VJ) {
    const float w = rw.c + self_row_velocity(rw, VA, VB, VJ);
    const float nl = fmaxf(rw.lam - w * rw.inv, 0.0f);
    const float d = nl - rw.lam;
    rw.lam = nl;
    self_row_apply(ln, rw, d, VA, VB, VJ);
  }

  // a tangential row of a leg-leg contact (LLM_SPEC_SELF_FRICTION, round 6): box bounds +- hi = mu x the contact's normal multiplier, as the oracle's sweep takes them
  static LL_HD void self_fric_turn(const L& ln, SelfRow& rw, float hi, F& VA, F& VB, F& VJ) {
    const float w = rw.c + self_row_velocity(rw, VA, VB, VJ);
    const float nl = fminf(fmaxf(rw.lam - w * rw.inv, -hi), hi);
    const float d = nl - rw.lam;
    rw.lam = nl;
    self_row_apply(ln, rw, d, VA, VB, VJ);
  }
  // btPlaneSpace1(n): the two tangents Bullet pairs with a contact normal (world coordinates; the oracle's dirs[1], dirs[2])
  static LL_HD void plane_space(const V3<float>& n, V3<float>& p, V3<float>& q) {
    if (fabsf(n.z) > 0.7071067811865475f) {
      const float a = n.y * n.y + n.z * n.z, kk = 1.0f / sqrtf(a);
      p = mk3<float>(0.0f, -n.z * kk, n.y * kk);
      q = mk3<float>(a * kk, -n.x * p.z, n.x * p.y);
    } else {
      const float a = n.x * n.x + n.y * n.y, kk = 1.0f / sqrtf(a);
      p = mk3<float>(-n.y * kk, n.x * kk, 0.0f);
      q = mk3<float>(-n.z * p.y, n.z * p.x, a * kk);
    }
  }
  // the bias of a leg-leg / robot-robot NORMAL row at depth d (a separated point within the margin may close its gap; a penetrating one is pushed out at the ERP of its depth)
  static LL_HD float row_bias(const StepParams& P, float d, float inv_dt) {
    return (d > 0.0f) ? d * inv_dt : fmaxf(d * ((d > P.erp_deep_below ? P.erp : P.erp_deep) * inv_dt), -P.max_depen);
  }
  // one row of a leg-leg contact along direction ub (F0): the two points move with the base alike, so the row has joint parts only (sgn: + own leg, - other leg, 0 elsewhere;
  // e1 .. e3: the lever arms of the lane's joints about the contact point)
  static LL_HD void self_row_build(const L& ln, SelfRow& rw, const V3l& ub, const F& sgn, const V3l& e1, const V3l& e2, const V3l& e3, const LegFactor& lf,
                                   const float* Sb, const float* Sd, const F* qs, float bias, bool have) {
    const F zero = ln.lane_f(0.0f);
    F sjt[3];
    float sgt[6];
    sjt[0] = sgn * dot(ub, e1); sjt[1] = sgn * dot(ub, e2); sjt[2] = sgn * dot(ub, e3);
    float vrow = L::qsum(sjt[0] * qs[0] + sjt[1] * qs[1] + sjt[2] * qs[2]);      // the base moves both points alike: no base part
    lm_fwd(lf, sjt);
    SV<F> yj = scale(lf.y1, sjt[0]) + scale(lf.y2, sjt[1]) + scale(lf.y3, sjt[2]);
    F g6[6] = {zero - yj.a.x, zero - yj.a.y, zero - yj.a.z, zero - yj.l.x, zero - yj.l.y, zero - yj.l.z};
    L::qsum6(g6, sgt);
    fwd6(Sb, Sd, sgt);
    float nn = L::qsum(sjt[0] * sjt[0] + sjt[1] * sjt[1] + sjt[2] * sjt[2]);
    for (int i = 0; i < 6; i++) nn += sgt[i] * sgt[i];
    self_row_pack(ln, sjt, sgt, rw);
    rw.c = vrow + bias;
    rw.inv = have ? 1.0f / nn : 0.0f;
    rw.lam = 0.0f;
    if (!have) self_row_clear(ln, rw);
  }

  // one shared row of a robot-robot contact along direction ub_ (my F0; the world direction points from robot 1 to robot 0 and carries my sign) at point pb_: my half of it -- base
  // part [pb_ x ub_; ub_] and the joints of the leg that holds my capsule (`mine`) -- whitened; free velocity and diagonal are summed with the other robot's half in a fixed
  // order (robot 0's first), so both rows of the arena hold the same c and inv
  static LL_HD void pair_row_build(const L& ln, SelfRow& rw, const V3<float>& ub_, const V3<float>& pb_, const B& mine, const V3l& e1, const V3l& e2, const V3l& e3,
                                   const LegFactor& lf, const float* Sb, const float* Sd, const float* xi, const F* qs, const StepParams& P, float dsel, float inv_dt, bool normal, bool have, int me) {
    const F zero = ln.lane_f(0.0f);
    const V3l ub = cvt3<F>(ub_);
    F sjt[3];
    float sgt[6];
    sjt[0] = lm::sel(mine, dot(ub, e1), zero); sjt[1] = lm::sel(mine, dot(ub, e2), zero); sjt[2] = lm::sel(mine, dot(ub, e3), zero);
    const V3<float> pxu = cross(pb_, ub_);
    float vrow = L::qsum(sjt[0] * qs[0] + sjt[1] * qs[1] + sjt[2] * qs[2]) + pxu.x * xi[0] + pxu.y * xi[1] + pxu.z * xi[2] + ub_.x * xi[3] + ub_.y * xi[4] + ub_.z * xi[5];
    lm_fwd(lf, sjt);
    SV<F> yj = scale(lf.y1, sjt[0]) + scale(lf.y2, sjt[1]) + scale(lf.y3, sjt[2]);
    F g6[6] = {zero - yj.a.x, zero - yj.a.y, zero - yj.a.z, zero - yj.l.x, zero - yj.l.y, zero - yj.l.z};
    L::qsum6(g6, sgt);
    sgt[0] += pxu.x; sgt[1] += pxu.y; sgt[2] += pxu.z; sgt[3] += ub_.x; sgt[4] += ub_.y; sgt[5] += ub_.z;
    fwd6(Sb, Sd, sgt);
    float nn = L::qsum(sjt[0] * sjt[0] + sjt[1] * sjt[1] + sjt[2] * sjt[2]);
    for (int i = 0; i < 6; i++) nn += sgt[i] * sgt[i];
    self_row_pack(ln, sjt, sgt, rw);
    const float vo = ln.peer_u(vrow), no = ln.peer_u(nn);
    rw.c = ((me == 0) ? vrow + vo : vo + vrow) + (normal ? row_bias(P, dsel, inv_dt) : 0.0f);      // (the bias is formed HERE: handed in as a finished value it cost the larger-batch chase-tag build 26 spilt registers and 12 % of its time, profiles/r06_row_bias_ab.txt)
    rw.inv = have ? 1.0f / ((me == 0) ? nn + no : no + nn) : 0.0f;
    rw.lam = 0.0f;
    if (!have) self_row_clear(ln, rw);
  }
  // a tangential shared row (LLM_SPEC_PAIR_FRICTION): pair_turn with the box bounds +- hi = mu x the contact's normal multiplier
  static LL_HD void pair_fric_turn(const L& ln, SelfRow& rw, float hi, F& VA, F& VB, F& VJ, int me) {
    const float mine = self_row_velocity(rw, VA, VB, VJ);
    const float theirs = ln.peer_u(mine);
    const float w = rw.c + ((me == 0) ? mine + theirs : theirs + mine);
    const float nl = fminf(fmaxf(rw.lam - w * rw.inv, -hi), hi);
    const float d = nl - rw.lam;
    rw.lam = nl;
    self_row_apply(ln, rw, d, VA, VB, VJ);
  }

  // robot-robot row (SEPMC): the same frictionless turn, with the other robot's share of the row velocity fetched from its row; both
  // rows add the two shares in the same order (robot 0's first), so both apply the same multiplier
  static LL_HD void pair_turn(const L& ln, SelfRow& rw, F& VA, F& VB, F& VJ, int me) {
    const float mine = self_row_velocity(rw, VA, VB, VJ);
    const float theirs = ln.peer_u(mine);
    const float w = rw.c + ((me == 0) ? mine + theirs : theirs + mine);
    const float nl = fmaxf(rw.lam - w * rw.inv, 0.0f);
    const float d = nl - rw.lam;
    rw.lam = nl;
    self_row_apply(ln, rw, d, VA, VB, VJ);
  }

  // Where a candidate is tested against the terrain: its lowest point (the point the plane test uses), in world coordinates;
  // for a sphere the test is made at the centre with the radius subtracted (rs), so that a side wall is met sideways.
  static LL_HD void cand_eval_point(const L& ln, const Base& bs, const M3<float>& R, const M3<F>& lR, const V3l& lp, const V3l& ez_link, const V3l& A,
                                    const V3l& ax, const F& r, const F& az, const F& len, V3l& Ew, F& rs) {
    const F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f);
    F il = lm::sel(len < 1e-6f, zero, one / lm::max_(len, ln.lane_f(1e-12f)));
    V3l dir = mk3<F>((ez_link.x - az * ax.x) * il, (ez_link.y - az * ax.y) * il, (ez_link.z - az * ax.z) * il);
    V3l x = A - scale(dir, r);
    V3l Pb = lp + mul(lR, x);
    V3l Pw = mul(R, Pb);
    rs = lm::sel(dot(ax, ax) < 0.5f, r, zero);
    Ew = mk3<F>(Pw.x + bs.p.x, Pw.y + bs.p.y, Pw.z + bs.p.z + rs);
  }
  // one row of a contact at point Pb (F0) along direction uu: joint part through the lever arms d1..d3, base part [Pb x uu; uu],
  // both whitened; c = its free velocity (+ bias for the normal row)
  static LL_HD void contact_row(const L& ln, Row& rw, const V3l& uu, const V3l& Pb, const V3l& d1, const V3l& d2, const V3l& d3, const LegFactor& lf,
                                const float* Sb, const float* Sd, const float* xi, const F* qs, const F& bias, const B& cvalid) {
    F gt[6], jt[3];
    contact_row(ln, rw, uu, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, bias, cvalid, gt, jt);
  }
  // ... leaving the whitened coefficients gt[6], jt[3] with the caller (the cone-coupled friction solve forms cross Gram scalars from them)
  static LL_HD void contact_row(const L& ln, Row& rw, const V3l& uu, const V3l& Pb, const V3l& d1, const V3l& d2, const V3l& d3, const LegFactor& lf,
                                const float* Sb, const float* Sd, const float* xi, const F* qs, const F& bias, const B& cvalid, F* gt, F* jt) {
    jt[0] = dot(uu, d1); jt[1] = dot(uu, d2); jt[2] = dot(uu, d3);
    V3l pxu = cross(Pb, uu);
    // free row velocity J_b xi + J_l qd*
    F vrow = pxu.x * xi[0] + pxu.y * xi[1] + pxu.z * xi[2] + uu.x * xi[3] + uu.y * xi[4] + uu.z * xi[5] + jt[0] * qs[0] + jt[1] * qs[1] + jt[2] * qs[2];
    lm_fwd(lf, jt);
    SV<F> yj = scale(lf.y1, jt[0]) + scale(lf.y2, jt[1]) + scale(lf.y3, jt[2]);
    gt[0] = pxu.x - yj.a.x; gt[1] = pxu.y - yj.a.y; gt[2] = pxu.z - yj.a.z;
    gt[3] = uu.x - yj.l.x; gt[4] = uu.y - yj.l.y; gt[5] = uu.z - yj.l.z;
    fwd6(Sb, Sd, gt);
    F nn = jt[0] * jt[0] + jt[1] * jt[1] + jt[2] * jt[2];
    for (int i = 0; i < 6; i++) nn = nn + gt[i] * gt[i];
    rw.c = vrow + bias;
    rw.inv = lm::sel(cvalid, ln.lane_f(1.0f) / nn, ln.lane_f(0.0f));
    finish_row<false>(ln, rw, gt, jt);
  }
  // ---- the three rows of a contact with their five Gram blocks on the matrix cores, in the background (lane policies with kGramPipe: lanes.hpp WithGramPipe) ----
  // Same rows, same scalars as contact_row x 3 + cross_gram x 2; what changes is WHO forms the base part of the Gram scalars and WHEN: the six MFMAs of a block
  // are issued one at a time between pieces of the NEXT row's coefficient arithmetic (row_coeffs' tick points), the blocks of the last row and the two cross blocks
  // between the pieces of the rows' own finishing work, and every block is collected (16 register swaps) long after the matrix cores have finished it.
  struct RowCo { F gt[6], jt[3]; };                     // a row's whitened coefficients
  struct NoTick { template <int K_> LL_HD void at() const {} };
  struct GramTick {                                      // one block in the background: acc += outer(x[K], y[K]) at tick point K
    typename L::GramAcc& a;
    const F* x;
    const F* y;
    template <int K_> LL_HD void at() const { L::template gram_mfma<K_>(a, x[K_], y[K_]); }
  };
  // contact_row up to the whitened coefficients gt[6], jt[3], c and inv; tk.at<0..5>() are spread over its ~110 instructions
  template <class TK>
  static LL_HD void row_coeffs(const L& ln, Row& rw, RowCo& co, const V3l& uu, const V3l& Pb, const V3l& d1, const V3l& d2, const V3l& d3, const LegFactor& lf,
                               const float* Sb, const float* Sd, const float* xi, const F* qs, const F& bias, const B& cvalid, const TK& tk) {
    F* gt = co.gt;
    F* jt = co.jt;
    jt[0] = dot(uu, d1); jt[1] = dot(uu, d2); jt[2] = dot(uu, d3);
    V3l pxu = cross(Pb, uu);
    tk.template at<0>();
    F vrow = pxu.x * xi[0] + pxu.y * xi[1] + pxu.z * xi[2] + uu.x * xi[3] + uu.y * xi[4] + uu.z * xi[5] + jt[0] * qs[0] + jt[1] * qs[1] + jt[2] * qs[2];
    lm_fwd(lf, jt);
    tk.template at<1>();
    SV<F> yj = scale(lf.y1, jt[0]) + scale(lf.y2, jt[1]) + scale(lf.y3, jt[2]);
    tk.template at<2>();
    gt[0] = pxu.x - yj.a.x; gt[1] = pxu.y - yj.a.y; gt[2] = pxu.z - yj.a.z;
    gt[3] = uu.x - yj.l.x; gt[4] = uu.y - yj.l.y; gt[5] = uu.z - yj.l.z;
    for (int i = 0; i < 4; i++) {                        // fwd6, rows 0 .. 3
      F sacc = gt[i];
      for (int k = 0; k < i; k++) sacc = sacc - gt[k] * Sb[i * (i + 1) / 2 + k];
      gt[i] = sacc * Sd[i];
    }
    tk.template at<3>();
    for (int i = 4; i < 6; i++) {                        // fwd6, rows 4, 5
      F sacc = gt[i];
      for (int k = 0; k < i; k++) sacc = sacc - gt[k] * Sb[i * (i + 1) / 2 + k];
      gt[i] = sacc * Sd[i];
    }
    tk.template at<4>();
    F nn = jt[0] * jt[0] + jt[1] * jt[1] + jt[2] * jt[2];
    for (int i = 0; i < 6; i++) nn = nn + gt[i] * gt[i];
    rw.c = vrow + bias;
    rw.inv = lm::sel(cvalid, ln.lane_f(1.0f) / nn, ln.lane_f(0.0f));
    tk.template at<5>();
  }
  static LL_HD void neg_scaled(const L& ln, const F& inv, const RowCo& co, F* yg, F* yj) {      // finish_row's yg, yj: the coefficients times -inv
    F ninv = ln.lane_f(0.0f) - inv;
    for (int i = 0; i < 6; i++) yg[i] = co.gt[i] * ninv;
    for (int i = 0; i < 3; i++) yj[i] = co.jt[i] * ninv;
  }
  static LL_HD void joint_part(const F* yj, const F* jt_oth, F* nj) {                             // the rows of one leg also meet in its joints (finish_row's nj)
    nj[0] = yj[0] * L::template subbcast<0>(jt_oth[0]) + yj[1] * L::template subbcast<0>(jt_oth[1]) + yj[2] * L::template subbcast<0>(jt_oth[2]);
    nj[1] = yj[0] * L::template subbcast<1>(jt_oth[0]) + yj[1] * L::template subbcast<1>(jt_oth[1]) + yj[2] * L::template subbcast<1>(jt_oth[2]);
    nj[2] = yj[0] * L::template subbcast<2>(jt_oth[0]) + yj[1] * L::template subbcast<2>(jt_oth[1]) + yj[2] * L::template subbcast<2>(jt_oth[2]);
    nj[3] = yj[0] * L::template subbcast<3>(jt_oth[0]) + yj[1] * L::template subbcast<3>(jt_oth[1]) + yj[2] * L::template subbcast<3>(jt_oth[2]);
  }
  // out[L] = (base part, collected from the matrix cores) + [leg of L == my leg] nj[L & 3]
  static LL_HD void gram_finish(const typename L::GramAcc& a, const F* lf4, const F* nj, F* out) {
    F d[16];
    L::gram_collect(a, d);
    for (int L_ = 0; L_ < 16; L_++) out[L_] = d[L_] + lf4[L_ >> 2] * nj[L_ & 3];
  }
  static LL_HD void row_perm_a(const L& ln, Row& r, const RowCo& co) {                             // finish_row's lane-permuted coefficients, in two pieces
    F t[4];
    permute4(ln, co.gt[0], co.gt[1], co.gt[2], co.gt[3], t);
    r.ca01 = L::pair(t[0], t[1]); r.ca23 = L::pair(t[2], t[3]);
    r.lam = ln.lane_f(0.0f);
  }
  static LL_HD void row_perm_b(const L& ln, Row& r, const RowCo& co) {
    F t[4];
    B odd = lm::odd_(ln.sub());
    r.cb01 = L::pair(lm::sel(odd, co.gt[5], co.gt[4]), lm::sel(odd, co.gt[4], co.gt[5]));
    permute4(ln, co.jt[0], co.jt[1], co.jt[2], ln.lane_f(0.0f), t);
    r.cj01 = L::pair(t[0], t[1]); r.cj23 = L::pair(t[2], t[3]);
  }
  static LL_HD void contact_rows_cone_piped(const L& ln, Row& rn, Row& r1, Row& r2, ConeX& cx, const V3l& un, const V3l& ut1, const V3l& ut2, const V3l& Pb,
                                            const V3l& d1, const V3l& d2, const V3l& d3, const LegFactor& lf, const float* Sb, const float* Sd, const float* xi,
                                            const F* qs, const F& bias, const B& cvalid) {
    const F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f);
    RowCo cn, c1, c2;
    F ygn[6], yjn[3], yg1[6], yj1[3], yg2[6], yj2[3];
    typename L::GramAcc an, a1, a2, a12, a21;
    row_coeffs(ln, rn, cn, un, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, bias, cvalid, NoTick());
    neg_scaled(ln, rn.inv, cn, ygn, yjn);
    row_coeffs(ln, r1, c1, ut1, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, zero, cvalid, GramTick{an, cn.gt, ygn});      // block (n, n) behind the t1 row's arithmetic
    neg_scaled(ln, r1.inv, c1, yg1, yj1);
    row_coeffs(ln, r2, c2, ut2, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, zero, cvalid, GramTick{a1, c1.gt, yg1});      // block (t1, t1) behind the t2 row's
    neg_scaled(ln, r2.inv, c2, yg2, yj2);
    // the last three blocks -- (t2, t2), n12 = (t1 against the t2 rows), n21 = (t2 against the t1 rows) -- behind the rows' finishing work, round robin
    const F lf4[4] = {lm::sel(ln.is_leg(0), one, zero), lm::sel(ln.is_leg(1), one, zero), lm::sel(ln.is_leg(2), one, zero), lm::sel(ln.is_leg(3), one, zero)};
    F njn[4], nj1[4], nj2[4], nj12[4], nj21[4];
#define LL_TK3(K_) do { L::template gram_mfma<K_>(a2, c2.gt[K_], yg2[K_]); } while (0)
#define LL_TK12(K_) do { L::template gram_mfma<K_>(a12, c2.gt[K_], yg1[K_]); } while (0)
#define LL_TK21(K_) do { L::template gram_mfma<K_>(a21, c1.gt[K_], yg2[K_]); } while (0)
    LL_TK3(0);
    joint_part(yjn, cn.jt, njn);
    LL_TK12(0);
    gram_finish(an, lf4, njn, rn.nk);                                   // (n, n): finished two rows ago
    LL_TK21(0);
    row_perm_a(ln, rn, cn);
    LL_TK3(1);
    row_perm_b(ln, rn, cn);
    LL_TK12(1);
    joint_part(yj1, c1.jt, nj1);
    LL_TK21(1);
    joint_part(yj2, c2.jt, nj2);
    LL_TK3(2);
    row_perm_a(ln, r1, c1);
    LL_TK12(2);
    row_perm_b(ln, r1, c1);
    LL_TK21(2);
    joint_part(yj1, c2.jt, nj12);
    LL_TK3(3);
    joint_part(yj2, c1.jt, nj21);
    LL_TK12(3);
    row_perm_a(ln, r2, c2);
    LL_TK21(3);
    row_perm_b(ln, r2, c2);
    LL_TK3(4);
    LL_TK12(4);
    LL_TK21(4);
    gram_finish(a1, lf4, nj1, r1.nk);                                   // (t1, t1): finished a row ago
    LL_TK3(5);
    LL_TK12(5);
    LL_TK21(5);
#undef LL_TK3
#undef LL_TK12
#undef LL_TK21
    gram_finish(a2, lf4, nj2, r2.nk);
    gram_finish(a12, lf4, nj12, cx.n12);
    gram_finish(a21, lf4, nj21, cx.n21);
  }

  // LR:137-141: tau = kp (target - q) + kd (0 - qd), clipped to +-max_tau -- the `forces=` the reference hands to
  // setJointMotorControlArray(TORQUE_CONTROL) before every stepSimulation (golden G8; ll_probe_pd_torque runs exactly this)
  static LL_HD void pd_torque(const L& ln, const StepParams& P, const F* q, const F* qd, const F* tgt, F* tau, float max_tau = 0.0f) {
    const float mt = max_tau > 0.0f ? max_tau : P.max_tau;
    for (int j = 0; j < 3; j++) {
      F t = (tgt[j] - q[j]) * P.kp + (ln.lane_f(0.0f) - qd[j]) * P.kd;
      tau[j] = lm::min_(lm::max_(t, ln.lane_f(-mt)), ln.lane_f(mt));
    }
  }
  // the target of a control step: joint angles at its start + the policy's action (PLE:199-200), clipped to +-3 rad (LR:126-127)
  static LL_HD void pd_target(const L& ln, const F* q, const F* act, F* tgt) {
    for (int j = 0; j < 3; j++) tgt[j] = lm::min_(lm::max_(q[j] + act[j], ln.lane_f(-3.0f)), ln.lane_f(3.0f));
  }
  // parity probe (golden G8): rows of [q 12 | qd 12 | x 12] -> tau 12; x is a target (mode 0: LeggedRobot.apply_action's
  // argument, clipped here as LR:126-127 does) or an action (mode 1: PLE:199-200 forms the target from it)
  static LL_HD void probe_pd(const L& ln, const StepParams& P, const float* in, float* out, int mode) {
    F q[3], qd[3], x[3], tgt[3], tau[3];
    for (int j = 0; j < 3; j++) { q[j] = ln.ldl(in, j, 3); qd[j] = ln.ldl(in, 12 + j, 3); x[j] = ln.ldl(in, 24 + j, 3); }
    if (mode == 1) pd_target(ln, q, x, tgt);
    else for (int j = 0; j < 3; j++) tgt[j] = lm::min_(lm::max_(x[j], ln.lane_f(-3.0f)), ln.lane_f(3.0f));
    pd_torque(ln, P, q, qd, tgt, tau);
    for (int j = 0; j < 3; j++) ln.stl(out, j, 3, tau[j]);
  }
  // btMultiBody::m_maxCoordinateVelocity: finite velocities are clipped to +-vmax with one v_med3; a NaN or an infinity must NOT become a bound (v_med3 /
  // fmin / fmax would make it one) -- the non-finite guard of the step tail has to see it -- so x * 0 (0 for a finite x, NaN otherwise) rides along: 2 instructions a value
  static LL_HD float clip1(float x, float vmax) { return __builtin_fmaf(x, 0.0f, lm::med3_(x, -vmax, vmax)); }
  static LL_HD void clip_velocities(const L& ln, float* xi, F* qs, float vmax) {
    for (int i = 0; i < 6; i++) xi[i] = clip1(xi[i], vmax);
    F hi = ln.lane_f(vmax), lo = ln.lane_f(-vmax), zero = ln.lane_f(0.0f);
    for (int j = 0; j < 3; j++) qs[j] = lm::med3_(qs[j], lo, hi) + qs[j] * zero;
  }
  static LL_HD void substep(const L& ln, const StepParams& P, Base& bs, F* q, F* qd, const F* tgt, int env = 0, int sidx = -1,
                            const SubstepExtra* ex = nullptr) {
    if (P.friction_mode == 2 && pmc_wants_xrows(P)) substep_impl<false, false, true, true>(ln, P, bs, q, qd, tgt, env, sidx, ex, nullptr);
    else if (P.friction_mode == 2) substep_impl<false, false, true>(ln, P, bs, q, qd, tgt, env, sidx, ex, nullptr);      // (host tests: emu_substep)
    else substep_impl<false>(ln, P, bs, q, qd, tgt, env, sidx, ex, nullptr);
  }
  // TERRAIN: contact candidates are also tested against ex->shapes, and a contact's normal is that of the shape it touches
  // PAIR: contacts with the other robot of a SEPMC arena (the neighbouring row) are found and solved too
  // XROWS (round 6): the build that can carry the extended contact rows -- the two tangential rows of a leg-leg contact (LLM_SPEC_SELF_FRICTION) and, with PAIR, of a robot-robot
  // contact (LLM_SPEC_PAIR_FRICTION), and up to four robot-robot contacts per pair (LLM_SPEC_MAX_PAIR).  Engine twins of what had been oracle-only switches.  A build of its own because
  // the rows cost the plain build registers even when they are switched off (the contract kernel ran 2.5 % slower with the code merely present: profiles/r06_self_friction_ab.txt);
  // the launch picks it when one of the three switches is off its default (pmc_wants_xrows).
  template <bool TERRAIN, bool PAIR = false, bool CONE = false, bool XROWS = false>
  static LL_HD void substep_impl(const L& ln, const StepParams& P_in, Base& bs, F* q, F* qd, const F* tgt, int env, int sidx, const SubstepExtra* ex,
                                 const LinkC* held) {   // held: the own-link constants if the caller keeps them in registers, or null
#define PMC_TSS(k) do { if (sidx == 5) PMC_TS(k); } while (0)
    PMC_PHASE("sub.kinematics");
    const StepParams& P = (L::kParamsReload > 1) ? ln.params(P_in) : P_in;
    const float* legc = P.legc;
    const float* bc = P.basec;
    const float dt = P.dt;
    ln.refresh_consts();
    M3<float> R = qmat(bs.q);
    SV<float> v0;
    v0.a = mulT(R, bs.w);
    v0.l = mulT(R, bs.v);
    V3u ezb = mk3<float>(R.m[6], R.m[7], R.m[8]);

    // --- PD torque with clip (LR:126-141) + URDF joint damping --------------------------------------------
    F tau[3];
    pd_torque(ln, P, q, qd, tgt, tau, (PAIR && ex && ex->pair_me == 1) ? P.max_tau1 : 0.0f);      // (SEPMC: robot 1 may have its own limit, LR:244)
    for (int j = 0; j < 3; j++) tau[j] = tau[j] - ln.legc(legc, LC_JDAMP + j) * qd[j];

    // --- leg kinematics, velocities -------------------------------------------------------------------------
    LegKin k = leg_fk(ln, legc, q[0], q[1], q[2]);
    F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f);
    SV<F> S1, S2, S3v;
    S1.a = mk3<F>(one, zero, zero); S1.l = k.s1;
    S2.a = k.a2; S2.l = k.s2;
    S3v.a = k.a2; S3v.l = k.s3;
    // Link-level work (inertia about the F0 origin, bias force, F = I^c S) is done ONCE per link: sub-lane k < 3 works on link k + 1
    // (sub-lane 3 on a link of zero mass), composite quantities are suffix sums over the sub-lanes, and what every sub-lane needs
    // afterwards (joint-space inertia, F1..F3, the leg's total bias force and inertia) is handed round with DPP moves.
    const B sub_is0 = ln.is_sub(0), sub_is1 = ln.is_sub(1);
    const F qd1m = lm::sel(sub_is0, zero, qd[1]), qd2m = lm::sel(lm::or_(sub_is0, sub_is1), zero, qd[2]);
    SV<F> vb = cvt6<F>(v0);
    SV<F> v1 = vb + scale(S1, qd[0]);
    SV<F> v2 = v1 + scale(S2, qd[1]);
    SV<F> vk = v1 + scale(S2, qd1m) + scale(S3v, qd2m);                      // velocity of the own link
    // velocity-product acceleration of the own link (frame falling with gravity, base acceleration zero)
    SV<F> ak = scale(crm(vb, S1), qd[0]) + scale(crm(v1, S2), qd1m) + scale(crm(v2, S3v), qd2m);
    M3<F> Rk;
    for (int i = 0; i < 9; i++) Rk.m[i] = lm::sel(sub_is0, k.R1.m[i], lm::sel(sub_is1, k.R2.m[i], k.R3.m[i]));
    V3l pk = mk3<F>(lm::sel(sub_is0, k.p1.x, lm::sel(sub_is1, k.p2.x, k.p3.x)), lm::sel(sub_is0, k.p1.y, lm::sel(sub_is1, k.p2.y, k.p3.y)),
                    lm::sel(sub_is0, k.p1.z, lm::sel(sub_is1, k.p2.z, k.p3.z)));
    SV<F> Sk;                                                                 // motion axis of the own joint
    Sk.a = mk3<F>(lm::sel(sub_is0, one, zero), lm::sel(sub_is0, zero, k.a2.y), lm::sel(sub_is0, zero, k.a2.z));
    Sk.l = mk3<F>(lm::sel(sub_is0, k.s1.x, lm::sel(sub_is1, k.s2.x, k.s3.x)), lm::sel(sub_is0, k.s1.y, lm::sel(sub_is1, k.s2.y, k.s3.y)),
                  lm::sel(sub_is0, k.s1.z, lm::sel(sub_is1, k.s2.z, k.s3.z)));

    // --- own link: inertia, bias force ------------------------------------------------------------------------
    V3l ck;
    S3<F> ick;
    const LinkC lk = held ? *held : own_link(ln);
    RI<F> Ik = own_link_inertia(lk, Rk, pk, &ck, &ick);
    SV<F> fk = apply(Ik, ak) + crf(vk, apply(Ik, vk)) + scale(damping_force<F>(vk, ck, ick, Ik.m, P.link_damping), ln.lane_f(-1.0f));
    if (ex && ex->has_push) {                                                 // on the FR hip link: leg 0, sub-lane 0
      V3l fb = scale(mul(Rk, mk3<F>(ln.lane_f(ex->push[0]), ln.lane_f(ex->push[1]), ln.lane_f(ex->push[2]))), lm::sel(lm::and_(ln.is_leg(0), sub_is0), one, zero));
      SV<F> fe;
      fe.a = cross(ck, fb); fe.l = fb;
      fk = fk + scale(fe, ln.lane_f(-1.0f));
    }
    // force the joint of the own link carries = own + outboard links; the joint's share of it against the torque
    F fs[6];
    sv_to6(fk, fs);
    L::sufsum6(fs);
    SV<F> fsk;
    fsk.a = mk3<F>(fs[0], fs[1], fs[2]); fsk.l = mk3<F>(fs[3], fs[4], fs[5]);
    F tauk = lm::sel(sub_is0, tau[0], lm::sel(sub_is1, tau[1], tau[2]));
    F b[3];
    L::spread3(tauk - dot(Sk, fsk), b);
    F f123a[6];
    L::template subbcast6<0>(fs, f123a);                                      // the leg's total, for the base
    SV<F> f123;
    f123.a = mk3<F>(f123a[0], f123a[1], f123a[2]); f123.l = mk3<F>(f123a[3], f123a[4], f123a[5]);

    // --- composite inertias (suffix sums), joint-space inertia of the leg, coupling to the base -----------------------
    F ci[12] = {Ik.m, Ik.h.x, Ik.h.y, Ik.h.z, Ik.io.xx, Ik.io.xy, Ik.io.xz, Ik.io.yy, Ik.io.yz, Ik.io.zz, zero, zero};
    L::sufsum6(ci); L::sufsum4(ci + 6);
    RI<F> Ick;                                                                // I^c of the own joint: own link + outboard links
    Ick.m = ci[0]; Ick.h = mk3<F>(ci[1], ci[2], ci[3]);
    Ick.io.xx = ci[4]; Ick.io.xy = ci[5]; Ick.io.xz = ci[6]; Ick.io.yy = ci[7]; Ick.io.yz = ci[8]; Ick.io.zz = ci[9];
    F cin[12];                                                                // I^c of the whole leg (sub-lane 0's), everywhere
    L::template subbcast6<0>(ci, cin); L::template subbcast6<0>(ci + 6, cin + 6);
    SV<F> Fk = apply(Ick, Sk);
    F Fk6[6], F16[6], F26[6], F36[6];
    sv_to6(Fk, Fk6);
    L::template subbcast6<0>(Fk6, F16); L::template subbcast6_after<1>(Fk6, F26, F16[0]); L::template subbcast6_after<2>(Fk6, F36, F26[0]);
    SV<F> F1, F2, F3;
    F1.a = mk3<F>(F16[0], F16[1], F16[2]); F1.l = mk3<F>(F16[3], F16[4], F16[5]);
    F2.a = mk3<F>(F26[0], F26[1], F26[2]); F2.l = mk3<F>(F26[3], F26[4], F26[5]);
    F3.a = mk3<F>(F36[0], F36[1], F36[2]); F3.l = mk3<F>(F36[3], F36[4], F36[5]);
    F dcol[3] = {dot(S1, Fk), dot(S2, Fk), dot(S3v, Fk)}, mt[6];              // column k of the joint-space inertia
    L::gather_tri3(dcol, mt);
    F m11 = mt[0], m12 = mt[1], m22 = mt[2], m13 = mt[3], m23 = mt[4], m33 = mt[5];
    LegFactor lf;
    lf.i11 = lm::rsqrt_(m11); lf.l11 = m11 * lf.i11;
    lf.l21 = m12 * lf.i11; lf.l31 = m13 * lf.i11;
    F d22 = m22 - lf.l21 * lf.l21;
    lf.i22 = lm::rsqrt_(d22); lf.l22 = d22 * lf.i22;
    lf.l32 = (m23 - lf.l31 * lf.l21) * lf.i22;
    F d33 = m33 - lf.l31 * lf.l31 - lf.l32 * lf.l32;
    lf.i33 = lm::rsqrt_(d33); lf.l33 = d33 * lf.i33;
    lf.y1 = scale(F1, lf.i11);
    lf.y2 = scale(F2 + scale(lf.y1, zero - lf.l21), lf.i22);
    lf.y3 = scale(F3 + scale(lf.y1, zero - lf.l31) + scale(lf.y2, zero - lf.l32), lf.i33);

    PMC_TSS(21);
    PMC_PHASE("sub.base_factor");
    // --- base: S = I_base + sum I^c_leg - sum Y Y^T ; packed lower triangle, index order [wx wy wz vx vy vz] ----
    float Sb[21], Sd[6];
    {
      float cs[12];
      L::qsum6(cin, cs); L::qsum6(cin + 6, cs + 6);
      float m = ln.basec(bc, BC_MASS) + cs[0];
      float hx = ln.basec(bc, BC_H + 0) + cs[1], hy = ln.basec(bc, BC_H + 1) + cs[2], hz = ln.basec(bc, BC_H + 2) + cs[3];
      float ixx = ln.basec(bc, BC_IO + 0) + cs[4], ixy = ln.basec(bc, BC_IO + 1) + cs[5], ixz = ln.basec(bc, BC_IO + 2) + cs[6];
      float iyy = ln.basec(bc, BC_IO + 3) + cs[7], iyz = ln.basec(bc, BC_IO + 4) + cs[8], izz = ln.basec(bc, BC_IO + 5) + cs[9];
      // [[Io, hx],[-hx, m]] with hx = skew(h): rows 3..5 x cols 0..2 hold -skew(h) = [[0,hz,-hy],[-hz,0,hx],[hy,-hx,0]]
      Sb[0] = ixx;
      Sb[1] = ixy; Sb[2] = iyy;
      Sb[3] = ixz; Sb[4] = iyz; Sb[5] = izz;
      Sb[6] = 0.0f; Sb[7] = hz; Sb[8] = -hy; Sb[9] = m;
      Sb[10] = -hz; Sb[11] = 0.0f; Sb[12] = hx; Sb[13] = 0.0f; Sb[14] = m;
      Sb[15] = hy; Sb[16] = -hx; Sb[17] = 0.0f; Sb[18] = 0.0f; Sb[19] = 0.0f; Sb[20] = m;
      F y1[6], y2[6], y3[6];
      sv_to6(lf.y1, y1); sv_to6(lf.y2, y2); sv_to6(lf.y3, y3);
      F yy[24];
      float ys[24];
      for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) yy[i * (i + 1) / 2 + j] = y1[i] * y1[j] + y2[i] * y2[j] + y3[i] * y3[j];
      yy[21] = yy[22] = yy[23] = zero;
      L::qsum6(yy, ys); L::qsum6(yy + 6, ys + 6); L::qsum6(yy + 12, ys + 12); L::qsum6(yy + 18, ys + 18);
      for (int i = 0; i < 21; i++) Sb[i] -= ys[i];
      chol6(Sb, Sd);
    }

    PMC_TSS(22);
    PMC_PHASE("sub.free_accel");
    // --- unconstrained accelerations ---------------------------------------------------------------------------
    lm_fwd(lf, b);                                              // bt = Lm^-1 (tau - C_l)
    SV<F> z = scale(lf.y1, b[0]) + scale(lf.y2, b[1]) + scale(lf.y3, b[2]);
    float xb[6];
    {
      RI<float> I0;
      I0.m = ln.basec(bc, BC_MASS);
      I0.h = mk3<float>(ln.basec(bc, BC_H), ln.basec(bc, BC_H + 1), ln.basec(bc, BC_H + 2));
      I0.io.xx = ln.basec(bc, BC_IO); I0.io.xy = ln.basec(bc, BC_IO + 1); I0.io.xz = ln.basec(bc, BC_IO + 2); I0.io.yy = ln.basec(bc, BC_IO + 3); I0.io.yz = ln.basec(bc, BC_IO + 4); I0.io.zz = ln.basec(bc, BC_IO + 5);
      S3<float> ic0;
      ic0.xx = ln.basec(bc, BC_ICOM); ic0.xy = ln.basec(bc, BC_ICOM + 1); ic0.xz = ln.basec(bc, BC_ICOM + 2); ic0.yy = ln.basec(bc, BC_ICOM + 3); ic0.yz = ln.basec(bc, BC_ICOM + 4); ic0.zz = ln.basec(bc, BC_ICOM + 5);
      V3u com0 = mk3<float>(ln.basec(bc, BC_COM), ln.basec(bc, BC_COM + 1), ln.basec(bc, BC_COM + 2));
      SV<float> f0 = crf(v0, apply(I0, v0)) + scale(damping_force<float>(v0, com0, ic0, I0.m, P.link_damping), -1.0f);
      F fz[6] = {f123.a.x + z.a.x, f123.a.y + z.a.y, f123.a.z + z.a.z, f123.l.x + z.l.x, f123.l.y + z.l.y, f123.l.z + z.l.z};
      float fs[6];
      L::qsum6(fz, fs);
      xb[0] = -(f0.a.x + fs[0]); xb[1] = -(f0.a.y + fs[1]); xb[2] = -(f0.a.z + fs[2]);
      xb[3] = -(f0.l.x + fs[3]); xb[4] = -(f0.l.y + fs[4]); xb[5] = -(f0.l.z + fs[5]);
      fwd6(Sb, Sd, xb);
      bwd6(Sb, Sd, xb);
    }
    SV<float> ab;
    ab.a = mk3<float>(xb[0], xb[1], xb[2]); ab.l = mk3<float>(xb[3], xb[4], xb[5]);
    F u[3];
    u[0] = b[0] - dot(lf.y1, ab); u[1] = b[1] - dot(lf.y2, ab); u[2] = b[2] - dot(lf.y3, ab);
    lm_bwd(lf, u);                                              // qdd

    // --- velocities after the unconstrained update (frozen F0 coordinates) -------------------------------------------
    float xi[6];
    {
      V3u wxv = cross(v0.a, v0.l);
      float g = -P.gravity;
      xi[0] = v0.a.x + dt * ab.a.x; xi[1] = v0.a.y + dt * ab.a.y; xi[2] = v0.a.z + dt * ab.a.z;
      xi[3] = v0.l.x + dt * (ab.l.x + wxv.x + g * ezb.x); xi[4] = v0.l.y + dt * (ab.l.y + wxv.y + g * ezb.y); xi[5] = v0.l.z + dt * (ab.l.z + wxv.z + g * ezb.z);
    }
    F qs[3];
    for (int j = 0; j < 3; j++) qs[j] = qd[j] + u[j] * dt;
    // btMultiBody::m_maxCoordinateVelocity (LLM_MAX_COORD_VEL): every generalized velocity clipped when a delta is applied -- here and after the solve.
    // Inert in every gait; it keeps a robot sane that was reset onto a discontinuity of the mocap data (include/llenv_model.h)
    const float vmax = P.max_coord_vel;
    clip_velocities(ln, xi, qs, vmax);

    PMC_TSS(23);
    PMC_PHASE("sub.candidates");
    // --- contact candidates (DESIGN.md "contact candidates"): 28 points per leg, 7 per sub-lane, grouped by link --------------
    //   jj 0..3 -> group A, 4..5 -> group B, 6 -> group C;  links  A: [3,3,2,2]  B: [2,2,2,1]  C: [3,0,0,0]  by sub-lane
    const float inv_dt = 1.0f / dt;
    V3l ez = cvt3<F>(ezb);
    V3l ez1 = mulT(k.R1, ez), ez2 = mulT(k.R2, ez), ez3 = mulT(k.R3, ez);
    F z1 = dot(ez, k.p1) + bs.p.z, z2 = dot(ez, k.p2) + bs.p.z, z3 = dot(ez, k.p3) + bs.p.z, z0b = ln.lane_f(bs.p.z);
    B sub_lt2 = L::i2f(ln.sub()) < 1.5f, sub_lt3 = L::i2f(ln.sub()) < 2.5f, sub_0 = L::i2f(ln.sub()) < 0.5f;
    V3l gez[3];
    F gz0[3];
    gez[0] = mk3<F>(lm::sel(sub_lt2, ez3.x, ez2.x), lm::sel(sub_lt2, ez3.y, ez2.y), lm::sel(sub_lt2, ez3.z, ez2.z));
    gz0[0] = lm::sel(sub_lt2, z3, z2);
    gez[1] = mk3<F>(lm::sel(sub_lt3, ez2.x, ez1.x), lm::sel(sub_lt3, ez2.y, ez1.y), lm::sel(sub_lt3, ez2.z, ez1.z));
    gz0[1] = lm::sel(sub_lt3, z2, z1);
    gez[2] = mk3<F>(lm::sel(sub_0, ez3.x, ez.x), lm::sel(sub_0, ez3.y, ez.y), lm::sel(sub_0, ez3.z, ez.z));
    gz0[2] = lm::sel(sub_0, z3, z0b);
    const F far_ = ln.lane_f(1.0e30f);
    // TERRAIN: link frames of the three candidate groups of this sub-lane (A: [3,3,2,2], B: [2,2,2,1], C: [3,0,0,0]), in F0
    const int n_shapes = (TERRAIN && ex) ? ex->n_shapes : 0;
    const bool terr = TERRAIN && L::any(ln.lane_f(n_shapes > 0 ? 1.0f : 0.0f) > 0.5f);
    M3<F> tgR[3];
    V3l tgp[3];
    if (TERRAIN) {
      if (terr) {
        for (int i = 0; i < 9; i++) {
          tgR[0].m[i] = lm::sel(sub_lt2, k.R3.m[i], k.R2.m[i]);
          tgR[1].m[i] = lm::sel(sub_lt3, k.R2.m[i], k.R1.m[i]);
          tgR[2].m[i] = lm::sel(sub_0, k.R3.m[i], ln.lane_f((i % 4 == 0) ? 1.0f : 0.0f));
        }
        tgp[0] = mk3<F>(lm::sel(sub_lt2, k.p3.x, k.p2.x), lm::sel(sub_lt2, k.p3.y, k.p2.y), lm::sel(sub_lt2, k.p3.z, k.p2.z));
        tgp[1] = mk3<F>(lm::sel(sub_lt3, k.p2.x, k.p1.x), lm::sel(sub_lt3, k.p2.y, k.p1.y), lm::sel(sub_lt3, k.p2.z, k.p1.z));
        tgp[2] = mk3<F>(lm::sel(sub_0, k.p3.x, zero), lm::sel(sub_0, k.p3.y, zero), lm::sel(sub_0, k.p3.z, zero));
      }
    }
    // candidates per sub-lane: the eighth (mid-link sphere) and the ninth (a terrain edge under the trunk, reverse_edge) only with terrain
    constexpr int NT = TERRAIN ? 8 : 7, NC = TERRAIN ? (XROWS ? 10 : 9) : 7;   // (terrain builds: jj = 7 mid-link sphere, 8 a terrain edge under the trunk; their XROWS builds: 9 a terrain edge across a leg box)
    constexpr float STRIDE = TERRAIN ? 16.0f : 8.0f;          // candidate index = STRIDE * sub + jj: the (sub, jj) order of the oracle's enumeration
    F depth[NC];
    const bool want_touch = TERRAIN && ex && ex->want_touch;
    F tch_st = one, tch_fl = one;                             // 0 once a body link touches a static / the flag
    for (int jj = 0; jj < NT; jj++) {
      const int g = jj < 4 ? 0 : (jj < 6 ? 1 : (jj < 7 ? 2 : 0));   // (candidate 7 sits on group A's link)
      V3l A = mk3<F>(ln.candc(jj * CF_WORDS + CF_A), ln.candc(jj * CF_WORDS + CF_A + 1), ln.candc(jj * CF_WORDS + CF_A + 2));
      V3l ax = mk3<F>(ln.candc(jj * CF_WORDS + CF_AX), ln.candc(jj * CF_WORDS + CF_AX + 1), ln.candc(jj * CF_WORDS + CF_AX + 2));
      F r = ln.candc(jj * CF_WORDS + CF_R), link = ln.candc(jj * CF_WORDS + CF_LINK);
      F az = dot(gez[g], ax);
      F len = lm::sqrt_(lm::max_(one - az * az, zero));      // 1 for spheres and vertices (ax = 0)
      F dpt = gz0[g] + dot(gez[g], A) - r * len;
      F d_st = dpt, d_fl = far_;
      if (TERRAIN) {
        if (terr) {
          V3l Ew;
          F rs;
          BoxRec nx;
          if constexpr (L::kPrefetchShapes) nx = load_box(ex->shapes);    // (lanes.hpp WithShapePrefetch; one record past the list is readable row scratch)
          cand_eval_point(ln, bs, R, tgR[g], tgp[g], gez[g], A, ax, r, az, len, Ew, rs);
          for (int si = 0; si < n_shapes; si++) {
            F ds, isb;
            V3l ns;
            if constexpr (L::kPrefetchShapes) {
              const BoxRec rec = nx;
              nx = load_box(ex->shapes + (si + 1) * 8);
              terrain_sdf_rec<F>(ln, ex, rec, Ew, ds, ns, isb);
            } else {
              terrain_sdf<F>(ln, ex, si, Ew, ds, ns, isb);
            }
            dpt = lm::min_(dpt, ds - rs);
            if (want_touch) {
              B isf = ln.lane_f(si == ex->flag_shape ? 1.0f : 0.0f) > 0.5f;
              d_fl = lm::sel(isf, lm::min_(d_fl, ds - rs), d_fl);
              d_st = lm::sel(isf, d_st, lm::min_(d_st, ds - rs));
            }
          }
        }
        if (want_touch) {
          B body = lm::and_(link > 0.5f, ln.candc(jj * CF_WORDS + CF_KIND) < 0.5f);
          tch_st = lm::sel(lm::and_(body, d_st < P.margin_dist), zero, tch_st);
          tch_fl = lm::sel(lm::and_(body, d_fl < P.margin_dist), zero, tch_fl);
        }
      }
      depth[jj] = lm::sel(lm::and_(dpt < P.margin_dist, link > -0.5f), dpt, far_);
    }
    if constexpr (TERRAIN) {
      depth[8] = far_;
      if constexpr (XROWS) depth[9] = far_;
      if (terr) {                                             // sub-lanes 0, 1: the two ends of the leg's terrain edge under the body box
        F rd;
        V3l rP, rn;
        reverse_edge(ln, P, ex, bs, R, L::i2f(ln.sub()), rd, rP, rn);
        depth[8] = lm::sel(lm::and_(sub_lt2, rd < P.margin_dist), rd, far_);
        if constexpr (XROWS) if (P.leg_edges) {               // every sub-lane: terrain edge `sub` of the listed boxes across the leg's thigh and shank boxes (round 6)
          F ld, ll, lsh;
          V3l lP, lnw;
          leg_edge(ln, P, ex, bs, R, k, legc, L::i2f(ln.sub()), ld, lP, lnw, ll, lsh);
          depth[9] = lm::sel(ld < P.margin_dist, ld, far_);
          if (want_touch) {                                   // a leg box on a box's edge is a leg link touching that box (the flag, or a static)
            const B on = ld < P.margin_dist, isf = lm::abs_(lsh - (float)ex->flag_shape) < 0.5f;
            tch_fl = lm::sel(lm::and_(on, isf), zero, tch_fl);
            tch_st = lm::sel(lm::and_(on, lm::not_(isf)), zero, tch_st);
          }
        }
      }
      if (want_touch) {
        ex->touch_static = L::rmin(tch_st) < 0.5f ? 1.0f : 0.0f;
        ex->touch_flag = L::rmin(tch_fl) < 0.5f ? 1.0f : 0.0f;
      }
    }
    PMC_PHASE("sub.selection");
    // the leg keeps its 4 deepest candidates, slot s = s-th pick: four rounds of (quad min, claim).  Candidates within LLM_SELECT_EPS of
    // the deepest count as equally deep and the lowest candidate index (sub-lane, then position) wins: symmetric poses put several
    // points at the same depth up to rounding, and which of them is kept must not depend on the arithmetic.
    F my_depth = far_, my_sub = zero, my_jj = zero;
    LL_UNROLL
    for (int s = 0; s < PMC_K; s++) {
      if (s >= P.max_contacts) break;                       // (spec override LLM_SPEC_MAX_CONTACTS_PER_LEG; 4 unless a deviation study says otherwise)
      F m = depth[0];
      for (int jj = 1; jj < NC; jj++) m = lm::min_(m, depth[jj]);
      F mq = L::submin(m);
      F thr = mq + (float)LLM_SELECT_EPS;
      // first own candidate within the tolerance of the leg's deepest: key = jj if depth <= thr, jj + 100 otherwise, as
      // clamp(jj + (depth - thr) * 1e30, jj, jj + 100) -- a v_fma and a v_med3 per candidate instead of a compare feeding a select
      // (which costs two wait states on gfx950); the smallest key is the answer
      // (the difference is formed first: at depth == thr the key is exactly jj on the GPU's fused multiply-add as on the host's multiply and add)
      F am = ln.lane_f(100.0f);
      for (int jj = NC - 1; jj >= 0; jj--) am = lm::min_(am, lm::med3_((depth[jj] - thr) * 1.0e30f + (float)jj, ln.lane_f((float)jj), ln.lane_f((float)jj + 100.0f)));
      F code = lm::sel(lm::and_(am < 50.0f, mq < far_), L::i2f(ln.sub()) * STRIDE + am, ln.lane_f(1000.0f));
      F wcode = L::submin(code);                            // lowest candidate index among them (1000 = none)
      B winner = lm::and_(code <= wcode, code < 500.0f);
      F dsel = zero;
      for (int jj = 0; jj < NC; jj++) {
        B hit = lm::and_(winner, lm::abs_(am - (float)jj) < 0.5f);
        dsel = lm::sel(hit, depth[jj], dsel);
        depth[jj] = lm::sel(hit, far_, depth[jj]);
      }
      F wdepth = L::subsum(dsel);
      F wsub = lm::rint_(wcode * (1.0f / STRIDE) - (0.5f - 0.5f / STRIDE));   // floor(wcode / STRIDE) for wcode = STRIDE sub + jj, jj <= STRIDE / 2
      B owner = ln.is_sub(s);
      my_depth = lm::sel(owner, lm::sel(wcode < 500.0f, wdepth, far_), my_depth);
      my_sub = lm::sel(owner, wsub, my_sub);
      my_jj = lm::sel(owner, wcode - wsub * STRIDE, my_jj);
    }
    // store the kept candidates in candidate-index order (near-ties in depth must not reorder the solve): each slot lane
    // ranks its candidate among the leg's four and picks up the one whose rank equals its slot
    {
      F idx = lm::sel(my_depth < 1.0e29f, my_sub * STRIDE + my_jj, ln.lane_f(1000.0f) + L::i2f(ln.sub()));
      F i0 = L::template subbcast<0>(idx), i1 = L::template subbcast<1>(idx), i2 = L::template subbcast<2>(idx), i3 = L::template subbcast<3>(idx);
      F rank = lm::sel(i0 < idx, one, zero) + lm::sel(i1 < idx, one, zero) + lm::sel(i2 < idx, one, zero) + lm::sel(i3 < idx, one, zero);
      F me = L::i2f(ln.sub());
      F nd = my_depth, ns = my_sub, nj = my_jj;
      pick_rank<0>(ln, rank, me, my_depth, my_sub, my_jj, nd, ns, nj);
      pick_rank<1>(ln, rank, me, my_depth, my_sub, my_jj, nd, ns, nj);
      pick_rank<2>(ln, rank, me, my_depth, my_sub, my_jj, nd, ns, nj);
      pick_rank<3>(ln, rank, me, my_depth, my_sub, my_jj, nd, ns, nj);
      my_depth = nd; my_sub = ns; my_jj = nj;
    }
    const B cvalid = lm::and_(my_depth < 1.0e29f, ln.lane_f(PMC_ABL(1) ? 0.0f : 1.0f) > 0.5f);
    bool any_c[4], any_l[4] = {false, false, false, false};
    any_c[0] = L::any(lm::and_(cvalid, ln.is_sub(0))); any_c[1] = L::any(lm::and_(cvalid, ln.is_sub(1)));
    any_c[2] = L::any(lm::and_(cvalid, ln.is_sub(2))); any_c[3] = L::any(lm::and_(cvalid, ln.is_sub(3)));
    const bool any_contact = any_c[0] || any_c[1] || any_c[2] || any_c[3];

    PMC_TSS(24);
    PMC_PHASE("sub.limit_rows");
    // --- rows -------------------------------------------------------------------------------------------------------------------
    Row rl, rn, r1, r2;
    ConeX cx;                          // (CONE builds only; otherwise never touched and never allocated)
    F mu = zero;
    {   // joint-limit row of joint j = sub (sub-lane 3 holds none)
      B has = sub_lt3;
      F qj = lm::sel(sub_0, q[0], lm::sel(sub_lt2, q[1], q[2])), qsj = lm::sel(sub_0, qs[0], lm::sel(sub_lt2, qs[1], qs[2]));
      F qlo = lm::sel(sub_0, ln.legc(legc, LC_QLO), lm::sel(sub_lt2, ln.legc(legc, LC_QLO + 1), ln.legc(legc, LC_QLO + 2)));
      F qhi = lm::sel(sub_0, ln.legc(legc, LC_QHI), lm::sel(sub_lt2, ln.legc(legc, LC_QHI + 1), ln.legc(legc, LC_QHI + 2)));
      F dlo = qj - qlo, dhi = qhi - qj;
      B lower = dlo <= dhi;
      F d = lm::sel(lower, dlo, dhi), sg = lm::sel(lower, one, zero - one);
      F ljt[3], lgt[6];
      ljt[0] = lm::sel(sub_0, sg, zero);
      ljt[1] = lm::sel(lm::and_(sub_lt2, lm::not_(sub_0)), sg, zero);
      ljt[2] = lm::sel(lm::and_(sub_lt3, lm::not_(sub_lt2)), sg, zero);
      lm_fwd(lf, ljt);
      SV<F> g6 = scale(scale(lf.y1, ljt[0]) + scale(lf.y2, ljt[1]) + scale(lf.y3, ljt[2]), zero - one);
      sv_to6(g6, lgt);
      fwd6(Sb, Sd, lgt);
      F nn = ljt[0] * ljt[0] + ljt[1] * ljt[1] + ljt[2] * ljt[2];
      for (int i = 0; i < 6; i++) nn = nn + lgt[i] * lgt[i];
      // (the second ERP -- Bullet: none beyond 0.04 rad -- under a wave-uniform test of its own: nothing of it is live in the common path of the builds that are short of registers)
      F lerp = ln.lane_f(P.limit_erp * inv_dt);
      if (P.limit_erp_deep != P.limit_erp) lerp = lm::sel(d > P.erp_deep_below, lerp, ln.lane_f(P.limit_erp_deep * inv_dt));
      rl.c = sg * qsj + lm::sel(d > 0.0f, d * inv_dt, d * lerp);
      // which rows enter the solve.  limit_speculative (rounds 1 - 4): every row that can act within the substep (free approach speed below the gate);
      // otherwise Bullet's rule (btMultiBodyJointLimitConstraint): only a joint that is past its limit has a row -- rare, so most substeps
      // of most waves skip the limit section altogether (any_l below)
      B lvalid = lm::and_(lm::and_(has, P.limit_speculative ? (rl.c < P.limit_gate) : lm::not_(d > 0.0f)), ln.lane_f(PMC_ABL(2) ? 0.0f : 1.0f) > 0.5f);
      rl.inv = lm::sel(lvalid, one / nn, zero);
      any_l[0] = L::any(lm::and_(lvalid, ln.is_sub(0))); any_l[1] = L::any(lm::and_(lvalid, ln.is_sub(1)));
      any_l[2] = L::any(lm::and_(lvalid, ln.is_sub(2)));
      if (any_l[0] || any_l[1] || any_l[2]) {
        PMC_PHASE("sub.limit_gram");
        finish_row<true>(ln, rl, lgt, ljt);
      }
    }
    PMC_TSS(25);
    PMC_PHASE("sub.contact_geometry");
    if (any_contact) {
      // geometry of this lane's contact: candidate (my_sub, my_jj) of the leg, re-evaluated from the table
      // (a reverse candidate, jj = 8, has no table entry: it reads entry 7's and replaces what it needs below)
      const B isrevl = (TERRAIN && XROWS) ? lm::and_(cvalid, my_jj > 8.5f) : (zero > one);                     // a terrain edge across a leg box (leg_edge)
      const B isrev = TERRAIN ? lm::and_(lm::and_(cvalid, my_jj > 7.5f), lm::not_(isrevl)) : (zero > one);   // a terrain edge under the trunk (reverse_edge)
      // (a slot without a candidate decodes the 'none' code to indices outside the table: clamped, its row is dead anyway)
      I wsub = L::f2i(lm::min_(my_sub, ln.lane_f(3.0f))), wbase = L::f2i(lm::min_(my_jj, ln.lane_f(7.0f))) * CF_WORDS;
      V3l A = mk3<F>(ln.candc_of(wsub, wbase + CF_A), ln.candc_of(wsub, wbase + CF_A + 1), ln.candc_of(wsub, wbase + CF_A + 2));
      V3l ax = mk3<F>(ln.candc_of(wsub, wbase + CF_AX), ln.candc_of(wsub, wbase + CF_AX + 1), ln.candc_of(wsub, wbase + CF_AX + 2));
      V3l fb = mk3<F>(ln.candc_of(wsub, wbase + CF_FB), ln.candc_of(wsub, wbase + CF_FB + 1), ln.candc_of(wsub, wbase + CF_FB + 2));
      F r = ln.candc_of(wsub, wbase + CF_R), link = ln.candc_of(wsub, wbase + CF_LINK), kind = ln.candc_of(wsub, wbase + CF_KIND);
      if (TERRAIN) link = lm::sel(isrev, zero, link);          // the body box
      mu = lm::sel(kind > 0.5f, ln.lane_f(ex ? ex->mu_foot : P.mu_foot), ln.lane_f(P.mu_link));
      B l1 = link < 1.5f, l2 = link < 2.5f, l0 = link < 0.5f;       // link: 0 base, 1 hip, 2 thigh, 3 shank
      V3l ezk = mk3<F>(lm::sel(l0, ez.x, lm::sel(l1, ez1.x, lm::sel(l2, ez2.x, ez3.x))), lm::sel(l0, ez.y, lm::sel(l1, ez1.y, lm::sel(l2, ez2.y, ez3.y))),
                       lm::sel(l0, ez.z, lm::sel(l1, ez1.z, lm::sel(l2, ez2.z, ez3.z))));
      F az = dot(ezk, ax);
      F len = lm::sqrt_(lm::max_(one - az * az, zero));
      B degenerate = len < 1e-6f;
      F il = lm::sel(degenerate, zero, one / lm::max_(len, ln.lane_f(1e-12f)));
      V3l dir = mk3<F>(lm::sel(degenerate, fb.x, (ezk.x - az * ax.x) * il), lm::sel(degenerate, fb.y, (ezk.y - az * ax.y) * il),
                       lm::sel(degenerate, fb.z, (ezk.z - az * ax.z) * il));
      V3l x = A - scale(dir, r);
      // P_b = p_k + R_k x   (link 0: the base frame itself)
      V3l x1 = k.p1 + mul(k.R1, x), x2 = k.p2 + mul(k.R2, x), x3 = k.p3 + mul(k.R3, x);
      V3l Pb = mk3<F>(lm::sel(l0, x.x, lm::sel(l1, x1.x, lm::sel(l2, x2.x, x3.x))), lm::sel(l0, x.y, lm::sel(l1, x1.y, lm::sel(l2, x2.y, x3.y))),
                      lm::sel(l0, x.z, lm::sel(l1, x1.z, lm::sel(l2, x2.z, x3.z))));
      // TERRAIN: which surface the kept candidate touches -- the plane (normal +z) or the nearest shape -- decides the row directions
      V3l un = cvt3<F>(ezb), ut1 = mk3<F>(ln.lane_f(-R.m[3]), ln.lane_f(-R.m[4]), ln.lane_f(-R.m[5])), ut2 = mk3<F>(ln.lane_f(R.m[0]), ln.lane_f(R.m[1]), ln.lane_f(R.m[2]));
      if (TERRAIN) {
        if (terr) {
          M3<F> lR;
          V3l lp;
          for (int i = 0; i < 9; i++) lR.m[i] = lm::sel(l0, ln.lane_f((i % 4 == 0) ? 1.0f : 0.0f), lm::sel(l1, k.R1.m[i], lm::sel(l2, k.R2.m[i], k.R3.m[i])));
          lp = mk3<F>(lm::sel(l0, zero, lm::sel(l1, k.p1.x, lm::sel(l2, k.p2.x, k.p3.x))), lm::sel(l0, zero, lm::sel(l1, k.p1.y, lm::sel(l2, k.p2.y, k.p3.y))),
                      lm::sel(l0, zero, lm::sel(l1, k.p1.z, lm::sel(l2, k.p2.z, k.p3.z))));
          V3l Ew;
          F rs;
          BoxRec nx;
          if constexpr (L::kPrefetchShapes) nx = load_box(ex->shapes);
          cand_eval_point(ln, bs, R, lR, lp, ezk, A, ax, r, az, len, Ew, rs);
          F best = Ew.z - rs;                                  // the plane
          V3l nw = mk3<F>(zero, zero, one);
          F scale_mu = one;
          for (int si = 0; si < n_shapes; si++) {
            F ds, isb;
            V3l ns;
            if constexpr (L::kPrefetchShapes) {
              const BoxRec rec = nx;
              nx = load_box(ex->shapes + (si + 1) * 8);
              terrain_sdf_rec<F>(ln, ex, rec, Ew, ds, ns, isb);
            } else {
              terrain_sdf<F>(ln, ex, si, Ew, ds, ns, isb);
            }
            B win = (ds - rs) < best;
            best = lm::sel(win, ds - rs, best);
            nw = mk3<F>(lm::sel(win, ns.x, nw.x), lm::sel(win, ns.y, nw.y), lm::sel(win, ns.z, nw.z));
            scale_mu = lm::sel(win, lm::sel(isb > 0.5f, ln.lane_f(ex->box_mu_scale), one), scale_mu);
          }
          if (L::any(isrev)) {                                  // the terrain edge's point and the body box face's normal instead
            F rd;
            V3l rP, rn;
            reverse_edge(ln, P, ex, bs, R, my_sub, rd, rP, rn);
            Pb = mk3<F>(lm::sel(isrev, rP.x, Pb.x), lm::sel(isrev, rP.y, Pb.y), lm::sel(isrev, rP.z, Pb.z));
            nw = mk3<F>(lm::sel(isrev, rn.x, nw.x), lm::sel(isrev, rn.y, nw.y), lm::sel(isrev, rn.z, nw.z));
            scale_mu = lm::sel(isrev, ln.lane_f(ex->box_mu_scale), scale_mu);
            rs = lm::sel(isrev, zero, rs);
          }
          if constexpr (XROWS) if (L::any(isrevl)) {            // ... or the terrain edge's point and the LEG box face's normal; the row acts on that box's link
            F rd, rl, rsh;
            V3l rP, rn;
            leg_edge(ln, P, ex, bs, R, k, legc, my_sub, rd, rP, rn, rl, rsh);
            Pb = mk3<F>(lm::sel(isrevl, rP.x, Pb.x), lm::sel(isrevl, rP.y, Pb.y), lm::sel(isrevl, rP.z, Pb.z));
            nw = mk3<F>(lm::sel(isrevl, rn.x, nw.x), lm::sel(isrevl, rn.y, nw.y), lm::sel(isrevl, rn.z, nw.z));
            scale_mu = lm::sel(isrevl, ln.lane_f(ex->box_mu_scale), scale_mu);
            rs = lm::sel(isrevl, zero, rs);
            link = lm::sel(isrevl, rl, link);
          }
          mu = mu * scale_mu;
          // btPlaneSpace1(n): two tangents; for n = +z they are -y and +x, the directions of the flat-ground rows
          B steep = lm::abs_(nw.z) > 0.7071067811865475f;
          F a_s = nw.y * nw.y + nw.z * nw.z, a_f = nw.x * nw.x + nw.y * nw.y;
          F ks = lm::rsqrt_(lm::max_(a_s, ln.lane_f(1e-12f))), kf = lm::rsqrt_(lm::max_(a_f, ln.lane_f(1e-12f)));
          V3l p_s = mk3<F>(zero, zero - nw.z * ks, nw.y * ks), q_s = mk3<F>(a_s * ks, zero - nw.x * (nw.y * ks), nw.x * (zero - nw.z * ks));
          V3l p_f = mk3<F>(zero - nw.y * kf, nw.x * kf, zero), q_f = mk3<F>(zero - nw.z * (nw.x * kf), nw.z * (zero - nw.y * kf), a_f * kf);
          V3l t1w = mk3<F>(lm::sel(steep, p_s.x, p_f.x), lm::sel(steep, p_s.y, p_f.y), lm::sel(steep, p_s.z, p_f.z));
          V3l t2w = mk3<F>(lm::sel(steep, q_s.x, q_f.x), lm::sel(steep, q_s.y, q_f.y), lm::sel(steep, q_s.z, q_f.z));
          un = mulT(R, nw); ut1 = mulT(R, t1w); ut2 = mulT(R, t2w);
          Pb = Pb + scale(ez - un, rs);                         // a sphere touches where the surface's normal leaves it, not at its lowest point
        }
      }
      F depth_c = my_depth;
      F cerp = ln.lane_f(P.erp * inv_dt);
      if (P.erp_deep != P.erp) cerp = lm::sel(depth_c > P.erp_deep_below, cerp, ln.lane_f(P.erp_deep * inv_dt));       // (LLM_SPEC_ERP_DEEP: not the spec; wave-uniform, see the limit rows)
      F bias = lm::sel(depth_c > 0.0f, depth_c * inv_dt, lm::max_(depth_c * cerp, ln.lane_f(-P.max_depen)));
      // joint j moves the point iff the point's link is at or below joint j: link >= j+1
      F on1 = lm::sel(link > 0.5f, one, zero), on2 = lm::sel(link > 1.5f, one, zero), on3 = lm::sel(link > 2.5f, one, zero);
      V3l rr1 = Pb - k.p1, rr2 = Pb - k.p2, rr3 = Pb - k.p3;
      V3l a1v = mk3<F>(one, zero, zero);
      V3l d1 = scale(cross(a1v, rr1), on1), d2 = scale(cross(k.a2, rr2), on2), d3 = scale(cross(k.a2, rr3), on3);
      if (P.friction_dirs) {
        // LLM_SPEC_FRICTION_DIRS = 1 (deviation study; Bullet's default direction rule as published in convertMultiBodyContact): the first
        // friction direction runs along the lateral velocity of the contact point after the unconstrained update, the second is t1 x n;
        // a point that does not slide keeps btPlaneSpace1(n)
        V3l wv = mk3<F>(ln.lane_f(xi[0]), ln.lane_f(xi[1]), ln.lane_f(xi[2]));
        V3l vp = cross(wv, Pb) + scale(d1, qs[0]) + scale(d2, qs[1]) + scale(d3, qs[2]);
        vp = mk3<F>(vp.x + xi[3], vp.y + xi[4], vp.z + xi[5]);
        F vn = dot(vp, un);
        V3l lat = vp - scale(un, vn);
        F l2 = dot(lat, lat);
        B slides = l2 > 1.1920929e-7f;
        F il = lm::rsqrt_(lm::max_(l2, ln.lane_f(1e-30f)));
        V3l a1 = scale(lat, il), a2 = cross(a1, un);
        ut1 = mk3<F>(lm::sel(slides, a1.x, ut1.x), lm::sel(slides, a1.y, ut1.y), lm::sel(slides, a1.z, ut1.z));
        ut2 = mk3<F>(lm::sel(slides, a2.x, ut2.x), lm::sel(slides, a2.y, ut2.y), lm::sel(slides, a2.z, ut2.z));
      }
      PMC_PHASE("sub.contact_rows");
      // rows n = +z, t1 = -y, t2 = +x (world), expressed in F0
      if constexpr (CONE && L::kGramPipe && !L::kConeInLds) {
        contact_rows_cone_piped(ln, rn, r1, r2, cx, un, ut1, ut2, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, bias, cvalid);
      } else {
      contact_row(ln, rn, un, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, bias, cvalid);
      if (CONE) {
        F g1[6], j1[3], g2[6], j2[3];
        contact_row(ln, r1, ut1, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, zero, cvalid, g1, j1);
        contact_row(ln, r2, ut2, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, zero, cvalid, g2, j2);
        cross_gram(ln, r1.inv, g1, j1, g2, j2, cx.n12);
        cross_gram(ln, r2.inv, g2, j2, g1, j1, cx.n21);
        if (L::kConeInLds)                                   // the 256-register builds: the 32 scalars wait in the row's LDS scratch, the round reads them block by block
          for (int S = 0; S < 4; S++) {
            ln.cone_store(0, S, cx.n12[S], cx.n12[4 + S], cx.n12[8 + S], cx.n12[12 + S]);
            ln.cone_store(1, S, cx.n21[S], cx.n21[4 + S], cx.n21[8 + S], cx.n21[12 + S]);
          }
      } else {
        contact_row(ln, r1, ut1, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, zero, cvalid);
        contact_row(ln, r2, ut2, Pb, d1, d2, d3, lf, Sb, Sd, xi, qs, zero, cvalid);
      }
      }
    }

    PMC_PHASE("sub.self_detect");
    // --- self-collision (LR:212-217: links of different legs): each leg is two capsules, the closest pairs within the
    //     margin give up to two frictionless rows.  Lane (leg g, sub s) tests capsule (s & 2 ? shank : thigh) of its own leg against
    //     capsule (s & 1 ? shank : thigh) of the previous leg, and -- legs 0 and 1 only -- of the leg two away: 24 pairs in two passes.
    SelfRow sr[2], sf[2][2];             // sf[slot][0 / 1]: the contact's two tangential rows (LLM_SPEC_SELF_FRICTION > 0 only)
    self_row_clear(ln, sr[0]); self_row_clear(ln, sr[1]);
    const bool self_fric = XROWS && P.self_friction > 0.0f;
    if (self_fric) { self_row_clear(ln, sf[0][0]); self_row_clear(ln, sf[0][1]); self_row_clear(ln, sf[1][0]); self_row_clear(ln, sf[1][1]); }
    bool any_self = false;
    int n_self_w = 0;                   // self-collision slots in use by some env of the wave
    if (P.self_collision > 0.5f && !PMC_ABL(512)) {                                 // (ablation 512: no self-collision at all)
      V3l TA = k.p2 + mul(k.R2, ld3c(ln, legc, LC_CAPS)), TB = k.p2 + mul(k.R2, ld3c(ln, legc, LC_CAPS + 3));
      V3l SA = k.p3 + mul(k.R3, ld3c(ln, legc, LC_CAPS + 6)), SB = k.p3 + mul(k.R3, ld3c(ln, legc, LC_CAPS + 9));
      F rT = ln.legc(legc, LC_CAPS + 12), rS = ln.legc(legc, LC_CAPS + 13);
      B oth_s = lm::odd_(ln.sub()), own_s = lm::bit1_(ln.sub());
      V3l a1 = mk3<F>(lm::sel(own_s, SA.x, TA.x), lm::sel(own_s, SA.y, TA.y), lm::sel(own_s, SA.z, TA.z));
      V3l b1 = mk3<F>(lm::sel(own_s, SB.x, TB.x), lm::sel(own_s, SB.y, TB.y), lm::sel(own_s, SB.z, TB.z));
      F r1 = lm::sel(own_s, rS, rT);
      F cd[2], cpair[2];
      V3l cP[2], cN[2];
      F subf = L::i2f(ln.sub()), legf = ln.legf();
      for (int pass = 0; pass < 2; pass++) {
        V3l oTA, oTB, oSA, oSB;
        F orT, orS;
        if (pass == 0) {
          oTA = mk3<F>(L::from_prev_leg(TA.x), L::from_prev_leg(TA.y), L::from_prev_leg(TA.z)); oTB = mk3<F>(L::from_prev_leg(TB.x), L::from_prev_leg(TB.y), L::from_prev_leg(TB.z));
          oSA = mk3<F>(L::from_prev_leg(SA.x), L::from_prev_leg(SA.y), L::from_prev_leg(SA.z)); oSB = mk3<F>(L::from_prev_leg(SB.x), L::from_prev_leg(SB.y), L::from_prev_leg(SB.z));
          orT = L::from_prev_leg(rT); orS = L::from_prev_leg(rS);
        } else {
          oTA = mk3<F>(L::from_leg2(TA.x), L::from_leg2(TA.y), L::from_leg2(TA.z)); oTB = mk3<F>(L::from_leg2(TB.x), L::from_leg2(TB.y), L::from_leg2(TB.z));
          oSA = mk3<F>(L::from_leg2(SA.x), L::from_leg2(SA.y), L::from_leg2(SA.z)); oSB = mk3<F>(L::from_leg2(SB.x), L::from_leg2(SB.y), L::from_leg2(SB.z));
          orT = L::from_leg2(rT); orS = L::from_leg2(rS);
        }
        V3l a2 = mk3<F>(lm::sel(oth_s, oSA.x, oTA.x), lm::sel(oth_s, oSA.y, oTA.y), lm::sel(oth_s, oSA.z, oTA.z));
        V3l b2 = mk3<F>(lm::sel(oth_s, oSB.x, oTB.x), lm::sel(oth_s, oSB.y, oTB.y), lm::sel(oth_s, oSB.z, oTB.z));
        F r2 = lm::sel(oth_s, orS, orT);
        // the closest points of two nearly parallel axes depend on which segment the algorithm treats first: the spec's pair (A, B)
        // has A = the previous leg in pass 0 (the OTHER capsule) and A = the own leg in pass 1 -- the oracle's order
        V3l c1, c2;
        if (pass == 0) seg_seg(ln, a2, b2, a1, b1, c2, c1);
        else seg_seg(ln, a1, b1, a2, b2, c1, c2);
        V3l dd = c1 - c2;
        F len = lm::sqrt_(dot(dd, dd));
        F il = one / lm::max_(len, ln.lane_f(1e-9f));
        cN[pass] = scale(dd, il);                                              // from the other leg's capsule to this leg's
        cP[pass] = scale((c1 - scale(cN[pass], r1)) + (c2 + scale(cN[pass], r2)), ln.lane_f(0.5f));
        F dep = len - r1 - r2;
        // the oracle's pair index (tie-break and identity): leg pairs (0,1) (1,2) (2,3) (3,0) (0,2) (1,3), A first;
        // capsule bits: bit 0 = A's, bit 1 = B's.  Pass 0: A = the previous leg (other), B = own; pass 1: A = own, B = other.
        F lp = (pass == 0) ? lm::sel(legf < 0.5f, ln.lane_f(3.0f), legf - one) : legf + ln.lane_f(4.0f);
        F sp = (pass == 0) ? subf : lm::sel(own_s, one, zero) + lm::sel(oth_s, ln.lane_f(2.0f), zero);
        cpair[pass] = lp * 4.0f + sp;
        B valid = lm::and_(dep < P.self_margin, len > 1e-9f);
        if (pass == 1) valid = lm::and_(valid, legf < 1.5f);                    // pairs {0,2} and {1,3}, once each
        cd[pass] = lm::sel(valid, dep, far_);
      }
      any_self = L::any(lm::min_(cd[0], cd[1]) < 1.0e29f) && !PMC_ABL(256) && P.max_self > 0;          // (ablation 256: detection only)
      if (any_self) {
        PMC_PHASE("sub.self_rows");
        LL_UNROLL
        for (int slot = 0; slot < 2; slot++) {
          if (slot >= P.max_self) break;                                             // (spec override LLM_SPEC_MAX_SELF)
          if (slot == 1 && !L::any(lm::min_(cd[0], cd[1]) < 1.0e29f)) break;        // nobody in the wave has a second one
          n_self_w = slot + 1;
          F dlane = lm::min_(cd[0], cd[1]);
          float dmin = L::rmin(dlane);
          B at0 = cd[0] <= ln.lane_f(dmin + (float)LLM_SELECT_EPS), at1 = cd[1] <= ln.lane_f(dmin + (float)LLM_SELECT_EPS);   // equally deep within the tolerance: lower pair index
          F code = lm::min_(lm::sel(at0, cpair[0], ln.lane_f(1.0e9f)), lm::sel(at1, cpair[1], ln.lane_f(1.0e9f)));
          float cmin = L::rmin(code);
          const bool have = dmin < 1.0e29f;
          B win0 = lm::and_(at0, lm::abs_(cpair[0] - cmin) < 0.5f), win1 = lm::and_(lm::and_(at1, lm::abs_(cpair[1] - cmin) < 0.5f), lm::not_(win0));
          F w6[6];
          float u6[6];
          w6[0] = lm::sel(win0, cP[0].x, lm::sel(win1, cP[1].x, zero)); w6[1] = lm::sel(win0, cP[0].y, lm::sel(win1, cP[1].y, zero));
          w6[2] = lm::sel(win0, cP[0].z, lm::sel(win1, cP[1].z, zero)); w6[3] = lm::sel(win0, cN[0].x, lm::sel(win1, cN[1].x, zero));
          w6[4] = lm::sel(win0, cN[0].y, lm::sel(win1, cN[1].y, zero)); w6[5] = lm::sel(win0, cN[0].z, lm::sel(win1, cN[1].z, zero));
          L::rsum6(w6, u6);
          const float dsel = have ? L::rmin(lm::sel(win0, cd[0], lm::sel(win1, cd[1], far_))) : 0.0f;   // the chosen pair's own depth (within the tolerance of dmin)
          cd[0] = lm::sel(win0, far_, cd[0]); cd[1] = lm::sel(win1, far_, cd[1]);
          // who is who: pair index -> (own leg: sign +, other leg: sign -) and the links (2 thigh, 3 shank)
          const int pair = have ? (int)(cmin + 0.5f) : 0, lpi = pair >> 2, spi = pair & 3;
          const int legA = lpi < 4 ? lpi : lpi - 4, legB = lpi < 4 ? ((lpi + 1) & 3) : lpi - 2;
          const int own = lpi < 4 ? legB : legA, oth = lpi < 4 ? legA : legB;
          const int own_link = 2 + (lpi < 4 ? (spi >> 1) : (spi & 1)), oth_link = 2 + (lpi < 4 ? (spi & 1) : (spi >> 1));
          B is_own = ln.is_leg(own), is_oth = ln.is_leg(oth);
          F sgn = lm::sel(is_own, one, lm::sel(is_oth, zero - one, zero));
          F link = lm::sel(is_own, ln.lane_f((float)own_link), ln.lane_f((float)oth_link));
          V3l Pb = mk3<F>(ln.lane_f(u6[0]), ln.lane_f(u6[1]), ln.lane_f(u6[2])), nb = mk3<F>(ln.lane_f(u6[3]), ln.lane_f(u6[4]), ln.lane_f(u6[5]));
          F on3 = lm::sel(link > 2.5f, one, zero);
          V3l a1v = mk3<F>(one, zero, zero);
          V3l e1 = cross(a1v, Pb - k.p1), e2 = cross(k.a2, Pb - k.p2), e3 = scale(cross(k.a2, Pb - k.p3), on3);
          self_row_build(ln, sr[slot], nb, sgn, e1, e2, e3, lf, Sb, Sd, qs, row_bias(P, dsel, inv_dt), have);
          if (self_fric) {
            // LLM_SPEC_SELF_FRICTION (round 6, the engine twin of the oracle's switch): two tangential rows along btPlaneSpace1 of the WORLD normal, behind the normal row
            V3<float> nw = mul(R, mk3<float>(u6[3], u6[4], u6[5])), t1w, t2w;
            plane_space(nw, t1w, t2w);
            self_row_build(ln, sf[slot][0], cvt3<F>(mulT(R, t1w)), sgn, e1, e2, e3, lf, Sb, Sd, qs, 0.0f, have);
            self_row_build(ln, sf[slot][1], cvt3<F>(mulT(R, t2w)), sgn, e1, e2, e3, lf, Sb, Sd, qs, 0.0f, have);
          }
        }
      }
    }
    // --- robot-robot contact (SEPMC): each robot is ten capsules -- thigh and shank-with-foot of every leg as in the
    //     self-collision test, and two for the trunk.  Both rows evaluate all 100 pairs with robot 0's capsule as the first segment, on
    //     bit-identical inputs (world end points computed once and copied across), so both find the same two deepest contacts.
    constexpr int NPAIR = (PAIR && XROWS) ? LLM_MAX_PAIR_CAP : 2;       // robot-robot contacts a build can carry (LLM_SPEC_MAX_PAIR: 2 by default, up to 4 in the XROWS build)
    SelfRow pr[NPAIR], pf[(PAIR && XROWS) ? NPAIR : 1][2];              // pf[slot][0 / 1]: the contact's two tangential rows (LLM_SPEC_PAIR_FRICTION > 0, XROWS builds)
    bool any_pair = false;
    int n_pair_w = 0;
    const bool pair_fric = PAIR && XROWS && P.pair_friction > 0.0f;
    const int max_pair = (PAIR && XROWS) ? P.max_pair : 2;
    if (PAIR) {
      for (int i_ = 0; i_ < NPAIR; i_++) self_row_clear(ln, pr[i_]);
      if (PAIR && XROWS) for (int i_ = 0; i_ < NPAIR; i_++) { self_row_clear(ln, pf[i_][0]); self_row_clear(ln, pf[i_][1]); }
      if (ex->want_touch) ex->touch_robot = 0.0f;
      if (L::any(ln.lane_f(ex->pair_active ? 1.0f : 0.0f) > 0.5f)) {
        const int me = ex->pair_me;
        const B act = ln.lane_f(ex->pair_active ? 1.0f : 0.0f) > 0.5f, i_am_0 = ln.lane_f(me == 0 ? 1.0f : 0.0f) > 0.5f;
        // my capsules in world coordinates: [0] thigh, [1] shank, [2] the trunk capsule this leg holds
        V3l wa[3], wb[3];
        F wr[3];
        {
          V3l ab[3] = {k.p2 + mul(k.R2, ld3c(ln, legc, LC_CAPS)), k.p3 + mul(k.R3, ld3c(ln, legc, LC_CAPS + 6)), ld3c(ln, legc, LC_TRUNKCAP)};
          V3l bb[3] = {k.p2 + mul(k.R2, ld3c(ln, legc, LC_CAPS + 3)), k.p3 + mul(k.R3, ld3c(ln, legc, LC_CAPS + 9)), ld3c(ln, legc, LC_TRUNKCAP + 3)};
          for (int c = 0; c < 3; c++) {
            V3l x = mul(R, ab[c]), y = mul(R, bb[c]);
            wa[c] = mk3<F>(x.x + bs.p.x, x.y + bs.p.y, x.z + bs.p.z);
            wb[c] = mk3<F>(y.x + bs.p.x, y.y + bs.p.y, y.z + bs.p.z);
          }
          wr[0] = ln.legc(legc, LC_CAPS + 12); wr[1] = ln.legc(legc, LC_CAPS + 13); wr[2] = ln.legc(legc, LC_TRUNKCAP + 6);
        }
        V3l oa[3], ob[3];                                    // the other robot's, same lane
        for (int c = 0; c < 3; c++) {
          oa[c] = mk3<F>(ln.peer(wa[c].x), ln.peer(wa[c].y), ln.peer(wa[c].z));
          ob[c] = mk3<F>(ln.peer(wb[c].x), ln.peer(wb[c].y), ln.peer(wb[c].z));
        }
        const B s_odd = lm::odd_(ln.sub()), s_hi = lm::bit1_(ln.sub()), g_odd = lm::odd_(ln.leg());
        const F subf = L::i2f(ln.sub()), legf = ln.legf();
        const F lo_bit = lm::sel(s_odd, one, zero), hi_bit = lm::sel(s_hi, one, zero);
        // Every lane keeps its two best (depth, pair id) with point and normal: exact for the two deepest of the 100 pairs.  LLM_SPEC_MAX_PAIR > 2 (XROWS builds): the pairs are
        // walked a second time without the two already taken, for the third and fourth deepest (a lane's own two best may both be among the row's four: one scan cannot know).
        float taken[2] = {-1.0f, -1.0f};
        LL_UNROLL
        for (int scan = 0; scan < ((PAIR && XROWS) ? 2 : 1); scan++) {
        if (scan == 1 && (max_pair <= 2 || !any_pair)) break;
        F bd[2] = {far_, far_}, bid[2] = {far_, far_};
        V3l bP[2], bN[2];
        bP[0] = bP[1] = bN[0] = bN[1] = mk3<F>(zero, zero, zero);
        F tch = one;
        for (int pass = 0; pass < 7; pass++) {
          V3l ma, mb, ta, tb;                                // my capsule, their capsule
          F mr, tr, mi, ti;                                  // radii and capsule indices (0..7 legs, 8..9 trunk)
          B pass_ok = act;
          if (pass < 4) {                                    // my leg capsule (sub bit 0) x their leg (leg + pass) capsule (sub bit 1)
            ma = mk3<F>(lm::sel(s_odd, wa[1].x, wa[0].x), lm::sel(s_odd, wa[1].y, wa[0].y), lm::sel(s_odd, wa[1].z, wa[0].z));
            mb = mk3<F>(lm::sel(s_odd, wb[1].x, wb[0].x), lm::sel(s_odd, wb[1].y, wb[0].y), lm::sel(s_odd, wb[1].z, wb[0].z));
            mr = lm::sel(s_odd, wr[1], wr[0]);
            mi = legf * 2.0f + lo_bit;
            V3l ra[2], rb[2];
            for (int c = 0; c < 2; c++) {
              if (pass == 0) { ra[c] = oa[c]; rb[c] = ob[c]; }
              else if (pass == 1) { ra[c] = mk3<F>(L::from_next_leg(oa[c].x), L::from_next_leg(oa[c].y), L::from_next_leg(oa[c].z)); rb[c] = mk3<F>(L::from_next_leg(ob[c].x), L::from_next_leg(ob[c].y), L::from_next_leg(ob[c].z)); }
              else if (pass == 2) { ra[c] = mk3<F>(L::from_leg2(oa[c].x), L::from_leg2(oa[c].y), L::from_leg2(oa[c].z)); rb[c] = mk3<F>(L::from_leg2(ob[c].x), L::from_leg2(ob[c].y), L::from_leg2(ob[c].z)); }
              else { ra[c] = mk3<F>(L::from_prev_leg(oa[c].x), L::from_prev_leg(oa[c].y), L::from_prev_leg(oa[c].z)); rb[c] = mk3<F>(L::from_prev_leg(ob[c].x), L::from_prev_leg(ob[c].y), L::from_prev_leg(ob[c].z)); }
            }
            ta = mk3<F>(lm::sel(s_hi, ra[1].x, ra[0].x), lm::sel(s_hi, ra[1].y, ra[0].y), lm::sel(s_hi, ra[1].z, ra[0].z));
            tb = mk3<F>(lm::sel(s_hi, rb[1].x, rb[0].x), lm::sel(s_hi, rb[1].y, rb[0].y), lm::sel(s_hi, rb[1].z, rb[0].z));
            tr = lm::sel(s_hi, wr[1], wr[0]);                 // (same model: their radii are mine)
            F tl = legf + (float)pass;
            tl = lm::sel(tl > 3.5f, tl - 4.0f, tl);
            ti = tl * 2.0f + hi_bit;
          } else if (pass == 4) {                            // my trunk capsule (sub bit 1) x their leg `leg` capsule (sub bit 0)
            B own_t = lm::or_(lm::and_(s_hi, g_odd), lm::and_(lm::not_(s_hi), lm::not_(g_odd)));     // the capsule this leg holds is the wanted one
            V3l na = mk3<F>(L::from_next_leg(wa[2].x), L::from_next_leg(wa[2].y), L::from_next_leg(wa[2].z)), nb_ = mk3<F>(L::from_next_leg(wb[2].x), L::from_next_leg(wb[2].y), L::from_next_leg(wb[2].z));
            ma = mk3<F>(lm::sel(own_t, wa[2].x, na.x), lm::sel(own_t, wa[2].y, na.y), lm::sel(own_t, wa[2].z, na.z));
            mb = mk3<F>(lm::sel(own_t, wb[2].x, nb_.x), lm::sel(own_t, wb[2].y, nb_.y), lm::sel(own_t, wb[2].z, nb_.z));
            mr = wr[2]; mi = hi_bit + 8.0f;
            ta = mk3<F>(lm::sel(s_odd, oa[1].x, oa[0].x), lm::sel(s_odd, oa[1].y, oa[0].y), lm::sel(s_odd, oa[1].z, oa[0].z));
            tb = mk3<F>(lm::sel(s_odd, ob[1].x, ob[0].x), lm::sel(s_odd, ob[1].y, ob[0].y), lm::sel(s_odd, ob[1].z, ob[0].z));
            tr = lm::sel(s_odd, wr[1], wr[0]); ti = legf * 2.0f + lo_bit;
          } else if (pass == 5) {                            // my leg capsule (sub bit 0) x their trunk capsule (sub bit 1)
            B own_t = lm::or_(lm::and_(s_hi, g_odd), lm::and_(lm::not_(s_hi), lm::not_(g_odd)));
            V3l na = mk3<F>(L::from_next_leg(oa[2].x), L::from_next_leg(oa[2].y), L::from_next_leg(oa[2].z)), nb_ = mk3<F>(L::from_next_leg(ob[2].x), L::from_next_leg(ob[2].y), L::from_next_leg(ob[2].z));
            ta = mk3<F>(lm::sel(own_t, oa[2].x, na.x), lm::sel(own_t, oa[2].y, na.y), lm::sel(own_t, oa[2].z, na.z));
            tb = mk3<F>(lm::sel(own_t, ob[2].x, nb_.x), lm::sel(own_t, ob[2].y, nb_.y), lm::sel(own_t, ob[2].z, nb_.z));
            tr = wr[2]; ti = hi_bit + 8.0f;
            ma = mk3<F>(lm::sel(s_odd, wa[1].x, wa[0].x), lm::sel(s_odd, wa[1].y, wa[0].y), lm::sel(s_odd, wa[1].z, wa[0].z));
            mb = mk3<F>(lm::sel(s_odd, wb[1].x, wb[0].x), lm::sel(s_odd, wb[1].y, wb[0].y), lm::sel(s_odd, wb[1].z, wb[0].z));
            mr = lm::sel(s_odd, wr[1], wr[0]); mi = legf * 2.0f + lo_bit;
          } else {                                           // trunk x trunk, on the lanes of leg 0: mine (sub bit 0) x theirs (sub bit 1)
            V3l nma = mk3<F>(L::from_next_leg(wa[2].x), L::from_next_leg(wa[2].y), L::from_next_leg(wa[2].z)), nmb = mk3<F>(L::from_next_leg(wb[2].x), L::from_next_leg(wb[2].y), L::from_next_leg(wb[2].z));
            V3l nta = mk3<F>(L::from_next_leg(oa[2].x), L::from_next_leg(oa[2].y), L::from_next_leg(oa[2].z)), ntb = mk3<F>(L::from_next_leg(ob[2].x), L::from_next_leg(ob[2].y), L::from_next_leg(ob[2].z));
            ma = mk3<F>(lm::sel(s_odd, nma.x, wa[2].x), lm::sel(s_odd, nma.y, wa[2].y), lm::sel(s_odd, nma.z, wa[2].z));
            mb = mk3<F>(lm::sel(s_odd, nmb.x, wb[2].x), lm::sel(s_odd, nmb.y, wb[2].y), lm::sel(s_odd, nmb.z, wb[2].z));
            ta = mk3<F>(lm::sel(s_hi, nta.x, oa[2].x), lm::sel(s_hi, nta.y, oa[2].y), lm::sel(s_hi, nta.z, oa[2].z));
            tb = mk3<F>(lm::sel(s_hi, ntb.x, ob[2].x), lm::sel(s_hi, ntb.y, ob[2].y), lm::sel(s_hi, ntb.z, ob[2].z));
            mr = wr[2]; tr = wr[2]; mi = lo_bit + 8.0f; ti = hi_bit + 8.0f;
            pass_ok = lm::and_(pass_ok, legf < 0.5f);
          }
          // canonical order: robot 0's capsule is the first segment
          V3l a1 = mk3<F>(lm::sel(i_am_0, ma.x, ta.x), lm::sel(i_am_0, ma.y, ta.y), lm::sel(i_am_0, ma.z, ta.z)), b1 = mk3<F>(lm::sel(i_am_0, mb.x, tb.x), lm::sel(i_am_0, mb.y, tb.y), lm::sel(i_am_0, mb.z, tb.z));
          V3l a2 = mk3<F>(lm::sel(i_am_0, ta.x, ma.x), lm::sel(i_am_0, ta.y, ma.y), lm::sel(i_am_0, ta.z, ma.z)), b2 = mk3<F>(lm::sel(i_am_0, tb.x, mb.x), lm::sel(i_am_0, tb.y, mb.y), lm::sel(i_am_0, tb.z, mb.z));
          F r1 = lm::sel(i_am_0, mr, tr), r2 = lm::sel(i_am_0, tr, mr);
          F id = lm::sel(i_am_0, mi * 10.0f + ti, ti * 10.0f + mi);
          V3l c1, c2;
          F ps, pt;
          seg_seg_st(ln, a1, b1, a2, b2, c1, c2, ps, pt);
          V3l dd = c1 - c2;                                  // from robot 1's capsule to robot 0's
          F len = lm::sqrt_(dot(dd, dd));
          V3l nn_ = scale(dd, one / lm::max_(len, ln.lane_f(1e-9f)));
          V3l pp = scale((c1 - scale(nn_, r1)) + (c2 + scale(nn_, r2)), ln.lane_f(0.5f));
          F dep = len - r1 - r2;
          B valid = lm::and_(pass_ok, lm::and_(dep < P.margin_dist, len > 1e-9f));
          if (scan == 1) valid = lm::and_(valid, lm::and_(lm::abs_(id - taken[0]) > 0.5f, lm::abs_(id - taken[1]) > 0.5f));
          // my side of it: a leg / wheel link unless it is the foot end of a shank capsule
          F my_par = lm::sel(i_am_0, ps, pt);
          B my_body = lm::and_(mi < 7.5f, lm::not_(lm::and_(lm::abs_(mi - lm::rint_(mi * 0.5f) * 2.0f) > 0.5f, my_par > 0.9f)));
          tch = lm::sel(lm::and_(valid, my_body), zero, tch);
          F d = lm::sel(valid, dep, far_);
          const float se = (float)LLM_SELECT_EPS;               // deeper by more than the tolerance, or equally deep with the lower id
          B beat0 = lm::and_(d < 1.0e29f, lm::or_(d < bd[0] - se, lm::and_(d <= bd[0] + se, id < bid[0])));
          B beat1 = lm::and_(lm::and_(lm::not_(beat0), d < 1.0e29f), lm::or_(d < bd[1] - se, lm::and_(d <= bd[1] + se, id < bid[1])));
          // shift down, then insert
          bd[1] = lm::sel(beat0, bd[0], lm::sel(beat1, d, bd[1])); bid[1] = lm::sel(beat0, bid[0], lm::sel(beat1, id, bid[1]));
          bP[1] = mk3<F>(lm::sel(beat0, bP[0].x, lm::sel(beat1, pp.x, bP[1].x)), lm::sel(beat0, bP[0].y, lm::sel(beat1, pp.y, bP[1].y)), lm::sel(beat0, bP[0].z, lm::sel(beat1, pp.z, bP[1].z)));
          bN[1] = mk3<F>(lm::sel(beat0, bN[0].x, lm::sel(beat1, nn_.x, bN[1].x)), lm::sel(beat0, bN[0].y, lm::sel(beat1, nn_.y, bN[1].y)), lm::sel(beat0, bN[0].z, lm::sel(beat1, nn_.z, bN[1].z)));
          bd[0] = lm::sel(beat0, d, bd[0]); bid[0] = lm::sel(beat0, id, bid[0]);
          bP[0] = mk3<F>(lm::sel(beat0, pp.x, bP[0].x), lm::sel(beat0, pp.y, bP[0].y), lm::sel(beat0, pp.z, bP[0].z));
          bN[0] = mk3<F>(lm::sel(beat0, nn_.x, bN[0].x), lm::sel(beat0, nn_.y, bN[0].y), lm::sel(beat0, nn_.z, bN[0].z));
        }
        if (scan == 0 && ex->want_touch) ex->touch_robot = L::rmin(tch) < 0.5f ? 1.0f : 0.0f;
        const bool any_here = L::any(bd[0] < 1.0e29f);
        if (scan == 0) any_pair = any_here && max_pair > 0;
        if (any_here && max_pair > 0) {
          LL_UNROLL
          for (int s2 = 0; s2 < 2; s2++) {
            const int slot = 2 * scan + s2;
            if (slot >= max_pair) break;
            if (s2 == 1 && !L::any(bd[0] < 1.0e29f)) break;
            n_pair_w = slot + 1;
            const float dmin = L::rmin(bd[0]);
            B at = bd[0] <= ln.lane_f(dmin + (float)LLM_SELECT_EPS);                       // equally deep within the tolerance: lower pair id
            const float imin = L::rmin(lm::sel(at, bid[0], ln.lane_f(1.0e9f)));
            const bool have = dmin < 1.0e29f;
            B win = lm::and_(at, lm::abs_(bid[0] - imin) < 0.5f);
            const float dsel = have ? L::rmin(lm::sel(win, bd[0], far_)) : 0.0f;
            // several lanes may hold the same pair (never: each pair is evaluated by exactly one lane of a row)
            F w6[6] = {lm::sel(win, bP[0].x, zero), lm::sel(win, bP[0].y, zero), lm::sel(win, bP[0].z, zero), lm::sel(win, bN[0].x, zero), lm::sel(win, bN[0].y, zero), lm::sel(win, bN[0].z, zero)};
            float u6[6];
            L::rsum6(w6, u6);
            // the winner lane promotes its second best
            bd[0] = lm::sel(win, bd[1], bd[0]); bid[0] = lm::sel(win, bid[1], bid[0]);
            bP[0] = mk3<F>(lm::sel(win, bP[1].x, bP[0].x), lm::sel(win, bP[1].y, bP[0].y), lm::sel(win, bP[1].z, bP[0].z));
            bN[0] = mk3<F>(lm::sel(win, bN[1].x, bN[0].x), lm::sel(win, bN[1].y, bN[0].y), lm::sel(win, bN[1].z, bN[0].z));
            bd[1] = lm::sel(win, far_, bd[1]);
            const int pid = have ? (int)(imin + 0.5f) : 0, ia = pid / 10, ib = pid - 10 * ia;
            const int ic = me == 0 ? ia : ib;                          // my capsule: 0..7 leg (leg = ic >> 1, thigh / shank), 8..9 trunk
            const float sg = me == 0 ? 1.0f : -1.0f;                   // the normal points from robot 1 to robot 0
            // into my base frame
            const V3<float> pw = mk3<float>(u6[0] - bs.p.x, u6[1] - bs.p.y, u6[2] - bs.p.z), nw = mk3<float>(u6[3] * sg, u6[4] * sg, u6[5] * sg);
            const V3<float> pb_ = mulT(R, pw), nb_ = mulT(R, nw);
            V3l Pb = cvt3<F>(pb_), nb = cvt3<F>(nb_);
            B mine = lm::and_(ln.is_leg(ic >> 1), ln.lane_f(ic < 8 ? 1.0f : 0.0f) > 0.5f);
            F on3 = ln.lane_f((ic & 1) ? 1.0f : 0.0f);
            V3l a1v = mk3<F>(one, zero, zero);
            V3l e1 = cross(a1v, Pb - k.p1), e2 = cross(k.a2, Pb - k.p2), e3 = scale(cross(k.a2, Pb - k.p3), on3);
            if (scan == 0) taken[s2] = have ? imin : -1.0f;
            pair_row_build(ln, pr[slot], nb_, pb_, mine, e1, e2, e3, lf, Sb, Sd, xi, qs, P, dsel, inv_dt, true, have, me);
            if (pair_fric) {
              // LLM_SPEC_PAIR_FRICTION (round 6, the engine twin of the oracle's switch): two tangential rows along btPlaneSpace1 of the contact's WORLD normal (robot 1 -> robot 0,
              // the same bits in both rows of the arena), each solved right behind its normal row inside +- mu x that row's multiplier
              V3<float> t1w, t2w;
              plane_space(mk3<float>(u6[3], u6[4], u6[5]), t1w, t2w);
              pair_row_build(ln, pf[slot][0], mulT(R, scale(t1w, sg)), pb_, mine, e1, e2, e3, lf, Sb, Sd, xi, qs, P, 0.0f, inv_dt, false, have, me);
              pair_row_build(ln, pf[slot][1], mulT(R, scale(t2w, sg)), pb_, mine, e1, e2, e3, lf, Sb, Sd, xi, qs, P, 0.0f, inv_dt, false, have, me);
            }
          }
        }
        }
      }
    }
    PMC_TSS(26);
    PMC_PHASE("sub.pgs_setup");
#if defined(PMC_ABLATION)
    if (PMC_ABL(16) && ln.is_lane(0)) P.counters[4 + (long)env * PMC_TS_SLOTS + 28] += (unsigned long long)(any_self ? 1 : 0);
    if (PMC_ABL(16) && ln.is_lane(0)) {   // solver occupancy: active 4-turn blocks of this wave, contact and limit
      P.counters[4 + (long)env * PMC_TS_SLOTS + 30] += (unsigned long long)(any_c[0] + any_c[1] + any_c[2] + any_c[3]);
      P.counters[4 + (long)env * PMC_TS_SLOTS + 31] += (unsigned long long)(any_l[0] + any_l[1] + any_l[2]);
    }
#endif
    // --- projected Gauss-Seidel in whitened coordinates, rows in registers ------------------------------------------------------
    // Order of the spec: limit rows (joint, leg), then normal rows (slot, leg), t1 rows, t2 rows.
    F VA = zero, VB = zero, VJ = zero;     // sum gt * lambda (env-uniform dx[6]) and sum jt * lambda (leg-uniform dq[3]), scattered over the sub-lanes
    const F big = ln.lane_f(3.0e38f);
    const bool any_limit = any_l[0] || any_l[1] || any_l[2];
    ln.prepare_turn_masks();
    // The iteration loop, with the wave-uniform tests of its body (limit rows? contact rows? leg-leg rows?) hoisted out of it where the build is worth the code: a lone wave
    // pays ~12 cycles for a branch it does not take and ~29 for one it takes (tools/branch_probe.hip, profiles/r06_branch_probe.txt; the issue ledger had charged 4.4), and the
    // three tests ran a hundred times a control step.  LIM_K / CON_K: 1 = known true, 0 = known false, -1 = tested inside (the generic loop); SELF_K: the number of leg-leg slots in use (0, 1, 2), -1 = tested.  Same operations in
    // the same order whichever copy runs: bit-identical.
    auto pgs_loop = [&](auto lim_k, auto con_k, auto self_k) __attribute__((always_inline)) {
      constexpr int LIM_K = decltype(lim_k)::value, CON_K = decltype(con_k)::value, SELF_K = decltype(self_k)::value;
      LL_NOUNROLL
      for (int it = 0; it < P.n_iter; it++) {
        PMC_PHASE("pgs.limit_round");
        if (LIM_K > 0 || (LIM_K < 0 && any_limit)) gs_round<true, true>(ln, rl, big, VA, VB, VJ);
        PMC_PHASE("pgs.normal_round");
        if (CON_K > 0 || (CON_K < 0 && any_contact)) {
          gs_round<false, true>(ln, rn, big, VA, VB, VJ);
          F hi = mu * rn.lam;
          PMC_PHASE("pgs.cone_round");
          if (CONE) {
            gs_cone_round(ln, r1, r2, cx, hi, VA, VB, VJ);
          } else {
            gs_round<false, false>(ln, r1, hi, VA, VB, VJ);
            gs_round<false, false>(ln, r2, hi, VA, VB, VJ);
          }
        }
        PMC_PHASE("pgs.self_turns");
        if (SELF_K > 0 || (SELF_K < 0 && any_self)) {                          // then the self-collision rows, one after the other (with LLM_SPEC_SELF_FRICTION each followed by its two tangential rows)
          // (a tangential row whose normal multiplier is zero is bounded to [0, 0]: unless it still carries a multiplier of its own its turn changes nothing -- skipped when that holds
          //  for every env of the wave: exact, and most leg-leg contacts of a step are within the margin without pressing)
          self_turn(ln, sr[0], VA, VB, VJ);
          if (self_fric && L::any(ln.lane_f((sr[0].lam > 0.0f || sf[0][0].lam != 0.0f || sf[0][1].lam != 0.0f) ? 1.0f : 0.0f) > 0.5f)) {
            self_fric_turn(ln, sf[0][0], P.self_friction * sr[0].lam, VA, VB, VJ); self_fric_turn(ln, sf[0][1], P.self_friction * sr[0].lam, VA, VB, VJ);
          }
          if (SELF_K > 1 || (SELF_K < 0 && n_self_w > 1)) {
            self_turn(ln, sr[1], VA, VB, VJ);
            if (self_fric && L::any(ln.lane_f((sr[1].lam > 0.0f || sf[1][0].lam != 0.0f || sf[1][1].lam != 0.0f) ? 1.0f : 0.0f) > 0.5f)) {
              self_fric_turn(ln, sf[1][0], P.self_friction * sr[1].lam, VA, VB, VJ); self_fric_turn(ln, sf[1][1], P.self_friction * sr[1].lam, VA, VB, VJ);
            }
          }
        }
        if (PAIR) {
          if (any_pair) {                                                      // last, the rows shared with the other robot (each followed by its two tangential rows under LLM_SPEC_PAIR_FRICTION)
            LL_UNROLL
            for (int slot = 0; slot < NPAIR; slot++) {
              if (slot >= n_pair_w) break;
              pair_turn(ln, pr[slot], VA, VB, VJ, ex->pair_me);
              if constexpr (PAIR && XROWS) {
                if (pair_fric && L::any(ln.lane_f((pr[slot].lam > 0.0f || pf[slot][0].lam != 0.0f || pf[slot][1].lam != 0.0f) ? 1.0f : 0.0f) > 0.5f)) {      // (see the leg-leg rows)
                  pair_fric_turn(ln, pf[slot][0], P.pair_friction * pr[slot].lam, VA, VB, VJ, ex->pair_me);
                  pair_fric_turn(ln, pf[slot][1], P.pair_friction * pr[slot].lam, VA, VB, VJ, ex->pair_me);
                }
              }
            }
          }
        }
      }
    };
    {
      typedef std::integral_constant<int, 1> K1;
      typedef std::integral_constant<int, 0> K0;
      typedef std::integral_constant<int, -1> KT;
      constexpr bool HOIST = PMC_PGS_HOIST && (!PAIR || PMC_PGS_HOIST_PAIR) && !XROWS && (!TERRAIN || L::kConeInLds || PMC_PGS_HOIST_TERRAIN);      // (the XROWS builds keep the one generic loop; which of the other builds take the copies was decided by A/B: profiles/r06_pgs_hoist_ab.txt)
      if (HOIST && any_contact) {
        typedef std::integral_constant<int, 2> K2;
        const int ns = any_self ? (n_self_w > 1 ? 2 : 1) : 0;           // leg-leg slots in use by some env of the wave
        if (any_limit) { if (ns == 0) pgs_loop(K1(), K1(), K0()); else if (ns == 1) pgs_loop(K1(), K1(), K1()); else pgs_loop(K1(), K1(), K2()); }
        else           { if (ns == 0) pgs_loop(K0(), K1(), K0()); else if (ns == 1) pgs_loop(K0(), K1(), K1()); else pgs_loop(K0(), K1(), K2()); }
      } else pgs_loop(KT(), KT(), KT());
    }

    PMC_TSS(27);
    PMC_PHASE("sub.integrate");
    // back to velocities: d(xi) = Lb^-T dx ; d(qd) = Lm^-T (dq - Y^T d(xi))
    float dx[6] = {L::template vel_dx<0>(VA, VB), L::template vel_dx<1>(VA, VB), L::template vel_dx<2>(VA, VB),
                   L::template vel_dx<3>(VA, VB), L::template vel_dx<4>(VA, VB), L::template vel_dx<5>(VA, VB)};
    F dq[3] = {L::template vel_dq<0>(VJ), L::template vel_dq<1>(VJ), L::template vel_dq<2>(VJ)};
    bwd6(Sb, Sd, dx);
    SV<float> dxi;
    dxi.a = mk3<float>(dx[0], dx[1], dx[2]); dxi.l = mk3<float>(dx[3], dx[4], dx[5]);
    F du[3];
    du[0] = dq[0] - dot(lf.y1, dxi); du[1] = dq[1] - dot(lf.y2, dxi); du[2] = dq[2] - dot(lf.y3, dxi);
    lm_bwd(lf, du);
    for (int i = 0; i < 6; i++) xi[i] += dx[i];
    for (int j = 0; j < 3; j++) qs[j] = qs[j] + du[j];
    clip_velocities(ln, xi, qs, vmax);

    // --- integrate positions with the new velocities (semi-implicit Euler) ---------------------------------------------------------
    bs.w = mul(R, mk3<float>(xi[0], xi[1], xi[2]));
    bs.v = mul(R, mk3<float>(xi[3], xi[4], xi[5]));
    bs.p = bs.p + scale(bs.v, dt);
    Q4 dqt = quat_of_small_rotvec(scale(bs.w, dt));
    bs.q = qnormalize(qmul(dqt, qnormalize(bs.q)));
    for (int j = 0; j < 3; j++) {
      qd[j] = qs[j];
      q[j] = q[j] + qs[j] * dt;
    }
  }

  // ---------------------------------------------------------------------------------------------------
  // mocap reference (ML:65-166)
  // ---------------------------------------------------------------------------------------------------
  struct RefPose {
    V3u p;
    Q4 q;
    V3u v, w;
    F jp[3], jv[3];
  };
  // One interpolation site of a clip (ML:88-166) in two halves.  mocap_gather() does everything that touches the float64
  // rows and leaves floats; mocap_finish() is pure arithmetic.  A caller gathers all its sites first, so that the loads of
  // a control step are in flight together instead of one round trip per site (one wave per SIMD hides no latency).
  struct RefRaw {
    V3u p, vlin;        // interpolated position; finite-difference velocity (ML:137-140)
    Q4 qc, dl;          // current-frame quaternion; next - current, formed in float64 so the small rotation keeps its precision
    float ff;
    F jp[3], jv[3];
  };
  static LL_HD RefRaw mocap_gather(const L& ln, const double* fc, const double* fn, double frac, double frame_step) {
    RefRaw o;
    const double inv = 1.0 / frame_step;
    const double d0 = fn[0] - fc[0], d1 = fn[1] - fc[1], d2 = fn[2] - fc[2];
    o.p = mk3<float>((float)(fc[0] + frac * d0), (float)(fc[1] + frac * d1), (float)(fc[2] + frac * d2));
    o.vlin = mk3<float>((float)(d0 * inv), (float)(d1 * inv), (float)(d2 * inv));
    o.qc.x = (float)fc[3]; o.qc.y = (float)fc[4]; o.qc.z = (float)fc[5]; o.qc.w = (float)fc[6];
    o.dl.x = (float)(fn[3] - fc[3]); o.dl.y = (float)(fn[4] - fc[4]); o.dl.z = (float)(fn[5] - fc[5]); o.dl.w = (float)(fn[6] - fc[6]);
    o.ff = (float)frac;
    for (int j = 0; j < 3; j++) {
      D c = ln.lddl(fc, 7 + j, 3), n = ln.lddl(fn, 7 + j, 3);
      o.jp[j] = L::d2f(c + (n - c) * frac);                 // ML:157
      o.jv[j] = L::d2f((n - c) * inv);                      // ML:158
    }
    return o;
  }
  static LL_HD RefPose mocap_finish(const RefRaw& r, double frame_step, bool want_vel) {
    RefPose o;
    o.p = r.p;
    Q4 qci = qconj(r.qc);
    Q4 e = qmul(qci, r.dl);                                 // qc^-1 qn = 1 + qc^-1 (qn - qc)
    V3u rv = rotvec_of(mk3<float>(e.x, e.y, e.z), 1.0f + e.w);   // ML:127-134 (scipy Slerp)
    Q4 dq = quat_of_rotvec(mk3<float>(rv.x * r.ff, rv.y * r.ff, rv.z * r.ff));
    o.q = qmul(r.qc, dq);
    for (int j = 0; j < 3; j++) { o.jp[j] = r.jp[j]; o.jv[j] = r.jv[j]; }
    if (want_vel) {
      o.v = r.vlin;
      Q4 e2 = qmul(r.dl, qci);                              // qn qc^-1 = 1 + (qn - qc) qc^-1            ML:143-149
      V3u rv2 = rotvec_of(mk3<float>(e2.x, e2.y, e2.z), 1.0f + e2.w);
      float ang = sqrtf(rv2.x * rv2.x + rv2.y * rv2.y + rv2.z * rv2.z);
      float kk = ang / (ang + 1e-8f) * (float)(1.0 / frame_step);
      o.w = mk3<float>(rv2.x * kk, rv2.y * kk, rv2.z * kk);
    } else {
      o.v = mk3<float>(0.f, 0.f, 0.f);
      o.w = o.v;
    }
    return o;
  }

  // ---------------------------------------------------------------------------------------------------
  // observation (PLE:247-317).  The obs row doubles as the history store: frames 1,2 of the previous row are
  // frames 0,1 of the new one (deque maxlen 3, PLE:145-147); `fill` = reset pre-fill (PLE:282-290).
  // obs_gather() reads (history, the four future mocap sites), obs_emit() writes; every load of the row is issued
  // before its first store because the compiler must assume the two alias.
  // ---------------------------------------------------------------------------------------------------
  static constexpr int OBS_HIST_CHUNKS = 5;   // 2 * prop_dim <= 66 floats, 16 per chunk
  // The four future-goal sites (1/30, 1/15, 1/3, 1 s ahead; ML:75-86) are env-level work: four slerps and four relative rotations with
  // their atan2 / sin / cos.  Leg l of the row takes site l -- the base-pose part of a site is gathered and computed lane-varying,
  // one site's worth of instructions instead of four; only the reference joint angles stay per (site, leg).
  struct FutBase {
    V3l p;               // interpolated base position of this leg's site
    Q4T<F> qc, dl;       // current-frame quaternion; next - current (formed in float64)
    F ff;                // frame fraction
  };
  struct ObsIn {
    F h[OBS_HIST_CHUNKS], ha[2];
    FutBase futb;
    F fut_jp[4][3];      // reference joint angles of the lane's leg at the four sites
  };
  static LL_HD void obs_gather_futures(const L& ln, const StepParams& P, ObsIn& in, const double* clip_rows, int frame_id, double frac) {
    const double hz[4] = {1. / 30., 1. / 15., 1. / 3., 1.};                     // ML:75-86
    int fidh[4];
    double ffh[4];
    LL_UNROLL
    for (int h = 0; h < 4; h++) {
      double t = P.frame_step * frac + hz[h];
      fidh[h] = (int)floor(t / P.frame_step);
      ffh[h] = t / P.frame_step - fidh[h];
      const double* fc = clip_rows + (long)(frame_id + fidh[h]) * 19;
      for (int j = 0; j < 3; j++) {
        D c = ln.lddl(fc, 7 + j, 3), n = ln.lddl(fc + 19, 7 + j, 3);
        in.fut_jp[h][j] = L::d2f(c + (n - c) * ffh[h]);                          // ML:157
      }
    }
    const I idx = ln.pick4i((frame_id + fidh[0]) * 19, (frame_id + fidh[1]) * 19, (frame_id + fidh[2]) * 19, (frame_id + fidh[3]) * 19);
    const D ffd = ln.pick4d(ffh[0], ffh[1], ffh[2], ffh[3]);
    D c[7], n[7];
    for (int i = 0; i < 7; i++) { c[i] = ln.ldd_idx(clip_rows, idx + i); n[i] = ln.ldd_idx(clip_rows, idx + (19 + i)); }
    in.futb.p = mk3<F>(L::d2f(c[0] + ffd * (n[0] - c[0])), L::d2f(c[1] + ffd * (n[1] - c[1])), L::d2f(c[2] + ffd * (n[2] - c[2])));
    in.futb.qc.x = L::d2f(c[3]); in.futb.qc.y = L::d2f(c[4]); in.futb.qc.z = L::d2f(c[5]); in.futb.qc.w = L::d2f(c[6]);
    in.futb.dl.x = L::d2f(n[3] - c[3]); in.futb.dl.y = L::d2f(n[4] - c[4]); in.futb.dl.z = L::d2f(n[5] - c[5]); in.futb.dl.w = L::d2f(n[6] - c[6]);
    in.futb.ff = L::d2f(ffd);
  }
  static LL_HD ObsIn obs_gather(const L& ln, const StepParams& P, const float* hist_row, bool fill, const double* clip_rows, int frame_id,
                                double frac) {
    ObsIn in;
    const int Pd = P.prop_dim;
    const long a0 = 3L * Pd;
    if (!fill) {
      for (int c = 0; c < OBS_HIST_CHUNKS; c++) in.h[c] = ln.ld16(hist_row + Pd, 16 * c, 2 * Pd);
      for (int c = 0; c < 2; c++) in.ha[c] = ln.ld16(hist_row + a0 + 12, 16 * c, 24);
    } else {
      for (int c = 0; c < OBS_HIST_CHUNKS; c++) in.h[c] = ln.lane_f(0.0f);
      in.ha[0] = in.ha[1] = ln.lane_f(0.0f);
    }
    obs_gather_futures(ln, P, in, clip_rows, frame_id, frac);
    return in;
  }
  // the part every env of the family shares: prop and action histories with the newest frame (PLE:247-260, :276-290; PGE:251-290)
  static LL_HD void obs_emit_core(const L& ln, const StepParams& P, float* row, bool fill, const ObsIn& in, const Base& bs, const M3<float>& R,
                                  const F* q, const F* qd, const F* act) {
    const int Pd = P.prop_dim;
    B lane3 = ln.legf() < 2.5f;
    const long a0 = 3L * Pd;
    if (!fill) {                                                    // history shift
      for (int c = 0; c < OBS_HIST_CHUNKS; c++) ln.st16(row, 16 * c, 2 * Pd, in.h[c]);
      for (int c = 0; c < 2; c++) ln.st16(row + a0, 16 * c, 24, in.ha[c]);
    }
    // --- newest prop frame (PLE:247-260); a reset pre-fills all three frames with it (PLE:282-290) ---
    V3u wl = mulT(R, bs.w), vl = mulT(R, bs.v);
    LL_NOUNROLL
    for (int kf = 2; kf >= (fill ? 0 : 2); kf--) {
      long fb = (long)kf * Pd;
      if (P.prop_off[0] >= 0) for (int j = 0; j < 3; j++) ln.stl(row, fb + P.prop_off[0] + j, 3, q[j]);      // joint_pos
      if (P.prop_off[1] >= 0) for (int j = 0; j < 3; j++) ln.stl(row, fb + P.prop_off[1] + j, 3, qd[j]);     // joint_vel
      if (P.prop_off[2] >= 0) ln.stl_if(lane3, row, fb + P.prop_off[2], 1, ln.pick3(vl.x, vl.y, vl.z));      // R^-1 v
      if (P.prop_off[3] >= 0) ln.stl_if(lane3, row, fb + P.prop_off[3], 1, ln.pick3(wl.x, wl.y, wl.z));      // R^-1 w
      if (P.prop_off[4] >= 0) ln.stl_if(lane3, row, fb + P.prop_off[4], 1, ln.pick3(R.m[6], R.m[7], R.m[8])); // R[2,:]
      for (int j = 0; j < 3; j++) ln.stl(row, a0 + 12 * kf + j, 3, act[j]);                                   // raw action (quirk Q3)
    }
  }
  static LL_HD void obs_emit(const L& ln, const StepParams& P, float* row, bool fill, const ObsIn& in, const Base& bs, const M3<float>& R,
                             const F* q, const F* qd, const F* act) {
    obs_emit_core(ln, P, row, fill, in, bs, R, q, qd, act);
    B lane3 = ln.legf() < 2.5f;
    // --- future goals (ML:75-86 + PLE:299-317): leg l computes site l (see FutBase) and stores its six numbers ---
    (void)lane3;
    long f0 = 3L * P.prop_dim + 36;
    const F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f);
    Q4 qbi = qconj(qnormalize(bs.q));
    const FutBase& fb = in.futb;
    Q4T<F> e = qmul_t(qconj_t(fb.qc, zero), fb.dl);                                 // qc^-1 qn = 1 + qc^-1 (qn - qc)
    V3l rv = rotvec_of_t(mk3<F>(e.x, e.y, e.z), one + e.w, zero);                    // ML:127-134 (scipy Slerp)
    Q4T<F> qf = qmul_t(fb.qc, quat_of_rotvec_t(mk3<F>(rv.x * fb.ff, rv.y * fb.ff, rv.z * fb.ff)));
    V3l dp = mulT(R, mk3<F>(fb.p.x - bs.p.x, fb.p.y - bs.p.y, fb.p.z - bs.p.z));
    V3l aa;
    axis_angle_scaled_t(qmul_t(qbi, qnormalize_t(qf)), &aa, zero);
    ln.stl(row, f0 + 0, 18, dp.x); ln.stl(row, f0 + 1, 18, dp.y); ln.stl(row, f0 + 2, 18, dp.z);
    ln.stl(row, f0 + 3, 18, aa.x); ln.stl(row, f0 + 4, 18, aa.y); ln.stl(row, f0 + 5, 18, aa.z);
    LL_UNROLL
    for (int h = 0; h < 4; h++)
      for (int j = 0; j < 3; j++) ln.stl(row, f0 + 18 * h + 6 + j, 3, in.fut_jp[h][j]);
  }

  // ---------------------------------------------------------------------------------------------------
  // state I/O (SoA [field][n_envs])
  // ---------------------------------------------------------------------------------------------------
  static LL_HD void load_state(const L& ln, const float* s, int N, int env, Base& bs, F* q, F* qd) {
    bs.p = mk3<float>(s[0 * N + env], s[1 * N + env], s[2 * N + env]);
    bs.q.x = s[3 * N + env]; bs.q.y = s[4 * N + env]; bs.q.z = s[5 * N + env]; bs.q.w = s[6 * N + env];
    bs.v = mk3<float>(s[7 * N + env], s[8 * N + env], s[9 * N + env]);
    bs.w = mk3<float>(s[10 * N + env], s[11 * N + env], s[12 * N + env]);
    for (int j = 0; j < 3; j++) {
      q[j] = ln.ldl(s, (long)(13 + j) * N + env, 3L * N);
      qd[j] = ln.ldl(s, (long)(25 + j) * N + env, 3L * N);
    }
  }
  static LL_HD void store_state(const L& ln, float* s, int N, int env, const Base& bs, const F* q, const F* qd) {
    B lane3 = ln.legf() < 2.5f;
    ln.stl_if(lane3, s, env, N, ln.pick3(bs.p.x, bs.p.y, bs.p.z));
    ln.stl(s, 3L * N + env, N, lm::sel(ln.is_leg(3), ln.lane_f(bs.q.w), ln.pick3(bs.q.x, bs.q.y, bs.q.z)));
    ln.stl_if(lane3, s, 7L * N + env, N, ln.pick3(bs.v.x, bs.v.y, bs.v.z));
    ln.stl_if(lane3, s, 10L * N + env, N, ln.pick3(bs.w.x, bs.w.y, bs.w.z));
    for (int j = 0; j < 3; j++) {
      ln.stl(s, (long)(13 + j) * N + env, 3L * N, q[j]);
      ln.stl(s, (long)(25 + j) * N + env, 3L * N, qd[j]);
    }
  }

  // ---------------------------------------------------------------------------------------------------
  // reset one env at (clip, t0): PLE:150-171 + ML:48-57.  Writes state, ghost, bookkeeping and the first obs.
  // ---------------------------------------------------------------------------------------------------
  static LL_HD void reset_env(const L& ln, const StepParams& P, int env, int clip, double t0) {
    const int N = P.n_envs;
    int fid = (int)floor(t0 / P.frame_step);                                  // ML:52
    double frac = (t0 - fid * P.frame_step) / P.frame_step;                    // ML:53
    const double* rows = P.frames + (long)P.clip_off[clip] * 19;
    RefRaw rr = mocap_gather(ln, rows + (long)fid * 19, rows + (long)(fid + 1) * 19, frac, P.frame_step);
    float* row = P.obs + (long)env * P.obs_dim;
    ObsIn oin = obs_gather(ln, P, row, true, rows, fid, frac);
    RefPose rp = mocap_finish(rr, P.frame_step, true);
    Base bs;
    bs.p = rp.p; bs.q = rp.q; bs.v = rp.v; bs.w = rp.w;
    store_state(ln, P.kin, N, env, bs, rp.jp, rp.jv);                          // PLE:162
    store_state(ln, P.state, N, env, bs, rp.jp, rp.jv);                        // PLE:163
    M3<float> R = qmat(qnormalize(bs.q));
    F zero3[3] = {ln.lane_f(0.0f), ln.lane_f(0.0f), ln.lane_f(0.0f)};
    obs_emit(ln, P, row, true, oin, bs, R, rp.jp, rp.jv, zero3);               // PLE:168-170
    V3l fw = foot_world(ln, P.legc, bs.p, R, rp.jp[0], rp.jp[1], rp.jp[2]);
    for (int c = 0; c < 3; c++) {
      F v = (c == 0) ? fw.x : (c == 1 ? fw.y : fw.z);
      ln.stl(P.feet, (long)c * N + env, 3L * N, v);
      ln.stl(P.feet, (long)(12 + c) * N + env, 3L * N, v);
    }
    // quad-uniform bookkeeping: every lane of the quad writes the same value
    P.time[env] = t0;
    P.clip[env] = clip;
    P.ep_steps[env] = 0;
    P.reward_sum[env] = 0.0f;
    if (P.set_obstacle) P.ob_id[env] = 0;                                     // PLE:179
  }

  // PLE:262-268 + PLE:341-346 for the batch: advance the episode's obstacle, then test the robot's collision shapes against
  // the box (0.05 x 1.0 x 2h, centred on the ground at the peak's (x, y), yawed to the mocap heading).  Shapes are
  // represented by their candidate points: box vertices, sphere centres and cylinder cap centres with their radius.
  static LL_HD bool obstacle_contact(const L& ln, const StepParams& P, int env, int clip, double t, const V3u& p, const M3<float>& R,
                                     const LegKin& k) {
    const int oc = P.ob_cnt[clip];
    if (oc <= 0) return false;                                          // PLE:342 `self._obstacle is not None`
    const double* tab = P.ob_table + (long)P.ob_off[clip] * 4;
    // getContactPoints (PLE:343) reports the contacts of the last stepSimulation -- the box where it stood during the substeps;
    // _update_obstacle (PLE:229, :262-268) has already moved it on for the next step by then
    int ob = P.ob_id[env];
    const float cx = (float)tab[ob * 4 + 0], cy = (float)tab[ob * 4 + 1], yaw = (float)tab[ob * 4 + 2];
    while (ob < oc - 1 && t > tab[ob * 4 + 3] + 0.5) ob++;              // PLE:264-265
    P.ob_id[env] = ob;
    const float cyaw = cosf(yaw), syaw = sinf(yaw), hx = 0.025f, hy = 0.5f, hz = P.ob_half_height;   // PLE:184
    F zero = ln.lane_f(0.0f), one = ln.lane_f(1.0f);
    B sub_lt2 = L::i2f(ln.sub()) < 1.5f, sub_lt3 = L::i2f(ln.sub()) < 2.5f, sub_0 = L::i2f(ln.sub()) < 0.5f;
    // link frames of the three candidate groups of this sub-lane (A: [3,3,2,2], B: [2,2,2,1], C: [3,0,0,0])
    M3<F> gR[3];
    V3l gp[3];
    for (int i = 0; i < 9; i++) {
      gR[0].m[i] = lm::sel(sub_lt2, k.R3.m[i], k.R2.m[i]);
      gR[1].m[i] = lm::sel(sub_lt3, k.R2.m[i], k.R1.m[i]);
      gR[2].m[i] = lm::sel(sub_0, k.R3.m[i], ln.lane_f((i % 4 == 0) ? 1.0f : 0.0f));
    }
    gp[0] = mk3<F>(lm::sel(sub_lt2, k.p3.x, k.p2.x), lm::sel(sub_lt2, k.p3.y, k.p2.y), lm::sel(sub_lt2, k.p3.z, k.p2.z));
    gp[1] = mk3<F>(lm::sel(sub_lt3, k.p2.x, k.p1.x), lm::sel(sub_lt3, k.p2.y, k.p1.y), lm::sel(sub_lt3, k.p2.z, k.p1.z));
    gp[2] = mk3<F>(lm::sel(sub_0, k.p3.x, zero), lm::sel(sub_0, k.p3.y, zero), lm::sel(sub_0, k.p3.z, zero));
    F hit = zero;
    for (int jj = 0; jj < 7; jj++) {
      const int g = jj < 4 ? 0 : (jj < 6 ? 1 : 2);
      V3l A = mk3<F>(ln.candc(jj * CF_WORDS + CF_A), ln.candc(jj * CF_WORDS + CF_A + 1), ln.candc(jj * CF_WORDS + CF_A + 2));
      F r = ln.candc(jj * CF_WORDS + CF_R), link = ln.candc(jj * CF_WORDS + CF_LINK);
      V3l Pb = gp[g] + mul(gR[g], A);
      V3l Pw = mul(R, Pb);
      F dxw = Pw.x + (p.x - cx), dyw = Pw.y + (p.y - cy), lz = Pw.z + p.z;
      F lx = dxw * cyaw + dyw * syaw, ly = dyw * cyaw - dxw * syaw;
      F qx = lm::abs_(lx) - hx, qy = lm::abs_(ly) - hy, qz = lm::abs_(lz) - hz;
      F ox = lm::max_(qx, zero), oy = lm::max_(qy, zero), oz = lm::max_(qz, zero);
      F sdf = lm::sqrt_(ox * ox + oy * oy + oz * oz) + lm::min_(lm::max_(qx, lm::max_(qy, qz)), zero);
      hit = lm::sel(lm::and_(sdf - r < P.margin_dist, link > -0.5f), one, hit);
    }
    return L::qsum(L::subsum(hit)) > 0.5f;
  }

  // sample (clip, t0) for a new episode: ML:59-63 + ML:50-51, Philox stream keyed on (seed; env, episode)
  static LL_HD void sample_start(const L& ln, const StepParams& P, int env, uint32_t episode, int* clip, double* t0, const double* cdf = nullptr) {
    if (!cdf) cdf = P.cdf;
    uint32_t r[4];
    philox4x32((uint32_t)env, episode, 0x5eedu, 0u, (uint32_t)P.seed, (uint32_t)(P.seed >> 32), r);
    double u1 = u01_from(r[0], r[1]), u2 = u01_from(r[2], r[3]);
    // first i with u1 < cdf[i] (np.random.choice's inverse-cdf search) == the number of entries <= u1, cdf being non-decreasing
    int c = ln.count_le16(cdf, P.n_clips, u1);
    if (c > P.n_clips - 1) c = P.n_clips - 1;
    *clip = c;
    *t0 = u2 * (P.frame_step * (double)(P.clip_len[c] - P.margin - 1));
  }

  // ---------------------------------------------------------------------------------------------------
  // the control step
  // ---------------------------------------------------------------------------------------------------
  // act_in: the env's actions, one register per joint of the lane's leg (read from P.actions or drawn by the caller)
  // OBST (set_obstacle builds): the jump obstacle of the episode is a static box the robot collides with during the substeps
  // sl: index of this control step inside its launch (ll_step_random_n runs n_steps of them back to back; 0 otherwise)
  template <bool OBST = false, bool CONE = false, bool XROWS = false>
  static LL_HD void step_env(const L& ln, const StepParams& P_in, int env, const F* act_in, int sl = 0) {
    const StepParams& P = ln.params(P_in);
    const int N = P.n_envs;
    Base bs;
    F q[3], qd[3], act[3], tgt[3];
    PMC_TS(0);
    PMC_PHASE("step.entry_loads");
    if (PMC_ABL(8)) return;
    load_state(ln, P.state, N, env, bs, q, qd);
    for (int j = 0; j < 3; j++) act[j] = act_in[j];
    pd_target(ln, q, act, tgt);                                              // PLE:199-200, LR:126-127
    double t = P.time[env], t_loc = t;
    const int clip = P.clip[env];
    const int clen = P.clip_len[clip];
    const double* rows = P.frames + (long)P.clip_off[clip] * 19;
    PMC_TS(1);
    SubstepExtra ex;
    if (OBST) {
      // PLE:182-193: 0.05 x 1.0 x 2h box (createMultiBody, mass 0, Bullet's default friction 0.5) at the pose the last reset /
      // _update_obstacle gave it; it takes part in the substeps when the step starts with the base within LLM_OBSTACLE_REACH of it
      ex.want_touch = false; ex.flag_shape = -1; ex.pair_active = false; ex.pair_me = 0; ex.has_push = false;
      ex.mu_foot = P.mu_foot; ex.box_mu_scale = (float)(LLM_LINK_FRICTION / LLM_PLANE_FRICTION);
      ex.n_shapes = 0; ex.shapes = ln.row_scratch();
      const int oc = P.ob_cnt[clip];
      if (oc > 0) {
        const double* tab = P.ob_table + ((long)P.ob_off[clip] + P.ob_id[env]) * 4;
        ex.ycx = (float)tab[0]; ex.ycy = (float)tab[1];
        const float yaw = (float)tab[2], ddx = bs.p.x - ex.ycx, ddy = bs.p.y - ex.ycy;
        ex.ycs = cosf(yaw); ex.ysn = sinf(yaw); ex.yawed = true;
        if (ddx * ddx + ddy * ddy < (float)(LLM_OBSTACLE_REACH * LLM_OBSTACLE_REACH)) ex.n_shapes = 1;
      }
      float* rec = ln.row_scratch();
      if (ln.lane0()) {
        rec[0] = -0.025f; rec[1] = 0.025f; rec[2] = -0.5f; rec[3] = 0.5f; rec[4] = -P.ob_half_height; rec[5] = P.ob_half_height; rec[6] = 0.0f; rec[7] = 0.0f;
      }
      ln.row_sync();
    }
    LinkC lkh;
    const LinkC* held = nullptr;
    if (L::kHoldLink) { lkh = own_link_held(ln, P.legc); held = &lkh; }
    PMC_PHASE("step.substep_loop");
    for (int s = 0; s < P.n_sub; s++) {                                      // PLE:202
      if (OBST) substep_impl<true, false, CONE, XROWS>(ln, P, bs, q, qd, tgt, env, s, &ex, held);
      else substep_impl<false, false, CONE, XROWS>(ln, P, bs, q, qd, tgt, env, s, nullptr, held);   // PLE:204-206
      t_loc = t;                                                             // PLE:208 motion.step(time BEFORE the increment), quirk Q2
      t += P.dt_d;                                                           // PLE:210
    PMC_TS(10 + (s < 20 ? s : 20));
    }
    PMC_PHASE("tail.mocap_gather");
    if (PMC_ABL(4)) { store_state(ln, P.state, N, env, bs, q, qd); P.time[env] = t; return; }
    int fid = (int)floor(t_loc / P.frame_step);                              // ML:66
    {                                                                        // keep a done-but-still-stepped env inside its clip
      int fmax = clen - P.frame_rate - 3;
      if (fid > fmax) fid = fmax;
    }
    double frac = (t_loc - fid * P.frame_step) / P.frame_step;               // ML:67

    // --- gather: every load the rest of the step needs is issued here, ahead of the first store (one wave per SIMD hides
    //     no memory latency, and the compiler may not move a load above a store it cannot prove distinct) ---
    float* row = P.obs + (long)env * P.obs_dim;
    RefRaw rr = mocap_gather(ln, rows + (long)fid * 19, rows + (long)(fid + 1) * 19, frac, P.frame_step);                  // PLE:217
    ObsIn oin = obs_gather(ln, P, row, false, rows, fid, frac);
    constexpr int TRAJ_CHUNKS = 13;                                          // obs_dim <= 207
    F told[TRAJ_CHUNKS];
    float t_neglogp = 0.0f, t_value = 0.0f;
    if (P.traj) {
      LL_UNROLL
      for (int c = 0; c < TRAJ_CHUNKS; c++) told[c] = ln.ld16(row, 16 * c, P.obs_dim);
      if (P.neglogp) t_neglogp = P.neglogp[env];
      if (P.value) t_value = P.value[env];
    }
    const int steps = P.ep_steps[env] + 1;                                    // PLE:197
    const float rsum0 = P.reward_sum[env];
    const double max_steps = P.max_steps[clip];
    const uint32_t ep0 = P.ep_count[env];
    RefPose rp = mocap_finish(rr, P.frame_step, true);

    PMC_TS(2);
    PMC_PHASE("tail.reward");
    if (P.scripted_state) {   // parity hook: the caller plays PyBullet (how the golden harness drove the reference)
      const float* ss = P.scripted_state + (long)env * 37;
      bs.p = mk3<float>(ss[0], ss[1], ss[2]);
      bs.q.x = ss[3]; bs.q.y = ss[4]; bs.q.z = ss[5]; bs.q.w = ss[6];
      bs.v = mk3<float>(ss[7], ss[8], ss[9]);
      bs.w = mk3<float>(ss[10], ss[11], ss[12]);
      for (int j = 0; j < 3; j++) { q[j] = ln.ldl(ss, 13 + j, 3); qd[j] = ln.ldl(ss, 25 + j, 3); }
    }
    // non-finite guard
    F fin = q[0] + q[1] + q[2] + qd[0] + qd[1] + qd[2];
    float chk = L::qsum(fin) + bs.p.x + bs.p.y + bs.p.z + bs.q.x + bs.q.y + bs.q.z + bs.q.w + bs.v.x + bs.v.y + bs.v.z + bs.w.x + bs.w.y + bs.w.z;
    bool bad = !(fabsf(chk) < 1e30f);

    Q4 qn = qnormalize(bs.q);
    M3<float> R = qmat(qn);
    const int ar = P.auto_reset;

    // --- reward (PLE:350-426) ---
    Base gb;
    gb.p = rp.p; gb.q = rp.q; gb.v = rp.v; gb.w = rp.w;
    Q4 gqn = qnormalize(rp.q);
    M3<float> Rg = qmat(gqn);
    V3l fd = foot_world(ln, P.legc, bs.p, R, q[0], q[1], q[2]);                                 // PLE:397
    V3l fk = foot_world(ln, P.legc, gb.p, Rg, rp.jp[0], rp.jp[1], rp.jp[2]);                    // PLE:398
    if (P.scripted_feet) {    // parity hook: getLinkStates answered by the caller
      const float* sf = P.scripted_feet + (long)env * 24;
      fd = mk3<F>(ln.ldl(sf, 0, 3), ln.ldl(sf, 1, 3), ln.ldl(sf, 2, 3));
      fk = mk3<F>(ln.ldl(sf, 12, 3), ln.ldl(sf, 13, 3), ln.ldl(sf, 14, 3));
    }
    F ejp = ln.lane_f(0.0f), ejv = ln.lane_f(0.0f);
    for (int j = 0; j < 3; j++) {
      F d = q[j] - rp.jp[j], dv = qd[j] - rp.jv[j];
      ejp = ejp + d * d;
      ejv = ejv + dv * dv;
    }
    V3l df = fd - fk;
    float e_jp = L::qsum(ejp), e_jv = L::qsum(ejv), e_ee = L::qsum(dot(df, df));
    V3u dpv = bs.p - gb.p, dvv = bs.v - gb.v, dwv = bs.w - gb.w;
    float e_p = dot(dpv, dpv), e_v = dot(dvv, dvv), e_w = dot(dwv, dwv);
    V3u aa;
    float angle = axis_angle_scaled(qmul(gqn, qconj(qn)), &aa);                                  // PLE:410-411
    float reward = P.rw[0] * expf(-1.0f * e_jp) + P.rw[1] * expf(-0.1f * e_jv) + P.rw[2] * expf(-40.0f * e_ee) +
                   P.rw[3] * expf(-20.0f * e_p - 10.0f * angle * angle) + P.rw[4] * expf(-2.0f * e_v - 0.2f * e_w);   // PLE:386-425
    if (bad) reward = 0.0f;

    PMC_TS(3);
    PMC_PHASE("tail.termination");
    // --- termination (PLE:337-348) ---
    int reason = 0;
    {
      float left_z = R.m[2] * R.m[3] - R.m[5] * R.m[0];                       // up.x*fwd.y - up.y*fwd.x   LR:171-172
      if (left_z > 0.70710678118654752f || left_z < -0.70710678118654752f) reason |= LLS_DONE_FALL;
      if (R.m[8] < 0.5f) reason |= LLS_DONE_FALL;                             // cos(60 deg)  LR:176
      if (fid >= clen - P.margin - 1) reason |= LLS_DONE_CLIP_END;   // ML:168-172
      if (fabsf(angle) > 1.0f || e_p > 1.0f) reason |= LLS_DONE_DIVERGED;     // PLE:319-335
      if (bad) reason |= LLS_DONE_NONFINITE;
      if (P.set_obstacle && !bad) {
        PMC_PHASE("tail.obstacle_check");
        LegKin kf = leg_fk(ln, P.legc, q[0], q[1], q[2]);
        if (obstacle_contact(ln, P, env, clip, t, bs.p, R, kf)) reason |= LLS_DONE_COLLISION;   // PLE:341-346
      }
    }
    PMC_PHASE("tail.termination_end");
    const float rsum = rsum0 + reward;                                        // PLE:231

    PMC_TS(4);
    PMC_PHASE("tail.unroll_row");
    // --- this transition's row of the env's unroll (distill_actor.py:118-162): the observation the policy acted on
    //     in the learner's flatten order (future first: dict keys sorted), its action, neglogp and value as the policy reported
    //     them, the reward and the not-done mask; R is filled in by ll_finish_unroll.  Written before the obs row is replaced. ---
    if (P.traj) {
      const int W = P.obs_dim + LL_UNROLL_EXTRA, od = P.obs_dim;
      int tslot = P.traj_slot + sl, tbuf = P.traj_buf;                        // the launch's first step writes (traj_buf, traj_slot)
      while (tslot >= P.traj_unroll) { tslot -= P.traj_unroll; tbuf = (tbuf + 1 == P.traj_nbuf) ? 0 : tbuf + 1; }
      float* tr = P.traj + ((((long)tbuf * N + env) * P.traj_unroll) + tslot) * W;
      LL_UNROLL
      for (int c = 0; c < TRAJ_CHUNKS; c++) ln.st16_rot(tr, 16 * c, od, od - 72, told[c]);
      for (int j = 0; j < 3; j++) ln.stl(tr, od + j, 3, act[j]);
      tr[od + 12] = t_neglogp;
      tr[od + 13] = 0.0f;
      tr[od + 14] = t_value;
      tr[od + 15] = reward;
      tr[od + 16] = reason ? 0.0f : 1.0f;
    }

    PMC_TS(5);
    PMC_PHASE("tail.episode_bookkeeping");
    // --- end of episode (PLE:235-240): publish the per-clip statistics; the last workgroup of the kernel folds them into the
    //     table (highest env index wins when several envs finish the same clip in one step == sequential overwrite order) ---
    int steps_out = steps, clip_out = clip;
    float rsum_out = rsum;
    bool fill = false;
    F oq[3] = {q[0], q[1], q[2]}, oqd[3] = {qd[0], qd[1], qd[2]}, oact[3] = {act[0], act[1], act[2]};
    F gjp[3] = {rp.jp[0], rp.jp[1], rp.jp[2]}, gjv[3] = {rp.jv[0], rp.jv[1], rp.jv[2]};
    if (reason) {
      PMC_PHASE("tail.episode_end_reseed");
      float avg_r = (float)((double)rsum / max_steps), avg_l = (float)((double)steps / (max_steps + 1.0));
      if (bad) avg_r = 0.0f;
      // (every control step of a launch has its own slots -- folded by the last wave to finish that step, in step order: the later step wins, then
      // the higher env, the order in which one actor would have seen the episodes end)
      unsigned long long tag = ((unsigned long long)(unsigned)(env + 1)) << 32;
      publish_max(ln, P.pending_reward + (long)sl * P.n_clips + clip, tag | (unsigned long long)f2u(avg_r));
      publish_max(ln, P.pending_len + (long)sl * P.n_clips + clip, tag | (unsigned long long)f2u(avg_l));
      count_add(ln, P.counters + 1);
      if (bad) count_add(ln, P.counters + 2);
      {
        int b = 0;
        for (int s2 = steps; s2 > 1 && b < 15; s2 >>= 1) b++;                 // floor(log2(steps)), capped
        count_add(ln, P.ep_hist + b);
      }
      if (ar) {
        // auto-reset inside the step (PLE:150-171 + ML:48-63): the env continues from a freshly sampled (clip, t0); only what
        // differs from a running env is done here -- the pose and the four future sites are re-read at the new place, and
        // the common tail below writes the one observation, state and ghost of the step
        if (P.keep_term_obs) obs_emit(ln, P, P.term_obs + (long)env * P.obs_dim, false, oin, bs, R, q, qd, act);   // PLE:227
        int nclip;
        double nt0;
        sample_start(ln, P, env, ep0 + 1, &nclip, &nt0, table_for_reseed(ln, P, sl));
        const int fid2 = (int)floor(nt0 / P.frame_step);                      // ML:52
        const double frac2 = (nt0 - fid2 * P.frame_step) / P.frame_step;      // ML:53
        const double* rows2 = P.frames + (long)P.clip_off[nclip] * 19;
        RefRaw rr2 = mocap_gather(ln, rows2 + (long)fid2 * 19, rows2 + (long)(fid2 + 1) * 19, frac2, P.frame_step);
        obs_gather_futures(ln, P, oin, rows2, fid2, frac2);
        RefPose rp2 = mocap_finish(rr2, P.frame_step, true);
        bs.p = rp2.p; bs.q = rp2.q; bs.v = rp2.v; bs.w = rp2.w;               // PLE:162-163 dynamic robot := ghost := mocap
        gb = bs;
        R = qmat(qnormalize(bs.q));
        for (int j = 0; j < 3; j++) {
          oq[j] = rp2.jp[j]; oqd[j] = rp2.jv[j]; gjp[j] = rp2.jp[j]; gjv[j] = rp2.jv[j];
          oact[j] = ln.lane_f(0.0f);
        }
        fd = foot_world(ln, P.legc, bs.p, R, oq[0], oq[1], oq[2]);
        fk = fd;
        fill = true;                                                          // PLE:168-170
        t = nt0; steps_out = 0; rsum_out = 0.0f; clip_out = nclip;
        P.ep_count[env] = ep0 + 1;
        if (P.set_obstacle) P.ob_id[env] = 0;                                 // PLE:179
      }
    }
    PMC_TS(6);
    PMC_PHASE("tail.obs_row");
    // --- observation (PLE:227), state, ghost, feet, bookkeeping ---
    obs_emit(ln, P, row, fill, oin, bs, R, oq, oqd, oact);
    PMC_TS(8);
    PMC_PHASE("tail.state_stores");
    store_state(ln, P.state, N, env, bs, oq, oqd);
    store_state(ln, P.kin, N, env, gb, gjp, gjv);
    PMC_TS(9);
    for (int c = 0; c < 3; c++) {
      ln.stl(P.feet, (long)c * N + env, 3L * N, (c == 0) ? fd.x : (c == 1 ? fd.y : fd.z));
      ln.stl(P.feet, (long)(12 + c) * N + env, 3L * N, (c == 0) ? fk.x : (c == 1 ? fk.y : fk.z));
    }
    P.time[env] = t;
    P.clip[env] = clip_out;
    P.ep_steps[env] = steps_out;
    P.reward_sum[env] = rsum_out;
    P.reward[env] = reward;
    P.done[env] = reason ? 1 : 0;
    P.done_reason[env] = (uint8_t)reason;
    PMC_TS(7);
  }

  // The sampling table an episode that re-seeds in control step `sl` of the running launch draws from: the table as steps 0 .. sl - 1 of the launch
  // left it (PLE:235-240; what k single launches give).  Step 0 reads `cdf` (the previous launch's last fold); later steps read the version the
  // last wave to finish step sl - 1 has written (llenv.hip), WAITING for it if it is not there yet -- which takes a wave that has run a whole
  // control step ahead of the slowest one.  Waiting is safe exactly when every wave of the launch has started (then each is running or done and the
  // chain of folds ends); while some have not -- another kernel holds their SIMDs -- the episode takes the newest version there is and is
  // counted in counters[3] (ll_get_table_sync: zero in every run that has the chip to itself).
  static LL_HD const double* table_for_reseed(const L& ln, const StepParams& P, int sl) {
    if (sl == 0 || !P.table_versions) return P.cdf;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned int serial = P.launch_serial;
    int j = sl - 1;
    if (__hip_atomic_load(P.resident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) {
      while (__hip_atomic_load(P.ver_ready + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != serial) __builtin_amdgcn_s_sleep(4);
    } else {
      while (j >= 0 && __hip_atomic_load(P.ver_ready + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != serial) j--;
      if (j != sl - 1) count_add(ln, P.counters + 3);
    }
    LL_VER_FENCE(__ATOMIC_ACQUIRE);
    asm volatile("" ::: "memory");                 // the version is read after its mark (device-scope loads, issued in order; and an acquire fence at agent scope: LL_VER_FENCE)
    return j < 0 ? P.cdf : P.cdf_ver + (long)(j + 1) * P.n_clips;
#else
    (void)ln;
    return P.cdf_ver + (long)sl * P.n_clips;       // (the CPU build runs a launch step-major and folds after every step: emul.cpp)
#endif
  }

  static LL_HD uint32_t f2u(float x) {
    union { float f; uint32_t u; } c;
    c.f = x;
    return c.u;
  }
  static LL_HD void count_add(const L& ln, unsigned long long* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ln.is_lane(0)) atomicAdd(p, 1ull);
#else
    *p += 1ull;
#endif
  }
  static LL_HD void publish_max(const L& ln, unsigned long long* p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ln.is_lane(0)) atomicMax(p, v);
#else
    if (*p < v) *p = v;
#endif
  }
};
