/*
 * pmc_oracle.c -- CPU float64 restatement of the reference's PMC tracking-env hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (lifelike_agility_and_play_amd/, the C ABI, bench's timed
 * GPU leg) may import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / reported baseline.
 *
 * What it restates (file:line under /root/reference/src/lifelike/sim_envs/pybullet_envs/):
 *   PLE = primitive_level_env/primitive_level_env.py, ML = primitive_level_env/motion_lib.py,
 *   LR  = legged_robot/legged_robot.py
 * Everything outside PyBullet follows the reference line by line and is PINNED against golden vectors
 * produced by importing the reference (tests/golden/gen_golden.py -> pmc_golden.npz).
 *
 * Physics (what `stepSimulation()` does at PLE:206) lives in the third-party `pybullet` wheel, which
 * is un-pinned (setup.py:20), absent from /root/reference and not installable here:
 *     >>> PARITY UNPINNED for the physics substep <<<
 * It is restated from Bullet's published algorithm (btMultiBody: Featherstone articulated-body
 * forward dynamics, semi-implicit Euler, sequential-impulse (PGS) contact + joint-limit solve with
 * ERP) -- see DESIGN.md "physics spec".  The oracle deliberately uses the textbook generic
 * formulation (dense 6x6 spatial algebra, generic tree ABA, explicit Jacobians, explicit
 * M^-1 by unit responses, lambda-space PGS) so that it is an independent check of the
 * structure-exploiting HIP kernel.
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/llenv.h"
#include "../include/llenv_model.h"

#define NB 13
#define NDOF 18
#define KC LLM_MAX_CONTACTS_PER_LEG
#define MAXC (4 * KC)
#define MAXROWS (12 + 3 * MAXC + 3 * LLM_MAX_SELF)   /* limits + contacts (n, t1, t2) + self-collision rows (n, and t1, t2 when LLM_SPEC_SELF_FRICTION > 0) */

/* ------------------------------------------------------------------------------------------------ */
/* small linear algebra                                                                             */
/* ------------------------------------------------------------------------------------------------ */
static void v3cross(const double* a, const double* b, double* o) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static double v3dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3v(const double* m, const double* v, double* o) { /* o = M v (row-major) */
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m3tv(const double* m, const double* v, double* o) { /* o = M^T v */
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m3m(const double* a, const double* b, double* o) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(o, t, sizeof t);
}

/* ---- quaternions, xyzw, following scipy.spatial.transform.Rotation semantics --------------------- */
static void q_normalize(const double* q, double* o) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) o[i] = q[i] / n;
}
static void q_mul(const double* p, const double* q, double* o) { /* Hamilton product p (x) q  == scipy `P * Q` */
  double x = p[3] * q[0] + p[0] * q[3] + p[1] * q[2] - p[2] * q[1];
  double y = p[3] * q[1] - p[0] * q[2] + p[1] * q[3] + p[2] * q[0];
  double z = p[3] * q[2] + p[0] * q[1] - p[1] * q[0] + p[2] * q[3];
  double w = p[3] * q[3] - p[0] * q[0] - p[1] * q[1] - p[2] * q[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void q_conj(const double* q, double* o) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
static void q_to_mat(const double* q, double* m) { /* unit q -> R (row-major), scipy as_matrix */
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w, xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
  m[0] = x2 - y2 - z2 + w2; m[1] = 2 * (xy - zw); m[2] = 2 * (xz + yw);
  m[3] = 2 * (xy + zw); m[4] = -x2 + y2 - z2 + w2; m[5] = 2 * (yz - xw);
  m[6] = 2 * (xz - yw); m[7] = 2 * (yz + xw); m[8] = -x2 - y2 + z2 + w2;
}
static void q_as_rotvec(const double* qin, double* rv) { /* scipy as_rotvec: shortest arc, series below 1e-3 */
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  double angle = 2.0 * atan2(nrm, q[3]);
  double scale;
  if (angle <= 1e-3) {
    double a2 = angle * angle;
    scale = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
  } else {
    scale = angle / sin(angle / 2.0);
  }
  rv[0] = scale * q[0]; rv[1] = scale * q[1]; rv[2] = scale * q[2];
}
static void q_from_rotvec(const double* rv, double* q) { /* scipy from_rotvec */
  double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  double scale;
  if (angle <= 1e-3) {
    double a2 = angle * angle;
    scale = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
  } else {
    scale = sin(angle / 2.0) / angle;
  }
  q[0] = scale * rv[0]; q[1] = scale * rv[1]; q[2] = scale * rv[2]; q[3] = cos(angle / 2.0);
}
/* PLE:19-23 quat2axisangle followed by axis*angle */
static double quat_axis_angle(const double* quat, double* axis) {
  double qn[4], rv[3];
  q_normalize(quat, qn);
  q_as_rotvec(qn, rv);
  double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  for (int i = 0; i < 3; i++) axis[i] = rv[i] / (angle + 1e-8);
  return angle;
}

/* ------------------------------------------------------------------------------------------------ */
/* mocap (ML:65-172)                                                                                */
/* ------------------------------------------------------------------------------------------------ */
/* ML:65-67 */
void orc_mocap_locate(double t, double frame_step, int32_t* frame_id, double* frac) {
  int32_t f = (int32_t)floor(t / frame_step);
  *frame_id = f;
  *frac = (t - f * frame_step) / frame_step;
}

/* ML:88-166 _get_states_info_by_interpolation (interpolation=True, free_joint=True) -> state37 */
void orc_mocap_interp(const double* fc, const double* fn, double frac, double frame_step, double* out) {
  /* base_pos_interpolation ML:118-124 */
  for (int i = 0; i < 3; i++) out[i] = fc[i] + frac * (fn[i] - fc[i]);
  /* base_orn_interpolation ML:127-134 : scipy Slerp([0,1], [qc,qn])(frac) */
  double qc[4], qn[4], qci[4], d[4], rv[3], dq[4];
  q_normalize(fc + 3, qc);
  q_normalize(fn + 3, qn);
  q_conj(qc, qci);
  q_mul(qci, qn, d);
  q_as_rotvec(d, rv);
  rv[0] *= frac; rv[1] *= frac; rv[2] *= frac;
  q_from_rotvec(rv, dq);
  q_mul(qc, dq, out + 3);
  /* base_lin_vel_interpolation ML:137-140 */
  for (int i = 0; i < 3; i++) out[7 + i] = (fn[i] - fc[i]) / frame_step;
  /* base_ang_vel_interpolation ML:143-149 : rotvec(qn * qc^-1), axis*angle/dt with the 1e-8 renorm */
  double e[4], rv2[3];
  q_mul(qn, qci, e);
  q_as_rotvec(e, rv2);
  double angle = sqrt(rv2[0] * rv2[0] + rv2[1] * rv2[1] + rv2[2] * rv2[2]);
  for (int i = 0; i < 3; i++) out[10 + i] = ((rv2[i] / (angle + 1e-8)) * angle) / frame_step;
  /* joint_interpolation ML:152-160 */
  for (int i = 0; i < 12; i++) {
    out[13 + i] = fc[7 + i] + frac * (fn[7 + i] - fc[7 + i]);
    out[25 + i] = (fn[7 + i] - fc[7 + i]) / frame_step;
  }
}

/* ML:75-86 get_states_info_future; frames points at row `frame_id` of the clip. out: 4 x state37 */
void orc_mocap_future(const double* frames_at_fid, double frac, double frame_step, double* out) {
  const double time_future[4] = {1. / 30., 1. / 15., 1. / 3., 1.}; /* ML:44 */
  for (int i = 0; i < 4; i++) {
    double t = frame_step * frac + time_future[i];
    int fid = (int)floor(t / frame_step);
    double ff = t / frame_step - fid;
    orc_mocap_interp(frames_at_fid + (size_t)fid * 19, frames_at_fid + (size_t)(fid + 1) * 19, ff, frame_step, out + 37 * i);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* observation (PLE:247-317)                                                                        */
/* ------------------------------------------------------------------------------------------------ */
/* PLE:247-260: full prop in the order given by prop_order (LL_PROP_* ids, -1 terminated); returns frame size */
int orc_prop(const double* s, const int32_t* prop_order, double* out) {
  double qn[4], R[9];
  q_normalize(s + 3, qn);
  q_to_mat(qn, R);
  int n = 0;
  for (int k = 0; k < 5 && prop_order[k] >= 0; k++) {
    switch (prop_order[k]) {
      case LL_PROP_JOINT_POS: memcpy(out + n, s + 13, 12 * sizeof(double)); n += 12; break;   /* PLE:249 */
      case LL_PROP_JOINT_VEL: memcpy(out + n, s + 25, 12 * sizeof(double)); n += 12; break;   /* PLE:250 */
      case LL_PROP_ROOT_LIN_VEL_LOC: m3tv(R, s + 7, out + n); n += 3; break;                   /* PLE:251 R^-1 v */
      case LL_PROP_ROOT_ANG_VEL_LOC: m3tv(R, s + 10, out + n); n += 3; break;                  /* PLE:252 R^-1 w */
      case LL_PROP_E_G: out[n] = R[6]; out[n + 1] = R[7]; out[n + 2] = R[8]; n += 3; break;   /* PLE:253 R[2,:] */
      default: return -1;
    }
  }
  return n;
}

/* PLE:299-317 calculate_future. fut: 4 x (pos3, quat4, joint_pos12) = 76 doubles -> out 72 */
void orc_calc_future(const double* base_pos, const double* base_orn, const double* fut, double* out) {
  double qb[4], qbi[4], Rb[9];
  q_normalize(base_orn, qb);
  q_conj(qb, qbi);
  q_to_mat(qb, Rb);
  for (int i = 0; i < 4; i++) {
    const double* f = fut + 19 * i;
    double qi[4], d[4], axis[3], pd[3];
    q_normalize(f + 3, qi);
    q_mul(qbi, qi, d);                                  /* PLE:307 */
    double angle = quat_axis_angle(d, axis);           /* PLE:308 */
    for (int k = 0; k < 3; k++) pd[k] = f[k] - base_pos[k];   /* PLE:310 */
    m3tv(Rb, pd, out + 18 * i);                         /* PLE:311 r_b.inv().apply */
    for (int k = 0; k < 3; k++) out[18 * i + 3 + k] = axis[k] * angle;   /* PLE:313 */
    memcpy(out + 18 * i + 6, f + 7, 12 * sizeof(double));                 /* PLE:315 */
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* reward + termination (PLE:319-426, LR:158-179)                                                   */
/* ------------------------------------------------------------------------------------------------ */
static double base_angle_between(const double* q_dyn, const double* q_kin) { /* PLE:327-328, :410-411 */
  double a[4], b[4], bi[4], d[4], axis[3];
  q_normalize(q_kin, a);
  q_normalize(q_dyn, b);
  q_conj(b, bi);
  q_mul(a, bi, d);
  return quat_axis_angle(d, axis);
}

double orc_reward(const double* dyn, const double* kin, const double* feet_dyn, const double* feet_kin, const double* w_in) {
  double w[5], sw = 0;
  for (int i = 0; i < 5; i++) sw += w_in[i];             /* PLE:365 */
  for (int i = 0; i < 5; i++) w[i] = w_in[i] / sw;       /* PLE:366-370 */
  double e_jp = 0, e_jv = 0, e_ee = 0, e_p = 0, e_v = 0, e_w = 0;
  for (int i = 0; i < 12; i++) {
    double d = dyn[13 + i] - kin[13 + i]; e_jp += d * d;             /* PLE:384-386 */
    double dv = dyn[25 + i] - kin[25 + i]; e_jv += dv * dv;          /* PLE:389-391 */
    double de = feet_dyn[i] - feet_kin[i]; e_ee += de * de;          /* PLE:402 */
  }
  for (int i = 0; i < 3; i++) {
    double d = dyn[i] - kin[i]; e_p += d * d;                         /* PLE:413 */
    double dv = dyn[7 + i] - kin[7 + i]; e_v += dv * dv;              /* PLE:418 */
    double dw = dyn[10 + i] - kin[10 + i]; e_w += dw * dw;            /* PLE:419 */
  }
  double angle = base_angle_between(dyn + 3, kin + 3);
  double r_jp = exp(-1.0 * e_jp), r_jv = exp(-0.1 * e_jv), r_ee = exp(-40.0 * e_ee);           /* PLE:373-375 */
  double r_pose = exp(-20.0 * e_p + -10.0 * (angle * angle));                                     /* PLE:413-414 */
  double r_vel = exp(-2 * e_v + -0.2 * e_w);                                                      /* PLE:418-419 */
  return w[0] * r_jp + w[1] * r_jv + w[2] * r_ee + w[3] * r_pose + w[4] * r_vel;                  /* PLE:421-425 */
}

int orc_check_fall(const double* quat) { /* LR:158-179 */
  double qn[4], R[9];
  q_normalize(quat, qn);
  q_to_mat(qn, R);
  double fwd[3] = {R[0], R[3], R[6]}, up[3] = {R[2], R[5], R[8]};
  double left_z = up[0] * fwd[1] - up[1] * fwd[0];
  int term = 0;
  if (left_z > sin(45.0 * M_PI / 180.0) || left_z < sin(-45.0 * M_PI / 180.0)) term = 1;
  if (up[2] < cos(60.0 * M_PI / 180.0)) term = 1;
  return term;
}

int orc_check_diverged(const double* dyn, const double* kin) { /* PLE:319-335 */
  int term = 0;
  double angle = base_angle_between(dyn + 3, kin + 3);
  if (fabs(angle) > 1.0) term = 1;
  double e = 0;
  for (int i = 0; i < 3; i++) e += (dyn[i] - kin[i]) * (dyn[i] - kin[i]);
  if (e > 1.0) term = 1;
  return term;
}

/* ------------------------------------------------------------------------------------------------ */
/* robot model                                                                                      */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  int type;
  double size[3], pos[3], rot[9];
} OPrim;

typedef struct {
  int parent[NB];
  double r[NB][3], axis[NB][3], mass[NB], com[NB][3], Ic[NB][9];
  double I6[NB][36]; /* spatial inertia about the body origin, body axes */
  double qlo[12], qhi[12], damp[12], foot[4][3];
  OPrim base_prims[LLM_N_BASE_PRIMS], leg_prims[4][LLM_N_LEG_PRIMS];
} OModel;

static void skew(const double* v, double* m) {
  m[0] = 0; m[1] = -v[2]; m[2] = v[1]; m[3] = v[2]; m[4] = 0; m[5] = -v[0]; m[6] = -v[1]; m[7] = v[0]; m[8] = 0;
}

static void rigid_inertia(double m, const double* c, const double* Ic, double* I6) {
  /* [[Ic + m cx cx^T, m cx], [m cx^T, m 1]]  (Featherstone RBDA eq. 2.63) */
  double cx[9], cxcxT[9];
  skew(c, cx);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += cx[3 * i + k] * cx[3 * j + k];
      cxcxT[3 * i + j] = s;
    }
  memset(I6, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      I6[6 * i + j] = Ic[3 * i + j] + m * cxcxT[3 * i + j];
      I6[6 * i + 3 + j] = m * cx[3 * i + j];
      I6[6 * (3 + i) + j] = m * cx[3 * j + i];
    }
  for (int i = 0; i < 3; i++) I6[6 * (3 + i) + 3 + i] = m;
}

static void read_prim(const double* b, OPrim* p) {
  p->type = (int)b[0];
  memcpy(p->size, b + 1, 24); memcpy(p->pos, b + 4, 24); memcpy(p->rot, b + 7, 72);
}

static void model_from_blob(const double* b, OModel* M) {
  memset(M, 0, sizeof *M);
  M->parent[0] = -1;
  M->mass[0] = b[LLM_OFF_BASE_MASS];
  memcpy(M->com[0], b + LLM_OFF_BASE_COM, 24);
  memcpy(M->Ic[0], b + LLM_OFF_BASE_INERTIA, 72);
  for (int i = 0; i < 12; i++) {
    int bi = i + 1;
    M->parent[bi] = (i % 3 == 0) ? 0 : bi - 1;
    memcpy(M->r[bi], b + LLM_OFF_JOINT_ORIGIN + 3 * i, 24);
    memcpy(M->axis[bi], b + LLM_OFF_JOINT_AXIS + 3 * i, 24);
    M->mass[bi] = b[LLM_OFF_LINK_MASS + i];
    memcpy(M->com[bi], b + LLM_OFF_LINK_COM + 3 * i, 24);
    memcpy(M->Ic[bi], b + LLM_OFF_LINK_INERTIA + 9 * i, 72);
    M->qlo[i] = b[LLM_OFF_Q_LO + i]; M->qhi[i] = b[LLM_OFF_Q_HI + i]; M->damp[i] = b[LLM_OFF_DAMPING + i];
  }
  for (int i = 0; i < NB; i++) rigid_inertia(M->mass[i], M->com[i], M->Ic[i], M->I6[i]);
  memcpy(M->foot, b + LLM_OFF_FOOT_POS, 96);
  for (int i = 0; i < LLM_N_BASE_PRIMS; i++) read_prim(b + LLM_OFF_BASE_PRIMS + i * LLM_PRIM_STRIDE, &M->base_prims[i]);
  for (int l = 0; l < 4; l++)
    for (int i = 0; i < LLM_N_LEG_PRIMS; i++)
      read_prim(b + LLM_OFF_LEG_PRIMS + (l * LLM_N_LEG_PRIMS + i) * LLM_PRIM_STRIDE, &M->leg_prims[l][i]);
}

/* ------------------------------------------------------------------------------------------------ */
/* kinematics                                                                                       */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  double E[NB][9];   /* coordinate rotation parent -> body  ( = (rotation of body in parent)^T ) */
  double Rw[NB][9];  /* body -> world rotation */
  double pw[NB][3];  /* body origin in world */
  double v[NB][6];   /* spatial velocity, body coordinates [ang; lin] */
} OKin;

static void axis_angle_mat(const double* a, double q, double* m) { /* Rodrigues, rotation by q about unit a */
  double c = cos(q), s = sin(q), t = 1 - c;
  m[0] = t * a[0] * a[0] + c; m[1] = t * a[0] * a[1] - s * a[2]; m[2] = t * a[0] * a[2] + s * a[1];
  m[3] = t * a[0] * a[1] + s * a[2]; m[4] = t * a[1] * a[1] + c; m[5] = t * a[1] * a[2] - s * a[0];
  m[6] = t * a[0] * a[2] - s * a[1]; m[7] = t * a[1] * a[2] + s * a[0]; m[8] = t * a[2] * a[2] + c;
}

/* motion transform parent->child: v_c = [E w ; E (v + w x r)] */
static void xform_motion(const double* E, const double* r, const double* vp, double* vc) {
  double t[3], u[3];
  v3cross(vp, r, t);
  for (int i = 0; i < 3; i++) u[i] = vp[3 + i] + t[i];
  m3v(E, vp, vc);
  m3v(E, u, vc + 3);
}
/* force transform child->parent: f_p = [E^T n + r x (E^T f) ; E^T f] */
static void xform_force_T(const double* E, const double* r, const double* fc, double* fp) {
  double n[3], f[3], t[3];
  m3tv(E, fc, n);
  m3tv(E, fc + 3, f);
  v3cross(r, f, t);
  for (int i = 0; i < 3; i++) { fp[i] = n[i] + t[i]; fp[3 + i] = f[i]; }
}
static void xform_matrix(const double* E, const double* r, double* X) { /* 6x6 [E 0; -E rx, E] */
  double rx[9], Erx[9];
  skew(r, rx);
  m3m(E, rx, Erx);
  memset(X, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      X[6 * i + j] = E[3 * i + j];
      X[6 * (3 + i) + 3 + j] = E[3 * i + j];
      X[6 * (3 + i) + j] = -Erx[3 * i + j];
    }
}

/* positions + velocities of every body.  state = LR:86-106 layout; nu (optional) overrides velocities
 * with generalized body-frame velocity [w_b, v_b, qd] */
static void kinematics(const OModel* M, const double* state, const double* nu, OKin* K) {
  double qn[4];
  q_normalize(state + 3, qn);
  q_to_mat(qn, K->Rw[0]);
  memcpy(K->pw[0], state, 24);
  for (int i = 0; i < 9; i++) K->E[0][i] = (i % 4 == 0);
  if (nu) {
    memcpy(K->v[0], nu, 48);
  } else {
    m3tv(K->Rw[0], state + 10, K->v[0]);
    m3tv(K->Rw[0], state + 7, K->v[0] + 3);
  }
  for (int b = 1; b < NB; b++) {
    int p = M->parent[b];
    double q = state[13 + b - 1], qd = nu ? nu[6 + b - 1] : state[25 + b - 1];
    double Rj[9], t[3];
    axis_angle_mat(M->axis[b], q, Rj);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) K->E[b][3 * i + j] = Rj[3 * j + i];
    m3m(K->Rw[p], Rj, K->Rw[b]);
    m3v(K->Rw[p], M->r[b], t);
    for (int i = 0; i < 3; i++) K->pw[b][i] = K->pw[p][i] + t[i];
    xform_motion(K->E[b], M->r[b], K->v[p], K->v[b]);
    for (int i = 0; i < 3; i++) K->v[b][i] += M->axis[b][i] * qd;
  }
}

/* LR:199-205 compute_end_effector_info: world positions of the 4 foot links */
void orc_fk_feet_model(const OModel* M, const double* state, double* feet) {
  OKin K;
  kinematics(M, state, NULL, &K);
  for (int l = 0; l < 4; l++) {
    int b = 3 + 3 * l;
    double t[3];
    m3v(K.Rw[b], M->foot[l], t);
    for (int i = 0; i < 3; i++) feet[3 * l + i] = K.pw[b][i] + t[i];
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* articulated-body algorithm (Featherstone, RBDA table 7.1 + floating base 9.4), body coordinates   */
/* ------------------------------------------------------------------------------------------------ */
static void cross_motion(const double* v, const double* m, double* o) { /* v x m */
  double a[3], b[3], c[3];
  v3cross(v, m, a);
  v3cross(v, m + 3, b);
  v3cross(v + 3, m, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
static void cross_force(const double* v, const double* f, double* o) { /* v x* f */
  double a[3], b[3], c[3];
  v3cross(v, f, a);
  v3cross(v + 3, f + 3, b);
  v3cross(v, f + 3, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
static void m6v(const double* m, const double* v, double* o) {
  double t[6];
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int j = 0; j < 6; j++) s += m[6 * i + j] * v[j];
    t[i] = s;
  }
  memcpy(o, t, sizeof t);
}
static int solve6(const double* A, const double* b, double* x) { /* SPD solve by Cholesky */
  double L[36];
  memset(L, 0, sizeof L);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0)) return -1;
        L[6 * i + i] = sqrt(s);
      } else {
        L[6 * i + j] = s / L[6 * j + j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k];
    y[i] = s / L[6 * i + i];
  }
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k];
    x[i] = s / L[6 * i + i];
  }
  return 0;
}

/* fext[b] = external spatial force acting ON body b, body coordinates (NULL = none).
 * out: a0 = spatial acceleration of the base (body coords), qdd[12].  with_bias=0 drops velocity terms. */
static double g_spec[LLM_SPEC_COUNT];   /* (defined with its defaults below) */
static int aba(const OModel* M, const OKin* K, const double* qd, const double* tau, const double (*fext)[6],
               int with_bias, double* a0, double* qdd) {
  double IA[NB][36], pA[NB][6], c[NB][6], U[NB][6], D[NB], u[NB];
  for (int b = 0; b < NB; b++) {
    memcpy(IA[b], M->I6[b], sizeof IA[b]);
    double Iv[6];
    if (with_bias) {
      m6v(M->I6[b], K->v[b], Iv);
      cross_force(K->v[b], Iv, pA[b]);
      if (g_spec[LLM_SPEC_GYRO] < 0.5) {       /* audit switch: btMultiBody::setUseGyroTerm(false) drops w x (I_c w), a pure couple */
        double Iw[3], g[3];
        m3v(M->Ic[b], K->v[b], Iw);
        v3cross(K->v[b], Iw, g);
        for (int i = 0; i < 3; i++) pA[b][i] -= g[i];
      }
    } else {
      memset(pA[b], 0, sizeof pA[b]);
    }
    if (fext)
      for (int i = 0; i < 6; i++) pA[b][i] -= fext[b][i];
    memset(c[b], 0, sizeof c[b]);
    if (b > 0 && with_bias) {
      double Sqd[6] = {M->axis[b][0] * qd[b - 1], M->axis[b][1] * qd[b - 1], M->axis[b][2] * qd[b - 1], 0, 0, 0};
      cross_motion(K->v[b], Sqd, c[b]);
    }
  }
  for (int b = NB - 1; b >= 1; b--) {
    int p = M->parent[b];
    double S[6] = {M->axis[b][0], M->axis[b][1], M->axis[b][2], 0, 0, 0};
    m6v(IA[b], S, U[b]);
    D[b] = 0;
    for (int i = 0; i < 6; i++) D[b] += S[i] * U[b][i];
    double sp = 0;
    for (int i = 0; i < 6; i++) sp += S[i] * pA[b][i];
    u[b] = tau[b - 1] - sp;
    double Ia[36], pa[6], Iac[6];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) Ia[6 * i + j] = IA[b][6 * i + j] - U[b][i] * U[b][j] / D[b];
    m6v(Ia, c[b], Iac);
    for (int i = 0; i < 6; i++) pa[i] = pA[b][i] + Iac[i] + U[b][i] * u[b] / D[b];
    double X[36], XtIa[36], fp[6];
    xform_matrix(K->E[b], M->r[b], X);
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += X[6 * k + i] * Ia[6 * k + j];
        XtIa[6 * i + j] = s;
      }
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += XtIa[6 * i + k] * X[6 * k + j];
        IA[p][6 * i + j] += s;
      }
    xform_force_T(K->E[b], M->r[b], pa, fp);
    for (int i = 0; i < 6; i++) pA[p][i] += fp[i];
  }
  double rhs[6], a[NB][6];
  for (int i = 0; i < 6; i++) rhs[i] = -pA[0][i];
  if (solve6(IA[0], rhs, a[0])) return -1;
  for (int b = 1; b < NB; b++) {
    int p = M->parent[b];
    double ap[6];
    xform_motion(K->E[b], M->r[b], a[p], ap);
    for (int i = 0; i < 6; i++) ap[i] += c[b][i];
    double s = 0;
    for (int i = 0; i < 6; i++) s += U[b][i] * ap[i];
    qdd[b - 1] = (u[b] - s) / D[b];
    for (int i = 0; i < 6; i++) a[b][i] = ap[i];
    for (int i = 0; i < 3; i++) a[b][i] += M->axis[b][i] * qdd[b - 1];
  }
  memcpy(a0, a[0], 48);
  return 0;
}

/* gravity + Bullet's per-link velocity damping as external spatial forces (body coords) */
/* Spec overrides (include/llenv_model.h LLM_SPEC_*): process-wide, defaults = the constants of that header.  The engine has the same
 * switches (ll_set_spec_param); tools/deviation_table.py moves them in both to measure what each of this build's own choices is worth. */
static double g_spec[LLM_SPEC_COUNT] = {LLM_LIMIT_GATE, LLM_MAX_DEPEN_SPEED, LLM_LINK_DAMPING, LLM_MAX_CONTACTS_PER_LEG, 1.0, LLM_SELF_MARGIN,
                                        LLM_MAX_SELF, LLM_ERP, LLM_CONTACT_MARGIN, LLM_SELF_FRICTION, 0.0, 1.0, LLM_SELECT_EPS, LLM_FRICTION_MODE, 0.0, LLM_MAX_COORD_VEL, LLM_LIMIT_ERP, LLM_PAIR_FRICTION, LLM_MAX_PAIR, 0.0, LLM_LIMIT_SPECULATIVE, 1.0, 0.0, LLM_ERP_DEEP, LLM_ERP_DEEP_BELOW, LLM_LIMIT_ERP_DEEP, LLM_LEG_EDGES};
int orc_set_spec_param(int id, double v) {
  if (id < 0 || id >= LLM_SPEC_COUNT) return -1;
  if (id == LLM_SPEC_MAX_CONTACTS_PER_LEG && !(v >= 1 && v <= LLM_MAX_CONTACTS_PER_LEG)) return -1;
  if (id == LLM_SPEC_MAX_SELF && !(v >= 0 && v <= LLM_MAX_SELF)) return -1;
  if (id == LLM_SPEC_MAX_PAIR && !(v >= 0 && v <= 4)) return -1;
  if (id == LLM_SPEC_FRICTION_MODE && !(v == 0 || v == 1 || v == 2 || v == 3)) return -1;
  g_spec[id] = v;
  return 0;
}
double orc_get_spec_param(int id) { return (id >= 0 && id < LLM_SPEC_COUNT) ? g_spec[id] : NAN; }
void orc_reset_spec(void) {
  const double d[LLM_SPEC_COUNT] = {LLM_LIMIT_GATE, LLM_MAX_DEPEN_SPEED, LLM_LINK_DAMPING, LLM_MAX_CONTACTS_PER_LEG, 1.0, LLM_SELF_MARGIN,
                                    LLM_MAX_SELF, LLM_ERP, LLM_CONTACT_MARGIN, LLM_SELF_FRICTION, 0.0, 1.0, LLM_SELECT_EPS, LLM_FRICTION_MODE, 0.0, LLM_MAX_COORD_VEL, LLM_LIMIT_ERP, LLM_PAIR_FRICTION, LLM_MAX_PAIR, 0.0, LLM_LIMIT_SPECULATIVE, 1.0, 0.0, LLM_ERP_DEEP, LLM_ERP_DEEP_BELOW, LLM_LIMIT_ERP_DEEP, LLM_LEG_EDGES};
  memcpy(g_spec, d, sizeof d);
}
/* bias of a unilateral row from its signed distance: a separated row may close the gap within the substep; a penetrating one is pushed
 * out by ERP per substep -- with a second, deeper ERP when LLM_SPEC_ERP_DEEP is set, and (contact rows only) capped at LLM_SPEC_MAX_DEPEN_SPEED */
static double row_bias(double depth, double dt, double erp, int capped) {
  if (depth > 0) return depth / dt;
  /* (capped = a contact row; the others are joint-limit rows, whose deep ERP may be given separately) */
  const double deep = (!capped && g_spec[LLM_SPEC_LIMIT_ERP_DEEP] >= 0) ? g_spec[LLM_SPEC_LIMIT_ERP_DEEP] : g_spec[LLM_SPEC_ERP_DEEP];
  const double e = (deep >= 0 && !(depth > g_spec[LLM_SPEC_ERP_DEEP_BELOW])) ? deep : erp;
  const double b = e * depth / dt;
  return capped ? fmax(b, -g_spec[LLM_SPEC_MAX_DEPEN_SPEED]) : b;
}
#define g_link_damping (g_spec[LLM_SPEC_LINK_DAMPING])
#define g_self_collision (g_spec[LLM_SPEC_SELF_COLLISION] > 0.5)
void orc_set_self_collision(int on) { g_spec[LLM_SPEC_SELF_COLLISION] = on ? 1.0 : 0.0; } /* tests: compare with / without */
void orc_set_link_damping(double k) { g_spec[LLM_SPEC_LINK_DAMPING] = k; } /* tests: 0 makes free flight conservative */
static void external_forces(const OModel* M, const OKin* K, double (*fext)[6]) {
  const double k1 = g_link_damping, k2 = g_link_damping;
  for (int b = 0; b < NB; b++) {
    const double* w = K->v[b];
    double vc[3], t[3], f[3], n[3], Iw[3], gw[3] = {0, 0, -M->mass[b] * LLM_GRAVITY}, gb[3];
    v3cross(w, M->com[b], t);
    for (int i = 0; i < 3; i++) vc[i] = K->v[b][3 + i] + t[i];
    double sv = sqrt(v3dot(vc, vc)), sw = sqrt(v3dot(w, w));
    m3tv(K->Rw[b], gw, gb);
    m3v(M->Ic[b], w, Iw);
    for (int i = 0; i < 3; i++) {
      f[i] = gb[i] - M->mass[b] * vc[i] * (k1 + k2 * sv);
      n[i] = -Iw[i] * (k1 + k2 * sw);
    }
    v3cross(M->com[b], f, t);
    for (int i = 0; i < 3; i++) { fext[b][i] = n[i] + t[i]; fext[b][3 + i] = f[i]; }
  }
}

/* unconstrained forward dynamics: generalized acceleration [a0(6) body-coord spatial ; qdd(12)] */
int orc_forward_dynamics_model(const OModel* M, const double* state, const double* tau, double* acc18) {
  OKin K;
  double fext[NB][6];
  kinematics(M, state, NULL, &K);
  external_forces(M, &K, fext);
  return aba(M, &K, state + 25, tau, fext, 1, acc18, acc18 + 6);
}

/* ------------------------------------------------------------------------------------------------ */
/* contacts                                                                                         */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  int body;       /* body index the point is attached to */
  double P[3];    /* world position of the contact point on the robot surface */
  double depth;   /* signed distance to the plane z=0 (negative = penetrating) */
  double mu;
  int leg, slot;
  int cand;       /* index of the candidate within the leg's fixed enumeration (identity across substeps: the warm-start key) */
  double n[3];    /* unit normal of the surface it touches (plane: +z), world */
} OContact;

/* EPMC terrain (epmc_step.hpp): axis-aligned boxes  x0 x1 y0 y1 z0 z1 rod r ; rod = +1 / -1 / 0: thin cylinders of
 * radius r along y on the two top / bottom x-edges.  Signed distance and outward normal of point E; same spec as the kernel's
 * shape_sdf (face of least penetration inside, max(q) outside). */
typedef struct {
  int n; const double* rec; double box_mu_scale;
  /* yawed != 0 (the PMC jump obstacle, PLE:182-193): the records are given in a frame turned by yaw about z around (cx, cy) */
  int yawed; double cx, cy, cs, sn;
} OTerrain;
/* shape_sdf of record si at world point E for a terrain that may be yawed: distance, and the outward normal in WORLD coordinates */
static double shape_sdf(const double* s, const double* E, double* n, int* is_box);
static double terrain_sdf(const OTerrain* T, int si, const double* E, double* n, int* is_box) {
  if (!T->yawed) return shape_sdf(T->rec + 8 * si, E, n, is_box);
  const double dx = E[0] - T->cx, dy = E[1] - T->cy;
  const double El[3] = {dx * T->cs + dy * T->sn, dy * T->cs - dx * T->sn, E[2]};
  double nl[3];
  const double d = shape_sdf(T->rec + 8 * si, El, nl, is_box);
  n[0] = nl[0] * T->cs - nl[1] * T->sn; n[1] = nl[0] * T->sn + nl[1] * T->cs; n[2] = nl[2];
  return d;
}
static double shape_sdf(const double* s, const double* E, double* n, int* is_box) {
  double a0[3] = {s[0] - E[0], s[2] - E[1], s[4] - E[2]}, a1[3] = {E[0] - s[1], E[1] - s[3], E[2] - s[5]};
  double q[3], sg[3];
  for (int i = 0; i < 3; i++) { q[i] = a0[i] > a1[i] ? a0[i] : a1[i]; sg[i] = a1[i] > a0[i] ? 1.0 : -1.0; }
  double d = q[0]; int ax = 0;
  if (q[1] > d) { d = q[1]; ax = 1; }
  if (q[2] > d) { d = q[2]; ax = 2; }
  n[0] = n[1] = n[2] = 0; n[ax] = sg[ax];
  *is_box = 1;
  /* outside the box: the Euclidean distance to it and the direction away from its nearest point -- over a face the same as above,
   * diagonally outside an edge or a corner the rounded distance, so that a sphere or a link's mid-span meets an edge with the
   * normal pointing from the edge to its centre */
  {
    const double o0 = q[0] > 0 ? q[0] : 0, o1 = q[1] > 0 ? q[1] : 0, o2 = q[2] > 0 ? q[2] : 0, e2 = o0 * o0 + o1 * o1 + o2 * o2;
    if (e2 > 0.0) {
      const double e = sqrt(e2);
      d = e; n[0] = sg[0] * o0 / e; n[1] = sg[1] * o1 / e; n[2] = sg[2] * o2 / e;
    }
  }
  if (s[6] != 0.0) {
    double ze = s[6] > 0 ? s[5] : s[4];
    for (int e = 0; e < 2; e++) {
      double dx = E[0] - s[e], dz = E[2] - ze, len = sqrt(dx * dx + dz * dz), dr = len - s[7];
      if (q[1] <= 0.0 && dr < d && len > 1e-6) { d = dr; n[0] = dx / len; n[1] = 0; n[2] = dz / len; *is_box = 0; }
    }
  }
  return d;
}

/* DESIGN.md "contact candidates" (spec v2, order independent): every leg has 28 candidate points in a fixed index order
 *   0 foot | 1-3 shank box v0-2 | 4,5 wheel caps | 6 shank v3 | 7-10 shank v4-7 | 11,12 thigh cyl 0 caps | 13 body box vertex z- |
 *   14-17 thigh box v0-3 | 18,19 thigh cyl 1 caps | 20 body box vertex z+ | 21-24 thigh box v4-7 | 25,26 hip cyl caps | 27 handle
 * and keeps the KC candidates of smallest depth below the margin (ties: lower index); the kept ones fill the slots in
 * candidate-index order.  (The index order is the one of the kernel's table, pmc_tables.hpp pmc_build_cand_table.) */
typedef struct { double P[3], depth, mu, rs, n[3]; int body, valid, shape; } OCand;   /* rs: radius when the primitive is a sphere (tested at its centre) */
#define NCAND 40   /* per leg: index = 10 * sub + jj; jj = 7 (mid-link spheres), jj = 8 (terrain edges under the trunk) and jj = 9 (terrain edges across the leg boxes, round 6) exist only with terrain */

static void cand_point(const OPrim* p, const double* Rw, const double* pw, int which, OCand* c) {
  double ctr[3], t[3], Rp[9];
  m3v(Rw, p->pos, t);
  for (int i = 0; i < 3; i++) ctr[i] = pw[i] + t[i];
  m3m(Rw, p->rot, Rp);
  if (p->type == LLM_PRIM_SPHERE) {
    c->P[0] = ctr[0]; c->P[1] = ctr[1]; c->P[2] = ctr[2] - p->size[0];
  } else if (p->type == LLM_PRIM_BOX) {
    double l[3] = {(which & 1) ? p->size[0] : -p->size[0], (which & 2) ? p->size[1] : -p->size[1], (which & 4) ? p->size[2] : -p->size[2]};
    m3v(Rp, l, c->P);
    for (int i = 0; i < 3; i++) c->P[i] += ctr[i];
  } else { /* cylinder, axis = local z: lowest rim point of cap `which` (0: +h, 1: -h) */
    double a[3] = {Rp[2], Rp[5], Rp[8]};
    double proj[3] = {-a[2] * a[0], -a[2] * a[1], 1.0 - a[2] * a[2]};
    double len = sqrt(v3dot(proj, proj)), dir[3];
    if (len < 1e-6) { dir[0] = Rp[0]; dir[1] = Rp[3]; dir[2] = Rp[6]; }
    else { for (int i = 0; i < 3; i++) dir[i] = proj[i] / len; }
    double sg = which ? -1.0 : 1.0;
    for (int i = 0; i < 3; i++) c->P[i] = ctr[i] + sg * p->size[1] * a[i] - p->size[0] * dir[i];
  }
  c->depth = c->P[2];
  c->rs = p->type == LLM_PRIM_SPHERE ? p->size[0] : 0.0;
  c->n[0] = 0; c->n[1] = 0; c->n[2] = 1;
  c->valid = 1;
}

static void capsule(const OModel* M, const OKin* K, int leg, int which, double* a, double* b, double* r, int* body);
/* the candidate points of leg l, index = 10 * sub + jj in the (sub, jj) order of the kernel's table; jj = 8, 9 are left invalid here (reverse_edge, leg_reverse_edge) */
static void enum_cands(const OModel* M, const OKin* K, int l, double mu_foot, double mu_link, const OTerrain* T, OCand* c) {
  memset(c, 0, NCAND * sizeof(OCand));
  int hip = 1 + 3 * l, thigh = 2 + 3 * l, shank = 3 + 3 * l, k = 0;
  const OPrim* lp = M->leg_prims[l];   /* 0 hip cyl | 1 thigh box, 2 thigh cyl 0, 3 thigh cyl 1, 4 wheel | 5 shank box, 6 foot */
#define CAND(prim, body_, which, mu_) do { cand_point(prim, K->Rw[body_], K->pw[body_], which, &c[k]); c[k].body = body_; c[k].mu = mu_; k++; } while (0)
#define MID(which_, f_) do { if (T) { double a_[3], b_[3], r_; int bd_; capsule(M, K, l, which_, a_, b_, &r_, &bd_); \
    for (int i = 0; i < 3; i++) { c[k].P[i] = a_[i] + (f_) * (b_[i] - a_[i]); } \
    c[k].P[2] -= r_; c[k].depth = c[k].P[2]; c[k].rs = r_; \
    c[k].n[0] = 0; c[k].n[1] = 0; c[k].n[2] = 1; c[k].valid = 1; c[k].body = bd_; c[k].mu = mu_link; } k++; } while (0)
  CAND(&lp[6], shank, 0, mu_foot);                                        /*  0      foot                         */
  for (int v = 0; v < 3; v++) CAND(&lp[5], shank, v, mu_link);             /*  1-3    shank box v0..v2             */
  for (int s2 = 0; s2 < 2; s2++) CAND(&lp[4], thigh, s2, mu_link);         /*  4,5    wheel caps                   */
  CAND(&lp[5], shank, 3, mu_link);                                        /*  6      shank box v3                 */
  MID(1, 1.0 / 3.0);                                                      /*  7      shank axis 1/3 (terrain)     */
  k++;                                                                    /*         (jj = 8: reverse_edge)       */
  k++;                                                                    /*         (jj = 9: leg_reverse_edge)   */
  for (int v = 4; v < 8; v++) CAND(&lp[5], shank, v, mu_link);             /*  7-10   shank box v4..v7             */
  for (int s2 = 0; s2 < 2; s2++) CAND(&lp[2], thigh, s2, mu_link);         /*  11,12  thigh cylinder 0 caps        */
  CAND(&M->base_prims[0], 0, l, mu_link);                                 /*  13     body box vertex (leg, z-)    */
  MID(1, 2.0 / 3.0);                                                      /*         shank axis 2/3 (terrain)     */
  k++;                                                                    /*         (jj = 8: reverse_edge)       */
  k++;                                                                    /*         (jj = 9: leg_reverse_edge)   */
  for (int v = 0; v < 4; v++) CAND(&lp[1], thigh, v, mu_link);             /*  14-17  thigh box v0..v3             */
  for (int s2 = 0; s2 < 2; s2++) CAND(&lp[3], thigh, s2, mu_link);         /*  18,19  thigh cylinder 1 caps        */
  CAND(&M->base_prims[0], 0, l + 4, mu_link);                             /*  20     body box vertex (leg, z+)    */
  MID(0, 1.0 / 3.0);                                                      /*         thigh axis 1/3 (terrain)     */
  k++;                                                                    /*         (jj = 8: reverse_edge)       */
  k++;                                                                    /*         (jj = 9: leg_reverse_edge)   */
  for (int v = 4; v < 8; v++) CAND(&lp[1], thigh, v, mu_link);             /*  21-24  thigh box v4..v7             */
  for (int s2 = 0; s2 < 2; s2++) CAND(&lp[0], hip, s2, mu_link);           /*  25,26  hip cylinder caps            */
  if (l == 0 || l == 2) CAND(&M->base_prims[l == 0 ? 1 : 2], 0, 0, mu_link); else k++;   /* 27 handle sphere (legs 0, 2) */
  MID(0, 2.0 / 3.0);                                                      /*         thigh axis 2/3 (terrain)     */
  k++;                                                                    /*         (jj = 8: reverse_edge)       */
  k++;                                                                    /*         (jj = 9: leg_reverse_edge)   */
#undef CAND
#undef MID
}
/* Reverse candidates ("edges under the trunk"): the robot's own candidate points are vertices and spheres, which cannot see
 * a step edge that crosses the flat of the body box between its corners.  So leg l also tests top edge l of every terrain box
 *   l = 0: x = x0,  1: x = x1 (both along y),  2: y = y0,  3: y = y1 (both along x),  all at z = z1 -- or, for a box that floats (its bottom z0 above
 *   LLM_FLOATING_MIN_Z: the hanging bars of bullet_static_entities.py:366-412), at z = z0: the edges the flat of the back meets from below (round 5)
 * against the body box, in the box's own frame (half extents h):
 *   1. the edge is cut to the box grown by the contact margin (nothing left: no candidate);
 *   2. the middle of what is left names the face of the body box the edge runs along: the axis of largest |p_i| - h_i;
 *   3. the edge is cut to the exact extents of the other two axes; the two ends of that piece are the candidates (sub-lanes 0 and 1 of
 *      the leg, candidate jj = 8), each with depth = its signed distance to the face's plane, contact point = the point of the
 *      terrain edge, normal = the face's inward normal (the way the body is pushed), friction partner = the terrain box.
 * Over several boxes each candidate keeps the deepest.  Returns 0 / 1; P, n in world coordinates. */
/* one terrain edge family (edge index e of every box of T: 0: x = x0, 1: x = x1, 2: y = y0, 3: y = y1, at the top -- at the bottom of a floating box) against ONE box of the robot
 * given in world coordinates (Bt: columns = the box's unit axes, cw: its centre, size: half extents): steps 1 - 3 of the rule above.  `end` = 0 / 1 evaluates that end of the
 * clipped piece, `end` = 2 both and keeps the deeper (the first on a tie).  The candidate in `out` is replaced when a deeper one is found (found_before: `out` already holds one). */
static int edge_vs_box(const OTerrain* T, int e, const double* Bt, const double* cw, const double* size, int end, int body, int found, int across, OCand* out) {
  const double margin = g_spec[LLM_SPEC_CONTACT_MARGIN];
  for (int si = 0; si < T->n; si++) {
    const double* r = T->rec + 8 * si;
    const double ze = r[4] > LLM_FLOATING_MIN_Z ? r[4] : r[5];                      /* a floating box (a hanging bar) offers its bottom edges */
    double al[3] = {e == 1 ? r[1] : r[0], e == 3 ? r[3] : r[2], ze}, bl[3] = {e == 0 ? r[0] : r[1], e == 2 ? r[2] : r[3], ze};
    double aw[3], bw[3], pa[3], d[3];
    if (T->yawed) {
      aw[0] = T->cx + al[0] * T->cs - al[1] * T->sn; aw[1] = T->cy + al[0] * T->sn + al[1] * T->cs; aw[2] = al[2];
      bw[0] = T->cx + bl[0] * T->cs - bl[1] * T->sn; bw[1] = T->cy + bl[0] * T->sn + bl[1] * T->cs; bw[2] = bl[2];
    } else { memcpy(aw, al, 24); memcpy(bw, bl, 24); }
    for (int i = 0; i < 3; i++) {            /* into the box's frame */
      pa[i] = 0; d[i] = 0;
      for (int k = 0; k < 3; k++) { pa[i] += Bt[3 * k + i] * (aw[k] - cw[k]); d[i] += Bt[3 * k + i] * (bw[k] - aw[k]); }
    }
    double lo[3], hi[3], t0 = 0, t1 = 1;
    for (int i = 0; i < 3; i++) {
      const double ds = fabs(d[i]) < 1e-9 ? 1e-9 : d[i], H = size[i] + margin;
      const double ta = (-H - pa[i]) / ds, tb = (H - pa[i]) / ds;
      if ((ta < tb ? ta : tb) > t0) t0 = ta < tb ? ta : tb;
      if ((ta < tb ? tb : ta) < t1) t1 = ta < tb ? tb : ta;
      const double ea = (-size[i] - pa[i]) / ds, eb = (size[i] - pa[i]) / ds;
      lo[i] = ea < eb ? ea : eb; hi[i] = ea < eb ? eb : ea;
    }
    if (t0 > t1) continue;
    const double tm = 0.5 * (t0 + t1);
    double q = -INFINITY, sg = 1; int ax = 0;
    /* across (the leg boxes): the face must be one the edge runs ACROSS -- the box axis the edge is most parallel to is not a candidate (a thin leg box is pierced
     * lengthwise by an edge through its two small faces; "the nearest face of the middle point" would be one of those, with a normal along the edge) */
    const int par = across ? ((fabs(d[0]) >= fabs(d[1]) && fabs(d[0]) >= fabs(d[2])) ? 0 : (fabs(d[1]) >= fabs(d[2]) ? 1 : 2)) : -1;
    for (int i = 0; i < 3; i++) {
      if (i == par) continue;
      const double pm = pa[i] + tm * d[i], qi = fabs(pm) - size[i];
      if (qi > q) { q = qi; ax = i; sg = pm >= 0 ? 1.0 : -1.0; }
    }
    double u0 = 0, u1 = 1;
    for (int i = 0; i < 3; i++) {
      if (i == ax) continue;
      if (lo[i] > u0) u0 = lo[i];
      if (hi[i] < u1) u1 = hi[i];
    }
    if (u0 > u1) continue;
    const double dep0 = sg * (pa[ax] + u0 * d[ax]) - size[ax], dep1 = sg * (pa[ax] + u1 * d[ax]) - size[ax];
    const double u = end == 0 ? u0 : (end == 1 ? u1 : (dep1 < dep0 ? u1 : u0));
    const double depth = sg * (pa[ax] + u * d[ax]) - size[ax];
    if (found && depth >= out->depth) continue;
    found = 1;
    out->depth = depth; out->rs = 0; out->body = body; out->valid = 2; out->shape = si;
    for (int i = 0; i < 3; i++) { out->P[i] = aw[i] + u * (bw[i] - aw[i]); out->n[i] = -sg * Bt[3 * i + ax]; }
  }
  return found;
}
static int reverse_edge(const OModel* M, const OKin* K, const OTerrain* T, int l, int s, OCand* out) {
  const OPrim* bx = &M->base_prims[0];
  double Bt[9], cw[3], t[3];
  m3m(K->Rw[0], bx->rot, Bt);
  m3v(K->Rw[0], bx->pos, t);
  for (int i = 0; i < 3; i++) cw[i] = K->pw[0][i] + t[i];
  return edge_vs_box(T, l, Bt, cw, bx->size, s, 0, 0, 0, out);
}
/* ... and the same for the LEG boxes (round 6): the robot's own candidates on a leg are vertices, rim points and two mid-span spheres per link, so a shank laid across the edge of a
 * hurdle between them would sink until one of those arrives.  Candidate jj = 9 of sub-lane s of leg l: edge s of every terrain box against the leg's thigh box, then its shank box,
 * both ends of every clipped piece -- the deepest of all of them (ties: the first in that order).  Normal = the leg box face's inward normal, body = the thigh / shank link. */
static int leg_reverse_edge(const OModel* M, const OKin* K, const OTerrain* T, int l, int s, OCand* out) {
  const OPrim* lp = M->leg_prims[l];
  const int bodies[2] = {2 + 3 * l, 3 + 3 * l}, prims[2] = {1, 5};
  int found = 0;
  for (int b = 0; b < 2; b++) {
    const OPrim* bx = &lp[prims[b]];
    double Bt[9], cw[3], t[3];
    m3m(K->Rw[bodies[b]], bx->rot, Bt);
    m3v(K->Rw[bodies[b]], bx->pos, t);
    for (int i = 0; i < 3; i++) cw[i] = K->pw[bodies[b]][i] + t[i];
    found = edge_vs_box(T, s, Bt, cw, bx->size, 2, bodies[b], found, 1, out);
  }
  return found;
}
/* Diagnostic for the tests: how close the last find_contacts calls of this thread came to the one discontinuity of the deepest-K rule --
 * a candidate's depth crossing (deepest of the round + LLM_SELECT_EPS), where the pick changes hands.  min over rounds and candidates of
 * |depth - deepest - eps|; reset by orc_selection_margin(1). */
static _Thread_local double tl_sel_margin = INFINITY;
double orc_selection_margin(int reset) { const double m = tl_sel_margin; if (reset) tl_sel_margin = INFINITY; return m; }
static int find_contacts(const OModel* M, const OKin* K, double mu_foot, double mu_link, const OTerrain* T, OContact* out) {
  int n = 0;
  for (int l = 0; l < 4; l++) {
    OCand c[NCAND];
    enum_cands(M, K, l, mu_foot, mu_link, T, c);
    if (T) {                                 /* the nearest surface decides depth, normal and friction partner */
      for (int i = 0; i < NCAND; i++) {
        if (!c[i].valid) continue;
        double E[3] = {c[i].P[0], c[i].P[1], c[i].P[2] + c[i].rs};
        for (int si = 0; si < T->n; si++) {
          double nn[3]; int isb;
          double d = terrain_sdf(T, si, E, nn, &isb) - c[i].rs;
          if (d < c[i].depth) {              /* valid: 1 plane, 2 box, 3 edge cylinder */
            c[i].depth = d; memcpy(c[i].n, nn, 24); c[i].valid = isb ? 2 : 3;
            for (int k = 0; k < 3; k++) c[i].P[k] = E[k] - c[i].rs * nn[k];      /* a sphere touches where the surface's normal leaves it */
          }
        }
      }
      for (int s = 0; s < 2 && g_spec[LLM_SPEC_TRUNK_EDGES] > 0.5; s++) { c[10 * s + 8].mu = mu_link; reverse_edge(M, K, T, l, s, &c[10 * s + 8]); }
      for (int s = 0; s < 4 && g_spec[LLM_SPEC_LEG_EDGES] > 0.5; s++) { c[10 * s + 9].mu = mu_link; leg_reverse_edge(M, K, T, l, s, &c[10 * s + 9]); }
    }
    int taken[NCAND] = {0}, nsel = 0;
    const int kc = (int)g_spec[LLM_SPEC_MAX_CONTACTS_PER_LEG];
    for (int s = 0; s < kc; s++) {          /* the KC deepest (ties: lower index) ... */
      int best = -1;
      double dmin = INFINITY;                /* candidates within LLM_SELECT_EPS of the deepest are equally deep: the lowest index wins */
      for (int i = 0; i < NCAND; i++)
        if (c[i].valid && !taken[i] && c[i].depth < g_spec[LLM_SPEC_CONTACT_MARGIN] && c[i].depth < dmin) dmin = c[i].depth;
      for (int i = 0; i < NCAND && best < 0; i++)
        if (c[i].valid && !taken[i] && c[i].depth < g_spec[LLM_SPEC_CONTACT_MARGIN] && c[i].depth <= dmin + g_spec[LLM_SPEC_SELECT_EPS]) best = i;
      if (best < 0) break;
      for (int i = 0; i < NCAND; i++)
        if (c[i].valid && !taken[i] && c[i].depth < g_spec[LLM_SPEC_CONTACT_MARGIN] && fabs(c[i].depth - dmin - g_spec[LLM_SPEC_SELECT_EPS]) < tl_sel_margin)
          tl_sel_margin = fabs(c[i].depth - dmin - g_spec[LLM_SPEC_SELECT_EPS]);
      taken[best] = 1;
      nsel++;
    }
    int slot = 0;
    for (int i = 0; i < NCAND; i++) {          /* ... stored in candidate-index order, so near-ties in depth cannot reorder the solve */
      if (!taken[i]) continue;
      out[n].body = c[i].body; memcpy(out[n].P, c[i].P, 24); out[n].depth = c[i].depth;
      out[n].mu = c[i].mu * (c[i].valid == 2 && T ? T->box_mu_scale : 1.0);
      memcpy(out[n].n, c[i].n, 24);
      out[n].leg = l; out[n].slot = slot++; out[n].cand = i;
      n++;
    }
    (void)nsel;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* self-collision (LR:212-217 URDF_USE_SELF_COLLISION + ..._EXCLUDE_ALL_PARENTS): only links of DIFFERENT legs can touch     */
/* (same-leg links and the body are ancestors of one another).  Spec of this build (unpinned): each leg is two  */
/* capsules -- thigh: the long axis of the thigh box, radius = its larger half thickness; shank: from the upper end of the    */
/* shank box to the foot centre, radius = the mean of the box's half thickness and the foot radius -- and the LLM_MAX_SELF   */
/* closest capsule pairs within the contact margin give one frictionless row each.                                          */
/* ------------------------------------------------------------------------------------------------ */
#define MAX_SELF LLM_MAX_SELF
typedef struct { int bodyA, bodyB; double P[3], n[3], depth; int pair; } OSelf;
static void capsule(const OModel* M, const OKin* K, int leg, int which, double* a, double* b, double* r, int* body) {
  const OPrim* lp = M->leg_prims[leg];
  double la[3], lb[3];
  if (which == 0) {                                   /* thigh */
    const OPrim* bx = &lp[1];
    for (int i = 0; i < 3; i++) { la[i] = bx->pos[i] + bx->size[0] * bx->rot[3 * i]; lb[i] = bx->pos[i] - bx->size[0] * bx->rot[3 * i]; }
    *r = bx->size[1] > bx->size[2] ? bx->size[1] : bx->size[2];
    *body = 2 + 3 * leg;
  } else {                                            /* shank + foot */
    const OPrim* bx = &lp[5]; const OPrim* ft = &lp[6];
    double d[3], len;
    for (int i = 0; i < 3; i++) d[i] = ft->pos[i] - bx->pos[i];
    len = sqrt(v3dot(d, d));
    for (int i = 0; i < 3; i++) { la[i] = bx->pos[i] - bx->size[0] * d[i] / len; lb[i] = ft->pos[i]; }
    *r = 0.5 * ((bx->size[1] > bx->size[2] ? bx->size[1] : bx->size[2]) + ft->size[0]);
    *body = 3 + 3 * leg;
  }
  double t[3];
  m3v(K->Rw[*body], la, t); for (int i = 0; i < 3; i++) a[i] = K->pw[*body][i] + t[i];
  m3v(K->Rw[*body], lb, t); for (int i = 0; i < 3; i++) b[i] = K->pw[*body][i] + t[i];
}
/* closest points of segments p1-q1, p2-q2 (Ericson, Real-Time Collision Detection 5.1.9) */
static void seg_seg(const double* p1, const double* q1, const double* p2, const double* q2, double* c1, double* c2) {
  double d1[3], d2[3], r[3];
  for (int i = 0; i < 3; i++) { d1[i] = q1[i] - p1[i]; d2[i] = q2[i] - p2[i]; r[i] = p1[i] - p2[i]; }
  double a = v3dot(d1, d1), e = v3dot(d2, d2), f = v3dot(d2, r), c = v3dot(d1, r), b = v3dot(d1, d2);
  /* Nearly parallel segments: den = a e sin^2(angle) -> 0 and the textbook quotient is 0 / 0 -- any s between the overlap's ends is a
   * closest point, and which one rounding picks decides where the contact row acts.  The spec regularises: the quotient is pulled
   * toward the middle of the overlap (sa, sb = the other segment's ends projected on this one) with weight LLM_SEG_PARALLEL_REG
   * relative to a e; for angles above a few degrees this is Ericson's closest point, for parallel capsules the middle of the overlap. */
  double sa = -c / a, sb = (b - c) / a;
  sa = sa < 0 ? 0 : (sa > 1 ? 1 : sa); sb = sb < 0 ? 0 : (sb > 1 ? 1 : sb);
  double reg = LLM_SEG_PARALLEL_REG * a * e;
  double den = a * e - b * b, s = (b * f - c * e + reg * 0.5 * (sa + sb)) / (den + reg), t;
  if (s < 0) s = 0;
  if (s > 1) s = 1;
  t = (b * s + f) / e;
  if (t < 0) { t = 0; s = -c / a; if (s < 0) s = 0; if (s > 1) s = 1; }
  else if (t > 1) { t = 1; s = (b - c) / a; if (s < 0) s = 0; if (s > 1) s = 1; }
  for (int i = 0; i < 3; i++) { c1[i] = p1[i] + s * d1[i]; c2[i] = p2[i] + t * d2[i]; }
}
/* pair index order: leg pairs (0,1) (1,2) (2,3) (3,0) (0,2) (1,3), then capsule pair (own, other) = (t,t) (s,t) (t,s) (s,s) */
static int find_self_contacts(const OModel* M, const OKin* K, OSelf* out) {
  static const int LA[6] = {0, 1, 2, 3, 0, 1}, LB[6] = {1, 2, 3, 0, 2, 3};
  OSelf cand[24];
  int nc = 0;
  for (int lp = 0; lp < 6; lp++)
    for (int sp = 0; sp < 4; sp++) {
      double a1[3], b1[3], a2[3], b2[3], r1, r2, c1[3], c2[3], d[3];
      int bA, bB;
      capsule(M, K, LA[lp], sp & 1, a1, b1, &r1, &bA);
      capsule(M, K, LB[lp], (sp >> 1) & 1, a2, b2, &r2, &bB);
      seg_seg(a1, b1, a2, b2, c1, c2);
      for (int i = 0; i < 3; i++) d[i] = c1[i] - c2[i];
      double len = sqrt(v3dot(d, d));
      if (len < 1e-9) continue;
      OSelf* o = &cand[nc++];
      o->bodyA = bA; o->bodyB = bB; o->depth = len - r1 - r2; o->pair = lp * 4 + sp;
      for (int i = 0; i < 3; i++) { o->n[i] = d[i] / len; o->P[i] = 0.5 * ((c1[i] - r1 * o->n[i]) + (c2[i] + r2 * o->n[i])); }
    }
  int n = 0, taken[24] = {0};
  for (int s = 0; s < (int)g_spec[LLM_SPEC_MAX_SELF]; s++) {
    int best = -1;
    double dmin = INFINITY;                  /* (cand[] is in pair-index order) */
    for (int i = 0; i < nc; i++)
      if (!taken[i] && cand[i].depth < g_spec[LLM_SPEC_SELF_MARGIN] && cand[i].depth < dmin) dmin = cand[i].depth;
    for (int i = 0; i < nc && best < 0; i++)
      if (!taken[i] && cand[i].depth < g_spec[LLM_SPEC_SELF_MARGIN] && cand[i].depth <= dmin + g_spec[LLM_SPEC_SELECT_EPS]) best = i;
    if (best < 0) break;
    for (int i = 0; i < nc; i++)             /* the same diagnostic as in find_contacts: a pair's distance near (closest + LLM_SELECT_EPS) */
      if (!taken[i] && cand[i].depth < g_spec[LLM_SPEC_SELF_MARGIN] && fabs(cand[i].depth - dmin - g_spec[LLM_SPEC_SELECT_EPS]) < tl_sel_margin)
        tl_sel_margin = fabs(cand[i].depth - dmin - g_spec[LLM_SPEC_SELECT_EPS]);
    taken[best] = 1;
    out[n++] = cand[best];
  }
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* one physics substep  ==  what stepSimulation() does at PLE:206 (spec: DESIGN.md)                  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  int n_contacts, n_rows;
  double lambda[MAXROWS];
  double acc_free[NDOF];
} OSubstepDiag;

static int substep_terrain(const OModel* M, double dt, int n_iter, double mu_foot, double* state, const double* tau_in, OSubstepDiag* diag,
                           const OTerrain* T, const double* push);
int orc_substep_model(const OModel* M, double dt, int n_iter, double mu_foot, double* state, const double* tau_in,
                      OSubstepDiag* diag) {
  return substep_terrain(M, dt, n_iter, mu_foot, state, tau_in, diag, NULL, NULL);
}
/* T (nullable): EPMC terrain; push (nullable): PR:72-77 applyExternalForce(linkIndex 0, LINK_FRAME) -- a force in the FR hip link's
 * frame at that link's centre of mass */
/* Everything one robot brings to the solve: kinematics, the velocities after the unconstrained update, its rows in the order of the
 * spec, M^-1 J^T.  The projected Gauss-Seidel then runs in velocity form (nu carries the impulses applied so far):
 * w = J nu + bias, lambda' = clamp(lambda - w / (J M^-1 J^T)), nu += M^-1 J^T (lambda' - lambda) -- the same iteration as the
 * A-matrix form, and the form in which rows shared between two robots (SEPMC) are natural. */
typedef struct {
  OKin K;
  double nu[NDOF];
  int nr, no, nc;
  double J[MAXROWS][NDOF], MiJt[MAXROWS][NDOF], Minv[NDOF][NDOF];
  double bias[MAXROWS], lo[MAXROWS], hi[MAXROWS], lam[MAXROWS], mu_row[MAXROWS], dinv[MAXROWS];
  int fric_of[MAXROWS], order[MAXROWS], lim_row[12], con_row[4][KC];
  int con_leg[MAXC], con_slot[MAXC], con_cand[MAXC];
  int ns, self_pair[LLM_MAX_SELF], self_row[LLM_MAX_SELF], self_rows;
  int cone, ellipse, first_self;       /* audit switch: the two friction rows of a ground / terrain contact are solved together inside the cone */
} ORows;

static int assemble_rows(const OModel* M, double dt, double mu_foot, const double* state, const double* tau_in, OSubstepDiag* diag,
                         const OTerrain* T, const double* push, ORows* W) {
#define K (W->K)
#define nu (W->nu)
#define J (W->J)
#define bias (W->bias)
#define lo (W->lo)
#define hi (W->hi)
#define fric_of (W->fric_of)
#define mu_row (W->mu_row)
#define lim_row (W->lim_row)
#define con_row (W->con_row)
  double fext[NB][6], acc[NDOF], tau[12];
  kinematics(M, state, NULL, &K);
  external_forces(M, &K, fext);
  if (push) {
    double t[3];
    v3cross(M->com[1], push, t);
    for (int i = 0; i < 3; i++) { fext[1][i] += t[i]; fext[1][3 + i] += push[i]; }
  }
  for (int i = 0; i < 12; i++) tau[i] = tau_in[i] - M->damp[i] * state[25 + i]; /* URDF joint damping (Bullet adds -d*qd) */
  if (aba(M, &K, state + 25, tau, fext, 1, acc, acc + 6)) return -1;
  if (diag) memcpy(diag->acc_free, acc, sizeof acc);

  /* velocities after the unconstrained update, generalized body-frame coordinates nu = [w_b, v_b, qd] */
  double wxv[3];
  v3cross(K.v[0], K.v[0] + 3, wxv);
  for (int i = 0; i < 3; i++) {
    nu[i] = K.v[0][i] + dt * acc[i];
    nu[3 + i] = K.v[0][3 + i] + dt * (acc[3 + i] + wxv[i]) - 0.0; /* classical accel of the origin, frozen frame */
  }
  /* NOTE: d/dt(world velocity) = R (a_lin + w x v); expressed in the (frozen) body frame of this substep */
  for (int i = 0; i < 12; i++) nu[6 + i] = state[25 + i] + dt * acc[6 + i];
  { const double vm = g_spec[LLM_SPEC_MAX_COORD_VEL];        /* btMultiBody::m_maxCoordinateVelocity (LLM_MAX_COORD_VEL); a NaN or an infinity is an error state, not a bound */
    for (int i = 0; i < NDOF; i++) nu[i] = !isfinite(nu[i]) ? NAN : (nu[i] > vm ? vm : (nu[i] < -vm ? -vm : nu[i])); }

  /* ---- constraint rows --------------------------------------------------------------------------- */
  OContact C[MAXC];
  int nc = find_contacts(M, &K, mu_foot, LLM_LINK_FRICTION * LLM_PLANE_FRICTION, T, C);
  int nr = 0;
  W->nc = nc;
  for (int c = 0; c < nc; c++) { W->con_leg[c] = C[c].leg; W->con_slot[c] = C[c].slot; }
  /* joint limits: one unilateral row per joint toward its nearer limit (URDF <limit>, Bullet
   * btMultiBodyJointLimitConstraint).  A row can only ever act if its free approach speed s*qd* + bias is small,
   * so rows with s*qd* + bias >= LLM_LIMIT_GATE are left out of the solve (DESIGN.md "joint limits"). */
  for (int i = 0; i < 12; i++) {
    double dl = state[13 + i] - M->qlo[i], dh = M->qhi[i] - state[13 + i];
    double d = dl <= dh ? dl : dh, sgn = dl <= dh ? 1.0 : -1.0;
    const double lerp = g_spec[LLM_SPEC_LIMIT_ERP] >= 0 ? g_spec[LLM_SPEC_LIMIT_ERP] : g_spec[LLM_SPEC_ERP];
    double bz = row_bias(d, dt, lerp, 0);
    lim_row[i] = -1;
    if (g_spec[LLM_SPEC_LIMIT_SPECULATIVE] < 0.5) {                              /* Bullet's rule: a row only once the limit is passed, and then whatever the speed */
      if (d > 0) continue;
    } else if (!(sgn * nu[6 + i] + bz < g_spec[LLM_SPEC_LIMIT_GATE])) continue;
    memset(J[nr], 0, sizeof J[nr]);
    J[nr][6 + i] = sgn;
    bias[nr] = bz;
    lo[nr] = 0; hi[nr] = INFINITY; fric_of[nr] = -1; mu_row[nr] = 0;
    lim_row[i] = nr;
    nr++;
  }
  /* contacts: normal + two friction rows, directions n=+z, t1=(0,-1,0), t2=(1,0,0) (btPlaneSpace1 of +z) */
  for (int l = 0; l < 4; l++) for (int k = 0; k < KC; k++) con_row[l][k] = -1;
  for (int c = 0; c < nc; c++) {
    int b = C[c].body;
    con_row[C[c].leg][C[c].slot] = nr;
    double ploc[3], d3[3];
    for (int i = 0; i < 3; i++) d3[i] = C[c].P[i] - K.pw[b][i];
    m3tv(K.Rw[b], d3, ploc);
    double dirs[3][3];                      /* n and btPlaneSpace1(n): for n = +z these are +z, -y, +x */
    {
      const double* nn = C[c].n;
      memcpy(dirs[0], nn, 24);
      if (fabs(nn[2]) > 0.7071067811865475) {
        double a = nn[1] * nn[1] + nn[2] * nn[2], kk = 1.0 / sqrt(a);
        dirs[1][0] = 0; dirs[1][1] = -nn[2] * kk; dirs[1][2] = nn[1] * kk;
        dirs[2][0] = a * kk; dirs[2][1] = -nn[0] * dirs[1][2]; dirs[2][2] = nn[0] * dirs[1][1];
      } else {
        double a = nn[0] * nn[0] + nn[1] * nn[1], kk = 1.0 / sqrt(a);
        dirs[1][0] = -nn[1] * kk; dirs[1][1] = nn[0] * kk; dirs[1][2] = 0;
        dirs[2][0] = -nn[2] * dirs[1][1]; dirs[2][1] = nn[2] * dirs[1][0]; dirs[2][2] = a * kk;
      }
      if (g_spec[LLM_SPEC_FRICTION_DIRS] > 0.5) {
        /* audit switch: t1 along the lateral velocity of the contact point (velocities after the unconstrained update, as Bullet converts
         * contacts after stepVelocities), t2 = t1 x n */
        OKin Kn;
        double t[3], vl[3], vw[3];
        kinematics(M, state, nu, &Kn);
        v3cross(Kn.v[b], ploc, t);
        for (int i = 0; i < 3; i++) vl[i] = Kn.v[b][3 + i] + t[i];
        m3v(K.Rw[b], vl, vw);
        const double vn = v3dot(vw, nn);
        double lat[3] = {vw[0] - vn * nn[0], vw[1] - vn * nn[1], vw[2] - vn * nn[2]};
        const double l2 = v3dot(lat, lat);
        if (l2 > 1.1920929e-7) {
          const double il = 1.0 / sqrt(l2);
          for (int i = 0; i < 3; i++) dirs[1][i] = lat[i] * il;
          v3cross(dirs[1], nn, dirs[2]);
        }
      }
    }
    for (int r = 0; r < 3; r++) {
      for (int d = 0; d < NDOF; d++) {
        double e[NDOF];
        memset(e, 0, sizeof e);
        e[d] = 1.0;
        OKin Kd;
        kinematics(M, state, e, &Kd);
        double t[3], vl[3], vw[3];
        v3cross(Kd.v[b], ploc, t);
        for (int i = 0; i < 3; i++) vl[i] = Kd.v[b][3 + i] + t[i];
        m3v(K.Rw[b], vl, vw);
        J[nr][d] = v3dot(dirs[r], vw);
      }
      if (r == 0) {
        bias[nr] = row_bias(C[c].depth, dt, g_spec[LLM_SPEC_ERP], 1);
        lo[nr] = 0; hi[nr] = INFINITY; fric_of[nr] = -1; mu_row[nr] = 0;
      } else {
        bias[nr] = 0; lo[nr] = 0; hi[nr] = 0; fric_of[nr] = nr - r; mu_row[nr] = C[c].mu;
      }
      nr++;
    }
  }
  /* self-collision rows: n . (v_A(P) - v_B(P)) >= -depth/dt; frictionless in the spec, with LLM_SPEC_SELF_FRICTION = mu > 0 (deviation
   * study) followed by two tangential rows along btPlaneSpace1(n) bounded by mu times the normal multiplier */
  OSelf SC[MAX_SELF];
  int ns = g_self_collision ? find_self_contacts(M, &K, SC) : 0, self_row[MAX_SELF];
  const double mu_self = g_spec[LLM_SPEC_SELF_FRICTION];
  const int self_rows = mu_self > 0 ? 3 : 1;
  W->ns = ns;
  for (int c = 0; c < ns; c++) {
    self_row[c] = nr;
    W->self_pair[c] = SC[c].pair;
    double dirs[3][3];
    {
      const double* nn = SC[c].n;
      memcpy(dirs[0], nn, 24);
      if (fabs(nn[2]) > 0.7071067811865475) {
        double a = nn[1] * nn[1] + nn[2] * nn[2], kk = 1.0 / sqrt(a);
        dirs[1][0] = 0; dirs[1][1] = -nn[2] * kk; dirs[1][2] = nn[1] * kk;
        dirs[2][0] = a * kk; dirs[2][1] = -nn[0] * dirs[1][2]; dirs[2][2] = nn[0] * dirs[1][1];
      } else {
        double a = nn[0] * nn[0] + nn[1] * nn[1], kk = 1.0 / sqrt(a);
        dirs[1][0] = -nn[1] * kk; dirs[1][1] = nn[0] * kk; dirs[1][2] = 0;
        dirs[2][0] = -nn[2] * dirs[1][1]; dirs[2][1] = nn[2] * dirs[1][0]; dirs[2][2] = a * kk;
      }
    }
    for (int r = 0; r < self_rows; r++) {
      for (int d = 0; d < NDOF; d++) {
        double e[NDOF];
        memset(e, 0, sizeof e);
        e[d] = 1.0;
        OKin Kd;
        kinematics(M, state, e, &Kd);
        double rel = 0;
        for (int side = 0; side < 2; side++) {
          int b = side ? SC[c].bodyB : SC[c].bodyA;
          double d3[3], ploc[3], t[3], vl[3], vw[3];
          for (int i = 0; i < 3; i++) d3[i] = SC[c].P[i] - K.pw[b][i];
          m3tv(K.Rw[b], d3, ploc);
          v3cross(Kd.v[b], ploc, t);
          for (int i = 0; i < 3; i++) vl[i] = Kd.v[b][3 + i] + t[i];
          m3v(K.Rw[b], vl, vw);
          rel += (side ? -1.0 : 1.0) * v3dot(dirs[r], vw);
        }
        J[nr][d] = rel;
      }
      if (r == 0) {
        bias[nr] = row_bias(SC[c].depth, dt, g_spec[LLM_SPEC_ERP], 1);
        lo[nr] = 0; hi[nr] = INFINITY; fric_of[nr] = -1; mu_row[nr] = 0;
      } else {
        bias[nr] = 0; lo[nr] = 0; hi[nr] = 0; fric_of[nr] = nr - r; mu_row[nr] = mu_self;
      }
      nr++;
    }
  }
  /* M^-1 J^T by unit responses of the ABA at zero velocity */
  {
    double zero12[12] = {0};
    for (int d = 0; d < NDOF; d++) {
      double fe[NB][6], t12[12] = {0}, col[NDOF];
      memset(fe, 0, sizeof fe);
      if (d < 6) fe[0][d] = 1.0; else t12[d - 6] = 1.0;
      if (aba(M, &K, zero12, t12, fe, 0, col, col + 6)) return -1;
      for (int i = 0; i < NDOF; i++) W->Minv[i][d] = col[i];
    }
    for (int r = 0; r < nr; r++) {
      double dd = 0;
      for (int i = 0; i < NDOF; i++) {
        double s = 0;
        for (int k = 0; k < NDOF; k++) s += W->Minv[i][k] * J[r][k];
        W->MiJt[r][i] = s;
      }
      for (int k = 0; k < NDOF; k++) dd += J[r][k] * W->MiJt[r][k];
      W->dinv[r] = 1.0 / dd;
      W->lam[r] = 0;
    }
  }
  /* Row order of the spec: the limit rows (joint, leg), then all normal rows (slot, leg), all t1 rows, all t2 rows,
   * then the self-collision rows. */
  int no = 0;
  for (int j = 0; j < 3; j++)
    for (int l = 0; l < 4; l++)
      if (lim_row[3 * l + j] >= 0) W->order[no++] = lim_row[3 * l + j];
  {
    /* contacts in solve order: slot-major (spec), or per body pair as a manifold lists them (audit switch LLM_SPEC_ROW_ORDER) */
    int oc[MAXC], noc = 0;
    if (g_spec[LLM_SPEC_ROW_ORDER] > 0.5) {
      for (int b = 0; b < NB; b++)
        for (int cand = 0; cand < NCAND; cand++)
          for (int c = 0; c < nc; c++)
            if (C[c].body == b && C[c].cand == cand) oc[noc++] = con_row[C[c].leg][C[c].slot];
    } else {
      for (int k = 0; k < KC; k++)
        for (int l = 0; l < 4; l++)
          if (con_row[l][k] >= 0) oc[noc++] = con_row[l][k];
    }
    for (int i = 0; i < noc; i++) W->order[no++] = oc[i];                       /* all normal rows */
    if (g_spec[LLM_SPEC_FRICTION_MODE] > 0.5 && g_spec[LLM_SPEC_FRICTION_MODE] < 2.5) {
      for (int i = 0; i < noc; i++) { W->order[no++] = oc[i] + 1; W->order[no++] = oc[i] + 2; }   /* t1, t2 of a contact adjacent */
    } else {
      for (int r = 1; r < 3; r++)
        for (int i = 0; i < noc; i++) W->order[no++] = oc[i] + r;                /* all t1 rows, then all t2 rows */
    }
  }
  W->cone = g_spec[LLM_SPEC_FRICTION_MODE] > 1.5 && g_spec[LLM_SPEC_FRICTION_MODE] < 2.5;
  W->ellipse = g_spec[LLM_SPEC_FRICTION_MODE] > 2.5;
  W->first_self = nr - ns * self_rows;
  for (int c = 0; c < ns; c++)
    for (int r = 0; r < self_rows; r++) W->order[no++] = self_row[c] + r;
  W->nr = nr; W->no = no;
  for (int c = 0; c < nc; c++) W->con_cand[c] = C[c].cand;
  for (int c = 0; c < ns; c++) W->self_row[c] = self_row[c];
  W->self_rows = self_rows;
  return 0;
#undef K
#undef nu
#undef J
#undef bias
#undef lo
#undef hi
#undef fric_of
#undef mu_row
#undef lim_row
#undef con_row
}

/* one Gauss-Seidel sweep over a robot's own rows (LR:261 numSolverIterations sweeps per substep) */
static void sweep_rows(ORows* W) {
  const int keep = g_spec[LLM_SPEC_FRICTION_KEEP] > 0.5;
  for (int oi = 0; oi < W->no; oi++) {
    const int r = W->order[oi];
    if (W->cone && r < W->first_self && W->fric_of[r] >= 0 && W->fric_of[r] == r - 1 && oi + 1 < W->no && W->order[oi + 1] == r + 1) {
      /* btMultiBodyConstraintSolver::resolveConeFrictionConstraintRows as published: both increments from the SAME velocity, the pair
       * scaled back onto the cone |(t1, t2)| <= mu * (normal multiplier), then both applied */
      const int r2 = r + 1, rn = W->fric_of[r];
      if (keep && !(W->lam[rn] > 0)) { oi++; continue; }        /* (audit switch LLM_SPEC_FRICTION_KEEP: "if (totalImpulse > 0)") */
      double w1 = W->bias[r], w2 = W->bias[r2];
      for (int k = 0; k < NDOF; k++) { w1 += W->J[r][k] * W->nu[k]; w2 += W->J[r2][k] * W->nu[k]; }
      double t1 = W->lam[r] - w1 * W->dinv[r], t2 = W->lam[r2] - w2 * W->dinv[r2];
      const double lim = W->mu_row[r] * W->lam[rn], len2 = t1 * t1 + t2 * t2;
      if (len2 > lim * lim) { const double sc = lim > 0 ? lim / sqrt(len2) : 0.0; t1 *= sc; t2 *= sc; }
      const double d1 = t1 - W->lam[r], d2 = t2 - W->lam[r2];
      W->lam[r] = t1; W->lam[r2] = t2;
      for (int k = 0; k < NDOF; k++) W->nu[k] += W->MiJt[r][k] * d1 + W->MiJt[r2][k] * d2;
      oi++;
      continue;
    }
    double w = W->bias[r];
    for (int k = 0; k < NDOF; k++) w += W->J[r][k] * W->nu[k];
    double l_new = W->lam[r] - w * W->dinv[r];
    double l_lo = W->lo[r], l_hi = W->hi[r];
    if (W->fric_of[r] >= 0) {
      if (keep && !(W->lam[W->fric_of[r]] > 0)) continue;
      l_hi = W->mu_row[r] * W->lam[W->fric_of[r]];
      /* mode 3: the spec's round structure (all t1, then all t2) with each friction bound shrunk to what the other row of the contact leaves
       * of the cone: |t1| <= sqrt((mu N)^2 - t2^2), |t2| <= sqrt((mu N)^2 - t1^2).  Not Bullet's coupled clip, but the same admissible set,
       * and it fits the kernel's rounds.  (Bounding only t2 -- t1 first come, first served -- starves the forward direction: tracked 0.63.) */
      if (W->ellipse && r < W->first_self) {     /* each of the two rows is bounded by what the OTHER's current multiplier leaves of the cone */
        const double other = W->lam[W->fric_of[r] == r - 2 ? r - 1 : r + 1];
        l_hi = sqrt(fmax(l_hi * l_hi - other * other, 0.0));
      }
      l_lo = -l_hi;
    }
    if (l_new < l_lo) l_new = l_lo;
    if (l_new > l_hi) l_new = l_hi;
    const double d = l_new - W->lam[r];
    W->lam[r] = l_new;
    for (int k = 0; k < NDOF; k++) W->nu[k] += W->MiJt[r][k] * d;
  }
}
/* ... and once more when the solver's result is written back (applyDeltaVeeMultiDof), before the positions are integrated */
static void clip_velocities(ORows* W) {
  const double vmax = g_spec[LLM_SPEC_MAX_COORD_VEL];
  for (int k = 0; k < NDOF; k++) W->nu[k] = !isfinite(W->nu[k]) ? NAN : (W->nu[k] > vmax ? vmax : (W->nu[k] < -vmax ? -vmax : W->nu[k]));
}

static void integrate_state(const ORows* W, double dt, double* state) {
  const double* nu = W->nu;
#define K (W->K)
  /* ---- integrate positions with the NEW velocities (semi-implicit Euler) -------------------------- */
  double vw[3], ww[3];
  m3v(K.Rw[0], nu, ww);
  m3v(K.Rw[0], nu + 3, vw);
  for (int i = 0; i < 3; i++) { state[7 + i] = vw[i]; state[10 + i] = ww[i]; state[i] += dt * vw[i]; }
  double rv[3] = {ww[0] * dt, ww[1] * dt, ww[2] * dt}, dq[4], qn[4], qo[4];
  q_from_rotvec(rv, dq);
  q_normalize(state + 3, qn);
  q_mul(dq, qn, qo);
  q_normalize(qo, state + 3);
  for (int i = 0; i < 12; i++) { state[25 + i] = nu[6 + i]; state[13 + i] += dt * nu[6 + i]; }
#undef K
}


/* Warm starting (LLM_SPEC_WARM_START = factor > 0; deviation study only -- the spec, like btMultiBodyConstraintSolver for multibody
 * contacts as far as this author recalls its source, starts every substep from zero multipliers): the multipliers a robot's rows
 * ended the previous substep with, keyed by what persists -- joint, (leg, candidate index), capsule pair. */
typedef struct { double lim[12], con[4][NCAND][3], self[24][3]; } OWarm;
static _Thread_local OWarm* tl_warm = NULL;   /* set by orc_step_env around its substeps; NULL for the stateless entry points */
static void warm_apply(ORows* W, const OWarm* c, double f) {
  for (int r = 0; r < W->nr; r++) W->lam[r] = 0;
  for (int i = 0; i < 12; i++) if (W->lim_row[i] >= 0) W->lam[W->lim_row[i]] = f * c->lim[i];
  for (int k = 0; k < W->nc; k++)
    for (int r = 0; r < 3; r++) W->lam[W->con_row[W->con_leg[k]][W->con_slot[k]] + r] = f * c->con[W->con_leg[k]][W->con_cand[k]][r];
  for (int k = 0; k < W->ns; k++)
    for (int r = 0; r < W->self_rows; r++) W->lam[W->self_row[k] + r] = f * c->self[W->self_pair[k]][r];
  for (int r = 0; r < W->nr; r++)
    if (W->lam[r] != 0)
      for (int k = 0; k < NDOF; k++) W->nu[k] += W->MiJt[r][k] * W->lam[r];
}
static void warm_store(const ORows* W, OWarm* c) {
  memset(c, 0, sizeof *c);
  for (int i = 0; i < 12; i++) if (W->lim_row[i] >= 0) c->lim[i] = W->lam[W->lim_row[i]];
  for (int k = 0; k < W->nc; k++)
    for (int r = 0; r < 3; r++) c->con[W->con_leg[k]][W->con_cand[k]][r] = W->lam[W->con_row[W->con_leg[k]][W->con_slot[k]] + r];
  for (int k = 0; k < W->ns; k++)
    for (int r = 0; r < W->self_rows; r++) c->self[W->self_pair[k]][r] = W->lam[W->self_row[k] + r];
}

static int substep_terrain(const OModel* M, double dt, int n_iter, double mu_foot, double* state, const double* tau_in, OSubstepDiag* diag,
                           const OTerrain* T, const double* push) {
  static _Thread_local ORows W;
  if (assemble_rows(M, dt, mu_foot, state, tau_in, diag, T, push, &W)) return -1;
  if (tl_warm && g_spec[LLM_SPEC_WARM_START] > 0) warm_apply(&W, tl_warm, g_spec[LLM_SPEC_WARM_START]);
  for (int it = 0; it < n_iter; it++) sweep_rows(&W);
  clip_velocities(&W);
  if (tl_warm) warm_store(&W, tl_warm);
  if (diag) {   /* fixed layout for the tests: [12 limit rows (0 when gated out)] [3 rows per contact, contact order] */
    diag->n_contacts = W.nc; diag->n_rows = 12 + 3 * W.nc;
    memset(diag->lambda, 0, sizeof diag->lambda);
    for (int i = 0; i < 12; i++) if (W.lim_row[i] >= 0) diag->lambda[i] = W.lam[W.lim_row[i]];
    for (int c = 0; c < W.nc; c++)
      for (int r = 0; r < 3; r++) diag->lambda[12 + 3 * c + r] = W.lam[W.con_row[W.con_leg[c]][W.con_slot[c]] + r];
  }
  integrate_state(&W, dt, state);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* SEPMC: two robots in one world (CTG:383-388).  Spec of this build (unpinned): each robot is ten capsules -- thigh and */
/* shank-with-foot per leg (as for self-collision) and two along the trunk box, radius its half height, side by side; the two       */
/* deepest of the 100 pairs within the contact margin (ties: lower pair id = 10 * capsule of robot 0 + capsule of robot 1) give one */
/* frictionless row each, solved after both robots' own rows in every iteration.                                                 */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { int body[2]; double P[3], n[3], depth; int id; } OPair;
static void pair_capsule(const OModel* M, const OKin* K, int idx, double* a, double* b, double* r, int* body) {
  if (idx < 8) { capsule(M, K, idx >> 1, idx & 1, a, b, r, body); return; }
  const OPrim* bb = &M->base_prims[0];
  const double rr = bb->size[2], hx = bb->size[0] - rr, oy = (idx == 8 ? 1.0 : -1.0) * (bb->size[1] - rr);
  double la[3], lb[3], t[3];
  for (int i = 0; i < 3; i++) {
    la[i] = bb->pos[i] + hx * bb->rot[3 * i] + oy * bb->rot[3 * i + 1];
    lb[i] = bb->pos[i] - hx * bb->rot[3 * i] + oy * bb->rot[3 * i + 1];
  }
  *r = rr; *body = 0;
  m3v(K->Rw[0], la, t); for (int i = 0; i < 3; i++) a[i] = K->pw[0][i] + t[i];
  m3v(K->Rw[0], lb, t); for (int i = 0; i < 3; i++) b[i] = K->pw[0][i] + t[i];
}
static int find_pair_contacts(const OModel* M, const OKin* K0, const OKin* K1, OPair* out) {
  static _Thread_local OPair cand[100];
  int nc = 0;
  for (int ia = 0; ia < 10; ia++)
    for (int ib = 0; ib < 10; ib++) {
      double a1[3], b1[3], a2[3], b2[3], r1, r2, c1[3], c2[3], d[3];
      int bA, bB;
      pair_capsule(M, K0, ia, a1, b1, &r1, &bA);
      pair_capsule(M, K1, ib, a2, b2, &r2, &bB);
      seg_seg(a1, b1, a2, b2, c1, c2);
      for (int i = 0; i < 3; i++) d[i] = c1[i] - c2[i];
      double len = sqrt(v3dot(d, d));
      if (len < 1e-9) continue;
      OPair* o = &cand[nc++];
      o->body[0] = bA; o->body[1] = bB; o->depth = len - r1 - r2; o->id = ia * 10 + ib;
      for (int i = 0; i < 3; i++) { o->n[i] = d[i] / len; o->P[i] = 0.5 * ((c1[i] - r1 * o->n[i]) + (c2[i] + r2 * o->n[i])); }
    }
  int n = 0, taken[100] = {0};
  const int max_pair = (int)(g_spec[LLM_SPEC_MAX_PAIR] + 0.5);      /* spec: 2 */
  for (int s = 0; s < max_pair; s++) {
    int best = -1;
    double dmin = INFINITY;                  /* (cand[] is in pair-id order) */
    for (int i = 0; i < nc; i++)
      if (!taken[i] && cand[i].depth < LLM_CONTACT_MARGIN && cand[i].depth < dmin) dmin = cand[i].depth;
    for (int i = 0; i < nc && best < 0; i++)
      if (!taken[i] && cand[i].depth < LLM_CONTACT_MARGIN && cand[i].depth <= dmin + g_spec[LLM_SPEC_SELECT_EPS]) best = i;
    if (best < 0) break;
    for (int i = 0; i < nc; i++)             /* selection diagnostic, as in find_contacts */
      if (!taken[i] && cand[i].depth < LLM_CONTACT_MARGIN && fabs(cand[i].depth - dmin - g_spec[LLM_SPEC_SELECT_EPS]) < tl_sel_margin)
        tl_sel_margin = fabs(cand[i].depth - dmin - g_spec[LLM_SPEC_SELECT_EPS]);
    taken[best] = 1;
    out[n++] = cand[best];
  }
  return n;
}
/* SEPMC contact bookkeeping, this build's spec: does a leg / wheel link of the robot -- any candidate point below the
 * trunk except the foot sphere (index 0 of every leg) -- have a contact point with the plane or a box other than `flag` / with the flag? */
static void touch_classes(const OModel* M, const OKin* K, const OTerrain* T, int flag, int* t_static, int* t_flag) {
  *t_static = 0; *t_flag = 0;
  for (int l = 0; l < 4; l++) {
    OCand c[NCAND];
    enum_cands(M, K, l, 0.0, 0.0, T, c);
    for (int i = 0; i < NCAND; i++) {
      if (!c[i].valid || c[i].body == 0 || i == 0) continue;
      if (c[i].P[2] < LLM_CONTACT_MARGIN) *t_static = 1;
      const double E[3] = {c[i].P[0], c[i].P[1], c[i].P[2] + c[i].rs};
      for (int si = 0; T && si < T->n; si++) {
        double nn[3]; int isb;
        if (terrain_sdf(T, si, E, nn, &isb) - c[i].rs < LLM_CONTACT_MARGIN) { if (si == flag) *t_flag = 1; else *t_static = 1; }
      }
    }
    for (int s = 0; T && s < 4 && g_spec[LLM_SPEC_LEG_EDGES] > 0.5; s++) {      /* a leg box lying on a box's edge is a leg link touching that box (round 6) */
      OCand e;
      memset(&e, 0, sizeof e);
      if (leg_reverse_edge(M, K, T, l, s, &e) && e.depth < LLM_CONTACT_MARGIN) { if (e.shape == flag) *t_flag = 1; else *t_static = 1; }
    }
  }
}
/* ... and with the other robot: a thigh capsule, or a shank capsule away from its foot end (closest-point parameter <= 0.9) */
static void pair_touch(const OModel* M, const OKin* K0, const OKin* K1, int* touch0, int* touch1) {
  *touch0 = 0; *touch1 = 0;
  for (int ia = 0; ia < 10; ia++)
    for (int ib = 0; ib < 10; ib++) {
      double a1[3], b1[3], a2[3], b2[3], r1, r2, c1[3], c2[3], d[3], u[3], w[3];
      int bA, bB;
      pair_capsule(M, K0, ia, a1, b1, &r1, &bA);
      pair_capsule(M, K1, ib, a2, b2, &r2, &bB);
      seg_seg(a1, b1, a2, b2, c1, c2);
      for (int i = 0; i < 3; i++) d[i] = c1[i] - c2[i];
      const double len = sqrt(v3dot(d, d));
      if (len < 1e-9 || len - r1 - r2 >= LLM_CONTACT_MARGIN) continue;
      for (int i = 0; i < 3; i++) { u[i] = c1[i] - a1[i]; w[i] = b1[i] - a1[i]; }
      const double s0 = sqrt(v3dot(u, u) / v3dot(w, w));
      for (int i = 0; i < 3; i++) { u[i] = c2[i] - a2[i]; w[i] = b2[i] - a2[i]; }
      const double s1 = sqrt(v3dot(u, u) / v3dot(w, w));
      if (ia < 8 && !((ia & 1) && s0 > 0.9)) *touch0 = 1;
      if (ib < 8 && !((ib & 1) && s1 > 0.9)) *touch1 = 1;
    }
}
int orc_touch_model(const OModel* M, const double* state0, const double* state1, const OTerrain* T0, int flag0, const OTerrain* T1, int flag1, int32_t* out6) {
  OKin K0, K1;
  kinematics(M, state0, NULL, &K0);
  kinematics(M, state1, NULL, &K1);
  int a, b;
  touch_classes(M, &K0, T0, flag0, &a, &b); out6[0] = a; out6[1] = b;
  touch_classes(M, &K1, T1, flag1, &a, &b); out6[3] = a; out6[4] = b;
  pair_touch(M, &K0, &K1, &a, &b); out6[2] = a; out6[5] = b;
  return 0;
}
/* states[2][37], taus[2][12], pushes[2] (nullable entries), one terrain for both; returns the number of shared rows */
int orc_substep_pair_model(const OModel* M, double dt, int n_iter, const double* mu_foot2, double* state0, double* state1, const double* tau0,
                           const double* tau1, const OTerrain* T0, const OTerrain* T1, const double* push0, const double* push1, double* pair_rows) {
  static _Thread_local ORows W[2];
  double* st[2] = {state0, state1};
  if (assemble_rows(M, dt, mu_foot2[0], state0, tau0, NULL, T0, push0, &W[0])) return -1;
  if (assemble_rows(M, dt, mu_foot2[1], state1, tau1, NULL, T1, push1, &W[1])) return -1;
  /* Shared rows.  Spec: up to 2 contacts per robot pair, one frictionless row each.  Audit switches (deviation study only):
   * LLM_SPEC_MAX_PAIR contacts (a manifold holds four), LLM_SPEC_PAIR_FRICTION = mu > 0 adds two tangential rows along btPlaneSpace1(n)
   * per contact, bounded by mu times the normal multiplier, solved right after their normal row. */
  OPair PC[4];
  const int np = find_pair_contacts(M, &W[0].K, &W[1].K, PC);
  const double mu_pair = g_spec[LLM_SPEC_PAIR_FRICTION];
  const int prows = mu_pair > 0 ? 3 : 1;
  static _Thread_local double Jp[12][2][NDOF], MJ[12][2][NDOF];
  double dinv[12], bias[12], lam[12];
  for (int c = 0; c < np; c++) {
    double dirs[3][3];
    {
      const double* nn = PC[c].n;
      memcpy(dirs[0], nn, 24);
      if (fabs(nn[2]) > 0.7071067811865475) {
        double a = nn[1] * nn[1] + nn[2] * nn[2], kk = 1.0 / sqrt(a);
        dirs[1][0] = 0; dirs[1][1] = -nn[2] * kk; dirs[1][2] = nn[1] * kk;
        dirs[2][0] = a * kk; dirs[2][1] = -nn[0] * dirs[1][2]; dirs[2][2] = nn[0] * dirs[1][1];
      } else {
        double a = nn[0] * nn[0] + nn[1] * nn[1], kk = 1.0 / sqrt(a);
        dirs[1][0] = -nn[1] * kk; dirs[1][1] = nn[0] * kk; dirs[1][2] = 0;
        dirs[2][0] = -nn[2] * dirs[1][1]; dirs[2][1] = nn[2] * dirs[1][0]; dirs[2][2] = a * kk;
      }
    }
    for (int r = 0; r < prows; r++) {
      const int q = 3 * c + r;
      double dd = 0;
      for (int side = 0; side < 2; side++) {
        const OKin* K = &W[side].K;
        const int b = PC[c].body[side];
        double d3[3], ploc[3];
        for (int i = 0; i < 3; i++) d3[i] = PC[c].P[i] - K->pw[b][i];
        m3tv(K->Rw[b], d3, ploc);
        for (int d = 0; d < NDOF; d++) {
          double e[NDOF], t[3], vl[3], vw[3];
          memset(e, 0, sizeof e);
          e[d] = 1.0;
          OKin Kd;
          kinematics(M, st[side], e, &Kd);
          v3cross(Kd.v[b], ploc, t);
          for (int i = 0; i < 3; i++) vl[i] = Kd.v[b][3 + i] + t[i];
          m3v(K->Rw[b], vl, vw);
          Jp[q][side][d] = (side ? -1.0 : 1.0) * v3dot(dirs[r], vw);       /* n points from robot 1 to robot 0 */
        }
        for (int i = 0; i < NDOF; i++) {
          double s = 0;
          for (int k = 0; k < NDOF; k++) s += W[side].Minv[i][k] * Jp[q][side][k];
          MJ[q][side][i] = s;
        }
        for (int k = 0; k < NDOF; k++) dd += Jp[q][side][k] * MJ[q][side][k];
      }
      dinv[q] = 1.0 / dd;
      bias[q] = r ? 0.0 : row_bias(PC[c].depth, dt, g_spec[LLM_SPEC_ERP], 1);
      lam[q] = 0;
    }
    if (pair_rows && c < 2) { for (int i = 0; i < 3; i++) { pair_rows[8 * c + i] = PC[c].P[i]; pair_rows[8 * c + 3 + i] = PC[c].n[i]; } pair_rows[8 * c + 6] = PC[c].depth; pair_rows[8 * c + 7] = PC[c].id; }
  }
  for (int it = 0; it < n_iter; it++) {
    sweep_rows(&W[0]);
    sweep_rows(&W[1]);
    for (int c = 0; c < np; c++)
      for (int r = 0; r < prows; r++) {
        const int q = 3 * c + r;
        double w = bias[q];
        for (int side = 0; side < 2; side++)
          for (int k = 0; k < NDOF; k++) w += Jp[q][side][k] * W[side].nu[k];
        double l_new = lam[q] - w * dinv[q];
        if (r == 0) { if (l_new < 0) l_new = 0; }
        else { const double lim = mu_pair * lam[3 * c]; if (l_new > lim) l_new = lim; if (l_new < -lim) l_new = -lim; }
        const double d = l_new - lam[q];
        lam[q] = l_new;
        for (int side = 0; side < 2; side++)
          for (int k = 0; k < NDOF; k++) W[side].nu[k] += MJ[q][side][k] * d;
      }
  }
  clip_velocities(&W[0]); clip_velocities(&W[1]);
  integrate_state(&W[0], dt, state0);
  integrate_state(&W[1], dt, state1);
  return np;
}

/* ------------------------------------------------------------------------------------------------ */
/* the environment (PLE:26-435) -- a batch of envs sharing one prioritized-sampling table            */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  double state[37], kin[37], time, reward_sum, frac;
  double hist_prop[LL_STACK][LL_PROP_FRAME_MAX], hist_act[LL_STACK][12];
  int hist_n, clip, frame_id, ep_steps, done_reason, ob_id;
  double feet_dyn[12], feet_kin[12];
  OWarm warm;     /* multipliers of the previous substep (only read when LLM_SPEC_WARM_START > 0) */
} OEnv;

typedef struct {
  OModel model;
  ll_config cfg;
  int n_envs, n_clips, frame_rate, margin, n_sub, prop_dim, obs_dim;
  double dt, policy_step, frame_step, mu_foot;
  double* frames;     /* [total][19] */
  int32_t *clip_off, *clip_len;
  double *max_steps, *prob, *avg_reward_sum, *avg_episode_len;
  int32_t *ob_off, *ob_cnt;   /* jump obstacles per clip (utils/obstacle.py), table rows = x, y, yaw, peak time */
  double* ob_table;
  OEnv* envs;
} OBatch;

OBatch* orc_create(const ll_config* cfg, const double* blob, int blob_len) {
  if (blob_len != LLM_BLOB_LEN || cfg->abi_version != LL_ABI_VERSION) return NULL;
  OBatch* B = (OBatch*)calloc(1, sizeof(OBatch));
  model_from_blob(blob, &B->model);
  B->cfg = *cfg;
  B->n_envs = cfg->n_envs;
  B->policy_step = 1.0 / cfg->control_freq;                      /* PLE:47 */
  B->dt = 1.0 / cfg->sim_freq;                                   /* PLE:49 */
  B->n_sub = (int)(B->policy_step / B->dt);                      /* PLE:52 */
  B->mu_foot = cfg->foot_lateral_friction * LLM_PLANE_FRICTION;  /* LR:304-308 x plane.urdf:5 */
  B->prop_dim = 0;
  for (int k = 0; k < 5 && cfg->prop_order[k] >= 0; k++)
    B->prop_dim += (cfg->prop_order[k] <= LL_PROP_JOINT_VEL) ? 12 : 3;    /* PLE:102-111 */
  B->obs_dim = LL_STACK * B->prop_dim + LL_STACK * 12 + LL_FUTURE_DIM;   /* PLE:114-121 */
  B->envs = (OEnv*)calloc(B->n_envs, sizeof(OEnv));
  return B;
}

void orc_destroy(OBatch* B) {
  if (!B) return;
  free(B->frames); free(B->clip_off); free(B->clip_len); free(B->max_steps); free(B->prob);
  free(B->avg_reward_sum); free(B->avg_episode_len); free(B->ob_off); free(B->ob_cnt); free(B->ob_table); free(B->envs); free(B);
}

int orc_obs_dim(const OBatch* B) { return B->obs_dim; }

/* ML:19-46 */
int orc_load_mocap(OBatch* B, const double* frames, const int32_t* clip_len, int n_clips, double frame_step) {
  size_t total = 0;
  for (int c = 0; c < n_clips; c++) total += clip_len[c];
  B->frames = (double*)malloc(total * 19 * sizeof(double));
  memcpy(B->frames, frames, total * 19 * sizeof(double));
  B->n_clips = n_clips;
  B->clip_len = (int32_t*)malloc(n_clips * 4); B->clip_off = (int32_t*)malloc(n_clips * 4);
  B->max_steps = (double*)malloc(n_clips * 8); B->prob = (double*)malloc(n_clips * 8);
  B->avg_reward_sum = (double*)calloc(n_clips, 8); B->avg_episode_len = (double*)calloc(n_clips, 8);
  B->frame_step = frame_step;                                            /* ML:33 */
  B->frame_rate = (int)(1.0 / frame_step);                               /* ML:34 */
  B->margin = (int)ceil(B->policy_step / frame_step) + B->frame_rate + 2; /* ML:35 */
  int off = 0;
  for (int c = 0; c < n_clips; c++) {
    B->clip_len[c] = clip_len[c]; B->clip_off[c] = off; off += clip_len[c];
    B->max_steps[c] = (clip_len[c] - B->margin) * frame_step / B->policy_step;   /* ML:45 */
    B->prob[c] = 1.0 / n_clips;                                                  /* ML:46 */
  }
  return 0;
}

static const double* clip_row(const OBatch* B, int clip, int fid) { return B->frames + ((size_t)B->clip_off[clip] + fid) * 19; }

/* utils/obstacle.py:6-33 tables (computed by the host with scipy find_peaks, exactly as the reference does) */
int orc_load_obstacles(OBatch* B, const int32_t* count, const double* table, int n_clips) {
  if (n_clips != B->n_clips) return LL_EINVAL;
  B->ob_off = (int32_t*)malloc(n_clips * 4); B->ob_cnt = (int32_t*)malloc(n_clips * 4);
  int total = 0;
  for (int c = 0; c < n_clips; c++) { B->ob_off[c] = total; B->ob_cnt[c] = count[c]; total += count[c]; }
  B->ob_table = (double*)malloc((total > 0 ? total : 1) * 4 * sizeof(double));
  if (total > 0) memcpy(B->ob_table, table, (size_t)total * 4 * sizeof(double));
  return 0;
}

/* PLE:262-268 + PLE:341-346: advance the episode's obstacle, then test the robot's collision shapes against the box
 * (PLE:184: half extents 0.025, 0.5, obstacle_height; PLE:191: centred on the ground at the peak's x, y; yawed).
 * DESIGN.md "jump obstacle": shapes are represented by box vertices, sphere centres and cylinder cap centres with their
 * radius; contact = within the contact margin (what getContactPoints reports). */
static double box_sdf(const double* P, double cx, double cy, double yaw, double hz) {
  double c = cos(yaw), s = sin(yaw), dx = P[0] - cx, dy = P[1] - cy;
  double lx = dx * c + dy * s, ly = dy * c - dx * s, lz = P[2];
  double qx = fabs(lx) - 0.025, qy = fabs(ly) - 0.5, qz = fabs(lz) - hz;
  double ox = qx > 0 ? qx : 0, oy = qy > 0 ? qy : 0, oz = qz > 0 ? qz : 0;
  double m = qx > qy ? qx : qy;
  if (qz > m) m = qz;
  return sqrt(ox * ox + oy * oy + oz * oz) + (m < 0 ? m : 0);
}
static int prim_hits_box(const OPrim* p, const double* Rw, const double* pw, double cx, double cy, double yaw, double hz) {
  double ctr[3], t[3], Rp[9];
  m3v(Rw, p->pos, t);
  for (int i = 0; i < 3; i++) ctr[i] = pw[i] + t[i];
  m3m(Rw, p->rot, Rp);
  if (p->type == LLM_PRIM_SPHERE) return box_sdf(ctr, cx, cy, yaw, hz) - p->size[0] < LLM_CONTACT_MARGIN;
  if (p->type == LLM_PRIM_BOX) {
    for (int v = 0; v < 8; v++) {
      double l[3] = {(v & 1) ? p->size[0] : -p->size[0], (v & 2) ? p->size[1] : -p->size[1], (v & 4) ? p->size[2] : -p->size[2]}, P[3];
      m3v(Rp, l, P);
      for (int i = 0; i < 3; i++) P[i] += ctr[i];
      if (box_sdf(P, cx, cy, yaw, hz) < LLM_CONTACT_MARGIN) return 1;
    }
    return 0;
  }
  for (int s2 = 0; s2 < 2; s2++) {
    double P[3];
    for (int i = 0; i < 3; i++) P[i] = ctr[i] + (s2 ? -1.0 : 1.0) * p->size[1] * Rp[3 * i + 2];
    if (box_sdf(P, cx, cy, yaw, hz) - p->size[0] < LLM_CONTACT_MARGIN) return 1;
  }
  return 0;
}
static int obstacle_contact(OBatch* B, OEnv* e) {
  int oc = B->ob_cnt ? B->ob_cnt[e->clip] : 0;
  if (oc <= 0) return 0;
  const double* tab = B->ob_table + (size_t)B->ob_off[e->clip] * 4;
  /* getContactPoints (PLE:343) reports the contacts of the last stepSimulation, i.e. with the box where it stood during the substeps;
   * _update_obstacle (PLE:229, :262-268) has already moved it on for the next step by then */
  double cx = tab[e->ob_id * 4], cy = tab[e->ob_id * 4 + 1], yaw = tab[e->ob_id * 4 + 2], hz = B->cfg.obstacle_height;
  while (e->ob_id < oc - 1 && e->time > tab[e->ob_id * 4 + 3] + 0.5) e->ob_id++;          /* PLE:264-265 */
  OKin K;
  kinematics(&B->model, e->state, NULL, &K);
  static const int links[LLM_N_LEG_PRIMS] = LLM_LEG_PRIM_LINKS;
  for (int i = 0; i < LLM_N_BASE_PRIMS; i++)
    if (prim_hits_box(&B->model.base_prims[i], K.Rw[0], K.pw[0], cx, cy, yaw, hz)) return 1;
  for (int l = 0; l < 4; l++)
    for (int i = 0; i < LLM_N_LEG_PRIMS; i++) {
      int body = 1 + 3 * l + links[i];
      if (prim_hits_box(&B->model.leg_prims[l][i], K.Rw[body], K.pw[body], cx, cy, yaw, hz)) return 1;
    }
  return 0;
}

/* PLE:276-297 _prepare_obs */
static void prepare_obs(OBatch* B, OEnv* e, const double* action, double* obs) {
  double fut4[4 * 37], fut[76], prop[LL_PROP_FRAME_MAX];
  orc_mocap_future(clip_row(B, e->clip, e->frame_id), e->frac, B->frame_step, fut4);     /* PLE:166/222 */
  for (int i = 0; i < 4; i++) {
    memcpy(fut + 19 * i, fut4 + 37 * i, 7 * 8);
    memcpy(fut + 19 * i + 7, fut4 + 37 * i + 13, 12 * 8);
  }
  int P = B->prop_dim;
  orc_prop(e->state, B->cfg.prop_order, prop);
  if (e->hist_n == 0) {                                  /* PLE:282-283, :287-288 deque pre-fill */
    for (int k = 0; k < LL_STACK; k++) { memcpy(e->hist_prop[k], prop, P * 8); memcpy(e->hist_act[k], action, 96); }
    e->hist_n = LL_STACK;
  } else {                                               /* PLE:284, :289 append to a maxlen-3 deque */
    for (int k = 0; k < LL_STACK - 1; k++) { memcpy(e->hist_prop[k], e->hist_prop[k + 1], P * 8); memcpy(e->hist_act[k], e->hist_act[k + 1], 96); }
    memcpy(e->hist_prop[LL_STACK - 1], prop, P * 8); memcpy(e->hist_act[LL_STACK - 1], action, 96);
  }
  for (int k = 0; k < LL_STACK; k++) {
    memcpy(obs + k * P, e->hist_prop[k], P * 8);
    memcpy(obs + LL_STACK * P + 12 * k, e->hist_act[k], 96);
  }
  orc_calc_future(e->state, e->state + 3, fut, obs + LL_STACK * P + LL_STACK * 12);
}

/* PLE:150-171 with explicit (clip, t0) -- the parity mode of SURVEY 8c; obs_out [obs_dim] */
int orc_reset_env(OBatch* B, int env, int clip, double t0, double* obs_out) {
  if (env < 0 || env >= B->n_envs || clip < 0 || clip >= B->n_clips) return LL_EINVAL;
  OEnv* e = &B->envs[env];
  e->clip = clip; e->time = t0; e->reward_sum = 0; e->ep_steps = 0; e->hist_n = 0; e->done_reason = 0; e->ob_id = 0;   /* PLE:179 */
  orc_mocap_locate(t0, B->frame_step, &e->frame_id, &e->frac);                              /* ML:52-53 */
  orc_mocap_interp(clip_row(B, clip, e->frame_id), clip_row(B, clip, e->frame_id + 1), e->frac, B->frame_step, e->kin);
  memcpy(e->state, e->kin, sizeof e->kin);                                                  /* PLE:162-163 */
  memset(&e->warm, 0, sizeof e->warm);
  double zero[12] = {0};
  prepare_obs(B, e, zero, obs_out);                                                         /* PLE:168-170 */
  return 0;
}

/* upper bound of the start-time window of a clip, ML:50 */
double orc_motion_duration(const OBatch* B, int clip) { return B->frame_step * (B->clip_len[clip] - B->margin - 1); }

/* LeggedRobot.apply_action, LR:119-148 (noise=None): the `forces=` handed to setJointMotorControlArray(TORQUE_CONTROL).
 * Pinned by golden G8 (tests/golden/pmc_config_golden.npz) at 1e-12. */
void orc_pd_torque(double kp, double kd, double max_tau, const double* q, const double* qd, const double* tgt_joint_pos, double* tau) {
  for (int i = 0; i < 12; i++) {
    double tgt = tgt_joint_pos[i];
    if (tgt > 3.0) tgt = 3.0;                                             /* LR:126-127 np.clip(+-3) */
    if (tgt < -3.0) tgt = -3.0;
    double t = kp * (tgt - q[i]) + kd * (0.0 - qd[i]);                    /* LR:139 (tgt_joint_vel = 0, LR:128) */
    if (t > max_tau) t = max_tau;                                         /* LR:140 clip_value */
    if (t < -max_tau) t = -max_tau;
    tau[i] = t;
  }
}

/*
 * PLE:195-245 for one env.  scripted_dyn / feet_* (nullable) replace the physics result / FK, which is
 * how the golden harness (fake BulletClient) drove the reference.
 */
int orc_step_env(OBatch* B, int env, const double* action, const double* scripted_dyn, const double* feet_dyn_in,
                 const double* feet_kin_in, double* obs_out, double* reward_out, int* done_out) {
  OEnv* e = &B->envs[env];
  e->ep_steps += 1;                                                       /* PLE:197 */
  double tgt[12], tau[12];
  for (int i = 0; i < 12; i++) tgt[i] = e->state[13 + i] + action[i];     /* PLE:199-200 (clipped inside apply_action, LR:126-127) */
  int bad = 0;
  /* PLE:182-193: the jump obstacle is a static body (createMultiBody, mass 0) the robot collides with: 0.05 x 1.0 x 2h box at the
   * pose the last reset / _update_obstacle gave it, Bullet's default lateral friction 0.5 */
  OTerrain OT;
  double ob_rec[8];
  const OTerrain* Tob = NULL;
  if (B->cfg.set_obstacle && B->ob_cnt && B->ob_cnt[e->clip] > 0) {
    const double* tab = B->ob_table + ((size_t)B->ob_off[e->clip] + e->ob_id) * 4;
    const double dx = e->state[0] - tab[0], dy = e->state[1] - tab[1], hz = B->cfg.obstacle_height;
    if (dx * dx + dy * dy < LLM_OBSTACLE_REACH * LLM_OBSTACLE_REACH) {
      const double rec[8] = {-0.025, 0.025, -0.5, 0.5, -hz, hz, 0.0, 0.0};
      memcpy(ob_rec, rec, sizeof rec);
      OTerrain t = {1, ob_rec, LLM_LINK_FRICTION / LLM_PLANE_FRICTION, 1, tab[0], tab[1], cos(tab[2]), sin(tab[2])};
      OT = t;
      Tob = &OT;
    }
  }
  for (int s = 0; s < B->n_sub; s++) {                                    /* PLE:202 */
    orc_pd_torque(B->cfg.kp, B->cfg.kd, B->cfg.max_tau, e->state + 13, e->state + 25, tgt, tau);   /* LR:137-141 */
    if (!scripted_dyn) {
      tl_warm = &e->warm;
      if (substep_terrain(&B->model, B->dt, B->cfg.solver_iterations, B->mu_foot, e->state, tau, NULL, Tob, NULL)) bad = 1;   /* PLE:206 */
      tl_warm = NULL;
    }
    orc_mocap_locate(e->time, B->frame_step, &e->frame_id, &e->frac);      /* PLE:208 (time BEFORE the increment, quirk Q2) */
    e->time += B->dt;                                                      /* PLE:210 */
  }
  if (scripted_dyn) memcpy(e->state, scripted_dyn, sizeof e->state);
  for (int i = 0; i < 37; i++) if (!isfinite(e->state[i])) bad = 1;
  orc_mocap_interp(clip_row(B, e->clip, e->frame_id), clip_row(B, e->clip, e->frame_id + 1), e->frac, B->frame_step, e->kin); /* PLE:217-218 */
  prepare_obs(B, e, action, obs_out);                                      /* PLE:221-227 (raw action, quirk Q3) */
  if (feet_dyn_in) memcpy(e->feet_dyn, feet_dyn_in, 96); else orc_fk_feet_model(&B->model, e->state, e->feet_dyn);   /* PLE:397 */
  if (feet_kin_in) memcpy(e->feet_kin, feet_kin_in, 96); else orc_fk_feet_model(&B->model, e->kin, e->feet_kin);     /* PLE:398 */
  double r = orc_reward(e->state, e->kin, e->feet_dyn, e->feet_kin, B->cfg.reward_weights);                          /* PLE:230 */
  e->reward_sum += r;                                                      /* PLE:231 */
  int reason = 0;
  if (orc_check_fall(e->state + 3)) reason |= LL_DONE_FALL;                                     /* PLE:338 */
  if (e->frame_id >= B->clip_len[e->clip] - B->margin - 1) reason |= LL_DONE_CLIP_END;          /* PLE:339, ML:168-172 */
  if (orc_check_diverged(e->state, e->kin)) reason |= LL_DONE_DIVERGED;                         /* PLE:340 */
  if (bad) reason |= LL_DONE_NONFINITE;
  if (B->cfg.set_obstacle && !bad && obstacle_contact(B, e)) reason |= LL_DONE_COLLISION;      /* PLE:341-346 */
  e->done_reason = reason;
  if (reason)                                                              /* PLE:235-240 */
#pragma omp critical(orc_table)
  {
    int c = e->clip;
    B->avg_reward_sum[c] = e->reward_sum / B->max_steps[c];
    B->avg_episode_len[c] = e->ep_steps / (B->max_steps[c] + 1);
    double sum = 0;
    for (int k = 0; k < B->n_clips; k++) { B->prob[k] = pow(1 - B->avg_reward_sum[k], B->cfg.prioritized_sample_factor); sum += B->prob[k]; }
    for (int k = 0; k < B->n_clips; k++) B->prob[k] /= sum;
  }
  *reward_out = r;
  *done_out = reason != 0;
  return 0;
}

/* one substep with EPMC terrain and push (tests/epmc parity): shapes [n][8] as in the kernel, push [3] or NULL */
int orc_substep_terrain(const OBatch* B, double* state, const double* tau, double mu_foot, int n_shapes, const double* shapes, double box_mu_scale,
                        const double* push, int32_t* n_contacts, double* lambda_out) {
  OSubstepDiag d;
  OTerrain T = {n_shapes, shapes, box_mu_scale};
  int rc = substep_terrain(&B->model, B->dt, B->cfg.solver_iterations, mu_foot, state, tau, &d, n_shapes > 0 ? &T : NULL, push);
  if (n_contacts) *n_contacts = d.n_contacts;
  if (lambda_out) memcpy(lambda_out, d.lambda, sizeof d.lambda);
  return rc;
}

/* tests: the contacts find_contacts keeps in the given configuration: rows [leg, candidate index 9 sub + jj, depth, P(3), n(3), mu, body]; returns how many */
int orc_list_contacts(const OBatch* B, const double* state, double mu_foot, int n_shapes, const double* shapes, double box_mu_scale, double* out11) {
  OTerrain T = {n_shapes, shapes, box_mu_scale};
  OKin K;
  OContact C[MAXC];
  kinematics(&B->model, state, NULL, &K);
  const int n = find_contacts(&B->model, &K, mu_foot, LLM_LINK_FRICTION * LLM_PLANE_FRICTION, n_shapes > 0 ? &T : NULL, C);
  for (int i = 0; i < n; i++) {
    double* o = out11 + 11 * i;
    o[0] = C[i].leg; o[1] = C[i].cand; o[2] = C[i].depth; memcpy(o + 3, C[i].P, 24); memcpy(o + 6, C[i].n, 24); o[9] = C[i].mu; o[10] = C[i].body;
  }
  return n;
}
/* SEPMC: one substep of two robots in one world; shapes0 / shapes1 = the terrain records within reach of each (the flag among them);
 * pair_rows8 (nullable): [2][8] = P 3, n 3 (from robot 1 to robot 0), depth, pair id of the shared rows.  Returns their number. */
int orc_substep_pair(const OBatch* B, double* state0, double* state1, const double* tau0, const double* tau1, double mu_foot, int n_shapes0,
                     const double* shapes0, int n_shapes1, const double* shapes1, double box_mu_scale, const double* push0, const double* push1,
                     double* pair_rows8) {
  OTerrain T0 = {n_shapes0, shapes0, box_mu_scale}, T1 = {n_shapes1, shapes1, box_mu_scale};
  const double mu2[2] = {mu_foot, mu_foot};
  return orc_substep_pair_model(&B->model, B->dt, B->cfg.solver_iterations, mu2, state0, state1, tau0, tau1, n_shapes0 > 0 ? &T0 : NULL,
                                n_shapes1 > 0 ? &T1 : NULL, push0, push1, pair_rows8);
}
/* SEPMC: the contact classes of both robots in the given configuration: out6 = robot 0 {static, flag, robot}, robot 1 {static, flag, robot} */
int orc_touch(const OBatch* B, const double* state0, const double* state1, int n_shapes0, const double* shapes0, int flag0, int n_shapes1,
              const double* shapes1, int flag1, int32_t* out6) {
  OTerrain T0 = {n_shapes0, shapes0, 1.0}, T1 = {n_shapes1, shapes1, 1.0};
  return orc_touch_model(&B->model, state0, state1, n_shapes0 > 0 ? &T0 : NULL, flag0, n_shapes1 > 0 ? &T1 : NULL, flag1, out6);
}
int orc_self_contacts(const OBatch* B, const double* state, double* rows8) {
  OKin K;
  kinematics(&B->model, state, NULL, &K);
  OSelf sc[MAX_SELF];
  int n = find_self_contacts(&B->model, &K, sc);
  for (int i = 0; i < n; i++) {
    rows8[8 * i] = sc[i].depth; rows8[8 * i + 1] = sc[i].pair;
    for (int k = 0; k < 3; k++) { rows8[8 * i + 2 + k] = sc[i].P[k]; rows8[8 * i + 5 + k] = sc[i].n[k]; }
  }
  return n;
}

/* accessors used by the tests */
void orc_get_state(const OBatch* B, int env, double* s37) { memcpy(s37, B->envs[env].state, 37 * 8); }
void orc_set_state(OBatch* B, int env, const double* s37) { memcpy(B->envs[env].state, s37, 37 * 8); }
void orc_get_ref_state(const OBatch* B, int env, double* s37) { memcpy(s37, B->envs[env].kin, 37 * 8); }
void orc_get_feet(const OBatch* B, int env, double* fd, double* fk) { memcpy(fd, B->envs[env].feet_dyn, 96); memcpy(fk, B->envs[env].feet_kin, 96); }
void orc_get_episode_info(const OBatch* B, int env, int32_t* clip, double* time, int32_t* steps, double* rsum, int32_t* reason, int32_t* frame_id, double* frac) {
  const OEnv* e = &B->envs[env];
  *clip = e->clip; *time = e->time; *steps = e->ep_steps; *rsum = e->reward_sum; *reason = e->done_reason; *frame_id = e->frame_id; *frac = e->frac;
}
void orc_get_sampling_table(const OBatch* B, double* prob, double* avg_r, double* avg_len) {
  memcpy(prob, B->prob, B->n_clips * 8); memcpy(avg_r, B->avg_reward_sum, B->n_clips * 8); memcpy(avg_len, B->avg_episode_len, B->n_clips * 8);
}
void orc_set_sampling_table(OBatch* B, const double* prob, const double* avg_r) {
  memcpy(B->prob, prob, B->n_clips * 8); memcpy(B->avg_reward_sum, avg_r, B->n_clips * 8);
}
void orc_get_margin(const OBatch* B, int32_t* margin, int32_t* frame_rate, double* max_steps) {
  *margin = B->margin; *frame_rate = B->frame_rate; memcpy(max_steps, B->max_steps, B->n_clips * 8);
}

/* thin wrappers taking the batch's model (ctypes-friendly) */
void orc_fk_feet(const OBatch* B, const double* state, double* feet) { orc_fk_feet_model(&B->model, state, feet); }
int orc_forward_dynamics(const OBatch* B, const double* state, const double* tau, double* acc18) {
  return orc_forward_dynamics_model(&B->model, state, tau, acc18);
}
int orc_substep(const OBatch* B, double* state, const double* tau, int32_t* n_contacts, double* lambda_out, double* acc_free) {
  OSubstepDiag d;
  int rc = orc_substep_model(&B->model, B->dt, B->cfg.solver_iterations, B->mu_foot, state, tau, &d);
  if (n_contacts) *n_contacts = d.n_contacts;
  if (lambda_out) memcpy(lambda_out, d.lambda, d.n_rows * sizeof(double));
  if (acc_free) memcpy(acc_free, d.acc_free, sizeof d.acc_free);
  return rc;
}
/* inverse-dynamics style check value: kinetic + potential energy of a state (tests) */
double orc_energy(const OBatch* B, const double* state) {
  const OModel* M = &B->model;
  OKin K;
  kinematics(M, state, NULL, &K);
  double E = 0;
  for (int b = 0; b < NB; b++) {
    double Iv[6];
    m6v(M->I6[b], K.v[b], Iv);
    double ke = 0;
    for (int i = 0; i < 6; i++) ke += 0.5 * K.v[b][i] * Iv[i];
    double cw[3];
    m3v(K.Rw[b], M->com[b], cw);
    E += ke + M->mass[b] * LLM_GRAVITY * (K.pw[b][2] + cw[2]);
  }
  return E;
}

/* linear momentum (world) and angular momentum about the world origin (tests) */
void orc_momentum(const OBatch* B, const double* state, double* out6) {
  const OModel* M = &B->model;
  OKin K;
  kinematics(M, state, NULL, &K);
  memset(out6, 0, 48);
  for (int b = 0; b < NB; b++) {
    double h[6], n[3], f[3], t[3];
    m6v(M->I6[b], K.v[b], h);              /* spatial momentum about the body origin, body coords */
    m3v(K.Rw[b], h, n);
    m3v(K.Rw[b], h + 3, f);
    v3cross(K.pw[b], f, t);
    for (int i = 0; i < 3; i++) { out6[i] += f[i]; out6[3 + i] += n[i] + t[i]; }
  }
}

/* step every env (index order), as bench.py's cpu_baseline leg times it */
int orc_step_all(OBatch* B, const double* actions, double* obs, double* reward, int32_t* done) {
  for (int i = 0; i < B->n_envs; i++) {
    int d;
    orc_step_env(B, i, actions + 12 * i, NULL, NULL, NULL, obs + (size_t)B->obs_dim * i, reward + i, &d);
    done[i] = d;
  }
  return 0;
}

/* the same over the host's cores (envs are independent; the per-clip table update is the one shared write and is serialised,
 * in whatever order the threads get there) -- only bench.py's cpu_baseline leg uses this */
int orc_step_all_mt(OBatch* B, const double* actions, double* obs, double* reward, int32_t* done, int n_threads) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int i = 0; i < B->n_envs; i++) {
    int d;
    orc_step_env(B, i, actions + 12 * i, NULL, NULL, NULL, obs + (size_t)B->obs_dim * i, reward + i, &d);
    done[i] = d;
  }
  return 0;
}
