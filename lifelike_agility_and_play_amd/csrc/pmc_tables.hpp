// pmc_tables.hpp -- host side: model blob (include/llenv_model.h) + ll_config -> kernel constant tables and scalars.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/llenv.h"
#include "../../include/llenv_model.h"
#include "pmc_params.hpp"

struct PmcPrimView {
  int type;
  const double *size, *pos, *rot;
};
static inline PmcPrimView pmc_prim(const double* b) {
  PmcPrimView p;
  p.type = (int)b[0]; p.size = b + 1; p.pos = b + 4; p.rot = b + 7;
  return p;
}

// Returns "" on success, else a message.  legc: [LC_COUNT][4], basec: [BC_COUNT]
static inline std::string pmc_build_tables(const double* blob, int blob_len, std::vector<float>& legc, std::vector<float>& basec) {
  if (blob_len != LLM_BLOB_LEN) return "model blob has the wrong length";
  legc.assign(LC_COUNT * 4, 0.0f);
  basec.assign(BC_COUNT, 0.0f);
  auto L = [&](int field, int leg) -> float& { return legc[field * 4 + leg]; };
  static const int links[LLM_N_LEG_PRIMS] = LLM_LEG_PRIM_LINKS;
  (void)links;
  for (int l = 0; l < 4; l++) {
    for (int k = 0; k < 3; k++) {
      int i = 3 * l + k;
      const double* ax = blob + LLM_OFF_JOINT_AXIS + 3 * i;
      // the kernel's closed-form leg kinematics assume hip axis +x, thigh/shank axis -y (true for max.urdf)
      const double want[3] = {k == 0 ? 1.0 : 0.0, k == 0 ? 0.0 : -1.0, 0.0};
      for (int c = 0; c < 3; c++)
        if (fabs(ax[c] - want[c]) > 1e-9) return "unsupported joint axis layout (kernel expects hip +x, thigh/shank -y)";
      for (int c = 0; c < 3; c++) {
        L((k == 0 ? LC_R1 : (k == 1 ? LC_R2 : LC_R3)) + c, l) = (float)blob[LLM_OFF_JOINT_ORIGIN + 3 * i + c];
        L(LC_COM + 3 * k + c, l) = (float)blob[LLM_OFF_LINK_COM + 3 * i + c];
      }
      L(LC_M + k, l) = (float)blob[LLM_OFF_LINK_MASS + i];
      const double* I = blob + LLM_OFF_LINK_INERTIA + 9 * i;
      L(LC_IC + 6 * k + 0, l) = (float)I[0]; L(LC_IC + 6 * k + 1, l) = (float)I[1]; L(LC_IC + 6 * k + 2, l) = (float)I[2];
      L(LC_IC + 6 * k + 3, l) = (float)I[4]; L(LC_IC + 6 * k + 4, l) = (float)I[5]; L(LC_IC + 6 * k + 5, l) = (float)I[8];
      L(LC_QLO + k, l) = (float)blob[LLM_OFF_Q_LO + i];
      L(LC_QHI + k, l) = (float)blob[LLM_OFF_Q_HI + i];
      L(LC_JDAMP + k, l) = (float)blob[LLM_OFF_DAMPING + i];
    }
    for (int c = 0; c < 3; c++) L(LC_FOOT + c, l) = (float)blob[LLM_OFF_FOOT_POS + 3 * l + c];
    L(LC_BSX, l) = (l & 1) ? 1.0f : -1.0f;
    L(LC_BSY, l) = (l & 2) ? 1.0f : -1.0f;
    int handle = (l == 0) ? 1 : (l == 2 ? 2 : -1);
    if (handle > 0) {
      PmcPrimView h = pmc_prim(blob + LLM_OFF_BASE_PRIMS + handle * LLM_PRIM_STRIDE);
      if (h.type != LLM_PRIM_SPHERE) return "base primitive 1/2 must be the handle spheres";
      for (int c = 0; c < 3; c++) L(LC_HANDLE + c, l) = (float)h.pos[c];
      L(LC_HANDLE + 3, l) = (float)h.size[0];
      L(LC_HAS_HANDLE, l) = 1.0f;
    }
    // leg primitive slots: 0 hip cyl | 1 thigh box, 2 thigh cyl, 3 thigh cyl, 4 wheel cyl | 5 shank box, 6 foot sphere
    static const int slot_field[LLM_N_LEG_PRIMS] = {LC_HIPCYL, LC_THBOX, LC_THCYL0, LC_THCYL1, LC_WHEEL, LC_SHBOX, LC_FOOTSPH};
    static const int slot_type[LLM_N_LEG_PRIMS] = {LLM_PRIM_CYL, LLM_PRIM_BOX, LLM_PRIM_CYL, LLM_PRIM_CYL, LLM_PRIM_CYL, LLM_PRIM_BOX, LLM_PRIM_SPHERE};
    for (int s = 0; s < LLM_N_LEG_PRIMS; s++) {
      PmcPrimView p = pmc_prim(blob + LLM_OFF_LEG_PRIMS + (l * LLM_N_LEG_PRIMS + s) * LLM_PRIM_STRIDE);
      if (p.type != slot_type[s]) return "leg primitive slots do not match the expected layout";
      int f = slot_field[s];
      for (int c = 0; c < 3; c++) L(f + c, l) = (float)p.pos[c];
      if (p.type == LLM_PRIM_SPHERE) {
        L(f + 3, l) = (float)p.size[0];
      } else if (p.type == LLM_PRIM_BOX) {
        for (int a = 0; a < 3; a++)
          for (int c = 0; c < 3; c++) L(f + 3 + 3 * a + c, l) = (float)(p.rot[3 * c + a] * p.size[a]);   // column a of rot, scaled
      } else {
        for (int c = 0; c < 3; c++) { L(f + 3 + c, l) = (float)p.rot[3 * c + 2]; L(f + 6 + c, l) = (float)p.rot[3 * c + 0]; }
        L(f + 9, l) = (float)p.size[0];
        L(f + 10, l) = (float)p.size[1];
      }
    }
  }
  // self-collision capsules: thigh = long axis of the thigh box, radius its larger half thickness; shank = from the
  // upper end of the shank box to the foot centre, radius the mean of the box's half thickness and the foot radius
  for (int l = 0; l < 4; l++) {
    PmcPrimView tb = pmc_prim(blob + LLM_OFF_LEG_PRIMS + (l * LLM_N_LEG_PRIMS + 1) * LLM_PRIM_STRIDE);
    PmcPrimView sb = pmc_prim(blob + LLM_OFF_LEG_PRIMS + (l * LLM_N_LEG_PRIMS + 5) * LLM_PRIM_STRIDE);
    PmcPrimView ft = pmc_prim(blob + LLM_OFF_LEG_PRIMS + (l * LLM_N_LEG_PRIMS + 6) * LLM_PRIM_STRIDE);
    double d[3], len = 0;
    for (int c = 0; c < 3; c++) { d[c] = ft.pos[c] - sb.pos[c]; len += d[c] * d[c]; }
    len = sqrt(len);
    for (int c = 0; c < 3; c++) {
      L(LC_CAPS + 0 + c, l) = (float)(tb.pos[c] + tb.size[0] * tb.rot[3 * c]);
      L(LC_CAPS + 3 + c, l) = (float)(tb.pos[c] - tb.size[0] * tb.rot[3 * c]);
      L(LC_CAPS + 6 + c, l) = (float)(sb.pos[c] - sb.size[0] * d[c] / len);
      L(LC_CAPS + 9 + c, l) = (float)ft.pos[c];
    }
    L(LC_CAPS + 12, l) = (float)(tb.size[1] > tb.size[2] ? tb.size[1] : tb.size[2]);
    L(LC_CAPS + 13, l) = (float)(0.5 * ((sb.size[1] > sb.size[2] ? sb.size[1] : sb.size[2]) + ft.size[0]));
  }
  // robot-robot collision (SEPMC): the body box as two capsules along its long axis, radius = its half height,
  // side by side so that they span its width; legs 0, 2 hold the first, legs 1, 3 the second
  {
    PmcPrimView bb = pmc_prim(blob + LLM_OFF_BASE_PRIMS);
    const double r = bb.size[2], hx = bb.size[0] - r, oy = bb.size[1] - r;
    for (int l = 0; l < 4; l++) {
      const double sy = (l & 1) ? -oy : oy;
      for (int c = 0; c < 3; c++) {
        L(LC_TRUNKCAP + 0 + c, l) = (float)(bb.pos[c] + hx * bb.rot[3 * c] + sy * bb.rot[3 * c + 1]);
        L(LC_TRUNKCAP + 3 + c, l) = (float)(bb.pos[c] - hx * bb.rot[3 * c] + sy * bb.rot[3 * c + 1]);
      }
      L(LC_TRUNKCAP + 6, l) = (float)r;
    }
  }
  // base
  double m = blob[LLM_OFF_BASE_MASS];
  const double* c = blob + LLM_OFF_BASE_COM;
  const double* I = blob + LLM_OFF_BASE_INERTIA;
  basec[BC_MASS] = (float)m;
  for (int k = 0; k < 3; k++) { basec[BC_H + k] = (float)(m * c[k]); basec[BC_COM + k] = (float)c[k]; }
  double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  basec[BC_IO + 0] = (float)(I[0] + m * (cc - c[0] * c[0])); basec[BC_IO + 1] = (float)(I[1] - m * c[0] * c[1]);
  basec[BC_IO + 2] = (float)(I[2] - m * c[0] * c[2]); basec[BC_IO + 3] = (float)(I[4] + m * (cc - c[1] * c[1]));
  basec[BC_IO + 4] = (float)(I[5] - m * c[1] * c[2]); basec[BC_IO + 5] = (float)(I[8] + m * (cc - c[2] * c[2]));
  basec[BC_ICOM + 0] = (float)I[0]; basec[BC_ICOM + 1] = (float)I[1]; basec[BC_ICOM + 2] = (float)I[2];
  basec[BC_ICOM + 3] = (float)I[4]; basec[BC_ICOM + 4] = (float)I[5]; basec[BC_ICOM + 5] = (float)I[8];
  PmcPrimView bx = pmc_prim(blob + LLM_OFF_BASE_PRIMS);
  if (bx.type != LLM_PRIM_BOX) return "base primitive 0 must be the body box";
  for (int k = 0; k < 3; k++) basec[BC_BOX + k] = (float)bx.pos[k];
  for (int a = 0; a < 3; a++)
    for (int k = 0; k < 3; k++) basec[BC_BOX + 3 + 3 * a + k] = (float)(bx.rot[3 * k + a] * bx.size[a]);
  return "";
}

// Contact candidate table (pmc_params.hpp CandField): 28 points per leg, 7 per sub-lane, laid out so that candidates jj 0..3
// of a sub-lane sit on one link (group A), jj 4..5 on another (B), jj 6 on a third (C):
//   sub 0: foot, shank box v0 v1 v2 | wheel caps +,- | shank box v3
//   sub 1: shank box v4..v7         | thigh cylinder 0 caps | body box vertex (leg, z-)
//   sub 2: thigh box v0..v3         | thigh cylinder 1 caps | body box vertex (leg, z+)
//   sub 3: thigh box v4..v7         | hip cylinder caps     | handle sphere (legs 0 and 2; invalid elsewhere)
static inline void pmc_put_cand(std::vector<float>& t, int lane16, int jj, const double* A, const double* ax, const double* fb, double r, int link, int kind) {
  for (int c = 0; c < 3; c++) {
    t[(jj * CF_WORDS + CF_A + c) * 16 + lane16] = (float)A[c];
    t[(jj * CF_WORDS + CF_AX + c) * 16 + lane16] = ax ? (float)ax[c] : 0.0f;
    t[(jj * CF_WORDS + CF_FB + c) * 16 + lane16] = fb ? (float)fb[c] : 0.0f;
  }
  t[(jj * CF_WORDS + CF_R) * 16 + lane16] = (float)r;
  t[(jj * CF_WORDS + CF_LINK) * 16 + lane16] = (float)link;
  t[(jj * CF_WORDS + CF_KIND) * 16 + lane16] = (float)kind;
}
static inline void pmc_box_vertex(const PmcPrimView& p, int v, double* A) {
  for (int c = 0; c < 3; c++) {
    A[c] = p.pos[c];
    for (int a = 0; a < 3; a++) A[c] += (((v >> a) & 1) ? 1.0 : -1.0) * p.size[a] * p.rot[3 * c + a];
  }
}
static inline void pmc_cyl_cap(const PmcPrimView& p, int s, double* A, double* ax, double* fb) {
  for (int c = 0; c < 3; c++) {
    ax[c] = p.rot[3 * c + 2];
    fb[c] = p.rot[3 * c + 0];
    A[c] = p.pos[c] + (s ? -1.0 : 1.0) * p.size[1] * ax[c];
  }
}
static inline std::string pmc_build_cand_table(const double* blob, std::vector<float>& t) {
  t.assign(CAND_TABLE_WORDS * 16, 0.0f);
  PmcPrimView bbox = pmc_prim(blob + LLM_OFF_BASE_PRIMS);
  for (int l = 0; l < 4; l++) {
    PmcPrimView lp[LLM_N_LEG_PRIMS];
    for (int s = 0; s < LLM_N_LEG_PRIMS; s++) lp[s] = pmc_prim(blob + LLM_OFF_LEG_PRIMS + (l * LLM_N_LEG_PRIMS + s) * LLM_PRIM_STRIDE);
    double A[3], ax[3], fb[3];
    const int L0 = l * 4;
    // sub 0
    pmc_put_cand(t, L0 + 0, 0, lp[6].pos, nullptr, nullptr, lp[6].size[0], 3, 1);
    for (int v = 0; v < 3; v++) { pmc_box_vertex(lp[5], v, A); pmc_put_cand(t, L0 + 0, 1 + v, A, nullptr, nullptr, 0.0, 3, 0); }
    for (int s = 0; s < 2; s++) { pmc_cyl_cap(lp[4], s, A, ax, fb); pmc_put_cand(t, L0 + 0, 4 + s, A, ax, fb, lp[4].size[0], 2, 0); }
    pmc_box_vertex(lp[5], 3, A); pmc_put_cand(t, L0 + 0, 6, A, nullptr, nullptr, 0.0, 3, 0);
    // sub 1
    for (int v = 0; v < 4; v++) { pmc_box_vertex(lp[5], 4 + v, A); pmc_put_cand(t, L0 + 1, v, A, nullptr, nullptr, 0.0, 3, 0); }
    for (int s = 0; s < 2; s++) { pmc_cyl_cap(lp[2], s, A, ax, fb); pmc_put_cand(t, L0 + 1, 4 + s, A, ax, fb, lp[2].size[0], 2, 0); }
    pmc_box_vertex(bbox, l, A); pmc_put_cand(t, L0 + 1, 6, A, nullptr, nullptr, 0.0, 0, 0);
    // sub 2
    for (int v = 0; v < 4; v++) { pmc_box_vertex(lp[1], v, A); pmc_put_cand(t, L0 + 2, v, A, nullptr, nullptr, 0.0, 2, 0); }
    for (int s = 0; s < 2; s++) { pmc_cyl_cap(lp[3], s, A, ax, fb); pmc_put_cand(t, L0 + 2, 4 + s, A, ax, fb, lp[3].size[0], 2, 0); }
    pmc_box_vertex(bbox, l + 4, A); pmc_put_cand(t, L0 + 2, 6, A, nullptr, nullptr, 0.0, 0, 0);
    // sub 3
    for (int v = 0; v < 4; v++) { pmc_box_vertex(lp[1], 4 + v, A); pmc_put_cand(t, L0 + 3, v, A, nullptr, nullptr, 0.0, 2, 0); }
    for (int s = 0; s < 2; s++) { pmc_cyl_cap(lp[0], s, A, ax, fb); pmc_put_cand(t, L0 + 3, 4 + s, A, ax, fb, lp[0].size[0], 1, 0); }
    if (l == 0 || l == 2) {
      PmcPrimView h = pmc_prim(blob + LLM_OFF_BASE_PRIMS + (l == 0 ? 1 : 2) * LLM_PRIM_STRIDE);
      pmc_put_cand(t, L0 + 3, 6, h.pos, nullptr, nullptr, h.size[0], 0, 0);
    } else {
      const double z3[3] = {0, 0, 0};
      pmc_put_cand(t, L0 + 3, 6, z3, nullptr, nullptr, 0.0, -1, 0);
    }
    // the link each sub-lane owns in the dynamics (pmc_step.hpp own_link): LK_WORDS constants per lane
    for (int sub = 0; sub < 3; sub++) {
      const int i = l * 3 + sub;
      const double* I = blob + LLM_OFF_LINK_INERTIA + 9 * i;
      const double v[LK_WORDS] = {blob[LLM_OFF_LINK_MASS + i], blob[LLM_OFF_LINK_COM + 3 * i], blob[LLM_OFF_LINK_COM + 3 * i + 1], blob[LLM_OFF_LINK_COM + 3 * i + 2],
                                  I[0], I[1], I[2], I[4], I[5], I[8]};
      for (int w = 0; w < LK_WORDS; w++) t[(LK_BASE + w) * 16 + L0 + sub] = (float)v[w];
    }
    // candidate 7 (terrain only): a sphere of the capsule radius at 1/3 and 2/3 of the shank axis (subs 0, 1) and of the
    // thigh axis (subs 2, 3) -- what meets a hurdle or step edge between the end points of a link
    {
      double d[3], len = 0, sa[3], sbv[3], ta[3], tb2[3];
      for (int c = 0; c < 3; c++) { d[c] = lp[6].pos[c] - lp[5].pos[c]; len += d[c] * d[c]; }
      len = sqrt(len);
      for (int c = 0; c < 3; c++) {
        sa[c] = lp[5].pos[c] - lp[5].size[0] * d[c] / len; sbv[c] = lp[6].pos[c];
        ta[c] = lp[1].pos[c] + lp[1].size[0] * lp[1].rot[3 * c]; tb2[c] = lp[1].pos[c] - lp[1].size[0] * lp[1].rot[3 * c];
      }
      const double rs = 0.5 * ((lp[5].size[1] > lp[5].size[2] ? lp[5].size[1] : lp[5].size[2]) + lp[6].size[0]);
      const double rt = lp[1].size[1] > lp[1].size[2] ? lp[1].size[1] : lp[1].size[2];
      for (int sub = 0; sub < 4; sub++) {
        const double f = (sub & 1) ? 2.0 / 3.0 : 1.0 / 3.0;
        double A2[3];
        for (int c = 0; c < 3; c++) A2[c] = sub < 2 ? sa[c] + f * (sbv[c] - sa[c]) : ta[c] + f * (tb2[c] - ta[c]);
        pmc_put_cand(t, L0 + sub, 7, A2, nullptr, nullptr, sub < 2 ? rs : rt, sub < 2 ? 3 : 2, 0);
      }
    }
  }
  return "";
}

// the ERPs as the kernel uses them, from what the spec switches say (LLM_SPEC_ERP, _LIMIT_ERP, _ERP_DEEP: "< 0" = follow the contact ERP / no second ERP)
static inline void pmc_resolve_erps(StepParams& P) {
  P.limit_erp = P.spec_limit_erp >= 0.0f ? P.spec_limit_erp : P.erp;
  P.erp_deep = P.spec_erp_deep >= 0.0f ? P.spec_erp_deep : P.erp;
  P.limit_erp_deep = P.spec_limit_erp_deep >= 0.0f ? P.spec_limit_erp_deep : (P.spec_erp_deep >= 0.0f ? P.spec_erp_deep : P.limit_erp);
}

// scalar part of StepParams from the reference-style config; returns "" or an error
static inline std::string pmc_fill_params(const ll_config& cfg, StepParams& P) {
  if (cfg.abi_version != LL_ABI_VERSION) return "ll_config.abi_version mismatch";
  if (cfg.n_envs <= 0) return "n_envs must be positive";
  if (!(cfg.control_freq > 0) || !(cfg.sim_freq > 0)) return "control_freq and sim_freq must be positive";
  memset(&P, 0, sizeof P);
  P.n_steps = 1;
  P.n_envs = cfg.n_envs;
  P.policy_step = 1.0 / cfg.control_freq;                 // PLE:47
  P.dt_d = 1.0 / cfg.sim_freq;                            // PLE:49
  P.n_sub = (int)(P.policy_step / P.dt_d);                // PLE:52
  if (P.n_sub < 1) return "sim_freq must be >= control_freq";
  P.dt = (float)P.dt_d;
  P.n_iter = cfg.solver_iterations > 0 ? cfg.solver_iterations : 10;   // LR:261
  P.auto_reset = cfg.auto_reset;
  P.keep_term_obs = (cfg.auto_reset && cfg.keep_terminal_obs) ? 1 : 0;
  P.kp = (float)cfg.kp; P.kd = (float)cfg.kd; P.max_tau = (float)cfg.max_tau;
  P.mu_foot = (float)(cfg.foot_lateral_friction * LLM_PLANE_FRICTION);   // LR:304-308 x plane.urdf:5
  P.mu_link = (float)(LLM_LINK_FRICTION * LLM_PLANE_FRICTION);
  P.gravity = (float)LLM_GRAVITY;
  P.link_damping = (float)LLM_LINK_DAMPING;
  P.erp = (float)LLM_ERP;
  P.margin_dist = (float)LLM_CONTACT_MARGIN;
  P.limit_gate = (float)LLM_LIMIT_GATE;
  P.self_collision = 1.0f;
  P.max_depen = (float)LLM_MAX_DEPEN_SPEED; P.self_margin = (float)LLM_SELF_MARGIN;
  P.max_contacts = LLM_MAX_CONTACTS_PER_LEG; P.max_self = LLM_MAX_SELF;
  P.self_friction = (float)LLM_SELF_FRICTION; P.pair_friction = (float)LLM_PAIR_FRICTION; P.max_pair = LLM_MAX_PAIR; P.leg_edges = LLM_LEG_EDGES;
  P.friction_mode = LLM_FRICTION_MODE;
  P.max_coord_vel = (float)LLM_MAX_COORD_VEL;
  P.limit_speculative = LLM_LIMIT_SPECULATIVE;
  P.spec_limit_erp = (float)LLM_LIMIT_ERP; P.spec_erp_deep = (float)LLM_ERP_DEEP; P.spec_limit_erp_deep = (float)LLM_LIMIT_ERP_DEEP; P.erp_deep_below = (float)LLM_ERP_DEEP_BELOW;
  pmc_resolve_erps(P);
  double sw = 0;
  for (int i = 0; i < 5; i++) sw += cfg.reward_weights[i];               // PLE:365
  if (!(sw > 0)) return "reward_weights must sum to a positive number";
  for (int i = 0; i < 5; i++) P.rw[i] = (float)(cfg.reward_weights[i] / sw);
  int off = 0;
  for (int k = 0; k < 5; k++) P.prop_off[k] = -1;
  for (int k = 0; k < 5 && cfg.prop_order[k] >= 0; k++) {               // PLE:101-113
    int id = cfg.prop_order[k];
    if (id > 4 || P.prop_off[id] >= 0) return "bad prop_order";
    P.prop_off[id] = off;
    off += (id <= LL_PROP_JOINT_VEL) ? 12 : 3;
  }
  if (off == 0) return "prop_type must not be empty";
  P.prop_dim = off;
  P.obs_dim = LL_STACK * off + LL_STACK * 12 + LL_FUTURE_DIM;           // PLE:114-121
  P.sample_factor = cfg.prioritized_sample_factor;
  P.set_obstacle = cfg.set_obstacle ? 1 : 0;
#if defined(PMC_ABLATION)
  if (const char* dbg = getenv("LL_DEBUG_FLAGS")) P.debug_flags = atoi(dbg);
#endif
  P.ob_half_height = (float)cfg.obstacle_height;                          // PLE:184 halfExtents z
  P.seed = cfg.seed;
  return "";
}

// mocap-derived scalars (ML:33-35)
static inline void pmc_fill_mocap(StepParams& P, int n_clips, double frame_step) {
  P.n_clips = n_clips;
  P.frame_step = frame_step;
  P.frame_rate = (int)(1.0 / frame_step);
  P.margin = (int)ceil(P.policy_step / frame_step) + P.frame_rate + 2;
}

// ll_set_spec_param / ll_get_spec_param (include/llenv_model.h LLM_SPEC_*): the constants of the physics spec that are this build's own
// choice, movable at run time for the deviation study.  Returns "" or a message.
inline std::string pmc_set_spec_param(StepParams& P, int id, double v) {
  switch (id) {
    case LLM_SPEC_LIMIT_GATE: P.limit_gate = (float)v; break;
    case LLM_SPEC_MAX_DEPEN_SPEED: P.max_depen = (float)v; break;
    case LLM_SPEC_LINK_DAMPING: P.link_damping = (float)v; break;
    case LLM_SPEC_MAX_CONTACTS_PER_LEG:
      if (!(v >= 1 && v <= LLM_MAX_CONTACTS_PER_LEG)) return "max contacts per leg must be 1..4";
      P.max_contacts = (int)v; break;
    case LLM_SPEC_SELF_COLLISION: P.self_collision = v > 0.5 ? 1.0f : 0.0f; break;
    case LLM_SPEC_SELF_MARGIN: P.self_margin = (float)v; break;
    case LLM_SPEC_MAX_SELF:
      if (!(v >= 0 && v <= LLM_MAX_SELF)) return "self-collision rows per robot must be 0..2";
      P.max_self = (int)v; break;
    case LLM_SPEC_ERP: P.erp = (float)v; pmc_resolve_erps(P); break;
    case LLM_SPEC_LIMIT_ERP: P.spec_limit_erp = (float)v; pmc_resolve_erps(P); break;
    case LLM_SPEC_ERP_DEEP: P.spec_erp_deep = (float)v; pmc_resolve_erps(P); break;
    case LLM_SPEC_ERP_DEEP_BELOW: P.erp_deep_below = (float)v; break;
    case LLM_SPEC_LIMIT_ERP_DEEP: P.spec_limit_erp_deep = (float)v; pmc_resolve_erps(P); break;
    case LLM_SPEC_LIMIT_SPECULATIVE:
      if (!(v == 0.0 || v == 1.0)) return "limit_speculative must be 0 or 1";
      P.limit_speculative = (int)v; break;
    case LLM_SPEC_CONTACT_MARGIN: P.margin_dist = (float)v; break;
    case LLM_SPEC_SELF_FRICTION:
      if (!(v >= 0.0 && v <= 4.0)) return "self_friction must be in [0, 4]";
      P.self_friction = (float)v; break;
    case LLM_SPEC_PAIR_FRICTION:
      if (!(v >= 0.0 && v <= 4.0)) return "pair_friction must be in [0, 4]";
      P.pair_friction = (float)v; break;
    case LLM_SPEC_MAX_PAIR:
      if (!(v >= 0 && v <= LLM_MAX_PAIR_CAP)) return "robot-robot rows per pair must be 0..4";
      P.max_pair = (int)v; break;
    case LLM_SPEC_LEG_EDGES:
      if (!(v == 0.0 || v == 1.0)) return "leg_edges must be 0 or 1";
      P.leg_edges = (int)v; break;
    case LLM_SPEC_FRICTION_KEEP:
    case LLM_SPEC_WARM_START:
      if (v != 0.0) return "this switch exists in the oracle only (tools/deviation_table.py reports what it is worth)";
      break;
    case LLM_SPEC_TRUNK_EDGES:
      if (v != 1.0) return "this switch exists in the oracle only (a test instrument)";
      break;
    case LLM_SPEC_SELECT_EPS:
      if (v != LLM_SELECT_EPS) return "this switch exists in the oracle only (a test instrument)";
      break;
    case LLM_SPEC_FRICTION_DIRS:
      if (!(v == 0.0 || v == 1.0)) return "friction_dirs must be 0 or 1";
      P.friction_dirs = (int)v; break;
    case LLM_SPEC_FRICTION_MODE:
      if (!(v == 0.0 || v == 2.0)) return "friction_mode: the engine has 0 (pyramid, the spec) and 2 (cone-coupled, btMultiBodyConstraintSolver's published default); 1 and 3 exist in the oracle only";
      P.friction_mode = (int)v; break;
    case LLM_SPEC_MAX_COORD_VEL:
      if (!(v > 0.0)) return "max_coord_vel must be positive (1e30: no clip)";
      P.max_coord_vel = (float)(v < 3.0e38 ? v : 3.0e38); break;
    case LLM_SPEC_ROW_ORDER: case LLM_SPEC_GYRO:
      if (id == LLM_SPEC_GYRO && v == 1.0) break;
      return "this switch exists in the oracle only (round-3 audit against Bullet's published solver: profiles/r03_deviation_table.md)";
    default: return "unknown spec parameter id";
  }
  return "";
}
inline double pmc_get_spec_param(const StepParams& P, int id) {
  switch (id) {
    case LLM_SPEC_LIMIT_GATE: return P.limit_gate;
    case LLM_SPEC_MAX_DEPEN_SPEED: return P.max_depen;
    case LLM_SPEC_LINK_DAMPING: return P.link_damping;
    case LLM_SPEC_MAX_CONTACTS_PER_LEG: return P.max_contacts;
    case LLM_SPEC_SELF_COLLISION: return P.self_collision;
    case LLM_SPEC_SELF_MARGIN: return P.self_margin;
    case LLM_SPEC_MAX_SELF: return P.max_self;
    case LLM_SPEC_ERP: return P.erp;
    case LLM_SPEC_CONTACT_MARGIN: return P.margin_dist;
    case LLM_SPEC_TRUNK_EDGES: return 1.0;
    case LLM_SPEC_SELECT_EPS: return LLM_SELECT_EPS;
    case LLM_SPEC_FRICTION_DIRS: return P.friction_dirs;
    case LLM_SPEC_FRICTION_MODE: return P.friction_mode;
    case LLM_SPEC_MAX_COORD_VEL: return P.max_coord_vel;
    case LLM_SPEC_LIMIT_ERP: return P.spec_limit_erp;
    case LLM_SPEC_ERP_DEEP: return P.spec_erp_deep;
    case LLM_SPEC_ERP_DEEP_BELOW: return P.erp_deep_below;
    case LLM_SPEC_LIMIT_ERP_DEEP: return P.spec_limit_erp_deep;
    case LLM_SPEC_MAX_PAIR: return P.max_pair;
    case LLM_SPEC_SELF_FRICTION: return P.self_friction;
    case LLM_SPEC_PAIR_FRICTION: return P.pair_friction;
    case LLM_SPEC_LEG_EDGES: return P.leg_edges;
    case LLM_SPEC_LIMIT_SPECULATIVE: return P.limit_speculative;
    case LLM_SPEC_GYRO: return 1.0;
    default: return 0.0;
  }
}
