// sepmc_engine.hpp -- host-side engine behind include/llenv_sepmc.h.  It owns a PmcEngine with 2 * n_arenas rows (one per
// robot: model tables, state / obs / reward / done / action buffers, counters, backend) and adds the SEPMC per-row buffers.
// The backend must provide launch_sepmc_step / launch_sepmc_reset next to the PMC and EPMC launches.
#pragma once
#include <string.h>

#include "../../include/llenv_sepmc.h"
#include "pmc_engine.hpp"
#include "sepmc_step.hpp"

template <class BK>
struct SepmcEngine {
  PmcEngine<BK> base;
  ll_sepmc_config cfg;
  SepmcParams S;
  bool have_reset = false, reset_scripted = false;
  float *d_scr_state = nullptr, *d_scr_frac = nullptr, *d_scr_draws = nullptr, *d_reset_draws = nullptr, *d_prev_orn = nullptr;
  uint8_t *d_scr_hit = nullptr, *d_scr_vis = nullptr;
  int32_t *d_scr_contacts = nullptr, *d_row_ids = nullptr;
  int scr_draws_cap = 0, pending_step_draws = 0;

  static ll_config base_config(const ll_sepmc_config& c) {
    ll_config b;
    memset(&b, 0, sizeof b);
    b.abi_version = LL_ABI_VERSION;
    b.n_envs = 2 * c.n_arenas; b.device = c.device; b.auto_reset = c.auto_reset;
    b.control_freq = c.control_freq; b.sim_freq = 500.0;                   // CTG:53
    b.kp = c.kp; b.kd = c.kd; b.max_tau = c.max_tau;
    b.foot_lateral_friction = c.friction_range[0];
    for (int i = 0; i < 5; i++) { b.reward_weights[i] = 1.0; b.prop_order[i] = c.prop_order[i]; }
    b.solver_iterations = c.solver_iterations;
    b.seed = c.seed;
    return b;
  }

  SepmcEngine(const ll_sepmc_config& c, const double* blob, int blob_len, const double* init37) : base(base_config(c), blob, blob_len), cfg(c) {
    if (c.abi_version != LL_ABI_VERSION) throw PmcError(LL_EINVAL, "ll_sepmc_config.abi_version mismatch");
    if (c.n_arenas <= 0 || c.max_steps <= 0) throw PmcError(LL_EINVAL, "bad n_arenas / max_steps");
    if (c.push_enabled && (c.push_interval_step <= 0 || c.push_duration_step > c.push_interval_step))
      throw PmcError(LL_EINVAL, "push schedule: duration_time <= interval_time required (PR:34)");
    StepParams& P = base.P;
    const size_t N = (size_t)P.n_envs;                                      // robot rows
    P.obs_dim = 3 * P.prop_dim + 36 + LLS_OBS_DIM_FIXED;
    P.obs = base.template dalloc<float>(N * P.obs_dim);
    memset(&S, 0, sizeof S);
    EpmcParams& E = S.e;
    E.max_steps = c.max_steps;
    E.push_enabled = c.push_enabled ? 1 : 0; E.push_count0 = c.push_count0;
    E.push_interval_step = c.push_interval_step; E.push_duration_step = c.push_duration_step;
    E.friction_lo = (float)c.friction_range[0]; E.friction_hi = (float)c.friction_range[1];
    E.hforce_lo = (float)c.horizontal_force[0]; E.hforce_hi = (float)c.horizontal_force[1];
    E.vforce_lo = (float)c.vertical_force[0]; E.vforce_hi = (float)c.vertical_force[1];
    E.push_ratio = (float)c.push_strength_ratio; E.plane_friction = (float)LLM_PLANE_FRICTION;
    E.aux_radius = -1.0f; E.box_friction = 0.5f; E.terrain_contacts = 1;
    for (int i = 0; i < 4; i++) { E.noise_on[i] = c.noise_enabled[i] ? 1 : 0; E.noise_lo[i] = (float)c.noise_range[i][0]; E.noise_hi[i] = (float)c.noise_range[i][1]; }
    float init[37];
    for (int i = 0; i < 37; i++) init[i] = (float)init37[i];
    float* d_init = base.template dalloc<float>(37);
    base.bk.h2d(d_init, init, sizeof init);
    E.init_state = d_init;
    E.boxes = base.template dalloc<float>(N * EPMC_MAX_BOXES * EPMC_BOX_WORDS + EPMC_BOX_WORDS);   // + one record: the ray loops read one box ahead
    E.ray_pose = base.template dalloc<float>(N * EPMC_RAY_POSE);                                    // what the ray kernel needs of a row (epmc_step.hpp percept_rays; backends that cast the rays inside the step never touch it)
    E.push_trace = base.template dalloc<float>(N * P.n_sub * 4);
    if (N <= 512) {
      E.ray_trace = base.template dalloc<float>(N * EPMC_N_RAYS * 8);      // diagnostics / parity only
      S.vis_trace = base.template dalloc<float>(N * 16 * 8);
    }
    S.robot_contacts = 1;
    S.rand_cube = c.rand_cube ? 1 : 0; S.hurdle = c.hurdle ? 1 : 0; S.hole = c.hole ? 1 : 0;
    S.cos_visible = (float)cos(c.visible_angle); S.control_spd = (float)c.control_spd;
    base.P.max_tau1 = c.max_tau_robot1 > 0 ? (float)c.max_tau_robot1 : 0.0f;
    S.sp = base.template dalloc<float>(N * SEPMC_SP_STRIDE);
    S.info = base.template dalloc<float>(N * 4);
    d_reset_draws = base.template dalloc<float>((N / 2) * EPMC_MAX_DRAWS);
    d_prev_orn = base.template dalloc<float>((N / 2) * 4);
    d_row_ids = base.template dalloc<int32_t>(N);
    std::vector<float> sp(N * SEPMC_SP_STRIDE, 0.0f);
    for (size_t r = 0; r < N; r++)
      for (int i = 0; i < 4; i++) sp[r * SEPMC_SP_STRIDE + SP_INIT_ORN + i] = init[3 + i];
    base.bk.h2d(S.sp, sp.data(), sp.size() * 4);
  }

  void ensure_script_buffers(int n_draws) {
    const size_t N = base.P.n_envs, A = N / 2;
    if (!d_scr_state) {
      d_scr_state = base.template dalloc<float>(N * 37);
      d_scr_hit = base.template dalloc<uint8_t>(N * EPMC_N_RAYS);
      d_scr_frac = base.template dalloc<float>(N * EPMC_N_RAYS);
      d_scr_vis = base.template dalloc<uint8_t>(A * SEPMC_N_VIS);
      d_scr_contacts = base.template dalloc<int32_t>(A * SEPMC_MAX_CONTACTS * 4);
    }
    if (n_draws > scr_draws_cap) {
      d_scr_draws = base.template dalloc<float>(A * (size_t)n_draws);
      scr_draws_cap = n_draws;
    }
  }
  void script_reset(const uint8_t* h_hit, const float* h_frac, const uint8_t* h_vis) {
    ensure_script_buffers(0);
    const size_t N = base.P.n_envs;
    base.bk.sync();
    base.bk.h2d(d_scr_hit, h_hit, N * EPMC_N_RAYS);
    base.bk.h2d(d_scr_frac, h_frac, N * EPMC_N_RAYS * 4);
    base.bk.h2d(d_scr_vis, h_vis, (N / 2) * SEPMC_N_VIS);
    reset_scripted = true;
  }
  void reset(const int32_t* arena_ids, int n, const float* h_draws, const float* h_prev_orn) {
    const int A = base.P.n_envs / 2;
    if (!arena_ids) n = A;
    if (n <= 0 || n > A) throw PmcError(LL_EINVAL, "bad arena count");
    base.bk.sync();
    if (arena_ids) {
      std::vector<int32_t> rows(2 * (size_t)n);
      for (int i = 0; i < n; i++) {
        if (arena_ids[i] < 0 || arena_ids[i] >= A) throw PmcError(LL_EINVAL, "arena id out of range");
        rows[2 * i] = 2 * arena_ids[i]; rows[2 * i + 1] = 2 * arena_ids[i] + 1;
      }
      base.bk.h2d(d_row_ids, rows.data(), rows.size() * 4);
    }
    if (h_draws) base.bk.h2d(d_reset_draws, h_draws, (size_t)n * EPMC_MAX_DRAWS * 4);
    if (h_prev_orn) base.bk.h2d(d_prev_orn, h_prev_orn, (size_t)n * 4 * 4);
    SepmcParams Q = S;
    if (reset_scripted) { Q.e.scr_ray_hit = d_scr_hit; Q.e.scr_ray_frac = d_scr_frac; Q.scr_vis = d_scr_vis; Q.scr_on = 1; }
    base.bk.launch_sepmc_reset(base.P, Q, arena_ids ? d_row_ids : nullptr, 2 * n, h_draws ? d_reset_draws : nullptr, h_prev_orn ? d_prev_orn : nullptr);
    reset_scripted = false;
    have_reset = true;
  }
  void set_step_draws(const float* h_draws, int n_draws) {
    if (n_draws < 0) throw PmcError(LL_EINVAL, "negative draw count");
    ensure_script_buffers(n_draws);
    base.bk.sync();
    if (n_draws > 0) base.bk.h2d(d_scr_draws, h_draws, (size_t)(base.P.n_envs / 2) * n_draws * 4);
    pending_step_draws = n_draws > 0 ? n_draws : -1;
  }
  void step(const float* d_act) {
    if (!have_reset) throw PmcError(LL_ESTATE, "ll_sepmc_reset must be called before ll_sepmc_step");
    StepParams Q = base.P;
    Q.actions = d_act ? d_act : base.d_actions;
    SepmcParams R = S;
    if (pending_step_draws != 0) {
      if (!d_scr_draws) ensure_script_buffers(1);
      R.e.scr_draws = d_scr_draws; R.e.scr_n_draws = pending_step_draws > 0 ? pending_step_draws : 0;
      pending_step_draws = 0;
    }
    base.bk.launch_sepmc_step(Q, R);
    base.P.step_count += 1;
  }
  // n_steps control steps of the random-policy loop in one launch (pmc_engine.hpp step_random_n): the draws of a step are keyed on (arena,
  // episode, draw counter), so nothing but the actions' Philox step index moves from step to step
  void step_random_n(float sigma, int n_steps) {
    if (!have_reset) throw PmcError(LL_ESTATE, "ll_sepmc_reset must be called before ll_sepmc_step_random_n");
    if (!(sigma > 0.0f) || n_steps <= 0) throw PmcError(LL_EINVAL, "sigma and n_steps must be positive");
    if (pending_step_draws != 0) throw PmcError(LL_ESTATE, "scripted draws apply to single steps only");
    StepParams Q = base.P;
    Q.actions = base.d_actions; Q.action_sigma = sigma; Q.n_steps = n_steps;
    base.bk.launch_sepmc_step(Q, S);
    base.P.step_count += (uint64_t)n_steps;
  }
  void step_scripted(const float* h_actions, const float* h_state, const uint8_t* h_hit, const float* h_frac, const uint8_t* h_vis, const int32_t* h_contacts,
                     const float* h_draws, int n_draws) {
    if (!have_reset) throw PmcError(LL_ESTATE, "ll_sepmc_reset must be called before ll_sepmc_step_scripted");
    const size_t N = base.P.n_envs, A = N / 2;
    ensure_script_buffers(n_draws);
    base.bk.sync();
    base.bk.h2d(base.d_actions, h_actions, N * 12 * 4);
    base.bk.h2d(d_scr_state, h_state, N * 37 * 4);
    base.bk.h2d(d_scr_hit, h_hit, N * EPMC_N_RAYS);
    base.bk.h2d(d_scr_frac, h_frac, N * EPMC_N_RAYS * 4);
    base.bk.h2d(d_scr_vis, h_vis, A * SEPMC_N_VIS);
    base.bk.h2d(d_scr_contacts, h_contacts, A * SEPMC_MAX_CONTACTS * 4 * 4);
    if (h_draws && n_draws > 0) base.bk.h2d(d_scr_draws, h_draws, A * (size_t)n_draws * 4);
    StepParams Q = base.P;
    Q.actions = base.d_actions;
    SepmcParams R = S;
    R.e.scr_state = d_scr_state; R.e.scr_ray_hit = d_scr_hit; R.e.scr_ray_frac = d_scr_frac;
    R.scr_vis = d_scr_vis; R.scr_contacts = d_scr_contacts; R.scr_on = 1;
    R.e.scr_draws = d_scr_draws; R.e.scr_n_draws = (h_draws && n_draws > 0) ? n_draws : 0;
    if (!d_scr_draws) { ensure_script_buffers(1); R.e.scr_draws = d_scr_draws; }
    base.bk.launch_sepmc_step(Q, R);
    base.P.step_count += 1;
  }
};
