// epmc_engine.hpp -- host-side engine behind include/llenv_epmc.h.  It owns a PmcEngine for everything the two envs share
// (model tables, state / obs / reward / done / action buffers, counters, backend) and adds the EPMC per-env buffers.
// The backend must provide launch_epmc_step / launch_epmc_reset next to the PMC launches.
#pragma once
#include <string.h>

#include "../../include/llenv_epmc.h"
#include "epmc_step.hpp"
#include "pmc_engine.hpp"

template <class BK>
struct EpmcEngine {
  PmcEngine<BK> base;
  ll_epmc_config cfg;
  EpmcParams E;
  bool have_reset = false;
  float *d_scr_state = nullptr, *d_scr_frac = nullptr, *d_scr_draws = nullptr, *d_reset_draws = nullptr, *d_prev_orn = nullptr;
  uint8_t* d_scr_hit = nullptr;
  int scr_draws_cap = 0;
  bool reset_rays_scripted = false;

  static ll_config base_config(const ll_epmc_config& c) {
    ll_config b;
    memset(&b, 0, sizeof b);
    b.abi_version = LL_ABI_VERSION;
    b.n_envs = c.n_envs; b.device = c.device; b.auto_reset = c.auto_reset;
    b.control_freq = c.control_freq; b.sim_freq = 500.0;                   // PGE:82 time_step = 1/500, not configurable
    b.kp = c.kp; b.kd = c.kd; b.max_tau = c.max_tau;
    b.foot_lateral_friction = c.friction_range[0];                         // per-episode value travels in SubstepExtra
    for (int i = 0; i < 5; i++) { b.reward_weights[i] = 1.0; b.prop_order[i] = c.prop_order[i]; }
    b.solver_iterations = c.solver_iterations;
    b.seed = c.seed;
    return b;
  }

  EpmcEngine(const ll_epmc_config& c, const double* blob, int blob_len, const double* init37) : base(base_config(c), blob, blob_len), cfg(c) {
    if (c.abi_version != LL_ABI_VERSION) throw PmcError(LL_EINVAL, "ll_epmc_config.abi_version mismatch");
    if (c.element_id < 0 || c.element_id > 3) throw PmcError(LL_EINVAL, "Unknown element id.");                 // BSE:249-250
    if (c.max_steps <= 0 || c.cmd_vary_freq_range[0] <= 0 || c.cmd_vary_freq_range[1] <= c.cmd_vary_freq_range[0])
      throw PmcError(LL_EINVAL, "bad max_steps / cmd_vary_freq_range");
    if (c.push_enabled && (c.push_interval_step <= 0 || c.push_duration_step > c.push_interval_step))
      throw PmcError(LL_EINVAL, "push schedule: duration_time <= interval_time required (PR:34)");
    StepParams& P = base.P;
    const size_t N = (size_t)P.n_envs;
    P.obs_dim = 3 * P.prop_dim + 36 + LLE_OBS_DIM_FIXED;
    P.obs = base.template dalloc<float>(N * P.obs_dim);
    memset(&E, 0, sizeof E);
    E.element_id = c.element_id; E.max_steps = c.max_steps;
    E.push_enabled = c.push_enabled ? 1 : 0; E.push_count0 = c.push_count0;
    E.push_interval_step = c.push_interval_step; E.push_duration_step = c.push_duration_step;
    E.cmd_freq_lo = c.cmd_vary_freq_range[0]; E.cmd_freq_hi = c.cmd_vary_freq_range[1];
    E.friction_lo = (float)c.friction_range[0]; E.friction_hi = (float)c.friction_range[1];
    E.hforce_lo = (float)c.horizontal_force[0]; E.hforce_hi = (float)c.horizontal_force[1];
    E.vforce_lo = (float)c.vertical_force[0]; E.vforce_hi = (float)c.vertical_force[1];
    E.push_ratio = (float)c.push_strength_ratio; E.plane_friction = (float)LLM_PLANE_FRICTION;
    E.spd_lo = (float)c.target_spd_range[0]; E.spd_hi = (float)c.target_spd_range[1];
    E.aux_radius = (float)c.auxiliary_radius;
    E.hole_gap_lo = (float)c.hole_gap_height[0]; E.hole_gap_hi = (float)c.hole_gap_height[1];
    E.box_friction = 0.5f; E.terrain_contacts = 1;
    for (int i = 0; i < 4; i++) { E.noise_on[i] = c.noise_enabled[i] ? 1 : 0; E.noise_lo[i] = (float)c.noise_range[i][0]; E.noise_hi[i] = (float)c.noise_range[i][1]; }
    float init[37];
    for (int i = 0; i < 37; i++) init[i] = (float)init37[i];
    float* d_init = base.template dalloc<float>(37);
    base.bk.h2d(d_init, init, sizeof init);
    E.init_state = d_init;
    E.ep = base.template dalloc<float>(N * EPMC_EP_STRIDE);
    E.info = base.template dalloc<float>(N * 6);
    E.statics = base.template dalloc<float>(N * EPMC_MAX_STATICS * 8);
    E.boxes = base.template dalloc<float>(N * EPMC_MAX_BOXES * EPMC_BOX_WORDS + EPMC_BOX_WORDS);   // + one record: the ray loops read one box ahead
    E.ray_pose = base.template dalloc<float>(N * EPMC_RAY_POSE);                                    // what the ray kernel needs of a row (epmc_step.hpp percept_rays; backends that cast the rays inside the step never touch it)
    E.push_trace = base.template dalloc<float>(N * P.n_sub * 4);
    if (N <= 512) E.ray_trace = base.template dalloc<float>(N * EPMC_N_RAYS * 8);    // diagnostics / parity only
    d_reset_draws = base.template dalloc<float>(N * EPMC_MAX_DRAWS);
    d_prev_orn = base.template dalloc<float>(N * 4);
    // every env starts from its own copy of the start orientation (rotated in place at each reset, PGE:186-190)
    std::vector<float> ep(N * EPMC_EP_STRIDE, 0.0f);
    for (size_t e = 0; e < N; e++)
      for (int i = 0; i < 4; i++) ep[e * EPMC_EP_STRIDE + EP_INIT_ORN + i] = init[3 + i];
    base.bk.h2d(E.ep, ep.data(), ep.size() * 4);
  }

  void ensure_script_buffers(int n_draws) {
    const size_t N = base.P.n_envs;
    if (!d_scr_state) {
      d_scr_state = base.template dalloc<float>(N * 37);
      d_scr_hit = base.template dalloc<uint8_t>(N * EPMC_N_RAYS);
      d_scr_frac = base.template dalloc<float>(N * EPMC_N_RAYS);
    }
    if (n_draws > scr_draws_cap) {
      d_scr_draws = base.template dalloc<float>(N * (size_t)n_draws);
      scr_draws_cap = n_draws;
    }
  }

  void script_reset_rays(const uint8_t* h_hit, const float* h_frac) {
    ensure_script_buffers(0);
    const size_t N = base.P.n_envs;
    base.bk.sync();
    base.bk.h2d(d_scr_hit, h_hit, N * EPMC_N_RAYS);
    base.bk.h2d(d_scr_frac, h_frac, N * EPMC_N_RAYS * 4);
    reset_rays_scripted = true;
  }

  void reset(const int32_t* env_ids, int n, const float* h_draws, const float* h_prev_orn) {
    const int N = base.P.n_envs;
    if (!env_ids) n = N;
    if (n <= 0 || n > N) throw PmcError(LL_EINVAL, "bad env count");
    base.bk.sync();
    if (env_ids) {
      for (int i = 0; i < n; i++)
        if (env_ids[i] < 0 || env_ids[i] >= N) throw PmcError(LL_EINVAL, "env id out of range");
      base.bk.h2d(base.d_reset_ids, env_ids, n * 4);
    }
    if (h_draws) base.bk.h2d(d_reset_draws, h_draws, (size_t)n * EPMC_MAX_DRAWS * 4);
    if (h_prev_orn) base.bk.h2d(d_prev_orn, h_prev_orn, (size_t)n * 4 * 4);
    EpmcParams Q = E;
    if (reset_rays_scripted) { Q.scr_ray_hit = d_scr_hit; Q.scr_ray_frac = d_scr_frac; }
    base.bk.launch_epmc_reset(base.P, Q, env_ids ? base.d_reset_ids : nullptr, n, h_draws ? d_reset_draws : nullptr, h_prev_orn ? d_prev_orn : nullptr);
    reset_rays_scripted = false;
    have_reset = true;
  }

  int pending_step_draws = 0;
  void set_step_draws(const float* h_draws, int n_draws) {
    if (n_draws < 0) throw PmcError(LL_EINVAL, "negative draw count");
    ensure_script_buffers(n_draws);
    base.bk.sync();
    if (n_draws > 0) base.bk.h2d(d_scr_draws, h_draws, (size_t)base.P.n_envs * n_draws * 4);
    pending_step_draws = n_draws > 0 ? n_draws : -1;              // -1: the step must not draw at all
  }
  void step(const float* d_act) {
    if (!have_reset) throw PmcError(LL_ESTATE, "ll_epmc_reset must be called before ll_epmc_step");
    StepParams Q = base.P;
    Q.actions = d_act ? d_act : base.d_actions;
    EpmcParams R = E;
    if (pending_step_draws != 0) {
      if (!d_scr_draws) ensure_script_buffers(1);
      R.scr_draws = d_scr_draws; R.scr_n_draws = pending_step_draws > 0 ? pending_step_draws : 0;
      pending_step_draws = 0;
    }
    base.bk.launch_epmc_step(Q, R);
    base.P.step_count += 1;
  }

  // n_steps control steps of the random-policy loop in one launch (pmc_engine.hpp step_random_n): the draws of a step are keyed on (env,
  // episode, draw counter), so nothing but the actions' Philox step index moves from step to step
  void step_random_n(float sigma, int n_steps) {
    if (!have_reset) throw PmcError(LL_ESTATE, "ll_epmc_reset must be called before ll_epmc_step_random_n");
    if (!(sigma > 0.0f) || n_steps <= 0) throw PmcError(LL_EINVAL, "sigma and n_steps must be positive");
    if (pending_step_draws != 0) throw PmcError(LL_ESTATE, "scripted draws apply to single steps only");
    StepParams Q = base.P;
    Q.actions = base.d_actions; Q.action_sigma = sigma; Q.n_steps = n_steps;
    base.bk.launch_epmc_step(Q, E);
    base.P.step_count += (uint64_t)n_steps;
  }
  void step_scripted(const float* h_actions, const float* h_state, const uint8_t* h_hit, const float* h_frac, const float* h_draws, int n_draws) {
    if (!have_reset) throw PmcError(LL_ESTATE, "ll_epmc_reset must be called before ll_epmc_step_scripted");
    const size_t N = base.P.n_envs;
    ensure_script_buffers(n_draws);
    base.bk.sync();
    base.bk.h2d(base.d_actions, h_actions, N * 12 * 4);
    base.bk.h2d(d_scr_state, h_state, N * 37 * 4);
    base.bk.h2d(d_scr_hit, h_hit, N * EPMC_N_RAYS);
    base.bk.h2d(d_scr_frac, h_frac, N * EPMC_N_RAYS * 4);
    if (h_draws && n_draws > 0) base.bk.h2d(d_scr_draws, h_draws, N * (size_t)n_draws * 4);
    StepParams Q = base.P;
    Q.actions = base.d_actions;
    EpmcParams R = E;
    R.scr_state = d_scr_state; R.scr_ray_hit = d_scr_hit; R.scr_ray_frac = d_scr_frac;
    if (h_draws && n_draws > 0) { R.scr_draws = d_scr_draws; R.scr_n_draws = n_draws; }
    base.bk.launch_epmc_step(Q, R);
    base.P.step_count += 1;
  }
};
