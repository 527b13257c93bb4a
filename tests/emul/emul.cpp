// emul.cpp -- TEST INFRASTRUCTURE.  Runs the kernel body (csrc/pmc_step.hpp) and the engine host logic
// (csrc/pmc_engine.hpp) on the host through lanes_host.hpp, exporting the same ll_* entry points as the product
// library so one Python harness drives both.  Purpose: debug the kernel's logic against the oracle on a machine
// without a GPU.  Built only by the tests into tests/emul/_build/; the product never loads it.
#include "lanes_host.hpp"

#include <stdlib.h>
#include <string.h>

#include "../../lifelike_agility_and_play_amd/csrc/epmc_engine.hpp"
#include "../../lifelike_agility_and_play_amd/csrc/sepmc_engine.hpp"
#include "../../lifelike_agility_and_play_amd/csrc/pmc_engine.hpp"
#include "../../lifelike_agility_and_play_amd/csrc/pmc_step.hpp"

typedef Pmc<HostLanes> K;
typedef WithGramPipeHost<HostLanes> HostLanesPipe;      // what the one-wave-per-SIMD PMC cone kernels run (LL_EMUL_GRAM_PIPE): the contact rows' Gram blocks as a background job (lanes.hpp WithGramPipe)
typedef WithConeInLdsHost<HostLanes> HostLanesLds;      // what the larger-batch GPU builds run (LL_EMUL_PARK): the cone round's cross scalars through the row scratch

struct HostBackend {
  explicit HostBackend(int) {}
  void set_stream(void*) {}
  void* stream_handle() { return nullptr; }
  void* alloc(size_t bytes) { return malloc(bytes ? bytes : 4); }
  void release(void* p) { free(p); }
  void zero(void* p, size_t bytes) { memset(p, 0, bytes); }
  void h2d(void* d, const void* h, size_t bytes) { memcpy(d, h, bytes); }
  void d2h(void* h, const void* d, size_t bytes) { memcpy(h, d, bytes); }
  void sync() {}
  void launch_step(const StepParams& P) {
    HostLanes ln(P.candc);
    for (int sl = 0; sl < P.n_steps; sl++) {               // ll_step_random_n: step-major, the table folded after every step into that step's version
      if (P.action_sigma > 0.0f) {
        StepParams Q = P;
        Q.step_count = P.step_count + (uint64_t)sl;
        launch_actions(Q, P.actions_out, P.action_sigma);
      }
      for (int env = 0; env < P.n_envs; env++) {
        fN act[3];
        for (int j = 0; j < 3; j++) act[j] = ln.ldl(P.actions, (long)env * 12 + j, 3);
        if (P.friction_mode == 2 && getenv("LL_EMUL_PARK")) {                 // the larger-batch GPU build's variant (tests)
          HostLanesLds lq(P.candc);
          if (P.set_obstacle) Pmc<HostLanesLds>::step_env<true, true>(lq, P, env, act, sl);
          else Pmc<HostLanesLds>::step_env<false, true>(lq, P, env, act, sl);
        }
        else if (P.friction_mode == 2 && getenv("LL_EMUL_GRAM_PIPE")) {
          HostLanesPipe lq(P.candc);
          if (P.set_obstacle) Pmc<HostLanesPipe>::step_env<true, true>(lq, P, env, act, sl);
          else Pmc<HostLanesPipe>::step_env<false, true>(lq, P, env, act, sl);
        }
        else if (P.friction_mode == 2 && pmc_wants_xrows(P)) {                   // the extended contact rows (llenv.hip launch_step): the XROWS build
          if (P.set_obstacle) K::step_env<true, true, true>(ln, P, env, act, sl);
          else K::step_env<false, true, true>(ln, P, env, act, sl);
        }
        else if (P.set_obstacle && P.friction_mode == 2) K::step_env<true, true>(ln, P, env, act, sl);
        else if (P.set_obstacle) K::step_env<true>(ln, P, env, act, sl);
        else if (P.friction_mode == 2) K::step_env<false, true>(ln, P, env, act, sl);
        else K::step_env<false>(ln, P, env, act, sl);
      }
      pmc_finalize_table(P, sl, sl == P.n_steps - 1);
    }
  }
  bool co_resident(const StepParams&) const { return true; }
  void launch_reset(const StepParams& P, const int32_t* ids, int n, const int32_t* clip, const double* t0) {
    HostLanes ln(P.candc);
    for (int i = 0; i < n; i++) {
      int env = ids ? ids[i] : i, c;
      double t;
      uint32_t ep = P.ep_count[env] + 1;
      K::sample_start(ln, P, env, ep, &c, &t);
      P.ep_count[env] = ep;
      if (clip) c = clip[i];
      if (t0) t = t0[i];
      K::reset_env(ln, P, env, c, t);
      P.done[env] = 0;
      P.done_reason[env] = 0;
    }
  }
  void launch_probe_pd(const StepParams& P, const float* in, float* out, int n, int mode) {
    HostLanes ln(P.candc);
    for (int i = 0; i < n; i++) K::probe_pd(ln, P, in + (long)i * 36, out + (long)i * 12, mode);
  }
  void launch_gae(float* block, int n_envs, int unroll, int W, int od, float gamma, float lam, const float* bootstrap) {
    for (int env = 0; env < n_envs; env++) {
      float* rows = block + (size_t)env * unroll * W;
      float adv = 0.0f, vnext = bootstrap[env];
      for (int t = unroll - 1; t >= 0; t--) {
        float* r = rows + (size_t)t * W + od;
        const float V = r[14], m = r[16];
        const float delta = r[15] + gamma * vnext * m - V;
        adv = delta + gamma * lam * m * adv;
        r[13] = adv + V;
        vnext = V;
      }
    }
  }
  void launch_actions(const StepParams& P, float* actions, float sigma) {
    for (int gid = 0; gid < P.n_envs * 3; gid++) {
      uint32_t r[4];
      philox4x32((uint32_t)gid, (uint32_t)P.step_count, (uint32_t)(P.step_count >> 32), 0xAC710u, (uint32_t)P.seed, (uint32_t)(P.seed >> 32), r);
      const float k = 2.3283064365386963e-10f;
      float u1 = fminf(((float)r[0] + 1.0f) * k, 1.0f), u2 = (float)r[1] * k, u3 = fminf(((float)r[2] + 1.0f) * k, 1.0f), u4 = (float)r[3] * k;
      float m1 = sqrtf(-2.0f * logf(u1)), m2 = sqrtf(-2.0f * logf(u3));
      actions[4 * gid + 0] = sigma * m1 * cosf(6.283185307179586f * u2); actions[4 * gid + 1] = sigma * m1 * sinf(6.283185307179586f * u2);
      actions[4 * gid + 2] = sigma * m2 * cosf(6.283185307179586f * u4); actions[4 * gid + 3] = sigma * m2 * sinf(6.283185307179586f * u4);
    }
  }
  void draw_step_actions(const StepParams& P, int sl) {
    if (!(P.action_sigma > 0.0f)) return;
    StepParams Q = P;
    Q.step_count = P.step_count + (uint64_t)sl;
    launch_actions(Q, P.actions_out, P.action_sigma);
  }
  // LL_SPLIT_RAYS (as llenv.hip HipBackend; read per call here so that a test can run both ways in one process): the 778 rays of a row after the step, by Epmc::percept_row_host
  static bool rays_split(const EpmcParams& E) { const char* v = getenv("LL_SPLIT_RAYS"); return v && atoi(v) >= 1 && !E.scr_ray_hit; }
  void launch_epmc_step(const StepParams& P, const EpmcParams& E_in) {
    HostLanes ln(P.candc);
    EpmcParams E = E_in;
    E.split_rays = rays_split(E_in) ? 1 : 0;
    for (int sl = 0; sl < P.n_steps; sl++) {
      draw_step_actions(P, sl);
      for (int env = 0; env < P.n_envs; env++) {
        fN act[3];
        for (int j = 0; j < 3; j++) act[j] = ln.ldl(P.actions, (long)env * 12 + j, 3);
        const bool park = getenv("LL_EMUL_PARK") != nullptr, cone = P.friction_mode == 2;      // park: the larger-batch GPU build's variant (tests)
        if (park) { if (cone) { HostLanesLds lq(P.candc); Epmc<HostLanesLds>::step_env<true, true>(lq, P, E, env, act); } else Epmc<HostLanes>::step_env<true>(ln, P, E, env, act); }
        else if (cone && pmc_wants_xrows_terrain(P)) Epmc<HostLanes>::step_env<false, true, true>(ln, P, E, env, act);
        else      { if (cone) Epmc<HostLanes>::step_env<false, true>(ln, P, E, env, act); else Epmc<HostLanes>::step_env(ln, P, E, env, act); }
        if (E.split_rays) Epmc<HostLanes>::percept_row_host(P, E, env);
      }
    }
  }
  void launch_epmc_reset(const StepParams& P, const EpmcParams& E, const int32_t* ids, int n, const float* draws, const float* prev_orn) {
    HostLanes ln(P.candc);
    for (int i = 0; i < n; i++)
      Epmc<HostLanes>::reset_env(ln, P, E, ids ? ids[i] : i, draws ? draws + (long)i * EPMC_MAX_DRAWS : nullptr, prev_orn ? prev_orn + (long)i * 4 : nullptr);
  }
  // SEPMC: the two robots of an arena run as two threads that meet in HostLanes::peer (on the GPU: two rows of one wave)
  template <class LANES = HostLanes, class FN>
  static void run_pairs(const StepParams& P, int n_rows, FN fn) {
    for (int i = 0; i + 1 < n_rows; i += 2) {
      PairLink link;
      std::thread t[2];
      for (int side = 0; side < 2; side++)
        t[side] = std::thread([&, side]() {
          LANES ln(P.candc);
          ln.link_ = &link; ln.side_ = side;
          fn(ln, i + side);
        });
      t[0].join(); t[1].join();
    }
  }
  void launch_sepmc_step(const StepParams& P, const SepmcParams& S_in) {
    SepmcParams S = S_in;
    S.e.split_rays = rays_split(S_in.e) ? 1 : 0;
    for (int sl = 0; sl < P.n_steps; sl++) {
      draw_step_actions(P, sl);
      const bool park = getenv("LL_EMUL_PARK") != nullptr, cone = P.friction_mode == 2;
      if (park && cone)
        run_pairs<HostLanesLds>(P, P.n_envs, [&](HostLanesLds& ln, int row) {
          fN act[3];
          for (int j = 0; j < 3; j++) act[j] = ln.ldl(P.actions, (long)row * 12 + j, 3);
          Sepmc<HostLanesLds>::step_env<true, true>(ln, P, S, row, act);
        });
      else
        run_pairs(P, P.n_envs, [&](HostLanes& ln, int row) {
          fN act[3];
          for (int j = 0; j < 3; j++) act[j] = ln.ldl(P.actions, (long)row * 12 + j, 3);
          if (park) Sepmc<HostLanes>::step_env<true>(ln, P, S, row, act);
          else if (cone && pmc_wants_xrows_terrain(P)) Sepmc<HostLanes>::step_env<false, true, true>(ln, P, S, row, act);
          else if (cone) Sepmc<HostLanes>::step_env<false, true>(ln, P, S, row, act);
          else Sepmc<HostLanes>::step_env(ln, P, S, row, act);
        });
      if (S.e.split_rays)
        for (int row = 0; row < P.n_envs; row++) Epmc<HostLanes>::percept_row_host(P, S.e, row);
    }
  }
  void launch_sepmc_reset(const StepParams& P, const SepmcParams& S, const int32_t* ids, int n, const float* draws, const float* prev_orn) {
    run_pairs(P, n, [&](HostLanes& ln, int i) {
      Sepmc<HostLanes>::reset_env(ln, P, S, ids ? ids[i] : i, draws ? draws + (long)(i >> 1) * EPMC_MAX_DRAWS : nullptr, prev_orn ? prev_orn + (long)(i >> 1) * 4 : nullptr);
    });
  }
  void enable_timing(bool) {}
  void collect_timing(double* avg_ms, int* n, long long* steps = nullptr) { *avg_ms = 0; *n = 0; if (steps) *steps = 0; }
};

typedef PmcEngine<HostBackend> ENGINE;
#include "../../lifelike_agility_and_play_amd/csrc/pmc_capi.inc"
typedef EpmcEngine<HostBackend> EPMC_ENGINE;
#include "../../lifelike_agility_and_play_amd/csrc/epmc_capi.inc"
typedef SepmcEngine<HostBackend> SEPMC_ENGINE;
#include "../../lifelike_agility_and_play_amd/csrc/sepmc_capi.inc"
static_assert(sizeof(HostLanes::scratch_) == PMC_ROW_SCRATCH * sizeof(float), "lanes_host.hpp: the row scratch must have the size lanes.hpp states");
static_assert(HostLanes::kConeLdsAt == CONE_LDS_AT, "lanes_host.hpp: the cone cross scalars must live where lanes.hpp puts them");

extern "C" {
// single physics substep on explicit state (staged comparison with the oracle); tgt = PD target joint angles
int emu_substep(ll_engine* h, float* state37, const float* tgt12) {
  const StepParams& P = h->e->P;
  HostLanes ln(P.candc);
  K::Base bs;
  bs.p = mk3<float>(state37[0], state37[1], state37[2]);
  bs.q.x = state37[3]; bs.q.y = state37[4]; bs.q.z = state37[5]; bs.q.w = state37[6];
  bs.v = mk3<float>(state37[7], state37[8], state37[9]);
  bs.w = mk3<float>(state37[10], state37[11], state37[12]);
  fN q[3], qd[3], tgt[3];
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < EW; i++) { int l = i >> 2; q[j].v[i] = state37[13 + 3 * l + j]; qd[j].v[i] = state37[25 + 3 * l + j]; tgt[j].v[i] = tgt12[3 * l + j]; }
  K::substep(ln, P, bs, q, qd, tgt);
  state37[0] = bs.p.x; state37[1] = bs.p.y; state37[2] = bs.p.z;
  state37[3] = bs.q.x; state37[4] = bs.q.y; state37[5] = bs.q.z; state37[6] = bs.q.w;
  state37[7] = bs.v.x; state37[8] = bs.v.y; state37[9] = bs.v.z;
  state37[10] = bs.w.x; state37[11] = bs.w.y; state37[12] = bs.w.z;
  for (int j = 0; j < 3; j++)
    for (int l = 0; l < 4; l++) { state37[13 + 3 * l + j] = q[j].v[4 * l]; state37[25 + 3 * l + j] = qd[j].v[4 * l]; }
  return 0;
}
}
