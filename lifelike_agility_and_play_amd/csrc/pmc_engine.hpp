// pmc_engine.hpp -- host-side engine behind the C ABI (include/llenv.h): buffer ownership, SoA<->row
// conversion at the boundary, launch sequencing.  Generic over a Backend that allocates memory and launches the
// three kernels (pre-step, step, reset); the product instantiates it with the HIP backend (llenv.hip).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "pmc_params.hpp"
#include "pmc_tables.hpp"

struct PmcError : public std::runtime_error {
  int code;
  PmcError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

template <class BK>
struct PmcEngine {
  BK bk;
  ll_config cfg;
  StepParams P;
  bool have_mocap = false, have_reset = false;
  // table state (device, float64)
  double *d_avg_reward = nullptr, *d_avg_len = nullptr, *d_prob = nullptr, *d_cdf = nullptr;
  float* d_actions = nullptr;       // engine-owned action buffer
  float* d_traj = nullptr;          // optional unroll buffers [n_buffers][n_envs][unroll][obs_dim + LL_UNROLL_EXTRA]
  int traj_unroll = 0, traj_buffers = 0;
  float *d_neglogp = nullptr, *d_value = nullptr;   // [n_envs] policy outputs that accompany the actions (ll_pg_ptrs)
  int32_t* d_reset_ids = nullptr;   // scratch for ll_reset
  int32_t* d_reset_clip = nullptr;
  double* d_reset_t0 = nullptr;
  std::vector<void*> allocs;
  std::vector<double> h_max_steps;
  std::vector<int32_t> h_clip_len;

  template <class T>
  T* dalloc(size_t n) {
    void* p = bk.alloc(n * sizeof(T));
    bk.zero(p, n * sizeof(T));
    allocs.push_back(p);
    return (T*)p;
  }

  PmcEngine(const ll_config& c, const double* blob, int blob_len) : bk(c.device), cfg(c) {
    std::string e = pmc_fill_params(cfg, P);
    if (!e.empty()) throw PmcError(LL_EINVAL, e);
    std::vector<float> legc, basec;
    e = pmc_build_tables(blob, blob_len, legc, basec);
    if (!e.empty()) throw PmcError(LL_EINVAL, e);
    const size_t N = (size_t)P.n_envs;
    float* lc = dalloc<float>(legc.size());
    float* bc = dalloc<float>(basec.size());
    bk.h2d(lc, legc.data(), legc.size() * 4);
    bk.h2d(bc, basec.data(), basec.size() * 4);
    P.legc = lc; P.basec = bc;
    std::vector<float> candc;
    e = pmc_build_cand_table(blob, candc);
    if (!e.empty()) throw PmcError(LL_EINVAL, e);
    float* cc = dalloc<float>(candc.size());
    bk.h2d(cc, candc.data(), candc.size() * 4);
    P.candc = cc;
    P.state = dalloc<float>(37 * N); P.kin = dalloc<float>(37 * N); P.feet = dalloc<float>(24 * N);
    P.time = dalloc<double>(N); P.clip = dalloc<int32_t>(N); P.ep_steps = dalloc<int32_t>(N);
    P.reward_sum = dalloc<float>(N); P.ep_count = dalloc<uint32_t>(N);
    P.obs = dalloc<float>(N * P.obs_dim); P.term_obs = dalloc<float>(N * P.obs_dim);
    P.reward = dalloc<float>(N); P.done = dalloc<uint8_t>(N); P.done_reason = dalloc<uint8_t>(N);
    d_actions = dalloc<float>(N * 12);
    d_neglogp = dalloc<float>(N); d_value = dalloc<float>(N);
    P.actions = d_actions;
    P.counters = dalloc<unsigned long long>(4 + (size_t)PMC_TS_SLOTS * N);
    P.ep_hist = dalloc<unsigned long long>(16);
    P.block_ticket = dalloc<unsigned int>(LL_MAX_STEPS_PER_LAUNCH);
    P.ver_ready = dalloc<unsigned int>(LL_MAX_STEPS_PER_LAUNCH);
    P.resident = dalloc<unsigned int>(2);
    P.actions_out = d_actions;
    d_reset_ids = dalloc<int32_t>(N); d_reset_clip = dalloc<int32_t>(N); d_reset_t0 = dalloc<double>(N);
  }
  ~PmcEngine() {
    for (void* p : allocs) bk.release(p);
  }

  // ML:19-46
  void load_mocap(const double* frames, const int32_t* clip_len, int n_clips, double frame_step) {
    if (have_mocap) throw PmcError(LL_ESTATE, "mocap table already loaded");
    if (n_clips <= 0 || !(frame_step > 0)) throw PmcError(LL_EINVAL, "bad mocap table");
    pmc_fill_mocap(P, n_clips, frame_step);
    size_t total = 0;
    std::vector<int32_t> off(n_clips);
    h_max_steps.resize(n_clips);
    h_clip_len.assign(clip_len, clip_len + n_clips);
    for (int c = 0; c < n_clips; c++) {
      if (clip_len[c] <= P.margin + 1 + P.frame_rate + 2) throw PmcError(LL_EINVAL, "clip shorter than the sampling margin (ML:35,50)");
      off[c] = (int32_t)total;
      total += clip_len[c];
      h_max_steps[c] = (clip_len[c] - P.margin) * frame_step / P.policy_step;    // ML:45
    }
    double* fr = dalloc<double>(total * 19);
    bk.h2d(fr, frames, total * 19 * 8);
    int32_t* dco = dalloc<int32_t>(n_clips);
    int32_t* dcl = dalloc<int32_t>(n_clips);
    double* dms = dalloc<double>(n_clips);
    bk.h2d(dco, off.data(), n_clips * 4);
    bk.h2d(dcl, clip_len, n_clips * 4);
    bk.h2d(dms, h_max_steps.data(), n_clips * 8);
    P.frames = fr; P.clip_off = dco; P.clip_len = dcl; P.max_steps = dms;
    d_avg_reward = dalloc<double>(n_clips); d_avg_len = dalloc<double>(n_clips);
    d_prob = dalloc<double>(n_clips); d_cdf = dalloc<double>(n_clips);
    P.pending_reward = dalloc<unsigned long long>((size_t)LL_MAX_STEPS_PER_LAUNCH * n_clips);
    P.pending_len = dalloc<unsigned long long>((size_t)LL_MAX_STEPS_PER_LAUNCH * n_clips);
    P.cdf_ver = dalloc<double>((size_t)LL_MAX_STEPS_PER_LAUNCH * n_clips);
    std::vector<double> prob(n_clips, 1.0 / n_clips), cdf(n_clips);            // ML:46
    double acc = 0;
    for (int c = 0; c < n_clips; c++) { acc += prob[c]; cdf[c] = acc; }
    cdf[n_clips - 1] = 1.0;
    bk.h2d(d_prob, prob.data(), n_clips * 8);
    bk.h2d(d_cdf, cdf.data(), n_clips * 8);
    P.cdf = d_cdf; P.prob = d_prob; P.avg_reward = d_avg_reward; P.avg_len = d_avg_len;
    have_mocap = true;
  }

  // utils/obstacle.py:6-33 tables, computed by the host (scipy find_peaks, as the reference does)
  bool have_obstacles = false;
  void load_obstacles(const int32_t* count, const double* table, int n_clips) {
    need(true, false);
    if (n_clips != P.n_clips) throw PmcError(LL_EINVAL, "obstacle table does not match the clip table");
    std::vector<int32_t> off(n_clips);
    int total = 0;
    for (int c = 0; c < n_clips; c++) {
      if (count[c] < 0) throw PmcError(LL_EINVAL, "negative obstacle count");
      off[c] = total;
      total += count[c];
    }
    int32_t* doff = dalloc<int32_t>(n_clips);
    int32_t* dcnt = dalloc<int32_t>(n_clips);
    double* dtab = dalloc<double>((size_t)(total > 0 ? total : 1) * 4);
    bk.h2d(doff, off.data(), n_clips * 4);
    bk.h2d(dcnt, count, n_clips * 4);
    if (total > 0) bk.h2d(dtab, table, (size_t)total * 4 * 8);
    P.ob_off = doff; P.ob_cnt = dcnt; P.ob_table = dtab;
    P.ob_id = dalloc<int32_t>(P.n_envs);
    have_obstacles = true;
  }

  void need(bool mocap, bool reset) const {
    if (mocap && !have_mocap) throw PmcError(LL_ESTATE, "ll_load_mocap must be called first");
    if (reset && !have_reset) throw PmcError(LL_ESTATE, "ll_reset must be called before ll_step");
    if (reset && P.set_obstacle && !have_obstacles) throw PmcError(LL_ESTATE, "set_obstacle needs ll_load_obstacles");
  }

  // PLE:150-171
  void reset(const int32_t* env_ids, int n, const int32_t* clip, const double* t0) {
    need(true, false);
    const int N = P.n_envs;
    if (!env_ids) n = N;
    if (n < 0 || n > N) throw PmcError(LL_EINVAL, "bad env count in ll_reset");
    if (P.set_obstacle && !have_obstacles) throw PmcError(LL_ESTATE, "set_obstacle needs ll_load_obstacles");
    if (n == 0) return;                                  // `reset(env_ids=where(done))` on a step that finished nobody
    if (env_ids) {
      std::vector<char> seen(N, 0);                      // two rows for one env would race on its state and episode counter
      for (int i = 0; i < n; i++) {
        if (env_ids[i] < 0 || env_ids[i] >= N) throw PmcError(LL_EINVAL, "env id out of range");
        if (seen[env_ids[i]]) throw PmcError(LL_EINVAL, "env id listed twice in ll_reset");
        seen[env_ids[i]] = 1;
      }
      bk.h2d(d_reset_ids, env_ids, n * 4);
    }
    if (clip) {
      for (int i = 0; i < n; i++)
        if (clip[i] < 0 || clip[i] >= P.n_clips) throw PmcError(LL_EINVAL, "clip index out of range");
      bk.h2d(d_reset_clip, clip, n * 4);
    }
    if (t0) {
      // A start time is only meaningful inside the clip it belongs to, and the first observation looks frame_rate + 2 rows
      // ahead of it (ML:75-86): the admissible range is the reference's own sampling range (ML:50-51).
      if (!clip) throw PmcError(LL_EINVAL, "ll_reset: explicit start times need explicit clip indices");
      for (int i = 0; i < n; i++) {
        const double tmax = P.frame_step * (double)(h_clip_len[clip[i]] - P.margin - 1);
        if (!(t0[i] >= 0) || t0[i] > tmax) throw PmcError(LL_EINVAL, "start time outside the clip's sampling range (ML:50)");
      }
      bk.h2d(d_reset_t0, t0, n * 8);
    }
    bk.launch_reset(P, env_ids ? d_reset_ids : nullptr, n, clip ? d_reset_clip : nullptr, t0 ? d_reset_t0 : nullptr);
    have_reset = true;
  }

  // PLE:195-245
  // One launch: the step kernel folds the statistics of the episodes it finishes into the sampling table itself (its last
  // workgroup does, PLE:235-240), and with sigma > 0 it also draws the actions a ~ N(0, sigma^2) it then applies -- the same
  // Philox stream as fill_random_actions(), so step_random(s) == fill_random_actions(s); step(nullptr).
  void step(const float* d_act, float sigma = 0.0f, int n_steps = 1) {
    need(true, true);
    // A launch runs at most LL_MAX_STEPS_PER_LAUNCH control steps (the per-step slots of the sampling table), and a multi-step launch whose grid
    // the chip cannot hold at once runs as single launches: its waves could not wait for each other's finished episodes (bk.co_resident)
    const int per = (n_steps > 1 && !bk.co_resident(P)) ? 1 : LL_MAX_STEPS_PER_LAUNCH;
    for (int done = 0; done < n_steps;) {
      const int k = n_steps - done < per ? n_steps - done : per;
      StepParams Q = P;
      Q.actions = d_act ? d_act : d_actions;
      Q.action_sigma = sigma;
      Q.n_steps = k;
      if (++launch_serial == 0) launch_serial = 1;
      Q.launch_serial = launch_serial;
      Q.table_versions = (k > 1) ? 1 : 0;
      set_unroll_slot(Q);
      bk.launch_step(Q);
      P.step_count += (uint64_t)k;
      done += k;
    }
  }
  uint32_t launch_serial = 0;
  void step_random(float sigma) {
    if (!(sigma > 0.0f)) throw PmcError(LL_EINVAL, "sigma must be positive");
    step(nullptr, sigma);
  }
  // n_steps control steps of the random-policy loop in ONE launch: every wave walks its envs through the steps on its own, nothing
  // waits for the slowest wave of a step and there is no launch gap.  The sampling table is folded after EVERY step, by the last workgroup to
  // finish that step, into a version of its own; an episode that re-seeds at step s samples from the version steps 0 .. s - 1 left (PLE:235-240
  // as k single launches keep it), waiting for it if need be -- in practice only a wave that has run a whole step ahead of the slowest one waits.
  void step_random_n(float sigma, int n_steps) {
    if (!(sigma > 0.0f)) throw PmcError(LL_EINVAL, "sigma must be positive");
    if (n_steps <= 0) throw PmcError(LL_EINVAL, "n_steps must be positive");
    if (d_traj && (uint64_t)n_steps > (uint64_t)traj_unroll * (uint64_t)traj_buffers)
      // With unrolls recorded a launch writes n_steps consecutive rows of the ring: more than the ring holds and it would overwrite rows of
      // its own.  (Running from one block into the next is allowed -- the rows land where single steps would put them, bit for bit -- but then
      // the caller must have handed the next block over already, and every row of a launch carries the neglogp / value pair the pg buffers
      // held when it started: bench.py cuts its launches at unroll boundaries for that reason.)
      throw PmcError(LL_EINVAL, "ll_step_random_n: " + std::to_string(n_steps) + " steps do not fit the unroll ring of " + std::to_string(traj_unroll) + " x " +
                                std::to_string(traj_buffers) + " rows per env");
    step(nullptr, sigma, n_steps);
  }
  // parity hook: one control step whose physics result (and optionally foot positions) is supplied by the caller -- the
  // fake-BulletClient protocol of tests/golden/gen_golden.py -- so that everything around the physics can be compared with
  // the reference's own outputs directly
  float *d_script_state = nullptr, *d_script_feet = nullptr;
  void step_scripted(const float* d_act, const float* h_state, const float* h_feet) {
    need(true, true);
    const size_t N = P.n_envs;
    if (!d_script_state) { d_script_state = dalloc<float>(N * 37); d_script_feet = dalloc<float>(N * 24); }
    bk.sync();
    bk.h2d(d_script_state, h_state, N * 37 * 4);
    if (h_feet) bk.h2d(d_script_feet, h_feet, N * 24 * 4);
    StepParams Q = P;
    Q.actions = d_act ? d_act : d_actions;
    Q.scripted_state = d_script_state;
    Q.scripted_feet = h_feet ? d_script_feet : nullptr;
    set_unroll_slot(Q);
    bk.launch_step(Q);
    P.step_count += 1;
  }
  // parity probe for golden G8 (the torques the reference hands to PyBullet): runs the step kernel's own pd_target / pd_torque
  void probe_pd_torque(const float* h_rows, int n, int mode, float* h_tau) {
    if (n <= 0) throw PmcError(LL_EINVAL, "bad row count");
    float* d_in = (float*)bk.alloc((size_t)n * 36 * 4);
    float* d_out = (float*)bk.alloc((size_t)n * 12 * 4);
    try {
      bk.h2d(d_in, h_rows, (size_t)n * 36 * 4);
      bk.launch_probe_pd(P, d_in, d_out, n, mode);
      bk.sync();
      bk.d2h(h_tau, d_out, (size_t)n * 12 * 4);
    } catch (...) { bk.release(d_in); bk.release(d_out); throw; }
    bk.release(d_in); bk.release(d_out);
  }
  // record every env's transitions as unrolls in HBM, in the layout the learner consumes: n_buffers blocks of
  // [n_envs][unroll][row]; step s writes time step s % unroll of block (s / unroll) % n_buffers, so a finished block can be handed
  // off (RCCL gather) while the next one fills.
  void enable_unrolls(int unroll, int n_buffers) {
    if (unroll <= 0 || n_buffers <= 0) throw PmcError(LL_EINVAL, "unroll length and buffer count must be positive");
    if (d_traj) throw PmcError(LL_ESTATE, "unroll buffers already enabled");
    d_traj = dalloc<float>((size_t)n_buffers * P.n_envs * unroll * (P.obs_dim + LL_UNROLL_EXTRA));
    traj_unroll = unroll; traj_buffers = n_buffers;
    traj_base = P.step_count;       // unroll 0 starts with the NEXT control step, whatever ran before (warm-up, a previous learner phase)
  }
  uint64_t traj_base = 0;
  // index of the unroll the next control step writes into, and its time step there; unroll k lives in block k % n_buffers
  void unroll_position(int64_t* unroll_index, int* slot) const {
    if (!d_traj) throw PmcError(LL_ESTATE, "ll_enable_unrolls must be called first");
    const uint64_t rel = P.step_count - traj_base;
    *unroll_index = (int64_t)(rel / (uint64_t)traj_unroll);
    *slot = (int)(rel % (uint64_t)traj_unroll);
  }
  void set_unroll_slot(StepParams& Q) const {
    const uint64_t rel = P.step_count - traj_base;
    Q.traj = d_traj;
    Q.traj_unroll = traj_unroll;
    Q.traj_nbuf = traj_buffers;
    Q.traj_slot = traj_unroll ? (int)(rel % (uint64_t)traj_unroll) : 0;
    Q.traj_buf = traj_unroll ? (int)((rel / (uint64_t)traj_unroll) % (uint64_t)traj_buffers) : 0;
    Q.neglogp = d_neglogp; Q.value = d_value;
  }
  // TD(lambda) returns of one finished block (the actor-side post-processing of a PPO learner's data: R = GAE advantage + V):
  //   delta_t = r_t + gamma V_{t+1} m_t - V_t,  A_t = delta_t + gamma lam m_t A_{t+1},  R_t = A_t + V_t,  m_t = 1 - done_t,
  // V_T = bootstrap[env] (the value of the observation after the block's last step; masked when that step ended the episode)
  // the control step whose observation the value buffer was last written for by ll_policy_act_pg (~0: never -- the buffer holds zeros)
  uint64_t value_step = ~0ull;
  void finish_unroll(int buffer, float gamma, float lam, const float* d_bootstrap) {
    if (!d_traj) throw PmcError(LL_ESTATE, "ll_enable_unrolls must be called first");
    if (buffer < 0 || buffer >= traj_buffers) throw PmcError(LL_EINVAL, "buffer index out of range");
    // V_T must be the value of the observation AFTER the block's last step.  Right after that step the engine's value buffer still holds
    // V(obs_{T-1}), the estimate that went with the action just applied: a policy-gradient actor has to evaluate the new observation
    // (ll_policy_act_pg) before it closes the unroll, or hand the bootstrap values over explicitly.
    if (!d_bootstrap && value_step != ~0ull && value_step != P.step_count)
      throw PmcError(LL_ESTATE, "ll_finish_unroll: the value buffer was written for an earlier observation; call ll_policy_act_pg on the "
                                "observation that follows the unroll first, or pass d_bootstrap_value");
    bk.launch_gae(d_traj + (size_t)buffer * P.n_envs * traj_unroll * (P.obs_dim + LL_UNROLL_EXTRA), P.n_envs, traj_unroll, P.obs_dim + LL_UNROLL_EXTRA,
                  P.obs_dim, gamma, lam, d_bootstrap ? d_bootstrap : d_value);
  }
  void fill_random_actions(float sigma) {
    need(true, false);
    bk.launch_actions(P, d_actions, sigma);
  }

  // ---- boundary copies: rows on the host, SoA on the device ---------------------------------------------
  void get_soa(const float* d, int nf, float* h_rows) {
    const size_t N = P.n_envs;
    std::vector<float> tmp(nf * N);
    bk.sync();
    bk.d2h(tmp.data(), d, tmp.size() * 4);
    for (size_t e = 0; e < N; e++)
      for (int f = 0; f < nf; f++) h_rows[e * nf + f] = tmp[f * N + e];
  }
  void set_soa(float* d, int nf, const float* h_rows) {
    const size_t N = P.n_envs;
    std::vector<float> tmp(nf * N);
    for (size_t e = 0; e < N; e++)
      for (int f = 0; f < nf; f++) tmp[f * N + e] = h_rows[e * nf + f];
    bk.sync();
    bk.h2d(d, tmp.data(), tmp.size() * 4);
  }
  template <class T>
  void get_vec(const T* d, T* h, size_t n) {
    bk.sync();
    bk.d2h(h, d, n * sizeof(T));
  }
  void set_sampling_table(const double* avg_r) {
    need(true, false);
    const int C = P.n_clips;
    std::vector<double> prob(C), cdf(C);
    double sum = 0, acc = 0;
    for (int c = 0; c < C; c++) { prob[c] = pow(1.0 - avg_r[c], P.sample_factor); sum += prob[c]; }   // PLE:239-240
    for (int c = 0; c < C; c++) { prob[c] /= sum; acc += prob[c]; cdf[c] = acc; }
    cdf[C - 1] = 1.0;
    bk.sync();
    bk.h2d(d_avg_reward, avg_r, C * 8);
    bk.h2d(d_prob, prob.data(), C * 8);
    bk.h2d(d_cdf, cdf.data(), C * 8);
  }
};
