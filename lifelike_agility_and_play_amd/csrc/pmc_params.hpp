// pmc_params.hpp -- kernel argument block and constant-table layouts of the PMC step engine.
#pragma once
#include <stdint.h>

#define LL_UNROLL_EXTRA 17   // floats of an unroll row after the observation: A 12, neglogp, R, V, r, 1 - done
#define PMC_K 4            // contact slots per leg lane (== LLM_MAX_CONTACTS_PER_LEG)
#define PMC_WAVE 64
#define PMC_ENVS_PER_WAVE 4     // one env = one 16-lane DPP row

// ---- per-leg constant table: legc[field * 4 + leg] ---------------------------------------------------
enum LegConst {
  LC_R1 = 0,          // 3  hip joint origin in F0
  LC_R2 = 3,          // 3  thigh joint origin in hip frame
  LC_R3 = 6,          // 3  shank joint origin in thigh frame
  LC_M = 9,           // 3  link masses
  LC_COM = 12,        // 9  link COMs (link frame)
  LC_IC = 21,         // 18 link inertias about COM: xx xy xz yy yz zz
  LC_QLO = 39,        // 3
  LC_QHI = 42,        // 3
  LC_JDAMP = 45,      // 3
  LC_FOOT = 48,       // 3  foot link origin in the shank frame
  LC_BSX = 51,        // base-box vertex sign x owned by this lane
  LC_BSY = 52,        // base-box vertex sign y owned by this lane
  LC_HANDLE = 53,     // 4  handle sphere centre (F0) + radius
  LC_HAS_HANDLE = 57, // 1.0 if this lane owns a handle sphere
  LC_HIPCYL = 58,     // 11 cylinder on the hip link:   c(3) axis(3) fallback-dir(3) r h
  LC_THBOX = 69,      // 12 box on the thigh link:      c(3) ua(3) ub(3) uc(3)   (scaled half-axis vectors)
  LC_THCYL0 = 81,     // 11
  LC_THCYL1 = 92,     // 11
  LC_WHEEL = 103,     // 11 wheel cylinder (welded to the thigh)
  LC_SHBOX = 114,     // 12 box on the shank link
  LC_FOOTSPH = 126,   // 4  foot sphere on the shank link: c(3) r
  LC_CAPS = 130,      // 14 self-collision capsules: thigh a(3) b(3) in the thigh frame, shank a(3) b(3) in the shank frame, r_thigh, r_shank
  LC_TRUNKCAP = 144,  // 7  robot-robot collision (SEPMC): trunk capsule (leg & 1) of the two that stand for the body box: a(3) b(3) in F0, r
  LC_COUNT = 151
};

// ---- contact candidate table: candc[(jj * CF_WORDS + field) * 16 + leg * 4 + sub], 7 candidates per (leg, sub) --------------
//   a candidate is a point of a link:  P = A - r * n(ez_link),  n = unit(ez - (ez.ax) ax)  (fb if degenerate)
//   sphere: A = centre, ax = 0;  box vertex: A = vertex, r = 0;  cylinder cap: A = cap centre, ax = axis
enum CandField { CF_A = 0, CF_AX = 3, CF_FB = 6, CF_R = 9, CF_LINK = 10, CF_KIND = 11, CF_WORDS = 12 };
#define CAND_PER_SUB 8   // 7 on flat ground (PMC); the eighth, a sphere on the link's axis, only matters against terrain (TERRAIN builds)
// ... followed by LK_WORDS words per lane: the constants of the link the sub-lane owns in the dynamics (sub-lane k < 3: link k + 1 of the
// leg -- mass, COM in the link frame, inertia about the COM xx xy xz yy yz zz; sub-lane 3: zeros)
#define LK_BASE (CAND_PER_SUB * CF_WORDS)
#define LK_WORDS 10
#define CAND_TABLE_WORDS (CAND_PER_SUB * CF_WORDS + LK_WORDS)

// ---- base constant table (quad-uniform) ----------------------------------------------------------------
enum BaseConst {
  BC_MASS = 0,
  BC_H = 1,       // 3  m * com
  BC_IO = 4,      // 6  inertia about the F0 origin: xx xy xz yy yz zz
  BC_COM = 10,    // 3
  BC_ICOM = 13,   // 6  inertia about the COM
  BC_BOX = 19,    // 12 body box: c(3) ua(3) ub(3) uc(3)
  BC_COUNT = 31
};

// Timing ablations (tools/ablate.sh builds a separate library with -DPMC_ABLATION; the shipped one has no such switch):
// LL_DEBUG_FLAGS bit 1 drops contact rows, 2 drops limit rows, 4 returns after the substeps, 8 returns at kernel entry.
// Bit 16 records per-env wall-clock stamps (100 MHz) at PMC_TS(k) marks into the words after P.counters[4].
#if defined(PMC_ABLATION)
#define PMC_ABL(bit) ((P.debug_flags & (bit)) != 0)
#define PMC_TS_SLOTS 32
#define PMC_TS(k) do { if (PMC_ABL(16) && ln.is_lane(0)) P.counters[4 + (long)env * PMC_TS_SLOTS + (k)] = wall_clock64(); } while (0)
#else
#define PMC_ABL(bit) false
#define PMC_TS_SLOTS 0
#define PMC_TS(k) do { } while (0)
#endif
// The table versions of a multi-step launch (ver_ready, cdf_ver: llenv.hip step_done_fold, pmc_step.hpp table_for_reseed) travel by relaxed device-scope atomics, ordered by the
// s_waitcnt / issue order of gfx950.  LL_VER_FENCE says the same in the memory model: a release fence ahead of a version's mark, an acquire fence behind the load that saw it
// (agent scope; only the one folding wave per step and the re-seeding waves meet them).  -DLL_VER_FENCES=0 is the A/B leg without them.
#ifndef LL_VER_FENCES
#define LL_VER_FENCES 1
#endif
#if LL_VER_FENCES && defined(__HIP_DEVICE_COMPILE__)
#define LL_VER_FENCE(order) __builtin_amdgcn_fence(order, "agent")
#else
#define LL_VER_FENCE(order) do { } while (0)
#endif

// Static phase marks (tools/issue_ledger.py compiles an assembly LISTING with -DPMC_MARKS; no library is ever built with it): a comment in the instruction stream where a
// phase of the step begins, fenced by a scheduling barrier so that the instructions of a phase stay between its marks.  The ledger attributes the static instructions of the
// listing to phases with them (profiles/r06_issue_ledger.md); the shipped build has no marks and schedules across these points, so the marked listing is a close cousin, not the same code.
#if defined(PMC_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define PMC_PHASE(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; LLPHASE " name ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PMC_PHASE(name) do { } while (0)
#endif

#ifndef PMC_PGS_HOIST
#define PMC_PGS_HOIST 1           // pmc_step.hpp: the solver's iteration loop in six copies (limit rows? x 0 / 1 / 2 leg-leg slots) with its wave-uniform tests decided outside (0: the one generic loop)
#endif

#ifndef PMC_PGS_HOIST_PAIR
#define PMC_PGS_HOIST_PAIR 1      // the chase-tag builds too (- 0.5 % at 2048 arenas, - 1.4 % at 32768; 0: their one generic loop)
#endif
#ifndef PMC_PGS_HOIST_TERRAIN
#define PMC_PGS_HOIST_TERRAIN 0   // 1: the one-wave-per-SIMD playground build too (measured: no difference; its 256-register build has the copies, - 9 % at 65536 envs)
#endif

#define LL_MAX_STEPS_PER_LAUNCH 128   // control steps one launch of ll_step_random_n runs at most (longer calls are split); sizes the per-step table slots

struct StepParams {
  static constexpr bool kFirstKernelArgument = true;   // every kernel takes it first, by value: lanes.hpp WithParamsReload re-reads it from the kernarg segment
  int32_t n_envs, n_sub, n_iter, auto_reset;
  int32_t n_clips, prop_dim, obs_dim, frame_rate;
  int32_t margin, envs_per_wave;
  int32_t prop_off[5];      // offset of each LL_PROP_* key inside one prop frame, or -1
  int32_t keep_term_obs;    // auto-reset: also write the finished episode's last obs to term_obs
  float dt, kp, kd, max_tau;
  float max_tau1, pad6;     // SEPMC: robot 1's torque limit when it differs (> 0; max_tau given as a list draws one per LeggedRobot, LR:244)
  float mu_foot, mu_link, gravity, link_damping;
  float erp, margin_dist, limit_gate, self_collision;   // self_collision: 1 = links of different legs collide (LR:212-217)
  float rw[5];              // normalised reward weights (PLE:365-370)
  float max_depen;          // cap on the penetration-recovery speed of a contact row (LLM_MAX_DEPEN_SPEED)
  float self_margin;        // leg-leg capsule rows start within this distance (LLM_SELF_MARGIN)
  int32_t friction_dirs;    // LLM_SPEC_FRICTION_DIRS: 1 = first friction direction along the contact point's sliding velocity (default 0: btPlaneSpace1)
  int32_t friction_mode;    // LLM_SPEC_FRICTION_MODE: 2 (LLM_FRICTION_MODE, the spec since round 4) = the two friction rows of a contact solved together inside the
                            // cone (Bullet's published default; Pmc::gs_cone_round); 0 = the pyramid of rounds 1 - 3 (all t1 rows, then all t2 rows, box bounds).
                            // Every step kernel has both builds; the launch picks
  float limit_erp;          // ERP of the joint-limit rows as the kernel uses it (LLM_SPEC_LIMIT_ERP; the setter resolves "< 0 = erp")
  float erp_deep, limit_erp_deep, erp_deep_below;   // LLM_SPEC_ERP_DEEP / _BELOW as the kernel uses them: a row deeper than erp_deep_below takes erp_deep (limit rows: limit_erp_deep);
                            // without a second ERP both equal erp / limit_erp, so the kernel selects unconditionally
  float spec_erp_deep, spec_limit_erp, spec_limit_erp_deep;   // what ll_set_spec_param was given (< 0: follow erp), kept for ll_get_spec_param and for re-resolving when erp moves
  int32_t limit_speculative;   // LLM_SPEC_LIMIT_SPECULATIVE: 1 = a limit row inside the range too (gated by limit_gate); 0 = Bullet's rule, a row only once the limit is passed
  float max_coord_vel;      // LLM_SPEC_MAX_COORD_VEL (btMultiBody::m_maxCoordinateVelocity, 100): base twist and joint rates clipped after the unconstrained update and after the solve
  int32_t max_contacts, max_self;   // deepest-K per leg (LLM_MAX_CONTACTS_PER_LEG), self-collision rows per robot (LLM_MAX_SELF)
  float self_friction, pair_friction;   // LLM_SPEC_SELF_FRICTION / LLM_SPEC_PAIR_FRICTION (round 6: engine twins of the oracle switches): mu of the two tangential rows a leg-leg / robot-robot
                                        // contact carries behind its normal row (box bounds +- mu x the normal multiplier, btPlaneSpace1 directions); 0 = frictionless (the spec of rounds 1 - 5)
  int32_t max_pair, leg_edges;          // LLM_SPEC_MAX_PAIR: robot-robot rows per robot pair, 2 (rounds 1 - 5) .. 4 (a manifold's four points); LLM_SPEC_LEG_EDGES: terrain edges across the leg boxes are contact candidates (round 6)
  double dt_d, frame_step, policy_step, sample_factor;
  uint64_t seed;
  uint64_t step_count;      // control steps executed so far (salts the Philox stream)

  // per-env state, SoA [field][n_envs]
  float* state;             // 37 fields: pos3 quat4 linvel3 angvel3 q12 qd12
  float* kin;               // 37 fields: ghost (reference) state
  float* feet;              // 24 fields: dyn feet 4x3, ghost feet 4x3 (world)
  double* time;             // [n_envs] env time (PLE:210)
  int32_t* clip;            // [n_envs]
  int32_t* ep_steps;        // [n_envs]
  float* reward_sum;        // [n_envs]
  uint32_t* ep_count;       // [n_envs] episodes started (Philox counter)
  // per-env outputs, AoS rows
  float* obs;               // [n_envs][obs_dim]   also the history store (frames 1,2 feed the next step)
  float* term_obs;          // [n_envs][obs_dim]
  float* reward;            // [n_envs]
  uint8_t* done;            // [n_envs]
  uint8_t* done_reason;     // [n_envs]
  const float* actions;     // [n_envs][12]
  // parity hook (ll_step_scripted): physics result and foot positions supplied by the caller, rows [n_envs][37] / [n_envs][24]
  const float* scripted_state;
  const float* scripted_feet;
  // optional unroll buffers for the learner hand-off (ll_enable_unrolls): [n_buffers][n_envs][unroll][obs_dim + LL_UNROLL_EXTRA], one
  // env's unroll contiguous and its time step laid out the way the reference's actor flattens it (distill_actor.py:118-162):
  //   X: future 72 | prop | prop_a 36   (the observation dict's keys in sorted order)   | A 12 | neglogp | R | V | r | 1 - done
  float* traj;
  int32_t traj_slot, traj_buf, traj_unroll, traj_nbuf;   // slot / block of the launch's FIRST control step; a launch of n_steps walks on from there
  const float* neglogp;     // [n_envs] what the policy reported for the actions being applied (ll_pg_ptrs), copied into the unroll
  const float* value;       // [n_envs]
  // mocap
  const double* frames;     // [total][19]
  const int32_t* clip_off;
  const int32_t* clip_len;
  const double* max_steps;  // [n_clips]
  double* cdf;              // [n_clips] inclusive prefix sums of the sampling probabilities
  double* prob;             // [n_clips] p ~ (1 - avg_reward_sum)^factor, normalised (PLE:239-240)
  double* avg_reward;       // [n_clips] _avg_reward_sum (PLE:235-238)
  double* avg_len;          // [n_clips] avg_episode_len
  unsigned int* block_ticket;   // [LL_MAX_STEPS_PER_LAUNCH] per control step of the launch: workgroups that have finished it; the last one folds that step's
                                // finished episodes into the table (PLE:235-240), so a launch of k steps leaves behind -- and re-seeds from -- the tables k launches would
  // ---- the sampling table INSIDE a multi-step launch (ll_step_random_n): one version per control step, so that an episode that re-seeds at step s of a launch
  // samples from the table as the steps before s left it -- what k single launches do
  double* cdf_ver;              // [LL_MAX_STEPS_PER_LAUNCH][n_clips]: slot s (1 <= s < n_steps) = the CDF after step s - 1 of this launch; the launch's last step writes `cdf` itself
  unsigned int* ver_ready;      // [LL_MAX_STEPS_PER_LAUNCH]: slot s holds launch_serial once step s of this launch has been folded (device-scope release by the folding wave)
  unsigned int* resident;       // waves of the running launch that have started: a re-seeding wave only WAITS for a version when all of them have (then every wave is running
                                // or done, so the wait ends); otherwise it takes the newest version there is and counts the episode in counters[3]
  uint32_t launch_serial;       // distinguishes this launch's ver_ready marks from older ones (never 0)
  int32_t table_versions;       // 1: re-seeds inside a multi-step launch read the per-step versions (set by the engine when the grid is co-resident and n_steps > 1)
  float* actions_out;       // where a step that draws its own actions (action_sigma > 0) records them, [n_envs][12]
  float action_sigma;
  int32_t n_steps;           // control steps per launch (ll_step_random_n); 1 everywhere else
  // jump obstacles (set_obstacle): per clip offset/count into ob_table[total][4] = x, y, yaw, peak time
  const int32_t* ob_off;
  const int32_t* ob_cnt;
  const double* ob_table;
  int32_t* ob_id;           // [n_envs] current obstacle of the episode (PLE:179,:264-265)
  float ob_half_height, ob_pad;
  int32_t set_obstacle, debug_flags;   // debug_flags: read only by builds with -DPMC_ABLATION (tools/ablate.sh), 0 otherwise
  unsigned long long* pending_reward;  // [LL_MAX_STEPS_PER_LAUNCH][n_clips] per control step of the launch: packed (env+1)<<32 | float bits of reward_sum/max_steps
  unsigned long long* pending_len;     // [LL_MAX_STEPS_PER_LAUNCH][n_clips] packed (env+1)<<32 | float bits of avg_episode_len
  unsigned long long* counters;        // [4] env-steps, episodes, non-finite resets, episodes re-seeded from a table version older than the exact one (see `resident`)
  unsigned long long* ep_hist;         // [16] finished episodes by length: bucket b counts lengths in [2^b, 2^(b+1)) control steps (the last: and longer)
  // constants
  const float* legc;        // [LC_COUNT][4]
  const float* candc;       // [CAND_TABLE_WORDS][16]
  const float* basec;       // [BC_COUNT]
};

// the launch needs the build with the extended contact rows (Pmc::substep_impl<.., XROWS = true>): one of the round-6 switches is off its default
LL_HD bool pmc_wants_xrows(const StepParams& P) { return P.self_friction > 0.0f || P.pair_friction > 0.0f || P.max_pair != LLM_MAX_PAIR; }
// ... and in the terrain envs (EPMC, SEPMC) the same builds carry the terrain edges across leg boxes (LLM_SPEC_LEG_EDGES)
LL_HD bool pmc_wants_xrows_terrain(const StepParams& P) { return pmc_wants_xrows(P) || P.leg_edges != 0; }
