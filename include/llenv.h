/*
 * llenv.h -- C ABI of the MI355X-native batched PMC tracking-environment stepper.
 *
 * The reference (Tencent-RoboticsX/lifelike-agility-and-play) has no C interface: its hot
 * path is a Python gym env that drives the PyBullet shared library through ~24 API calls
 * (SURVEY.md 2.3).  This header declares the entry points that a maintainer of the
 * reference would bind with ctypes to replace that path; every function names the
 * reference interface (file:line under src/lifelike/sim_envs/pybullet_envs/) it stands for.
 *
 *   PLE = primitive_level_env/primitive_level_env.py     ML = primitive_level_env/motion_lib.py
 *   LR  = legged_robot/legged_robot.py                   CPE = create_pybullet_envs.py
 *
 * Conventions
 *   - every function returns 0 on success or a negative LL_E* code; ll_last_error() gives text;
 *   - one handle per GPU; calls on one handle are not re-entrant (the TLeague actor drives its
 *     env from a single thread, learning/actors/distill_actor.py:205-247);
 *   - the caller owns every buffer it passes in; buffers named d_* are DEVICE pointers,
 *     h_* are HOST pointers;  all arrays are float32 unless stated otherwise;
 *   - no torch types cross this boundary.
 */
#ifndef LLENV_H
#define LLENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LL_ABI_VERSION 2   /* 2: ll_enable_trajectory -> ll_enable_unrolls (round 2); ll_step_random_n, ll_unroll_position, ll_pg_mark_current, ll_kernel_time_stats;
                              ll_sepmc_config grew max_tau_robot1 at its end (llenv_sepmc.h); spec id LLM_SPEC_LIMIT_SPECULATIVE (llenv_model.h) */

/* per-env sizes (PLE:101-124) */
#define LL_N_JOINTS 12
#define LL_STATE_DIM 37     /* pos3 quat4(xyzw) linvel3 angvel3 q12 qd12 : LR:86-106 states_info */
#define LL_ACTION_DIM 12    /* PLE:124 */
#define LL_STACK 3          /* PLE:39 stack_frame_num (not overridable from the factory) */
#define LL_PROP_FRAME_MAX 33
#define LL_FUTURE_DIM 72    /* PLE:120 */
#define LL_OBS_DIM_MAX (LL_STACK * LL_PROP_FRAME_MAX + LL_STACK * LL_ACTION_DIM + LL_FUTURE_DIM) /* 207 */
#define LL_MOCAP_ROW 19     /* ML:91-96 */

/* prop_type keys (PLE:102-108), ids used in ll_config.prop_order */
#define LL_PROP_JOINT_POS 0
#define LL_PROP_JOINT_VEL 1
#define LL_PROP_ROOT_LIN_VEL_LOC 2
#define LL_PROP_ROOT_ANG_VEL_LOC 3
#define LL_PROP_E_G 4

/* reward_weights order (PLE:352-357) */
#define LL_RW_JOINT_POS 0
#define LL_RW_JOINT_VEL 1
#define LL_RW_END_EFFECTOR 2
#define LL_RW_ROOT_POSE 3
#define LL_RW_ROOT_VEL 4

/* error codes */
#define LL_OK 0
#define LL_EINVAL (-1)
#define LL_ENOMEM (-2)
#define LL_EHIP (-3)      /* a HIP runtime call failed; message in ll_last_error() */
#define LL_ESTATE (-4)    /* call sequence error (e.g. step before mocap load / reset) */
#define LL_ENODEV (-5)    /* no usable GPU: the product path never falls back to the CPU */

/* done-reason bits written to the done_reason buffer (PLE:337-348) */
#define LL_DONE_FALL 1        /* LR:158-179 */
#define LL_DONE_CLIP_END 2    /* ML:168-172 */
#define LL_DONE_DIVERGED 4    /* PLE:319-335 */
#define LL_DONE_COLLISION 8   /* PLE:341-346 (obstacle variant) */
#define LL_DONE_NONFINITE 16  /* engine guard: a NaN/Inf state terminates the episode */

/*
 * Environment configuration = the reference's env_config dict (CPE:28-59) plus batching knobs.
 * Field defaults below are the factory's (CPE), not PrimitiveLevelEnv's.
 */
typedef struct ll_config {
  int32_t abi_version;          /* must be LL_ABI_VERSION */
  int32_t n_envs;               /* number of environments stepped in lockstep on this GPU */
  int32_t device;               /* HIP device ordinal */
  int32_t auto_reset;           /* 1: envs that finish are re-seeded inside the step kernel
                                   0: reference semantics, caller resets (PLE never auto-resets) */
  double control_freq;          /* CPE:29 default 25.0 (scripts use 50.0) */
  double sim_freq;              /* CPE:30 default 500.0 */
  double kp;                    /* CPE:31 */
  double kd;                    /* CPE:32 default 1.0 (scripts use 0.5) */
  double max_tau;               /* CPE:33; a [lo,hi] list is drawn ONCE by the host (LR:244, quirk Q1) */
  double foot_lateral_friction; /* CPE:48 */
  double reward_weights[5];     /* CPE:42 / PLE:352-363 (pass PLE's defaults when the dict is None) */
  int32_t prop_order[5];        /* prop_type list as LL_PROP_* ids, -1 terminated (PLE:101-113) */
  int32_t set_obstacle;         /* CPE:39 (obstacle variant; flat-terrain configs pass 0) */
  double obstacle_height;       /* CPE:40 default 0.0 (quirk Q7) */
  double prioritized_sample_factor; /* CPE:38 */
  int32_t solver_iterations;    /* LR:261 numSolverIterations = 10 */
  int32_t keep_terminal_obs;    /* auto_reset=1 only: also emit the obs of the finished episode (ll_get_terminal_obs).
                                   0 (default): an env that finishes writes one obs per step, the first of its next episode */
  uint64_t seed;                /* Philox key for clip / start-time sampling and synthetic actions */
} ll_config;

typedef struct ll_engine ll_engine; /* opaque */

/* Text of the last error on this thread (never NULL). */
const char* ll_last_error(void);
int ll_abi_version(void);

/* Length (doubles) of the model blob produced by the URDF compiler (urdf_model.py, = loadURDF LR:208-220). */
int ll_model_blob_len(void);

/*
 * Build an engine: replaces PrimitiveLevelEnv.__init__ (PLE:27-148) +
 * LeggedRobot._init_dynamic_model/_init_kinematic_model (LR:207-302) + loadURDF(plane) (PLE:81-82).
 */
int ll_create(const ll_config* cfg, const double* model_blob, int blob_len, ll_engine** out);
int ll_destroy(ll_engine* e); /* PLE:428-435 close()/__del__ */

/*
 * Upload the packed clip table: replaces MotionLib._open_all_mocap_datas (ML:19-46).
 * h_frames: [sum(clip_len)][19] float32 rows (ML:91-96 layout), clips concatenated.
 */
int ll_load_mocap(ll_engine* e, const float* h_frames, const int32_t* h_clip_len, int n_clips, double frame_step);
/* Same with float64 rows -- the numbers the reference parses from JSON (ML:31).  Preferred: reference velocities are
 * finite differences of neighbouring rows over 1/120 s (ML:137-160), so float32 rows already cost ~1e-4 m/s. */
int ll_load_mocap_f64(ll_engine* e, const double* h_frames, const int32_t* h_clip_len, int n_clips, double frame_step);

/*
 * Jump obstacles (only used when ll_config.set_obstacle != 0): replaces utils/obstacle.py:6-33 as consumed at
 * PLE:173-193 (_create_obstacle) and PLE:262-268 (_update_obstacle).  h_count[n_clips] obstacles per clip,
 * h_table[sum(count)][4] = x, y, yaw, peak time (float64).  The box is 0.05 x 1.0 x 2*obstacle_height, centred on the
 * ground (PLE:184-193, createMultiBody with mass 0).  The box is a collision body during the substeps of a set_obstacle engine
 * (pmc_step_kernel<OCC, true>: the robot is decelerated by it in the very step that touches it) and the termination test of
 * PLE:341-346: any robot shape within the contact distance of the box, where it stood during the substeps, ends the episode
 * (LL_DONE_COLLISION).  DESIGN.md 4 "Jump obstacle".
 */
int ll_load_obstacles(ll_engine* e, const int32_t* h_count, const double* h_table, int n_clips);

/*
 * Reset environments: PLE:150-171 + ML:48-63.
 *   h_env_ids   NULL = all n_envs, else n ids
 *   h_clip_idx  NULL = sample from the prioritized table (ML:59-63) with the engine's Philox stream
 *   h_t0        NULL = sample U(0,1)*frame_step*(N-margin-1) (ML:50-51); else explicit start times (float64)
 * After the call the obs buffer rows of those envs hold the first observation.
 */
int ll_reset(ll_engine* e, const int32_t* h_env_ids, int n, const int32_t* h_clip_idx, const double* h_t0);

/*
 * One 50 Hz control step for every env: PLE:195-245 (10 x {LR:119-148 PD torque, stepSimulation},
 * ML:65-115 mocap lookup, PLE:276-317 obs, PLE:350-426 reward, PLE:337-348 termination,
 * PLE:235-240 sampling-table update).  The real-time sleep of PLE:241-244 is NOT reproduced.
 *   d_actions  device pointer [n_envs][12] float32, or NULL to use the engine's own action buffer
 *              (see ll_device_ptrs) -- e.g. after ll_fill_random_actions().
 * Asynchronous on the engine's stream; results land in the engine's device buffers.
 */
int ll_step(ll_engine* e, const float* d_actions);

/*
 * Parity hook: ll_step with the physics result supplied by the caller.  After the (still executed) substeps the dynamic
 * state of every env is replaced by h_state[n_envs][37] and, if h_feet is given, the foot positions of LR:199-205 by
 * h_feet[n_envs][24] (4x3 dynamic robot, 4x3 ghost): the protocol with which tests/golden/gen_golden.py drove the
 * reference through a fake BulletClient.  Lets the mocap / observation / reward / termination / sampling-table code be
 * compared with the reference's own outputs without any physics in between.
 */
int ll_step_scripted(ll_engine* e, const float* d_actions, const float* h_state, const float* h_feet);

/*
 * Parity probe for LeggedRobot.apply_action (LR:119-148) and the target of PLE:199-200: the torques the reference hands to
 * setJointMotorControlArray(TORQUE_CONTROL, forces=...) before every stepSimulation, computed on the device by the very
 * functions the step kernel calls (pmc_step.hpp pd_target / pd_torque).
 *   h_rows [n][36] = joint_pos 12 | joint_vel 12 | x 12   (reference joint order FR1..3, FL1..3, HR1..3, HL1..3)
 *   mode 0: x = tgt_joint_pos as passed to apply_action (clipped to +-3 rad there, LR:126-127)
 *   mode 1: x = the policy action of a control step (target = joint_pos + action, PLE:199-200)
 *   h_tau  [n][12]  kp (target - q) + kd (0 - qd), clipped to +-max_tau (LR:137-141)
 * Synchronous.  Golden: tests/golden/pmc_config_golden.npz (G8).
 */
int ll_probe_pd_torque(ll_engine* e, const float* h_rows, int n, int mode, float* h_tau);

/*
 * Deviation study: move one constant of the physics spec that is this build's own choice (ids LLM_SPEC_* in
 * llenv_model.h: limit-row gate, depenetration cap, per-link damping, deepest-K, leg-leg rows on/off, their margin and count, ERP,
 * contact margin).  Takes effect from the next step.  The shipped defaults are the spec; nothing in the product calls this.
 */
int ll_set_spec_param(ll_engine* e, int id, double value);
int ll_get_spec_param(ll_engine* e, int id, double* value);

/* Synthetic random policy a ~ N(0, sigma^2) per joint (sigma = exp(-2)), generated on
 * device by Philox keyed on (seed, env, step) into the engine's action buffer. */
int ll_fill_random_actions(ll_engine* e, float sigma);

/* ll_fill_random_actions(e, sigma) followed by ll_step(e, NULL), as ONE kernel launch: the step kernel draws the same
 * Philox stream itself, records the actions in the engine's action buffer and applies them.  Stands for the reference
 * actor's random-policy loop (`env.step(np.random.randn(12) * sigma)`, learning/actors: SURVEY 8d). */
int ll_step_random(ll_engine* e, float sigma);
/*
 * n_steps iterations of that loop -- `for _ in range(n_steps): env.step(np.random.randn(12) * sigma)` for every env -- as ONE launch.
 * The random policy needs nothing from the host between two steps, so every wavefront walks its own environments through the n_steps
 * control steps without waiting for the others: no launch gap, and a slow step of one wavefront (self-collision rows, a re-seed) is
 * not a slow step of the whole chip.  Same Philox streams, same per-step outputs in the unroll buffers (ll_enable_unrolls) as n_steps
 * calls of ll_step_random; the obs / reward / done buffers hold the LAST step's values.  One difference, stated: the prioritized
 * sampling table (PLE:235-240) is folded once per launch, by its last workgroup, in the order one actor would have seen the episodes
 * end (later step first, then higher env) -- episodes that re-seed inside the launch sample from the table as it stood when the launch
 * began.  With prioritized_sample_factor = 0 (uniform sampling) the result is bit-identical to n_steps single-step calls.
 * With unrolls recorded (ll_enable_unrolls) a launch of more than unroll_length x n_buffers steps would overwrite rows of its own:
 * LL_EINVAL.  A launch may run from one unroll into the next (the rows land where single steps put them), but the block it runs into
 * must have been handed over by then -- cut launches at unroll boundaries (ll_unroll_position) when blocks are gathered asynchronously --
 * and every row of a launch records the neglogp / value pair the pg buffers held when the launch started (a random policy: zeros).
 */
int ll_step_random_n(ll_engine* e, float sigma, int n_steps);

/* Block until all queued work on the engine's stream has finished. */
int ll_sync(ll_engine* e);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream) instead of the engine's own, so that the
 * producer of d_actions and the consumers of the obs buffer are stream-ordered with the step kernel. NULL restores
 * the engine's private (non-blocking) stream -- so the legacy default stream, whose handle IS NULL, cannot be shared: give
 * torch and the engine an explicit stream (gather.bind_torch_stream). */
int ll_set_stream(ll_engine* e, void* hip_stream);

/* Device buffers owned by the engine (valid until ll_destroy), for zero-copy consumers (torch, RCCL). */
typedef struct ll_device_ptrs_t {
  float* obs;            /* [n_envs][obs_dim]  prop | prop_a | future (PLE:292-296) */
  float* reward;         /* [n_envs] */
  uint8_t* done;         /* [n_envs] */
  uint8_t* done_reason;  /* [n_envs] LL_DONE_* bits */
  float* actions;        /* [n_envs][12] engine-owned action buffer */
  float* terminal_obs;   /* [n_envs][obs_dim] obs of the finished episode (auto_reset=1 and keep_terminal_obs=1 only) */
  int32_t obs_dim;
  int32_t n_envs;
  void* stream;          /* hipStream_t the engine launches on */
} ll_device_ptrs_t;
int ll_device_ptrs(ll_engine* e, ll_device_ptrs_t* out);

/*
 * Unroll buffers for the learner hand-off: replace the actor's unroll list and its
 * structure -> flatten -> concatenate packing, learning/actors/distill_actor.py:118-162.  From now on every ll_step also writes the
 * transition it computes into a device buffer [n_buffers][n_envs][unroll_length][row_floats]: step s goes to time step
 * s % unroll_length of block (s / unroll_length) % n_buffers, so ONE ENV'S UNROLL IS CONTIGUOUS and -- read as float32 -- is the
 * `unroll_np` the reference pushes for that env (golden: tests/golden/unroll_golden.npz).  A row is one flattened time step:
 *     X    future[72] | prop[3 * prop_dim] | prop_a[36]    the observation the action was chosen on; the observation dict's keys
 *                                                         in sorted order, as the flatten of the (absent) tleague data structure
 *                                                         walks them [assumption stated in tests/golden/gen_unroll_golden.py]
 *     A    action[12]
 *     neglogp, R, V                                        the learner's remaining inputs (pmc_net.py:61-96: X, A, neglogp, R, V):
 *                                                         neglogp and V as found in the engine's ll_pg_ptrs buffers when the step
 *                                                         ran (written there by ll_policy_act_pg; zero if nobody did), R by
 *                                                         ll_finish_unroll
 *     r, 1 - done                                          what ll_finish_unroll computes R from
 * row_floats = obs_dim + 17.  The buffer is owned by the engine.
 * Unroll 0 starts with the first control step AFTER this call (whatever ran before: warm-up, scripted steps, an earlier phase); unroll k
 * lives in block k % n_buffers and is complete when ll_unroll_position reports unroll_index = k + 1, time_step = 0.
 */
int ll_enable_unrolls(ll_engine* e, int unroll_length, int n_buffers, float** d_base, int* row_floats);
/* Where the NEXT control step writes: the index of its unroll (counted from ll_enable_unrolls) and its time step inside it. */
int ll_unroll_position(ll_engine* e, int64_t* unroll_index, int* time_step);
/* Device buffers [n_envs] in which a policy leaves -log p(a|obs) and V(obs) for the actions it wrote into the action buffer.
 * Whoever writes them -- ll_policy_act_pg or the caller's own policy (e.g. torch kernels on the engine's stream) -- calls
 * ll_pg_mark_current afterwards: the stale-bootstrap guard of ll_finish_unroll is armed by the first mark and from then on refuses a
 * NULL bootstrap whose mark is not of the current step, whoever wrote the buffer.  A caller that never marks is never checked. */
int ll_pg_ptrs(ll_engine* e, float** d_neglogp, float** d_value);
/* Tell the engine that those buffers now hold the policy's outputs for the CURRENT observation (the one the next ll_step acts on).
 * A caller that fills them (ll_policy_act_pg on the engine's stream) calls this right after; ll_finish_unroll uses the stamp to refuse a
 * stale bootstrap value. */
int ll_pg_mark_current(ll_engine* e);
/*
 * TD(lambda) returns of block `buffer` (what the actor of a PPO learner computes before it pushes an unroll; gamma, lam:
 * example_pmc_train.sh:21-22): delta_t = r_t + gamma V_{t+1} m_t - V_t, A_t = delta_t + gamma lam m_t A_{t+1}, R_t = A_t + V_t with
 * m_t = 1 - done_t and V_T = d_bootstrap_value[env] (NULL: the engine's value buffer, i.e. the policy's estimate for the
 * observation that follows the block).  Call order with NULL: step the block's last step, evaluate the policy on the NEW observation
 * (ll_policy_act_pg + ll_pg_mark_current), then ll_finish_unroll -- right after the step the value buffer still holds V(obs_{T-1}).
 * Once ll_pg_mark_current has ever been called, a NULL bootstrap whose stamp is not the current step fails with LL_ESTATE (a buffer nobody
 * ever marked holds zeros: the random-policy benchmark).  Asynchronous on the engine's stream.
 */
int ll_finish_unroll(ll_engine* e, int buffer, float gamma, float lam, const float* d_bootstrap_value);

/* Host copies (synchronise the stream first). */
int ll_get_obs(ll_engine* e, float* h_obs /*[n_envs][obs_dim]*/);
int ll_get_terminal_obs(ll_engine* e, float* h_obs /*[n_envs][obs_dim]*/);
int ll_get_reward_done(ll_engine* e, float* h_reward, uint8_t* h_done, uint8_t* h_done_reason);
int ll_set_actions(ll_engine* e, const float* h_actions /*[n_envs][12]*/);

/* Dynamic-robot state, LR:86-106 / LR:62-84 layout, float32 [n_envs][37] on the host. */
int ll_get_state(ll_engine* e, float* h_state);
int ll_set_state(ll_engine* e, const float* h_state);
/* Kinematic ghost state (what PLE:218 writes into the ghost robot), [n_envs][37]. */
int ll_get_ref_state(ll_engine* e, float* h_state);
/* Episode bookkeeping: clip index (ML:60), env time in seconds as float64 (PLE:210), steps (PLE:197). */
int ll_get_episode_info(ll_engine* e, int32_t* h_clip, double* h_time, int32_t* h_steps, float* h_reward_sum);
/* Prioritized sampling table (PLE:131-136): probability[n_clips], avg_reward_sum[n_clips], avg_episode_len[n_clips]. */
int ll_get_sampling_table(ll_engine* e, double* h_prob, double* h_avg_reward_sum, double* h_avg_episode_len);
int ll_set_sampling_table(ll_engine* e, const double* h_avg_reward_sum);

/* World positions of the four feet of the dynamic robot and of the ghost: LR:199-205 compute_end_effector_info. */
int ll_get_feet(ll_engine* e, float* h_feet_dyn /*[n_envs][4][3]*/, float* h_feet_ref /*[n_envs][4][3]*/);

/* Counters for bench/diagnostics: total env-steps executed, episodes finished, non-finite resets. */
int ll_get_counters(ll_engine* e, uint64_t* steps, uint64_t* episodes, uint64_t* nonfinite);
/* Episodes that re-seeded inside a multi-step launch (ll_step_random_n) from an OLDER version of the sampling table than the exact one (the table as the
 * steps before theirs left it, primitive_level_env.py:235-240): zero whenever the launch had the chip to itself.  A re-seeding wavefront waits for the
 * version it needs only once every wavefront of its launch has started; while another kernel holds some of their SIMDs it takes the newest version there is. */
int ll_get_table_sync(ll_engine* e, uint64_t* stale_reseeds);
/* Finished episodes by length since creation: counts16[b] = episodes of 2^b .. 2^(b+1) - 1 control steps (b = 15: and longer); SURVEY 8d asks
 * for it next to the throughput so that the reset frequency of a benchmark is visible. */
int ll_get_episode_histogram(ll_engine* e, uint64_t* counts16);

/* Average device time (ms) of the step kernel over the launches since the last call, measured with
 * HIP events on the engine's own stream (bench.py roofline leg); also returns the launch count. */
int ll_kernel_time_ms(ll_engine* e, double* avg_ms, int* n_launches);
/* The same, plus the number of control steps those launches executed (ll_step_random_n launches run several). */
int ll_kernel_time_stats(ll_engine* e, double* avg_launch_ms, int* n_launches, int64_t* n_control_steps);
int ll_enable_kernel_timing(ll_engine* e, int on);

#ifdef __cplusplus
}
#endif
#endif /* LLENV_H */
