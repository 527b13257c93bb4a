/*
 * llenv_epmc.h -- C ABI of the batched EPMC (environmental-level) env: SURVEY.md 8f-1.
 *
 * Replaces, for N environments on one GPU, `PlayGroundEnv.reset()/.step()`
 *   PGE = src/lifelike/sim_envs/pybullet_envs/max_game_elements/playground_env.py
 *   BSE = src/lifelike/sim_envs/pybullet_envs/max_game_elements/bullet_static_entities.py
 *   PR  = src/lifelike/sim_envs/pybullet_envs/randomizer/push_randomizer.py
 * behind the factory `create_playground_game` (create_pybullet_envs.py:67-101).  Same conventions as llenv.h: extern "C",
 * plain pointers and sizes, int return codes (LL_OK / LL_E*), ll_last_error() for the text; one engine per GPU, calls on
 * an engine are not re-entrant.  The robot model blob is the one ll_create takes (llenv_model.h).
 *
 * Scope: the whole env logic (terrain generation, 778 rays against plane + boxes, observation, the joystick and average-speed
 * rewards, termination, push schedule, per-episode friction) and the articulated-body physics of llenv.h with the push force
 * and the terrain boxes / edge cylinders as obstacles.
 */
#ifndef LLENV_EPMC_H
#define LLENV_EPMC_H

#include <stdint.h>

#include "llenv.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LLE_OBS_DIM_FIXED 781 /* percep_2d 325 + percep_1d 128 + percep_front 325 + target 3 (PGE:128-139); + prop, prop_a */
#define LLE_N_RAYS 778        /* 325 height + 128 horizontal + 325 front (PGE:381-447) */
#define LLE_MAX_STATICS 104   /* rows of ll_epmc_get_statics: boxes, auxiliary cylinders, the target marker, in creation order */
#define LLE_MAX_DRAWS 64      /* draws one reset can consume */

#define LLE_DONE_FALL 1  /* LR:159-179 */
#define LLE_DONE_TIME 2  /* counter >= max_steps, PGE:368 */
#define LLE_DONE_REACH 4 /* within 0.5 m of the target, PGE:371 */
#define LLE_DONE_NONFINITE 16

typedef struct ll_epmc_config {
  int32_t abi_version; /* LL_ABI_VERSION (llenv.h) */
  int32_t n_envs;
  int32_t device;
  int32_t auto_reset;      /* 1: a finished env is re-seeded inside the step kernel; 0: reference semantics */
  double control_freq;     /* CPE:79 default 50.0 */
  double kp, kd, max_tau;  /* CPE:80-82 defaults 50, 1.0, 16 */
  int32_t max_steps;       /* CPE:84 default 1000 */
  int32_t prop_order[5];   /* LL_PROP_* ids, -1 terminated (PGE:108-122) */
  int32_t element_id;      /* env_randomize_config['element_id']: 0 joystick, 1 hurdles, 2 holes, 3 cubes (BSE:236-250) */
  int32_t solver_iterations;
  double friction_range[2];     /* PGE:92, :209 foot lateral friction ~ U(range) per episode */
  /* PR:8-54.  The three counts are (-start_time) // dt, interval_time // dt, duration_time // dt evaluated by the HOST with
   * Python's float floor division (e.g. 1.0 // 0.002 == 499.0, not 500): the kernel must not re-derive them */
  int32_t push_enabled;
  int32_t push_count0, push_interval_step, push_duration_step;
  double horizontal_force[2], vertical_force[2], push_strength_ratio;
  int32_t cmd_vary_freq_range[2];  /* PGE:169 default [25, 200] */
  double target_spd_range[2];      /* PGE:313 */
  double auxiliary_radius;         /* BSE:16; < 0 = None */
  double hole_gap_height[2];       /* hole_config min/max_gap_height (BSE:374-375), used when element_id == 2 */
  int32_t noise_enabled[4];        /* obs_randomization keys pos_x_bias, pos_y_bias, yaw_bias, pos_z_bias (PGE:176-179) */
  double noise_range[4][2];
  uint64_t seed;
} ll_epmc_config;

typedef struct ll_epmc_engine ll_epmc_engine;

/* PGE:58-174 + LR:207-264: builds N envs.  init_state37 = LeggedRobot.get_init_states_info() (LR:116-117) as
 * pos3 quat4(xyzw) linvel3 angvel3 q12 qd12; every env starts its own copy, which randomize_init_states then rotates IN
 * PLACE at every reset (PGE:186-190: the yaw draws of successive episodes accumulate). */
int ll_epmc_create(const ll_epmc_config* cfg, const double* model_blob, int blob_len, const double* init_state37, ll_epmc_engine** out);
int ll_epmc_destroy(ll_epmc_engine* e);

/* PGE:196-249 for env_ids[0..n) (NULL = all).  h_draws (nullable) = n rows of LLE_MAX_DRAWS uniforms in [0,1) consumed in
 * the reference's draw order instead of the engine's Philox stream (parity: replay of a recorded np.random log);
 * h_prev_orn (nullable) = n x 4 start orientations to rotate from (instead of the env's accumulated one). */
int ll_epmc_reset(ll_epmc_engine* e, const int32_t* env_ids, int n, const float* h_draws, const float* h_prev_orn);

/* PGE:299-364 for every env, one kernel launch.  d_actions: device [n_envs][12] or NULL (engine buffer). */
int ll_epmc_step(ll_epmc_engine* e, const float* d_actions);
int ll_epmc_set_actions(ll_epmc_engine* e, const float* h_actions);

/* Parity hook (how gen_epmc_golden.py drove the reference through its fake BulletClient): one control step in which the
 * caller supplies what PyBullet would have returned -- the robot state after the ten substeps (h_state [n_envs][37]),
 * the answers to the three rayTestBatch calls (h_ray_hit [n_envs][778] 0/1, h_ray_frac [n_envs][778]) -- and, optionally,
 * the uniforms the step consumes (h_draws [n_envs][n_draws]; PGE:303, :313, PR:88-98). */
int ll_epmc_step_scripted(ll_epmc_engine* e, const float* h_actions, const float* h_state, const uint8_t* h_ray_hit, const float* h_ray_frac,
                          const float* h_draws, int n_draws);
/* Uniforms in [0,1) for the draws of the NEXT ll_epmc_step only, [n_envs][n_draws], consumed in the reference's order (joystick
 * target angle PGE:303, target speed PGE:313, push direction / horizontal / vertical PR:88-98) instead of the engine's Philox
 * stream: lets a host that owns the random stream (the 1-env shim replays NumPy's global one) keep it authoritative. */
int ll_epmc_set_step_draws(ll_epmc_engine* e, const float* h_draws, int n_draws);
/* scripted ray answers for the NEXT ll_epmc_reset only (the reset observation casts rays too) */
int ll_epmc_script_reset_rays(ll_epmc_engine* e, const uint8_t* h_ray_hit, const float* h_ray_frac);

/* the constants of the physics spec that are this build's own choice (include/llenv_model.h LLM_SPEC_*), as ll_set_spec_param / ll_get_spec_param
 * of include/llenv.h: the robot and its solver are the PMC engine's */
int ll_epmc_set_spec_param(ll_epmc_engine* e, int id, double value);
int ll_epmc_get_spec_param(ll_epmc_engine* e, int id, double* value);
int ll_epmc_sync(ll_epmc_engine* e);
int ll_epmc_obs_dim(ll_epmc_engine* e);

int ll_epmc_get_obs(ll_epmc_engine* e, float* h_obs /*[n_envs][obs_dim]: prop | prop_a | percep_2d | percep_1d | percep_front | target*/);
int ll_epmc_get_reward_done(ll_epmc_engine* e, float* h_reward, uint8_t* h_done, uint8_t* h_done_reason);
int ll_epmc_get_state(ll_epmc_engine* e, float* h_state37);
int ll_epmc_set_state(ll_epmc_engine* e, const float* h_state37);
/* per env: target_pos 3, target_spd, foot friction, cmd_vary_freq, counter, push force 3, noise 4, last_pos_diff_len,
 * init_pos_diff_len (-1 = None), total_spd, max_spd  -> 19 floats */
int ll_epmc_get_episode(ll_epmc_engine* e, float* h_rows19);
/* the info dict of the episode an env last finished (PGE:356-362): ave_spd, max_spd, reward_vel, reward_rotation,
 * reward_dist, reward_avg_spd */
int ll_epmc_get_info(ll_epmc_engine* e, float* h_rows6);
/* BSE bodies of each env in creation order: rows [kind 0 box / 1 cylinder along y, x, y, z, a, b, c, 0]; h_count [n_envs] */
int ll_epmc_get_statics(ll_epmc_engine* e, float* h_rows /*[n_envs][LLE_MAX_STATICS][8]*/, int32_t* h_count);
/* the rays of the last observation of each env: from, to [n_envs][778][3], hit [n_envs][778], fraction [n_envs][778] */
int ll_epmc_get_rays(ll_epmc_engine* e, float* h_from, float* h_to, uint8_t* h_hit, float* h_frac);
/* the push force applied before each substep of the last step, [n_envs][n_sub][4]: on/off, fx, fy, fz (PR:56-86) */
int ll_epmc_get_push_trace(ll_epmc_engine* e, float* h_rows, int32_t* n_sub);
int ll_epmc_get_counters(ll_epmc_engine* e, uint64_t* env_steps, uint64_t* episodes, uint64_t* nonfinite);
int ll_epmc_device_ptrs(ll_epmc_engine* e, ll_device_ptrs_t* out);
int ll_epmc_kernel_time_ms(ll_epmc_engine* e, double* avg_ms, int* n_launches);
int ll_epmc_enable_kernel_timing(ll_epmc_engine* e, int on);
int ll_epmc_fill_random_actions(ll_epmc_engine* e, float sigma);
/* n_steps iterations of { ll_epmc_fill_random_actions(sigma); ll_epmc_step(NULL) } as ONE launch (llenv.h ll_step_random_n): every wavefront
 * walks its own envs through the steps without waiting for the others.  The draws of a step (terrain, targets, pushes, re-seeds) are
 * keyed on (env, episode, draw index), so the result is bit-identical to n_steps single steps. */
int ll_epmc_step_random_n(ll_epmc_engine* e, float sigma, int n_steps);
/* ll_epmc_kernel_time_ms plus the number of control steps the timed launches ran */
int ll_epmc_kernel_time_stats(ll_epmc_engine* e, double* avg_launch_ms, int* n_launches, int64_t* n_control_steps);

#ifdef __cplusplus
}
#endif
#endif
