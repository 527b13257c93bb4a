/*
 * llenv_sepmc.h -- C ABI of the batched SEPMC (strategic-level, two robots per arena) env: SURVEY.md 8f-2.
 *
 * Replaces, for N arenas on one GPU, `ChaseTagGameEnv.reset()/.step()`
 *   CTG = src/lifelike/sim_envs/pybullet_envs/max_game/chase_tag_game_env.py
 *   BS4 = src/lifelike/sim_envs/pybullet_envs/max_game/bullet_static_entities.py (BulletStaticsV4, :830-1019)
 *   PR  = src/lifelike/sim_envs/pybullet_envs/randomizer/push_randomizer.py
 * behind the factory `create_chase_tag_game` (create_pybullet_envs.py:104-140).  Same conventions as llenv.h / llenv_epmc.h.
 * Every per-robot array is laid out [arena][robot] ("row" = 2 * arena + robot); the robot model blob is the one ll_create takes.
 */
#ifndef LLENV_SEPMC_H
#define LLENV_SEPMC_H

#include <stdint.h>

#include "llenv.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LLS_OBS_DIM_FIXED 830 /* percept_2d 325 + percept_1d 128 + percept_front 325 + percept_vec 5 + oppo_info 15 + oppo_info_cheat 15 +
                                 flag_info 7 + flag_info_cheat 7 + with_flag 2 + control_spd 1 (CTG:105-124); + prop, prop_a */
#define LLS_N_RAYS 778        /* per robot, as llenv_epmc.h */
#define LLS_N_VIS 21          /* rayTest slots per arena: 0 = base to base, 1 + 10 i + k = head of robot i to convex point k of the other
                                 (k: feet 0-3, wheels 4-7, front / hind handle 8-9; LR:150-156) */
#define LLS_MAX_BOXES 12      /* 4 walls + 5 cubes + hurdle + bar; the flag is reported separately */
#define LLS_MAX_DRAWS 64
#define LLS_MAX_CONTACTS 8

#define LLS_DONE_FALL 1       /* robot 0 fell (CTG:457-462; robot 1 falling does not end the episode) */
#define LLS_DONE_TIME 2
#define LLS_DONE_CATCH 8      /* a leg / wheel link of robot 0 touches robot 1 (CTG:442-450) */
#define LLS_DONE_NONFINITE 16

/* bodies of a contact record (ll_sepmc_step_scripted) and of ll_sepmc_get_episode's who fields */
#define LLS_BODY_PLANE 0
#define LLS_BODY_STATIC 1
#define LLS_BODY_FLAG 2
#define LLS_BODY_ROBOT0 3
#define LLS_BODY_ROBOT1 4

typedef struct ll_sepmc_config {
  int32_t abi_version; /* LL_ABI_VERSION (llenv.h) */
  int32_t n_arenas;
  int32_t device;
  int32_t auto_reset;
  double control_freq;     /* CPE:115 default 25.0 */
  double kp, kd, max_tau;  /* CPE:116-118 defaults 50, 1.0, 18 */
  int32_t max_steps;       /* CPE:120 default 1000 */
  int32_t prop_order[5];   /* LL_PROP_* ids, -1 terminated (CTG:90-104) */
  int32_t rand_cube, hurdle, hole; /* element_config (BS4:848-853) */
  int32_t solver_iterations;
  double friction_range[2];     /* CTG:279 */
  /* PR:8-54, as ll_epmc_config: counts evaluated by the host with Python's float floor division */
  int32_t push_enabled;
  int32_t push_count0, push_interval_step, push_duration_step;
  double horizontal_force[2], vertical_force[2], push_strength_ratio;
  double visible_angle;            /* CTG:31 default pi */
  double control_spd;              /* env_randomize_config['control_spd'] (CTG:361); < 0 = absent: the episode's draw */
  int32_t noise_enabled[4];        /* obs_randomization keys pos_x_bias, pos_y_bias, yaw_bias, pos_z_bias (CTG:207-210) */
  double noise_range[4][2];
  uint64_t seed;
  double max_tau_robot1;           /* torque limit of robot 1 when the two differ (max_tau given as a [lo, hi] list: one draw per LeggedRobot, LR:244,
                                      CTG:62-72); <= 0: the same as max_tau */
} ll_sepmc_config;

typedef struct ll_sepmc_engine ll_sepmc_engine;

/* CTG:22-161 for N arenas.  init_state37 = LeggedRobot.get_init_states_info() (LR:116-117); each arena starts its own copy of the
 * start orientation, which BOTH robots rotate in place at every reset (CTG:224-229). */
int ll_sepmc_create(const ll_sepmc_config* cfg, const double* model_blob, int blob_len, const double* init_state37, ll_sepmc_engine** out);
int ll_sepmc_destroy(ll_sepmc_engine* e);

/* CTG:263-310 for arena_ids[0..n) (NULL = all).  h_draws (nullable): n rows of LLS_MAX_DRAWS uniforms consumed in the reference's
 * draw order; h_prev_orn (nullable): n x 4 start orientations to rotate from. */
int ll_sepmc_reset(ll_sepmc_engine* e, const int32_t* arena_ids, int n, const float* h_draws, const float* h_prev_orn);
/* CTG:378-424 for every arena, one kernel launch.  d_actions: device [n_arenas][2][12] or NULL (engine buffer). */
int ll_sepmc_step(ll_sepmc_engine* e, const float* d_actions);
int ll_sepmc_set_actions(ll_sepmc_engine* e, const float* h_actions);
int ll_sepmc_fill_random_actions(ll_sepmc_engine* e, float sigma);
/* n_steps iterations of { ll_sepmc_fill_random_actions(sigma); ll_sepmc_step(NULL) } as ONE launch (llenv.h ll_step_random_n): every wavefront
 * walks its own arenas through the steps without waiting for the others.  The draws of a step (terrain, targets, pushes, re-seeds) are
 * keyed on (arena, episode, draw index), so the result is bit-identical to n_steps single steps. */
int ll_sepmc_step_random_n(ll_sepmc_engine* e, float sigma, int n_steps);
/* ll_sepmc_kernel_time_ms plus the number of control steps the timed launches ran */
int ll_sepmc_kernel_time_stats(ll_sepmc_engine* e, double* avg_launch_ms, int* n_launches, int64_t* n_control_steps);

/* Parity hook (how gen_sepmc_golden.py drove the reference through its fake BulletClient): one control step in which the caller
 * supplies what PyBullet would have returned -- both robot states after the ten substeps (h_state [n_arenas][2][37]), the answers
 * to the rayTestBatch calls per robot in THIS library's ray order (height 325, horizontal 128, front 325), the rayTest answers per
 * slot (h_vis_blocked [n_arenas][LLS_N_VIS]), the getContactPoints() list (h_contacts [n_arenas][LLS_MAX_CONTACTS][4]: bodyA,
 * bodyB, linkA, linkB; bodyA = -9 ends it) -- and the uniforms the step consumes (h_draws [n_arenas][n_draws]). */
int ll_sepmc_step_scripted(ll_sepmc_engine* e, const float* h_actions, const float* h_state, const uint8_t* h_ray_hit, const float* h_ray_frac,
                           const uint8_t* h_vis_blocked, const int32_t* h_contacts, const float* h_draws, int n_draws);
int ll_sepmc_set_step_draws(ll_sepmc_engine* e, const float* h_draws, int n_draws);
/* scripted ray / visibility answers for the NEXT ll_sepmc_reset only */
int ll_sepmc_script_reset(ll_sepmc_engine* e, const uint8_t* h_ray_hit, const float* h_ray_frac, const uint8_t* h_vis_blocked);

/* the constants of the physics spec that are this build's own choice (include/llenv_model.h LLM_SPEC_*), as ll_set_spec_param / ll_get_spec_param
 * of include/llenv.h: the robot and its solver are the PMC engine's */
int ll_sepmc_set_spec_param(ll_sepmc_engine* e, int id, double value);
int ll_sepmc_get_spec_param(ll_sepmc_engine* e, int id, double* value);
int ll_sepmc_sync(ll_sepmc_engine* e);
int ll_sepmc_obs_dim(ll_sepmc_engine* e);

int ll_sepmc_get_obs(ll_sepmc_engine* e, float* h_obs /*[n_arenas][2][obs_dim]: prop | prop_a | percept_2d | percept_1d | percept_front |
                                                          percept_vec | oppo_info | oppo_info_cheat | flag_info | flag_info_cheat | with_flag | control_spd*/);
int ll_sepmc_get_reward_done(ll_sepmc_engine* e, float* h_reward /*[n_arenas][2]*/, uint8_t* h_done /*[n_arenas]*/, uint8_t* h_done_reason);
int ll_sepmc_get_state(ll_sepmc_engine* e, float* h_state37 /*[n_arenas][2][37]*/);
int ll_sepmc_set_state(ll_sepmc_engine* e, const float* h_state37);
/* per arena: flag 3, with_flag[0], foot friction, episodic_fix_spd, counter, push force 3, noise 4, last_two_rob_pos_diff_len,
 * last_esc_flag_pos_diff_len, switch_flag_at_this_frame, oppo_visible 2, who-touches (_detect_body_contact, CTG:426-440) of robot 0 and
 * of the robot that could take the flag (LLS_BODY_*, -1 none) -> 21 floats */
int ll_sepmc_get_episode(ll_sepmc_engine* e, float* h_rows21);
/* avg_spd0, avg_spd1, max_spd0, max_spd1 of the last step (CTG:404-409) */
int ll_sepmc_get_info(ll_sepmc_engine* e, float* h_rows4);
/* the arena's boxes in creation order (walls, cubes, hurdle, bar): rows [x, y, z, hx, hy, hz]; h_count [n_arenas] */
int ll_sepmc_get_boxes(ll_sepmc_engine* e, float* h_rows /*[n_arenas][LLS_MAX_BOXES][6]*/, int32_t* h_count);
/* the perception rays of the last observation of each robot: from, to [n_arenas][2][778][3], hit, fraction [n_arenas][2][778] */
int ll_sepmc_get_rays(ll_sepmc_engine* e, float* h_from, float* h_to, uint8_t* h_hit, float* h_frac);
/* the visibility segments of the last observation by slot: from 3, to 3, blocked, asked (this library evaluates all 21; the reference
 * stops at the first clear one) -> [n_arenas][LLS_N_VIS][8] */
int ll_sepmc_get_vis(ll_sepmc_engine* e, float* h_rows);
/* the push force on each robot before each substep of the last step, [n_arenas][2][n_sub][4]: on/off, fx, fy, fz (PR:78-86) */
int ll_sepmc_get_push_trace(ll_sepmc_engine* e, float* h_rows, int32_t* n_sub);
int ll_sepmc_get_counters(ll_sepmc_engine* e, uint64_t* arena_steps, uint64_t* episodes, uint64_t* nonfinite);
int ll_sepmc_device_ptrs(ll_sepmc_engine* e, ll_device_ptrs_t* out);
int ll_sepmc_kernel_time_ms(ll_sepmc_engine* e, double* avg_ms, int* n_launches);
int ll_sepmc_enable_kernel_timing(ll_sepmc_engine* e, int on);

#ifdef __cplusplus
}
#endif
#endif
