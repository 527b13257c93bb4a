// epmc_step.hpp -- the EPMC (environmental-level) control step, generic over the lane policy like pmc_step.hpp.
//
// What it replaces, cited as
//   PGE = src/lifelike/sim_envs/pybullet_envs/max_game_elements/playground_env.py
//   BSE = src/lifelike/sim_envs/pybullet_envs/max_game_elements/bullet_static_entities.py
//   PR  = src/lifelike/sim_envs/pybullet_envs/randomizer/push_randomizer.py
//
// The physics substep is Pmc<L>::substep (pmc_step.hpp) with two additions: the episode's foot friction and the push force
// on the FR hip link.  Everything else here is per-env scalar work (every lane of the env's row computes it; stores come
// from lane 0) except the 778 rays, which the 16 lanes of the row share out.
#pragma once
#include "pmc_step.hpp"

#define EPMC_N_HEIGHT 325
#define EPMC_N_HORIZ 128
#define EPMC_N_FRONT 325
#define EPMC_N_RAYS 778
#define EPMC_MAX_STATICS 104
#define EPMC_MAX_BOXES 40
#define EPMC_MAX_DRAWS 64
#define EPMC_BOX_WORDS 8
#define EPMC_MAX_NEAR 8       // boxes within reach of the robot's contact candidates during one control step
#define EPMC_EP_STRIDE 40
// -DLL_NO_FUSED_RAYS=1: an A/B build whose step kernels carry no ray code at all (they always leave the ray pose; run it with LL_SPLIT_RAYS=2) -- what the dead code costs the split path
#ifndef LL_NO_FUSED_RAYS
#define LL_NO_FUSED_RAYS 0
#endif
#define EPMC_RAY_POSE 24        // floats per row in EpmcParams::ray_pose: pos 3 | R 9 | yaw | noise_z | n_boxes | 1 if the last box is overridden | that box record 8
#define EPMC_LIST_A 320         // row-scratch words: behind the staged box records the three ray lists (height grid, fan, front rays), then 64 spare words.
#define EPMC_LIST_A_MAX 10      // (PMC_ROW_SCRATCH = 688 words per row is what eight workgroups per CU can afford: 9392 B of tables + 4 x 2752 B <= 160 KB / 8)
#define EPMC_LIST_B (EPMC_LIST_A + EPMC_LIST_A_MAX * EPMC_BOX_WORDS)
#define EPMC_LIST_B_MAX 16
#define EPMC_LIST_C (EPMC_LIST_B + EPMC_LIST_B_MAX * EPMC_BOX_WORDS)
#define EPMC_LIST_C_MAX 12
#define EPMC_SPARE (EPMC_LIST_C + EPMC_LIST_C_MAX * EPMC_BOX_WORDS)
#define RAY_CHUNK (L::kRayChunk)  // rays a lane carries through one walk of its family's box list (lanes.hpp WithRayChunk; 325 rays / 16 lanes = 21 = 3 x 7)
#define EPMC_PARK_AT EPMC_SPARE  // row-scratch word where the episode scalars wait during the substep loop (step_env<PARK>)
static_assert(EPMC_SPARE + 64 == PMC_ROW_SCRATCH, "row scratch layout: boxes, three ray lists, 64 spare words");

// per-env scalar row (EpmcParams::ep)
enum EpmcEpField {
  EP_TARGET = 0,      // 3
  EP_TARGET_SPD = 3,
  EP_FRICTION = 4,    // the episode's foot lateralFriction (PGE:209)
  EP_CMD_FREQ = 5,
  EP_COUNTER = 6,
  EP_PUSH_FORCE = 7,  // 3
  EP_NOISE = 10,      // 4: pos_x, pos_y, yaw, pos_z bias
  EP_LAST_DIFF = 14,
  EP_INIT_DIFF = 15,  // -1 = None (element 0)
  EP_TOTAL_SPD = 16,
  EP_MAX_SPD = 17,
  EP_REW = 18,        // 4: reward_vel, reward_rotation, reward_dist, reward_avg_spd
  EP_PUSH_COUNT = 22,
  EP_STEP_DRAWS = 23, // uniforms consumed by the steps of this episode (Philox counter)
  EP_INIT_ORN = 24,   // 4: the start orientation, rotated in place at every reset (PGE:186-190)
  EP_EPISODE = 28,
  EP_N_BOXES = 29,
  EP_N_STATICS = 30,
};

struct EpmcParams {
  int32_t element_id, max_steps, push_enabled, push_count0;
  int32_t push_interval_step, push_duration_step, cmd_freq_lo, cmd_freq_hi;
  float friction_lo, friction_hi, hforce_lo, hforce_hi;
  float vforce_lo, vforce_hi, push_ratio, plane_friction;
  float spd_lo, spd_hi, aux_radius, hole_gap_lo;      // aux_radius < 0: no auxiliary cylinders
  float hole_gap_hi, box_friction;                    // lateralFriction of a body made by createMultiBody: Bullet's default 0.5
  int32_t terrain_contacts, split_rays;               // terrain_contacts 0: the boxes are seen by the rays only.  split_rays: the 778 rays of the observation are cast by a kernel of
                                                      // their own behind the step kernel (round 6, percept_rays below): observe() leaves the row's ray pose in ray_pose instead of casting
  int32_t noise_on[4];
  float noise_lo[4], noise_hi[4];
  const float* init_state;  // [37] LeggedRobot.get_init_states_info(), LR:116-117
  // per-env buffers
  float* ep;               // [n_envs][EPMC_EP_STRIDE]
  float* info;             // [n_envs][6]
  float* statics;          // [n_envs][EPMC_MAX_STATICS][8]
  float* boxes;            // [n_envs][EPMC_MAX_BOXES][8]  x0 x1 y0 y1 z0 z1 - - of what the rays (and later the contacts) see
  float* push_trace;       // [n_envs][n_sub][4]
  float* ray_trace;        // optional [n_envs][778][8]: from 3, to 3, hit, fraction
  float* ray_pose;         // [n_envs][EPMC_RAY_POSE]: what percept_rays needs of a row (position with its noise, rotation, yaw, height noise, box count)
  // parity hooks (null in production)
  const float* scr_state;    // [n_envs][37]
  const uint8_t* scr_ray_hit;  // [n_envs][778]
  const float* scr_ray_frac;   // [n_envs][778]
  const float* scr_draws;    // [n_envs][scr_n_draws]
  int32_t scr_n_draws, pad4;
};

// Uniform draws in the reference's call order: a recorded stream (parity) or Philox keyed on (seed; env, episode, index, salt).
struct EpmcDraws {
  const float* scr;
  int n_scr, used;
  uint64_t seed;
  uint32_t env, episode, salt;
  uint32_t blk[4];          // the four words of Philox block used / 4 (one evaluation serves four consecutive draws)
  int have = -1;            // index of the block held in blk, -1: none
  LL_HD float u01() {
    const int i = used++;
    if (scr) return i < n_scr ? scr[i] : 0.5f;
    if ((i >> 2) != have) {
      have = i >> 2;
      philox4x32(env, episode, (uint32_t)have, salt, (uint32_t)seed, (uint32_t)(seed >> 32), blk);
    }
    const int k = i & 3;
    const uint32_t w = k == 0 ? blk[0] : (k == 1 ? blk[1] : (k == 2 ? blk[2] : blk[3]));
    return (float)(w >> 8) * (1.0f / 16777216.0f);
  }
  LL_HD float uniform(float a, float b) { return a + (b - a) * u01(); }
  LL_HD int randint(int a, int b) {                      // np.random.randint(a, b): a .. b-1
    int k = (int)((float)(b - a) * u01());
    if (k > b - a - 1) k = b - a - 1;
    return a + k;
  }
};

template <class L>
struct Epmc {
  typedef Pmc<L> K;
  typedef typename L::F F;
  typedef typename K::Base Base;

  // ------------------------------------------------------------------------------------------------------------
  // terrain (BSE:22-503): rows [kind, x, y, z, a, b, c, 0] in creation order + the compact box list the rays use
  // ------------------------------------------------------------------------------------------------------------
  struct Terrain {
    float* rows;
    float* boxes;
    int n_rows, n_boxes;
    bool store;
    float gap, aux;
    LL_HD void row(float kind, float x, float y, float z, float a, float b, float c) {
      if (store && n_rows < EPMC_MAX_STATICS) {
        float* r = rows + n_rows * 8;
        r[0] = kind; r[1] = x; r[2] = y; r[3] = z; r[4] = a; r[5] = b; r[6] = c; r[7] = 0.0f;
      }
      n_rows++;
    }
    LL_HD void box(float x, float y, float z, float l, float w, float h, float flag) {
      row(0.0f, x, y, z, l * 0.5f, w * 0.5f, h * 0.5f);
      if (l > 0.0f && n_boxes < EPMC_MAX_BOXES) {
        if (store) {                                                            // record: x0 x1 y0 y1 | z0 z1 rod r
          float* b = boxes + n_boxes * EPMC_BOX_WORDS;
          b[0] = x - l * 0.5f; b[1] = x + l * 0.5f; b[2] = y - w * 0.5f; b[3] = y + w * 0.5f; b[4] = z - h * 0.5f; b[5] = z + h * 0.5f;
          b[6] = (flag != 0.0f && aux >= 0.0f) ? flag : 0.0f; b[7] = aux >= 0.0f ? aux : 0.0f;      // edge rods, for the contacts
        }
        n_boxes++;
      }
      if (flag != 0.0f && aux >= 0.0f) {                                       // BSE:43-104 edge cylinders along y
        row(1.0f, x - l * 0.5f, y, z + h * 0.5f * flag, aux, w, 0.0f);
        row(1.0f, x + l * 0.5f, y, z + h * 0.5f * flag, aux, w, 0.0f);
      }
    }
  };
  static LL_HD float hurdle(Terrain& T, EpmcDraws& d, float cur) {              // BSE:309-364
    float height = d.uniform(0.05f, 0.15f);
    float distance = d.uniform(1.0f, 3.0f);
    T.box(cur + distance * 0.5f, 0.0f, height * 0.5f, 0.1f, T.gap, height, 1.0f);
    return cur + distance + 0.1f;
  }
  static LL_HD float hole(Terrain& T, EpmcDraws& d, const EpmcParams& E, float cur) {   // BSE:366-425
    float distance = d.uniform(1.0f, 3.0f);
    float gap_height = d.uniform(E.hole_gap_lo, E.hole_gap_hi);
    T.box(cur + distance * 0.5f, 0.0f, 0.15f + gap_height, 0.1f, T.gap, 0.3f, -1.0f);
    return cur + distance + 0.1f;
  }
  static LL_HD float cube_set(Terrain& T, EpmcDraws& d, float cur) {            // BSE:427-503, easy=True (BSE:247)
    cur += d.uniform(0.0f, 1.0f);
    T.box(1.75f + cur, 0.0f, 0.125f, 0.5f, T.gap, 0.25f, 1.0f);
    T.box(1.0f + cur, 0.0f, 0.05f, 0.5f, T.gap, 0.1f, 1.0f);
    cur += 2.0f;
    T.box(cur + 0.5f, 0.0f, 0.125f, 0.5f, T.gap, 0.25f, 1.0f);
    T.box(cur + 1.25f, 0.0f, 0.05f, 0.5f, T.gap, 0.1f, 1.0f);
    return cur + 3.0f;
  }
  static LL_HD void gen_terrain(Terrain& T, EpmcDraws& d, const EpmcParams& E, float* target) {
    target[0] = 8.0f; target[1] = 0.0f; target[2] = 0.0f;
    if (E.element_id == 0) {                                                   // joystick: the target marker only
      T.box(8.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
      return;
    }
    float width = d.uniform(0.02f, 0.5f);                                       // PGE:165, BSE:166-170
    float gap = d.uniform(1.0f, 20.0f);                                         // PGE:166
    T.gap = gap;
    float aux = T.aux;
    T.aux = -1.0f;                                                              // the walls carry no edge cylinders
    T.box(5.0f, gap * 0.5f + width * 0.5f, 1.0f, 200.0f, width, 2.0f, 0.0f);
    T.box(5.0f, -(gap * 0.5f + width * 0.5f), 1.0f, 200.0f, width, 2.0f, 0.0f);
    T.aux = aux;
    float cur = 0.0f;
    if (E.element_id == 1 || E.element_id == 2) {                              // BSE:200-222
      int n = d.randint(1, 10);
      for (int i = 0; i < n; i++) cur = (E.element_id == 1) ? hurdle(T, d, cur) : hole(T, d, E, cur);
      target[0] = cur + d.uniform(-1.0f, 1.0f);
      T.box(target[0], 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
      for (int i = 0; i < n; i++) cur = (E.element_id == 1) ? hurdle(T, d, cur) : hole(T, d, E, cur);
    } else {                                                                    // BSE:187-198 cubes
      int n = d.randint(1, 5);
      for (int i = 0; i < n; i++) cur = cube_set(T, d, cur);
      target[0] = cur + d.uniform(-3.0f, 3.0f);
      T.box(target[0], 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
      for (int i = 0; i < n; i++) cur = cube_set(T, d, cur);
    }
  }

  // ------------------------------------------------------------------------------------------------------------
  // rays (PGE:25-52, :381-447).  cast(): this build's spec of rayTestBatch(mask 6): plane z = 0 and the boxes.
  // ------------------------------------------------------------------------------------------------------------
  // entry parameter of segment f + s d (0 <= s <= 1) into the box lo/hi record x0 x1 y0 y1 z0 z1, or 3e38 (an origin inside: no hit)
  static LL_HD float slab(const float* f, const float* d, const float* inv, const float* bx) {
    float te = -3.0e38f, tl = 3.0e38f;
    for (int a = 0; a < 3; a++) {
      const float lo = bx[2 * a], hi = bx[2 * a + 1];
      if (d[a] == 0.0f) {
        if (!(f[a] >= lo && f[a] <= hi)) { te = 3.0e38f; tl = -3.0e38f; }
      } else {
        float t1 = (lo - f[a]) * inv[a], t2 = (hi - f[a]) * inv[a];
        te = fmaxf(te, fminf(t1, t2));
        tl = fminf(tl, fmaxf(t1, t2));
      }
    }
    return (te <= tl && te >= 0.0f && te <= 1.0f) ? te : 3.0e38f;
  }
  // `cand` = bit b set for every box the segment may meet (a conservative pre-selection made once per env and ray family)
  static LL_HD float cast(const float* f, const float* t, const float* boxes, unsigned long long cand, bool* hit_out) {
    const float d[3] = {t[0] - f[0], t[1] - f[1], t[2] - f[2]};
    float inv[3];
    for (int a = 0; a < 3; a++) inv[a] = d[a] != 0.0f ? 1.0f / d[a] : 0.0f;
    float best = 3.0e38f;
    if (d[2] < 0.0f) {
      float tz = -f[2] * inv[2];
      if (tz >= 0.0f && tz <= 1.0f) best = tz;
    }
    // a segment can only meet a box whose x and y extents overlap its own: most boxes are thin bars far from the ray
    const float sx0 = fminf(f[0], t[0]), sx1 = fmaxf(f[0], t[0]), sy0 = fminf(f[1], t[1]), sy1 = fmaxf(f[1], t[1]);
    while (cand) {
      const int b = __builtin_ctzll(cand);
      cand &= cand - 1;
      const float* bx = boxes + b * EPMC_BOX_WORDS;                           // one 16-byte read decides most boxes
      if (bx[1] < sx0 || bx[0] > sx1 || bx[3] < sy0 || bx[2] > sy1) continue;
      best = fminf(best, slab(f, d, inv, bx));
    }
    *hit_out = best < 2.0f;
    return best < 2.0f ? best : 1.0f;
  }
  static LL_HD void grid_point(int i, float x0, float x1, float y0, float y1, float* gx, float* gy) {   // utils/constants.py:5-10, 25 x 13
    const int ix = i / 13, iy = i - 13 * ix;
    *gx = x0 + (x1 - x0) * (float)ix * (1.0f / 24.0f);
    *gy = y0 + (y1 - y0) * (float)iy * (1.0f / 12.0f);
  }
  // end points of ray r in call order: 325 height, 128 horizontal, 325 front
  static LL_HD void ray_ends(int r, const float* pos, const M3<float>& R, float yaw, float* f, float* t) {
    if (r < EPMC_N_HEIGHT) {                                                    // PGE:431-441
      float gx, gy;
      grid_point(r, -1.2f, 1.2f, -0.6f, 0.6f, &gx, &gy);
      f[0] = t[0] = R.m[0] * gx + R.m[1] * gy + pos[0];
      f[1] = t[1] = R.m[3] * gx + R.m[4] * gy + pos[1];
      f[2] = 10.0f; t[2] = -10.0f;
    } else if (r < EPMC_N_HEIGHT + EPMC_N_HORIZ) {                              // PGE:30-38
      const float a = yaw + 6.283185307179586f * (float)(r - EPMC_N_HEIGHT) * (1.0f / 128.0f);
      f[0] = pos[0]; f[1] = pos[1]; f[2] = pos[2];
      t[0] = pos[0] + 20.0f * cosf(a); t[1] = pos[1] + 20.0f * sinf(a); t[2] = pos[2];
    } else {                                                                    // PGE:405-419: (0, gy, gz) -> (3, gy, gz) in the base frame
      float gy, gz;
      grid_point(r - EPMC_N_HEIGHT - EPMC_N_HORIZ, -0.25f, 0.25f, -0.3f, 0.1f, &gy, &gz);
      for (int a = 0; a < 3; a++) {
        const float o = R.m[3 * a + 1] * gy + R.m[3 * a + 2] * gz + pos[a];
        f[a] = o; t[a] = o + 3.0f * R.m[3 * a];
      }
    }
  }
  // sin / cos of a moderate angle (|x| < ~1e3): the scalar twin of Pmc::sincos_joint (Cody-Waite to [-pi/4, pi/4] + minimax polynomials)
  static LL_HD void sincos_f(float x, float* s, float* c) {
    const float kf = rintf(x * 0.63661977236758134f);
    float r = x - kf * 1.5707962512969971f;
    r = r - kf * 7.5497894158615964e-08f;
    const float r2 = r * r;
    const float sp = r + r * r2 * (-0.16666654611f + r2 * (0.0083321608736f + r2 * (-0.00019515295891f)));
    const float cp = 1.0f + r2 * (-0.5f + r2 * (0.041666645683f + r2 * (-0.0013887316255f + r2 * 0.000024433157117f)));
    const int k = (int)kf;
    const float ss = (k & 1) ? cp : sp, cs = (k & 1) ? sp : cp;
    *s = (k & 2) ? -ss : ss;
    *c = ((k + 1) & 2) ? -cs : cs;
  }
  // (box records are read with load_box, pmc_math.hpp: a = x0 x1 y0 y1, c = z0 z1 - -, one box ahead of their use so that the read
  // latency hides behind the previous box's arithmetic -- a wave has nothing else to switch to)
  // one axis of the slab test of origin o, direction d (inv = 1/d, anything if d == 0) against [lo, hi]: narrows [te, tl]
  static LL_HD void slab_axis(float lo, float hi, float o, float d, float inv, float& te, float& tl) {
    const float t1 = (lo - o) * inv, t2 = (hi - o) * inv;
    const bool par = d == 0.0f, inside = (o >= lo) & (o <= hi);
    const float a = par ? (inside ? -3.0e38f : 3.0e38f) : fminf(t1, t2);
    const float b = par ? (inside ? 3.0e38f : -3.0e38f) : fmaxf(t1, t2);
    te = fmaxf(te, a);
    tl = fminf(tl, b);
  }
  // The three percep arrays straight into the obs row; lanes share the rays out.  `boxes` = the env's n_boxes records staged at
  // the start of the row scratch.  Each family is a dense, branch-light loop: lane 0 first compacts the boxes a family can meet (see
  // below) into three lists in the row scratch; a ray then walks its family's list with one 32-byte LDS read per box and no per-ray set-up beyond its origin
  // (the front rays share one direction, the fan one origin; the fan's directions come from one sin/cos per lane and eight exact
  // 45-degree turns).
  static LL_HD void observe_rays(const L& ln, const StepParams& P, const EpmcParams& E, int env, const float* pos, const M3<float>& R, float yaw,
                                 const float* noise, const float* boxes, int n_boxes, float* percep) {
    if (PMC_ABL(32)) return;                                                     // ablation build only: no rays
    if (E.scr_ray_hit) {                                                         // parity hook: the caller plays rayTestBatch
      for (int r = ln.ray_first(); r < EPMC_N_RAYS; r += ln.ray_stride()) {
        float f[3], t[3];
        ray_ends(r, pos, R, yaw, f, t);
        const bool hit = E.scr_ray_hit[(long)env * EPMC_N_RAYS + r] != 0;
        const float frac = E.scr_ray_frac[(long)env * EPMC_N_RAYS + r];
        emit_ray(E, env, r, f, t, hit, frac, noise, percep);
      }
      return;
    }
    // --- the three compact lists (lane 0 writes, the row reads after the sync).  A box is listed for a family iff its extents overlap the
    //     bounding box of that family's ray bundle (grown by 1 mm against rounding): the 2.4 m x 1.2 m rectangle of the height grid as it lies
    //     in the world, the 3 m long prism swept by the front rays, and for the fan -- level rays of 20 m -- the boxes whose height range
    //     contains the base height.  Exact: a ray only ever meets a box inside its bundle's bounds.
    float* listA = ln.row_scratch() + EPMC_LIST_A;
    float* listB = ln.row_scratch() + EPMC_LIST_B;
    float* listC = ln.row_scratch() + EPMC_LIST_C;
    const float slack = 1.0e-3f;
    const float hgx = fabsf(R.m[0]) * 1.2f + fabsf(R.m[1]) * 0.6f + slack, hgy = fabsf(R.m[3]) * 1.2f + fabsf(R.m[4]) * 0.6f + slack;
    float flo[3], fhi[3];
    for (int a = 0; a < 3; a++) {
      const float ry = fabsf(R.m[3 * a + 1]) * 0.25f, z0 = R.m[3 * a + 2] * -0.3f, z1 = R.m[3 * a + 2] * 0.1f, da = 3.0f * R.m[3 * a];
      flo[a] = pos[a] - ry + fminf(z0, z1) + fminf(da, 0.0f) - slack;
      fhi[a] = pos[a] + ry + fmaxf(z0, z1) + fmaxf(da, 0.0f) + slack;
    }
    // (round 4: the lanes of the row take 16 boxes at a time -- lane j tests box b0 + j against the three bounds, a ballot over the row counts
    // and places the hits -- instead of every lane walking all boxes behind lane 0's stores: the lists come out in the same order)
    int nA = 0, nB = 0, nC = 0;
    const int me = ln.ray_first(), W = ln.ray_stride();
    const uint32_t below = (1u << me) - 1u;
    for (int b0 = 0; b0 < n_boxes; b0 += W) {
      const int b = b0 + me;
      const bool live = b < n_boxes;
      const float* bx = boxes + (live ? b : 0) * EPMC_BOX_WORDS;
      const BoxRec r = load_box(bx);
      const bool grid = live & (r.a.y >= pos[0] - hgx) & (r.a.x <= pos[0] + hgx) & (r.a.w >= pos[1] - hgy) & (r.a.z <= pos[1] + hgy);
      const bool front = live & (r.a.y >= flo[0]) & (r.a.x <= fhi[0]) & (r.a.w >= flo[1]) & (r.a.z <= fhi[1]) & (r.c.y >= flo[2]) & (r.c.x <= fhi[2]);
      const bool fan = live & (r.a.y >= pos[0] - 20.1f) & (r.a.x <= pos[0] + 20.1f) & (r.a.w >= pos[1] - 20.1f) & (r.a.z <= pos[1] + 20.1f) & (r.c.x <= pos[2]) & (r.c.y >= pos[2]);
      const uint32_t mA = ln.row_ballot(grid), mB = ln.row_ballot(fan), mC = ln.row_ballot(front);
      const int pA = nA + __builtin_popcount(mA & below), pB = nB + __builtin_popcount(mB & below), pC = nC + __builtin_popcount(mC & below);
      if (grid && pA < EPMC_LIST_A_MAX) store_box(listA + pA * EPMC_BOX_WORDS, r);
      if (fan && pB < EPMC_LIST_B_MAX) store_box(listB + pB * EPMC_BOX_WORDS, r);
      if (front && pC < EPMC_LIST_C_MAX) store_box(listC + pC * EPMC_BOX_WORDS, r);
      nA += __builtin_popcount(mA); nB += __builtin_popcount(mB); nC += __builtin_popcount(mC);
    }
    ln.row_sync();
    const float* LA = nA <= EPMC_LIST_A_MAX ? listA : boxes;                     // a list that does not fit: walk all boxes (the tests are exact anyway)
    const float* LB = nB <= EPMC_LIST_B_MAX ? listB : boxes;
    const float* LC = nC <= EPMC_LIST_C_MAX ? listC : boxes;
    const int cA = nA <= EPMC_LIST_A_MAX ? nA : n_boxes, cB = nB <= EPMC_LIST_B_MAX ? nB : n_boxes, cC = nC <= EPMC_LIST_C_MAX ? nC : n_boxes;
    // --- height grid (PGE:431-447): straight down from z = 10 to -10: the highest top under (x, y), or the plane ---
    // (round 4: RAY_CHUNK rays of a lane at a time with the list's records read ONCE per chunk -- box outermost, the rays' running answers in
    // registers -- instead of one LDS round trip per (ray, box): a lone wave has nothing to hide that latency behind.  Same arithmetic per ray.)
    if (!PMC_ABL(128))
    for (int r0 = ln.ray_first(); r0 < EPMC_N_HEIGHT; r0 += RAY_CHUNK * ln.ray_stride()) {
      float x[RAY_CHUNK], y[RAY_CHUNK], top[RAY_CHUNK];
      LL_UNROLL
      for (int j = 0; j < RAY_CHUNK; j++) {
        const int r = r0 + j * ln.ray_stride();
        float gx, gy;
        grid_point(r < EPMC_N_HEIGHT ? r : 0, -1.2f, 1.2f, -0.6f, 0.6f, &gx, &gy);
        x[j] = R.m[0] * gx + R.m[1] * gy + pos[0]; y[j] = R.m[3] * gx + R.m[4] * gy + pos[1];
        top[j] = 0.0f;
      }
      BoxRec nx = load_box(LA);
      for (int b = 0; b < cA; b++) {
        const BoxRec bx = nx;
        nx = load_box(LA + (b + 1) * EPMC_BOX_WORDS);                             // (one record past the list is readable scratch)
        LL_UNROLL
        for (int j = 0; j < RAY_CHUNK; j++) {
          const bool in = (x[j] >= bx.a.x) & (x[j] <= bx.a.y) & (y[j] >= bx.a.z) & (y[j] <= bx.a.w);
          top[j] = in ? fmaxf(top[j], bx.c.y) : top[j];
        }
      }
      LL_UNROLL
      for (int j = 0; j < RAY_CHUNK; j++) {
        const int r = r0 + j * ln.ray_stride();
        if (r >= EPMC_N_HEIGHT) break;
        const float frac = (10.0f - top[j]) * 0.05f;
        float v = 10.0f + frac * -20.0f;                                          // PGE:442 hit height
        if (E.noise_on[3]) v = (v > 0.01f && v < 0.6f) ? v + noise[3] : 0.0f;      // PGE:443-446
        percep[r] = v;
        if (E.ray_trace) {
          float* tr = E.ray_trace + ((long)env * EPMC_N_RAYS + r) * 8;
          tr[0] = x[j]; tr[1] = y[j]; tr[2] = 10.0f; tr[3] = x[j]; tr[4] = y[j]; tr[5] = -10.0f; tr[6] = 1.0f; tr[7] = frac;
        }
      }
    }
    if (PMC_ABL(64)) return;                                                     // ablation: height rays only
    // --- horizontal fan (PGE:30-38, :399): 128 level rays of 20 m from the base position; a miss measures |(0,0,0) - origin| ---
    {
      float sy, cy;
      sincos_f(yaw, &sy, &cy);
      const float miss = sqrtf(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
      for (int k = ln.ray_first(); k < EPMC_N_HORIZ; k += ln.ray_stride()) {
        float sk, ck;
        sincos_f(6.283185307179586f * (float)k * (1.0f / 128.0f), &sk, &ck);
        const float dx = 20.0f * (cy * ck - sy * sk), dy = 20.0f * (sy * ck + cy * sk);
        const float ix = dx != 0.0f ? 1.0f / dx : 0.0f, iy = dy != 0.0f ? 1.0f / dy : 0.0f;
        float best = 3.0e38f;
        BoxRec nx = load_box(LB);
        for (int b = 0; b < cB; b++) {
          const BoxRec bx = nx;
          nx = load_box(LB + (b + 1) * EPMC_BOX_WORDS);
          float te = -3.0e38f, tl = 3.0e38f;
          slab_axis(bx.a.x, bx.a.y, pos[0], dx, ix, te, tl);
          slab_axis(bx.a.z, bx.a.w, pos[1], dy, iy, te, tl);
          const bool ok = (pos[2] >= bx.c.x) & (pos[2] <= bx.c.y) & (te <= tl) & (te >= 0.0f) & (te <= 1.0f);
          best = ok ? fminf(best, te) : best;
        }
        const bool hit = best < 2.0f;
        const int r = EPMC_N_HEIGHT + k;
        percep[r] = hit ? best * 20.0f : miss;
        if (E.ray_trace) {
          float* tr = E.ray_trace + ((long)env * EPMC_N_RAYS + r) * 8;
          tr[0] = pos[0]; tr[1] = pos[1]; tr[2] = pos[2]; tr[3] = pos[0] + dx; tr[4] = pos[1] + dy; tr[5] = pos[2]; tr[6] = hit ? 1.0f : 0.0f; tr[7] = hit ? best : 1.0f;
        }
      }
    }
    if (PMC_ABL(128)) return;                                                    // ablation: height rays and the fan only
    // --- front rays (PGE:405-427): from (0, gy, gz) to (3, gy, gz) in the base frame; a miss counts as the far end ---
    {
      const float d[3] = {3.0f * R.m[0], 3.0f * R.m[3], 3.0f * R.m[6]};
      float inv[3];
      for (int a = 0; a < 3; a++) inv[a] = d[a] != 0.0f ? 1.0f / d[a] : 0.0f;
      for (int i0 = ln.ray_first(); i0 < EPMC_N_FRONT; i0 += RAY_CHUNK * ln.ray_stride()) {
        float o[RAY_CHUNK][3], best[RAY_CHUNK];
        LL_UNROLL
        for (int j = 0; j < RAY_CHUNK; j++) {
          const int i = i0 + j * ln.ray_stride();
          float gy, gz;
          grid_point(i < EPMC_N_FRONT ? i : 0, -0.25f, 0.25f, -0.3f, 0.1f, &gy, &gz);
          LL_UNROLL
          for (int a = 0; a < 3; a++) o[j][a] = R.m[3 * a + 1] * gy + R.m[3 * a + 2] * gz + pos[a];
          best[j] = 3.0e38f;
          if (d[2] < 0.0f) {
            const float tz = -o[j][2] * inv[2];
            if (tz >= 0.0f && tz <= 1.0f) best[j] = tz;
          }
        }
        BoxRec nx = load_box(LC);
        for (int b = 0; b < cC; b++) {
          const BoxRec bx = nx;
          nx = load_box(LC + (b + 1) * EPMC_BOX_WORDS);
          LL_UNROLL
          for (int j = 0; j < RAY_CHUNK; j++) {
            float te = -3.0e38f, tl = 3.0e38f;
            slab_axis(bx.a.x, bx.a.y, o[j][0], d[0], inv[0], te, tl);
            slab_axis(bx.a.z, bx.a.w, o[j][1], d[1], inv[1], te, tl);
            slab_axis(bx.c.x, bx.c.y, o[j][2], d[2], inv[2], te, tl);
            const bool ok = (te <= tl) & (te >= 0.0f) & (te <= 1.0f);
            best[j] = ok ? fminf(best[j], te) : best[j];
          }
        }
        LL_UNROLL
        for (int j = 0; j < RAY_CHUNK; j++) {
          const int i = i0 + j * ln.ray_stride();
          if (i >= EPMC_N_FRONT) break;
          const bool hit = best[j] < 2.0f;
          const int r = EPMC_N_HEIGHT + EPMC_N_HORIZ + i;
          percep[r] = hit ? best[j] * 3.0f : 3.0f;
          if (E.ray_trace) {
            float* tr = E.ray_trace + ((long)env * EPMC_N_RAYS + r) * 8;
            LL_UNROLL
            for (int a = 0; a < 3; a++) { tr[a] = o[j][a]; tr[3 + a] = o[j][a] + d[a]; }
            tr[6] = hit ? 1.0f : 0.0f; tr[7] = hit ? best[j] : 1.0f;
          }
        }
      }
    }
  }
  // ---- the rays as a kernel of their own (round 6; llenv.hip epmc_percept_kernel) ----------------------------------------------------------------------------
  // Inside the one-wave-per-SIMD step kernel the 3.2 M slab tests of a step run at the lone wave's pace (one instruction per ~5 cycles, nothing to hide a read behind:
  // 0.043 of the hurdle step's 0.289 ms, profiles/r04_epmc_ray_ablation.txt).  Cast by a second kernel -- a workgroup of four waves per row, sixteen waves per SIMD -- they
  // are full-rate work.  observe() leaves the row's ray pose (leave_ray_pose), percept_rays() reads it back and writes the three percep arrays of the row: ray r of the row is
  // taken by worker r_first + k r_stride.  Per ray the arithmetic is observe_rays' own, expression for expression (same helpers); a ray's answer is a minimum / maximum over
  // boxes, so neither the order of the boxes nor the compact lists of observe_rays (exact pre-selections) change a bit of it: tests hold the two paths equal.
  // last_box: the record the rays must see in place of the row's LAST box, or null (SEPMC: the flag where it stood while the step ran -- the step kernel moves the flag's box
  // behind the observation, CTG:515-579, and the ray kernel runs behind the step kernel)
  static LL_HD void leave_ray_pose(const L& ln, const EpmcParams& E, int row, const float* pos, const M3<float>& R, float yaw, const float* noise, int n_boxes, const float* last_box = nullptr) {
    if (!ln.lane0()) return;
    float* rec = E.ray_pose + (long)row * EPMC_RAY_POSE;
    rec[0] = pos[0]; rec[1] = pos[1]; rec[2] = pos[2];
    for (int i = 0; i < 9; i++) rec[3 + i] = R.m[i];
    rec[12] = yaw; rec[13] = noise[3]; rec[14] = (float)n_boxes; rec[15] = last_box ? 1.0f : 0.0f;
    for (int i = 0; i < EPMC_BOX_WORDS; i++) rec[16 + i] = last_box ? last_box[i] : 0.0f;
  }
  // box b of the row as the rays of this observation must see it
  static LL_HD BoxRec ray_box(const float* rec, const float* row_boxes, int b) {
    return load_box((rec[15] != 0.0f && b == (int)rec[14] - 1) ? rec + 16 : row_boxes + b * EPMC_BOX_WORDS);
  }
  // which of the row's boxes a ray family can meet at all: observe_rays' three bounds tests (family 0 height grid, 1 fan, 2 front rays)
  struct RayBounds { float hgx, hgy, flo[3], fhi[3]; };
  static LL_HD RayBounds ray_bounds(const float* pos, const M3<float>& R) {
    RayBounds b;
    const float slack = 1.0e-3f;
    b.hgx = fabsf(R.m[0]) * 1.2f + fabsf(R.m[1]) * 0.6f + slack; b.hgy = fabsf(R.m[3]) * 1.2f + fabsf(R.m[4]) * 0.6f + slack;
    for (int a = 0; a < 3; a++) {
      const float ry = fabsf(R.m[3 * a + 1]) * 0.25f, z0 = R.m[3 * a + 2] * -0.3f, z1 = R.m[3 * a + 2] * 0.1f, da = 3.0f * R.m[3 * a];
      b.flo[a] = pos[a] - ry + fminf(z0, z1) + fminf(da, 0.0f) - slack;
      b.fhi[a] = pos[a] + ry + fmaxf(z0, z1) + fmaxf(da, 0.0f) + slack;
    }
    return b;
  }
  static LL_HD bool box_in_family(int fam, const BoxRec& r, const float* pos, const RayBounds& b) {
    if (fam == 0) return (r.a.y >= pos[0] - b.hgx) & (r.a.x <= pos[0] + b.hgx) & (r.a.w >= pos[1] - b.hgy) & (r.a.z <= pos[1] + b.hgy);
    if (fam == 1) return (r.a.y >= pos[0] - 20.1f) & (r.a.x <= pos[0] + 20.1f) & (r.a.w >= pos[1] - 20.1f) & (r.a.z <= pos[1] + 20.1f) & (r.c.x <= pos[2]) & (r.c.y >= pos[2]);
    return (r.a.y >= b.flo[0]) & (r.a.x <= b.fhi[0]) & (r.a.w >= b.flo[1]) & (r.a.z <= b.fhi[1]) & (r.c.y >= b.flo[2]) & (r.c.x <= b.fhi[2]);
  }
  // lists[fam]: box records a family can meet (n[fam] of them), in any order; percep: the row's 778 floats in the obs row
  static LL_HD void percept_rays(const EpmcParams& E, int row, const float* rec, const float* const* lists, const int* n, float* percep, int r_first, int r_stride) {
    const float pos[3] = {rec[0], rec[1], rec[2]};
    M3<float> R;
    for (int i = 0; i < 9; i++) R.m[i] = rec[3 + i];
    const float yaw = rec[12], noise_z = rec[13];
    float sy, cy;
    sincos_f(yaw, &sy, &cy);
    const float miss = sqrtf(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
    const float d[3] = {3.0f * R.m[0], 3.0f * R.m[3], 3.0f * R.m[6]};
    float inv[3];
    for (int a = 0; a < 3; a++) inv[a] = d[a] != 0.0f ? 1.0f / d[a] : 0.0f;
    // family by family (every worker walks the same loop at the same time: no divergence between ray kinds inside a wavefront)
    for (int r = r_first; r < EPMC_N_HEIGHT; r += r_stride) {                    // observe_rays: height grid
      float* tr = E.ray_trace ? E.ray_trace + ((long)row * EPMC_N_RAYS + r) * 8 : nullptr;
      float gx, gy;
      grid_point(r, -1.2f, 1.2f, -0.6f, 0.6f, &gx, &gy);
      const float x = R.m[0] * gx + R.m[1] * gy + pos[0], y = R.m[3] * gx + R.m[4] * gy + pos[1];
      float top = 0.0f;
      for (int b = 0; b < n[0]; b++) {
        const BoxRec bx = load_box(lists[0] + b * EPMC_BOX_WORDS);
        const bool in = (x >= bx.a.x) & (x <= bx.a.y) & (y >= bx.a.z) & (y <= bx.a.w);
        top = in ? fmaxf(top, bx.c.y) : top;
      }
      const float frac = (10.0f - top) * 0.05f;
      float v = 10.0f + frac * -20.0f;
      if (E.noise_on[3]) v = (v > 0.01f && v < 0.6f) ? v + noise_z : 0.0f;
      percep[r] = v;
      if (tr) { tr[0] = x; tr[1] = y; tr[2] = 10.0f; tr[3] = x; tr[4] = y; tr[5] = -10.0f; tr[6] = 1.0f; tr[7] = frac; }
    }
    for (int k = r_first; k < EPMC_N_HORIZ; k += r_stride) {                     // observe_rays: horizontal fan
      const int r = EPMC_N_HEIGHT + k;
      float* tr = E.ray_trace ? E.ray_trace + ((long)row * EPMC_N_RAYS + r) * 8 : nullptr;
      float sk, ck;
      sincos_f(6.283185307179586f * (float)k * (1.0f / 128.0f), &sk, &ck);
      const float dx = 20.0f * (cy * ck - sy * sk), dy = 20.0f * (sy * ck + cy * sk);
      const float ix = dx != 0.0f ? 1.0f / dx : 0.0f, iy = dy != 0.0f ? 1.0f / dy : 0.0f;
      float best = 3.0e38f;
      for (int b = 0; b < n[1]; b++) {
        const BoxRec bx = load_box(lists[1] + b * EPMC_BOX_WORDS);
        float te = -3.0e38f, tl = 3.0e38f;
        slab_axis(bx.a.x, bx.a.y, pos[0], dx, ix, te, tl);
        slab_axis(bx.a.z, bx.a.w, pos[1], dy, iy, te, tl);
        const bool ok = (pos[2] >= bx.c.x) & (pos[2] <= bx.c.y) & (te <= tl) & (te >= 0.0f) & (te <= 1.0f);
        best = ok ? fminf(best, te) : best;
      }
      const bool hit = best < 2.0f;
      percep[r] = hit ? best * 20.0f : miss;
      if (tr) { tr[0] = pos[0]; tr[1] = pos[1]; tr[2] = pos[2]; tr[3] = pos[0] + dx; tr[4] = pos[1] + dy; tr[5] = pos[2]; tr[6] = hit ? 1.0f : 0.0f; tr[7] = hit ? best : 1.0f; }
    }
    for (int i = r_first; i < EPMC_N_FRONT; i += r_stride) {                     // observe_rays: front rays
      const int r = EPMC_N_HEIGHT + EPMC_N_HORIZ + i;
      float* tr = E.ray_trace ? E.ray_trace + ((long)row * EPMC_N_RAYS + r) * 8 : nullptr;
      float gy, gz, o[3];
      grid_point(i, -0.25f, 0.25f, -0.3f, 0.1f, &gy, &gz);
      for (int a = 0; a < 3; a++) o[a] = R.m[3 * a + 1] * gy + R.m[3 * a + 2] * gz + pos[a];
      float best = 3.0e38f;
      if (d[2] < 0.0f) {
        const float tz = -o[2] * inv[2];
        if (tz >= 0.0f && tz <= 1.0f) best = tz;
      }
      for (int b = 0; b < n[2]; b++) {
        const BoxRec bx = load_box(lists[2] + b * EPMC_BOX_WORDS);
        float te = -3.0e38f, tl = 3.0e38f;
        slab_axis(bx.a.x, bx.a.y, o[0], d[0], inv[0], te, tl);
        slab_axis(bx.a.z, bx.a.w, o[1], d[1], inv[1], te, tl);
        slab_axis(bx.c.x, bx.c.y, o[2], d[2], inv[2], te, tl);
        const bool ok = (te <= tl) & (te >= 0.0f) & (te <= 1.0f);
        best = ok ? fminf(best, te) : best;
      }
      const bool hit = best < 2.0f;
      percep[r] = hit ? best * 3.0f : 3.0f;
      if (tr) {
        for (int a = 0; a < 3; a++) { tr[a] = o[a]; tr[3 + a] = o[a] + d[a]; }
        tr[6] = hit ? 1.0f : 0.0f; tr[7] = hit ? best : 1.0f;
      }
    }
  }
  // the host statement of the second kernel for ONE row (emul.cpp, and the reference for the GPU kernel's staging): lists built by walking the row's boxes in order
  static LL_HD void percept_row_host(const StepParams& P, const EpmcParams& E, int row) {
    const float* rec = E.ray_pose + (long)row * EPMC_RAY_POSE;
    const int nb = (int)rec[14];
    const float* boxes = E.boxes + (long)row * EPMC_MAX_BOXES * EPMC_BOX_WORDS;
    M3<float> R;
    for (int i = 0; i < 9; i++) R.m[i] = rec[3 + i];
    const RayBounds bd = ray_bounds(rec, R);
    float buf[3][EPMC_MAX_BOXES * EPMC_BOX_WORDS];
    int n[3] = {0, 0, 0};
    for (int b = 0; b < nb; b++) {
      const BoxRec r = ray_box(rec, boxes, b);
      for (int fam = 0; fam < 3; fam++)
        if (box_in_family(fam, r, rec, bd)) store_box(buf[fam] + (n[fam]++) * EPMC_BOX_WORDS, r);
    }
    const float* lists[3] = {buf[0], buf[1], buf[2]};
    percept_rays(E, row, rec, lists, n, P.obs + (long)row * P.obs_dim + 3L * P.prop_dim + 36, 0, 1);
  }

  // what the env makes of one ray's answer (PGE:395-447), for the scripted path
  static LL_HD void emit_ray(const EpmcParams& E, int env, int r, const float* f, const float* t, bool hit, float frac, const float* noise, float* percep) {
    if (E.ray_trace) {
      float* tr = E.ray_trace + ((long)env * EPMC_N_RAYS + r) * 8;
      tr[0] = f[0]; tr[1] = f[1]; tr[2] = f[2]; tr[3] = t[0]; tr[4] = t[1]; tr[5] = t[2]; tr[6] = hit ? 1.0f : 0.0f; tr[7] = frac;
    }
    const float hx = hit ? f[0] + frac * (t[0] - f[0]) : 0.0f, hy = hit ? f[1] + frac * (t[1] - f[1]) : 0.0f, hz = hit ? f[2] + frac * (t[2] - f[2]) : 0.0f;
    float v;
    if (r < EPMC_N_HEIGHT) {                                                  // PGE:442-446 hit height; a miss reports (0,0,0)
      v = hz;
      if (E.noise_on[3]) v = (v > 0.01f && v < 0.6f) ? v + noise[3] : 0.0f;
    } else if (r < EPMC_N_HEIGHT + EPMC_N_HORIZ) {                            // PGE:399, :49-50: a miss measures |(0,0,0) - origin|
      const float dx = hx - f[0], dy = hy - f[1], dz = hz - f[2];
      v = sqrtf(dx * dx + dy * dy + dz * dz);
    } else {                                                                  // PGE:425-427: a miss counts as the far end
      const float px = hit ? hx : t[0], py = hit ? hy : t[1], pz = hit ? hz : t[2];
      const float dx = px - f[0], dy = py - f[1], dz = pz - f[2];
      v = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    percep[r] = v;
  }

  // ------------------------------------------------------------------------------------------------------------
  // observation: prop | prop_a history (as PMC) | percep_2d | percep_1d | percep_front | target (PGE:276-297, :381-403)
  // ------------------------------------------------------------------------------------------------------------
  static LL_HD void observe(const L& ln, const StepParams& P, const EpmcParams& E, int env, float* row, bool fill, const typename K::ObsIn& hist,
                            const Base& bs, const F* q, const F* qd, const F* act, const float* ep, const float* target, float target_spd) {
    const Q4 qn = qnormalize(bs.q);
    const M3<float> R = qmat(qn);
    K::obs_emit_core(ln, P, row, fill, hist, bs, R, q, qd, act);
    float pos[3] = {bs.p.x, bs.p.y, bs.p.z};
    float yaw = atan2f(R.m[3], R.m[0]);                                          // PGE:386
    if (E.noise_on[0]) { pos[0] += ep[EP_NOISE + 0]; pos[1] += ep[EP_NOISE + 1]; }   // PGE:388-391
    if (E.noise_on[2]) yaw += ep[EP_NOISE + 2];                                  // PGE:392-393
    const long a0 = 3L * P.prop_dim + 36;
    const int n_boxes = (int)ep[EP_N_BOXES];
    if (LL_NO_FUSED_RAYS || (E.split_rays && !E.scr_ray_hit)) {
      leave_ray_pose(ln, E, env, pos, R, yaw, ep + EP_NOISE, n_boxes);            // the rays of this observation are cast by the kernel behind this one
    } else {
      const float* boxes = ln.stage_row(E.boxes + (long)env * EPMC_MAX_BOXES * EPMC_BOX_WORDS, n_boxes * EPMC_BOX_WORDS);   // LDS on the GPU
      observe_rays(ln, P, E, env, pos, R, yaw, ep + EP_NOISE, boxes, n_boxes, row + a0);
    }
    // target_info (PGE:400-403): the (x, y) of R^-1 (target - position), normalised, then the commanded speed
    const float dx = target[0] - pos[0], dy = target[1] - pos[1], dz = target[2] - pos[2];
    const float lx = R.m[0] * dx + R.m[3] * dy + R.m[6] * dz, ly = R.m[1] * dx + R.m[4] * dy + R.m[7] * dz;
    const float inv = 1.0f / sqrtf(lx * lx + ly * ly);
    if (ln.lane0()) {
      row[a0 + EPMC_N_RAYS + 0] = lx * inv;
      row[a0 + EPMC_N_RAYS + 1] = ly * inv;
      row[a0 + EPMC_N_RAYS + 2] = target_spd;
    }
  }

  // ------------------------------------------------------------------------------------------------------------
  // reset (PGE:196-249): everything except the observation; the caller runs observe() afterwards
  // ------------------------------------------------------------------------------------------------------------
  static LL_HD void reset_scalars(const L& ln, const StepParams& P, const EpmcParams& E, int env, float* ep, EpmcDraws& d, Base& bs, F* q, F* qd,
                                  const float* prev_orn) {
    ep[EP_FRICTION] = d.uniform(E.friction_lo, E.friction_hi);                   // PGE:209
    if (E.push_enabled) {                                                        // PGE:213-214, PR:52-54
      ep[EP_PUSH_COUNT] = (float)E.push_count0;
      randomize_force(E, d, ep);
    }
    Terrain T;
    T.rows = E.statics + (long)env * EPMC_MAX_STATICS * 8;
    T.boxes = E.boxes + (long)env * EPMC_MAX_BOXES * EPMC_BOX_WORDS;
    T.n_rows = T.n_boxes = 0; T.store = ln.lane0(); T.gap = 0.0f; T.aux = E.aux_radius;
    gen_terrain(T, d, E, ep + EP_TARGET);                                         // PGE:216-221
    ln.row_sync();                                                                // the rays of this step read the new boxes
    ep[EP_N_BOXES] = (float)T.n_boxes; ep[EP_N_STATICS] = (float)(T.n_rows < EPMC_MAX_STATICS ? T.n_rows : EPMC_MAX_STATICS);
    ep[EP_CMD_FREQ] = (float)d.randint(E.cmd_freq_lo, E.cmd_freq_hi);            // PGE:223
    ep[EP_COUNTER] = 0.0f; ep[EP_TOTAL_SPD] = 0.0f; ep[EP_MAX_SPD] = 0.0f;
    for (int i = 0; i < 4; i++) ep[EP_REW + i] = 0.0f;
    for (int i = 0; i < 4; i++) ep[EP_NOISE + i] = E.noise_on[i] ? d.uniform(E.noise_lo[i], E.noise_hi[i]) : 0.0f;   // PGE:176-179 (dict order)
    // PGE:181-195: start pose = the stored pose rotated IN PLACE about z by 360 * rand() degrees, at (0, 0, 0.5)
    const float half = 0.5f * 360.0f * d.u01() * 0.017453292519943295f;
    Q4 prev = {prev_orn[0], prev_orn[1], prev_orn[2], prev_orn[3]};
    Q4 rz = {0.0f, 0.0f, sinf(half), cosf(half)};
    Q4 orn = qmul(prev, rz);
    ep[EP_INIT_ORN + 0] = orn.x; ep[EP_INIT_ORN + 1] = orn.y; ep[EP_INIT_ORN + 2] = orn.z; ep[EP_INIT_ORN + 3] = orn.w;
    bs.p = mk3<float>(0.0f, 0.0f, 0.5f);
    bs.q = orn;
    bs.v = mk3<float>(E.init_state[7], E.init_state[8], E.init_state[9]);
    bs.w = mk3<float>(E.init_state[10], E.init_state[11], E.init_state[12]);
    for (int j = 0; j < 3; j++) { q[j] = ln.ldl(E.init_state, 13 + j, 3); qd[j] = ln.ldl(E.init_state, 25 + j, 3); }
    const float ddx = bs.p.x - ep[EP_TARGET], ddy = bs.p.y - ep[EP_TARGET + 1];
    ep[EP_LAST_DIFF] = sqrtf(ddx * ddx + ddy * ddy);                              // PGE:191-192
    ep[EP_INIT_DIFF] = (E.element_id == 0) ? -1.0f : ep[EP_LAST_DIFF];
    // (target_spd survives a reset, PGE:172; it is redrawn at counter 0)
    ep[EP_STEP_DRAWS] = 0.0f;
  }
  static LL_HD void randomize_force(const EpmcParams& E, EpmcDraws& d, float* ep) {   // PR:88-98
    const float theta = d.uniform(0.0f, 6.283185307179586f);
    const float h = d.uniform(E.hforce_lo, E.hforce_hi), v = d.uniform(E.vforce_lo, E.vforce_hi);
    ep[EP_PUSH_FORCE + 0] = h * cosf(theta); ep[EP_PUSH_FORCE + 1] = h * sinf(theta); ep[EP_PUSH_FORCE + 2] = v;
  }
  static LL_HD void load_ep(const float* g, float* ep) {
    for (int i = 0; i < EPMC_EP_STRIDE; i++) ep[i] = g[i];
  }
  static LL_HD void store_ep(const L& ln, float* g, const float* ep) {
    if (ln.lane0()) for (int i = 0; i < EPMC_EP_STRIDE; i++) g[i] = ep[i];
  }
  static LL_HD bool check_fall(const M3<float>& R) {                                // LR:159-179
    const float left_z = R.m[2] * R.m[3] - R.m[5] * R.m[0];
    return left_z > 0.70710678118654752f || left_z < -0.70710678118654752f || R.m[8] < 0.5f;
  }

  // kernel body of ll_epmc_reset
  static LL_HD void reset_env(const L& ln, const StepParams& P, const EpmcParams& E, int env, const float* draws_row, const float* prev_orn_row) {
    const int N = P.n_envs;
    float ep[EPMC_EP_STRIDE];
    load_ep(E.ep + (long)env * EPMC_EP_STRIDE, ep);
    const uint32_t episode = (uint32_t)ep[EP_EPISODE] + 1u;
    ep[EP_EPISODE] = (float)episode;
    EpmcDraws d = {draws_row, EPMC_MAX_DRAWS, 0, P.seed, (uint32_t)env, episode, 0x7e44a1u};
    Base bs;
    F q[3], qd[3];
    float prev[4];
    for (int i = 0; i < 4; i++) prev[i] = prev_orn_row ? prev_orn_row[i] : ep[EP_INIT_ORN + i];
    reset_scalars(ln, P, E, env, ep, d, bs, q, qd, prev);
    F zero3[3] = {ln.lane_f(0.0f), ln.lane_f(0.0f), ln.lane_f(0.0f)};
    typename K::ObsIn hist;
    for (int c = 0; c < K::OBS_HIST_CHUNKS; c++) hist.h[c] = ln.lane_f(0.0f);
    hist.ha[0] = hist.ha[1] = ln.lane_f(0.0f);
    float* row = P.obs + (long)env * P.obs_dim;
    store_ep(ln, E.ep + (long)env * EPMC_EP_STRIDE, ep);                          // the boxes and scalars observe() reads
    observe(ln, P, E, env, row, true, hist, bs, q, qd, zero3, ep, ep + EP_TARGET, ep[EP_TARGET_SPD]);
    K::store_state(ln, P.state, N, env, bs, q, qd);
    P.done[env] = 0;
    P.done_reason[env] = 0;
  }

  // ------------------------------------------------------------------------------------------------------------
  // the control step (PGE:299-364)
  // ------------------------------------------------------------------------------------------------------------
  // PARK (the larger-batch build): the 40 per-env scalars wait in LDS and the history chunks are read after the substep loop (sepmc_step.hpp)
  template <bool PARK = false, bool CONE = false, bool XROWS = false>   // CONE: the cone-coupled friction solve (LLM_SPEC_FRICTION_MODE = 2, Pmc::gs_cone_round); XROWS: Pmc::substep_impl
  static LL_HD void step_env(const L& ln, const StepParams& P_in, const EpmcParams& E, int env, const F* act_in) {
    const StepParams& P = ln.params(P_in);
    const int N = P.n_envs;
    Base bs;
    F q[3], qd[3], act[3], tgt[3];
    K::load_state(ln, P.state, N, env, bs, q, qd);
    float ep[EPMC_EP_STRIDE];
    load_ep(E.ep + (long)env * EPMC_EP_STRIDE, ep);
    float* row = P.obs + (long)env * P.obs_dim;
    typename K::ObsIn hist;
    if (!PARK) {
      const int Pd = P.prop_dim;
      for (int c = 0; c < K::OBS_HIST_CHUNKS; c++) hist.h[c] = ln.ld16(row + Pd, 16 * c, 2 * Pd);
      for (int c = 0; c < 2; c++) hist.ha[c] = ln.ld16(row + 3L * Pd + 12, 16 * c, 24);
    }
    for (int j = 0; j < 3; j++) {
      act[j] = act_in[j];
      F t = q[j] + act[j];                                                     // PGE:323
      tgt[j] = lm::min_(lm::max_(t, ln.lane_f(-3.0f)), ln.lane_f(3.0f));        // LR:126-127
    }
    EpmcDraws d = {E.scr_draws ? E.scr_draws + (long)env * E.scr_n_draws : nullptr, E.scr_n_draws, E.scr_draws ? 0 : (int)ep[EP_STEP_DRAWS], P.seed, (uint32_t)env,
                   (uint32_t)ep[EP_EPISODE], 0x57e9d3u};
    const int counter = (int)ep[EP_COUNTER], cmd_freq = (int)ep[EP_CMD_FREQ];
    if (E.element_id == 0 && counter % cmd_freq == 0) {                          // PGE:300-311 joystick: a new target 100 m away
      const float ang = d.uniform(0.0f, 6.283185307179586f);
      ep[EP_TARGET + 0] = bs.p.x + cosf(ang) * 100.0f;
      ep[EP_TARGET + 1] = bs.p.y + sinf(ang) * 100.0f;
      ep[EP_TARGET + 2] = 0.0f;
      const float ddx = bs.p.x - ep[EP_TARGET], ddy = bs.p.y - ep[EP_TARGET + 1];
      ep[EP_LAST_DIFF] = sqrtf(ddx * ddx + ddy * ddy);
    }
    if (counter % cmd_freq == 0) ep[EP_TARGET_SPD] = d.uniform(E.spd_lo, E.spd_hi);   // PGE:312-313

    typename K::SubstepExtra ex;
    ex.want_touch = false; ex.flag_shape = -1; ex.pair_active = false; ex.pair_me = 0;
    ex.mu_foot = ep[EP_FRICTION] * E.plane_friction;
    // terrain within reach of the robot's contact candidates during this control step: boxes whose footprint, grown by 0.9 m
    // (leg reach 0.45 m + the distance covered in 20 ms + margin), contains the base; kept in the row's LDS scratch
    {
      float* near = ln.row_scratch();
      const float* allb = E.boxes + (long)env * EPMC_MAX_BOXES * EPMC_BOX_WORDS;
      const int nb = (int)ep[EP_N_BOXES];
      int n_near = 0;
      for (int b = 0; b < nb; b++) {
        const float* bx = allb + b * EPMC_BOX_WORDS;
        if (bs.p.x >= bx[0] - 0.9f && bs.p.x <= bx[1] + 0.9f && bs.p.y >= bx[2] - 0.9f && bs.p.y <= bx[3] + 0.9f && bs.p.z <= bx[5] + 0.9f) {
          if (n_near < EPMC_MAX_NEAR && ln.lane0())
            for (int i = 0; i < EPMC_BOX_WORDS; i++) near[n_near * EPMC_BOX_WORDS + i] = bx[i];
          n_near++;
        }
      }
      ln.row_sync();
      ex.shapes = near;
      ex.n_shapes = (E.terrain_contacts && !E.scr_state) ? (n_near < EPMC_MAX_NEAR ? n_near : EPMC_MAX_NEAR) : 0;
      ex.box_mu_scale = E.box_friction / E.plane_friction;
    }
    float* ptrace = E.push_trace + (long)env * P.n_sub * 4;
    if (PARK) ln.park_row(ep, EPMC_EP_STRIDE, EPMC_PARK_AT);                     // only the push counter and force are touched in the loop
    for (int s = 0; s < P.n_sub; s++) {                                          // PGE:326-331
      ex.has_push = false;
      if (E.push_enabled) {                                                      // PR:56-86, counted in substeps
        int c = (int)ep[EP_PUSH_COUNT] + 1;
        if (c > 0) {
          if (c % E.push_interval_step == 0) { randomize_force(E, d, ep); c = 0; }
          if (c < E.push_duration_step) {
            ex.has_push = true;
            for (int i = 0; i < 3; i++) ex.push[i] = ep[EP_PUSH_FORCE + i] * E.push_ratio;
          }
        }
        ep[EP_PUSH_COUNT] = (float)c;
      }
      if (ln.lane0()) {
        ptrace[s * 4 + 0] = ex.has_push ? 1.0f : 0.0f;
        for (int i = 0; i < 3; i++) ptrace[s * 4 + 1 + i] = ex.has_push ? ex.push[i] : 0.0f;
      }
      if (!E.scr_state) K::template substep_impl<true, false, CONE, XROWS>(ln, P, bs, q, qd, tgt, env, s, &ex, nullptr);   // PGE:328-330
    }
    if (PARK) {
      const float keep[4] = {ep[EP_PUSH_COUNT], ep[EP_PUSH_FORCE], ep[EP_PUSH_FORCE + 1], ep[EP_PUSH_FORCE + 2]};
      ln.unpark_row(ep, EPMC_EP_STRIDE, EPMC_PARK_AT);
      ep[EP_PUSH_COUNT] = keep[0]; ep[EP_PUSH_FORCE] = keep[1]; ep[EP_PUSH_FORCE + 1] = keep[2]; ep[EP_PUSH_FORCE + 2] = keep[3];
      const int Pd = P.prop_dim;
      for (int c = 0; c < K::OBS_HIST_CHUNKS; c++) hist.h[c] = ln.ld16(row + Pd, 16 * c, 2 * Pd);
      for (int c = 0; c < 2; c++) hist.ha[c] = ln.ld16(row + 3L * Pd + 12, 16 * c, 24);
    }
    if (E.scr_state) {   // parity hook: the caller plays PyBullet
      const float* ss = E.scr_state + (long)env * 37;
      bs.p = mk3<float>(ss[0], ss[1], ss[2]);
      bs.q.x = ss[3]; bs.q.y = ss[4]; bs.q.z = ss[5]; bs.q.w = ss[6];
      bs.v = mk3<float>(ss[7], ss[8], ss[9]);
      bs.w = mk3<float>(ss[10], ss[11], ss[12]);
      for (int j = 0; j < 3; j++) { q[j] = ln.ldl(ss, 13 + j, 3); qd[j] = ln.ldl(ss, 25 + j, 3); }
    }
    F fin = q[0] + q[1] + q[2] + qd[0] + qd[1] + qd[2];
    float chk = L::qsum(fin) + bs.p.x + bs.p.y + bs.p.z + bs.q.x + bs.q.y + bs.q.z + bs.q.w + bs.v.x + bs.v.y + bs.v.z + bs.w.x + bs.w.y + bs.w.z;
    const bool bad = !(fabsf(chk) < 1e30f);

    // --- termination (PGE:366-379) and reward (PGE:474-539) of this transition ---
    const int cnt = counter + 1;                                                 // PGE:342
    ep[EP_COUNTER] = (float)cnt;
    const M3<float> R = qmat(qnormalize(bs.q));
    const float gx = ep[EP_TARGET] - bs.p.x, gy = ep[EP_TARGET + 1] - bs.p.y;
    // (sums of two products written as product + fused multiply-add: which of the two -ffp-contract=fast would fuse differs between builds of the
    //  kernel, and ll_epmc_step_random_n must equal single steps bit for bit)
    const float dist = sqrtf(__builtin_fmaf(gx, gx, gy * gy));
    int reason = 0;
    if (check_fall(R)) reason |= 1;
    if (cnt >= E.max_steps) reason |= 2;
    const bool reach = dist < 0.5f;
    if (reach) reason |= 4;
    if (bad) reason |= 16;
    const float ux = gx / dist, uy = gy / dist;
    const float spd = fabsf(__builtin_fmaf(bs.v.x, ux, bs.v.y * uy));                           // PGE:478-480
    ep[EP_TOTAL_SPD] += spd;
    if (spd > ep[EP_MAX_SPD]) ep[EP_MAX_SPD] = spd;
    const float yaw = atan2f(R.m[3], R.m[0]);
    const float r_rot = expf((__builtin_fmaf(cosf(yaw), ux, sinf(yaw) * uy) - 1.0f) * 5.0f);
    const float inv_ms = 1.0f / (float)E.max_steps;
    float reward;
    if (E.element_id == 0) {                                                      // joystick, PGE:474-497
      const float r_vel = expf(-fabsf(spd - ep[EP_TARGET_SPD]));
      reward = r_vel * r_rot * inv_ms;
      ep[EP_REW + 1] = __builtin_fmaf(r_rot, inv_ms, ep[EP_REW + 1]);      // (explicit fused multiply-adds: see `dist` above -- round 6: the multi-step and the single-step builds had
      ep[EP_REW + 0] = __builtin_fmaf(r_vel, inv_ms, ep[EP_REW + 0]);      //  contracted these sums differently, a reward 50 ulp apart under cancellation)
    } else {                                                                      // average speed, PGE:499-539
      const float r_dist = ep[EP_INIT_DIFF] >= 0.0f ? (dist - ep[EP_LAST_DIFF]) / ep[EP_INIT_DIFF] : 0.0f;
      ep[EP_LAST_DIFF] = dist;
      const float s_rot = r_rot * inv_ms * 0.1f, s_dist = -r_dist * 0.1f;
      reward = __builtin_fmaf(s_rot, 2.0f, s_dist);
      ep[EP_REW + 1] = __builtin_fmaf(s_rot, 2.0f, ep[EP_REW + 1]);
      ep[EP_REW + 2] += s_dist;
      if (reach) {
        const float r_avg = expf(-fabsf(ep[EP_TOTAL_SPD] / (float)cnt - ep[EP_TARGET_SPD]));
        reward += r_avg;
        ep[EP_REW + 3] += r_avg;
      }
    }
    if (bad) reward = 0.0f;
    if (!E.scr_draws) ep[EP_STEP_DRAWS] = (float)d.used;

    F oact[3] = {act[0], act[1], act[2]};
    bool fill = false;
    if (reason) {
      if (ln.lane0()) {                                                           // PGE:356-362
        float* inf = E.info + (long)env * 6;
        inf[0] = ep[EP_TOTAL_SPD] / (float)cnt; inf[1] = ep[EP_MAX_SPD];
        for (int i = 0; i < 4; i++) inf[2 + i] = ep[EP_REW + i];
      }
      K::count_add(ln, P.counters + 1);
      if (bad) K::count_add(ln, P.counters + 2);
      if (P.auto_reset) {                                                         // re-seed inside the step: new terrain, start pose, then the common tail
        const uint32_t episode = (uint32_t)ep[EP_EPISODE] + 1u;
        ep[EP_EPISODE] = (float)episode;
        EpmcDraws dr = {nullptr, 0, 0, P.seed, (uint32_t)env, episode, 0x7e44a1u};
        float prev[4] = {ep[EP_INIT_ORN], ep[EP_INIT_ORN + 1], ep[EP_INIT_ORN + 2], ep[EP_INIT_ORN + 3]};
        reset_scalars(ln, P, E, env, ep, dr, bs, q, qd, prev);
        for (int j = 0; j < 3; j++) oact[j] = ln.lane_f(0.0f);
        fill = true;
      }
    }
    store_ep(ln, E.ep + (long)env * EPMC_EP_STRIDE, ep);
    observe(ln, P, E, env, row, fill, hist, bs, q, qd, oact, ep, ep + EP_TARGET, ep[EP_TARGET_SPD]);   // PGE:335-338 (raw action in the history)
    K::store_state(ln, P.state, N, env, bs, q, qd);
    P.reward[env] = reward;
    P.done[env] = reason ? 1 : 0;
    P.done_reason[env] = (uint8_t)reason;
  }
};
