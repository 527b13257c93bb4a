// sepmc_step.hpp -- the SEPMC (strategic-level) control step: two robots in one walled arena, generic over the lane policy.
//
// What it replaces, cited as
//   CTG = src/lifelike/sim_envs/pybullet_envs/max_game/chase_tag_game_env.py
//   BS4 = src/lifelike/sim_envs/pybullet_envs/max_game/bullet_static_entities.py  (BulletStaticsV4, :830-1019)
//   PR  = src/lifelike/sim_envs/pybullet_envs/randomizer/push_randomizer.py
//
// Execution model: a robot is a 16-lane row exactly as in pmc_step.hpp / epmc_step.hpp; the two robots of arena a are rows
// 2a and 2a+1 -- neighbours in the same wave -- and read each other's state with L::peer (v_permlane16_swap on gfx950).
// Arena-level logic (draws, flag, contact bookkeeping, termination, rewards) is computed by both rows from the same inputs, so
// it needs no further communication; every per-robot buffer (state, obs, reward, done, boxes, scalars) is indexed by row.
#pragma once
#include "epmc_step.hpp"

#define SEPMC_OBS_TAIL 52        // percept_vec 5, oppo_info 15, oppo_info_cheat 15, flag_info 7, flag_info_cheat 7, with_flag 2, control_spd 1
#define SEPMC_SP_STRIDE 40
#define SEPMC_PARK_AT EPMC_SPARE // row-scratch word: the 64 spare words behind the ray lists (epmc_step.hpp)
#define SEPMC_MAX_CONTACTS 8     // scripted getContactPoints records per arena
#define SEPMC_N_VIS 21           // visibility rays per arena: base to base, then head of robot i to the 10 convex points of the other
#define SEPMC_WALL_SOLID 1.0f     // metres the arena walls extend outwards for the contact tests (the rays see their true 1 cm)
#define SEPMC_VIS_SCRATCH EPMC_SPARE    // word of the row scratch where the visibility end points go (after the staged boxes and the ray lists)

enum SepmcField {
  SP_FLAG = 0,          // 3 flag position (CTG:231-236)
  SP_WITH_FLAG0 = 3,    // 1: robot 0 holds the flag (self.with_flag[0])
  SP_FRICTION = 4,
  SP_FIX_SPD = 5,
  SP_COUNTER = 6,
  SP_PUSH_FORCE = 7,    // 3 PushRandomizer._randomized_force
  SP_NOISE = 10,        // 4
  SP_LAST_TWO = 14,
  SP_LAST_ESC = 15,
  SP_TOTAL_SPD = 16,    // 2
  SP_MAX_SPD = 18,      // 2
  SP_PUSH_COUNT = 20,
  SP_STEP_DRAWS = 21,
  SP_INIT_ORN = 22,     // 4 the shared start orientation, rotated in place by both robots at every reset (CTG:224-229)
  SP_EPISODE = 26,
  SP_N_BOXES = 27,      // arena boxes; the flag is record n_boxes
  SP_SWITCH = 28,
  SP_VISIBLE = 29,      // this robot sees the other (CTG:472-493)
  SP_WHO0 = 30,         // _detect_body_contact(robot 0) of the last step: -1 none, 0 plane, 1 static, 2 flag, 3 robot 0, 4 robot 1
  SP_WHO_T = 31,        // ... of the robot that could take the flag
};

struct SepmcParams {
  EpmcParams e;         // the fields the two envs share: max_steps, push schedule, friction / force / noise ranges, init_state, per-row boxes,
                        // push_trace, ray_trace and the scripted-state / scripted-ray / scripted-draw hooks (all indexed by robot row)
  int32_t rand_cube, hurdle, hole, scr_on;
  int32_t robot_contacts, pad0;           // 0: the robots pass through each other (diagnostics)
  float cos_visible, control_spd;         // control_spd < 0: the episode's draw (CTG:264, :361)
  float* sp;                              // [rows][SEPMC_SP_STRIDE]
  float* info;                            // [rows][4] avg_spd0, avg_spd1, max_spd0, max_spd1 (CTG:404-409)
  float* vis_trace;                       // optional [rows][16][8]: from 3, to 3, blocked, valid -- by lane: leg * 4 + {foot, wheel, handle, base}
  const uint8_t* scr_vis;                 // [arenas][SEPMC_N_VIS] blocked flags (parity hook)
  const int32_t* scr_contacts;            // [arenas][SEPMC_MAX_CONTACTS][4] bodyA bodyB linkA linkB, bodyA = -9 ends the list
};

template <class L>
struct Sepmc {
  typedef Pmc<L> K;
  typedef Epmc<L> EP;
  typedef typename L::F F;
  typedef typename L::B B;
  typedef typename K::Base Base;
  typedef V3<F> V3l;

  // what a row knows about the other robot of its arena after an exchange
  struct Peer {
    Base bs;
    V3l foot, wheel, handle;      // world positions of the other robot's link origins, lane (leg g, .) holds those of ITS leg g
    float head[3];                // my own front handle (LR:154-156)
    float code;                   // the other robot's contact classes (bit 0 static, 1 flag, 2 robot)
  };

  static LL_HD void link_points(const L& ln, const StepParams& P, const Base& bs, const F* q, V3l& foot, V3l& wheel, V3l& handle) {
    const M3<float> R = qmat(qnormalize(bs.q));
    typename K::LegKin k = K::leg_fk(ln, P.legc, q[0], q[1], q[2]);
    V3l fb = k.p3 + mul(k.R3, K::ld3c(ln, P.legc, LC_FOOT));
    V3l wb = k.p2 + mul(k.R2, K::ld3c(ln, P.legc, LC_WHEEL));
    V3l hb = K::ld3c(ln, P.legc, LC_HANDLE);
    V3l fw = mul(R, fb), ww = mul(R, wb), hw = mul(R, hb);
    foot = mk3<F>(fw.x + bs.p.x, fw.y + bs.p.y, fw.z + bs.p.z);
    wheel = mk3<F>(ww.x + bs.p.x, ww.y + bs.p.y, ww.z + bs.p.z);
    handle = mk3<F>(hw.x + bs.p.x, hw.y + bs.p.y, hw.z + bs.p.z);
  }
  static LL_HD Peer exchange(const L& ln, const StepParams& P, const Base& bs, const F* q, float my_code) {
    Peer o;
    V3l foot, wheel, handle;
    link_points(ln, P, bs, q, foot, wheel, handle);
    o.head[0] = L::template bcast<0>(handle.x); o.head[1] = L::template bcast<0>(handle.y); o.head[2] = L::template bcast<0>(handle.z);
    o.foot = mk3<F>(ln.peer(foot.x), ln.peer(foot.y), ln.peer(foot.z));
    o.wheel = mk3<F>(ln.peer(wheel.x), ln.peer(wheel.y), ln.peer(wheel.z));
    o.handle = mk3<F>(ln.peer(handle.x), ln.peer(handle.y), ln.peer(handle.z));
    o.bs.p = mk3<float>(ln.peer_u(bs.p.x), ln.peer_u(bs.p.y), ln.peer_u(bs.p.z));
    o.bs.q.x = ln.peer_u(bs.q.x); o.bs.q.y = ln.peer_u(bs.q.y); o.bs.q.z = ln.peer_u(bs.q.z); o.bs.q.w = ln.peer_u(bs.q.w);
    o.bs.v = mk3<float>(ln.peer_u(bs.v.x), ln.peer_u(bs.v.y), ln.peer_u(bs.v.z));
    o.bs.w = mk3<float>(ln.peer_u(bs.w.x), ln.peer_u(bs.w.y), ln.peer_u(bs.w.z));
    o.code = ln.peer_u(my_code);
    return o;
  }

  // ------------------------------------------------------------------------------------------------------------
  // arena (BS4:841-1012): lo/hi records of the boxes in creation order; the flag is appended by the caller
  // ------------------------------------------------------------------------------------------------------------
  struct Arena {
    float* boxes;
    int n;
    bool store;
    LL_HD void box(float x, float y, float z, float l, float w, float h) {
      if (store && n < EPMC_MAX_BOXES - 1) {
        float* b = boxes + n * EPMC_BOX_WORDS;
        b[0] = x - l * 0.5f; b[1] = x + l * 0.5f; b[2] = y - w * 0.5f; b[3] = y + w * 0.5f; b[4] = z - h * 0.5f; b[5] = z + h * 0.5f; b[6] = 0.0f; b[7] = 0.0f;
      }
      if (n < EPMC_MAX_BOXES - 1) n++;
    }
  };
  static LL_HD void gen_arena(Arena& A, EpmcDraws& d, const SepmcParams& S) {
    A.box(0.0f, 2.5f, 1.0f, 5.0f, 0.01f, 2.0f); A.box(0.0f, -2.5f, 1.0f, 5.0f, 0.01f, 2.0f);       // BS4:863-902
    A.box(2.5f, 0.0f, 1.0f, 0.01f, 5.0f, 2.0f); A.box(-2.5f, 0.0f, 1.0f, 0.01f, 5.0f, 2.0f);
    if (S.rand_cube) {                                                                            // BS4:904-945
      const int n = d.randint(5, 6);
      for (int i = 0; i < n; i++) {
        const float h = d.uniform(0.05f, 0.25f);
        const float x = d.uniform(-2.0f, 2.0f), y = d.uniform(-2.0f, 2.0f);
        const float l = d.uniform(0.5f, 1.0f), w = d.uniform(0.5f, 1.0f);
        A.box(x, y, h * 0.5f, l, w, h);
      }
    }
    if (S.hurdle) {                                                                               // BS4:947-979
      const float h = d.uniform(0.05f, 0.15f);
      A.box(0.0f, 0.0f, h * 0.5f, 0.1f, 5.0f, h);
    }
    if (S.hole) {                                                                                 // BS4:981-1012
      const float gap = d.uniform(0.25f, 0.3f);
      A.box(0.0f, 0.0f, 0.15f + gap, 5.0f, 0.1f, 0.3f);
    }
  }
  static LL_HD void put_flag_box(const L& ln, float* boxes, int n, const float* flag) {             // CTG:163-189: 0.1 x 0.1 x 0.5
    if (ln.lane0()) {
      float* b = boxes + n * EPMC_BOX_WORDS;
      b[0] = flag[0] - 0.05f; b[1] = flag[0] + 0.05f; b[2] = flag[1] - 0.05f; b[3] = flag[1] + 0.05f; b[4] = flag[2] - 0.25f; b[5] = flag[2] + 0.25f;
      b[6] = 0.0f; b[7] = 0.0f;
    }
  }
  static LL_HD void load_sp(const float* g, float* sp) { for (int i = 0; i < SEPMC_SP_STRIDE; i++) sp[i] = g[i]; }
  static LL_HD void store_sp(const L& ln, float* g, const float* sp) {
    if (ln.lane0()) for (int i = 0; i < SEPMC_SP_STRIDE; i++) g[i] = sp[i];
  }
  static LL_HD void randomize_force(const EpmcParams& E, EpmcDraws& d, float* sp) {                  // PR:88-98
    const float theta = d.uniform(0.0f, 6.283185307179586f);
    const float h = d.uniform(E.hforce_lo, E.hforce_hi), v = d.uniform(E.vforce_lo, E.vforce_hi);
    sp[SP_PUSH_FORCE + 0] = h * cosf(theta); sp[SP_PUSH_FORCE + 1] = h * sinf(theta); sp[SP_PUSH_FORCE + 2] = v;
  }
  // CTG:231-243 (the positions are the robots' true base positions)
  static LL_HD void randomize_flag(EpmcDraws& d, float* sp, const float* pos0, const float* pos1) {
    sp[SP_FLAG + 0] = d.uniform(-2.0f, 2.0f); sp[SP_FLAG + 1] = d.uniform(-2.0f, 2.0f); sp[SP_FLAG + 2] = 0.25f;
    const float* o = sp[SP_WITH_FLAG0] > 0.5f ? pos1 : pos0;
    const float dx = sp[SP_FLAG] - o[0], dy = sp[SP_FLAG + 1] - o[1];
    sp[SP_LAST_ESC] = sqrtf(dx * dx + dy * dy);
  }

  // ------------------------------------------------------------------------------------------------------------
  // reset (CTG:263-299) up to and including randomize_init_states; the caller exchanges and observes afterwards
  // ------------------------------------------------------------------------------------------------------------
  static LL_HD void reset_scalars(const L& ln, const StepParams& P, const SepmcParams& S, int row, float* sp, EpmcDraws& d, Base& bs, F* q, F* qd,
                                  const float* prev_orn) {
    const EpmcParams& E = S.e;
    const int me = row & 1;
    sp[SP_FIX_SPD] = d.uniform(0.5f, 3.0f);                                        // CTG:264
    sp[SP_TOTAL_SPD] = sp[SP_TOTAL_SPD + 1] = sp[SP_MAX_SPD] = sp[SP_MAX_SPD + 1] = 0.0f;
    Arena A;
    A.boxes = E.boxes + (long)row * EPMC_MAX_BOXES * EPMC_BOX_WORDS; A.n = 0; A.store = ln.lane0();
    gen_arena(A, d, S);                                                             // CTG:267
    sp[SP_N_BOXES] = (float)A.n;
    sp[SP_WITH_FLAG0] = (float)d.randint(0, 2);                                    // CTG:268-269
    sp[SP_SWITCH] = 0.0f; sp[SP_COUNTER] = 0.0f;
    sp[SP_FRICTION] = d.uniform(E.friction_lo, E.friction_hi);                     // CTG:279
    if (E.push_enabled) {                                                          // CTG:284-285, PR:52-54
      sp[SP_PUSH_COUNT] = (float)E.push_count0;
      randomize_force(E, d, sp);
    }
    for (int i = 0; i < 4; i++) sp[SP_NOISE + i] = E.noise_on[i] ? d.uniform(E.noise_lo[i], E.noise_hi[i]) : 0.0f;   // CTG:207-210
    // CTG:212-229: both start positions, then for robot 0 and robot 1 in turn the SHARED start orientation is turned in place
    float pos[2][3];
    pos[0][0] = d.uniform(-2.0f, 2.0f); pos[0][1] = d.uniform(-2.0f, 2.0f); pos[0][2] = 0.5f;
    pos[1][0] = d.uniform(-2.0f, 2.0f); pos[1][1] = d.uniform(-2.0f, 2.0f); pos[1][2] = 0.5f;
    {
      const float dx = pos[1][0] - pos[0][0], dy = pos[1][1] - pos[0][1];
      sp[SP_LAST_TWO] = sqrtf(dx * dx + dy * dy);
    }
    Q4 orn = {prev_orn[0], prev_orn[1], prev_orn[2], prev_orn[3]}, mine = orn;
    for (int i = 0; i < 2; i++) {
      const float half = 0.5f * 360.0f * d.u01() * 0.017453292519943295f;
      Q4 rz = {0.0f, 0.0f, sinf(half), cosf(half)};
      orn = qmul(orn, rz);
      if (i == me) mine = orn;
    }
    sp[SP_INIT_ORN + 0] = orn.x; sp[SP_INIT_ORN + 1] = orn.y; sp[SP_INIT_ORN + 2] = orn.z; sp[SP_INIT_ORN + 3] = orn.w;
    bs.p = mk3<float>(pos[me][0], pos[me][1], pos[me][2]);
    bs.q = mine;
    bs.v = mk3<float>(E.init_state[7], E.init_state[8], E.init_state[9]);
    bs.w = mk3<float>(E.init_state[10], E.init_state[11], E.init_state[12]);
    for (int j = 0; j < 3; j++) { q[j] = ln.ldl(E.init_state, 13 + j, 3); qd[j] = ln.ldl(E.init_state, 25 + j, 3); }
    randomize_flag(d, sp, pos[0], pos[1]);                                          // CTG:230
    put_flag_box(ln, A.boxes, A.n, sp + SP_FLAG);
    ln.row_sync();
    sp[SP_STEP_DRAWS] = 0.0f;
  }

  // ------------------------------------------------------------------------------------------------------------
  // contact bookkeeping (CTG:426-470)
  // ------------------------------------------------------------------------------------------------------------
  static LL_HD bool body_link(int l) { return l >= 0 && l < 20 && (l % 5) != 3; }                    // leg + wheel links, CTG:427
  // _detect_body_contact on a scripted getContactPoints list: the other body of the FIRST record one of my body links is in
  static LL_HD int detect_scripted(const int32_t* c, int me_body) {
    for (int i = 0; i < SEPMC_MAX_CONTACTS; i++) {
      const int a = c[4 * i], b = c[4 * i + 1], la = c[4 * i + 2], lb = c[4 * i + 3];
      if (a == -9) break;
      if (a == me_body && b == me_body) continue;
      if (a == me_body) { if (body_link(la)) return b; }
      else if (b == me_body) { if (body_link(lb)) return a; }
    }
    return -1;
  }
  // ... and on this build's own contact classes, listed in the order plane / boxes, flag, other robot
  static LL_HD int detect_classes(float code, int other_body) {
    const int c = (int)code;
    return (c & 1) ? 1 : ((c & 2) ? 2 : ((c & 4) ? other_body : -1));
  }

  // ------------------------------------------------------------------------------------------------------------
  // observation of one robot (CTG:495-596 minus the contact part, CTG:331-363)
  // ------------------------------------------------------------------------------------------------------------
  static LL_HD void observe(const L& ln, const StepParams& P, const SepmcParams& S, int row, float* orow, bool fill, const typename K::ObsIn& hist,
                            const Base& bs, const F* q, const F* qd, const F* act, float* sp, const Peer& o, const float* flag_obs) {
    const EpmcParams& E = S.e;
    const int me = row & 1, arena = row >> 1;
    const M3<float> R = qmat(qnormalize(bs.q)), Ro = qmat(qnormalize(o.bs.q));
    K::obs_emit_core(ln, P, orow, fill, hist, bs, R, q, qd, act);
    float pos[3] = {bs.p.x, bs.p.y, bs.p.z}, opos[3] = {o.bs.p.x, o.bs.p.y, o.bs.p.z};
    float yaw = atan2f(R.m[3], R.m[0]), oyaw = atan2f(Ro.m[3], Ro.m[0]);                              // CTG:501
    if (E.noise_on[0]) {                                                                              // CTG:503-508
      pos[0] += sp[SP_NOISE + 0]; pos[1] += sp[SP_NOISE + 1];
      opos[0] += sp[SP_NOISE + 0]; opos[1] += sp[SP_NOISE + 1];
    }
    if (E.noise_on[2]) { yaw += sp[SP_NOISE + 2]; oyaw += sp[SP_NOISE + 2]; }                         // CTG:509-510
    const long a0 = 3L * P.prop_dim + 36;
    const int nb = (int)sp[SP_N_BOXES] + 1;                                                          // the arena and the flag
    const float* boxes = ln.stage_row(E.boxes + (long)row * EPMC_MAX_BOXES * EPMC_BOX_WORDS, nb * EPMC_BOX_WORDS);
    if (LL_NO_FUSED_RAYS || (E.split_rays && !E.scr_ray_hit)) EP::leave_ray_pose(ln, E, row, pos, R, yaw, sp + SP_NOISE, nb, boxes + (nb - 1) * EPMC_BOX_WORDS);  // (round 6: the 778 rays by the kernel behind this one, epmc_step.hpp percept_rays; the flag's box as it stands NOW)
    else EP::observe_rays(ln, P, E, row, pos, R, yaw, sp + SP_NOISE, boxes, nb, orow + a0);           // CTG:515-531
    // --- visibility (CTG:472-493): lane (leg g, sub s) owns one segment: my head -> the other's foot g / wheel g / handle (g = 0, 2),
    //     and lane (0, 3) the segment between the two (biased) base positions
    float* vsc = ln.row_scratch() + SEPMC_VIS_SCRATCH;
    {
      B s0 = ln.is_sub(0), s1 = ln.is_sub(1);
      F tx = lm::sel(s0, o.foot.x, lm::sel(s1, o.wheel.x, o.handle.x)), ty = lm::sel(s0, o.foot.y, lm::sel(s1, o.wheel.y, o.handle.y)),
        tz = lm::sel(s0, o.foot.z, lm::sel(s1, o.wheel.z, o.handle.z));
      ln.st16(vsc, 0, 16, tx); ln.st16(vsc + 16, 0, 16, ty); ln.st16(vsc + 32, 0, 16, tz);
      ln.row_sync();
    }
    float clear = 0.0f, root_clear = 0.0f;
    for (int k = ln.ray_first(); k < 16; k += ln.ray_stride()) {
      const int g = k >> 2, s = k & 3;
      const bool valid = s < 2 || (s == 2 && (g == 0 || g == 2)) || (s == 3 && g == 0);
      if (!valid) continue;
      float f[3], t[3];
      int slot;
      if (s == 3) {                                                                  // robot 0's position to robot 1's
        for (int a = 0; a < 3; a++) { f[a] = me == 0 ? pos[a] : opos[a]; t[a] = me == 0 ? opos[a] : pos[a]; }
        slot = 0;
      } else {
        for (int a = 0; a < 3; a++) { f[a] = o.head[a]; t[a] = vsc[16 * a + k]; }
        slot = 1 + 10 * me + (s == 0 ? g : (s == 1 ? 4 + g : 8 + (g >> 1)));
      }
      bool blocked;
      if (S.scr_on) blocked = S.scr_vis[(long)arena * SEPMC_N_VIS + slot] != 0;
      else EP::cast(f, t, boxes, (1ull << nb) - 1ull, &blocked);
      if (S.vis_trace) {
        float* tr = S.vis_trace + ((long)row * 16 + k) * 8;
        tr[0] = f[0]; tr[1] = f[1]; tr[2] = f[2]; tr[3] = t[0]; tr[4] = t[1]; tr[5] = t[2]; tr[6] = blocked ? 1.0f : 0.0f; tr[7] = 1.0f;
      }
      if (!blocked) { if (s == 3) root_clear = 1.0f; else clear = 1.0f; }
    }
    const bool test_visible = L::rmin(ln.lane_f(1.0f - root_clear)) < 0.5f || L::rmin(ln.lane_f(1.0f - clear)) < 0.5f;
    const float gdx = opos[0] - pos[0], gdy = opos[1] - pos[1], gdz = opos[2] - pos[2];              // CTG:539-540
    const float cth = (cosf(yaw) * gdx + sinf(yaw) * gdy) / sqrtf(gdx * gdx + gdy * gdy);
    const bool visible = (cth >= S.cos_visible) && test_visible;                                     // CTG:489-492
    sp[SP_VISIBLE] = visible ? 1.0f : 0.0f;
    if (ln.lane0()) {
      float* w = orow + a0 + EPMC_N_RAYS;
      w[0] = pos[0]; w[1] = pos[1]; w[2] = pos[2]; w[3] = cosf(yaw); w[4] = sinf(yaw);               // percept_vec, CTG:512-513, :534
      float op[15];
      op[0] = visible ? 1.0f : 0.0f;
      op[1] = opos[0]; op[2] = opos[1]; op[3] = opos[2];
      op[4] = R.m[0] * gdx + R.m[3] * gdy + R.m[6] * gdz; op[5] = R.m[1] * gdx + R.m[4] * gdy + R.m[7] * gdz; op[6] = R.m[2] * gdx + R.m[5] * gdy + R.m[8] * gdz;
      op[7] = cosf(oyaw - yaw); op[8] = sinf(oyaw - yaw);
      const V3<float> ov = o.bs.v, ow = o.bs.w;
      op[9] = R.m[0] * ov.x + R.m[3] * ov.y + R.m[6] * ov.z; op[10] = R.m[1] * ov.x + R.m[4] * ov.y + R.m[7] * ov.z; op[11] = R.m[2] * ov.x + R.m[5] * ov.y + R.m[8] * ov.z;
      op[12] = R.m[0] * ow.x + R.m[3] * ow.y + R.m[6] * ow.z; op[13] = R.m[1] * ow.x + R.m[4] * ow.y + R.m[7] * ow.z; op[14] = R.m[2] * ow.x + R.m[5] * ow.y + R.m[8] * ow.z;
      for (int i = 0; i < 15; i++) { w[5 + i] = visible ? op[i] : 0.0f; w[20 + i] = op[i]; }        // oppo_info, oppo_info_cheat (CTG:548-562)
      const float fx = flag_obs[0] - pos[0], fy = flag_obs[1] - pos[1], fz = flag_obs[2] - pos[2];
      float fl[7] = {1.0f, flag_obs[0], flag_obs[1], flag_obs[2], R.m[0] * fx + R.m[3] * fy + R.m[6] * fz, R.m[1] * fx + R.m[4] * fy + R.m[7] * fz,
                     R.m[2] * fx + R.m[5] * fy + R.m[8] * fz};
      for (int i = 0; i < 7; i++) { w[35 + i] = fl[i]; w[42 + i] = fl[i]; }                         // flag_info, flag_info_cheat (CTG:565-577)
      const float wf0 = sp[SP_WITH_FLAG0] > 0.5f ? 1.0f : 0.0f;
      w[49] = me == 0 ? wf0 : 1.0f - wf0; w[50] = me == 0 ? 1.0f - wf0 : wf0;                        // CTG:589
      w[51] = S.control_spd >= 0.0f ? S.control_spd : sp[SP_FIX_SPD];                                // CTG:361
    }
  }

  // kernel body of ll_sepmc_reset (CTG:263-310); draws_row / prev_orn_row are per ARENA
  static LL_HD void reset_env(const L& ln, const StepParams& P, const SepmcParams& S, int row, const float* draws_row, const float* prev_orn_row) {
    const int N = P.n_envs, arena = row >> 1;
    float sp[SEPMC_SP_STRIDE];
    load_sp(S.sp + (long)row * SEPMC_SP_STRIDE, sp);
    const uint32_t episode = (uint32_t)sp[SP_EPISODE] + 1u;
    sp[SP_EPISODE] = (float)episode;
    EpmcDraws d = {draws_row, EPMC_MAX_DRAWS, 0, P.seed, (uint32_t)arena, episode, 0x5e9a1du};
    Base bs;
    F q[3], qd[3];
    float prev[4];
    for (int i = 0; i < 4; i++) prev[i] = prev_orn_row ? prev_orn_row[i] : sp[SP_INIT_ORN + i];
    reset_scalars(ln, P, S, row, sp, d, bs, q, qd, prev);
    sp[SP_WHO0] = sp[SP_WHO_T] = -1.0f;
    Peer o = exchange(ln, P, bs, q, 0.0f);
    F zero3[3] = {ln.lane_f(0.0f), ln.lane_f(0.0f), ln.lane_f(0.0f)};
    typename K::ObsIn hist;
    for (int c = 0; c < K::OBS_HIST_CHUNKS; c++) hist.h[c] = ln.lane_f(0.0f);
    hist.ha[0] = hist.ha[1] = ln.lane_f(0.0f);
    float* orow = P.obs + (long)row * P.obs_dim;
    float flag_obs[3] = {sp[SP_FLAG], sp[SP_FLAG + 1], sp[SP_FLAG + 2]};
    observe(ln, P, S, row, orow, true, hist, bs, q, qd, zero3, sp, o, flag_obs);
    // (the reference also asks getContactPoints() here, CTG:579; before the first stepSimulation of an episode this build has none)
    store_sp(ln, S.sp + (long)row * SEPMC_SP_STRIDE, sp);
    K::store_state(ln, P.state, N, row, bs, q, qd);
    P.done[row] = 0;
    P.done_reason[row] = 0;
  }

  // ------------------------------------------------------------------------------------------------------------
  // the control step (CTG:378-424)
  // ------------------------------------------------------------------------------------------------------------
  // PARK (the larger-batch build): the 40 per-row scalars wait in LDS and the history chunks are read after the substep loop instead of
  // before it (two waves per SIMD hide that round trip; one wave per SIMD would pay it)
  template <bool PARK = false, bool CONE = false, bool XROWS = false>   // CONE: the cone-coupled friction solve (LLM_SPEC_FRICTION_MODE = 2, Pmc::gs_cone_round)
  static LL_HD void step_env(const L& ln, const StepParams& P_in, const SepmcParams& S, int row, const F* act_in) {
    const StepParams& P = ln.params(P_in);
    const EpmcParams& E = S.e;
    const int N = P.n_envs, me = row & 1, arena = row >> 1;
    Base bs;
    F q[3], qd[3], act[3], tgt[3];
    K::load_state(ln, P.state, N, row, bs, q, qd);
    float sp[SEPMC_SP_STRIDE];
    load_sp(S.sp + (long)row * SEPMC_SP_STRIDE, sp);
    float* orow = P.obs + (long)row * P.obs_dim;
    typename K::ObsIn hist;
    if (!PARK) {
      const int Pd = P.prop_dim;
      for (int c = 0; c < K::OBS_HIST_CHUNKS; c++) hist.h[c] = ln.ld16(orow + Pd, 16 * c, 2 * Pd);
      for (int c = 0; c < 2; c++) hist.ha[c] = ln.ld16(orow + 3L * Pd + 12, 16 * c, 24);
    }
    for (int j = 0; j < 3; j++) {
      act[j] = act_in[j];
      F t = q[j] + act[j];                                                       // CTG:380
      tgt[j] = lm::min_(lm::max_(t, ln.lane_f(-3.0f)), ln.lane_f(3.0f));          // LR:126-127
    }
    EpmcDraws d = {E.scr_draws ? E.scr_draws + (long)arena * E.scr_n_draws : nullptr, E.scr_n_draws, E.scr_draws ? 0 : (int)sp[SP_STEP_DRAWS], P.seed, (uint32_t)arena,
                   (uint32_t)sp[SP_EPISODE], 0x57e9d3u};
    typename K::SubstepExtra ex;
    ex.mu_foot = sp[SP_FRICTION] * E.plane_friction;
    ex.want_touch = false; ex.flag_shape = -1; ex.touch_static = ex.touch_flag = ex.touch_robot = 0.0f;
    {   // can the two robots meet during this control step?  (reach 0.55 m each + what 20 ms of motion adds)
      const float ddx = ln.peer_u(bs.p.x) - bs.p.x, ddy = ln.peer_u(bs.p.y) - bs.p.y, ddz = ln.peer_u(bs.p.z) - bs.p.z;
      ex.pair_active = !E.scr_state && S.robot_contacts && !PMC_ABL(2048) && (ddx * ddx + ddy * ddy + ddz * ddz < 1.5f * 1.5f);
      ex.pair_me = me;
    }
    const int nb = (int)sp[SP_N_BOXES];
    float* allb = E.boxes + (long)row * EPMC_MAX_BOXES * EPMC_BOX_WORDS;
    {   // boxes within reach of this robot during the control step (epmc_step.hpp), the flag among them
      float* near = ln.row_scratch();
      int n_near = 0;
      for (int b = 0; b <= nb; b++) {
        const float* bx = allb + b * EPMC_BOX_WORDS;
        if (bs.p.x >= bx[0] - 0.9f && bs.p.x <= bx[1] + 0.9f && bs.p.y >= bx[2] - 0.9f && bs.p.y <= bx[3] + 0.9f && bs.p.z <= bx[5] + 0.9f) {
          if (n_near < EPMC_MAX_NEAR) {
            if (ln.lane0()) {
              for (int i = 0; i < EPMC_BOX_WORDS; i++) near[n_near * EPMC_BOX_WORDS + i] = bx[i];
              // for contacts the four arena walls (boxes 0..3, BS4:897-902: 1 cm thick) are solid outwards: a point pressed more than
              // half-way into a thin box would otherwise be pushed out of its far side
              if (b < 4) near[n_near * EPMC_BOX_WORDS + (b == 0 ? 3 : (b == 1 ? 2 : (b == 2 ? 1 : 0)))] += (b == 0 || b == 2) ? SEPMC_WALL_SOLID : -SEPMC_WALL_SOLID;
              // ... and longer by the same at both ends, so that the four thickened walls close the corners (round 5: without it a point pressed into one wall next to a
              // corner was nearest to that wall's END face and left the arena through the pocket between the two thickened walls -- 1 robot in 16 M robot-steps of the soak)
              if (b < 4) { near[n_near * EPMC_BOX_WORDS + (b < 2 ? 0 : 2)] -= SEPMC_WALL_SOLID; near[n_near * EPMC_BOX_WORDS + (b < 2 ? 1 : 3)] += SEPMC_WALL_SOLID; }
            }
            if (b == nb) ex.flag_shape = n_near;
          }
          n_near++;
        }
      }
      ln.row_sync();
      ex.shapes = near;
      ex.n_shapes = (E.terrain_contacts && !E.scr_state && !PMC_ABL(4096)) ? (n_near < EPMC_MAX_NEAR ? n_near : EPMC_MAX_NEAR) : 0;
      ex.box_mu_scale = E.box_friction / E.plane_friction;
    }
    float* ptrace = E.push_trace + (long)row * P.n_sub * 4;
    const typename K::LinkC lkh = K::own_link_held(ln, P.legc);                    // (held in registers: faster in both SEPMC kernels, also with PARK)
    if (PARK) ln.park_row(sp, SEPMC_SP_STRIDE, SEPMC_PARK_AT);                     // only the push counter and force are touched in the loop
    for (int s = 0; s < P.n_sub; s++) {                                          // CTG:383-388
      ex.has_push = false;
      if (E.push_enabled) {                                                      // PR:56-86, the legged_robots branch :78-86
        int c = (int)sp[SP_PUSH_COUNT] + 1;
        if (c > 0) {
          if (c % E.push_interval_step == 0) { randomize_force(E, d, sp); c = 0; }
          if (c < E.push_duration_step) {
            ex.has_push = true;
            float f0[3] = {sp[SP_PUSH_FORCE], sp[SP_PUSH_FORCE + 1], sp[SP_PUSH_FORCE + 2]};
            randomize_force(E, d, sp);                                           // robot 0 got the old force; robot 1 gets this one
            for (int i = 0; i < 3; i++) ex.push[i] = (me == 0 ? f0[i] : sp[SP_PUSH_FORCE + i]) * E.push_ratio;
            randomize_force(E, d, sp);                                           // and a third is drawn for the next substep
          }
        }
        sp[SP_PUSH_COUNT] = (float)c;
      }
      if (ln.lane0()) {
        ptrace[s * 4 + 0] = ex.has_push ? 1.0f : 0.0f;
        for (int i = 0; i < 3; i++) ptrace[s * 4 + 1 + i] = ex.has_push ? ex.push[i] : 0.0f;
      }
      ex.want_touch = s == P.n_sub - 1;
      if (!E.scr_state) K::template substep_impl<true, true, CONE, XROWS>(ln, P, bs, q, qd, tgt, row, s, &ex, &lkh);
    }
    if (PARK) {
      const float keep[4] = {sp[SP_PUSH_COUNT], sp[SP_PUSH_FORCE], sp[SP_PUSH_FORCE + 1], sp[SP_PUSH_FORCE + 2]};
      ln.unpark_row(sp, SEPMC_SP_STRIDE, SEPMC_PARK_AT);
      sp[SP_PUSH_COUNT] = keep[0]; sp[SP_PUSH_FORCE] = keep[1]; sp[SP_PUSH_FORCE + 1] = keep[2]; sp[SP_PUSH_FORCE + 2] = keep[3];
    }
    if (PARK) {
      const int Pd = P.prop_dim;
      for (int c = 0; c < K::OBS_HIST_CHUNKS; c++) hist.h[c] = ln.ld16(orow + Pd, 16 * c, 2 * Pd);
      for (int c = 0; c < 2; c++) hist.ha[c] = ln.ld16(orow + 3L * Pd + 12, 16 * c, 24);
    }
    if (E.scr_state) {   // parity hook: the caller plays PyBullet
      const float* ss = E.scr_state + (long)row * 37;
      bs.p = mk3<float>(ss[0], ss[1], ss[2]);
      bs.q.x = ss[3]; bs.q.y = ss[4]; bs.q.z = ss[5]; bs.q.w = ss[6];
      bs.v = mk3<float>(ss[7], ss[8], ss[9]);
      bs.w = mk3<float>(ss[10], ss[11], ss[12]);
      for (int j = 0; j < 3; j++) { q[j] = ln.ldl(ss, 13 + j, 3); qd[j] = ln.ldl(ss, 25 + j, 3); }
    }
    F fin = q[0] + q[1] + q[2] + qd[0] + qd[1] + qd[2];
    float chk = L::qsum(fin) + bs.p.x + bs.p.y + bs.p.z + bs.q.x + bs.q.y + bs.q.z + bs.q.w + bs.v.x + bs.v.y + bs.v.z + bs.w.x + bs.w.y + bs.w.z;
    const float my_bad = !(fabsf(chk) < 1e30f) ? 1.0f : 0.0f;
    const float my_code = ex.touch_static * 1.0f + ex.touch_flag * 2.0f + ex.touch_robot * 4.0f + my_bad * 8.0f;
    Peer o = exchange(ln, P, bs, q, my_code);
    const bool bad = my_bad > 0.5f || ((int)o.code & 8);
    const float code0 = me == 0 ? my_code : o.code, code1 = me == 0 ? o.code : my_code;
    const Base& b0 = me == 0 ? bs : o.bs;
    const Base& b1 = me == 0 ? o.bs : bs;
    const float pos0[3] = {b0.p.x, b0.p.y, b0.p.z}, pos1[3] = {b1.p.x, b1.p.y, b1.p.z};

    // --- contact bookkeeping: flag hand-over (CTG:579-587), with the flag as the observation sees it kept aside ---
    const float flag_obs[3] = {sp[SP_FLAG], sp[SP_FLAG + 1], sp[SP_FLAG + 2]};
    int who0, who1;
    if (S.scr_on) {
      const int32_t* cl = S.scr_contacts + (long)arena * SEPMC_MAX_CONTACTS * 4;
      who0 = detect_scripted(cl, 3); who1 = detect_scripted(cl, 4);
    } else {
      who0 = detect_classes(code0, 4); who1 = detect_classes(code1, 3);
    }
    const bool wf0 = sp[SP_WITH_FLAG0] > 0.5f;
    const int who_t = wf0 ? who1 : who0;
    bool moved = false;
    if (who_t == 2) {
      sp[SP_WITH_FLAG0] = wf0 ? 0.0f : 1.0f;
      sp[SP_SWITCH] = 1.0f;
      randomize_flag(d, sp, pos0, pos1);
      moved = true;
    } else {
      sp[SP_SWITCH] = 0.0f;
    }
    sp[SP_WHO0] = (float)who0; sp[SP_WHO_T] = (float)who_t;
    // --- CTG:399-419 ---
    const float v0 = sqrtf(__builtin_fmaf(b0.v.x, b0.v.x, b0.v.y * b0.v.y)), v1 = sqrtf(__builtin_fmaf(b1.v.x, b1.v.x, b1.v.y * b1.v.y));   // CTG:370-376 (one stated rounding order: epmc_step.hpp)
    sp[SP_TOTAL_SPD] += v0; sp[SP_TOTAL_SPD + 1] += v1;
    if (v0 > sp[SP_MAX_SPD]) sp[SP_MAX_SPD] = v0;
    if (v1 > sp[SP_MAX_SPD + 1]) sp[SP_MAX_SPD + 1] = v1;
    const int cnt = (int)sp[SP_COUNTER] + 1;
    sp[SP_COUNTER] = (float)cnt;
    int reason = 0;
    if (EP::check_fall(qmat(qnormalize(b0.q)))) reason |= 1;                       // only robot 0's fall ends the episode (CTG:457, :462)
    if (cnt >= E.max_steps) reason |= 2;
    const bool caught = who0 == 4;                                                 // CTG:442-450
    if (caught) reason |= 8;
    if (bad) reason |= 16;
    const float sw = sp[SP_SWITCH];
    const bool now0 = sp[SP_WITH_FLAG0] > 0.5f;
    float reward = ((me == 0) == now0) ? sw : -sw;                                 // CTG:640-652
    if (caught) reward += ((me == 0) == now0) ? 1.0f : -1.0f;                      // CTG:411-419
    if (bad) reward = 0.0f;
    if (!E.scr_draws) sp[SP_STEP_DRAWS] = (float)d.used;
    if (ln.lane0()) {
      float* inf = S.info + (long)row * 4;
      inf[0] = sp[SP_TOTAL_SPD] / (float)cnt; inf[1] = sp[SP_TOTAL_SPD + 1] / (float)cnt; inf[2] = sp[SP_MAX_SPD]; inf[3] = sp[SP_MAX_SPD + 1];
    }

    F oact[3] = {act[0], act[1], act[2]};
    bool fill = false;
    float fobs[3] = {flag_obs[0], flag_obs[1], flag_obs[2]};
    if (reason) {
      if (me == 0) K::count_add(ln, P.counters + 1);
      if (bad && me == 0) K::count_add(ln, P.counters + 2);
      if (P.auto_reset && !PMC_ABL(8192)) {
        const uint32_t episode = (uint32_t)sp[SP_EPISODE] + 1u;
        sp[SP_EPISODE] = (float)episode;
        EpmcDraws dr = {nullptr, 0, 0, P.seed, (uint32_t)arena, episode, 0x5e9a1du};
        float prev[4] = {sp[SP_INIT_ORN], sp[SP_INIT_ORN + 1], sp[SP_INIT_ORN + 2], sp[SP_INIT_ORN + 3]};
        reset_scalars(ln, P, S, row, sp, dr, bs, q, qd, prev);
        o = exchange(ln, P, bs, q, 0.0f);
        for (int j = 0; j < 3; j++) oact[j] = ln.lane_f(0.0f);
        for (int i = 0; i < 3; i++) fobs[i] = sp[SP_FLAG + i];
        fill = true;
        moved = false;
      }
    }
    observe(ln, P, S, row, orow, fill, hist, bs, q, qd, oact, sp, o, fobs);         // the rays still see the flag where it was (CTG:515-577 run before :579)
    if (moved) { put_flag_box(ln, allb, nb, sp + SP_FLAG); }
    store_sp(ln, S.sp + (long)row * SEPMC_SP_STRIDE, sp);
    K::store_state(ln, P.state, N, row, bs, q, qd);
    P.reward[row] = reward;
    P.done[row] = reason ? 1 : 0;
    P.done_reason[row] = (uint8_t)reason;
  }
};
