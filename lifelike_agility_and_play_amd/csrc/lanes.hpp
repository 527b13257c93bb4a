// lanes.hpp -- the "row" execution model of the PMC step kernel.
//
// One environment is stepped by SIXTEEN lanes of a wavefront = one DPP row: lane = (leg, sub), leg = lane / 4 in
// LegOrder FR, FL, HR, HL, sub = lane % 4.  A wave64 carries 4 environments.  The MAX quadruped is a star (base + 4
// independent 3-joint chains):
//   * link-level work is SPLIT over the sub-lanes of a leg: sub-lane k < 3 owns link k + 1 (hip, thigh, shank) -- its sine / cosine, its
//     inertia about the base origin, its bias force, its column of the joint-space inertia -- and composite quantities are suffix sums over
//     the sub-lanes (DPP quad_perm); only what is genuinely per leg (the 3x3 Cholesky factor, Y = M_bl Lm^-T) is replicated four times;
//   * the 28 contact candidates of a leg are split 7 per sub-lane; contact slot s of leg l and its three constraint
//     rows live in the registers of lane (l, s); the limit row of joint j lives in lane (l, j);
//   * whatever couples legs goes through the base as a reduction / broadcast over the 16-lane row (DPP row_ror /
//     row_newbcast), whatever couples the slots of one leg through the 4-lane quad (DPP quad_perm) -- no LDS traffic
//     and no barrier anywhere in the solver.
//
// Values come in two classes:
//   * env-uniform  ("base" values: pose, twist, 6x6 factors, time...)  -- plain float/int/double;
//   * lane-varying (per leg, or per (leg, sub))                        -- L::F / L::I / L::D / L::B.
// The kernel body (pmc_step.hpp) is written once against this interface.  GpuLanes (below) maps it to hardware lanes.
// tests/emul/ instantiates the same source with a 16-wide host type to debug the kernel logic on a machine without
// a GPU; that build is test infrastructure and is never linked into the product library.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LL_HD __host__ __device__ __forceinline__
#define LL_D __device__ __forceinline__
#define LL_NOUNROLL _Pragma("nounroll")
#define LL_UNROLL _Pragma("unroll")
#else
#define LL_HD inline
#define LL_D inline
#define LL_NOUNROLL _Pragma("GCC unroll 1")
#define LL_UNROLL
#endif

namespace lm {
// scalar (env-uniform) overloads of the math vocabulary used by the generic code
LL_HD float sel(bool m, float a, float b) { return m ? a : b; }
LL_HD int sel(bool m, int a, int b) { return m ? a : b; }
LL_HD float sqrt_(float x) { return sqrtf(x); }
LL_HD float rsqrt_(float x) { return 1.0f / sqrtf(x); }
LL_HD float nfma_(float a, float b, float c) { return __builtin_fmaf(-a, b, c); }   // c - a * b as ONE rounding in every build (-ffp-contract=fast decides per build otherwise)
LL_HD float abs_(float x) { return fabsf(x); }
LL_HD float min_(float a, float b) { return fminf(a, b); }
LL_HD float max_(float a, float b) { return fmaxf(a, b); }
LL_HD bool and_(bool a, bool b) { return a && b; }
LL_HD bool or_(bool a, bool b) { return a || b; }
LL_HD bool not_(bool a) { return !a; }
LL_HD float rint_(float x) { return rintf(x); }
LL_HD float atan2_(float y, float x) { return atan2f(y, x); }
LL_HD float sin_(float x) { return sinf(x); }
LL_HD float cos_(float x) { return cosf(x); }
LL_HD float med3_(float x, float lo, float hi) {   // clamp for lo <= hi: one v_med3_f32 on the GPU
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
  return fminf(fmaxf(x, lo), hi);
#endif
}
LL_HD bool odd_(int k) { return (k & 1) != 0; }
LL_HD bool bit1_(int k) { return (k & 2) != 0; }
LL_HD bool eq_(int a, int b) { return a == b; }
}  // namespace lm

#define PMC_ROW 16   // lanes per environment
#ifndef LL_PEER_MODE
#define LL_PEER_MODE 0   // how a chase-tag robot reads its neighbour row: 0 = v_permlane16_swap (shipped); 1, 2: diagnostic forms (GpuLanes::peer)
#endif
#ifndef LL_CONE_PIPE
#define LL_CONE_PIPE 1   // cone turns: a turn's second select rides in the next turn's v_rsq wait state (GpuLanes::cone_turns4); 0: the round-4 turn, the A/B leg
#endif
#ifndef LL_MFMA_GRAM
#define LL_MFMA_GRAM 0   // 1: the Gram blocks of the solver's rows on the matrix cores (GpuLanes::gram16) instead of 96 v_fmac_f32_dpp per block (gram4).  Built and measured in round 5
                         // (profiles/r05_mfma_gram_ab.txt, one box): 344 fewer instructions per substep, and 0.1921 -> 0.1912 ms per control step at 4096 envs (0.5 %), EPMC 0.9 %, SEPMC 1.6 % SLOWER,
                         // the 256-register builds 8 - 10 % slower (the 16-register accumulator tuple costs them scratch; a lone wave does not hide the MFMA chain): off.  Kept as an experiment
#endif
#define CONE_LDS_AT 64        // row-scratch word where a row's cone cross scalars live during the substeps (16 lanes x 32 words; GpuLanes::cone_store)
#define PMC_ROW_SCRATCH 688   // floats of LDS scratch per env row (EPMC: 40 boxes x 8, three ray lists of 10, 16 and 12 records, 64 spare: epmc_step.hpp)

#if defined(__HIPCC__)
// DPP helpers.  ctrl encodings (gfx9 DPP16): quad_perm 0x00-0xFF, row_shr:n 0x110+n, row_ror:n 0x120+n, row_newbcast:n 0x150+n.
#define LL_DPP_MOV(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true))

typedef float ll_f2 __attribute__((ext_vector_type(2)));   // two floats in an aligned register pair: the operand of v_pk_*_f32

struct GpuLanes {
  using F = float;
  using F2 = ll_f2;
  static LL_D F2 pair(F a, F b) { F2 r; r.x = a; r.y = b; return r; }
  using I = int;
  using D = double;
  using B = bool;
  static constexpr int kWave = 64;

  int leg_, sub_, lane16_;
  float* lds_;
  static constexpr int kParamsReload = 0;      // see WithParamsReload below
  template <class T>
  LL_D const T& params(const T& as_passed) const { return as_passed; }
  mutable int cbase_;   // LDS word of the per-leg constant table (+ leg)
  mutable int tbase_;   // LDS word of the candidate table (+ lane16)
  int row_scratch_;     // LDS word where the per-row scratch areas start
  mutable unsigned long long tm_[16];

  static constexpr bool kHoldLink = false;   // pmc_step.hpp own_link: re-read the own-link constants every substep
  static constexpr bool kPrefetchShapes = false;   // see WithShapePrefetch below
  LL_D GpuLanes(float* lds) : leg_((threadIdx.x >> 2) & 3), sub_(threadIdx.x & 3), lane16_(threadIdx.x & 15), lds_(lds), cbase_(0), tbase_(0) {}

  // Stage the constant tables in LDS: legc [n_leg_fields][4] then candc [n_cand_words][16].  A constant then costs one
  // ds_read with an immediate offset instead of a VGPR held across the whole substep loop.
  LL_D void stage_consts(const float* legc, int n_leg_fields, const float* candc, int n_cand_words) {
    const int lane = threadIdx.x & 63;
    const int n1 = n_leg_fields * 4, n2 = n_cand_words * PMC_ROW;
    for (int i = lane; i < n1; i += kWave) lds_[i] = legc[i];
    for (int i = lane; i < n2; i += kWave) lds_[n1 + i] = candc[i];
    cbase_ = leg_;
    tbase_ = n1 + lane16_;
    row_scratch_ = n1 + n2 + 4 * 12;      // after the action stash of the step kernel
    __builtin_amdgcn_s_waitcnt(0);          // single-wave workgroup: program order + waitcnt is enough
  }
  // make the table offsets opaque again so the compiler re-reads constants per substep instead of hoisting them all
  LL_D void refresh_consts() const { asm volatile("" : "+v"(cbase_), "+v"(tbase_)); }
  // top of a control step inside a multi-step launch: nothing computed from the tables in one step may be carried to the next in
  // registers (the compiler would hoist every step-invariant value out of the step loop and spill the state instead)
  LL_D void new_step() { asm volatile("" : "+v"(cbase_), "+v"(tbase_), "+v"(leg_), "+v"(sub_), "+v"(lane16_) : : "memory"); }

  LL_D I leg() const { return leg_; }
  LL_D I sub() const { return sub_; }
  LL_D F legf() const { return (float)leg_; }
  LL_D B is_leg(int l) const { return leg_ == l; }
  LL_D B is_sub(int k) const { return sub_ == k; }
  LL_D B is_lane(int L) const { return lane16_ == L; }
  LL_D F lane_f(float x) const { return x; }
  // per-env scalar code runs on every lane of the row; lane 0 does its stores.  Work lists (rays) are dealt out 16 ways.
  LL_D bool lane0() const { return lane16_ == 0; }
  LL_D int ray_first() const { return lane16_; }
  LL_D int ray_stride() const { return PMC_ROW; }
  static constexpr int kRayChunk = 1;           // (see WithRayChunk)
  // which lanes of this row hold `pred`: bit j = lane j of the row (round 4: the rows' work lists are compacted 16 entries at a time)
  LL_D uint32_t row_ballot(bool pred) const { return (uint32_t)(__ballot(pred) >> (threadIdx.x & 48)) & 0xffffu; }
  // copy n floats (a multiple of 4, 16-byte aligned) of per-env data into the row's LDS scratch and return where they are: the
  // row's 16 lanes then read them at LDS latency instead of issuing a global load each
  LL_D float* row_scratch() const { return lds_ + row_scratch_ + (threadIdx.x >> 4) * PMC_ROW_SCRATCH; }
  LL_D const float* stage_row(const float* g, int n) const {
    float* dst = row_scratch();
    for (int i = lane16_ * 4; i < n; i += PMC_ROW * 4) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(g + i);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    return dst;
  }
  // ---- the cone round's cross scalars in the row scratch (WithConeInLds: the 256-register builds have no room for their 32 registers) ----
  // word CONE_LDS_AT + ((kind * 4 + S) * 16 + lane) * 4 + t holds scalar [4 t + S] of kind (0: ConeX::n12, 1: n21) of this lane: one ds_write_b128 /
  // ds_read_b128 per (kind, turn block S), consecutive lanes 16 B apart.  The words are idle during the substeps: in front of them the (at most eight)
  // near-box records the contact search reads, behind them the ray lists of the observation phase and the parked episode scalars (epmc_step.hpp).
  static constexpr bool kConeInLds = false;
  LL_D void cone_store(int kind, int S, F a, F b, F c, F d) const {
    *reinterpret_cast<float4*>(row_scratch() + CONE_LDS_AT + ((kind * 4 + S) * PMC_ROW + lane16_) * 4) = make_float4(a, b, c, d);
  }
  LL_D void cone_load(int kind, int S, F& a, F& b, F& c, F& d) const {
    const float4 v = *reinterpret_cast<const float4*>(row_scratch() + CONE_LDS_AT + ((kind * 4 + S) * PMC_ROW + lane16_) * 4);
    a = v.x; b = v.y; c = v.z; d = v.w;
  }
  LL_D void row_sync() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }   // lane 0's stores visible to the row
  // Park n row-uniform floats in the row's LDS scratch (word `at`) and take them back later: between the two calls their registers are
  // free -- the larger-batch EPMC / SEPMC builds carry 40 episode scalars through the substep loop otherwise and spill state around them.
  LL_D void park_row(const float* v, int n, int at) const {
    float* dst = row_scratch() + at;
    if (lane0()) for (int i = 0; i < n; i++) dst[i] = v[i];
    row_sync();
    asm volatile("" : : : "memory");                          // the loads of unpark_row must not be forwarded from these stores
  }
  LL_D void unpark_row(float* v, int n, int at) const {
    asm volatile("" : : : "memory");
    const float* src = row_scratch() + at;                    // (one wave: its LDS operations execute in program order)
    for (int i = 0; i < n; i++) v[i] = src[i];
  }

  // ---- reductions / broadcasts ------------------------------------------------------------------------------
  // sum over the four LEGS of a leg-uniform value (each leg's value is replicated in its 4 sub-lanes)
  static LL_D float qsum(F x) {
    float y = x + LL_DPP_MOV(x, 0x128);      // row_ror:8
    return y + LL_DPP_MOV(y, 0x124);         // row_ror:4
  }
  // sum over the four SUB-lanes of a leg (quad): result is leg-uniform
  static LL_D F subsum(F x) {
    float y = x + LL_DPP_MOV(x, 0xB1);       // quad_perm [1,0,3,2]
    return y + LL_DPP_MOV(y, 0x4E);          // quad_perm [2,3,0,1]
  }
  // value of x held by the same sub-lane of the PREVIOUS leg ((leg + 3) % 4) / of the leg two away (row_ror: data moves to
  // higher lanes, lane i receives lane (i - n) mod 16)
  static LL_D F from_prev_leg(F x) { return LL_DPP_MOV(x, 0x124); }
  static LL_D F from_leg2(F x) { return LL_DPP_MOV(x, 0x128); }
  static LL_D F from_next_leg(F x) { return LL_DPP_MOV(x, 0x12C); }   // row_ror:12: the same sub-lane of leg (leg + 1) % 4
  // value of x held by the same lane of the NEIGHBOURING row (row ^ 1 of the wave): the other robot of a SEPMC arena.
  // gfx950 v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second.
  // (LL_PEER_MODE, diagnostic builds only -- tools/diag_sepmc_builds.py: 1 = the same exchange through ds_bpermute_b32, 2 = the swap fenced by wait states)
  LL_D F peer(F x) const {
    unsigned u = __float_as_uint(x);
#if LL_PEER_MODE == 1
    return __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x ^ 16u) & 63u) << 2), (int)u));
#else
#if LL_PEER_MODE == 2
    asm volatile("s_nop 7" : "+v"(u));
#endif
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    unsigned o = (threadIdx.x & 16) ? r[0] : r[1];
#if LL_PEER_MODE == 2
    asm volatile("s_nop 7" : "+v"(o));
#endif
    return __uint_as_float(o);
#endif
  }
  LL_D float peer_u(float x) const { return peer(x); }
  // minimum over the 16 lanes of the row
  static LL_D float rmin(F x) {
    float y = fminf(x, LL_DPP_MOV(x, 0xB1));
    y = fminf(y, LL_DPP_MOV(y, 0x4E));
    y = fminf(y, LL_DPP_MOV(y, 0x124));
    return fminf(y, LL_DPP_MOV(y, 0x128));
  }
  static LL_D F submin(F x) {
    float y = fminf(x, LL_DPP_MOV(x, 0xB1));
    return fminf(y, LL_DPP_MOV(y, 0x4E));
  }
  // six row sums (all 16 lanes) at once; the DPP adds are interleaved so that no DPP read follows its producer by less
  // than two instructions (the compiler's own sequence pays an s_nop per DPP op)
#define LL_STEP6(CTRL)                                                                         \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
    "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
    "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  static LL_D void rsum6(const F* x, float* out) {
    float q0 = x[0], q1 = x[1], q2 = x[2], q3 = x[3], q4 = x[4], q5 = x[5];
    asm("s_nop 1\n\t" LL_STEP6("quad_perm:[1,0,3,2]") LL_STEP6("quad_perm:[2,3,0,1]") LL_STEP6("row_ror:4") LL_STEP6("row_ror:8") "s_nop 0"
        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5));
    out[0] = q0; out[1] = q1; out[2] = q2; out[3] = q3; out[4] = q4; out[5] = q5;
  }
  // six sums over the LEGS of leg-uniform values
  static LL_D void qsum6(const F* x, float* out) {
    float q0 = x[0], q1 = x[1], q2 = x[2], q3 = x[3], q4 = x[4], q5 = x[5];
    asm("s_nop 1\n\t" LL_STEP6("row_ror:8") LL_STEP6("row_ror:4") "s_nop 0" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5));
    out[0] = q0; out[1] = q1; out[2] = q2; out[3] = q3; out[4] = q4; out[5] = q5;
  }
#undef LL_STEP6
  // three sums over the sub-lanes of a leg (quad)
  static LL_D void subsum3(const F* x, F* out) {
    float q0 = x[0], q1 = x[1], q2 = x[2];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0"
        : "+v"(q0), "+v"(q1), "+v"(q2));
    out[0] = q0; out[1] = q1; out[2] = q2;
  }
  // ---- work split over the sub-lanes of a leg by LINK (pmc_step.hpp: sub-lane k < 3 owns link k + 1, sub-lane 3 a link of zero mass) ------
  // suffix sums over the sub-lanes: x[k] <- x[k] + x[k+1] + x[k+2]  (sub-lane 3 must hold zero), N values at once with the DPP adds
  // interleaved so that no DPP read follows its producer by less than two instructions
#define LL_SUF(I_, P_) "v_add_f32_dpp %" #I_ ", %" #I_ ", %" #I_ " quad_perm:" P_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  static LL_D void sufsum6(F* x) {
    asm("s_nop 1\n\t"
        LL_SUF(0, "[1,2,3,3]") LL_SUF(1, "[1,2,3,3]") LL_SUF(2, "[1,2,3,3]") LL_SUF(3, "[1,2,3,3]") LL_SUF(4, "[1,2,3,3]") LL_SUF(5, "[1,2,3,3]")
        LL_SUF(0, "[2,3,3,3]") LL_SUF(1, "[2,3,3,3]") LL_SUF(2, "[2,3,3,3]") LL_SUF(3, "[2,3,3,3]") LL_SUF(4, "[2,3,3,3]") LL_SUF(5, "[2,3,3,3]")
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
  }
  static LL_D void sufsum4(F* x) {
    asm("s_nop 1\n\t"
        LL_SUF(0, "[1,2,3,3]") LL_SUF(1, "[1,2,3,3]") LL_SUF(2, "[1,2,3,3]") LL_SUF(3, "[1,2,3,3]")
        LL_SUF(0, "[2,3,3,3]") LL_SUF(1, "[2,3,3,3]") LL_SUF(2, "[2,3,3,3]") LL_SUF(3, "[2,3,3,3]")
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
  }
#undef LL_SUF
  // out[i] = x[i] of sub-lane K of the own leg, six / three values at once (plain DPP moves; the leading wait states cover an x written
  // just before the call)
#define LL_BC(D_, S_, P_) "v_mov_b32_dpp %" #D_ ", %" #S_ " quad_perm:" P_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LL_BC6(P_)                                                                                                       \
    asm("s_nop 1\n\t" LL_BC(0, 6, P_) LL_BC(1, 7, P_) LL_BC(2, 8, P_) LL_BC(3, 9, P_) LL_BC(4, 10, P_) LL_BC(5, 11, P_)  \
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5])                      \
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]))
  template <int K_>
  static LL_D void subbcast6(const F* x, F* out) {
    if (K_ == 0) LL_BC6("[0,0,0,0]"); else if (K_ == 1) LL_BC6("[1,1,1,1]"); else if (K_ == 2) LL_BC6("[2,2,2,2]"); else LL_BC6("[3,3,3,3]");
  }
  // the same, directly behind another subbcast6 of the SAME x (`after`: one of that call's outputs, which orders the two): x has settled, no wait states (round 6)
#define LL_BC6S(P_)                                                                                                      \
    asm(LL_BC(0, 6, P_) LL_BC(1, 7, P_) LL_BC(2, 8, P_) LL_BC(3, 9, P_) LL_BC(4, 10, P_) LL_BC(5, 11, P_)                \
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5])                      \
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(after))
  template <int K_>
  static LL_D void subbcast6_after(const F* x, F* out, F after) {
    if (K_ == 0) LL_BC6S("[0,0,0,0]"); else if (K_ == 1) LL_BC6S("[1,1,1,1]"); else if (K_ == 2) LL_BC6S("[2,2,2,2]"); else LL_BC6S("[3,3,3,3]");
  }
#undef LL_BC6S
  // the lower triangle of a 3 x 3 matrix whose column k lives in sub-lane k as d[0..2] (row index): m11 m12 m22 m13 m23 m33 to every sub-lane
  static LL_D void gather_tri3(const F* d, F* m) {
    asm("s_nop 1\n\t" LL_BC(0, 6, "[0,0,0,0]") LL_BC(1, 6, "[1,1,1,1]") LL_BC(2, 7, "[1,1,1,1]") LL_BC(3, 6, "[2,2,2,2]") LL_BC(4, 7, "[2,2,2,2]") LL_BC(5, 8, "[2,2,2,2]")
        : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5])
        : "v"(d[0]), "v"(d[1]), "v"(d[2]));
  }
  // x of sub-lane 0, 1, 2 of the own leg
  static LL_D void spread3(F x, F* out) {
    asm("s_nop 1\n\t" LL_BC(0, 3, "[0,0,0,0]") LL_BC(1, 3, "[1,1,1,1]") LL_BC(2, 3, "[2,2,2,2]")
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]) : "v"(x));
  }
#undef LL_BC6
#undef LL_BC
  // value of lane L of the row (env-uniform), of sub-lane K of the own leg (leg-uniform)
  template <int L_>
  static LL_D float rbcast(F x) { return LL_DPP_MOV(x, 0x150 + L_); }
  template <int K_>
  static LL_D F subbcast(F x) { return LL_DPP_MOV(x, K_ | (K_ << 2) | (K_ << 4) | (K_ << 6)); }
  // value of leg l (taken from its sub-lane 0)
  template <int LEG_>
  static LL_D float bcast(F x) { return LL_DPP_MOV(x, 0x150 + 4 * LEG_); }
  // acc += rbcast<L>(x) * k as ONE instruction (VOP2 v_fmac_f32 with a DPP source); a VALU write followed by a DPP read of
  // the same VGPR needs 2 wait states, hence the leading s_nop 1
  template <int L_>
  static LL_D void fmac_rbcast(F& acc, F x, F k);
  // same without the wait states: only when x was NOT written by the two preceding instructions
  template <int L_>
  static LL_D void fmac_rbcast_settled(F& acc, F x, F k);
  // returns x after two wait states, so that following DPP reads of the result are hazard free
  static LL_D F settle(F x) { asm volatile("s_nop 1" : "+v"(x)); return x; }
  // g[4t + S_] += sum_i y[i] * (x[i] of lane 4t + S_), t = 0..3: the Gram scalars of a row against the four rows of turn block S_,
  // 24 v_fmac_f32 with a DPP row-broadcast source.  x and y are not written inside the block, so after the leading wait states no
  // DPP read-after-write hazard can occur.
  // (round 6) Blocks S_ = 1 .. 3 of one Gram chain follow block 0 directly: x has not been written since, so their two leading wait states -- s_nop 1 costs a lone wave
  // 8.2 cycles, profiles/r06_issue_probe.txt -- are dropped.  They take block S_ - 1's first scalar as an (unread) operand: the chain stays in order behind block 0's wait states.
  template <int S_>
  static LL_D void gram4(const F* x, const F* y, F* g) {
    float g0 = g[S_], g1 = g[4 + S_], g2 = g[8 + S_], g3 = g[12 + S_];
    const float after = S_ > 0 ? g[S_ - 1] : x[0];
#define LL_G1(O, L_, X, Y) "v_fmac_f32_dpp " O ", " X ", " Y " row_newbcast:" L_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LL_G6(O, L_) LL_G1(O, L_, "%4", "%10") LL_G1(O, L_, "%5", "%11") LL_G1(O, L_, "%6", "%12") LL_G1(O, L_, "%7", "%13") LL_G1(O, L_, "%8", "%14") LL_G1(O, L_, "%9", "%15")
#define LL_G24(NOP_, A_, B_, C_, D_)                                                                                                       \
    asm(NOP_ LL_G6("%0", A_) LL_G6("%1", B_) LL_G6("%2", C_) LL_G6("%3", D_)                                                             \
        : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3)                                                                                       \
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(after))
    if (S_ == 0) LL_G24("s_nop 1\n\t", "0", "4", "8", "12");
    else if (S_ == 1) LL_G24("", "1", "5", "9", "13");
    else if (S_ == 2) LL_G24("", "2", "6", "10", "14");
    else LL_G24("", "3", "7", "11", "15");
#undef LL_G24
#undef LL_G6
#undef LL_G1
    g[S_] = g0; g[4 + S_] = g1; g[8 + S_] = g2; g[12 + S_] = g3;
  }
  // The whole 16 x 16 Gram block of a round on the MATRIX cores (round 5):  g[L] = sum_i y[i] * (x[i] of lane L of my row), L = 0 .. 15, for the four env rows
  // of the wavefront at once.  v_mfma_f32_16x16x1_4b_f32 is four independent 16 x 16 outer products -- exactly the four env rows: A = x (lane i of a row holds
  // x_i[k]), B = y (lane j holds y_j[k]), six of them chained over the six base coefficients.  D_b[i][j] lands in lane 16 (i / 4) + j, register 4 b + i % 4:
  // lane j of row group g holds ITS OWN entries g_{b,j}[4 g .. 4 g + 3], only in the wrong row group -- a 4 x 4 block transpose between register block and row
  // group (8 v_permlane32_swap + 8 v_permlane16_swap, gfx950) brings them home, register index = column index.  6 MFMA + 16 swaps instead of 96 half-rate
  // v_fmac_f32_dpp (gram4 x 4); exact float32 (tools/mfma_gram_probe.hip: bit-identical to the shuffle statement on MI355X).  MFMA ignores EXEC; the swaps fetch from lanes outside
  // EXEC too (fi = 1: a wave whose last rows hold no env still owns the registers the matrix core wrote there -- with fi = 0 the first version read zeros and failed every test with a
  // partial wave).  EXPERIMENT, off by default (LL_MFMA_GRAM above): the fi = 1 form has not been through the GPU suite.
  static constexpr bool kGram16 = LL_MFMA_GRAM != 0;
  typedef float ll_f16v __attribute__((ext_vector_type(16)));
  static LL_D void gram16(const F* x, const F* y, F* g) {
    ll_f16v acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    for (int i = 0; i < 6; i++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(x[i], y[i], acc, 0, 0, 0);
    unsigned R[16];
    for (int i = 0; i < 16; i++) R[i] = __float_as_uint(acc[i]);
    for (int r = 0; r < 4; r++) {
      const auto a = __builtin_amdgcn_permlane32_swap(R[r], R[8 + r], true, false); R[r] = a[0]; R[8 + r] = a[1];
      const auto b = __builtin_amdgcn_permlane32_swap(R[4 + r], R[12 + r], true, false); R[4 + r] = b[0]; R[12 + r] = b[1];
    }
    for (int r = 0; r < 4; r++) {
      const auto a = __builtin_amdgcn_permlane16_swap(R[r], R[4 + r], true, false); R[r] = a[0]; R[4 + r] = a[1];
      const auto b = __builtin_amdgcn_permlane16_swap(R[8 + r], R[12 + r], true, false); R[8 + r] = b[0]; R[12 + r] = b[1];
    }
    for (int i = 0; i < 16; i++) g[i] = __uint_as_float(R[i]);
  }
  // ---- the Gram blocks as a BACKGROUND job of the matrix cores (round 6; WithGramPipe below) -------------------------------------------------------------------
  // Round 5's gram16 issued its six dependent MFMAs back to back and read the result at once: a wave issues in order, so it stood still for the whole chain
  // (six 8-pass instructions, 32 cycles each, plus the result latency) -- 0.5 % gained of the 5 % the removed instructions were worth.  Here the chain is a job the caller
  // feeds one MFMA at a time (gram_mfma<K>, K = 0 .. 5) between pieces of independent VALU work -- the next row's coefficients -- and collects much later
  // (gram_collect: the 16-swap block transpose of gram16), when the matrix cores have long finished.  Each step is fenced by scheduling barriers: the compiler
  // keeps the work between two steps between them.  Same arithmetic as gram16 (exact float32, accumulation order k = 0 .. 5 from zero).
  static constexpr bool kGramPipe = false;
  struct GramAcc { ll_f16v v; };
  template <int K_>
  static LL_D void gram_mfma(GramAcc& a, F x, F y) {
    __builtin_amdgcn_sched_barrier(0);
    if (K_ == 0) {
      ll_f16v z;
      for (int i = 0; i < 16; i++) z[i] = 0.0f;
      a.v = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, z, 0, 0, 0);
    } else {
      a.v = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a.v, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // g[L] = sum_k y[k] * (x[k] of lane L of my row): the finished accumulator's 4 x 4 block transpose between register block and row group (see gram16)
  static LL_D void gram_collect(const GramAcc& a, F* g) {
    unsigned R[16];
    for (int i = 0; i < 16; i++) R[i] = __float_as_uint(a.v[i]);
    for (int r = 0; r < 4; r++) {
      const auto p = __builtin_amdgcn_permlane32_swap(R[r], R[8 + r], true, false); R[r] = p[0]; R[8 + r] = p[1];
      const auto q = __builtin_amdgcn_permlane32_swap(R[4 + r], R[12 + r], true, false); R[4 + r] = q[0]; R[12 + r] = q[1];
    }
    for (int r = 0; r < 4; r++) {
      const auto p = __builtin_amdgcn_permlane16_swap(R[r], R[4 + r], true, false); R[r] = p[0]; R[4 + r] = p[1];
      const auto q = __builtin_amdgcn_permlane16_swap(R[8 + r], R[12 + r], true, false); R[8 + r] = q[0]; R[12 + r] = q[1];
    }
    for (int i = 0; i < 16; i++) g[i] = __uint_as_float(R[i]);
  }
  // Four Gauss-Seidel turns (lanes S, 4+S, 8+S, 12+S) as one block: v_med3 (clamp the pending increment), v_cndmask (the lane
  // whose turn it is keeps its increment; masks m0..m3 are the lane masks of the four turns), one wait state, v_fmac with a
  // DPP row broadcast (every lane's pending increment moves by nk * d).  4 issue slots per turn.
  template <int S_, bool NEG_LO = false>   // NEG_LO: the lower bound is -lo (a unilateral row passes its multiplier and saves the subtraction)
  LL_D void turns4(F& u, F& dl, F lo, F hi, F k0, F k1, F k2, F k3) const {
    const unsigned long long m0 = tm_[S_], m1 = tm_[4 + S_], m2 = tm_[8 + S_], m3 = tm_[12 + S_];
    float d;
    if (NEG_LO) lo = -lo;
#define LL_T1(K, M, L_)                                                                          \
    "v_med3_f32 %2, %0, %3, %4\n\t"                                                            \
    "v_cndmask_b32_e64 %1, %1, %2, " M "\n\t"                                                  \
    "s_nop 0\n\t"                                                                              \
    "v_fmac_f32_dpp %0, %2, " K " row_newbcast:" L_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    if (S_ == 0)      asm(LL_T1("%5", "%9", "0") LL_T1("%6", "%10", "4") LL_T1("%7", "%11", "8") LL_T1("%8", "%12", "12") : "+v"(u), "+v"(dl), "=&v"(d) : "v"(lo), "v"(hi), "v"(k0), "v"(k1), "v"(k2), "v"(k3), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
    else if (S_ == 1) asm(LL_T1("%5", "%9", "1") LL_T1("%6", "%10", "5") LL_T1("%7", "%11", "9") LL_T1("%8", "%12", "13") : "+v"(u), "+v"(dl), "=&v"(d) : "v"(lo), "v"(hi), "v"(k0), "v"(k1), "v"(k2), "v"(k3), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
    else if (S_ == 2) asm(LL_T1("%5", "%9", "2") LL_T1("%6", "%10", "6") LL_T1("%7", "%11", "10") LL_T1("%8", "%12", "14") : "+v"(u), "+v"(dl), "=&v"(d) : "v"(lo), "v"(hi), "v"(k0), "v"(k1), "v"(k2), "v"(k3), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
    else              asm(LL_T1("%5", "%9", "3") LL_T1("%6", "%10", "7") LL_T1("%7", "%11", "11") LL_T1("%8", "%12", "15") : "+v"(u), "+v"(dl), "=&v"(d) : "v"(lo), "v"(hi), "v"(k0), "v"(k1), "v"(k2), "v"(k3), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
#undef LL_T1
  }
  // Eight Gauss-Seidel turns as ONE block (H_ = 0: lanes 0,4,8,12, 1,5,9,13;  H_ = 1: lanes 2,6,10,14, 3,7,11,15): between two
  // separate asm statements the compiler puts a wait state of its own.
  template <int H_, bool NEG_LO = false>   // NEG_LO: the lower bound is -lo, taken with the instruction's own source negation
  LL_D void turns8(F& u, F& dl, F lo, F hi, const F* nk) const {
    constexpr int A = 2 * H_, B = 2 * H_ + 1;
    const unsigned long long m0 = tm_[A], m1 = tm_[4 + A], m2 = tm_[8 + A], m3 = tm_[12 + A], m4 = tm_[B], m5 = tm_[4 + B], m6 = tm_[8 + B], m7 = tm_[12 + B];
    float d;
#define LL_T1(LO, K, M, L_)                                                                      \
    "v_med3_f32 %2, %0, " LO ", %4\n\t"                                                        \
    "v_cndmask_b32_e64 %1, %1, %2, " M "\n\t"                                                  \
    "s_nop 0\n\t"                                                                              \
    "v_fmac_f32_dpp %0, %2, " K " row_newbcast:" L_ " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LL_T8(LO, L0, L1, L2, L3, L4, L5, L6, L7)                                                                                        \
      asm(LL_T1(LO, "%5", "%13", L0) LL_T1(LO, "%6", "%14", L1) LL_T1(LO, "%7", "%15", L2) LL_T1(LO, "%8", "%16", L3)                    \
          LL_T1(LO, "%9", "%17", L4) LL_T1(LO, "%10", "%18", L5) LL_T1(LO, "%11", "%19", L6) LL_T1(LO, "%12", "%20", L7)                 \
          : "+v"(u), "+v"(dl), "=&v"(d)                                                                                                  \
          : "v"(lo), "v"(hi), "v"(nk[A]), "v"(nk[4 + A]), "v"(nk[8 + A]), "v"(nk[12 + A]), "v"(nk[B]), "v"(nk[4 + B]), "v"(nk[8 + B]), "v"(nk[12 + B]), \
            "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "s"(m6), "s"(m7))
    if (H_ == 0 && !NEG_LO) LL_T8("%3", "0", "4", "8", "12", "1", "5", "9", "13");
    else if (H_ == 0) LL_T8("-%3", "0", "4", "8", "12", "1", "5", "9", "13");
    else if (!NEG_LO) LL_T8("%3", "2", "6", "10", "14", "3", "7", "11", "15");
    else LL_T8("-%3", "2", "6", "10", "14", "3", "7", "11", "15");
#undef LL_T8
#undef LL_T1
  }

  // Four CONE-COUPLED friction turns (lanes S, 4+S, 8+S, 12+S) as one block (pmc_step.hpp gs_cone_round).  The pair of friction rows of a lane
  // carries S = lambda + pending increment ("where the multiplier would go"); a turn scales the pair back onto the cone |(S1, S2)| <= lim
  // with sc = clamp(lim * rsq(S1^2 + S2^2), 0, 1) -- v_mul_legacy (0 * inf = 0: a row without normal force lands on 0 whatever the length,
  // and a pair of length zero under a positive bound on sc = 1) with the instruction's own clamp --, the lane whose turn it is keeps
  // e = S * sc - lambda for both rows, and every lane's S moves by the 2 x 2 coupling (k11 k12; k21 k22) of that lane's pair with its own.
  // 12 VALU instructions and one wait state (v_rsq result into a non-transcendental instruction) per turn; the DPP reads of e1 / e2 come
  // three and more instructions after their producers.  30 asm operands: the limit.
  template <int S_>
  LL_D void cone_turns4(F& S1, F& S2, F& d1, F& d2, F lam1, F lam2, F lim, const F* k11, const F* k12, const F* k21, const F* k22) const {
    const unsigned long long m0 = tm_[S_], m1 = tm_[4 + S_], m2 = tm_[8 + S_], m3 = tm_[12 + S_];
    float t, e1, e2;
#define LL_D1(X) " row_newbcast:" X " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LL_C1(K11, K12, K21, K22, M, L_)                                                           \
    "v_mul_f32_e32 %4, %1, %1\n\t"                                                               \
    "v_fmac_f32_e32 %4, %0, %0\n\t"                                                              \
    "v_rsq_f32_e32 %4, %4\n\t"                                                                   \
    "s_nop 0\n\t"                                                                                \
    "v_mul_legacy_f32_e64 %4, %9, %4 clamp\n\t"                                                  \
    "v_fma_f32 %5, %0, %4, -%7\n\t"                                                              \
    "v_fma_f32 %6, %1, %4, -%8\n\t"                                                              \
    "v_cndmask_b32_e64 %2, %2, %5, " M "\n\t"                                                    \
    "v_cndmask_b32_e64 %3, %3, %6, " M "\n\t"                                                    \
    "v_fmac_f32_dpp %0, %5, " K11 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %1, %5, " K21 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %0, %6, " K12 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %1, %6, " K22 LL_D1(L_)
#if LL_CONE_PIPE
    // (round 5) inside a block the second select of a turn (d2 <- e2 under the turn's mask) waits in the NEXT turn's wait state behind v_rsq instead of an s_nop: e2 is not
    // rewritten before that turn's second v_fma, the select of d2 is read by nobody inside the block.  3 issue slots less per block of four turns.
#define LL_CF(K11, K12, K21, K22, M, L_)      /* first turn of a block: its own d2 select is left to the next turn */                          \
    "v_mul_f32_e32 %4, %1, %1\n\t"                                                               \
    "v_fmac_f32_e32 %4, %0, %0\n\t"                                                              \
    "v_rsq_f32_e32 %4, %4\n\t"                                                                   \
    "s_nop 0\n\t"                                                                                \
    "v_mul_legacy_f32_e64 %4, %9, %4 clamp\n\t"                                                  \
    "v_fma_f32 %5, %0, %4, -%7\n\t"                                                              \
    "v_fma_f32 %6, %1, %4, -%8\n\t"                                                              \
    "v_cndmask_b32_e64 %2, %2, %5, " M "\n\t"                                                    \
    "v_fmac_f32_dpp %0, %5, " K11 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %1, %5, " K21 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %0, %6, " K12 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %1, %6, " K22 LL_D1(L_)
#define LL_CN(K11, K12, K21, K22, M, MP, L_)  /* later turns: the previous turn's d2 select (mask MP) in the wait state */                   \
    "v_mul_f32_e32 %4, %1, %1\n\t"                                                               \
    "v_fmac_f32_e32 %4, %0, %0\n\t"                                                              \
    "v_rsq_f32_e32 %4, %4\n\t"                                                                   \
    "v_cndmask_b32_e64 %3, %3, %6, " MP "\n\t"                                                   \
    "v_mul_legacy_f32_e64 %4, %9, %4 clamp\n\t"                                                  \
    "v_fma_f32 %5, %0, %4, -%7\n\t"                                                              \
    "v_fma_f32 %6, %1, %4, -%8\n\t"                                                              \
    "v_cndmask_b32_e64 %2, %2, %5, " M "\n\t"                                                    \
    "v_fmac_f32_dpp %0, %5, " K11 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %1, %5, " K21 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %0, %6, " K12 LL_D1(L_)                                                      \
    "v_fmac_f32_dpp %1, %6, " K22 LL_D1(L_)
#define LL_C4(L0, L1, L2, L3)                                                                                                            \
    asm(LL_CF("%10", "%14", "%18", "%22", "%26", L0) LL_CN("%11", "%15", "%19", "%23", "%27", "%26", L1)                                 \
        LL_CN("%12", "%16", "%20", "%24", "%28", "%27", L2) LL_CN("%13", "%17", "%21", "%25", "%29", "%28", L3)                          \
        "v_cndmask_b32_e64 %3, %3, %6, %29\n\t"                                                                                        \
        : "+v"(S1), "+v"(S2), "+v"(d1), "+v"(d2), "=&v"(t), "=&v"(e1), "=&v"(e2)                                                         \
        : "v"(lam1), "v"(lam2), "v"(lim),                                                                                                \
          "v"(k11[S_]), "v"(k11[4 + S_]), "v"(k11[8 + S_]), "v"(k11[12 + S_]), "v"(k12[S_]), "v"(k12[4 + S_]), "v"(k12[8 + S_]), "v"(k12[12 + S_]), \
          "v"(k21[S_]), "v"(k21[4 + S_]), "v"(k21[8 + S_]), "v"(k21[12 + S_]), "v"(k22[S_]), "v"(k22[4 + S_]), "v"(k22[8 + S_]), "v"(k22[12 + S_]), \
          "s"(m0), "s"(m1), "s"(m2), "s"(m3))
#else
#define LL_C4(L0, L1, L2, L3)                                                                                                            \
    asm(LL_C1("%10", "%14", "%18", "%22", "%26", L0) LL_C1("%11", "%15", "%19", "%23", "%27", L1)                                        \
        LL_C1("%12", "%16", "%20", "%24", "%28", L2) LL_C1("%13", "%17", "%21", "%25", "%29", L3)                                        \
        : "+v"(S1), "+v"(S2), "+v"(d1), "+v"(d2), "=&v"(t), "=&v"(e1), "=&v"(e2)                                                         \
        : "v"(lam1), "v"(lam2), "v"(lim),                                                                                                \
          "v"(k11[S_]), "v"(k11[4 + S_]), "v"(k11[8 + S_]), "v"(k11[12 + S_]), "v"(k12[S_]), "v"(k12[4 + S_]), "v"(k12[8 + S_]), "v"(k12[12 + S_]), \
          "v"(k21[S_]), "v"(k21[4 + S_]), "v"(k21[8 + S_]), "v"(k21[12 + S_]), "v"(k22[S_]), "v"(k22[4 + S_]), "v"(k22[8 + S_]), "v"(k22[12 + S_]), \
          "s"(m0), "s"(m1), "s"(m2), "s"(m3))
#endif
    if (S_ == 0) LL_C4("0", "4", "8", "12");
    else if (S_ == 1) LL_C4("1", "5", "9", "13");
    else if (S_ == 2) LL_C4("2", "6", "10", "14");
    else LL_C4("3", "7", "11", "15");
#undef LL_C4
#undef LL_C1
#undef LL_D1
#undef LL_CF
#undef LL_CN
  }

  // ---- the solver's velocity state, scattered over the sub-lanes (pmc_step.hpp gs_round) ------------------------------------------
  // Per env the projected Gauss-Seidel sweep carries the whitened base twist dx[6] and, per leg, the whitened joint rates dq[3].
  // They live in THREE registers:   VA: lane (leg, s) holds dx[s]      VB: dx[4 + (s & 1)]      VJ: dq_leg[s] (s = 3: zero)
  // and a row keeps its nine coefficients pre-permuted for its own lane (row_permute in pmc_step.hpp):
  //   ca[k] = gt[s ^ k],   cb[k] = gt[4 + ((s ^ k) & 1)],   cj[k] = jt[s ^ k] (index 3: zero)          k = 0..3 / 0..1 / 0..3
  // vel_dot:    c + gt . dx + jt . dq = c + sum_k ca[k] * VA[s ^ k] + ...: ten multiply-adds, seven of them with a quad_perm source;
  //             the three plain ones come first, so a register written just before the call has settled when DPP reads it.
  LL_D static F vel_dot(F c, F2 ca01, F2 ca23, F2 cb01, F2 cj01, F2 cj23, F VA, F VB, F VJ) {
    float w;
#define LL_Q(X) " quad_perm:" X " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("v_fma_f32 %0, %1, %11, %14\n\t"
        "v_fmac_f32_e32 %0, %5, %12\n\t"
        "v_fmac_f32_e32 %0, %7, %13\n\t"
        "v_fmac_f32_dpp %0, %11, %2" LL_Q("[1,0,3,2]")
        "v_fmac_f32_dpp %0, %11, %3" LL_Q("[2,3,0,1]")
        "v_fmac_f32_dpp %0, %11, %4" LL_Q("[3,2,1,0]")
        "v_fmac_f32_dpp %0, %12, %6" LL_Q("[1,0,3,2]")
        "v_fmac_f32_dpp %0, %13, %8" LL_Q("[1,0,3,2]")
        "v_fmac_f32_dpp %0, %13, %9" LL_Q("[2,3,0,1]")
        "v_fmac_f32_dpp %0, %13, %10" LL_Q("[3,2,1,0]")
        : "=&v"(w)
        : "v"(ca01.x), "v"(ca01.y), "v"(ca23.x), "v"(ca23.y), "v"(cb01.x), "v"(cb01.y), "v"(cj01.x), "v"(cj01.y), "v"(cj23.x), "v"(cj23.y), "v"(VA), "v"(VB),
          "v"(VJ), "v"(c));
    return w;
  }
  // vel_commit: every lane has committed the multiplier increment dl of its row;  dx += sum over the 16 lanes of gt * dl,
  // dq += sum over the leg's 4 lanes of jt * dl.  A transpose-reduce: because lane s multiplies dl with the coefficient of value
  // s ^ k, "own product + partner's product" lands value s in lane s after two quad exchanges -- 4 multiplies and 3 adds reduce four
  // values over a quad (an all-reduce takes 4 and 8), and the result is already laid out as VA / VB / VJ.  The ten products are five
  // v_pk_mul_f32 (the coefficients live in register pairs for this); 20 instructions; every DPP read is at least three instructions
  // behind its producer, so the block has no wait states.
  // (lam += dl rides along as the block's first instruction: with the s_nop behind it, the products -- computed by the compiler's own
  //  v_pk_mul_f32 just before the block -- have settled when the first DPP read comes, wherever the scheduler put them)
  LL_D static void vel_commit(F dl, F& lam, F2 ca01, F2 ca23, F2 cb01, F2 cj01, F2 cj23, F& VA, F& VB, F& VJ) {
    const F2 P01 = ca01 * dl, P23 = ca23 * dl, Q01 = cb01 * dl, J01 = cj01 * dl, J23 = cj23 * dl;
    float p0 = P01.x, p1 = P01.y, p2 = P23.x, p3 = P23.y, q0 = Q01.x, q1 = Q01.y, j0 = J01.x, j1 = J01.y, j2 = J23.x, j3 = J23.y;
#define LL_R(X) " " X " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("v_add_f32_e32 %8, %8, %14\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %3, %9, %3" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %4, %10, %4" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %5, %11, %5" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %6, %12, %6" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %7, %13, %7" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %3, %4, %3" LL_R("quad_perm:[2,3,0,1]")
        "v_add_f32_dpp %5, %5, %5" LL_R("quad_perm:[2,3,0,1]")
        "v_add_f32_dpp %6, %7, %6" LL_R("quad_perm:[2,3,0,1]")
        "v_add_f32_dpp %3, %3, %3" LL_R("row_ror:4")
        "v_add_f32_dpp %5, %5, %5" LL_R("row_ror:4")
        "v_add_f32_e32 %2, %2, %6\n\t"
        "v_add_f32_dpp %3, %3, %3" LL_R("row_ror:8")
        "v_add_f32_dpp %5, %5, %5" LL_R("row_ror:8")
        "v_add_f32_e32 %0, %0, %3\n\t"
        "v_add_f32_e32 %1, %1, %5"
        : "+v"(VA), "+v"(VB), "+v"(VJ), "+v"(p0), "+v"(p2), "+v"(q0), "+v"(j0), "+v"(j2), "+v"(lam)
        : "v"(p1), "v"(p3), "v"(q1), "v"(j1), "v"(j3), "v"(dl));
#undef LL_R
#undef LL_Q
  }
  // vel_commit for TWO rows of a lane at once (the friction pair of the cone-coupled round): the products of both rows are added before
  // the transpose-reduce, so the pair costs one reduce (five v_pk_mul, five v_pk_fma, 16 adds) instead of two.
  LL_D static void vel_commit2(F dl1, F& lam1, F2 a01, F2 a23, F2 b01, F2 j01, F2 j23, F dl2, F& lam2, F2 c01, F2 c23, F2 e01, F2 k01, F2 k23,
                               F& VA, F& VB, F& VJ) {
    // (explicit multiply, then fused multiply-add: the same roundings in every build of the kernel, whatever -ffp-contract=fast would pick)
    const F2 D2 = {dl2, dl2};
    const F2 P01 = __builtin_elementwise_fma(c01, D2, a01 * dl1), P23 = __builtin_elementwise_fma(c23, D2, a23 * dl1), Q01 = __builtin_elementwise_fma(e01, D2, b01 * dl1),
             J01 = __builtin_elementwise_fma(k01, D2, j01 * dl1), J23 = __builtin_elementwise_fma(k23, D2, j23 * dl1);
    float p0 = P01.x, p1 = P01.y, p2 = P23.x, p3 = P23.y, q0 = Q01.x, q1 = Q01.y, j0 = J01.x, j1 = J01.y, j2 = J23.x, j3 = J23.y;
#define LL_R(X) " " X " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("v_add_f32_e32 %8, %8, %15\n\t"
        "v_add_f32_e32 %9, %9, %16\n\t"
        "v_add_f32_dpp %3, %10, %3" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %4, %11, %4" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %5, %12, %5" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %6, %13, %6" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %7, %14, %7" LL_R("quad_perm:[1,0,3,2]")
        "v_add_f32_dpp %3, %4, %3" LL_R("quad_perm:[2,3,0,1]")
        "v_add_f32_dpp %5, %5, %5" LL_R("quad_perm:[2,3,0,1]")
        "v_add_f32_dpp %6, %7, %6" LL_R("quad_perm:[2,3,0,1]")
        "v_add_f32_dpp %3, %3, %3" LL_R("row_ror:4")
        "v_add_f32_dpp %5, %5, %5" LL_R("row_ror:4")
        "v_add_f32_e32 %2, %2, %6\n\t"
        "v_add_f32_dpp %3, %3, %3" LL_R("row_ror:8")
        "v_add_f32_dpp %5, %5, %5" LL_R("row_ror:8")
        "v_add_f32_e32 %0, %0, %3\n\t"
        "v_add_f32_e32 %1, %1, %5"
        : "+v"(VA), "+v"(VB), "+v"(VJ), "+v"(p0), "+v"(p2), "+v"(q0), "+v"(j0), "+v"(j2), "+v"(lam1), "+v"(lam2)
        : "v"(p1), "v"(p3), "v"(q1), "v"(j1), "v"(j3), "v"(dl1), "v"(dl2));
#undef LL_R
  }
  // what the scattered registers hold, for the code after the sweep: dx[i] (env-uniform), dq[j] (leg-uniform)
  template <int I_>
  static LL_D float vel_dx(F VA, F VB) { return I_ < 4 ? LL_DPP_MOV(VA, 0x150 + (I_ & 3)) : LL_DPP_MOV(VB, 0x150 + (I_ & 1)); }
  template <int J_>
  static LL_D F vel_dq(F VJ) { return LL_DPP_MOV(VJ, J_ | (J_ << 2) | (J_ << 4) | (J_ << 6)); }
  // the sixteen turn masks (lane t of every row), made opaque so they stay resident in SGPR pairs across the solver loop
  // instead of being rebuilt from a 32-bit half before every turn
  LL_D void prepare_turn_masks() const {
    for (int t = 0; t < 16; t++) {
      unsigned long long m = 0x0001000100010001ull << t;
      asm volatile("" : "+s"(m));
      tm_[t] = m;
    }
  }
  static LL_D bool any(B m) { return __any(m); }   // wave-level: guards wave-uniform branches

  // ---- constants -------------------------------------------------------------------------------------------------
  LL_D F legc(const float*, int field) const { return lds_[cbase_ + field * 4]; }
  LL_D F candc(int word) const { return lds_[tbase_ + word * PMC_ROW]; }                           // own (leg, sub) column
  LL_D float basec(const float* bc, int i) const { return bc[i]; }                                // env-uniform base constants
  LL_D F candc_of(I sub2, I word) const { return lds_[tbase_ - sub_ + sub2 + word * PMC_ROW]; }    // column of another sub-lane of the leg
  // lane pick for 3-vectors: leg 0 -> x, 1 -> y, 2,3 -> z
  LL_D F pick3(float x, float y, float z) const { return leg_ == 0 ? x : (leg_ == 1 ? y : z); }

  // ---- global memory, per-leg strided access: element (base + stride * leg); stores from sub-lane 0 only -----------
  LL_D F ldl(const float* p, long base, long stride) const { return p[base + stride * leg_]; }
  LL_D void stl(float* p, long base, long stride, F v) const { if (sub_ == 0) p[base + stride * leg_] = v; }
  LL_D void stl_if(B m, float* p, long base, long stride, F v) const { if (m && sub_ == 0) p[base + stride * leg_] = v; }
  LL_D D lddl(const double* p, long base, long stride) const { return p[base + stride * leg_]; }
  // per-leg values / gathers: leg l of the row works on item l (the four future-goal sites of an observation)
  LL_D I pick4i(int a, int b, int c, int d) const { return leg_ == 0 ? a : (leg_ == 1 ? b : (leg_ == 2 ? c : d)); }
  LL_D D pick4d(double a, double b, double c, double d) const { return leg_ == 0 ? a : (leg_ == 1 ? b : (leg_ == 2 ? c : d)); }
  LL_D D ldd_idx(const double* p, I idx) const { return p[idx]; }
  // cooperative copy of up to 16 consecutive floats: lane i moves element i0 + i (if below n)
  LL_D void copy16(float* dst, const float* src, int i0, int n) const { const int i = i0 + lane16_; if (i < n) dst[i] = src[i]; }
  // the two halves of copy16, so that a caller can issue every load of a row before its first store
  // (the index is clamped rather than the load predicated: a predicated load is a branch, and a wait, per chunk; n >= 1)
  LL_D F ld16(const float* src, int i0, int n) const { const int i = i0 + lane16_; const float v = src[i < n ? i : n - 1]; return i < n ? v : 0.0f; }
  LL_D void st16(float* dst, int i0, int n, F v) const { const int i = i0 + lane16_; if (i < n) dst[i] = v; }
  // the same with the row rotated: element i goes to i + (n - split) if i < split, else to i - split  (obs row prop | prop_a | future -> future | prop | prop_a)
  LL_D void st16_rot(float* dst, int i0, int n, int split, F v) const { const int i = i0 + lane16_; if (i < n) dst[i < split ? i + (n - split) : i - split] = v; }
  // number of entries of the non-decreasing table p[0..n) that are <= u: sixteen entries per round trip, row-summed
  LL_D int count_le16(const double* p, int n, double u) const {
    float c = 0.0f;
    for (int i0 = 0; i0 < n; i0 += PMC_ROW) {
      const int i = i0 + lane16_;
      // (device-scope load: inside a multi-step launch the table version was written by a wave on another XCD, whose L2 is not this one's)
      if (i < n && __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= u) c += 1.0f;
    }
    return (int)qsum(subsum(c));
  }
  static LL_D F d2f(D x) { return (float)x; }
  static LL_D F i2f(I x) { return (float)x; }
  static LL_D I f2i(F x) { return (int)x; }
};

// The same lanes with the per-leg constant table held in registers for the whole kernel instead of re-read from LDS in
// every substep.  For the occupancy-1 build only: with one wavefront per SIMD nothing hides an LDS round trip (measured:
// a quarter of the kernel's cycles sat in s_waitcnt on constant reads), while 256 AGPRs lie idle as spill space -- the
// register allocator parks the table there and a use costs one v_accvgpr_read instead of a ds_read plus its latency.
template <int N_LEG_FIELDS, int PIN_CANDS = 0, int N_BASE = 0, int LKB = 0>   // LKB > 0: the ten own-link words at candidate-table word LKB are held too
struct GpuLanesPinned : GpuLanes {
  float lc_[N_LEG_FIELDS];
  float lkp_[LKB > 0 ? 10 : 1];
  // PIN_CANDS > 0 (the PMC step kernel, which has the registers to spare): the fields of the lane's first PIN_CANDS contact candidates
  // that the per-substep candidate loop reads (A, ax, r, link: 8 of the 12 words) and the base constants are held too.  Measured
  // before: 40 s_waitcnt lgkmcnt per substep, each a few instructions behind its ds_read -- an LDS round trip nobody hides.
  float cc_[PIN_CANDS > 0 ? PIN_CANDS * 8 : 1];
  float bcr_[N_BASE > 0 ? N_BASE : 1];
  static constexpr bool kHoldLink = true;    // ... or hold them in registers across the substep loop (the occupancy-1 PMC kernel)
  LL_D GpuLanesPinned(float* lds) : GpuLanes(lds) {}
  LL_D void stage_consts(const float* legc, int n_leg_fields, const float* candc, int n_cand_words, const float* basec = nullptr) {
    GpuLanes::stage_consts(legc, n_leg_fields, candc, n_cand_words);
    LL_UNROLL
    for (int i = 0; i < N_LEG_FIELDS; i++) lc_[i] = lds_[i * 4 + leg_];
    LL_UNROLL
    for (int j = 0; j < PIN_CANDS; j++) {
      LL_UNROLL
      for (int f = 0; f < 8; f++) cc_[j * 8 + f] = lds_[tbase_ + (j * 12 + (f < 6 ? f : f + 3)) * PMC_ROW];
    }
    LL_UNROLL
    for (int i = 0; i < N_BASE; i++) bcr_[i] = basec[i];
    if (LKB > 0) {
      LL_UNROLL
      for (int i = 0; i < 10; i++) lkp_[i] = lds_[tbase_ + (LKB + i) * PMC_ROW];
    }
  }
  LL_D F legc(const float*, int field) const { return lc_[field]; }
  LL_D F candc(int word) const {
    if (LKB > 0 && word >= LKB) return lkp_[word - LKB];
    const int j = word / 12, f = word % 12;
    if (j < PIN_CANDS && (f < 6 || f == 9 || f == 10)) return cc_[j * 8 + (f < 6 ? f : f - 3)];
    return lds_[tbase_ + word * PMC_ROW];
  }
  LL_D float basec(const float* bc, int i) const { return N_BASE > 0 ? bcr_[i] : bc[i]; }
};

// The kernel's argument block (StepParams is every step kernel's FIRST argument; only step kernels call params()).  A lane policy wrapped
// in WithParamsReload hands out a reference to the argument block through a pointer the compiler has to treat as new at that point: scalar
// values derived from the arguments are then re-read from the argument block (scalar loads from the constant cache) where they are needed,
// instead of being computed once at kernel entry and parked in spilled SGPRs (a v_readlane per use: 576 of them in the multi-step PMC
// kernel before, 94 after).  LEVEL 1: once per control step; 2: also once per substep.  Which level pays is a matter of register
// allocation and measured per kernel (A/B on one box, tools/ab3.sh): PMC -2.2 % kernel time at level 1 (both builds); EPMC -0.8 % at
// level 2 (one wave per SIMD) / -1.2 % at level 1 (larger batches); SEPMC +0.3 % (one wave per SIMD: left alone); its larger-batch build
// lost 20 % to it while it carried 40 episode scalars through the substep loop and gains 8 % now that they wait in LDS (park_row).
template <class Base, int LEVEL>
struct WithParamsReload : Base {
  using Base::Base;
  static constexpr int kParamsReload = LEVEL;
  // The kernarg segment starts with the kernel's FIRST by-value argument: params() may only be asked for that one.  Every kernel that
  // instantiates this lane policy takes `StepParams P` first (llenv.hip: pmc_ / epmc_ / sepmc_step_kernel); the tag below makes any other
  // type a compile error instead of a silent read of the wrong bytes.
  template <class T>
  LL_D const T& params(const T&) const {
    static_assert(T::kFirstKernelArgument, "WithParamsReload::params(): only the kernel's first by-value argument lives at the start of the kernarg segment");
    const __attribute__((address_space(4))) void* k = (const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const T*)(const __attribute__((address_space(4))) T*)k;
  }
};

// Terrain builds (EPMC / SEPMC): read the terrain shape records of the contact-candidate loops one shape ahead (pmc_step.hpp shape_sdf_rec).
// Measured per kernel on one box (tools/ab.sh): EPMC at 65536 envs 20.4 -> 22.5 M env-steps/s (+10.6 %), at 4096 envs +0.3 % slower;
// SEPMC at 2048 arenas -1.1 % per step, at 32768 arenas 16.7 -> 16.1 M (-3.7 %): on for the larger-batch EPMC and the one-wave-per-SIMD SEPMC kernels.
template <class Base, bool ON>
struct WithShapePrefetch : Base {
  using Base::Base;
  static constexpr bool kPrefetchShapes = ON;
};

// Rays of the EPMC / SEPMC observation a lane carries through ONE walk of its family's box list (epmc_step.hpp observe_rays): 7 keeps the
// running answers of a third of a lane's 21 rays in registers and reads each box record once per chunk instead of once per ray.  Measured on
// one box (tools/gpu_tasks.sh rays, profiles/r04_ray_ab.txt): one wave per SIMD -1.5 % (hurdles) ... -3 % (chase-tag arenas with elements) on top of
// the row-parallel list building; the 256-register builds LOSE 11 % (EPMC 65536 envs) and 18 % (SEPMC 32768 arenas) to the extra live
// registers, so they keep one ray at a time.  The SEPMC one-wave-per-SIMD kernel (256 + 255 registers) fails its arena invariants on the
// GPU with 7 and passes with 3 (tools/diag_sepmc_rays.py): it runs 3.
template <class Base, int N>
struct WithRayChunk : Base {
  using Base::Base;
  static constexpr int kRayChunk = N;
};

// the Gram blocks of the contact rows as a background job of the matrix cores (GpuLanes::gram_mfma / gram_collect; pmc_step.hpp contact_rows_cone_piped): per kernel by A/B (llenv.hip LL_GRAM_PIPE*)
template <class Base>
struct WithGramPipe : Base {
  using Base::Base;
  static constexpr bool kGramPipe = true;
};

// the cone round keeps its cross scalars in the row's LDS scratch instead of 32 registers (the two-waves-per-SIMD PMC / EPMC builds and the
// one-wave-per-SIMD SEPMC builds, llenv.hip LL_CONE_LDS*; the launch must allocate the row scratch)
template <class Base>
struct WithConeInLds : Base {
  using Base::Base;
  static constexpr bool kConeInLds = true;
};

#define LL_FMAC_RBCAST(L_)                                                                                               \
  template <>                                                                                                           \
  LL_D void GpuLanes::fmac_rbcast<L_>(float& acc, float x, float k) {                                                    \
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:" #L_ " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(k)); \
  }                                                                                                                     \
  template <>                                                                                                           \
  LL_D void GpuLanes::fmac_rbcast_settled<L_>(float& acc, float x, float k) {                                            \
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #L_ " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(k)); \
  }
LL_FMAC_RBCAST(0) LL_FMAC_RBCAST(1) LL_FMAC_RBCAST(2) LL_FMAC_RBCAST(3) LL_FMAC_RBCAST(4) LL_FMAC_RBCAST(5) LL_FMAC_RBCAST(6) LL_FMAC_RBCAST(7)
LL_FMAC_RBCAST(8) LL_FMAC_RBCAST(9) LL_FMAC_RBCAST(10) LL_FMAC_RBCAST(11) LL_FMAC_RBCAST(12) LL_FMAC_RBCAST(13) LL_FMAC_RBCAST(14) LL_FMAC_RBCAST(15)
#undef LL_FMAC_RBCAST
#endif  // __HIPCC__
