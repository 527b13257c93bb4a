// lanes.hpp -- the "quad" execution model of the PMC step kernel.
//
// One environment is stepped by FOUR lanes of a wavefront, one per leg (LegOrder FR, FL, HR, HL): the MAX
// quadruped is a star (base + 4 independent 3-joint chains), so leg-local work (FK, leg inertia, contact
// candidates, constraint rows) runs lane-parallel and everything that couples legs goes through the base,
// i.e. through a reduction / broadcast over the quad (DPP quad_perm on gfx950: no LDS, no barrier).
//
// Values come in two classes:
//   * quad-uniform  ("base" values: pose, twist, 6x6 factors, time...) -- plain float/int/double;
//   * lane-varying  (one value per leg)                              -- L::F / L::I / L::D / L::B.
// The kernel body (pmc_step.hpp) is written once against this interface.  GpuLanes (below) maps it to one
// hardware lane per leg.  tests/emul/ instantiates the same source with a 4-wide host type to debug the
// kernel logic on a machine without a GPU; that build is test infrastructure and is never linked into
// the product library.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LL_HD __host__ __device__ __forceinline__
#define LL_D __device__ __forceinline__
#else
#define LL_HD inline
#define LL_D inline
#endif

namespace lm {
// scalar (quad-uniform) overloads of the math vocabulary used by the generic code
LL_HD float sel(bool m, float a, float b) { return m ? a : b; }
LL_HD int sel(bool m, int a, int b) { return m ? a : b; }
LL_HD float sqrt_(float x) { return sqrtf(x); }
LL_HD float rsqrt_(float x) { return 1.0f / sqrtf(x); }
LL_HD float sin_(float x) { return sinf(x); }
LL_HD float cos_(float x) { return cosf(x); }
LL_HD float atan2_(float y, float x) { return atan2f(y, x); }
LL_HD float exp_(float x) { return expf(x); }
LL_HD float abs_(float x) { return fabsf(x); }
LL_HD float min_(float a, float b) { return fminf(a, b); }
LL_HD float max_(float a, float b) { return fmaxf(a, b); }
LL_HD bool and_(bool a, bool b) { return a && b; }
LL_HD bool or_(bool a, bool b) { return a || b; }
LL_HD bool not_(bool a) { return !a; }
LL_HD float rint_(float x) { return rintf(x); }
LL_HD float med3_(float x, float lo, float hi) {   // clamp for lo <= hi: one v_med3_f32 on the GPU
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
  return fminf(fmaxf(x, lo), hi);
#endif
}
LL_HD bool odd_(int k) { return (k & 1) != 0; }
LL_HD bool bit1_(int k) { return (k & 2) != 0; }
}  // namespace lm

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------------
// GPU mapping: lane = leg.  64-thread workgroups = one wavefront = 16 environments.
// ---------------------------------------------------------------------------------------------------
struct GpuLanes {
  using F = float;
  using I = int;
  using D = double;
  using B = bool;
  static constexpr int kWave = 64;

  int leg_;        // 0..3
  int lane_;       // 0..63 within the wave
  float* lds_;     // workgroup LDS scratch, word w of this lane lives at lds_[w * 64 + lane_]
  mutable int cbase_;  // LDS word offset of the per-leg constant table copy (+ leg), see stage_consts()

  LL_D GpuLanes(float* lds) : leg_(threadIdx.x & 3), lane_(threadIdx.x & 63), lds_(lds), cbase_(0) {}

  // Copy the per-leg constant table [n_fields][4] behind the per-lane scratch (scratch_words * 64 floats) so that a
  // constant costs one ds_read with an immediate offset instead of a VGPR held across the whole substep loop.
  LL_D void stage_consts(const float* tbl, int n_fields, int scratch_words) {
    const int base = scratch_words * kWave;
    for (int i = lane_; i < n_fields * 4; i += kWave) lds_[base + i] = tbl[i];
    cbase_ = base + leg_;
    __builtin_amdgcn_s_waitcnt(0);          // single-wave workgroup: program order + waitcnt is enough
  }
  // make the table offset opaque again so the compiler re-reads constants per substep instead of hoisting ~130 of
  // them into registers for the whole 10-substep loop
  LL_D void refresh_consts() const { asm volatile("" : "+v"(cbase_)); }

  LL_D I leg() const { return leg_; }
  LL_D F legf() const { return (float)leg_; }
  LL_D B is_leg(int l) const { return leg_ == l; }
  LL_D F lane_f(float x) const { return x; }   // promote a uniform to lane-varying

  // quad reductions / broadcasts (DPP quad_perm, row-local, no LDS traffic)
  template <int S>
  static LL_D float bcast(F x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, true));
  }
  // acc += bcast<S>(x) * k as ONE instruction (VOP2 v_fmac_f32 with a DPP source).  x is usually produced by the
  // instruction just before: a VALU write followed by a DPP read of the same VGPR needs 2 wait states (s_nop 1).
  template <int S>
  static LL_D void fmac_bcast(F& acc, F x, F k);
  static LL_D float bcast_rt(F x, int src) {   // runtime (wave-uniform) source leg
    return __shfl(x, (int)((threadIdx.x & 60) | src), 64);
  }
  static LL_D float qsum(F x) {
    // x + swap-pairs, then + swap-halves of the quad: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E
    float y = x + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
    return y + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y), 0x4E, 0xf, 0xf, true));
  }
  // six quad sums at once: two DPP adds per value, interleaved so that no DPP read follows its producer by less than two
  // instructions (the compiler's own sequence pays an s_nop per DPP op)
  static LL_D void qsum6(const F* x, float* out) {
    float q0, q1, q2, q3, q4, q5;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %5, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]));
    out[0] = q0; out[1] = q1; out[2] = q2; out[3] = q3; out[4] = q4; out[5] = q5;
  }
  static LL_D bool qany(B m) { return qsum(m ? 1.0f : 0.0f) > 0.0f; }
  static LL_D bool any(B m) { return __any(m); }   // wave-level: guards wave-uniform branches

  // per-leg constant table [field][4]
  LL_D F legc(const float*, int field) const { return lds_[cbase_ + field * 4]; }
  // lane pick for 3-vectors: leg 0 -> x, 1 -> y, 2,3 -> z
  LL_D F pick3(float x, float y, float z) const { return leg_ == 0 ? x : (leg_ == 1 ? y : z); }

  // global memory, per-leg strided access: element (base + stride * leg)
  LL_D F ldl(const float* p, long base, long stride) const { return p[base + stride * leg_]; }
  LL_D void stl(float* p, long base, long stride, F v) const { p[base + stride * leg_] = v; }
  LL_D void stl_if(B m, float* p, long base, long stride, F v) const { if (m) p[base + stride * leg_] = v; }
  LL_D D lddl(const double* p, long base, long stride) const { return p[base + stride * leg_]; }
  static LL_D F d2f(D x) { return (float)x; }

  // LDS scratch: word w (uniform or lane-varying) of this lane
  LL_D F lds_ld(I w) const { return lds_[w * kWave + lane_]; }
  LL_D void lds_st(I w, F v) const { lds_[w * kWave + lane_] = v; }
  LL_D void lds_st_if(B m, I w, F v) const { if (m) lds_[w * kWave + lane_] = v; }
  // 16-byte groups: group g of this lane at float offset (base_word * 64) + (g * 64 + lane) * 4   (ds_*_b128, conflict free)
  LL_D void lds_ld4(int base_word, int g, F* out) const {
    const float4 v = reinterpret_cast<const float4*>(lds_ + base_word * kWave)[g * kWave + lane_];
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
  LL_D void lds_st4(int base_word, int g, const F* in) const {
    reinterpret_cast<float4*>(lds_ + base_word * kWave)[g * kWave + lane_] = make_float4(in[0], in[1], in[2], in[3]);
  }
  LL_D void lds_st1(int base_word, int g, int j, F v) const { lds_[base_word * kWave + (g * kWave + lane_) * 4 + j] = v; }
  static LL_D F i2f(I x) { return (float)x; }
  static LL_D I f2i(F x) { return (int)x; }
};
#define LL_FMAC_BCAST(S, PERM)                                                                                          \
  template <>                                                                                                           \
  LL_D void GpuLanes::fmac_bcast<S>(float& acc, float x, float k) {                                                     \
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(k)); \
  }
LL_FMAC_BCAST(0, "[0,0,0,0]")
LL_FMAC_BCAST(1, "[1,1,1,1]")
LL_FMAC_BCAST(2, "[2,2,2,2]")
LL_FMAC_BCAST(3, "[3,3,3,3]")
#undef LL_FMAC_BCAST
#endif  // __HIPCC__
