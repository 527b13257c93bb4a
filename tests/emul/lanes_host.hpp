// lanes_host.hpp -- TEST INFRASTRUCTURE: a 4-wide host stand-in for one quad of GPU lanes, so that the kernel
// body (lifelike_agility_and_play_amd/csrc/pmc_step.hpp) can be executed, one environment at a time, on a machine
// without a GPU.  It exists to debug the kernel's logic against the oracle here; it is compiled only by the tests
// and is never linked into, or reachable from, the product library.
#pragma once
#include <math.h>
#include <stdint.h>
#include <vector>

struct b4 { bool v[4]; };
struct f4 {
  float v[4];
  f4() {}
  explicit f4(float x) { for (int i = 0; i < 4; i++) v[i] = x; }
};
struct i4 {
  int v[4];
  i4() {}
  i4(int x) { for (int i = 0; i < 4; i++) v[i] = x; }
};
struct d4 {
  double v[4];
  d4() {}
  explicit d4(double x) { for (int i = 0; i < 4; i++) v[i] = x; }
};

#define EMU_BIN(T, OP) \
  inline T operator OP(const T& a, const T& b) { T r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] OP b.v[i]; return r; }
#define EMU_BIN_S(T, S, OP) \
  inline T operator OP(const T& a, S b) { T r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] OP b; return r; } \
  inline T operator OP(S a, const T& b) { T r; for (int i = 0; i < 4; i++) r.v[i] = a OP b.v[i]; return r; }
#define EMU_CMP(T, OP) \
  inline b4 operator OP(const T& a, const T& b) { b4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] OP b.v[i]; return r; }
#define EMU_CMP_S(T, S, OP) \
  inline b4 operator OP(const T& a, S b) { b4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] OP b; return r; }
EMU_BIN(f4, +) EMU_BIN(f4, -) EMU_BIN(f4, *) EMU_BIN(f4, /)
EMU_BIN_S(f4, float, +) EMU_BIN_S(f4, float, -) EMU_BIN_S(f4, float, *) EMU_BIN_S(f4, float, /)
EMU_CMP(f4, <) EMU_CMP(f4, >) EMU_CMP(f4, <=) EMU_CMP(f4, >=)
EMU_CMP_S(f4, float, <) EMU_CMP_S(f4, float, >) EMU_CMP_S(f4, float, <=) EMU_CMP_S(f4, float, >=)
EMU_BIN(i4, +) EMU_BIN(i4, -) EMU_BIN(i4, *)
EMU_BIN_S(i4, int, +) EMU_BIN_S(i4, int, -) EMU_BIN_S(i4, int, *)
EMU_CMP(i4, <) EMU_CMP(i4, >)
EMU_CMP_S(i4, int, <) EMU_CMP_S(i4, int, >)
EMU_BIN(d4, +) EMU_BIN(d4, -) EMU_BIN(d4, *)
EMU_BIN_S(d4, double, +) EMU_BIN_S(d4, double, -) EMU_BIN_S(d4, double, *)

namespace lm {
#define EMU_UN(NAME, FN) inline f4 NAME(const f4& a) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = FN(a.v[i]); return r; }
EMU_UN(sqrt_, sqrtf) EMU_UN(sin_, sinf) EMU_UN(cos_, cosf) EMU_UN(exp_, expf) EMU_UN(abs_, fabsf)
inline f4 rsqrt_(const f4& a) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = 1.0f / sqrtf(a.v[i]); return r; }
inline f4 min_(const f4& a, const f4& b) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = fminf(a.v[i], b.v[i]); return r; }
inline f4 max_(const f4& a, const f4& b) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = fmaxf(a.v[i], b.v[i]); return r; }
inline f4 sel(const b4& m, const f4& a, const f4& b) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline i4 sel(const b4& m, const i4& a, const i4& b) { i4 r; for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline b4 and_(const b4& a, const b4& b) { b4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
inline b4 or_(const b4& a, const b4& b) { b4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
inline f4 med3_(const f4& x, const f4& lo, const f4& hi) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = fminf(fmaxf(x.v[i], lo.v[i]), hi.v[i]); return r; }
inline f4 rint_(const f4& a) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = rintf(a.v[i]); return r; }
inline b4 odd_(const i4& k) { b4 r; for (int i = 0; i < 4; i++) r.v[i] = (k.v[i] & 1) != 0; return r; }
inline b4 bit1_(const i4& k) { b4 r; for (int i = 0; i < 4; i++) r.v[i] = (k.v[i] & 2) != 0; return r; }
inline b4 not_(const b4& a) { b4 r; for (int i = 0; i < 4; i++) r.v[i] = !a.v[i]; return r; }
}  // namespace lm

struct HostLanes {
  typedef f4 F;
  typedef i4 I;
  typedef d4 D;
  typedef b4 B;
  std::vector<float>* lds_;   // [word][4]

  explicit HostLanes(std::vector<float>* lds) : lds_(lds) {}
  I leg() const { i4 r; for (int i = 0; i < 4; i++) r.v[i] = i; return r; }
  F legf() const { f4 r; for (int i = 0; i < 4; i++) r.v[i] = (float)i; return r; }
  B is_leg(int l) const { b4 r; for (int i = 0; i < 4; i++) r.v[i] = (i == l); return r; }
  F lane_f(float x) const { return f4(x); }
  template <int S> static float bcast(const F& x) { return x.v[S]; }
  static float bcast_rt(const F& x, int s) { return x.v[s]; }
  template <int S> static void fmac_bcast(F& acc, const F& x, const F& k) { for (int i = 0; i < 4; i++) acc.v[i] = acc.v[i] + x.v[S] * k.v[i]; }
  static float qsum(const F& x) { return (x.v[0] + x.v[1]) + (x.v[2] + x.v[3]); }   // same association as the DPP tree
  static void qsum6(const F* x, float* out) { for (int i = 0; i < 6; i++) out[i] = qsum(x[i]); }
  static bool qany(const B& m) { return m.v[0] || m.v[1] || m.v[2] || m.v[3]; }
  static bool any(const B& m) { return qany(m); }
  void refresh_consts() const {}
  F legc(const float* tbl, int field) const { f4 r; for (int i = 0; i < 4; i++) r.v[i] = tbl[field * 4 + i]; return r; }
  F pick3(float x, float y, float z) const { f4 r; r.v[0] = x; r.v[1] = y; r.v[2] = z; r.v[3] = z; return r; }
  F ldl(const float* p, long base, long stride) const { f4 r; for (int i = 0; i < 4; i++) r.v[i] = p[base + stride * i]; return r; }
  void stl(float* p, long base, long stride, const F& v) const { for (int i = 0; i < 4; i++) p[base + stride * i] = v.v[i]; }
  void stl_if(const B& m, float* p, long base, long stride, const F& v) const { for (int i = 0; i < 4; i++) if (m.v[i]) p[base + stride * i] = v.v[i]; }
  D lddl(const double* p, long base, long stride) const { d4 r; for (int i = 0; i < 4; i++) r.v[i] = p[base + stride * i]; return r; }
  static F d2f(const D& x) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = (float)x.v[i]; return r; }
  F lds_ld(const I& w) const { f4 r; for (int i = 0; i < 4; i++) r.v[i] = (*lds_)[w.v[i] * 4 + i]; return r; }
  void lds_st(const I& w, const F& v) const { for (int i = 0; i < 4; i++) (*lds_)[w.v[i] * 4 + i] = v.v[i]; }
  void lds_st_if(const B& m, const I& w, const F& v) const { for (int i = 0; i < 4; i++) if (m.v[i]) (*lds_)[w.v[i] * 4 + i] = v.v[i]; }
  // 16-byte groups behind `base_word` per-lane words (the host array is [word][4 lanes]; groups use the same storage)
  void lds_ld4(int base_word, int g, F* out) const { for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) out[j].v[i] = (*lds_)[(base_word + g * 4 + j) * 4 + i]; }
  void lds_st4(int base_word, int g, const F* in) const { for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) (*lds_)[(base_word + g * 4 + j) * 4 + i] = in[j].v[i]; }
  void lds_st1(int base_word, int g, int j, const F& v) const { for (int i = 0; i < 4; i++) (*lds_)[(base_word + g * 4 + j) * 4 + i] = v.v[i]; }
  static F i2f(const I& x) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = (float)x.v[i]; return r; }
  static I f2i(const F& x) { i4 r; for (int i = 0; i < 4; i++) r.v[i] = (int)x.v[i]; return r; }
};
