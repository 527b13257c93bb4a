// pmc_math.hpp -- small fixed-size algebra used by the PMC step kernel, generic over the scalar type
// (float for quad-uniform values, L::F for lane-varying ones; see lanes.hpp).
#pragma once
#include "lanes.hpp"

// an 8-float record (terrain box: x0 x1 y0 y1 | z0 z1 rod r) as two 16-byte reads -- one LDS / global instruction each
struct alignas(16) F4 { float x, y, z, w; };
struct BoxRec { F4 a, c; };
LL_HD BoxRec load_box(const float* p) {
  BoxRec b;
  b.a = *reinterpret_cast<const F4*>(p);
  b.c = *reinterpret_cast<const F4*>(p + 4);
  return b;
}
LL_HD void store_box(float* p, const BoxRec& b) {
  *reinterpret_cast<F4*>(p) = b.a;
  *reinterpret_cast<F4*>(p + 4) = b.c;
}

template <class T>
struct V3 {
  T x, y, z;
};
template <class T>
LL_HD V3<T> mk3(T x, T y, T z) {
  V3<T> r;
  r.x = x; r.y = y; r.z = z;
  return r;
}
template <class A, class B>
LL_HD auto operator+(const V3<A>& a, const V3<B>& b) -> V3<decltype(a.x + b.x)> {
  return mk3<decltype(a.x + b.x)>(a.x + b.x, a.y + b.y, a.z + b.z);
}
template <class A, class B>
LL_HD auto operator-(const V3<A>& a, const V3<B>& b) -> V3<decltype(a.x - b.x)> {
  return mk3<decltype(a.x - b.x)>(a.x - b.x, a.y - b.y, a.z - b.z);
}
template <class A, class S>
LL_HD auto scale(const V3<A>& a, const S& s) -> V3<decltype(a.x * s)> {
  return mk3<decltype(a.x * s)>(a.x * s, a.y * s, a.z * s);
}
template <class A, class B>
LL_HD auto dot(const V3<A>& a, const V3<B>& b) -> decltype(a.x * b.x) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <class A, class B>
LL_HD auto cross(const V3<A>& a, const V3<B>& b) -> V3<decltype(a.x * b.x)> {
  return mk3<decltype(a.x * b.x)>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class T, class A>
LL_HD V3<T> cvt3(const V3<A>& a) {
  return mk3<T>(T(a.x), T(a.y), T(a.z));
}

// 3x3 matrix, row-major
template <class T>
struct M3 {
  T m[9];
};
template <class A, class B>
LL_HD auto mul(const M3<A>& a, const V3<B>& v) -> V3<decltype(a.m[0] * v.x)> {
  return mk3<decltype(a.m[0] * v.x)>(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
                                      a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
template <class A, class B>
LL_HD auto mulT(const M3<A>& a, const V3<B>& v) -> V3<decltype(a.m[0] * v.x)> {  // a^T v
  return mk3<decltype(a.m[0] * v.x)>(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
                                      a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
template <class A, class B>
LL_HD auto mul(const M3<A>& a, const M3<B>& b) -> M3<decltype(a.m[0] * b.m[0])> {
  M3<decltype(a.m[0] * b.m[0])> r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}

// symmetric 3x3
template <class T>
struct S3 {
  T xx, xy, xz, yy, yz, zz;
};
template <class A, class B>
LL_HD auto mul(const S3<A>& s, const V3<B>& v) -> V3<decltype(s.xx * v.x)> {
  return mk3<decltype(s.xx * v.x)>(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z,
                                    s.xz * v.x + s.yz * v.y + s.zz * v.z);
}
// R S R^T for a rotation R and symmetric S
template <class A, class B>
LL_HD auto rot_sym(const M3<A>& R, const S3<B>& s) -> S3<decltype(R.m[0] * s.xx)> {
  typedef decltype(R.m[0] * s.xx) T;
  T t[9];
  for (int i = 0; i < 3; i++) {
    t[3 * i + 0] = R.m[3 * i] * s.xx + R.m[3 * i + 1] * s.xy + R.m[3 * i + 2] * s.xz;
    t[3 * i + 1] = R.m[3 * i] * s.xy + R.m[3 * i + 1] * s.yy + R.m[3 * i + 2] * s.yz;
    t[3 * i + 2] = R.m[3 * i] * s.xz + R.m[3 * i + 1] * s.yz + R.m[3 * i + 2] * s.zz;
  }
  S3<T> o;
  o.xx = t[0] * R.m[0] + t[1] * R.m[1] + t[2] * R.m[2];
  o.xy = t[0] * R.m[3] + t[1] * R.m[4] + t[2] * R.m[5];
  o.xz = t[0] * R.m[6] + t[1] * R.m[7] + t[2] * R.m[8];
  o.yy = t[3] * R.m[3] + t[4] * R.m[4] + t[5] * R.m[5];
  o.yz = t[3] * R.m[6] + t[4] * R.m[7] + t[5] * R.m[8];
  o.zz = t[6] * R.m[6] + t[7] * R.m[7] + t[8] * R.m[8];
  return o;
}

// spatial vector: angular part a, linear part l (motion: [omega; v_O], force: [n_O; f])
template <class T>
struct SV {
  V3<T> a, l;
};
template <class A, class B>
LL_HD auto operator+(const SV<A>& p, const SV<B>& q) -> SV<decltype(p.a.x + q.a.x)> {
  SV<decltype(p.a.x + q.a.x)> r;
  r.a = p.a + q.a; r.l = p.l + q.l;
  return r;
}
template <class A, class S>
LL_HD auto scale(const SV<A>& p, const S& s) -> SV<decltype(p.a.x * s)> {
  SV<decltype(p.a.x * s)> r;
  r.a = scale(p.a, s); r.l = scale(p.l, s);
  return r;
}
template <class A, class B>
LL_HD auto dot(const SV<A>& p, const SV<B>& q) -> decltype(p.a.x * q.a.x) {
  return dot(p.a, q.a) + dot(p.l, q.l);
}
template <class T, class A>
LL_HD SV<T> cvt6(const SV<A>& p) {
  SV<T> r;
  r.a = cvt3<T>(p.a); r.l = cvt3<T>(p.l);
  return r;
}
// motion cross product v x m
template <class A, class B>
LL_HD auto crm(const SV<A>& v, const SV<B>& m) -> SV<decltype(v.a.x * m.a.x)> {
  SV<decltype(v.a.x * m.a.x)> r;
  r.a = cross(v.a, m.a);
  r.l = cross(v.a, m.l) + cross(v.l, m.a);
  return r;
}
// force cross product v x* f
template <class A, class B>
LL_HD auto crf(const SV<A>& v, const SV<B>& f) -> SV<decltype(v.a.x * f.a.x)> {
  SV<decltype(v.a.x * f.a.x)> r;
  r.a = cross(v.a, f.a) + cross(v.l, f.l);
  r.l = cross(v.a, f.l);
  return r;
}

// rigid-body inertia about the reference origin, 10 parameters
template <class T>
struct RI {
  T m;
  V3<T> h;   // m * com
  S3<T> io;  // rotational inertia about the origin
};
template <class A, class B>
LL_HD auto apply(const RI<A>& I, const SV<B>& v) -> SV<decltype(I.m * v.a.x)> {
  SV<decltype(I.m * v.a.x)> r;
  r.a = mul(I.io, v.a) + cross(I.h, v.l);
  r.l = scale(v.l, I.m) - cross(I.h, v.a);
  return r;
}
template <class T>
LL_HD RI<T> add(const RI<T>& p, const RI<T>& q) {
  RI<T> r;
  r.m = p.m + q.m;
  r.h = p.h + q.h;
  r.io.xx = p.io.xx + q.io.xx; r.io.xy = p.io.xy + q.io.xy; r.io.xz = p.io.xz + q.io.xz;
  r.io.yy = p.io.yy + q.io.yy; r.io.yz = p.io.yz + q.io.yz; r.io.zz = p.io.zz + q.io.zz;
  return r;
}

// ---- quaternions (xyzw), quad-uniform float -----------------------------------------------------------
struct Q4 {
  float x, y, z, w;
};
LL_HD Q4 qmul(const Q4& p, const Q4& q) {
  Q4 r;
  r.x = p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y;
  r.y = p.w * q.y - p.x * q.z + p.y * q.w + p.z * q.x;
  r.z = p.w * q.z + p.x * q.y - p.y * q.x + p.z * q.w;
  r.w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
  return r;
}
LL_HD Q4 qconj(const Q4& q) {
  Q4 r = {-q.x, -q.y, -q.z, q.w};
  return r;
}
LL_HD Q4 qnormalize(const Q4& q) {
  float n = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Q4 r = {q.x * n, q.y * n, q.z * n, q.w * n};
  return r;
}
LL_HD M3<float> qmat(const Q4& q) {  // unit quaternion -> rotation (world <- body)
  M3<float> R;
  float x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
  float xy = q.x * q.y, zw = q.z * q.w, xz = q.x * q.z, yw = q.y * q.w, yz = q.y * q.z, xw = q.x * q.w;
  R.m[0] = x2 - y2 - z2 + w2; R.m[1] = 2 * (xy - zw); R.m[2] = 2 * (xz + yw);
  R.m[3] = 2 * (xy + zw); R.m[4] = -x2 + y2 - z2 + w2; R.m[5] = 2 * (yz - xw);
  R.m[6] = 2 * (xz - yw); R.m[7] = 2 * (yz + xw); R.m[8] = -x2 - y2 + z2 + w2;
  return R;
}
// rotation vector of a quaternion given as (vector part v, scalar w), shortest arc (scipy as_rotvec)
LL_HD V3<float> rotvec_of(V3<float> v, float w) {
  if (w < 0.0f) { v.x = -v.x; v.y = -v.y; v.z = -v.z; w = -w; }
  float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  float angle = 2.0f * atan2f(n, w);
  float s;
  if (angle <= 1e-3f) {
    float a2 = angle * angle;
    s = 2.0f + a2 * (1.0f / 12.0f) + 7.0f * a2 * a2 * (1.0f / 2880.0f);
  } else {
    s = angle / sinf(0.5f * angle);
  }
  return mk3<float>(s * v.x, s * v.y, s * v.z);
}
LL_HD Q4 quat_of_rotvec(const V3<float>& rv) {  // scipy from_rotvec
  float angle = sqrtf(rv.x * rv.x + rv.y * rv.y + rv.z * rv.z);
  float s;
  if (angle <= 1e-3f) {
    float a2 = angle * angle;
    s = 0.5f - a2 * (1.0f / 48.0f) + a2 * a2 * (1.0f / 3840.0f);
  } else {
    s = sinf(0.5f * angle) / angle;
  }
  Q4 q = {s * rv.x, s * rv.y, s * rv.z, cosf(0.5f * angle)};
  return q;
}
// exp map for the per-substep base rotation w*dt (|angle| far below pi): polynomial sin/cos of the half angle, no
// library call.  Falls back to the general form for |angle| >= 1 (an angular rate above 500 rad/s at dt = 2 ms).
LL_HD Q4 quat_of_small_rotvec(const V3<float>& rv) {
  float a2 = rv.x * rv.x + rv.y * rv.y + rv.z * rv.z;
  if (a2 >= 1.0f) return quat_of_rotvec(rv);
  float h2 = 0.25f * a2;                           // (angle/2)^2
  // sin(h)/h * 0.5 and cos(h), h = angle/2, Taylor to h^8 (error < 1e-9 for h <= 0.5)
  float s = 0.5f * (1.0f + h2 * (-1.0f / 6.0f + h2 * (1.0f / 120.0f + h2 * (-1.0f / 5040.0f + h2 * (1.0f / 362880.0f)))));
  float c = 1.0f + h2 * (-0.5f + h2 * (1.0f / 24.0f + h2 * (-1.0f / 720.0f + h2 * (1.0f / 40320.0f))));
  Q4 q = {s * rv.x, s * rv.y, s * rv.z, c};
  return q;
}
// PLE:19-23 quat2axisangle then axis*angle; returns angle, writes axis*angle
LL_HD float axis_angle_scaled(const Q4& q, V3<float>* aa) {
  V3<float> rv = rotvec_of(mk3<float>(q.x, q.y, q.z), q.w);
  float angle = sqrtf(rv.x * rv.x + rv.y * rv.y + rv.z * rv.z);
  float k = angle / (angle + 1e-8f);
  *aa = mk3<float>(rv.x * k, rv.y * k, rv.z * k);
  return angle;
}

// ---- the same quaternion helpers for lane-varying scalars (T = L::F): branch-free, both sides of a case evaluated and selected.
//      The four future-goal sites of an observation are worked on by the four legs of the row at once (pmc_step.hpp obs_emit).
template <class T>
struct Q4T {
  T x, y, z, w;
};
template <class P, class Q>
LL_HD auto qmul_t(const P& p, const Q& q) -> Q4T<decltype(p.w * q.x)> {
  Q4T<decltype(p.w * q.x)> r;
  r.x = p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y;
  r.y = p.w * q.y - p.x * q.z + p.y * q.w + p.z * q.x;
  r.z = p.w * q.z + p.x * q.y - p.y * q.x + p.z * q.w;
  r.w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
  return r;
}
template <class T>
LL_HD Q4T<T> qconj_t(const Q4T<T>& q, const T& zero) {
  Q4T<T> r = {zero - q.x, zero - q.y, zero - q.z, q.w};
  return r;
}
template <class T>
LL_HD Q4T<T> qnormalize_t(const Q4T<T>& q) {
  T n = lm::rsqrt_(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Q4T<T> r = {q.x * n, q.y * n, q.z * n, q.w * n};
  return r;
}
template <class T>
LL_HD V3<T> rotvec_of_t(V3<T> v, T w, const T& zero) {       // rotvec_of for lane-varying arguments
  auto neg = w < 0.0f;
  v.x = lm::sel(neg, zero - v.x, v.x); v.y = lm::sel(neg, zero - v.y, v.y); v.z = lm::sel(neg, zero - v.z, v.z);
  w = lm::sel(neg, zero - w, w);
  T n = lm::sqrt_(v.x * v.x + v.y * v.y + v.z * v.z);
  T angle = lm::atan2_(n, w) * 2.0f;
  T a2 = angle * angle;
  T s_small = a2 * (1.0f / 12.0f) + (a2 * a2) * (7.0f / 2880.0f) + 2.0f;
  T s_big = angle / lm::sin_(angle * 0.5f);
  T s = lm::sel(angle <= 1e-3f, s_small, s_big);
  return mk3<T>(s * v.x, s * v.y, s * v.z);
}
template <class T>
LL_HD Q4T<T> quat_of_rotvec_t(const V3<T>& rv) {
  T angle = lm::sqrt_(rv.x * rv.x + rv.y * rv.y + rv.z * rv.z);
  T a2 = angle * angle;
  T s_small = (a2 * a2) * (1.0f / 3840.0f) - a2 * (1.0f / 48.0f) + 0.5f;
  T s_big = lm::sin_(angle * 0.5f) / angle;
  T s = lm::sel(angle <= 1e-3f, s_small, s_big);
  Q4T<T> q = {s * rv.x, s * rv.y, s * rv.z, lm::cos_(angle * 0.5f)};
  return q;
}
template <class T>
LL_HD T axis_angle_scaled_t(const Q4T<T>& q, V3<T>* aa, const T& zero) {
  V3<T> rv = rotvec_of_t(mk3<T>(q.x, q.y, q.z), q.w, zero);
  T angle = lm::sqrt_(rv.x * rv.x + rv.y * rv.y + rv.z * rv.z);
  T k = angle / (angle + 1e-8f);
  *aa = mk3<T>(rv.x * k, rv.y * k, rv.z * k);
  return angle;
}

// ---- Philox4x32-10 (counter-based RNG, quad-uniform) -------------------------------------------------------
LL_HD void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
LL_HD double u01_from(uint32_t hi, uint32_t lo) {  // uniform in [0,1) with 53 bits
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
