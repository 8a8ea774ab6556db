// Row-wise Poincare-ball math (k = -1).  A row of `dim` <= 256 floats lives in the registers of a group of G lanes,
// EPL elements per lane at columns (lane mod G) + G*e; reductions are DPP butterflies.  Two layouts:
//   RowVec   = RowT<64, 4>: one row per wave -- the fused training kernels, whose rows sit in LDS tiles;
//   RowT<16, EPL>         : four rows per wave, one per DPP row of 16 lanes -- the stand-alone HBM-bound kernels: every
//                           per-row scalar (norm, tanh, artanh, acosh, divisions) is then computed for four rows by one
//                           instruction, and a row sum needs no cross-row step at all.
// Formulas and clamp constants: /root/reference/math_.py (see oracle/gmath.py, oracle/manual.py which these
// transcribe line by line).
#pragma once
#include <type_traits>

#include "device_utils.h"

namespace hypad {

constexpr int MAX_EPL = 4;  // dim <= 256 at 64 lanes per row

template <int G_, int EPL_>
struct RowT {
  static constexpr int G = G_, EPL = EPL_;
  float v[EPL_];
};
using RowVec = RowT<64, MAX_EPL>;

template <class R = RowVec>
__device__ __forceinline__ R row_load(const float* p, int dim, int lane) {
  R r;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) {
    int c = (lane & (R::G - 1)) + R::G * e;
    r.v[e] = c < dim ? p[c] : 0.f;
  }
  return r;
}
template <class R>
__device__ __forceinline__ void row_store(float* p, const R& r, int dim, int lane) {
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) {
    int c = (lane & (R::G - 1)) + R::G * e;
    if (c < dim) p[c] = r.v[e];
  }
}
template <class R>
__device__ __forceinline__ float row_dot(const R& a, const R& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) s += a.v[e] * b.v[e];
  return R::G == 64 ? wave_sum(s) : row16_sum(s);
}

// ---- expmap0 (math_.py:1132-1136)
template <class R>
__device__ __forceinline__ R expmap0_row(const R& u) {
  float n = fmaxf(sqrtf(row_dot(u, u)), MIN_NORM);
  float f = tanhf(fminf(n, TANH_CLAMP)) / n;
  R p;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) p.v[e] = f * u.v[e];
  return p;
}
template <class R>
__device__ __forceinline__ R expmap0_row_bwd(const R& u, const R& dp) {
  float raw = sqrtf(row_dot(u, u));
  float n = fmaxf(raw, MIN_NORM);
  float t = tanhf(fminf(n, TANH_CLAMP));
  float f = t / n;
  float tp = n <= TANH_CLAMP ? 1.f - t * t : 0.f;
  float dfdn = (tp * n - t) / (n * n);
  float s = raw >= MIN_NORM ? row_dot(dp, u) * dfdn / n : 0.f;
  R du;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) du.v[e] = f * dp.v[e] + s * u.v[e];
  return du;
}

// ---- logmap0 (math_.py:1267-1270)
__device__ __forceinline__ float artanh_clamped(float x) {
  x = fminf(fmaxf(x, -1.f + ARTANH_EPS), 1.f - ARTANH_EPS);
  return 0.5f * (logf(1.f + x) - logf(1.f - x));
}
template <class R>
__device__ __forceinline__ R logmap0_row(const R& y) {
  float n = fmaxf(sqrtf(row_dot(y, y)), MIN_NORM);
  float f = artanh_clamped(n) / n;
  R o;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) o.v[e] = f * y.v[e];
  return o;
}
template <class R>
__device__ __forceinline__ R logmap0_row_bwd(const R& y, const R& go) {
  float raw = sqrtf(row_dot(y, y));
  float n = fmaxf(raw, MIN_NORM);
  float a = artanh_clamped(n);
  float f = a / n;
  // d artanh(clamp(n)) / dn: 1/(1-n^2) inside the clamp, 0 outside
  float ap = (n >= -1.f + ARTANH_EPS && n <= 1.f - ARTANH_EPS) ? 1.f / (1.f - n * n) : 0.f;
  float dfdn = (ap * n - a) / (n * n);
  float s = raw >= MIN_NORM ? row_dot(go, y) * dfdn / n : 0.f;
  R gy;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) gy.v[e] = f * go.v[e] + s * y.v[e];
  return gy;
}

// ---- mobius_add (math_.py:536-555)
template <class R>
__device__ __forceinline__ R mobius_add_row(const R& x, const R& y) {
  float x2 = row_dot(x, x), y2 = row_dot(y, y), xy = row_dot(x, y);
  float A = 1.f + 2.f * xy + y2, Bc = 1.f - x2;
  float D = fmaxf(1.f + 2.f * xy + x2 * y2, MIN_NORM);
  const float rD = 1.f / D;              // one division per row; an IEEE division per element is ~12 instructions each
  R m;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) m.v[e] = (A * x.v[e] + Bc * y.v[e]) * rD;
  return m;
}
template <class R>
__device__ __forceinline__ void mobius_add_row_bwd(const R& x, const R& y, const R& dm, R& dx, R& dy) {
  float x2 = row_dot(x, x), y2 = row_dot(y, y), xy = row_dot(x, y);
  float A = 1.f + 2.f * xy + y2, Bc = 1.f - x2;
  float Draw = 1.f + 2.f * xy + x2 * y2;
  float D = fmaxf(Draw, MIN_NORM);
  const float rD = 1.f / D;
  R m, dN;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) {
    m.v[e] = (A * x.v[e] + Bc * y.v[e]) * rD;
    dN.v[e] = dm.v[e] * rD;
  }
  float dD = Draw >= MIN_NORM ? -row_dot(dm, m) * rD : 0.f;
  float dA = row_dot(dN, x), dBc = row_dot(dN, y);
  float dxy = 2.f * dA + 2.f * dD, dx2 = -dBc + y2 * dD, dy2 = dA + x2 * dD;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) {
    dx.v[e] = A * dN.v[e] + 2.f * dx2 * x.v[e] + dxy * y.v[e];
    dy.v[e] = Bc * dN.v[e] + 2.f * dy2 * y.v[e] + dxy * x.v[e];
  }
}

// ---- project (math_.py:340-352)
template <class R>
__device__ __forceinline__ R project_row(const R& x) {
  float n = fmaxf(sqrtf(row_dot(x, x)), MIN_NORM);
  R o = x;
  if (n > BALL_MAXNORM) {
    const float sc = BALL_MAXNORM / n;
#pragma unroll
    for (int e = 0; e < R::EPL; ++e) o.v[e] = x.v[e] * sc;
  }
  return o;
}
template <class R>
__device__ __forceinline__ R project_row_bwd(const R& x, const R& go) {
  float raw = sqrtf(row_dot(x, x));
  float n = fmaxf(raw, MIN_NORM);
  if (!(n > BALL_MAXNORM)) return go;
  float s = raw >= MIN_NORM ? row_dot(x, go) / (n * n) : 0.f;
  R gx;
  const float sc = BALL_MAXNORM / n;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) gx.v[e] = sc * (go.v[e] - x.v[e] * s);
  return gx;
}

// ---- fused Moebius head epilogue (hyrnn_nets.py:27-34)
template <class R>
__device__ __forceinline__ R head_row(const R& u, const R& bias) {
  return project_row(mobius_add_row(expmap0_row(u), bias));
}
template <class R>
__device__ __forceinline__ void head_row_bwd(const R& u, const R& bias, const R& dr, R& du, R& db) {
  R p = expmap0_row(u);
  R m = mobius_add_row(p, bias);
  R dm = project_row_bwd(m, dr);
  R dp;
  mobius_add_row_bwd(p, bias, dm, dp, db);
  du = expmap0_row_bwd(u, dp);
}

// ---- row-wise Poincare distance (train.py:226-230)
template <class R>
__device__ __forceinline__ float rowdist_row(const R& u, const R& v) {
  R d;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) d.v[e] = u.v[e] - v.v[e];
  float sq = row_dot(d, d), un = row_dot(u, u), vn = row_dot(v, v);
  return acoshf(1.f + 2.f * sq / ((1.f - un) * (1.f - vn)) + 1e-7f);
}
template <class R>
__device__ __forceinline__ float rowdist_row_bwd(const R& u, const R& v, float gd, R& du, R& dv) {
  R d;
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) d.v[e] = u.v[e] - v.v[e];
  float sq = row_dot(d, d), un = row_dot(u, u), vn = row_dot(v, v);
  float den = (1.f - un) * (1.f - vn);
  float xt = 1.f + 2.f * sq / den + 1e-7f;
  float gx = gd / sqrtf(xt * xt - 1.f);
  float dsq = gx * 2.f / den;
  float dden = -gx * 2.f * sq / (den * den);
  float dun = -dden * (1.f - vn), dvn = -dden * (1.f - un);
#pragma unroll
  for (int e = 0; e < R::EPL; ++e) {
    du.v[e] = dsq * 2.f * d.v[e] + dun * 2.f * u.v[e];
    dv.v[e] = -dsq * 2.f * d.v[e] + dvn * 2.f * v.v[e];
  }
  return acoshf(xt);
}

// f(integral_constant<int, EPL>) with the 16-lanes-per-row EPL that fits `dim` (wave-uniform branch)
template <class F>
__device__ __forceinline__ void epl16_dispatch(int dim, F f) {
  if (dim <= 64) f(std::integral_constant<int, 4>{});
  else if (dim <= 112) f(std::integral_constant<int, 7>{});      // window 100: 7 elements per lane instead of 8 (the row chains are VALU-bound)
  else if (dim <= 128) f(std::integral_constant<int, 8>{});
  else f(std::integral_constant<int, 16>{});
}

}  // namespace hypad
