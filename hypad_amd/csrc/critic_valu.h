// Critic MLPs on the vector ALUs with their weights staged in LDS.
//
// The critics are 20-wide (models/tadgan.py:77-89,113-119): a layer is 16 x 20 x 20 MACs per tile -- far below what
// pays for an MFMA tile's fixed cost (measured ~2 000 cycles per 16x16 tile incl. address setup, scripts/diag_gemm.py).
// Here every thread owns one or two outputs and runs a plain dot product out of LDS (~200 cycles per layer), and
// the independent passes of a WGAN-GP critic update (real / fake / interpolated: train.py:21,35,72) are carried
// TOGETHER as 16*NP rows through each layer, so a critic iteration costs ~10 short phases instead of ~45.
#pragma once
#include "nets.h"

namespace hypad {

struct CriticBatchLds {
  float* act;   // [nh][R][LQ]   output of hidden layer li (post LeakyReLU and dropout)
  float* dm;    // [nh][R][LQ]   leaky'(pre) * dropout scale
  float* dl;    // [2][R][LQ]    ping-pong deltas
  float* out;   // [R]
  int LQ, R;
};
HD int critic_batch_lds_floats(int R, int L) { return (4 + 4 + 2) * R * (pad4(L) + 4) + pad4(R); }
__device__ __forceinline__ CriticBatchLds critic_batch_lds(float* base, int R, int L) {
  CriticBatchLds c;
  c.LQ = pad4(L) + 4; c.R = R;
  c.act = base; c.dm = base + 4 * R * c.LQ; c.dl = c.dm + 4 * R * c.LQ; c.out = c.dl + 2 * R * c.LQ;
  return c;
}

// dot product of two 16-byte aligned rows (K % 4 == 0) or of arbitrary rows (scalar tail-free fallback)
__device__ __forceinline__ float dot_rows(const float* __restrict__ x, const float* __restrict__ w, int K) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if ((K & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
#pragma unroll 5
    for (int k = 0; k < K; k += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(x + k);
      const float4 wv = *reinterpret_cast<const float4*>(w + k);
      a0 += xv.x * wv.x; a1 += xv.y * wv.y; a2 += xv.z * wv.z; a3 += xv.w * wv.w;
    }
  } else {
    int k = 0;
    for (; k + 1 < K; k += 2) { a0 += x[k] * w[k]; a1 += x[k + 1] * w[k + 1]; }
    if (k < K) a2 += x[k] * w[k];
  }
  return (a0 + a1) + (a2 + a3);
}

// out[r][c] = sum_k in[r][k] * W[c][k] + b[c]   (W row-major (N, K), in LDS or global through a flat pointer)
template <class Epi>
__device__ __forceinline__ void dense_rows_valu(const float* in, int ldin, int K, const float* W, const float* b, int R, int N, Epi epi) {
  const int total = R * N;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int r = idx / N, c = idx - r * N;
    epi(r, c, dot_rows(in + r * ldin, W + c * K, K) + (b ? b[c] : 0.f));
  }
}
// out[r][c] = sum_o d[r][o] * W[o][c]   (W row-major (No, C))
template <class Epi>
__device__ __forceinline__ void dense_rows_valu_t(const float* d, int ldd, int No, const float* W, int C, int R, Epi epi) {
  const int total = R * C;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int r = idx / C, c = idx - r * C;
    const float* dr = d + r * ldd;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int o = 0;
#pragma unroll 2
    for (; o + 3 < No; o += 4) {
      a0 += dr[o] * W[o * C + c]; a1 += dr[o + 1] * W[(o + 1) * C + c];
      a2 += dr[o + 2] * W[(o + 2) * C + c]; a3 += dr[o + 3] * W[(o + 3) * C + c];
    }
    for (; o < No; ++o) a0 += dr[o] * W[o * C + c];
    epi(r, c, (a0 + a1) + (a2 + a3));
  }
}

// Forward of R = 16*NP rows.  in(r) -> pointer to row r (in_dim floats, LDS).  drop(r) -> (DropSrc index, batch row).
// P: the critic's arena staged in LDS.
template <class InRow, class DropFn>
__device__ __forceinline__ void critic_batch_fwd(InRow in_row, const float* P, const CriticLayout& cl, int L, const CriticBatchLds& s,
                                                 DropFn drop) {
  const int R = s.R, LQ = s.LQ;
  for (int li = 0; li < cl.nh; ++li) {
    float* a = s.act + li * R * LQ;
    float* dmk = s.dm + li * R * LQ;
    const int K = li == 0 ? cl.in_dim : L;
    const float* W = P + cl.wof(li);
    const float* b = P + cl.bof(li);
    const int total = R * L;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int r = idx / L, c = idx - r * L;
      const float* x = li == 0 ? in_row(r) : s.act + (li - 1) * R * LQ + r * LQ;
      const float pre = dot_rows(x, W + c * K, K) + b[c];
      const float dd = leaky_slope(pre) * drop(li, r, c);
      dmk[r * LQ + c] = dd;
      a[r * LQ + c] = pre * dd;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < R) {
    const float* x = s.act + (cl.nh - 1) * R * LQ + threadIdx.x * LQ;
    const float* w = P + cl.wof(cl.nh);
    float acc = P[cl.bof(cl.nh)];
    for (int c = 0; c < L; ++c) acc += x[c] * w[c];
    s.out[threadIdx.x] = acc;
  }
  __syncthreads();
}

// First-order backward chain for per-row output gradients dout(r).  sink(li, delta [R][LQ]) after each layer
// (li = nh-1 .. 0).  Returns delta_0.
template <class DoutFn, class Sink>
__device__ __forceinline__ const float* critic_batch_bwd(DoutFn dout, const float* P, const CriticLayout& cl, int L,
                                                         const CriticBatchLds& s, Sink sink) {
  const int R = s.R, LQ = s.LQ;
  float* cur = s.dl;
  float* nxt = s.dl + R * LQ;
  {
    const float* wl = P + cl.wof(cl.nh);
    const float* d = s.dm + (cl.nh - 1) * R * LQ;
    for (int idx = threadIdx.x; idx < R * L; idx += blockDim.x) {
      const int r = idx / L, c = idx - r * L;
      cur[r * LQ + c] = dout(r) * wl[c] * d[r * LQ + c];
    }
  }
  __syncthreads();
  sink(cl.nh - 1, cur);
  for (int li = cl.nh - 2; li >= 0; --li) {
    const float* dd = s.dm + li * R * LQ;
    dense_rows_valu_t(cur, LQ, L, P + cl.wof(li + 1), L, R, [&](int r, int c, float v) { nxt[r * LQ + c] = v * dd[r * LQ + c]; });
    __syncthreads();
    sink(li, nxt);
    float* t = cur; cur = nxt; nxt = t;
  }
  return cur;
}

}  // namespace hypad
