// Small dense layers out of LDS on v_mfma_f32_16x16x4_f32: the critics (L <= 32 wide) as the training kernels run
// them -- weights staged in LDS with padded strides, biases folded in as a constant-one input column.
//
// Conventions: row strides are (multiple of 16) + 4 floats; every operand row is zero padded to the next multiple of 16
// columns; operand fetches are ds_read_b128: lane (i, q) supplies k = 16 g + 4 q + s to the s-th MFMA of k-group g (the
// reduction index may be permuted as long as A and B agree).  Written for 512-thread workgroups (8 waves).
#pragma once
#include "layout.h"
#include "tile_gemm.h"

namespace hypad {

constexpr int CM_NW = 8;                     // waves per workgroup the tile loops are dealt over

HD int up16(int n) { return (n + 15) & ~15; }

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}
// out[r][n] = sum_k A[r][k] W[n][k];  A: LDS [16 RT][lda], W: LDS [Nrows][ldw], both zero-padded to Kp columns.
// epi(r, n, value) for n < 16 ceil(Ncols / 16); columns >= Nrows repeat row Nrows - 1 of W (the caller overrides them).
// Tile t is computed by the wave whose wslot == t mod CM_NW.
template <class Epi>
__device__ __forceinline__ void lds_gemm_nt(const float* A, int lda, int RT, const float* W, int ldw, int Nrows, int Ncols, int Kp, int wslot,
                                            int lane, Epi epi) {
  const int j = lane & 15, q = lane >> 4, CT = (Ncols + 15) >> 4;
  for (int t = wslot; t < RT * CT; t += CM_NW) {
    const int rt = t / CT, ct = t - rt * CT;
    int n = ct * 16 + j; n = n < Nrows ? n : Nrows - 1;
    const float* a = A + (rt * 16 + j) * lda + 4 * q;
    const float* b = W + n * ldw + 4 * q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int g = 0; g < Kp; g += 16)
      acc = mfma4(*reinterpret_cast<const float4*>(a + g), *reinterpret_cast<const float4*>(b + g), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) epi(rt * 16 + 4 * q + r, ct * 16 + j, acc[r]);
  }
}
// out[r][c] = sum_o A[r][o] W[o][c];  A: LDS [16 RT][lda] zero-padded to Kp columns, W: LDS [No][ldw]; epi for
// c < 16 ceil(Ccols / 16), columns >= Cvalid repeat column Cvalid - 1.
template <class Epi>
__device__ __forceinline__ void lds_gemm_nn(const float* A, int lda, int RT, const float* W, int ldw, int No, int Cvalid, int Ccols, int Kp,
                                            int wslot, int lane, Epi epi) {
  const int j = lane & 15, q = lane >> 4, CT = (Ccols + 15) >> 4;
  for (int t = wslot; t < RT * CT; t += CM_NW) {
    const int rt = t / CT, ct = t - rt * CT;
    int c = ct * 16 + j; c = c < Cvalid ? c : Cvalid - 1;
    const float* a = A + (rt * 16 + j) * lda + 4 * q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int g = 0; g < Kp; g += 16) {
      const int o = g + 4 * q;
      float4 bv;
      bv.x = W[(o < No ? o : No - 1) * ldw + c];
      bv.y = W[(o + 1 < No ? o + 1 : No - 1) * ldw + c];
      bv.z = W[(o + 2 < No ? o + 2 : No - 1) * ldw + c];
      bv.w = W[(o + 3 < No ? o + 3 : No - 1) * ldw + c];
      acc = mfma4(*reinterpret_cast<const float4*>(a + g), bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) epi(rt * 16 + 4 * q + r, ct * 16 + j, acc[r]);
  }
}


// ---- one wave, one 16-row tile, all column tiles: the wave-local forms (no workgroup barrier between dependent layers;
// the caller separates a layer's LDS writes from the next layer's reads with wave_lds_fence()).  Column tiles go in
// pairs so that two independent accumulator chains share the MFMA pipe.
__device__ __forceinline__ void wave_lds_fence() {
  // LDS instructions of one wave execute in order; this only stops the compiler from moving a lane's reads above
  // other lanes' writes of the same wave (which its per-thread alias analysis cannot see)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class Epi>
__device__ __forceinline__ void wave_gemm_nt(const float* A, int lda, const float* W, int ldw, int Nrows, int Ncols, int Kp, int lane, Epi epi) {
  const int j = lane & 15, q = lane >> 4, CT = (Ncols + 15) >> 4;
  const float* a = A + j * lda + 4 * q;
  for (int ct = 0; ct < CT; ct += 2) {
    int n0 = ct * 16 + j, n1 = n0 + 16;
    n0 = n0 < Nrows ? n0 : Nrows - 1; n1 = n1 < Nrows ? n1 : Nrows - 1;
    const float* b0 = W + n0 * ldw + 4 * q;
    const float* b1 = W + n1 * ldw + 4 * q;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int g = 0; g < Kp; g += 16) {
      const float4 av = *reinterpret_cast<const float4*>(a + g);
      const float4 v0 = *reinterpret_cast<const float4*>(b0 + g), v1 = *reinterpret_cast<const float4*>(b1 + g);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, v0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, v1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, v0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, v1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, v0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, v1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, v0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, v1.w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      epi(4 * q + r, ct * 16 + j, acc0[r]);
      if (ct + 1 < CT) epi(4 * q + r, (ct + 1) * 16 + j, acc1[r]);
    }
  }
}
template <class Epi>
__device__ __forceinline__ void wave_gemm_nn(const float* A, int lda, const float* W, int ldw, int No, int Cvalid, int Ccols, int Kp, int lane,
                                             Epi epi) {
  const int j = lane & 15, q = lane >> 4, CT = (Ccols + 15) >> 4;
  const float* a = A + j * lda + 4 * q;
  for (int ct = 0; ct < CT; ct += 2) {
    int c0 = ct * 16 + j, c1 = c0 + 16;
    c0 = c0 < Cvalid ? c0 : Cvalid - 1; c1 = c1 < Cvalid ? c1 : Cvalid - 1;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int g = 0; g < Kp; g += 16) {
      const int o = g + 4 * q;
      const int o0 = (o < No ? o : No - 1) * ldw, o1 = (o + 1 < No ? o + 1 : No - 1) * ldw;
      const int o2 = (o + 2 < No ? o + 2 : No - 1) * ldw, o3 = (o + 3 < No ? o + 3 : No - 1) * ldw;
      const float4 av = *reinterpret_cast<const float4*>(a + g);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, W[o0 + c0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, W[o0 + c1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, W[o1 + c0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, W[o1 + c1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, W[o2 + c0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, W[o2 + c1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, W[o3 + c0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, W[o3 + c1], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      epi(4 * q + r, ct * 16 + j, acc0[r]);
      if (ct + 1 < CT) epi(4 * q + r, (ct + 1) * 16 + j, acc1[r]);
    }
  }
}

// ---------------------------------------------------------------------------------------- frozen critic, one 16-row tile
// (decoder_iteration, train.py:205-217: the critics only pass a gradient back to the generator)
struct CriticPad {            // padded LDS image of a critic's weights: bias of every layer in column K of its rows
  int Kin, Lp, ldin, LQ;      // padded reduction lengths (incl. the ones column) and row strides
  int w0, wh, wl, total;      // offsets (floats): [L][ldin] | [nh-1][L][LQ] | [LQ]
};
HD CriticPad critic_pad(int in_dim, int L, int nh) {
  CriticPad p;
  p.Kin = up16(in_dim + 1); p.Lp = up16(L + 1); p.ldin = p.Kin + 4; p.LQ = p.Lp + 4;
  p.w0 = 0; p.wh = L * p.ldin; p.wl = p.wh + (nh - 1) * L * p.LQ; p.total = p.wl + p.LQ;
  return p;
}
// global arena -> padded LDS image (zero padding included); whole workgroup, no barrier inside
__device__ __forceinline__ void stage_critic_padded(float* dst, const float* __restrict__ P, const CriticLayout& cl, int L, const CriticPad& cp) {
  const int in_dim = cl.in_dim, nh = cl.nh;
  for (int i = threadIdx.x; i < L * cp.ldin; i += blockDim.x) {
    const int n = i / cp.ldin, k = i - n * cp.ldin;
    dst[cp.w0 + i] = k < in_dim ? P[cl.wof(0) + n * in_dim + k] : (k == in_dim ? P[cl.bof(0) + n] : 0.f);
  }
  for (int i = threadIdx.x; i < (nh - 1) * L * cp.LQ; i += blockDim.x) {
    const int li = 1 + i / (L * cp.LQ), rem = i - (li - 1) * L * cp.LQ, n = rem / cp.LQ, k = rem - n * cp.LQ;
    dst[cp.wh + i] = k < L ? P[cl.wof(li) + n * L + k] : (k == L ? P[cl.bof(li) + n] : 0.f);
  }
  for (int k = threadIdx.x; k < cp.LQ; k += blockDim.x) dst[cp.wl + k] = k < L ? P[cl.wof(nh) + k] : (k == L ? P[cl.bof(nh)] : 0.f);
}
// scratch of one tile pass: in [16][ldin] | act [nh][16][LQ] | dm [nh][16][LQ] | dl [2][16][LQ]
HD int critic_tile_floats(const CriticPad& cp, int nh) { return 16 * cp.ldin + (2 * nh + 2) * 16 * cp.LQ; }

// Forward of 16 rows and the gradient of sum_r dout * out[r] with respect to the input rows.
//   Xs [16][ldx]: input rows (in_dim columns);  W: padded weights (stage_critic_padded);  scratch: critic_tile_floats
//   drop(li, r, c4) -> float4 / drop1(li, r, c) -> float: dropout keep-scales;  dX [16][lddx] <- d / d input;
//   returns the sum of the 16 outputs.  Starts and ends with a workgroup barrier.
template <class Drop, class Drop1>
__device__ __forceinline__ float critic_tile_fwd_bwd(const float* Xs, int ldx, const float* W, const CriticLayout& cl, int L, const CriticPad& cp,
                                                     float* scratch, float dout, Drop drop, Drop1 drop1, float* dX, int lddx) {
  const int in_dim = cl.in_dim, nh = cl.nh, ldin = cp.ldin, LQ = cp.LQ;
  float* in = scratch; float* act = in + 16 * ldin; float* dm = act + nh * 16 * LQ; float* dl = dm + nh * 16 * LQ;
  const float* w0 = W + cp.w0; const float* wh = W + cp.wh; const float* wl = W + cp.wl;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * ldin; i += blockDim.x) {
    const int r = i / ldin, c = i - r * ldin;
    in[i] = c < in_dim ? Xs[r * ldx + c] : (c == in_dim ? 1.f : 0.f);
  }
  for (int i = threadIdx.x; i < (nh + 2) * 16 * LQ; i += blockDim.x) {      // act | (dm) | dl: padding must be zero
    if (i < nh * 16 * LQ) act[i] = 0.f; else dl[i - nh * 16 * LQ] = 0.f;
  }
  if ((L & 3) == 0) {                       // four columns per thread: drop(li, r, c4) -> float4 (one Philox evaluation)
    const int L4 = L >> 2;
    for (int i = threadIdx.x; i < nh * 16 * L4; i += blockDim.x) {
      const int li = i / (16 * L4), rem = i - li * 16 * L4, r = rem / L4, c = 4 * (rem - r * L4);
      *reinterpret_cast<float4*>(dm + (li * 16 + r) * LQ + c) = drop(li, r, c);
    }
  } else {
    for (int i = threadIdx.x; i < nh * 16 * L; i += blockDim.x) {
      const int li = i / (16 * L), rem = i - li * 16 * L, r = rem / L, c = rem - r * L;
      dm[(li * 16 + r) * LQ + c] = drop1(li, r, c);
    }
  }
  __syncthreads();
  for (int li = 0; li < nh; ++li) {
    const float* A = li == 0 ? in : act + (li - 1) * 16 * LQ;
    const float* Wl = li == 0 ? w0 : wh + (li - 1) * L * LQ;
    float* ao = act + li * 16 * LQ; float* dmo = dm + li * 16 * LQ;
    lds_gemm_nt(A, li == 0 ? ldin : LQ, 1, Wl, li == 0 ? ldin : LQ, L, L + 1, li == 0 ? cp.Kin : cp.Lp, wave, lane, [&](int r, int c, float pre) {
      if (c < L) {
        const float dd = leaky_slope(pre) * dmo[r * LQ + c];
        dmo[r * LQ + c] = dd;
        ao[r * LQ + c] = pre * dd;
        if (li == nh - 1) dl[r * LQ + c] = dout * wl[c] * dd;
      } else if (c == L) {
        ao[r * LQ + c] = 1.f;
      }
    });
    __syncthreads();
  }
  float osum = 0.f;
  if (wave == CM_NW - 1) {
    float o = 0.f;
    if (lane < 16) {
      const float* x = act + ((nh - 1) * 16 + lane) * LQ;
      for (int c = 0; c <= L; ++c) o += x[c] * wl[c];
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) o += __shfl_xor(o, off, 64);
    osum = o;
  }
  float* cur = dl; float* nxt = dl + 16 * LQ;
  for (int li = nh - 2; li >= 0; --li) {
    const float* dmo = dm + li * 16 * LQ;
    lds_gemm_nn(cur, LQ, 1, wh + li * L * LQ, LQ, L, L, L, cp.Lp, wave, lane,
                [&](int r, int c, float v) { if (c < L) nxt[r * LQ + c] = v * dmo[r * LQ + c]; });
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  lds_gemm_nn(cur, LQ, 1, w0, ldin, L, in_dim, in_dim, cp.Lp, wave, lane, [&](int r, int c, float v) { if (c < in_dim) dX[r * lddx + c] = v; });
  // hand the sum to thread 0 through the (now dead) input tile
  if (wave == CM_NW - 1 && lane == 0) in[0] = osum;
  __syncthreads();
  return in[0];
}

// Forward only (eval mode, no dropout): out[r] for 16 rows, written to `outv` (LDS, 16 floats).  scratch: in [16][ldin] |
// act [2][16][LQ] (ping-pong).  Starts and ends with a workgroup barrier.
__device__ __forceinline__ void critic_tile_fwd(const float* Xs, int ldx, const float* W, const CriticLayout& cl, int L, const CriticPad& cp,
                                                float* scratch, float* outv) {
  const int in_dim = cl.in_dim, nh = cl.nh, ldin = cp.ldin, LQ = cp.LQ;
  float* in = scratch; float* act = in + 16 * ldin;
  const float* w0 = W + cp.w0; const float* wh = W + cp.wh; const float* wl = W + cp.wl;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * ldin; i += blockDim.x) {
    const int r = i / ldin, c = i - r * ldin;
    in[i] = c < in_dim ? Xs[r * ldx + c] : (c == in_dim ? 1.f : 0.f);
  }
  for (int i = threadIdx.x; i < 2 * 16 * LQ; i += blockDim.x) act[i] = 0.f;
  __syncthreads();
  for (int li = 0; li < nh; ++li) {
    const float* A = li == 0 ? in : act + ((li - 1) & 1) * 16 * LQ;
    const float* Wl = li == 0 ? w0 : wh + (li - 1) * L * LQ;
    float* ao = act + (li & 1) * 16 * LQ;
    lds_gemm_nt(A, li == 0 ? ldin : LQ, 1, Wl, li == 0 ? ldin : LQ, L, L + 1, li == 0 ? cp.Kin : cp.Lp, wave, lane, [&](int r, int c, float pre) {
      if (c < L) ao[r * LQ + c] = pre * leaky_slope(pre);
      else if (c == L) ao[r * LQ + c] = 1.f;
    });
    __syncthreads();
  }
  if (threadIdx.x < 16) {
    const float* x = act + ((nh - 1) & 1) * 16 * LQ + threadIdx.x * LQ;
    float o = 0.f;
    for (int c = 0; c <= L; ++c) o += x[c] * wl[c];
    outv[threadIdx.x] = o;
  }
  __syncthreads();
}

}  // namespace hypad
