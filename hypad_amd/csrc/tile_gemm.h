// Row-tile GEMMs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// A workgroup owns MT*16 rows of activations in LDS and streams weights from global memory (L2-resident,
// ~1 MB per model).  Two shapes cover every layer of the TadGAN networks:
//   gemm_nt :  Y[r][n] = sum_k X[r][k] * W[wrow(n)][k]  (+ bias)      forward of Linear / LSTM gates
//   gemm_nn :  Y[r][c] = sum_n D[r][n] * W[wrow(n)][c]                 backward-data
// wrow(n) = n + (n >= split ? gap : 0) lets the LSTM layers skip the f-gate block of W_ih (never used at
// seq_len 1 with c0 = 0: SURVEY.md A.2) while keeping PyTorch's [i,f,g,o] weight layout.
//
// MFMA 16x16x4 f32 lane map (cdna_hip_programming.md §3): lane l, j = l & 15, q = l >> 4
//   A[i = j][k = q],  B[k = q][col = j],  D[row = 4*q + reg][col = j].
// The reduction index may be permuted freely as long as A and B agree, which is what the 16-byte path does:
// one float4 per lane feeds four consecutive MFMAs (lane (.,q) carries k = k0 + 4q + i at step i).
#pragma once
#include "device_utils.h"

namespace hypad {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct RowMap {
  int split, gap;
  __device__ __forceinline__ int operator()(int n) const { return n + (n >= split ? gap : 0); }
};
__device__ __forceinline__ RowMap identity_map() { return RowMap{0x7fffffff, 0}; }
// compact (i,g,o) gate column c in [0,3H) -> row of the (4H, in) PyTorch weight
__device__ __forceinline__ RowMap lstm_gate_map(int H) { return RowMap{H, H}; }

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Y (LDS) [MT*16][ldy], columns ycol0 .. ycol0+N-1.  X (LDS) [MT*16][ldx], K columns, ldx % 4 == 0.
// W (global) rows of length ldw; bias pointers (global, indexed like W rows) may be null.
template <int MT>
__device__ void gemm_nt(const float* __restrict__ Xs, int ldx, const float* __restrict__ W, int ldw, int K, int N,
                        RowMap map, const float* __restrict__ bias0, const float* __restrict__ bias1,
                        float* __restrict__ Ys, int ldy, int ycol0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int ntiles = (N + 15) >> 4;
  const bool vec = ((K & 3) == 0) && ((ldw & 3) == 0) && aligned16(W);
  for (int t = wave; t < ntiles; t += nwaves) {
    const int n = t * 16 + j;
    const bool nv = n < N;
    const int wr = nv ? map(n) : 0;
    const float* __restrict__ wp = W + (size_t)wr * ldw;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (vec) {
#pragma unroll 2
      for (int k0 = 0; k0 < K; k0 += 16) {
        const int kk = k0 + 4 * q;
        const bool kv = kk < K;
        float4 b = (kv && nv) ? *reinterpret_cast<const float4*>(wp + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          float4 a = kv ? *reinterpret_cast<const float4*>(Xs + (m * 16 + j) * ldx + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[m], 0, 0, 0);
        }
      }
    } else {
#pragma unroll 4
      for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + q;
        const bool kv = k < K;
        const float b = (kv && nv) ? wp[k] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float a = kv ? Xs[(m * 16 + j) * ldx + k] : 0.f;
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        }
      }
    }
    if (nv) {
      float bsum = 0.f;
      if (bias0) bsum += bias0[wr];
      if (bias1) bsum += bias1[wr];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ys[(m * 16 + 4 * q + r) * ldy + ycol0 + n] = acc[m][r] + bsum;
    }
  }
}

// Y (LDS) [MT*16][ldy] columns 0..C-1  (+)= D (LDS) [MT*16][ldd] columns dcol0..dcol0+Nred-1  times  W rows wrow(n).
template <int MT>
__device__ void gemm_nn(const float* __restrict__ Ds, int ldd, int dcol0, const float* __restrict__ W, int ldw,
                        int Nred, RowMap map, int C, float* __restrict__ Ys, int ldy, bool accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int ntiles = (C + 15) >> 4;
  for (int t = wave; t < ntiles; t += nwaves) {
    const int c = t * 16 + j;
    const bool cv = c < C;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int n0 = 0; n0 < Nred; n0 += 4) {
      const int n = n0 + q;
      const bool nv = n < Nred;
      const float b = (nv && cv) ? W[(size_t)map(n) * ldw + c] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float a = nv ? Ds[(m * 16 + j) * ldd + dcol0 + n] : 0.f;
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
      }
    }
    if (cv) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* y = Ys + (m * 16 + 4 * q + r) * ldy + c;
          *y = accumulate ? *y + acc[m][r] : acc[m][r];
        }
    }
  }
}

}  // namespace hypad
