// Diagnostic build of the decoder forward with per-phase clock stamps (development aid, not on the product path):
// answers "where does a 16-row tile spend its time" -- shader cycles (s_memtime) and 100 MHz wall ticks per phase.
#include <hip/hip_runtime.h>

#include "../../include/hypad.h"
#include "nets.h"

using namespace hypad;

namespace {

__device__ __forceinline__ void stamp(long long* out, int& k) {
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * k] = (long long)__builtin_amdgcn_s_memtime();
    out[2 * k + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  ++k;
}

template <int MT>
__global__ void diag_decoder_kernel(const float* __restrict__ P, const float* __restrict__ z, float* __restrict__ hyper,
                                    int64_t rows, int S, int L, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = MT * 16;
  const int ldS = lds_stride(S);
  const int per_row = ldS > 6 * DEC_H + 4 ? ldS : 6 * DEC_H + 4;
  float* zs = smem;
  float* bufA = zs + R * LP;
  float* bufB = bufA + R * per_row;
  float* wst = bufB + R * per_row;
  const DecLayout dl = dec_layout(S, L, 1);
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int valid = (int)min((int64_t)R, rows - r0);
  long long* st = stamps + (size_t)blockIdx.x * 64;
  int k = 0;
  constexpr int ldA0 = 52, ldG = 6 * DEC_H + 4, ldH = 2 * DEC_H + 4;
  for (int rep = 0; rep < 2; ++rep) {
  stamp(st, k);                                                                                         // 0 start
  tile_load(zs, LP, z + r0 * L, L, R, L, valid);
  stamp(st, k);                                                                                         // 1 z loaded
  gemm_nt<MT>(zs, LP, P + dl.d1_w, L, L, DEC_D1, identity_map(), P + dl.d1_b, nullptr, bufB, ldA0, 0, wst);
  stamp(st, k);                                                                                         // 2 dense1
  lstm_gates_tile<MT>(bufB, ldA0, DEC_D1, P, dl.l[0][0], dl.l[0][1], DEC_H, bufA, ldG, wst);
  stamp(st, k);                                                                                         // 3 l0 gates (K=50 scalar path)
  lstm_cell_tile(bufA, ldG, DEC_H, R, bufB, ldH, nullptr, valid, 16);
  stamp(st, k);                                                                                         // 4 l0 cell
  lstm_gates_tile<MT>(bufB, ldH, 2 * DEC_H, P, dl.l[1][0], dl.l[1][1], DEC_H, bufA, ldG, wst);
  stamp(st, k);                                                                                         // 5 l1 gates (K=128 vec)
  lstm_cell_tile(bufA, ldG, DEC_H, R, bufB, ldH, nullptr, valid, 16);
  stamp(st, k);                                                                                         // 6 l1 cell
  gemm_nt<MT>(bufB, ldH, P + dl.d2_w, 2 * DEC_H, 2 * DEC_H, S, identity_map(), P + dl.d2_b, nullptr, bufA, ldS, 0, wst);
  stamp(st, k);                                                                                         // 7 dense2
  tile_for(R, S, [&](int r, int c) { bufA[r * ldS + c] = tanhf_(bufA[r * ldS + c]); });
  stamp(st, k);                                                                                         // 8 tanh
  gemm_nt<MT>(bufA, ldS, P + dl.head_w, S, S, S, identity_map(), nullptr, nullptr, bufB, ldS, 0, wst);
  stamp(st, k);                                                                                         // 9 head gemm
  head_rows_tile(bufB, ldS, R, S, P + dl.head_b);
  stamp(st, k);                                                                                         // 10 head rows
  tile_store(hyper + r0 * S, S, bufB, ldS, R, S, valid);
  stamp(st, k);                                                                                         // 11 store
  }
}

// the same decoder forward with MFMA-native packed weights (gemm_nt_packed): pk = packed arena, off[] = float offsets of
// {d1 W, d1 b, l0d0 W, l0d0 b, l0d1 W, l0d1 b, l1d0 W, l1d0 b, l1d1 W, l1d1 b, d2 W, d2 b, head W}
struct PackedOffs { int o[13]; };
__global__ void diag_decoder_packed_kernel(const float* __restrict__ pk, PackedOffs po, const float* __restrict__ head_b,
                                           const float* __restrict__ z, float* __restrict__ hyper, int64_t rows, int S, int L,
                                           long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16;
  const int ldS = lds_stride(S);
  const int per_row = ldS > 6 * DEC_H + 4 ? ldS : 6 * DEC_H + 4;
  float* zs = smem;
  float* bufA = zs + R * LP;
  float* bufB = bufA + R * per_row;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int valid = (int)min((int64_t)R, rows - r0);
  long long* st = stamps + (size_t)blockIdx.x * 64;
  int k = 0;
  constexpr int ldA0 = 52, ldG = 6 * DEC_H + 4, ldH = 2 * DEC_H + 4;
  for (int rep = 0; rep < 2; ++rep) {
  stamp(st, k);
  tile_load(zs, LP, z + r0 * L, L, R, L, valid);
  stamp(st, k);
  gemm_nt_packed<1>(zs, LP, L, DEC_D1, pk + po.o[0], pk + po.o[1], bufB, ldA0, 0);
  stamp(st, k);
  gemm_nt_packed<1>(bufB, ldA0, DEC_D1, 3 * DEC_H, pk + po.o[2], pk + po.o[3], bufA, ldG, 0);
  gemm_nt_packed<1>(bufB, ldA0, DEC_D1, 3 * DEC_H, pk + po.o[4], pk + po.o[5], bufA, ldG, 3 * DEC_H, (3 * DEC_H + 15) >> 4);
  stamp(st, k);
  lstm_cell_tile(bufA, ldG, DEC_H, R, bufB, ldH, nullptr, valid, 16);
  stamp(st, k);
  gemm_nt_packed<1>(bufB, ldH, 2 * DEC_H, 3 * DEC_H, pk + po.o[6], pk + po.o[7], bufA, ldG, 0);
  gemm_nt_packed<1>(bufB, ldH, 2 * DEC_H, 3 * DEC_H, pk + po.o[8], pk + po.o[9], bufA, ldG, 3 * DEC_H, (3 * DEC_H + 15) >> 4);
  stamp(st, k);
  lstm_cell_tile(bufA, ldG, DEC_H, R, bufB, ldH, nullptr, valid, 16);
  stamp(st, k);
  gemm_nt_packed<1>(bufB, ldH, 2 * DEC_H, S, pk + po.o[10], pk + po.o[11], bufA, ldS, 0);
  stamp(st, k);
  tile_for(R, S, [&](int r, int c) { bufA[r * ldS + c] = tanhf_(bufA[r * ldS + c]); });
  stamp(st, k);
  gemm_nt_packed<1>(bufA, ldS, S, S, pk + po.o[12], nullptr, bufB, ldS, 0);
  stamp(st, k);
  head_rows_tile(bufB, ldS, R, S, head_b);
  stamp(st, k);
  tile_store(hyper + r0 * S, S, bufB, ldS, R, S, valid);
  stamp(st, k);
  }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int hypad_diag_decoder_packed(const float* pk, const int* offs, const float* head_b, const float* z, float* hyper, int64_t rows,
                                         int S, int L, long long* stamps, void* stream) {
  PackedOffs po;
  for (int i = 0; i < 13; ++i) po.o[i] = offs[i];
  const int ldS = lds_stride(S);
  const int per_row = ldS > 6 * DEC_H + 4 ? ldS : 6 * DEC_H + 4;
  const size_t lds = (size_t)(16 * LP + 2 * 16 * per_row) * sizeof(float);
  const int nblk = (int)((rows + 15) / 16);
  hipLaunchKernelGGL(diag_decoder_packed_kernel, dim3(nblk), dim3(512), lds, (hipStream_t)stream, pk, po, head_b, z, hyper, rows, S, L, stamps);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int hypad_diag_decoder_timeline(const float* P, const float* z, float* hyper, int64_t rows, int S, int L, int mt,
                                           int threads, long long* stamps, hypad_stream_t s) {
  const int R = mt * 16;
  const int ldS = lds_stride(S);
  const int per_row = ldS > 6 * DEC_H + 4 ? ldS : 6 * DEC_H + 4;
  size_t lds = (size_t)(R * LP + 2 * R * per_row + 16 * WSTAGE_FLOATS) * sizeof(float);
  int blocks = (int)((rows + R - 1) / R);
  if (mt == 1) {
    (void)hipFuncSetAttribute((const void*)diag_decoder_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(diag_decoder_kernel<1>, dim3(blocks), dim3(threads), lds, (hipStream_t)s, P, z, hyper, rows, S, L, stamps);
  } else {
    (void)hipFuncSetAttribute((const void*)diag_decoder_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(diag_decoder_kernel<2>, dim3(blocks), dim3(threads), lds, (hipStream_t)s, P, z, hyper, rows, S, L, stamps);
  }
  return (int)hipGetLastError();
}

// ---- MFMA issue-rate microbenchmark: cycles per instruction for dependent / independent accumulator chains
namespace {
template <int NACC>
__global__ void diag_mfma_kernel(long long* out, float seed) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}
__global__ void diag_mfma32_kernel(long long* out, float seed) {
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = seed;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)(acc[0][0] + acc[1][0]); }
}
// co-issue: wave 0 runs fp32 MFMAs, wave 4 (same SIMD of a 512-thread workgroup) runs VALU work; each reports its own time.
// mode bit 0: wave 0 active (MFMA); bit 1: wave 4 active; bits 2-3: what wave 4 runs (0 v_fma_f32, 1 v_exp_f32, 2 MFMA, 3 ds_read)
template <int kind, int NV = 0>
__global__ void diag_coissue_kernel(long long* out, float seed, int mode, int w2) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = seed; lds[threadIdx.x + 512] = seed;
  __syncthreads();
  if ((wave == 0 && (mode & 1)) || (wave == w2 && (mode & 2) && kind == 2)) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    long long t0 = __builtin_amdgcn_s_memtime();
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
          for (int v = 0; v < NV; ++v) {      // the same wave interleaves NV VALU instructions behind every MFMA
            if (kind == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[v & 7]));
            else asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[v & 7]) : "v"(a), "v"(b));
          }
        }
    }
    acc[0][1] += x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { out[wave == 0 ? 0 : 1] = t1 - t0; out[2 + wave] = (long long)(acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + acc[0][1]); }
  } else if (wave == w2 && (mode & 2)) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    float a = 1.0001f, b = 0.5f;
    if (mode & 16) __builtin_amdgcn_s_setprio(3);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 128; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (kind == 0) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        else if (kind == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        else { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((int)((threadIdx.x & 63) * 4 + i * 256))); x[i] = v; }
      }
      if (kind == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += x[i];
    if ((threadIdx.x & 63) == 0) { out[1] = t1 - t0; out[2 + wave] = (long long)sum; }
  }
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int hypad_diag_coissue(int mode, int w2, long long* out, hypad_stream_t s) {
  const int kind = (mode >> 2) & 3;
  if (mode & 32) {
#define CO(K, N) hipLaunchKernelGGL((diag_coissue_kernel<K, N>), dim3(1), dim3(512), 0, (hipStream_t)s, out, 1.0f, mode, 4)
    if (kind == 0) { if (w2 == 0) CO(0, 0); else if (w2 == 1) CO(0, 1); else if (w2 == 2) CO(0, 2); else if (w2 == 4) CO(0, 4); else if (w2 == 6) CO(0, 6); else CO(0, 8); }
    else { if (w2 == 0) CO(1, 0); else if (w2 == 1) CO(1, 1); else if (w2 == 2) CO(1, 2); else if (w2 == 4) CO(1, 4); else if (w2 == 6) CO(1, 6); else CO(1, 8); }
#undef CO
    return (int)hipGetLastError();
  }
  if (kind == 0) hipLaunchKernelGGL(diag_coissue_kernel<0>, dim3(1), dim3(512), 0, (hipStream_t)s, out, 1.0f, mode, w2);
  else if (kind == 1) hipLaunchKernelGGL(diag_coissue_kernel<1>, dim3(1), dim3(512), 0, (hipStream_t)s, out, 1.0f, mode, w2);
  else if (kind == 2) hipLaunchKernelGGL(diag_coissue_kernel<2>, dim3(1), dim3(512), 0, (hipStream_t)s, out, 1.0f, mode, w2);
  else hipLaunchKernelGGL(diag_coissue_kernel<3>, dim3(1), dim3(512), 0, (hipStream_t)s, out, 1.0f, mode, w2);
  return (int)hipGetLastError();
}
extern "C" __attribute__((visibility("default"))) int hypad_diag_mfma(int variant, int threads, long long* out, hypad_stream_t s) {
  if (variant == 1) hipLaunchKernelGGL(diag_mfma_kernel<1>, dim3(1), dim3(threads), 0, (hipStream_t)s, out, 1.0f);
  else if (variant == 2) hipLaunchKernelGGL(diag_mfma_kernel<2>, dim3(1), dim3(threads), 0, (hipStream_t)s, out, 1.0f);
  else if (variant == 4) hipLaunchKernelGGL(diag_mfma_kernel<4>, dim3(1), dim3(threads), 0, (hipStream_t)s, out, 1.0f);
  else hipLaunchKernelGGL(diag_mfma32_kernel, dim3(1), dim3(threads), 0, (hipStream_t)s, out, 1.0f);
  return (int)hipGetLastError();
}

// ---- weight-tile load latency: the exact access pattern of nt_tile (16 rows x 8 float4 per lane), timed per tile
namespace {
__global__ void diag_load_kernel(const float* __restrict__ W, int ldw, int ntiles, int mode, long long* out, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  float acc = 0.f;
  for (int rep = 0; rep < 2; ++rep)
    for (int t = 0; t < ntiles; ++t) {
      const float* wp = W + (size_t)((wave * ntiles + t) * 16 + j) * ldw;
      long long t0 = __builtin_amdgcn_s_memtime();
      if (mode == 0) {
        float4 b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(wp + 16 * u + 4 * q);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += b[u].x + b[u].y + b[u].z + b[u].w;
      } else {   // one row per lane-quad, fully contiguous 1 KiB per instruction
        float4 b[8];
        const float* cp = W + (size_t)(wave * ntiles + t) * 16 * ldw;
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(cp + u * 256 + lane * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += b[u].x + b[u].y + b[u].z + b[u].w;
      }
      asm volatile("" :: "v"(acc));
      long long t1 = __builtin_amdgcn_s_memtime();
      if (lane == 0) out[(rep * 16 + wave) * 32 + t] = t1 - t0;
    }
  if (acc == 123.456f) sink[0] = acc;
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int hypad_diag_load(const float* W, int ldw, int ntiles, int mode, int threads, long long* out, float* sink, hypad_stream_t s) {
  hipLaunchKernelGGL(diag_load_kernel, dim3(1), dim3(threads), 0, (hipStream_t)s, W, ldw, ntiles, mode, out, sink);
  return (int)hipGetLastError();
}

// ---- gemm_nt / gemm_nn scaling probe: cycles for one call as a function of (K, N, MT, threads)
namespace {
template <int MT, int KIND>
__global__ void diag_gemm_kernel(const float* __restrict__ W, int K, int N, long long* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = pad4(K > N ? K : N) + 4;
  float* xs = smem;
  float* ys = xs + MT * 16 * ldx;
  float* wst = ys + MT * 16 * ldx;
  for (int i = threadIdx.x; i < MT * 16 * ldx; i += blockDim.x) xs[i] = 0.001f * (i % 97);
  __syncthreads();
  long long t[5];
  for (int rep = 0; rep < 4; ++rep) {
    __syncthreads();
    t[rep] = __builtin_amdgcn_s_memtime();
    if (KIND == 0) gemm_nt<MT>(xs, ldx, W, K, K, N, identity_map(), nullptr, nullptr, ys, ldx, 0, wst);
    else gemm_nn<MT>(xs, ldx, 0, W, K, N, identity_map(), K, ys, ldx, false);
    __syncthreads();
  }
  t[4] = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t[1] - t[0]; out[1] = t[4] - t[3]; out[2] = (long long)ys[0]; }
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int hypad_diag_gemm(const float* W, int K, int N, int mt, int kind, int threads, long long* out, hypad_stream_t s) {
  const int ldx = pad4(K > N ? K : N) + 4;
  size_t lds = (size_t)(2 * mt * 16 * ldx + 16 * WSTAGE_FLOATS) * sizeof(float);
#define LAUNCH(MT, KIND)                                                                                              \
  (void)hipFuncSetAttribute((const void*)diag_gemm_kernel<MT, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  hipLaunchKernelGGL((diag_gemm_kernel<MT, KIND>), dim3(1), dim3(threads), lds, (hipStream_t)s, W, K, N, out)
  if (mt == 1 && kind == 0) { LAUNCH(1, 0); } else if (mt == 2 && kind == 0) { LAUNCH(2, 0); }
  else if (mt == 1) { LAUNCH(1, 1); } else { LAUNCH(2, 1); }
  return (int)hipGetLastError();
}

// ---- four-row tiles on v_mfma_f32_4x4x1 against sixteen-row tiles on 16x16x4 over the same packed weights (scripts/diag_gemm4.py):
// cycles of one product call of a 512-thread workgroup, and both results (rows 0..3 must agree)
namespace {
// The same product for a FOUR-row tile, on v_mfma_f32_4x4x1_16B_f32 (sixteen independent 4 x 4 outer products per instruction, 8 cycles:
// the same 64 FLOP / clock / SIMD as 16x16x4) over the SAME packed blocks: lane (j, q) of block (tn, g) holds W[16 tn + j][16 g + 4 q .. + 3].
// MFMA block b = lane >> 2 = 4 q + (j >> 2) multiplies A[row = lane & 3][k] (taken from the lanes 4 b .. 4 b + 3: every lane loads its
// own row lane & 3 at ITS k = 16 g + 4 q + s) with W[16 tn + j][k]: the four lane groups q accumulate the k classes 4 q .. 4 q + 3 (mod 16)
// of the same 16 columns, and two swap-and-add steps (v_permlane16_swap / v_permlane32_swap) fold them: afterwards every lane (q, j)
// holds column 16 tn + j of all four rows, and takes row q into the epilogue -- one element per lane and tile.
// A latency-chain workgroup with 4 rows instead of 16 has a quarter of the MFMA cycles and a quarter of the epilogue elements -- but the
// same weight bytes to pull through its CU, and that is what the probe found to bound it: K = 128, N = 384 (196 KB of weights) takes
// 9.4 k cycles with four rows against 14.9 k with sixteen in this un-prefetched form = 21 bytes per clock; the generator kernel's
// prefetched form of that product already runs at 32 bytes per clock (6.1 k cycles), the rate the guide measures for L2-served rows
// per CU.  Rows are the only split of a chain that needs no exchange between workgroups, and it does not shrink a CU's weight bytes:
// a chain's ~1 MB of packed weights is ~33 k cycles of streaming whatever its tile height.  Kept here as the measurement, not used.
__device__ __forceinline__ float fold_q(float x) {            // sum over the four lane groups q (lanes l, l ^ 16, l ^ 32, l ^ 48)
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = __uint_as_float(a.x) + __uint_as_float(a.y);
  u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return __uint_as_float(b.x) + __uint_as_float(b.y);
}
// epi.prefetch4(n, row, ok) before the reduction, epi.emit4(row, n, value) after it: row = q in 0..3, n = output column.
template <bool PRE, class Epi, bool SC1 = false>
__device__ __forceinline__ void gemm4_nt_packed_epi(const float* __restrict__ Xs, int ldx, int K, int N, const float* __restrict__ Wp,
                                                    const float* __restrict__ bsum, int wave_rot, const PackedPre& pre, Epi& epi) {
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6, wave = (wave_id() + nwaves - wave_rot % nwaves) % nwaves;
  const int j = lane & 15, q = lane >> 4, arow = lane & 3;
  const int ntiles = (N + 15) >> 4, kg = (K + 15) >> 4;
  auto run = [&](int t, auto first_from_pre) __attribute__((always_inline)) {
    const WeightBlocks<SC1> wb(Wp, lane);
    const int b0 = t * kg;
    const int n = t * 16 + j;
    const float bs = (bsum && n < N) ? weight_scalar<SC1>(bsum + n) : 0.f;
    epi.prefetch4(n, q, n < N);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    auto consume = [&](const float4 (&w)[8], int g0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (g0 + u < kg) {                             // wave-uniform
          const int k0 = 16 * (g0 + u) + 4 * q;
          const int ka = k0 < ldx - 4 ? k0 : ldx - 4;
          const float4 a = *reinterpret_cast<const float4*>(Xs + arow * ldx + ka);
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(k0 < K ? a.x : 0.f, w[u].x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(k0 + 1 < K ? a.y : 0.f, w[u].y, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(k0 + 2 < K ? a.z : 0.f, w[u].z, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(k0 + 3 < K ? a.w : 0.f, w[u].w, acc2, 0, 0, 0);
        }
      }
    };
    int gbeg = 0;
    mfma_prio_begin<PRE>();
    if constexpr (decltype(first_from_pre)::value) {
      consume(pre.w, 0);
      gbeg = 8;
    }
    for (int g0 = gbeg; g0 < kg; g0 += 8) {
      float4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = wb(b0 + (g0 + u < kg ? g0 + u : kg - 1));
      __builtin_amdgcn_sched_barrier(0);
      consume(w, g0);
    }
    mfma_prio_end<PRE>();
    // this lane's row: q (fold the four k classes, then pick)
    const float v0 = fold_q(acc[0] + acc2[0]), v1 = fold_q(acc[1] + acc2[1]), v2 = fold_q(acc[2] + acc2[2]), v3 = fold_q(acc[3] + acc2[3]);
    const float v = q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3;
    if (n < N) epi.emit4(q, n, v + bs);
  };
  if (wave < ntiles) run(wave, std::integral_constant<bool, PRE>{});
  for (int t = wave + nwaves; t < ntiles; t += nwaves) run(t, std::false_type{});
}
template <class Act>
struct PlainEpi4 {
  float* Ys; int ldy, ycol0; Act act;
  __device__ __forceinline__ void prefetch4(int, int, bool) {}
  __device__ __forceinline__ void emit4(int row, int n, float v) { Ys[row * ldy + ycol0 + n] = act(v); }
};

__global__ void diag_gemm4_kernel(const float* __restrict__ Wp, const float* __restrict__ X, int K, int N, float* __restrict__ Y16, float* __restrict__ Y4,
                                  long long* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = lds_stride(K), ldy = lds_stride(N);
  float* xs = smem; float* ys = xs + 16 * ldx; float* y4 = ys + 16 * ldy;
  for (int i = threadIdx.x; i < 16 * ldx; i += blockDim.x) { const int r = i / ldx, c = i - r * ldx; xs[i] = c < K ? X[r * K + c] : 0.f; }
  __syncthreads();
  long long t[9];
  for (int rep = 0; rep < 4; ++rep) {
    __syncthreads();
    t[rep] = __builtin_amdgcn_s_memtime();
    gemm_nt_packed<1>(xs, ldx, K, N, Wp, nullptr, ys, ldy, 0);
    __syncthreads();
  }
  t[4] = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < 4; ++rep) {
    __syncthreads();
    t[5 + rep] = __builtin_amdgcn_s_memtime();
    PlainEpi4<ActIdentity> epi{y4, ldy, 0, ActIdentity{}};
    gemm4_nt_packed_epi<false, PlainEpi4<ActIdentity>, false>(xs, ldx, K, N, Wp, nullptr, 0, PackedPre{}, epi);
    __syncthreads();
  }
  const long long t9 = __builtin_amdgcn_s_memtime();
  for (int i = threadIdx.x; i < 16 * N; i += blockDim.x) Y16[i] = ys[(i / N) * ldy + i % N];
  for (int i = threadIdx.x; i < 4 * N; i += blockDim.x) Y4[i] = y4[(i / N) * ldy + i % N];
  if (threadIdx.x == 0) { out[0] = t[1] - t[0]; out[1] = t[4] - t[3]; out[2] = t[6] - t[5]; out[3] = t9 - t[8]; }
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int hypad_diag_gemm4(const float* Wp, const float* X, int K, int N, float* Y16, float* Y4, long long* out, hypad_stream_t s) {
  const size_t lds = (size_t)(16 * lds_stride(K) + 20 * lds_stride(N)) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)diag_gemm4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(diag_gemm4_kernel, dim3(1), dim3(512), lds, (hipStream_t)s, Wp, X, K, N, Y16, Y4, out);
  return (int)hipGetLastError();
}

// ---- ablation of one 16x128 weight tile (1 tile per wave): which part of the per-tile chain costs the time?
namespace {
template <int MODE>
__global__ void diag_tile_kernel(const float* __restrict__ W, long long* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int K = 128, ldx = 132;
  float* xs = smem;
  float* stage = xs + 16 * ldx + (threadIdx.x >> 6) * WSTAGE_FLOATS;
  for (int i = threadIdx.x; i < 16 * ldx; i += blockDim.x) xs[i] = 0.001f * (i % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const float* wp = W + (size_t)(wave * 16 + j) * K;
  long long t0 = 0, t1 = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int rep = 0; rep < 3; ++rep) {
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {                       // MFMA only, operands in registers
      float a = xs[lane], b = xs[lane + 64];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc2, 0, 0, 0);
      }
    } else if (MODE == 1) {                // direct fragment loads (8 x float4), A from LDS, no select
      float4 b[8], a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { b[u] = *reinterpret_cast<const float4*>(wp + 16 * u + 4 * q); a[u] = *reinterpret_cast<const float4*>(xs + j * ldx + 16 * u + 4 * q); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc2, 0, 0, 0);
      }
    } else if (MODE == 2) {                // only the 8 direct loads + a dependent add (no MFMA)
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(wp + 16 * u + 4 * q);
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += b[u].x + b[u].y + b[u].z + b[u].w;
      acc[0] += s;
    } else if (MODE == 3) {                // A reads from LDS + MFMA, B constant
      float4 a[8];
      float b = xs[lane];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(xs + j * ldx + 16 * u + 4 * q);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b, acc2, 0, 0, 0);
      }
    } else {                               // MODE 4: contiguous loads (8 rows x 128 B) -> LDS slab -> fragments -> MFMA
      const int lrow = lane >> 3, lcol = 4 * (lane & 7);
      float4 w[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) w[s][i] = *reinterpret_cast<const float4*>(W + (size_t)(wave * 16 + lrow + 8 * i) * K + 32 * s + lcol);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(stage + (lrow + 8 * i) * WSTAGE_LD + lcol) = w[s][i];
        float4 b[2], a[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { b[u] = *reinterpret_cast<const float4*>(stage + j * WSTAGE_LD + 16 * u + 4 * q); a[u] = *reinterpret_cast<const float4*>(xs + j * ldx + 32 * s + 16 * u + 4 * q); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc2, 0, 0, 0);
        }
      }
    }
    asm volatile("" :: "v"(acc[0]), "v"(acc2[0]));
    t1 = __builtin_amdgcn_s_memtime();
  }
  if (lane == 0) out[wave] = t1 - t0;
  if (acc[0] + acc2[0] == 123.456f) out[63] = 1;
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int hypad_diag_tile(const float* W, int mode, int threads, long long* out, hypad_stream_t s) {
  size_t lds = (size_t)(16 * 132 + 16 * WSTAGE_FLOATS) * sizeof(float);
  switch (mode) {
    case 0: hipLaunchKernelGGL(diag_tile_kernel<0>, dim3(1), dim3(threads), lds, (hipStream_t)s, W, out); break;
    case 1: hipLaunchKernelGGL(diag_tile_kernel<1>, dim3(1), dim3(threads), lds, (hipStream_t)s, W, out); break;
    case 2: hipLaunchKernelGGL(diag_tile_kernel<2>, dim3(1), dim3(threads), lds, (hipStream_t)s, W, out); break;
    case 3: hipLaunchKernelGGL(diag_tile_kernel<3>, dim3(1), dim3(threads), lds, (hipStream_t)s, W, out); break;
    default: hipLaunchKernelGGL(diag_tile_kernel<4>, dim3(1), dim3(threads), lds, (hipStream_t)s, W, out); break;
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- store16: 16-byte buffer stores vs rewriting
// their data registers.  Round 2 met intermittently lost / mixed values behind `buffer_store_dwordx4 ... s_off offen` whose data
// registers were rewritten a few instructions later (tile_gemm.h GBuf::st4).  This isolates it: a wave stores four registers
// holding a known pattern and overwrites those registers NOPS wait states later, all inside one asm block so that no compiler
// scheduling or hazard handling takes part; a second kernel counts the 16-byte slots that do not hold the pattern.
//   FORM 0: row offset in a scalar register (`s_off offen`)        FORM 1: offset folded into the vector offset (`0 offen`)
//   FORM 2: FORM 0 with the scalar register written by s_mov right before the store (SGPR written by SALU -> VMEM read)
//   NOPS 0: the overwrite is the next instruction; k > 0: `s_nop k-1` in between.
namespace {
__device__ __forceinline__ unsigned store16_pattern(unsigned slot, unsigned e) { return (slot * 2654435761u) ^ (0x9e3779b9u * (e + 1u)); }

template <int FORM, int NOPS>
__global__ __launch_bounds__(256) void diag_store16_kernel(unsigned* __restrict__ buf, int iters) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  const unsigned per_iter = gridDim.x * 256 * 16;                            // bytes of one iteration's slots
  for (int it = 0; it < iters; ++it) {
    const unsigned slot = (unsigned)it * (gridDim.x * 256) + tid;
    const unsigned a = store16_pattern(slot, 0), b = store16_pattern(slot, 1), c = store16_pattern(slot, 2), d = store16_pattern(slot, 3);
    const unsigned so = (unsigned)it * per_iter;                             // wave-uniform row offset
    const unsigned vo_lane = tid * 16, vo_full = vo_lane + so;
    const unsigned junk = ~a;
#define ST16_NOP(N) "s_nop " #N "\n\t"
#define ST16_BODY(STORE, NOPSTR)                                                                                             \
    asm volatile("v_mov_b32 v28, %[a]\n\tv_mov_b32 v29, %[b]\n\tv_mov_b32 v30, %[c]\n\tv_mov_b32 v31, %[d]\n\t"             \
                 "s_nop 4\n\t" STORE NOPSTR                                                                                   \
                 "v_mov_b32 v28, %[j]\n\tv_mov_b32 v29, %[j]\n\tv_mov_b32 v30, %[j]\n\tv_mov_b32 v31, %[j]\n\t"             \
                 : : [a] "v"(a), [b] "v"(b), [c] "v"(c), [d] "v"(d), [j] "v"(junk), [vl] "v"(vo_lane), [vf] "v"(vo_full), [rs] "s"(rs), [so] "s"(so) \
                 : "v28", "v29", "v30", "v31", "s33", "memory")
#define ST16_FORM(NOPSTR)                                                                                                    \
    do {                                                                                                                     \
      if constexpr (FORM == 0) ST16_BODY("buffer_store_dwordx4 v[28:31], %[vl], %[rs], %[so] offen\n\t", NOPSTR);          \
      else if constexpr (FORM == 1) ST16_BODY("buffer_store_dwordx4 v[28:31], %[vf], %[rs], 0 offen\n\t", NOPSTR);          \
      else ST16_BODY("s_mov_b32 s33, %[so]\n\tbuffer_store_dwordx4 v[28:31], %[vl], %[rs], s33 offen\n\t", NOPSTR);        \
    } while (0)
    if constexpr (NOPS == 0) ST16_FORM("");
    else if constexpr (NOPS == 1) ST16_FORM(ST16_NOP(0));
    else if constexpr (NOPS == 2) ST16_FORM(ST16_NOP(1));
    else if constexpr (NOPS == 3) ST16_FORM(ST16_NOP(2));
    else ST16_FORM(ST16_NOP(4));
#undef ST16_FORM
#undef ST16_BODY
#undef ST16_NOP
  }
}
__global__ __launch_bounds__(256) void diag_store16_check_kernel(const unsigned* __restrict__ buf, unsigned slots, unsigned long long* bad) {
  unsigned n = 0;
  for (unsigned s = blockIdx.x * 256 + threadIdx.x; s < slots; s += gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(buf)[s];
    if (v.x != store16_pattern(s, 0) || v.y != store16_pattern(s, 1) || v.z != store16_pattern(s, 2) || v.w != store16_pattern(s, 3)) ++n;
  }
  if (n) atomicAdd(bad, (unsigned long long)n);
}
}  // namespace

// buf: blocks * 256 * iters * 16 bytes (<= 2 GB); bad: device counter, zeroed by the caller.  Returns 0 or a HIP error.
extern "C" __attribute__((visibility("default"))) int hypad_diag_store16(int form, int nops, int blocks, int iters, unsigned* buf, unsigned long long* bad, hypad_stream_t s) {
  if (form < 0 || form > 2 || nops < 0 || nops > 4 || blocks < 1 || iters < 1 || (long long)blocks * 256 * iters * 16 > 0x7fffffffLL) return HYPAD_EINVAL;
#define ST16_LAUNCH(F, N) hipLaunchKernelGGL((diag_store16_kernel<F, N>), dim3(blocks), dim3(256), 0, (hipStream_t)s, buf, iters)
#define ST16_N(F) do { switch (nops) { case 0: ST16_LAUNCH(F, 0); break; case 1: ST16_LAUNCH(F, 1); break; case 2: ST16_LAUNCH(F, 2); break; \
                                       case 3: ST16_LAUNCH(F, 3); break; default: ST16_LAUNCH(F, 4); } } while (0)
  if (form == 0) ST16_N(0); else if (form == 1) ST16_N(1); else ST16_N(2);
#undef ST16_N
#undef ST16_LAUNCH
  hipLaunchKernelGGL(diag_store16_check_kernel, dim3(1024), dim3(256), 0, (hipStream_t)s, buf, (unsigned)(blocks * 256 * iters), bad);
  return (int)hipGetLastError();
}

// ---- two concurrent kernels on one XCD handing data to each other (round 5: could the dW + Adam launch run BESIDE the generator
// launch, fed through flags?).  Kernel `role` 0 writes `words` dwords of payload with PLAIN stores, drains, raises a flag word; role 1
// polls the flag with sc1 (L1-bypassing) loads, reads the payload with sc1 loads, checks it, and answers with its own payload + flag --
// `rounds` times.  Only the workgroup that finds itself on XCD `xcd` takes part (grid = 8 workgroups, dealt round-robin).
// out[0..1]: shader-clock cycles of the whole exchange per role; out[2..3]: payload words that arrived wrong; out[4..5]: 1 = a bounded
// wait gave up (the partner never showed: the two launches did not run beside each other).
namespace {
__global__ __launch_bounds__(256) void diag_pair_kernel(int role, int xcd, int rounds, int words, unsigned* flags, unsigned* payload, long long* out) {
  const unsigned my_xcc = __builtin_amdgcn_s_getreg(6164) & 0xfu;
  if ((int)my_xcc != xcd) return;
  __shared__ int give_up;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  unsigned* mine = payload + role * words;
  unsigned* theirs = payload + (1 - role) * words;
  long long bad = 0;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int r = 1; r <= rounds; ++r) {
    if (role == 0 || r > 0) {
      if (role == 1) {                                                   // wait for the partner's round r
        if (threadIdx.x == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(flags + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
            if (++spins > (1u << 22)) { give_up = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        __syncthreads();
        if (give_up) break;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(theirs, 0, 0x7fffffff, 0x00020000);
        for (int i = threadIdx.x; i < words; i += 256) {
          const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, i * 4, 0, 16);      // sc1
          if (v != (unsigned)(r * 131071 + i)) ++bad;
        }
      }
      for (int i = threadIdx.x; i < words; i += 256) mine[i] = (unsigned)(r * 131071 + i) + (role ? 7u : 0u);      // plain stores
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flags + role, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (role == 0) {                                                   // wait for the answer of round r
        if (threadIdx.x == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
            if (++spins > (1u << 22)) { give_up = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        __syncthreads();
        if (give_up) break;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(theirs, 0, 0x7fffffff, 0x00020000);
        for (int i = threadIdx.x; i < words; i += 256) {
          const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, i * 4, 0, 16);
          if (v != (unsigned)(r * 131071 + i) + 7u) ++bad;
        }
      }
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (bad) atomicAdd((unsigned long long*)(out + 2 + role), (unsigned long long)bad);
  if (threadIdx.x == 0) { out[role] = t1 - t0; out[4 + role] = give_up; }
}
}  // namespace
// flags: 2 words, payload: 2 * words dwords, out: 6 int64 -- all zeroed by the caller.  The two launches go to two streams.
extern "C" __attribute__((visibility("default"))) int hypad_diag_pair(int xcd, int rounds, int words, unsigned* flags, unsigned* payload, long long* out,
                                                                      hypad_stream_t s0, hypad_stream_t s1) {
  if (xcd < 0 || xcd > 7 || rounds < 1 || words < 1) return HYPAD_EINVAL;
  hipLaunchKernelGGL(diag_pair_kernel, dim3(8), dim3(256), 0, (hipStream_t)s0, 0, xcd, rounds, words, flags, payload, out);
  hipLaunchKernelGGL(diag_pair_kernel, dim3(8), dim3(256), 0, (hipStream_t)s1, 1, xcd, rounds, words, flags, payload, out);
  return (int)hipGetLastError();
}
