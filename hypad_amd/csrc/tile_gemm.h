// Row-tile GEMMs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// A workgroup owns MT*16 rows of activations in LDS and streams weights from global memory (Infinity-Cache / L2
// resident, ~1 MB per model).  Two shapes cover every layer of the TadGAN networks:
//   gemm_nt :  Y[r][n] = sum_k X[r][k] * W[wrow(n)][k]  (+ bias)      forward of Linear / LSTM gates
//   gemm_nn :  Y[r][c] = sum_n D[r][n] * W[wrow(n)][c]                 backward-data
// wrow(n) = n + (n >= split ? gap : 0) lets the LSTM layers skip the f-gate block of W_ih (never used at
// seq_len 1 with c0 = 0: SURVEY.md A.2) while keeping PyTorch's [i,f,g,o] weight layout.
//
// What bounds these layers (measured on MI355X, scripts/diag_*.py, profiles/README.md):
//   * a 16..48-row tile is ~1 CU-us of MFMA work, so everything is latency / issue bound;
//   * the MFMA B operand of gemm_nt is "16 weight rows x 4 consecutive k": loading it straight from the row-major
//     (out, in) weight is a 16-row gather of 64-byte half lines -- the CU's texture path sustains only ~19 B/clk
//     on that pattern (3 470 cycles per 8 KiB tile with 8 waves) against ~76 B/clk on full 128-byte lines.
// Hence:
//   gemm_nt stages every 16-row x 32-k weight slab through a small wave-private LDS tile: global loads are
//   8 rows x 128 B (full lines) per instruction, fragments are re-read from LDS with ds_read_b128;
//   gemm_nn reads W rows directly (its B operand is contiguous along the output index) as float4 = four 16-column
//   tiles per load, and splits the reduction range over the waves with a fixed-order (deterministic) combine;
//   all inner loops are branch-free (clamped addresses + zero-selects) so that the loads of a tile are issued
//   together and the MFMAs run back to back.
//
// MFMA 16x16x4 f32 lane map (cdna_hip_programming.md §3): lane l, j = l & 15, q = l >> 4
//   A[i = j][k = q],  B[k = q][col = j],  D[row = 4*q + reg][col = j].
// The reduction index may be permuted freely as long as A and B agree: a float4 per lane feeds four consecutive
// MFMAs (lane (.,q) carries k = k0 + 4q + i at step i).
#pragma once
#include <type_traits>
#include "device_utils.h"

// Wave priority around the fp32 MFMA streams of the latency-chain kernels.  Measured (scripts/diag_coissue.py): while one wave of a
// SIMD streams v_mfma_f32_16x16x4_f32 back to back, the other wave of that SIMD issues ONE VALU / LDS instruction per MFMA
// (38 cycles instead of 6): the fp32 MFMA occupies the SIMD's FMA lanes, so MFMA time and VALU time of a SIMD add up, and an
// epilogue or an address prologue next to a streaming wave is starved.  The latency-chain callers (PRE) run at priority 2 and
// drop to 0 inside the MFMA loops: the wave with scalar / VALU work gets its issue slots first.
namespace hypad {
template <bool ON> __device__ __forceinline__ void mfma_prio_begin() { if constexpr (ON) __builtin_amdgcn_s_setprio(0); }
template <bool ON> __device__ __forceinline__ void mfma_prio_end() { if constexpr (ON) __builtin_amdgcn_s_setprio(2); }


using f32x4 = __attribute__((ext_vector_type(4))) float;

struct RowMap {
  int split, gap;
  __device__ __forceinline__ int operator()(int n) const { return n + (n >= split ? gap : 0); }
};
__device__ __forceinline__ RowMap identity_map() { return RowMap{0x7fffffff, 0}; }
// compact (i,g,o) gate column c in [0,3H) -> row of the (4H, in) PyTorch weight
__device__ __forceinline__ RowMap lstm_gate_map(int H) { return RowMap{H, H}; }

constexpr int WSTAGE_LD = 36;                        // floats per row of the wave-private weight slab (32 + pad)
constexpr int WSTAGE_FLOATS = 16 * WSTAGE_LD;        // per wave
__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// ---------------------------------------------------------------------------------------------------- gemm_nt
// One output tile (16 weight rows n0..n0+15), K % V == 0, W rows V*4-byte aligned.  `stage`: this wave's LDS slab.
// Super-chunks of up to 128 k: all global loads first, then per 32-k slab: LDS write -> fragment reads -> MFMAs.
template <int MT, int V>
__device__ __forceinline__ void nt_tile_staged(const float* __restrict__ Xs, int ldx, const float* __restrict__ W, int ldw, int K,
                                               int N, int n0, RowMap map, float* __restrict__ stage, int lane, f32x4 (&acc)[MT]) {
  constexpr int RPI = 64 * V / 32;                   // weight rows covered by one load instruction (V=4: 8, V=2: 4)
  constexpr int NI = 16 / RPI;                       // load instructions per 32-k slab
  constexpr int SC = 4;                              // slabs per super-chunk (128 k)
  const int j = lane & 15, q = lane >> 4;
  const int lrow = lane / (32 / V), lcol = V * (lane % (32 / V));   // this lane's (row within instruction, k offset)
  const int klast = K - V;
  f32x4 acc2[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* rowp[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int n = n0 + lrow + RPI * i;
    n = n < N ? n : N - 1;                           // rows past N: valid duplicate, result dropped by the caller
    rowp[i] = W + (size_t)map(n) * ldw;
  }
  for (int k0 = 0; k0 < K; k0 += 32 * SC) {
    float wreg[SC][NI][V];
#pragma unroll
    for (int s = 0; s < SC; ++s) {
      int kk = k0 + 32 * s + lcol;
      kk = kk < klast ? kk : klast;                  // clamp: always a valid address (A is zeroed there)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (V == 4) {
          const float4 t = *reinterpret_cast<const float4*>(rowp[i] + kk);
          wreg[s][i][0] = t.x; wreg[s][i][1] = t.y; wreg[s][i][2] = t.z; wreg[s][i][3] = t.w;
        } else {
          const float2 t = *reinterpret_cast<const float2*>(rowp[i] + kk);
          wreg[s][i][0] = t.x; wreg[s][i][1] = t.y;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < SC; ++s) {
      const int ks = k0 + 32 * s;
      if (ks < K) {                                  // wave-uniform
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          float* d = stage + (lrow + RPI * i) * WSTAGE_LD + lcol;
          if (V == 4) *reinterpret_cast<float4*>(d) = make_float4(wreg[s][i][0], wreg[s][i][1], wreg[s][i][2], wreg[s][i][3]);
          else *reinterpret_cast<float2*>(d) = make_float2(wreg[s][i][0], wreg[s][i][1]);
        }
        float4 b[2], a[MT][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          b[u] = *reinterpret_cast<const float4*>(stage + j * WSTAGE_LD + 16 * u + 4 * q);
          int ka = ks + 16 * u + 4 * q;
          ka = ka < ldx - 4 ? ka : ldx - 4;          // stay inside the activation row (values beyond K are masked)
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m][u] = *reinterpret_cast<const float4*>(Xs + (m * 16 + j) * ldx + ka);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool live = ks + 16 * u + 4 * q + i < K;
            const float bv = f4get(b[u], i);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float av = live ? f4get(a[m][u], i) : 0.f;
              if (i & 1) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc2[m], 0, 0, 0);
              else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m], 0, 0, 0);
            }
          }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] += acc2[m];
}

// direct (un-staged) tile for odd K / unaligned weights: scalar loads, branch-free
template <int MT>
__device__ __forceinline__ void nt_tile_scalar(const float* __restrict__ Xs, int ldx, const float* __restrict__ wp, int K, int j, int q,
                                               f32x4 (&acc)[MT]) {
  constexpr int CH = MT == 1 ? 16 : 8;
  const int G = (K + 3) >> 2;
  for (int g0 = 0; g0 < G; g0 += CH) {
    float b[CH], a[MT][CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      int k = (g0 + u) * 4 + q;
      k = k < K ? k : K - 1;
      b[u] = wp[k];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m][u] = Xs[(m * 16 + j) * ldx + k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const bool live = (g0 + u) * 4 + q < K;
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? a[m][u] : 0.f, b[u], acc[m], 0, 0, 0);
    }
  }
}

// Y (LDS) [MT*16][ldy], columns ycol0 .. ycol0+N-1.  X (LDS) [MT*16][ldx], K columns, ldx % 4 == 0, X 16-byte aligned.
// W rows of length ldw (global, or LDS through a flat pointer); bias pointers (indexed like W rows) may be null.
// wstage: LDS, WSTAGE_FLOATS per wave of the workgroup (wave-private weight slabs).
template <int MT>
__device__ __forceinline__ void gemm_nt(const float* __restrict__ Xs, int ldx, const float* __restrict__ W, int ldw, int K, int N,
                        RowMap map, const float* __restrict__ bias0, const float* __restrict__ bias1,
                        float* __restrict__ Ys, int ldy, int ycol0, float* __restrict__ wstage, int wave_rot = 0) {
  // wave_rot: tile t goes to wave (t + wave_rot) mod nwaves -- back-to-back calls without a barrier between them (the two
  // directions of an LSTM layer) rotate the deal so the same waves do not get the odd tile every time
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6, wave = (wave_id() + nwaves - wave_rot % nwaves) % nwaves;
  const int j = lane & 15, q = lane >> 4;
  const int ntiles = (N + 15) >> 4;
  const uintptr_t wa = reinterpret_cast<uintptr_t>(W);
  const int vw = (((K | ldw) & 3) == 0 && (wa & 15) == 0) ? 4 : (((K | ldw) & 1) == 0 && (wa & 7) == 0) ? 2 : 1;
  float* stage = wstage + (threadIdx.x >> 6) * WSTAGE_FLOATS;
  for (int t = wave; t < ntiles; t += nwaves) {
    const int n = t * 16 + j;
    const bool nv = n < N;
    const int wr = map(nv ? n : N - 1);                // clamp: lanes past N read a valid row and drop the result
    float bsum = 0.f;
    if (bias0) bsum += bias0[wr];
    if (bias1) bsum += bias1[wr];
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (vw == 4) nt_tile_staged<MT, 4>(Xs, ldx, W, ldw, K, N, t * 16, map, stage, lane, acc);
    else if (vw == 2) nt_tile_staged<MT, 2>(Xs, ldx, W, ldw, K, N, t * 16, map, stage, lane, acc);
    else nt_tile_scalar<MT>(Xs, ldx, W + (size_t)wr * ldw, K, j, q, acc);
    if (nv) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ys[(m * 16 + 4 * q + r) * ldy + ycol0 + n] = acc[m][r] + bsum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- gemm_nt_packed
// Forward product against a weight stored in MFMA-native blocks: block (tn, g) = 64 lanes x float4, lane (j, q) holding
// W[16 tn + j][16 g + 4 q .. + 3] (rows past N / columns past K are zeros; LSTM gate rows already compacted to [i|g|o]).
// A tile's weights are then `kg` fully coalesced 1 KB loads straight into the B operand registers: no LDS re-shape.
// Wp: blocks of tile tn at Wp + tn * kg * 256 floats; bsum[n]: summed biases (may be null).
// The first batch of a wave's first tile can be requested ahead of time (weights do not depend on activations): issue
// gemm_nt_prefetch() before the previous stage's work and pass the result with PRE = true; the product then starts with its
// B operands in registers instead of an L2 round trip (~2 k cycles on a chain that runs every product once).
#ifndef HYPAD_R6_ABATCH
#define HYPAD_R6_ABATCH 1
#endif
struct PackedPre { float4 w[8]; };
// Packed weights through one of two doors.  SC1 = false: plain global loads (throughput callers).  SC1 = true: `sc1` buffer loads
// on a descriptor built from the (wave-uniform) matrix pointer -- 1 KB blocks at scalar offsets, the lane's 16 bytes at a
// constant vector offset: no vector address arithmetic at all, which is what the latency-chain kernels pay for (measured on the
// generator step: -1.7 us of 40.6; the sc1 bit itself costs nothing there -- every weight line is read once per workgroup).
// It is also the form a consumer must use for EVERY byte another workgroup of the same launch has just written with `sc1`
// stores (cdna_hip_programming.md Guideline 16 R1).
typedef unsigned int wl_u32x4 __attribute__((vector_size(16)));
typedef unsigned int wl_u32x2 __attribute__((vector_size(8)));
template <bool SC1>
struct WeightBlocks {
  const float4* wp; __amdgpu_buffer_rsrc_t rs; int voff;
  __device__ __forceinline__ WeightBlocks(const float* W, int lane) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0x7fffffff, 0x00020000); voff = lane * 16; wp = nullptr;
  }
  __device__ __forceinline__ float4 operator()(int blk) const {             // block index: wave-uniform
    if constexpr (SC1) return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, blk * 1024, 16));
    else return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, blk * 1024, 0));
  }
};
// Rows of a global fp32 array behind a buffer descriptor: a per-lane byte offset (computed once) + a scalar / constant offset per
// access -- the epilogues of the latency-chain kernels store and fetch ~20 values per lane and stage, and as flat accesses each
// one cost a 64-bit vector add on a SIMD whose fp32 MFMAs overlap with nothing.  (base: wave-uniform, may be null if unused)
struct GBuf {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ explicit GBuf(const float* base) : rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000)) {}
  __device__ __forceinline__ float ld(int vo, int so) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0)); }
  __device__ __forceinline__ void st(float v, int vo, int so) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vo, so, 0); }
  // (16 bytes: the whole vector is converted at once -- hipcc narrows an element-wise use of a b128 result to one dword)
  __device__ __forceinline__ float4 ld4(int vo, int so) const { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0)); }
  // 16-byte stores take NO scalar offset.  Hardware hazard (gfx950, isolated in round 3: csrc/diag.hip hypad_diag_store16,
  // scripts/diag_store16.py, profiles/r03_store16_hazard.txt): a VALU write of the data registers of a `buffer_store_dwordx4`
  // needs wait states behind the store -- ONE with the row offset in a scalar register (`s_off offen`: 1.1 % of the stores
  // carried the overwriting values with none, 0 of 1.3e8 with one), TWO with an immediate offset (`0 offen`: 24 % / 1.1 % / 0).
  // hipcc (ROCm 7.2) pads the immediate form (`s_nop 1`) but treats the scalar-offset form as hazard-free and pads nothing:
  // whenever its scheduler put the next writer of those registers right behind such a store, values were lost -- the
  // intermittent gradients of round 2 (2-5 of 6 test runs).  With the offset folded into the vector offset the compiler's own
  // padding covers it.  tests/test_cabi_and_host.py keeps every 16-byte buffer store of the sources on the immediate form.
  __device__ __forceinline__ void st4(const float4& v, int vo) const { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_u32x4, v), rs, vo, 0, 0); }
};
template <bool SC1>
__device__ __forceinline__ float weight_scalar(const float* p) {            // one float of handed-off data (bias sums, ...)
  if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool SC1 = false>
__device__ __forceinline__ PackedPre gemm_nt_prefetch(const float* __restrict__ Wp, int K, int N, int wave_rot = 0) {
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6, wave = (wave_id() + nwaves - wave_rot % nwaves) % nwaves;
  const int ntiles = (N + 15) >> 4, kg = (K + 15) >> 4;
  const WeightBlocks<SC1> wb(Wp, lane);
  const int b0 = (wave < ntiles ? wave : 0) * kg;
  PackedPre p;
#pragma unroll
  for (int u = 0; u < 8; ++u) p.w[u] = wb(b0 + (u < kg ? u : kg - 1));
  return p;
}
struct ActIdentity { __device__ __forceinline__ float operator()(float v) const { return v; } };
// The product with a caller-supplied epilogue object (MT * 16 rows):
//   epi.prefetch(n, q, ok)          before a tile's reduction: request whatever the epilogue needs for output column n and
//                                   rows 16 m + 4 q + r (ok: n < N) -- it arrives under the MFMAs;
//   epi.emit(m, r, row, n, value)   for every element of the tile (value = product + summed bias), n < N.
// An elementwise stage that follows a product costs a pass over LDS and a workgroup barrier on its own; in the epilogue it costs
// its arithmetic.
template <int MT, bool PRE, class Epi, bool SC1 = false>
__device__ __forceinline__ void gemm_nt_packed_epi(const float* __restrict__ Xs, int ldx, int K, int N, const float* __restrict__ Wp,
                                                   const float* __restrict__ bsum, int wave_rot, const PackedPre& pre, Epi& epi) {
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6, wave = (wave_id() + nwaves - wave_rot % nwaves) % nwaves;
  const int j = lane & 15, q = lane >> 4;
  const int ntiles = (N + 15) >> 4, kg = (K + 15) >> 4;
  auto run = [&](int t, auto first_from_pre) __attribute__((always_inline)) {
    const WeightBlocks<SC1> wb(Wp, lane);
    const int b0 = t * kg;
    const int n = t * 16 + j;
    const float bs = (bsum && n < N) ? weight_scalar<SC1>(bsum + n) : 0.f;
    // (what the epilogue needs is requested behind the LAST batch of weights: loads return in order, and only the epilogue waits for it)
    bool epi_asked = false;
    f32x4 acc[MT], acc2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto consume = [&](const float4 (&w)[8], int g0) __attribute__((always_inline)) {
#if HYPAD_R6_ABATCH
      if constexpr (SC1 && MT <= HYPAD_R6_ABATCH) {    // (the latency-chain callers: all A fragments of the batch requested before its first product)
        float4 av[8][MT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k0 = 16 * (g0 + u) + 4 * q;
          const int ka = k0 < ldx - 4 ? k0 : ldx - 4;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            av[u][m] = (g0 + u < kg) ? *reinterpret_cast<const float4*>(Xs + (m * 16 + j) * ldx + ka) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (g0 + u < kg) {
            const int k0 = 16 * (g0 + u) + 4 * q;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float4 a = av[u][m];
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 < K ? a.x : 0.f, w[u].x, acc[m], 0, 0, 0);
              acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 + 1 < K ? a.y : 0.f, w[u].y, acc2[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 + 2 < K ? a.z : 0.f, w[u].z, acc[m], 0, 0, 0);
              acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 + 3 < K ? a.w : 0.f, w[u].w, acc2[m], 0, 0, 0);
            }
          }
        }
        return;
      }
#endif
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (g0 + u < kg) {                             // wave-uniform
          const int k0 = 16 * (g0 + u) + 4 * q;
          const int ka = k0 < ldx - 4 ? k0 : ldx - 4;  // stay inside the activation row (the weights are zero past K)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float4 a = *reinterpret_cast<const float4*>(Xs + (m * 16 + j) * ldx + ka);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 < K ? a.x : 0.f, w[u].x, acc[m], 0, 0, 0);
            acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 + 1 < K ? a.y : 0.f, w[u].y, acc2[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 + 2 < K ? a.z : 0.f, w[u].z, acc[m], 0, 0, 0);
            acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0 + 3 < K ? a.w : 0.f, w[u].w, acc2[m], 0, 0, 0);
          }
        }
      }
    };
    int gbeg = 0;
    mfma_prio_begin<PRE>();
    if constexpr (decltype(first_from_pre)::value) {
      if (kg <= 8) { epi.prefetch(n, q, n < N); epi_asked = true; }
      consume(pre.w, 0);
      gbeg = 8;
    }
    for (int g0 = gbeg; g0 < kg; g0 += 8) {            // 8 k-groups (128 k) of weights in flight per lane
      float4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = wb(b0 + (g0 + u < kg ? g0 + u : kg - 1));
      if (g0 + 8 >= kg) { epi.prefetch(n, q, n < N); epi_asked = true; }
      __builtin_amdgcn_sched_barrier(0);
      consume(w, g0);
    }
    if (!epi_asked) epi.prefetch(n, q, n < N);
    mfma_prio_end<PRE>();
    if (n < N) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) epi.emit(m, r, m * 16 + 4 * q + r, n, acc[m][r] + acc2[m][r] + bs);
    }
  };
  if (wave < ntiles) run(wave, std::integral_constant<bool, PRE>{});
  for (int t = wave + nwaves; t < ntiles; t += nwaves) run(t, std::false_type{});
}
// The plain epilogue: Ys[row][ycol0 + n] = act(value) * escale[row][n].  escale (may be null): global [rows][es_ld] factors (a
// dropout mask in the backward pass), fetched before the reduction so that the epilogue does not wait for them.
template <int MT, class Act>
struct PlainEpi {
  float* Ys; int ldy, ycol0; Act act; const float* escale; int es_ld;
  float* gout; int gld, gps; // global mirror of the result rows (row stride gld; gps != 0: row r lives at (r >> 4) * gps + (r & 15)), or null
  float es[MT][4];
  int vo_g;                   // this lane's byte offset into gout: row 4 q, column n
  __device__ __forceinline__ void prefetch(int n, int q, bool ok) {
    vo_g = (4 * q * gld + n) * 4;
    const GBuf eb(escale);
    const int vo_e = (4 * q * es_ld + (ok ? n : 0)) * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) es[m][r] = (escale && ok) ? eb.ld(vo_e, ((m * 16 + r) * es_ld) * 4) : 1.f;
  }
  __device__ __forceinline__ void emit(int m, int r, int row, int n, float v) {
    const float o = act(v) * es[m][r];
    Ys[row * ldy + ycol0 + n] = o;
    // (row = 16 m + 4 q + r: the lane part is vo_g, the rest a scalar)
    if (gout) GBuf(gout).st(o, vo_g, ((gps ? m * gps : m * 16) + r) * gld * 4);
  }
};
template <int MT, bool PRE = false, class Act = ActIdentity, bool SC1 = false>
__device__ __forceinline__ void gemm_nt_packed(const float* __restrict__ Xs, int ldx, int K, int N, const float* __restrict__ Wp,
                                               const float* __restrict__ bsum, float* __restrict__ Ys, int ldy, int ycol0, int wave_rot = 0,
                                               const PackedPre& pre = PackedPre{}, Act act = Act{},
                                               const float* __restrict__ escale = nullptr, int es_ld = 0,
                                               float* __restrict__ gout = nullptr, int gld = 0, int gps = 0) {
  PlainEpi<MT, Act> epi{Ys, ldy, ycol0, act, escale, es_ld, gout, gld, gps, {}};
  gemm_nt_packed_epi<MT, PRE, PlainEpi<MT, Act>, SC1>(Xs, ldx, K, N, Wp, bsum, wave_rot, pre, epi);
}

// ---------------------------------------------------------------------------------------------------- gemm_nn
// Y (LDS) [MT*16][ldy] columns 0..C-1  (+)= D (LDS) [MT*16][ldd] columns dcol0..dcol0+Nred-1  times  W rows wrow(n).
// Contains __syncthreads(): every wave of the workgroup must call it.  The caller needs no barrier between two
// accumulating calls on the same Y.
template <int MT>
__device__ __forceinline__ void gemm_nn(const float* __restrict__ Ds, int ldd, int dcol0, const float* __restrict__ W, int ldw,
                        int Nred, RowMap map, int C, float* __restrict__ Ys, int ldy, bool accumulate) {
  const int lane = threadIdx.x & 63, wave = wave_id(), nwaves = blockDim.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const bool vec = ((C | ldw) & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 && Nred >= 64;
  if (vec) {
    // quad path: one float4 of W = columns c0+4j..c0+4j+3 of row n -> four 16-column tiles (tile i owns columns
    // c0 + 4*jj + i).  Waves = (column quad) x (slice of the reduction range); partial tiles are combined in a fixed order.
    const int nquads = (C + 63) >> 6;
    int nsplit = nwaves / nquads;
    if (nsplit < 1) nsplit = 1;
    const int G = (Nred + 3) >> 2;
    const int gper = (G + nsplit - 1) / nsplit;
    const int rounds = (nquads + (nwaves / nsplit) - 1) / (nwaves / nsplit);     // quads per wave slot (normally 1)
    for (int rd = 0; rd < rounds; ++rd) {
      const int quad = rd * (nwaves / nsplit) + wave / nsplit;
      const int sp = wave % nsplit;
      const bool work = quad < nquads && wave < (nwaves / nsplit) * nsplit;
      const int c0 = quad * 64;
      int cb = c0 + 4 * j;
      const bool cvalid = work && cb < C;
      cb = cvalid ? cb : 0;
      f32x4 acc[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (work) {
        const int gbeg = sp * gper, gend = (gbeg + gper < G) ? gbeg + gper : G;
        constexpr int CH = MT == 1 ? 8 : 4;
        for (int g0 = gbeg; g0 < gend; g0 += CH) {
          float4 b[CH];
          float a[MT][CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            int n = (g0 + u) * 4 + q;
            n = n < Nred ? n : Nred - 1;
            b[u] = *reinterpret_cast<const float4*>(W + (size_t)map(n) * ldw + cb);
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m][u] = Ds[(m * 16 + j) * ldd + dcol0 + n];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const bool live = (g0 + u) < gend && (g0 + u) * 4 + q < Nred;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float av = live ? a[m][u] : 0.f;
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f4get(b[u], i), acc[m][i], 0, 0, 0);
            }
          }
        }
      }
      // fixed-order combine: slice 0 stores (or adds to the existing Y), slices 1.. add in turn
      for (int s = 0; s < nsplit; ++s) {
        if (cvalid && sp == s) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float4* y = reinterpret_cast<float4*>(Ys + (m * 16 + 4 * q + r) * ldy + cb);
              float4 v = make_float4(acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]);
              if (s > 0 || accumulate) { const float4 o = *y; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
              *y = v;
            }
        }
        __syncthreads();
      }
    }
    return;
  }
  // scalar path (narrow / odd outputs): one 16-column tile per wave, branch-free
  const int ntiles = (C + 15) >> 4;
  constexpr int CH = MT == 1 ? 16 : 8;
  const int G = (Nred + 3) >> 2;
  for (int t = wave; t < ntiles; t += nwaves) {
    const int c = t * 16 + j;
    const bool cv = c < C;
    const int cc = cv ? c : C - 1;
    f32x4 acc[MT], acc2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int g0 = 0; g0 < G; g0 += CH) {
      float b[CH];
      float a[MT][CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        int n = (g0 + u) * 4 + q;
        n = n < Nred ? n : Nred - 1;
        b[u] = W[(size_t)map(n) * ldw + cc];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m][u] = Ds[(m * 16 + j) * ldd + dcol0 + n];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const bool live = (g0 + u) * 4 + q < Nred;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (u & 1) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? a[m][u] : 0.f, b[u], acc2[m], 0, 0, 0);
          else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? a[m][u] : 0.f, b[u], acc[m], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] += acc2[m];
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
  __syncthreads();
}

}  // namespace hypad
