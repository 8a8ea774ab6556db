// Tile-level building blocks of the TadGAN networks: a workgroup carries MT*16 rows through whole layers,
// activations in LDS, weights streamed from the parameter arena.  Closed forms follow SURVEY.md A.2 and
// oracle/manual.py (the CPU derivation sheet these functions transcribe).  All loops are written for any
// workgroup size that is a multiple of 64 (waves over rows, lanes over columns).
#pragma once
#include "layout.h"
#include "rowops.h"
#include "tile_gemm.h"

namespace hypad {

constexpr int LP = 32;  // padded row stride for latent / critic-width tiles (L <= MAX_L = 32)

// ---------------------------------------------------------------------------------------- dropout source
struct DropSrc {
  int mode;               // 0 none (eval), 1 injected masks, 2 device Philox
  const float* ptr;       // injected: [(layer)][batch][ncols]
  int batch;              // rows per layer block of the injected array
  uint64_t seed;
  uint32_t tick, stream, sig;
  float p;
  __device__ __forceinline__ float get(int layer, int grow, int c, int ncols) const {
    if (mode == 0) return 1.f;
    if (mode == 1) return ptr[((size_t)layer * batch + grow) * ncols + c];
    return rng_dropout(seed, tick, stream + layer, sig, (uint32_t)(grow * ncols + c), p);
  }
  // keep-scales of columns c4 .. c4+3 (c4 % 4 == 0, ncols % 4 == 0): the same numbers as four get() calls, from one
  // Philox evaluation in the device-RNG mode
  __device__ __forceinline__ float4 get4(int layer, int grow, int c4, int ncols) const {
    if (mode == 0) return make_float4(1.f, 1.f, 1.f, 1.f);
    if (mode == 1) return *reinterpret_cast<const float4*>(ptr + ((size_t)layer * batch + grow) * ncols + c4);
    Philox ph(seed);
    const uint4 r = ph((uint32_t)(grow * ncols + c4) >> 2, stream + layer, tick, sig);
    const float keep = 1.0f / (1.0f - p);
    return make_float4(u32_to_unit(r.x) >= p ? keep : 0.f, u32_to_unit(r.y) >= p ? keep : 0.f,
                       u32_to_unit(r.z) >= p ? keep : 0.f, u32_to_unit(r.w) >= p ? keep : 0.f);
  }
};
__device__ __forceinline__ DropSrc no_drop() { return DropSrc{0, nullptr, 0, 0, 0, 0, 0, 0.f}; }

// ---------------------------------------------------------------------------------------- tile movement
// Multi-pass tiles: LDS row r belongs to pass r>>4; its global row is (r>>4)*ps + (r&15) (ps = rows between passes;
// ps = 16 means plain contiguous rows).
__device__ __forceinline__ int64_t prow(int r, int ps) { return (int64_t)(r >> 4) * ps + (r & 15); }

// global (row stride gld) -> LDS [rows][ld]; rows >= valid are zero-filled
__device__ __forceinline__ void tile_load(float* __restrict__ dst, int ld, const float* __restrict__ src, int64_t gld,
                                          int rows, int cols, int valid) {
  tile_for(rows, cols, [&](int r, int c) { dst[r * ld + c] = r < valid ? src[(int64_t)r * gld + c] : 0.f; });
}
__device__ __forceinline__ void tile_load_rows(float* __restrict__ dst, int ld, const float* __restrict__ base, int64_t gld,
                                               const int32_t* __restrict__ row_index, int row0, int rows, int cols, int valid) {
  const int lane = threadIdx.x & 63, wave = wave_id(), nw = blockDim.x >> 6;
  // a gathered row is two dependent memory round trips (index, then row): a wave takes its rows two at a time so that
  // the round trips of the pair overlap
  for (int r = wave; r < rows; r += 2 * nw) {
    const int r2 = r + nw;
    const bool ok = r < valid, ok2 = r2 < rows && r2 < valid;
    const int64_t gr = ok ? (row_index ? (int64_t)row_index[row0 + r] : (int64_t)(row0 + r)) : 0;
    const int64_t gr2 = ok2 ? (row_index ? (int64_t)row_index[row0 + r2] : (int64_t)(row0 + r2)) : 0;
    for (int c = lane; c < cols; c += 64) {
      const float a = ok ? base[gr * gld + c] : 0.f;
      const float b = ok2 ? base[gr2 * gld + c] : 0.f;
      dst[r * ld + c] = a;
      if (r2 < rows) dst[r2 * ld + c] = b;
    }
  }
}
// LDS [rows][ld] -> global (row stride gld): 16 bytes per lane when the shapes allow it (a scalar sweep is one dependent
// LDS read -> store per element and thread, 12 in a row for a 16 x 384 tile)
__device__ __forceinline__ bool tile_vec4_ok(const float* g, int64_t gld, int ld, int cols) {
  return ((cols | ld | (int)gld) & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0;
}
__device__ __forceinline__ void tile_store(float* __restrict__ dst, int64_t gld, const float* __restrict__ src, int ld,
                                           int rows, int cols, int valid) {
  if (tile_vec4_ok(dst, gld, ld, cols)) {
    tile_for(rows, cols >> 2, [&](int r, int c4) {
      if (r < valid) *reinterpret_cast<float4*>(dst + (int64_t)r * gld + 4 * c4) = *reinterpret_cast<const float4*>(src + r * ld + 4 * c4);
    });
    return;
  }
  tile_for(rows, cols, [&](int r, int c) { if (r < valid) dst[(int64_t)r * gld + c] = src[r * ld + c]; });
}
// The same two through a buffer descriptor on the tile's first row (src / dst: wave-uniform): 32-bit lane offsets instead of 64-bit
// vector address arithmetic (the scoring kernel kept a 64-bit lane offset alive from its first load to its last store -- in scratch).
__device__ __forceinline__ void tile_load_b(float* __restrict__ dst, int ld, const float* __restrict__ src, int gld, int rows, int cols, int valid) {
  const GBuf b(src);
  tile_for(rows, cols, [&](int r, int c) { dst[r * ld + c] = r < valid ? b.ld((r * gld + c) * 4, 0) : 0.f; });
}
// (the lane number comes from mbcnt here, two instructions: derived from threadIdx.x the compiler recognises the kernel's first lane
// offset in it and keeps that one register alive -- spilled -- until the stores at the kernel's end)
__device__ __forceinline__ void tile_store_b(float* __restrict__ dst, int gld, const float* __restrict__ src, int ld, int rows, int cols, int valid) {
  const GBuf b(dst);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), wave = wave_id(), nw = blockDim.x >> 6;
  if (tile_vec4_ok(dst, gld, ld, cols)) {
    for (int r = wave; r < rows && r < valid; r += nw)
      for (int c4 = lane; c4 < (cols >> 2); c4 += 64) b.st4(*reinterpret_cast<const float4*>(src + r * ld + 4 * c4), (r * gld + 4 * c4) * 4);
    return;
  }
  for (int r = wave; r < rows && r < valid; r += nw)
    for (int c = lane; c < cols; c += 64) b.st(src[r * ld + c], (r * gld + c) * 4, 0);
}
__device__ __forceinline__ void tile_store_p(float* __restrict__ dst, int64_t gld, int ps, const float* __restrict__ src, int ld,
                                             int rows, int cols, int valid) {
  if (tile_vec4_ok(dst, gld, ld, cols)) {
    tile_for(rows, cols >> 2, [&](int r, int c4) {
      if (r < valid) *reinterpret_cast<float4*>(dst + prow(r, ps) * gld + 4 * c4) = *reinterpret_cast<const float4*>(src + r * ld + 4 * c4);
    });
    return;
  }
  tile_for(rows, cols, [&](int r, int c) { if (r < valid) dst[prow(r, ps) * gld + c] = src[r * ld + c]; });
}
__device__ __forceinline__ void tile_load_p(float* __restrict__ dst, int ld, const float* __restrict__ src, int64_t gld, int ps,
                                            int rows, int cols, int valid) {
  tile_for(rows, cols, [&](int r, int c) { dst[r * ld + c] = r < valid ? src[prow(r, ps) * gld + c] : 0.f; });
}

// ---------------------------------------------------------------------------------------- LSTM layer, T = 1
// Gate pre-activations for both directions into Gs[rows][ldg] as compact [dir][i|g|o] blocks of H columns.
template <int MT>
__device__ __forceinline__ void lstm_gates_tile(const float* As, int lda, int K, const float* P, const LstmDir& d0,
                                                const LstmDir& d1, int H, float* Gs, int ldg, float* wst) {
  gemm_nt<MT>(As, lda, P + d0.w_ih, K, K, 3 * H, lstm_gate_map(H), P + d0.b_ih, P + d0.b_hh, Gs, ldg, 0, wst);
  gemm_nt<MT>(As, lda, P + d1.w_ih, K, K, 3 * H, lstm_gate_map(H), P + d1.b_ih, P + d1.b_hh, Gs, ldg, 3 * H, wst, (3 * H + 15) >> 4);
}
// cell: c = sig(i)*tanh(g), h = sig(o)*tanh(c).  Hs[rows][ldh] <- [h_fwd | h_rev].
// gates_save (global, may be null): row r at gates_save + prow(r)*8H, layout [dir][i,g,o,tanh c][H].
__device__ __forceinline__ void lstm_cell_tile(const float* Gs, int ldg, int H, int rows, float* Hs, int ldh,
                                               float* gates_save, int valid, int ps = 16) {
  tile_for(rows, 2 * H, [&](int r, int c) {
    const int d = c >= H ? 1 : 0, jj = c - d * H;
    const float* g = Gs + r * ldg + d * 3 * H;
    const float gi = sigmoidf_(g[jj]), gg = tanhf_(g[H + jj]), go = sigmoidf_(g[2 * H + jj]);
    const float tc = tanhf_(gi * gg);
    Hs[r * ldh + c] = go * tc;
    if (gates_save && r < valid) {
      float* s = gates_save + prow(r, ps) * 8 * H + d * 4 * H + jj;
      s[0] = gi; s[H] = gg; s[2 * H] = go; s[3 * H] = tc;
    }
  });
}
// backward of the cell: dHs[rows][ldh] -> dGs[rows][ldg] compact [dir][di|dg|do] (oracle/manual.py lstm_dir_bwd).
// The saved gates come from the global workspace: a thread's (up to) four elements -- rows wave, wave + nw, ...; columns
// lane, lane + 64, ... -- have all their 16 gate values requested before the first is used (one memory round trip; an
// element-at-a-time loop paid one per element on a chain that runs this once per layer).
__device__ __forceinline__ void lstm_cell_bwd_tile(const float* dHs, int ldh, const float* gates_saved, int H, int rows,
                                                   float* dGs, int ldg, int valid, int ps = 16) {
  const int lane = threadIdx.x & 63, wave = wave_id(), nw = blockDim.x >> 6;
  constexpr int RB = 2, CB = 2;                         // rows x column steps per batch (16 rows on 8 waves, 2H <= 128: one batch)
  for (int r0 = wave; r0 < rows; r0 += RB * nw)
    for (int c0 = lane; c0 < 2 * H; c0 += CB * 64) {
      float gi[RB][CB], gg[RB][CB], go[RB][CB], tc[RB][CB], dh[RB][CB];
#pragma unroll
      for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) {
          const int r = r0 + a * nw, c = c0 + b * 64;
          const bool ok = r < rows && r < valid && c < 2 * H;
          const int rr = ok ? r : 0, cc = ok ? c : 0;
          const int d = cc >= H ? 1 : 0, jj = cc - d * H;
          const float* s = gates_saved + prow(rr, ps) * 8 * H + d * 4 * H + jj;
          gi[a][b] = s[0]; gg[a][b] = s[H]; go[a][b] = s[2 * H]; tc[a][b] = s[3 * H];
          dh[a][b] = dHs[(r < rows ? r : 0) * ldh + cc];
        }
#pragma unroll
      for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) {
          const int r = r0 + a * nw, c = c0 + b * 64;
          if (r < rows && c < 2 * H) {
            const int d = c >= H ? 1 : 0, jj = c - d * H;
            float di = 0.f, dg = 0.f, dov = 0.f;
            if (r < valid) {
              dov = dh[a][b] * tc[a][b] * go[a][b] * (1.f - go[a][b]);
              const float dc = dh[a][b] * go[a][b] * (1.f - tc[a][b] * tc[a][b]);
              di = dc * gg[a][b] * gi[a][b] * (1.f - gi[a][b]);
              dg = dc * gi[a][b] * (1.f - gg[a][b] * gg[a][b]);
            }
            float* o = dGs + r * ldg + d * 3 * H;
            o[jj] = di; o[H + jj] = dg; o[2 * H + jj] = dov;
          }
        }
    }
}
// The same cell backward as the epilogue of the product that makes dH (tile_gemm.h gemm_nt_packed_epi, 16 rows): output column
// n = d H + unit of dH turns into the three gate deltas of that unit, written to dGs[row][d 3H + {0, H, 2H} + unit]; the saved
// gates (and an optional [rows][mask_ld] factor on dH: the inter-layer dropout's backward) are requested before the reduction.
struct LstmCellBwdEpi {
  const float* gates_saved; int H; float* dGs; int ldg; int valid; int ps; const float* mask; int mask_ld;
  float* gout;               // global mirror [rows][6H] of the gate deltas (the weight-gradient kernel's operand rows), or null
  float gi[4], gg[4], go[4], tc[4], ms[4];
  int vo_g;                  // this lane's byte offset into gout: row 4 q, column d 3H + unit
  __device__ __forceinline__ void prefetch(int n, int q, bool ok) {
    const int nn = ok ? n : 0, d = nn >= H ? 1 : 0, jj = nn - d * H;
    // (16 rows: prow is the identity; rows past `valid` read row 0 -- their deltas are zeroed below.  Buffer addressing: one
    // lane offset, the row and the gate as scalar / constant offsets)
    const GBuf sb(gates_saved), mb(mask);
    const int vo_s = (4 * q * 8 * H + (d * H + jj) * 4) * 4, vo_m = (4 * q * mask_ld + nn) * 4;      // (lstm_layer_fwd_packed's layout)
    vo_g = (4 * q * 6 * H + d * 3 * H + jj) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool rv = 4 * q + r < valid;
      const int vs = rv ? vo_s : (d * H + jj) * 16, so = rv ? r * 8 * H * 4 : 0;
      const float4 g4 = sb.ld4(vs, so);
      gi[r] = g4.x; gg[r] = g4.y; go[r] = g4.z; tc[r] = g4.w;
      ms[r] = mask ? mb.ld(vo_m, r * mask_ld * 4) : 1.f;
    }
  }
  __device__ __forceinline__ void emit(int, int r, int row, int n, float v) {
    const int d = n >= H ? 1 : 0, jj = n - d * H;
    float di = 0.f, dg = 0.f, dov = 0.f;
    if (row < valid) {
      const float dh = v * ms[r];
      dov = dh * tc[r] * go[r] * (1.f - go[r]);
      const float dc = dh * go[r] * (1.f - tc[r] * tc[r]);
      di = dc * gg[r] * gi[r] * (1.f - gi[r]);
      dg = dc * gi[r] * (1.f - gg[r] * gg[r]);
    }
    float* o = dGs + row * ldg + d * 3 * H;
    o[jj] = di; o[H + jj] = dg; o[2 * H + jj] = dov;
    if (gout && row < valid) {
      const GBuf gb(gout);
      const int so = r * 6 * H * 4;
      gb.st(di, vo_g, so); gb.st(dg, vo_g + H * 4, so); gb.st(dov, vo_g + 2 * H * 4, so);
    }
  }
};
// dA[rows][K] = dG_fwd * W_ih_fwd + dG_rev * W_ih_rev  (compact gate columns -> PyTorch weight rows)
template <int MT>
__device__ __forceinline__ void lstm_bwd_data_tile(const float* dGs, int ldg, const float* P, const LstmDir& d0,
                                                   const LstmDir& d1, int H, int K, float* dAs, int lda) {
  gemm_nn<MT>(dGs, ldg, 0, P + d0.w_ih, K, 3 * H, lstm_gate_map(H), K, dAs, lda, false);
  gemm_nn<MT>(dGs, ldg, 3 * H, P + d1.w_ih, K, 3 * H, lstm_gate_map(H), K, dAs, lda, true);
}

// ---------------------------------------------------------------------------------------- critics (one 16-row tile)
// LDS scratch of a critic pass
struct CriticLds {
  float* act;    // [nh][16][LP]   act[li] = output of hidden layer li (post leaky + dropout)
  float* dm;     // [nh][16][LP]   dm[li]  = leaky'(pre) * dropout scale
  float* dl;     // [2][16][LP]    ping-pong deltas
  float* out;    // [16]
};
constexpr int CRITIC_LDS_FLOATS = (4 + 4 + 2) * 16 * LP + 16;
__device__ __forceinline__ CriticLds critic_lds(float* base) {
  CriticLds c;
  c.act = base; c.dm = base + 4 * 16 * LP; c.dl = c.dm + 4 * 16 * LP; c.out = c.dl + 2 * 16 * LP;
  return c;
}

// forward.  Xs: LDS [16][ldx] input (in_dim columns).  grow0: global batch row of tile row 0 (dropout indexing).
// P may point at a copy of the critic's arena staged in LDS (flat pointer).
__device__ __forceinline__ void critic_fwd_tile(const float* Xs, int ldx, const float* P, const CriticLayout& cl, int L,
                                                const CriticLds& s, const DropSrc& drop, int grow0, float* wst) {
  const float* in = Xs;
  int ldin = ldx, kin = cl.in_dim;
  for (int li = 0; li < cl.nh; ++li) {
    float* a = s.act + li * 16 * LP;
    float* d = s.dm + li * 16 * LP;
    gemm_nt<1>(in, ldin, P + cl.wof(li), kin, kin, L, identity_map(), P + cl.bof(li), nullptr, a, LP, 0, wst);
    __syncthreads();
    tile_for(16, L, [&](int r, int c) {
      const float pre = a[r * LP + c];
      const float dd = leaky_slope(pre) * drop.get(li, grow0 + r, c, L);
      d[r * LP + c] = dd;
      a[r * LP + c] = pre * dd;
    });
    __syncthreads();
    in = a; ldin = LP; kin = L;
  }
  // last layer (1, L): one thread per row
  if (threadIdx.x < 16) {
    const float* w = P + cl.wof(cl.nh);
    float acc = P[cl.bof(cl.nh)];
    for (int c = 0; c < L; ++c) acc += in[threadIdx.x * LP + c] * w[c];
    s.out[threadIdx.x] = acc;
  }
  __syncthreads();
}

// first-order backward chain for a per-row output gradient dout[r] (LDS [16]).
// Calls sink(li, delta_ptr) with delta_li in LDS [16][LP] for li = nh-1 .. 0 (the caller stores what it needs;
// delta of the last layer is dout itself).  Returns a pointer to delta_0.
template <class Sink>
__device__ __forceinline__ const float* critic_bwd_chain_tile(const float* dout, const float* P, const CriticLayout& cl,
                                                              int L, const CriticLds& s, Sink sink) {
  float* cur = s.dl;
  float* nxt = s.dl + 16 * LP;
  const float* wl = P + cl.wof(cl.nh);
  const float* d = s.dm + (cl.nh - 1) * 16 * LP;
  tile_for(16, L, [&](int r, int c) { cur[r * LP + c] = dout[r] * wl[c] * d[r * LP + c]; });
  __syncthreads();
  sink(cl.nh - 1, cur);
  for (int li = cl.nh - 2; li >= 0; --li) {
    gemm_nn<1>(cur, LP, 0, P + cl.wof(li + 1), L, L, identity_map(), L, nxt, LP, false);
    __syncthreads();
    const float* dd = s.dm + li * 16 * LP;
    tile_for(16, L, [&](int r, int c) { nxt[r * LP + c] *= dd[r * LP + c]; });
    __syncthreads();
    sink(li, nxt);
    float* t = cur; cur = nxt; nxt = t;
  }
  return cur;
}

// ---------------------------------------------------------------------------------------- encoder (one tile)
// Xs [16][ldx] (S columns) -> z in Zs [16][LP].  bufG: LDS scratch >= 16*ldg floats (ldg >= 6*ENC_H),
// bufH: LDS scratch [16][ldh >= 2*ENC_H].  Optionally saves gates / h for the backward.
__device__ __forceinline__ void encoder_fwd_tile(const float* Xs, int ldx, int S, int L, const float* P, const EncLayout& el,
                                                 float* bufG, int ldg, float* bufH, int ldh, float* Zs,
                                                 float* gates_save, float* h_save, int valid, float* wst) {
  lstm_gates_tile<1>(Xs, ldx, S, P, el.dir[0], el.dir[1], ENC_H, bufG, ldg, wst);
  __syncthreads();
  lstm_cell_tile(bufG, ldg, ENC_H, 16, bufH, ldh, gates_save, valid);
  __syncthreads();
  if (h_save) tile_store(h_save, 2 * ENC_H, bufH, ldh, 16, 2 * ENC_H, valid);
  gemm_nt<1>(bufH, ldh, P + el.dense_w, 2 * ENC_H, 2 * ENC_H, L, identity_map(), P + el.dense_b, nullptr, Zs, LP, 0, wst);
  __syncthreads();
}

// ---------------------------------------------------------------------------------------- decoder (MT tiles)
struct DecSave {           // global workspace rows for this tile (null = do not save); pass p at + p*ps rows
  int ps;
  float* a0;               // [rows][DEC_D1]
  float* g0;               // [rows][8*DEC_H]
  float* h0d;              // [rows][2*DEC_H]   (post dropout)
  float* mask;             // [rows][2*DEC_H]   dropout keep-scale
  float* g1;               // [rows][8*DEC_H]
  float* h1;               // [rows][2*DEC_H]
  float* e = nullptr;      // [rows][S] tanh output (latency-chain callers: written from the last product's epilogue)
  long long* stamps = nullptr;   // development aid
};
// Zs [rows][LP] -> tanh output E in bufA [rows][ldS].  bufA/bufB: LDS, each >= rows * max(ldS, 6*DEC_H + 4).
// drop: inter-layer dropout (p = 0.2, models/tadgan.py:37); grow(r) = batch row of tile row r for the mask.
template <int MT, class RowFn>
__device__ __forceinline__ void decoder_trunk_fwd_tile(const float* Zs, int L, int S, const float* P, const DecLayout& dl,
                                                       float* bufA, float* bufB, int ldS, const DropSrc& drop, RowFn grow,
                                                       const DecSave& sv, int valid, float* wst) {
  constexpr int rows = MT * 16;
  constexpr int ldA0 = 52, ldG = 6 * DEC_H + 4, ldH = 2 * DEC_H + 4;
  // dense1
  gemm_nt<MT>(Zs, LP, P + dl.d1_w, L, L, DEC_D1, identity_map(), P + dl.d1_b, nullptr, bufB, ldA0, 0, wst);
  __syncthreads();
  if (sv.a0) tile_store_p(sv.a0, DEC_D1, sv.ps, bufB, ldA0, rows, DEC_D1, valid);
  // layer 0
  lstm_gates_tile<MT>(bufB, ldA0, DEC_D1, P, dl.l[0][0], dl.l[0][1], DEC_H, bufA, ldG, wst);
  __syncthreads();
  lstm_cell_tile(bufA, ldG, DEC_H, rows, bufB, ldH, sv.g0, valid, sv.ps);
  __syncthreads();
  if (drop.mode != 0) {
    for (int i = threadIdx.x; i < rows * (2 * DEC_H / 4); i += blockDim.x) {       // four columns per thread: one Philox evaluation
      const int r = i / (2 * DEC_H / 4), c = 4 * (i - r * (2 * DEC_H / 4));
      const float4 m = drop.get4(0, grow(r), c, 2 * DEC_H);
      float4* h = reinterpret_cast<float4*>(bufB + r * ldH + c);
      float4 hv = *h;
      hv.x *= m.x; hv.y *= m.y; hv.z *= m.z; hv.w *= m.w;
      *h = hv;
      if (sv.mask && r < valid) *reinterpret_cast<float4*>(sv.mask + prow(r, sv.ps) * 2 * DEC_H + c) = m;
    }
    __syncthreads();
  }
  if (sv.h0d) tile_store_p(sv.h0d, 2 * DEC_H, sv.ps, bufB, ldH, rows, 2 * DEC_H, valid);
  // layer 1
  lstm_gates_tile<MT>(bufB, ldH, 2 * DEC_H, P, dl.l[1][0], dl.l[1][1], DEC_H, bufA, ldG, wst);
  __syncthreads();
  lstm_cell_tile(bufA, ldG, DEC_H, rows, bufB, ldH, sv.g1, valid, sv.ps);
  __syncthreads();
  if (sv.h1) tile_store_p(sv.h1, 2 * DEC_H, sv.ps, bufB, ldH, rows, 2 * DEC_H, valid);
  // dense2 + tanh
  gemm_nt<MT>(bufB, ldH, P + dl.d2_w, 2 * DEC_H, 2 * DEC_H, S, identity_map(), P + dl.d2_b, nullptr, bufA, ldS, 0, wst);
  __syncthreads();
  tile_for(rows, S, [&](int r, int c) { bufA[r * ldS + c] = tanhf_(bufA[r * ldS + c]); });
  __syncthreads();
}

// ---------------------------------------------------------------------------------------- packed-weight forms (training)
// The same encoder / decoder trunk against the MFMA-native copies of the weights (layout.h GenPack, gemm_nt_packed).
// One LSTM layer at T = 1 from the packed gate matrices, cell included: wave w owns hidden units [16 ub, 16 ub + 16) of
// direction d (w = nb d + ub, nb = up16(H) / 16 <= 4): it computes their i, g and o tiles -- same accumulator layout, so
// the gates of one (row, unit) meet in one lane's registers -- and applies c = sig(i) tanh(g), h = sig(o) tanh(c) right
// there.  No gate tile in LDS, no separate cell pass, no barrier in between.
// Hs[rows][ldh] <- [h_fwd | h_rev]; gates_save as in lstm_cell_tile.  pb: summed biases, gate x of unit u at x up16(H) + u.
// The first batch of weights of a wave's first task, requested ahead of time (weights do not depend on activations): the
// caller issues lstm_layer_prefetch() before the previous stage's epilogue / barrier and the layer starts with its
// operands already in registers instead of an L2 round trip.
struct LstmPre { float4 bi[4], bg[4], bo[4]; };
template <bool SC1 = false>
__device__ __forceinline__ LstmPre lstm_layer_prefetch(const float* __restrict__ pk0, const float* __restrict__ pk1, int H, int K) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int Hp = (H + 15) & ~15, nb = Hp >> 4, kg = (K + 15) >> 4;
  const int task = wave < 2 * nb ? wave : 0;
  const int d = task / nb, ub = task - d * nb;
  const WeightBlocks<SC1> wb(d ? pk1 : pk0, lane);
  const int bi0 = ub * kg, bg0 = bi0 + nb * kg, bo0 = bg0 + nb * kg;
  LstmPre p;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = u < kg ? u : kg - 1;
    p.bi[u] = wb(bi0 + o); p.bg[u] = wb(bg0 + o); p.bo[u] = wb(bo0 + o);
  }
  return p;
}
// development aid (scripts/diag_gen.py): per-wave marks 40-42 inside one layer (entry, products done, first task done)
#if HYPAD_DIAG
#define LSTAMP(k) do { if (xst && (threadIdx.x & 63) == 0) xst[(k) * 8 + (threadIdx.x >> 6)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define LSTAMP(k) do { (void)xst; } while (0)
#endif
template <int MT, bool PRE = false, bool SC1 = false>
__device__ __forceinline__ void lstm_layer_fwd_packed(const float* __restrict__ As, int lda, int K, const float* __restrict__ pk0,
                                                      const float* __restrict__ pb0, const float* __restrict__ pk1,
                                                      const float* __restrict__ pb1, int H, float* __restrict__ Hs, int ldh,
                                                      float* __restrict__ gates_save, int valid, int ps = 16,
                                                      const LstmPre& pre = LstmPre{}, float* __restrict__ h_out = nullptr,
                                                      long long* xst = nullptr) {      // h_out: global mirror of Hs
  const int lane = threadIdx.x & 63, wave = wave_id(), nwaves = blockDim.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int Hp = (H + 15) & ~15, nb = Hp >> 4, kg = (K + 15) >> 4;
  // one task = 16 units of one direction: the i, g, o gate tiles side by side, the cell in the epilogue
  auto run = [&](int task, auto first_from_pre) __attribute__((always_inline)) {
    const int d = task / nb, ub = task - d * nb;
    const WeightBlocks<SC1> wb(d ? pk1 : pk0, lane);
    const int bi0 = ub * kg, bg0 = bi0 + nb * kg, bo0 = bg0 + nb * kg;
    const float* pb = d ? pb1 : pb0;
    // (the biases are requested here, ahead of the products: behind them they are an exposed memory round trip per task)
    const int jj = 16 * ub + j;
    float b_i = 0.f, b_g = 0.f, b_o = 0.f;
    if constexpr (PRE) {
      b_i = weight_scalar<SC1>(pb + (jj < H ? jj : 0)); b_g = weight_scalar<SC1>(pb + Hp + (jj < H ? jj : 0));
      b_o = weight_scalar<SC1>(pb + 2 * Hp + (jj < H ? jj : 0));
    }
    f32x4 ai[MT], ag[MT], ao[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { ai[m] = f32x4{0.f, 0.f, 0.f, 0.f}; ag[m] = ai[m]; ao[m] = ai[m]; }
    auto consume = [&](auto nb_, const float4* bi, const float4* bg, const float4* bo, int g0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < decltype(nb_)::value; ++u) {
        if (g0 + u < kg) {                             // wave-uniform
          const int k0 = 16 * (g0 + u) + 4 * q;
          const int ka = k0 < lda - 4 ? k0 : lda - 4;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float4 a = *reinterpret_cast<const float4*>(As + (m * 16 + j) * lda + ka);
            const float a0 = k0 < K ? a.x : 0.f, a1 = k0 + 1 < K ? a.y : 0.f, a2 = k0 + 2 < K ? a.z : 0.f, a3 = k0 + 3 < K ? a.w : 0.f;
            ai[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bi[u].x, ai[m], 0, 0, 0);
            ag[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bg[u].x, ag[m], 0, 0, 0);
            ao[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bo[u].x, ao[m], 0, 0, 0);
            ai[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bi[u].y, ai[m], 0, 0, 0);
            ag[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bg[u].y, ag[m], 0, 0, 0);
            ao[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bo[u].y, ao[m], 0, 0, 0);
            ai[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bi[u].z, ai[m], 0, 0, 0);
            ag[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bg[u].z, ag[m], 0, 0, 0);
            ao[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bo[u].z, ao[m], 0, 0, 0);
            ai[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bi[u].w, ai[m], 0, 0, 0);
            ag[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bg[u].w, ag[m], 0, 0, 0);
            ao[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bo[u].w, ao[m], 0, 0, 0);
          }
        }
      }
    };
    auto fetch = [&](auto nb_, float4* bi, float4* bg, float4* bo, int g0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < decltype(nb_)::value; ++u) {
        const int o = g0 + u < kg ? g0 + u : kg - 1;
        bi[u] = wb(bi0 + o); bg[u] = wb(bg0 + o); bo[u] = wb(bo0 + o);
      }
    };
    mfma_prio_begin<PRE>();
    if constexpr (decltype(first_from_pre)::value) {
      // latency-chain callers: the first batch is already in registers (lstm_layer_prefetch), and two batches of four
      // k-groups stay in flight -- the next one is requested before the current one is consumed
      float4 xi[4], xg[4], xo[4], yi[4], yg[4], yo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { xi[u] = pre.bi[u]; xg[u] = pre.bg[u]; xo[u] = pre.bo[u]; }
      constexpr std::integral_constant<int, 4> four{};
      for (int g0 = 0; g0 < kg; g0 += 8) {
        if (g0 + 4 < kg) fetch(four, yi, yg, yo, g0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        consume(four, xi, xg, xo, g0);
        if (g0 + 8 < kg) fetch(four, xi, xg, xo, g0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        if (g0 + 4 < kg) consume(four, yi, yg, yo, g0 + 4);
      }
    } else {
      // throughput callers (many workgroups per CU hide the latency): one batch in flight, half the registers; with two row tiles a
      // weight block feeds 24 MFMAs and two k-groups per batch are enough (the accumulators of the second tile take the registers)
      constexpr int NB = MT >= 2 ? 2 : 4;
      constexpr std::integral_constant<int, NB> nb_c{};
      for (int g0 = 0; g0 < kg; g0 += NB) {
        float4 bi[NB], bg[NB], bo[NB];
        fetch(nb_c, bi, bg, bo, g0);
        __builtin_amdgcn_sched_barrier(0);
        consume(nb_c, bi, bg, bo, g0);
      }
    }
    mfma_prio_end<PRE>();
    LSTAMP(41);
    if (jj < H) {
      if constexpr (!PRE) { b_i = weight_scalar<SC1>(pb + jj); b_g = weight_scalar<SC1>(pb + Hp + jj); b_o = weight_scalar<SC1>(pb + 2 * Hp + jj); }      // (throughput callers: fewer live registers)
      // saved rows through buffer addressing: the lane's offset (row 4 q, this unit) once, row and gate as scalar / constant offsets
      const GBuf hb(h_out), gb(gates_save);
      // (saved gates of the packed path: [row][dir][unit][i, g, o, tanh c] -- one 16-byte store here, one 16-byte load in the
      // cell backward's epilogue -- not the [dir][gate][unit] rows of the stand-alone LSTM entry points)
      const int vo_h = (4 * q * 2 * H + d * H + jj) * 4, vo_g = (4 * q * 8 * H + (d * H + jj) * 4) * 4;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m * 16 + 4 * q + r;
          const float gi = sigmoidf_(ai[m][r] + b_i), gg = tanhf_(ag[m][r] + b_g), go = sigmoidf_(ao[m][r] + b_o);
          const float tc = tanhf_(gi * gg);
          Hs[row * ldh + d * H + jj] = go * tc;
          const int prs = m * ps + r;                      // prow(row, ps) = m ps + 4 q + r: the part that is not the lane's
          if (h_out && row < valid) hb.st(go * tc, vo_h, prs * 2 * H * 4);
          if (gates_save && row < valid) {
            gb.st4(make_float4(gi, gg, go, tc), vo_g + prs * 8 * H * 4);      // (row offset in the vector offset: see GBuf::st4)
          }
        }
    }
  };
  LSTAMP(40);
  if (wave < 2 * nb) run(wave, std::integral_constant<bool, PRE>{});
  LSTAMP(42);
  for (int task = wave + nwaves; task < 2 * nb; task += nwaves) run(task, std::false_type{});
}

// PRE: the caller requested the layer's first weights earlier (lstm_layer_prefetch on gp.enc_g) and passes them in.
template <bool PRE = false, bool SC1 = false, int MT = 1>
__device__ __forceinline__ void encoder_fwd_tile_packed(const float* Xs, int ldx, int S, int L, const float* pk, const GenPack& gp,
                                                        float* bufG, int ldg, float* bufH, int ldh, float* Zs,
                                                        float* gates_save, float* h_save, int valid, const LstmPre& pre = LstmPre{}) {
  (void)bufG; (void)ldg;
  PackedPre pred{};
  if constexpr (PRE) pred = gemm_nt_prefetch<SC1>(pk + gp.enc_d, 2 * ENC_H, L);   // the dense layer's weights, one stage ahead
  lstm_layer_fwd_packed<MT, PRE, SC1>(Xs, ldx, S, pk + gp.enc_g[0], pk + gp.enc_gb[0], pk + gp.enc_g[1], pk + gp.enc_gb[1], ENC_H, bufH, ldh,
                                gates_save, valid, 16, pre, h_save);          // (h_save written from the epilogue)
  __syncthreads();
  gemm_nt_packed<MT, PRE, ActIdentity, SC1>(bufH, ldh, 2 * ENC_H, L, pk + gp.enc_d, pk + gp.enc_db, Zs, LP, 0, 0, pred);
  __syncthreads();
}
// PRE: the caller requested d1's first weights earlier (gemm_nt_prefetch on gp.d1) and passes them in.  next_W (may be
// null): packed weights of the product that follows the trunk; their first batch is requested before the last product
// here and returned.
template <int MT, bool PRE = false, bool SC1 = false, class RowFn>
__device__ __forceinline__ PackedPre decoder_trunk_fwd_tile_packed(const float* Zs, int L, int S, const float* pk, const GenPack& gp,
                                                                   float* bufA, float* bufB, int ldS, const DropSrc& drop, RowFn grow,
                                                                   const DecSave& sv, int valid, const PackedPre& pred1 = PackedPre{},
                                                                   const float* next_W = nullptr, int next_K = 0, int next_N = 0,
                                                                   float* Eout = nullptr) {            // Eout: where the tanh output goes (default bufA)
  constexpr int rows = MT * 16;
  constexpr int ldA0 = 52, ldH = 2 * DEC_H + 4;
// development aid (scripts/diag_gen.py): per-wave shader-clock marks 16-19, 27-31 of the generator kernel's timeline
#define TSTAMP(k) do { if (sv.stamps && (threadIdx.x & 63) == 0) sv.stamps[(k) * 8 + (threadIdx.x >> 6)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  TSTAMP(16);
  // PRE (latency-chain callers): every product's first weights are requested one stage ahead -- they do not depend on the
  // activations.  Throughput callers (scoring, the critic phase's precompute) keep the registers for occupancy instead.
  LstmPre pre0{}, pre1{};
  PackedPre pre2{};
  if constexpr (PRE) pre0 = lstm_layer_prefetch<SC1>(pk + gp.l_g[0][0], pk + gp.l_g[0][1], DEC_H, DEC_D1);
  // (the saved activations go to the workspace from the epilogues: valid == rows for the callers that save)
  gemm_nt_packed<MT, PRE, ActIdentity, SC1>(Zs, LP, L, DEC_D1, pk + gp.d1, pk + gp.d1b, bufB, ldA0, 0, 0, pred1, ActIdentity{}, nullptr, 0, sv.a0, DEC_D1, sv.ps);
  TSTAMP(17);
  __syncthreads();
  // layer 0: input a0 in bufB [rows][ldA0] -> h0 in bufA [rows][ldH] (the cell runs in the gate product's epilogue)
  if constexpr (PRE) pre1 = lstm_layer_prefetch<SC1>(pk + gp.l_g[1][0], pk + gp.l_g[1][1], DEC_H, 2 * DEC_H);
  lstm_layer_fwd_packed<MT, PRE, SC1>(bufB, ldA0, DEC_D1, pk + gp.l_g[0][0], pk + gp.l_gb[0][0], pk + gp.l_g[0][1], pk + gp.l_gb[0][1], DEC_H, bufA,
                                  ldH, sv.g0, valid, sv.ps, pre0);
  TSTAMP(18);
  __syncthreads();
  TSTAMP(19);
  if (drop.mode != 0) {
    for (int i = threadIdx.x; i < rows * (2 * DEC_H / 4); i += blockDim.x) {
      const int r = i / (2 * DEC_H / 4), c = 4 * (i - r * (2 * DEC_H / 4));
      const float4 m = drop.get4(0, grow(r), c, 2 * DEC_H);
      float4* h = reinterpret_cast<float4*>(bufA + r * ldH + c);
      float4 hv = *h;
      hv.x *= m.x; hv.y *= m.y; hv.z *= m.z; hv.w *= m.w;
      *h = hv;
      if (sv.mask && r < valid) *reinterpret_cast<float4*>(sv.mask + prow(r, sv.ps) * 2 * DEC_H + c) = m;
      if (sv.h0d && r < valid) *reinterpret_cast<float4*>(sv.h0d + prow(r, sv.ps) * 2 * DEC_H + c) = hv;      // saved from the same pass
    }
    __syncthreads();
  } else if (sv.h0d) {
    tile_store_p(sv.h0d, 2 * DEC_H, sv.ps, bufA, ldH, rows, 2 * DEC_H, valid);
  }
  TSTAMP(27);
  TSTAMP(28);
  // layer 1: h0 (dropped) in bufA -> h1 in bufB
  if constexpr (PRE) pre2 = gemm_nt_prefetch<SC1>(pk + gp.d2, 2 * DEC_H, S);
  lstm_layer_fwd_packed<MT, PRE, SC1>(bufA, ldH, 2 * DEC_H, pk + gp.l_g[1][0], pk + gp.l_gb[1][0], pk + gp.l_g[1][1], pk + gp.l_gb[1][1], DEC_H, bufB,
                                  ldH, sv.g1, valid, sv.ps, pre1, sv.h1, sv.stamps);
  TSTAMP(29);
  __syncthreads();
  TSTAMP(30);
  PackedPre nxt{};
  if (next_W) nxt = gemm_nt_prefetch<SC1>(next_W, next_K, next_N);
  struct Tanh { __device__ __forceinline__ float operator()(float v) const { return tanhf_(v); } };
  gemm_nt_packed<MT, PRE, Tanh, SC1>(bufB, ldH, 2 * DEC_H, S, pk + gp.d2, pk + gp.d2b, Eout ? Eout : bufA, ldS, 0, 0, pre2, Tanh{}, nullptr, 0,
                                sv.e, S, sv.ps);                               // tanh in the epilogue
  TSTAMP(31);
  __syncthreads();
  return nxt;
#undef TSTAMP
}

// Moebius head on LDS rows: Us[rows][ld] (u = e W_h^T) -> in place r = project(mobius_add(expmap0(u), bias)).
// Four rows per wave, one per 16-lane DPP row (RowT<16, EPL>): the chain's per-row scalars cost one instruction for four
// rows and its ~5 reductions are 4 DPP steps each.
__device__ __forceinline__ void head_rows_tile(float* Us, int ld, int rows, int S, const float* bias_g) {
  const int lane = threadIdx.x & 63, wave = wave_id(), nw = blockDim.x >> 6, sub = lane >> 4;
  epl16_dispatch(S, [&](auto tag) {
    using R = RowT<16, decltype(tag)::value>;
    const R b = row_load<R>(bias_g, S, lane);
    for (int r0 = wave * 4; r0 < rows; r0 += nw * 4) {
      const int r = r0 + sub;
      const R u = row_load<R>(Us + (r < rows ? r : rows - 1) * ld, S, lane);
      const R o = head_row(u, b);
      if (r < rows) row_store(Us + r * ld, o, S, lane);
    }
  });
}

// copy n floats (n % 4 == 0, 16-byte aligned) global -> LDS: the weights of a small network, whole workgroup
__device__ __forceinline__ void stage_params(float* __restrict__ dst, const float* __restrict__ src, int n) {
  for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4)
    *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
}

}  // namespace hypad
