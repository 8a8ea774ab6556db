// Critic phase of an epoch (train.py:306-328) restructured for the GPU.
//
// During the n_critics passes the generator is frozen (train.py:306-309), so decoder(z_i) and encoder(x_i) of ALL
// critic iterations of the phase do not depend on anything the phase updates.  They are hoisted out of the
// sequential chain and computed by ONE wide launch (`critic_phase_precompute_kernel`: iterations x row tiles x
// signals x {decoder, encoder} workgroups -- 1 160 workgroups for 29 batches x 5 passes, it fills the chip).
// What remains sequential -- critic forward on real / fake / interpolated rows, the whole-batch gradient penalty, its
// double backward, the weight gradients and Adam -- is ~0.8 MFLOP per iteration and fits ONE workgroup per
// (signal, critic): `critic_fused_pair_kernel` keeps everything in LDS / registers (no workspace round trip, no
// inter-workgroup reduction for the whole-batch norm, one launch per (critic_x || critic_z) pair instead of three).
//
// Gradient-penalty algebra used to make it single-pass: the second-order chain is linear in u = coef * g with
// coef = 20 (||g|| - 1) / ||g|| known only after all rows; the kernel accumulates the GP part of every weight
// gradient with the *unscaled* g in its own accumulators and applies coef at the end (oracle/manual.py
// critic_gp_pairs is linear in `ugrad`).
#include <hip/hip_runtime.h>

#include "../../include/hypad.h"
#include "train_common.h"

using namespace hypad;
using namespace hypad::train;

namespace {

constexpr int FT = 512;     // threads of the fused critic kernel (8 waves)
constexpr int MAXT = 4;     // weight tiles per wave (critic_x: 28 tiles over 8 waves)

struct PhaseArgs {
  float* gen_pre;           // (n_signals, n_iters, B, S)  decoder(z_it)
  float* zenc_pre;          // (n_signals, n_iters, B, L)  encoder(x_it)
  const int32_t* row_index; // (n_iters, B)
  int n_iters;
  int it;                   // iteration index of a fused launch
  long long* stamps;        // development aid: per-stage shader-clock stamps of the fused kernel (64 per critic), or null
};
long long* g_stamps = nullptr;
#define STAMP(k) do { if (ph.stamps && threadIdx.x == 0) ph.stamps[blockIdx.z * 64 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

// ---------------------------------------------------------------------------------------------- precompute
__global__ __launch_bounds__(TB) void critic_phase_precompute_kernel(IterArgs a, PhaseArgs ph) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int sig = blockIdx.y, tile = blockIdx.x, S = a.S, L = a.L, B = a.B;
  const int it = blockIdx.z >> 1, role = blockIdx.z & 1;
  const LdsPlan lp = lds_plan(S, 16, 16, 0);
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  float* wst = smem + lp.wst;
  const uint32_t tick = (uint32_t)a.counters[3] + (uint32_t)it;
  const int g0 = tile * 16;
  if (role == 0) {          // x_ = decoder(z), train-mode dropout (train.py:24-33)
    const DecLayout dl = dec_layout(S, L, a.hyperbolic);
    const float* PD = a.P.dec + (int64_t)sig * a.pd;
    tile_for(16, L, [&](int r, int c) { zs[r * LP + c] = rng_normal(a.seed, tick, RS_Z, (uint32_t)sig, (uint32_t)((g0 + r) * L + c)); });
    __syncthreads();
    DropSrc ddrop = drop_src(a, sig, nullptr, RS_DROP_DEC0, tick, 0.2f);
    DecSave none{16, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    decoder_trunk_fwd_tile<1>(zs, L, S, PD, dl, bufA, bufB, lp.ldS, ddrop, [g0](int r) { return g0 + r; }, none, 16, wst);
    float* gen = bufA;
    if (a.hyperbolic) {
      gemm_nt<1>(bufA, lp.ldS, PD + dl.head_w, S, S, S, identity_map(), nullptr, nullptr, bufB, lp.ldS, 0, wst);
      __syncthreads();
      head_rows_tile(bufB, lp.ldS, 16, S, PD + dl.head_b);
      __syncthreads();
      gen = bufB;
    }
    tile_store(ph.gen_pre + (((int64_t)sig * ph.n_iters + it) * B + g0) * S, S, gen, lp.ldS, 16, S, 16);
  } else {                  // z_ = encoder(x)  (train.py:111)
    const EncLayout el = enc_layout(S, L);
    const float* PE = a.P.enc + (int64_t)sig * a.pe;
    tile_load_rows(xs, lp.ldS, a.x + sig * a.x_sig_stride, S, ph.row_index + (int64_t)it * B, g0, 16, S, 16);
    __syncthreads();
    encoder_fwd_tile(xs, lp.ldS, S, L, PE, el, bufA, 6 * ENC_H + 4, bufB, 2 * ENC_H + 4, zs, nullptr, nullptr, 16, wst);
    tile_store(ph.zenc_pre + (((int64_t)sig * ph.n_iters + it) * B + g0) * L, L, zs, LP, 16, L, 16);
  }
}

// ---------------------------------------------------------------------------------------------- fused critic iteration
// LDS plan (floats).  Row strides are (multiple of 16) + 4: every matrix product of the iteration runs on
// v_mfma_f32_16x16x4_f32 with ds_read_b128 operand fetches (lane (i, q) supplies k = 16 g + 4 q + s to the s-th MFMA of
// k-group g -- the reduction index may be permuted as long as A and B agree), which needs 16-byte aligned rows and
// zero padding up to the next multiple of 16 columns.  The padding is written once (the whole region is zeroed at kernel
// start) and never touched again.
struct FusedLds {
  int in0;      // [48][ldin]  rows 0-15 real, 16-31 fake, 32-47 interpolated; after the first backward rows 32-47 hold g
  int act;      // [nh][48][LQ] layer outputs; rows 32-47 are overwritten by the second-order chain ep_li
  int dm;       // [nh][48][LQ] leaky'(pre) * dropout scale
  int dl;       // [nh+1][48][LQ] first-order deltas of every layer (dl[nh]: column 0 = d loss / d out)
  int w0;       // [L][ldin]
  int wh;       // [nh-1][L][LQ]
  int wl;       // [LQ]
  int bias;     // [nh+1][LQ]
  int dbias;    // [nh+1][LQ]
  int red;      // [64]
  int total, ldin, LQ, Kin, Lp;
};
HD int up16(int n) { return (n + 15) & ~15; }
HD FusedLds fused_lds(int in_dim, int L, int nh) {
  FusedLds f; int o = 0;
  f.Kin = up16(in_dim); f.Lp = up16(L);
  f.ldin = f.Kin + 4; f.LQ = f.Lp + 4;
  f.in0 = o; o += 48 * f.ldin;
  f.act = o; o += nh * 48 * f.LQ;
  f.dm = o; o += nh * 48 * f.LQ;
  f.dl = o; o += (nh + 1) * 48 * f.LQ;
  f.w0 = o; o += L * f.ldin;
  f.wh = o; o += (nh - 1) * L * f.LQ;
  f.wl = o; o += f.LQ;
  f.bias = o; o += (nh + 1) * f.LQ;
  f.dbias = o; o += (nh + 1) * f.LQ;
  f.red = o; o += 64;
  f.total = o;
  return f;
}

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}
// out[r][n] = sum_k A[r][k] W[n][k];  A: LDS [16 RT][lda], W: LDS [N][ldw], both zero-padded to Kp columns.
// epi(r, n, value) is called for n < 16 ceil(N / 16); columns >= N repeat column N - 1 (the caller drops them).
template <class Epi>
__device__ __forceinline__ void lds_gemm_nt(const float* A, int lda, int RT, const float* W, int ldw, int N, int Kp, int wave, int lane, Epi epi) {
  const int j = lane & 15, q = lane >> 4, CT = (N + 15) >> 4;
  for (int t = wave; t < RT * CT; t += FT / 64) {
    const int rt = t / CT, ct = t - rt * CT;
    int n = ct * 16 + j; n = n < N ? n : N - 1;
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
// out[r][c] = sum_o A[r][o] W[o][c];  A: LDS [16 RT][lda] zero-padded to Kp columns, W: LDS [No][ldw], c < C.
template <class Epi>
__device__ __forceinline__ void lds_gemm_nn(const float* A, int lda, int RT, const float* W, int ldw, int No, int C, int Kp, int wave, int lane,
                                            Epi epi) {
  const int j = lane & 15, q = lane >> 4, CT = (C + 15) >> 4;
  for (int t = wave; t < RT * CT; t += FT / 64) {
    const int rt = t / CT, ct = t - rt * CT;
    int c = ct * 16 + j; c = c < C ? c : C - 1;
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

// four consecutive uniforms / two consecutive normals of a stream: the same numbers rng_uniform / rng_normal give for
// idx = 4 group + e / 2 pair + e, one Philox evaluation instead of four / two
__device__ __forceinline__ float4 rng_uniform4(uint64_t seed, uint32_t tick, uint32_t stream, uint32_t sig, uint32_t group) {
  Philox ph(seed);
  const uint4 r = ph(group, stream, tick, sig);
  return make_float4(u32_to_unit(r.x), u32_to_unit(r.y), u32_to_unit(r.z), u32_to_unit(r.w));
}
__device__ __forceinline__ float2 rng_normal2(uint64_t seed, uint32_t tick, uint32_t stream, uint32_t sig, uint32_t pair) {
  Philox ph(seed);
  const uint4 r = ph(pair, stream, tick, sig);
  const float a = sqrtf(-2.0f * __logf(u32_to_unit_open(r.x))) * __cosf(6.28318530717958647692f * u32_to_unit(r.y));
  const float b = sqrtf(-2.0f * __logf(u32_to_unit_open(r.z))) * __cosf(6.28318530717958647692f * u32_to_unit(r.w));
  return make_float2(a, b);
}

constexpr int NG = 2;       // element groups (4 consecutive elements of the 16 x in_dim chunk) per thread: 16 * 256 / 4 / 512

template <bool IS_X>
__device__ __forceinline__ void critic_fused_body(const IterArgs& a, const PhaseArgs& ph, float* smem) {
  const int sig = blockIdx.y, L = a.L, B = a.B, S = a.S;
  const int in_dim = IS_X ? S : L;
  constexpr int nh = IS_X ? 4 : 2;
  const CriticLayout cl = IS_X ? cx_layout(S, L) : cz_layout(L);
  const FusedLds fl = fused_lds(in_dim, L, nh);
  const int ldin = fl.ldin, LQ = fl.LQ, Kin = fl.Kin, Lp = fl.Lp;
  float* in0 = smem + fl.in0; float* act = smem + fl.act; float* dm = smem + fl.dm; float* dl = smem + fl.dl;
  float* w0 = smem + fl.w0; float* wh = smem + fl.wh; float* wl = smem + fl.wl; float* bias = smem + fl.bias;
  float* dbias = smem + fl.dbias; float* red = smem + fl.red;
  float* Pg = (IS_X ? a.P.cx + (int64_t)sig * a.pcx : a.P.cz + (int64_t)sig * a.pcz);
  float* Mg = (IS_X ? a.M.cx + (int64_t)sig * a.pcx : a.M.cz + (int64_t)sig * a.pcz);
  float* Vg = (IS_X ? a.V.cx + (int64_t)sig * a.pcx : a.V.cz + (int64_t)sig * a.pcz);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const uint32_t tick = (uint32_t)a.counters[3] + (uint32_t)ph.it;     // counters advance once per phase (no reader/writer race)
  const int step = a.counters[a.opt] + ph.it + 1;
  const float invB = 1.f / B;
  const float keep = 1.f / (1.f - cl.p_drop);
  const int32_t* ridx = ph.row_index + (int64_t)ph.it * B;
  const float* fake_rows = IS_X ? ph.gen_pre + ((int64_t)sig * ph.n_iters + ph.it) * B * S
                                : ph.zenc_pre + ((int64_t)sig * ph.n_iters + ph.it) * B * L;
  const float* xbase = a.x + sig * a.x_sig_stride;
  const int ngroups = 4 * in_dim;              // 16 rows * in_dim / 4

  STAMP(0);
  // ---- zero the whole plan (padding!), stage the weights with padded strides, constant d loss / d out
  for (int i = threadIdx.x; i < fl.total; i += FT) smem[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < L * in_dim; i += FT) { const int n = i / in_dim, k = i - n * in_dim; w0[n * ldin + k] = Pg[cl.w[0] + i]; }
  for (int li = 1; li < nh; ++li)
    for (int i = threadIdx.x; i < L * L; i += FT) { const int n = i / L, k = i - n * L; wh[((li - 1) * L + n) * LQ + k] = Pg[cl.w[li] + i]; }
  for (int i = threadIdx.x; i < L; i += FT) wl[i] = Pg[cl.w[nh] + i];
  for (int i = threadIdx.x; i < (nh + 1) * L; i += FT) { const int li = i / L, c = i - li * L; if (li < nh || c == 0) bias[li * LQ + c] = Pg[cl.b[li] + c]; }
  if (threadIdx.x < 48) { const int p = threadIdx.x >> 4; dl[(nh * 48 + threadIdx.x) * LQ] = p == 0 ? -invB : p == 1 ? invB : 1.f; }

  // weight tiles: tile t = wave + 8 i belongs to this wave (accumulators stay in its registers for the whole iteration)
  const int tk0 = (in_dim + 15) >> 4, tn = (L + 15) >> 4, tkh = (L + 15) >> 4;
  const int tiles0 = tn * tk0, tilesh = tn * tkh, ntiles = tiles0 + (nh - 1) * tilesh + tkh;
  auto tile_desc = [&](int t, int& li, int& n0, int& k0) {
    if (t < tiles0) { li = 0; n0 = (t / tk0) * 16; k0 = (t % tk0) * 16; return; }
    t -= tiles0;
    li = 1 + t / tilesh;
    t -= (li - 1) * tilesh;
    n0 = (t / tkh) * 16; k0 = (t % tkh) * 16;
  };
  f32x4 acc_rf[MAXT], acc_gp[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) { acc_rf[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_gp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float gsq = 0.f;

  // prefetch of a chunk's real / fake rows (critic_x: both from HBM; critic_z: fake only, real is noise)
  float pre_x[NG][4], pre_f[NG][4];
  auto prefetch = [&](int g0) {
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int gi = threadIdx.x + u * FT;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = 4 * gi + e;
        const int r = f / in_dim, c = f - r * in_dim;
        const bool ok = gi < ngroups;
        if (IS_X) pre_x[u][e] = ok ? xbase[(int64_t)(ridx ? ridx[g0 + r] : g0 + r) * S + c] : 0.f;
        pre_f[u][e] = ok ? fake_rows[(int64_t)(g0 + r) * in_dim + c] : 0.f;
      }
    }
  };
  prefetch(0);
  __syncthreads();
  STAMP(1);
  int sk = 2;

  for (int g0 = 0; g0 < B; g0 += 16) {
    // ---- P0: real / fake / interpolated rows, dropout scales of every layer and pass
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int gi = threadIdx.x + u * FT;
      if (gi < ngroups) {
        const uint32_t grp = (uint32_t)(g0 * in_dim) / 4 + gi;
        const float4 al = rng_uniform4(a.seed, tick, RS_ALPHA, (uint32_t)sig, grp);
        const float alv[4] = {al.x, al.y, al.z, al.w};
        float xr[4];
        if (IS_X) {
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[e] = pre_x[u][e];
        } else {
          const float2 n0 = rng_normal2(a.seed, tick, RS_Z, (uint32_t)sig, 2 * grp), n1 = rng_normal2(a.seed, tick, RS_Z, (uint32_t)sig, 2 * grp + 1);
          xr[0] = n0.x; xr[1] = n0.y; xr[2] = n1.x; xr[3] = n1.y;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 4 * gi + e;
          const int r = f / in_dim, c = f - r * in_dim;
          const float xv = xr[e], fv = pre_f[u][e];
          in0[r * ldin + c] = xv;
          in0[(16 + r) * ldin + c] = fv;
          in0[(32 + r) * ldin + c] = alv[e] * xv + (1.f - alv[e]) * fv;
        }
      }
    }
    if (g0 < 16) STAMP(sk++);
    if (g0 + 16 < B) prefetch(g0 + 16);
    {
      const int per = 4 * L;                    // groups per (pass, layer)
      for (int w = threadIdx.x; w < 3 * nh * per; w += FT) {
        const int pl = w / per, gi = w - pl * per;
        const int p = pl / nh, li = pl - p * nh;
        float v[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.drop_mode == 2) {
          const int pass = p == 2 ? 2 : (IS_X ? p : 1 - p);          // stream numbering of the per-iteration entry points
          const float4 uu = rng_uniform4(a.seed, tick, RS_DROP_CRITIC + 8 * pass + li, (uint32_t)sig, (uint32_t)(g0 * L) / 4 + gi);
          v[0] = uu.x >= cl.p_drop ? keep : 0.f; v[1] = uu.y >= cl.p_drop ? keep : 0.f;
          v[2] = uu.z >= cl.p_drop ? keep : 0.f; v[3] = uu.w >= cl.p_drop ? keep : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 4 * gi + e;
          const int r = f / L, c = f - r * L;
          dm[(li * 48 + p * 16 + r) * LQ + c] = v[e];
        }
      }
    }
    __syncthreads();
    if (g0 < 16) STAMP(sk++);
    // ---- P1..: forward, 48 rows.  The last hidden layer's epilogue also starts the backward chain.
    for (int li = 0; li < nh; ++li) {
      const float* A = li == 0 ? in0 : act + (li - 1) * 48 * LQ;
      const float* W = li == 0 ? w0 : wh + (li - 1) * L * LQ;
      float* ao = act + li * 48 * LQ; float* dmo = dm + li * 48 * LQ;
      const float* bb = bias + li * LQ;
      float* dtop = dl + (nh - 1) * 48 * LQ;
      const float* dout = dl + nh * 48 * LQ;
      lds_gemm_nt(A, li == 0 ? ldin : LQ, 3, W, li == 0 ? ldin : LQ, L, li == 0 ? Kin : Lp, wave, lane, [&](int r, int c, float v) {
        if (c < L) {
          const float pre = v + bb[c];
          const float dd = leaky_slope(pre) * dmo[r * LQ + c];
          dmo[r * LQ + c] = dd;
          ao[r * LQ + c] = pre * dd;
          if (li == nh - 1) dtop[r * LQ + c] = dout[r * LQ] * wl[c] * dd;
        }
      });
      __syncthreads();
      if (g0 < 16) STAMP(sk++);
    }
    // ---- critic outputs (loss terms) on the last wave, which owns no tile of the next products
    if (wave == FT / 64 - 1) {
      float o = 0.f;
      if (lane < 32) {
        const float* x = act + ((nh - 1) * 48 + lane) * LQ;
        o = bias[nh * LQ];
        for (int c = 0; c < L; ++c) o += x[c] * wl[c];
      }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) o += __shfl_xor(o, off, 64);
      if (lane == 0) red[32] += o;
      if (lane == 16) red[33] += o;
    }
    // ---- first-order backward chain, all 48 rows, every layer's delta kept
    for (int li = nh - 2; li >= 0; --li) {
      float* dst = dl + li * 48 * LQ; const float* dmo = dm + li * 48 * LQ;
      lds_gemm_nn(dl + (li + 1) * 48 * LQ, LQ, 3, wh + li * L * LQ, LQ, L, L, Lp, wave, lane,
                  [&](int r, int c, float v) { if (c < L) dst[r * LQ + c] = v * dmo[r * LQ + c]; });
      __syncthreads();
      if (g0 < 16) STAMP(sk++);
    }
    // ---- g = delta_0 W_0 on the interpolated rows (unscaled) -> in0 rows 32-47, sum of squares
    lds_gemm_nn(dl + 32 * LQ, LQ, 1, w0, ldin, L, in_dim, Lp, wave, lane, [&](int r, int c, float v) {
      if (c < in_dim) { in0[(32 + r) * ldin + c] = v; gsq += v * v; }
    });
    __syncthreads();
    if (g0 < 16) STAMP(sk++);
    // ---- unscaled second-order chain: ep_0 = (g W_0^T) * dm_0, ep_li = (ep_{li-1} W_li^T) * dm_li  -> act rows 32-47
    for (int li = 0; li < nh; ++li) {
      const float* A = li == 0 ? in0 + 32 * ldin : act + ((li - 1) * 48 + 32) * LQ;
      const float* W = li == 0 ? w0 : wh + (li - 1) * L * LQ;
      float* eo = act + (li * 48 + 32) * LQ; const float* dmo = dm + (li * 48 + 32) * LQ;
      lds_gemm_nt(A, li == 0 ? ldin : LQ, 1, W, li == 0 ? ldin : LQ, L, li == 0 ? Kin : Lp, wave, lane,
                  [&](int r, int c, float v) { if (c < L) eo[r * LQ + c] = v * dmo[r * LQ + c]; });
      __syncthreads();
      if (g0 < 16) STAMP(sk++);
    }
    // ---- bias gradients (real + fake rows only: LeakyReLU'' = 0 leaves the GP rows without a bias term)
    for (int idx = threadIdx.x; idx < (nh + 1) * L; idx += FT) {
      const int li = idx / L, c = idx - li * L;
      if (li < nh || c == 0) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; r += 2) { s0 += dl[(li * 48 + r) * LQ + c]; s1 += dl[(li * 48 + r + 1) * LQ + c]; }
        dbias[li * LQ + c] += s0 + s1;
      }
    }
    if (g0 < 16) STAMP(sk++);
    // ---- weight-gradient tiles: dW += left^T right over this chunk's rows (rows 0-31 -> acc_rf, GP rows -> acc_gp)
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = wave + (FT / 64) * i;
      if (t < ntiles) {
        int li, n0, k0;
        tile_desc(t, li, n0, k0);
        const int N = li == nh ? 1 : L, K = li == 0 ? in_dim : L;
        const int nj = n0 + j < N ? n0 + j : N - 1, kj = k0 + j < K ? k0 + j : K - 1;   // clamped; dropped at the update
        const float* left = dl + li * 48 * LQ + nj;
        const float* right = li == 0 ? in0 + kj : act + (li - 1) * 48 * LQ + kj;
        const int ldr = li == 0 ? ldin : LQ;
        float la[12], rb[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) { la[u] = left[(4 * u + q) * LQ]; rb[u] = right[(4 * u + q) * ldr]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc_rf[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc_rf[i], 0, 0, 0);
#pragma unroll
        for (int u = 8; u < 12; ++u) acc_gp[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc_gp[i], 0, 0, 0);
      }
    }
    __syncthreads();
    if (g0 < 16) STAMP(sk++);
  }
  STAMP(40);

  // ---- whole-batch norm, loss, Adam
  float tot = wave_sum(gsq);
  if (lane == 0) red[wave] = tot;
  __syncthreads();
  float gsum = 0.f;
  for (int w = 0; w < FT / 64; ++w) gsum += red[w];
  const float nrm = sqrtf(gsum + 1e-12f);
  const float gp = (nrm - 1.f) * (nrm - 1.f);
  const float coef = 20.f * (nrm - 1.f) / nrm;
  if (threadIdx.x == 0) {
    float* lo = a.losses + sig * a.loss_sig_stride;
    const float sreal = red[32], sfake = red[33];
    lo[0] = sfake * invB - sreal * invB + 10.f * gp;
    lo[1] = gp; lo[2] = sreal * invB; lo[3] = sfake * invB;
  }
  const AdamCoef co = adam_coef(a.lr, a.b1, a.b2, a.eps, 0.f, 0, 0, step);
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + (FT / 64) * i;
    if (t < ntiles) {
      int li, n0, k0;
      tile_desc(t, li, n0, k0);
      const int N = li == nh ? 1 : L, K = li == 0 ? in_dim : L;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * q + r, k = k0 + j;
        if (n < N && k < K) {
          const int64_t o = cl.w[li] + (int64_t)n * K + k;
          float p = Pg[o], m = Mg[o], v = Vg[o];
          adam_update(p, m, v, acc_rf[i][r] + coef * acc_gp[i][r], co);
          Pg[o] = p; Mg[o] = m; Vg[o] = v;
        }
      }
    }
  }
  for (int idx = threadIdx.x; idx < (nh + 1) * L; idx += FT) {
    const int li = idx / L, c = idx - li * L;
    if (li < nh || c == 0) {
      const int64_t o = cl.b[li] + c;
      float p = Pg[o], m = Mg[o], v = Vg[o];
      adam_update(p, m, v, dbias[li * LQ + c], co);
      Pg[o] = p; Mg[o] = m; Vg[o] = v;
    }
  }
  STAMP(41);
}

__global__ __launch_bounds__(FT) void critic_fused_pair_kernel(IterArgs ax, IterArgs az, PhaseArgs ph) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.z == 0) critic_fused_body<true>(ax, ph, smem); else critic_fused_body<false>(az, ph, smem);
}

__global__ void advance_counters_kernel(int32_t* counters, int n) {
  if (threadIdx.x == 0) { counters[0] += n; counters[1] += n; counters[3] += n; }
}

}  // namespace

namespace hypad {
namespace train {

size_t critic_phase_floats_per_iter(const hypad_dims& d) {
  return (size_t)d.n_signals * d.batch * (pad4(d.signal_shape) + pad4(d.latent_dim));
}

// Runs the critic phase of an epoch (train.py:315-328) for n_iters = n_critics * n_batches (critic_x || critic_z)
// iterations.  ax / az: arguments of the two iterations (row_index and losses are set per iteration here).  extra:
// `extra_floats` floats of scratch; the phase is cut into chunks of as many iterations as fit.  losses: iteration `it`
// writes rows 2*it (critic_x) and 2*it+1 (critic_z) of each signal's loss table.  ev (optional, 3 events): recorded
// before the precompute, before the first fused launch and after it (profiling; only with n_iters == 1).
int run_critic_phase(IterArgs ax, IterArgs az, const int32_t* row_index, int n_iters, float* losses, float* extra, size_t extra_floats,
                     int n_signals, hipStream_t s, hipEvent_t* ev) {
  hypad_dims d; d.signal_shape = ax.S; d.latent_dim = ax.L; d.batch = ax.B; d.hyperbolic = ax.hyperbolic; d.n_signals = n_signals;
  const size_t per_iter = critic_phase_floats_per_iter(d);
  int chunk = (int)(extra_floats / per_iter);
  if (chunk < 1) return HYPAD_EWORKSPACE;
  if (chunk > n_iters) chunk = n_iters;
  // the generator-side randomness (z, decoder dropout) keeps the critic_x seed; critic_z draws from its own
  az.seed = ax.seed ^ 0x5851F42D4C957F2DULL;
  const size_t lds_pre = (size_t)lds_plan(ax.S, 16, 16, 0).total * sizeof(float);
  if (lds_pre > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)critic_phase_precompute_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pre);
    if (e != hipSuccess) return (int)e;
  }
  const FusedLds fx = fused_lds(ax.S, ax.L, 4);
  const FusedLds fz = fused_lds(az.L, az.L, 2);
  const size_t lds = (size_t)(fx.total > fz.total ? fx.total : fz.total) * sizeof(float);
  if (lds > 160 * 1024) return HYPAD_EUNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)critic_fused_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  for (int it0 = 0; it0 < n_iters; it0 += chunk) {
    const int n = n_iters - it0 < chunk ? n_iters - it0 : chunk;
    PhaseArgs ph;
    ph.n_iters = n;
    ph.row_index = row_index + (int64_t)it0 * ax.B;
    ph.gen_pre = extra;
    ph.zenc_pre = extra + (size_t)n_signals * n * ax.B * pad4(ax.S);
    ph.it = 0;
    ph.stamps = g_stamps;
    if (ev) (void)hipEventRecord(ev[0], s);
    hipLaunchKernelGGL(critic_phase_precompute_kernel, dim3(ax.B / 16, n_signals, 2 * n), dim3(TB), lds_pre, s, ax, ph);
    HYPAD_CHECK_LAUNCH();
    if (ev) (void)hipEventRecord(ev[1], s);
    for (int it = 0; it < n; ++it) {
      ph.it = it;
      ax.losses = losses + (int64_t)(2 * (it0 + it)) * 4;
      az.losses = losses + (int64_t)(2 * (it0 + it) + 1) * 4;
      hipLaunchKernelGGL(critic_fused_pair_kernel, dim3(1, n_signals, 2), dim3(FT), lds, s, ax, az, ph);
      HYPAD_CHECK_LAUNCH();
      if (ev && it == 0) (void)hipEventRecord(ev[2], s);
    }
    hipLaunchKernelGGL(advance_counters_kernel, dim3(1), dim3(64), 0, s, ax.counters, n);
    HYPAD_CHECK_LAUNCH();
  }
  return HYPAD_OK;
}

}  // namespace train
}  // namespace hypad

// development aid (not declared in hypad.h): device buffer of 128 int64 that the next fused launches stamp, or null
extern "C" void hypad_diag_set_fused_stamps(long long* p) { g_stamps = p; }
