// The three WGAN-GP minibatch steps of train.py (critic_x_iteration :18-104, critic_z_iteration :107-186,
// decoder_iteration :189-249) with the optimizer step fused in.
//
// Launch structure (every kernel: one model per blockIdx.y):
//   hypad_critic_{x,z}_iteration (per-iteration entry points; also hypad_train_epoch's fallback for shapes that
//   critic_fused.hip cannot take):
//                          pass kernel  (row tiles: frozen generator forward, 3 critic passes, first GP backward, partial norms)
//                          gp kernel    (row tiles: whole-batch norm -> second-order chain)
//                          dw_adam      (16x16 weight tiles: dW = left^T right on MFMA, Adam in registers)
//   hypad_decoder_iteration:
//                          pack kernel  (MFMA-native copies of the generator weights into the workspace; inside an epoch the
//                                        dW kernel keeps them current instead)
//                          gen kernel   (one workgroup per (row tile, chain): G = z -> decoder -> critic_x and back,
//                                        R = x -> encoder -> critic_z, decoder, reconstruction loss and back; every
//                                        (delta, activation) pair written for dw_adam)
//                          dw_adam      (Adam or Riemannian Adam; writes arena + packed copies)
//   hypad_train_epoch:     pack once -> critic phase (critic_fused.hip: precompute + one launch per critic_x || critic_z
//                          iteration) -> n_batches x (gen, dw_adam)
//   hypad_score_forward_packed: pack + the test-loop body on the same building blocks.
// No gradient buffer exists: a weight's gradient tile lives in MFMA accumulators and is consumed by the update.
// Formulas: oracle/manual.py (CPU derivation sheet, validated against autograd and the reference fixtures).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include <vector>

#include "../../include/hypad.h"
#include "critic_mfma.h"
#include "train_common.h"

using namespace hypad;
using namespace hypad::train;

namespace {

// ------------------------------------------------------------------------------------------------ critic passes
// The three passes of a WGAN-GP critic update on one 16-row tile -- real (dout -1/B), fake (+1/B), interpolated
// (dout 1: the first backward of the gradient penalty, train.py:72-81) -- carried together as 48 rows on the VALU path
// (critic_valu.h).  Writes every (left, right) pair the dW kernel needs, the interpolated pass's masks for the
// gradient-penalty kernel, g = d(prob)/d(interpolated) and the tile's partial sums.
// real / fake: LDS tiles (in_dim columns).  inter: LDS [16][ldi] scratch; gbuf: LDS [16][ldgb] for g (may alias `real`:
// it is written after the last read of `real`).  P: the critic's weights (staged in LDS when they fit).
__device__ __forceinline__ void critic_three_passes(const IterArgs& a, int sig, int tile, uint32_t tick, const float* real, int ldr,
                                                    const float* fake, int ldf, float* inter, int ldi, float* gbuf, int ldgb,
                                                    const float* P, const CriticLayout& cl, const CriticBatchLds& cb, float* ws,
                                                    const CritWs& cw, int mask_real, int mask_fake, int mask_inter, float* red) {
  const int L = a.L, B = a.B, nh = cl.nh, in_dim = cl.in_dim, LQ = cb.LQ;
  const int g0 = tile * 16;
  const float* mbase = a.masks ? a.masks + sig * a.mask_sig_stride : nullptr;
  const int64_t mblk = (int64_t)nh * B * L;
  const int midx[3] = {mask_real, mask_fake, mask_inter};
  DropSrc ds[3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
    ds[p] = drop_src(a, sig, mbase ? mbase + midx[p] * mblk : nullptr, RS_DROP_CRITIC + 8 * midx[p], tick, cl.p_drop);
  // interpolation (train.py:64-69 / 149-154)
  const float* ainj = a.alpha ? a.alpha + ((int64_t)sig * B + g0) * in_dim : nullptr;
  tile_for(16, in_dim, [&](int r, int c) {
    float al = ainj ? ainj[r * in_dim + c] : rng_uniform(a.seed, tick, RS_ALPHA, (uint32_t)(sig + a.rng_sig0), (uint32_t)((g0 + r) * in_dim + c));
    inter[r * ldi + c] = al * real[r * ldr + c] + (1.f - al) * fake[r * ldf + c];
  });
  __syncthreads();
  critic_batch_fwd([&](int r) { return r < 16 ? real + r * ldr : r < 32 ? fake + (r - 16) * ldf : inter + (r - 32) * ldi; },
                   P, cl, L, cb,
                   [&](int li, int r, int c) {
                     const int p = r >> 4;
                     return p == 0 ? ds[0].get(li, g0 + (r & 15), c, L) : p == 1 ? ds[1].get(li, g0 + (r & 15), c, L)
                                                                                   : ds[2].get(li, g0 + (r & 15), c, L);
                   });
  // what dW needs from the forward: layer inputs of the real / fake passes, masks of the interpolated pass
  tile_store(ws + cw.in_right + (int64_t)g0 * in_dim, in_dim, real, ldr, 16, in_dim, 16);
  tile_store(ws + cw.in_right + ((int64_t)B + g0) * in_dim, in_dim, fake, ldf, 16, in_dim, 16);
  for (int li = 0; li < nh; ++li) {
    tile_store_p(ws + cw.act[li] + (int64_t)g0 * L, L, B, cb.act + li * cb.R * LQ, LQ, 32, L, 32);
    tile_store(ws + cw.dm[li] + (int64_t)g0 * L, L, cb.dm + (li * cb.R + 32) * LQ, LQ, 16, L, 16);
  }
  float* sums = red + 32;
  if (threadIdx.x < 2) {
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += cb.out[threadIdx.x * 16 + r];
    sums[threadIdx.x] = s;
  }
  if (threadIdx.x < 48) {
    const int p = threadIdx.x >> 4;
    ws[cw.left[nh] + (int64_t)p * B + g0 + (threadIdx.x & 15)] = p == 0 ? -1.f / B : p == 1 ? 1.f / B : 1.f;
  }
  const float invB = 1.f / B;
  const float* d0 = critic_batch_bwd([&](int r) { return r < 16 ? -invB : r < 32 ? invB : 1.f; }, P, cl, L, cb,
                                     [&](int li, const float* delta) { tile_store_p(ws + cw.left[li] + (int64_t)g0 * L, L, B, delta, LQ, 48, L, 48); });
  // g = delta_0 W_0 for the interpolated rows (train.py:75-81)
  dense_rows_valu_t(d0 + 32 * LQ, LQ, L, P + cl.wof(0), in_dim, 16, [&](int r, int c, float v) { gbuf[r * ldgb + c] = v; });
  __syncthreads();
  tile_store(ws + cw.in_right + ((int64_t)2 * B + g0) * in_dim, in_dim, gbuf, ldgb, 16, in_dim, 16);
  float sq = 0.f;
  tile_for(16, in_dim, [&](int r, int c) { float v = gbuf[r * ldgb + c]; sq += v * v; });
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    float* part = ws + cw.partial + tile * 4;
    part[0] = sq; part[1] = sums[0]; part[2] = sums[1]; part[3] = 0.f;
  }
}

__device__ __forceinline__ void load_z(const IterArgs& a, int sig, int tile, uint32_t tick, float* zs /* [16][LP] */) {
  const float* zinj = a.z ? a.z + ((int64_t)sig * a.B + tile * 16) * a.L : nullptr;
  const int L = a.L;
  if (zinj || (L & 1)) {
    tile_for(16, L, [&](int r, int c) {
      zs[r * LP + c] = zinj ? zinj[r * L + c] : rng_normal(a.seed, tick, RS_Z, (uint32_t)(sig + a.rng_sig0), (uint32_t)((tile * 16 + r) * L + c));
    });
  } else {                                  // two normals per Philox evaluation: the same numbers as rng_normal per element
    for (int i = threadIdx.x; i < 16 * (L >> 1); i += blockDim.x) {
      const int r = i / (L >> 1), c = 2 * (i - r * (L >> 1));
      const uint32_t idx = (uint32_t)((tile * 16 + r) * L + c);
      Philox ph(a.seed);
      const uint4 rr = ph(idx >> 1, RS_Z, tick, (uint32_t)(sig + a.rng_sig0));
      zs[r * LP + c] = sqrtf(-2.0f * __logf(u32_to_unit_open(rr.x))) * __cosf(6.28318530717958647692f * u32_to_unit(rr.y));
      zs[r * LP + c + 1] = sqrtf(-2.0f * __logf(u32_to_unit_open(rr.z))) * __cosf(6.28318530717958647692f * u32_to_unit(rr.w));
    }
  }
}

// ---- critic_x pass (train.py:18-81)
__device__ __forceinline__ void cx_pass_body(const IterArgs& a, float* smem) {
  const int sig = blockIdx.y, tile = blockIdx.x, S = a.S, L = a.L, B = a.B;
  const CriticLayout cl = cx_layout(S, L);
  const LdsPlan lp = lds_plan(S, 16, 16, cl.total, critic_batch_lds_floats(48, L));
  const DecLayout dl = dec_layout(S, L, a.hyperbolic);
  const CritWs cw = crit_ws(B, S, L, 4);
  const float* PD = a.P.dec + (int64_t)sig * a.pd;
  const float* PC = a.P.cx + (int64_t)sig * a.pcx;
  float* ws = a.ws + sig * a.ws_sig_stride;
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  float* red = smem + lp.small + 3 * 16 * LP;
  const CriticBatchLds cs = critic_batch_lds(smem + lp.crit, 48, L);
  const uint32_t tick = (uint32_t)a.counters[3];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.counters[a.opt] += 1;
  if (lp.stage) { stage_params(smem + lp.cparams, PC, cl.total); PC = smem + lp.cparams; }

  tile_load_rows(xs, lp.ldS, a.x + sig * a.x_sig_stride, a.x_ld, a.row_index ? a.row_index + (int64_t)sig * a.ri_sig_stride : nullptr, tile * 16, 16, S, 16);
  load_z(a, sig, tile, tick, zs);
  __syncthreads();
  // decoder (frozen, train-mode dropout): x_ = decoder(z)
  const float* mbase = a.masks ? a.masks + sig * a.mask_sig_stride : nullptr;
  DropSrc ddrop = drop_src(a, sig, mbase ? mbase + (int64_t)12 * B * L : nullptr, RS_DROP_DEC0, tick, 0.2f);
  DecSave none{16, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const int grow0 = tile * 16;
  float* wst = smem + lp.wst;
  decoder_trunk_fwd_tile<1>(zs, L, S, PD, dl, bufA, bufB, lp.ldS, ddrop, [grow0](int r) { return grow0 + r; }, none, 16, wst);
  float* gen = bufA; float* other = bufB;
  if (a.hyperbolic) {
    gemm_nt<1>(bufA, lp.ldS, PD + dl.head_w, S, S, S, identity_map(), nullptr, nullptr, bufB, lp.ldS, 0, wst);
    __syncthreads();
    head_rows_tile(bufB, lp.ldS, 16, S, PD + dl.head_b);
    __syncthreads();
    gen = bufB; other = bufA;
  }
  // `other` receives the interpolation; g reuses the x tile (dead once the interpolation exists)
  critic_three_passes(a, sig, tile, tick, xs, lp.ldS, gen, lp.ldS, other, lp.ldS, xs, lp.ldS, PC, cl, cs, ws, cw, 0, 1, 2, red);
}

// ---- critic_z pass (train.py:107-166)
__device__ __forceinline__ void cz_pass_body(const IterArgs& a, float* smem) {
  const int sig = blockIdx.y, tile = blockIdx.x, S = a.S, L = a.L, B = a.B;
  const CriticLayout cl = cz_layout(L);
  const LdsPlan lp = lds_plan(S, 16, 16, cl.total, critic_batch_lds_floats(48, L));
  const EncLayout el = enc_layout(S, L);
  const CritWs cw = crit_ws(B, L, L, 2);
  const float* PE = a.P.enc + (int64_t)sig * a.pe;
  const float* PC = a.P.cz + (int64_t)sig * a.pcz;
  float* ws = a.ws + sig * a.ws_sig_stride;
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  float* small = smem + lp.small;
  float* red = small + 3 * 16 * LP;
  const CriticBatchLds cs = critic_batch_lds(smem + lp.crit, 48, L);
  const uint32_t tick = (uint32_t)a.counters[3];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.counters[a.opt] += 1;
  if (lp.stage) { stage_params(smem + lp.cparams, PC, cl.total); PC = smem + lp.cparams; }

  tile_load_rows(xs, lp.ldS, a.x + sig * a.x_sig_stride, a.x_ld, a.row_index ? a.row_index + (int64_t)sig * a.ri_sig_stride : nullptr, tile * 16, 16, S, 16);
  load_z(a, sig, tile, tick, zs);                       // real = z ~ N(0,1)
  __syncthreads();
  float* zenc = zs + 16 * LP;                           // fake = encoder(x)
  encoder_fwd_tile(xs, lp.ldS, S, L, PE, el, bufA, ENC_LDG, bufB, ENC_LDH, zenc, nullptr, nullptr, 16, smem + lp.wst);
  // injected mask order (hypad.h): fake | valid | interpolated
  critic_three_passes(a, sig, tile, tick, zs, LP, zenc, LP, small, LP, small + 16 * LP, LP, PC, cl, cs, ws, cw, 1, 0, 2, red);
}

// ---- gradient-penalty body: whole-batch norm and the second-order chain (oracle/manual.py
// critic_gp_pairs): GP rows of every `right` matrix.
template <bool IS_X>
__device__ __forceinline__ void gp_body(const IterArgs& a, float* smem) {
  const int sig = blockIdx.y, tile = blockIdx.x, L = a.L, B = a.B;
  const int in_dim = IS_X ? a.S : a.L;
  const int nh = IS_X ? 4 : 2;
  const CriticLayout cl = IS_X ? cx_layout(a.S, L) : cz_layout(L);
  const CritWs cw = crit_ws(B, in_dim, L, nh);
  const float* PC = IS_X ? a.P.cx + (int64_t)sig * a.pcx : a.P.cz + (int64_t)sig * a.pcz;
  float* ws = a.ws + sig * a.ws_sig_stride;
  const int ldu = pad4(in_dim) + 4;
  float* us = smem;                         // [16][ldu]
  float* e0 = us + 16 * ldu;                // [16][LP]
  float* e1 = e0 + 16 * LP;                 // [16][LP]
  float* pc = e1 + 16 * LP;                 // staged critic weights
  stage_params(pc, PC, cl.total);
  PC = pc;
  const int ntiles = B / 16;
  float gsum = 0.f, sreal = 0.f, sfake = 0.f;
  for (int t = 0; t < ntiles; ++t) {        // fixed order: deterministic
    const float* part = ws + cw.partial + t * 4;
    gsum += part[0]; sreal += part[1]; sfake += part[2];
  }
  const float nrm = sqrtf(gsum + 1e-12f);                       // train.py:90
  const float gp = (nrm - 1.f) * (nrm - 1.f);
  const float coef = 10.f * 2.f * (nrm - 1.f) / nrm;            // d(10 gp)/d g = coef * g
  if (tile == 0 && threadIdx.x == 0) {
    float* lo = a.losses + sig * a.loss_sig_stride;
    lo[0] = sfake / B - sreal / B + 10.f * gp;                  // train.py:98-99
    lo[1] = gp; lo[2] = sreal / B; lo[3] = sfake / B;
  }
  float* grow = ws + cw.in_right + ((int64_t)2 * B + tile * 16) * in_dim;
  tile_for(16, in_dim, [&](int r, int c) {
    const float u = coef * grow[r * in_dim + c];
    grow[r * in_dim + c] = u;                // right of layer 0 (GP rows) = ugrad
    us[r * ldu + c] = u;
  });
  __syncthreads();
  dense_rows_valu(us, ldu, in_dim, PC + cl.wof(0), nullptr, 16, L, [&](int r, int c, float v) { e0[r * LP + c] = v; });
  __syncthreads();
  float* cur = e0; float* nxt = e1;
  for (int li = 1; li <= nh; ++li) {
    // ep_{li-1} = e_{li-1} * dm_{li-1}: right of layer li (GP rows)
    const float* dm = ws + cw.dm[li - 1] + (int64_t)tile * 16 * L;
    float* dst = ws + cw.act[li - 1] + ((int64_t)2 * B + tile * 16) * L;
    tile_for(16, L, [&](int r, int c) {
      const float v = cur[r * LP + c] * dm[r * L + c];
      cur[r * LP + c] = v;
      dst[r * L + c] = v;
    });
    __syncthreads();
    if (li < nh) {
      dense_rows_valu(cur, LP, L, PC + cl.wof(li), nullptr, 16, L, [&](int r, int c, float v) { nxt[r * LP + c] = v; });
      __syncthreads();
      float* t = cur; cur = nxt; nxt = t;
    }
  }
}
HD int gp_lds_floats(int in_dim, int critic_floats) { return 16 * (pad4(in_dim) + 4) + 2 * 16 * LP + critic_floats; }

// ------------------------------------------------------------------------------------------------ generator body
// decoder_iteration (train.py:189-249).  Its three chains are independent until the weight gradients are summed:
//   role 0 (G):  z ~ N(0,1) -> decoder -> critic_x -> -mean(fake_x)                      and back through the decoder;
//   role 1 (R):  x -> encoder -> decoder -> reconstruction loss against x (hyperbolic: Poincare distance between the
//                Moebius heads of x_rec and of x) and back through decoder and encoder;
//   role 2 (Z):  x -> encoder -> critic_z -> -mean(fake_z) and back through the encoder (gen_role_z).  The encoder's
//                backward is linear in the gradient of its output, so the two contributions (through the decoder, chain R;
//                through critic_z, chain Z) are propagated separately and meet in the weight-gradient reduction, which runs
//                over both chains' operand rows (train_common.h GenWs).  Chain Z repeats the encoder forward of its 16 rows
//                -- 13 k cycles on its own CU -- and takes critic_z (7 k cycles, five barrier-separated 20-wide stages) off
//                chain R, the longest of the three.
// One workgroup per (16-row tile, role): blockIdx.z = role.  Operand rows in the workspace are pass-major as before:
// pass 0 = decoder(z), pass 1 = decoder(encoder(x)), pass 2 (hyperbolic only) = hyperbolic_linear(x).
struct GenLds {
  int hb, xs, zs, bufA, bufB, small, cw, ct, total, ldS;
};
HD GenLds gen_lds(int S, int L, int hyper, int role) {
  GenLds p;
  p.ldS = lds_stride(S);
  const CriticPad cp = role == 0 ? critic_pad(S, L, 4) : critic_pad(L, L, 2);      // (chain R stages no critic; same plan as Z)
  const int rows_head = (role == 1 && hyper) ? 32 : 16;
  const int a = 16 * (6 * DEC_H + 4), b = rows_head * p.ldS;
  const int bufFloats = a > b ? a : b;
  int o = 0;
  p.hb = o; o += (role == 1 && hyper) ? 16 * p.ldS : 0;   // chain R, hyperbolic: the head's 32 input rows = [tanh output | real windows],
  p.xs = o; o += role >= 1 ? 16 * p.ldS : 0;              // contiguous: the windows are gathered straight into the second half
  p.zs = o; o += 32 * LP;
  p.bufA = o; o += bufFloats;
  p.bufB = o; o += bufFloats;
  p.small = o; o += 3 * 16 * LP + 64;             // dzc | dzs | spare, then 64 reduction slots
  p.cw = o; o += cp.total;                        // the role's frozen critic, padded image (critic_mfma.h)
  p.ct = o; o += critic_tile_floats(cp, role == 0 ? 4 : 2);
  p.total = o;
  return p;
}

// Chain Z of the generator step (see above): encoder forward of the tile's windows, critic_z forward / backward, encoder
// backward of that gradient; operand rows into the second halves of xg / enc_h / dzenc / dgenc.
template <int SC, int LC, int BC, bool WSC1 = false>
__device__ __forceinline__ void gen_role_z(const IterArgs& a, float* smem) {
  const int S = SC ? SC : a.S, L = LC ? LC : a.L, B = BC ? BC : a.B;
  const int sig = blockIdx.y + a.sig0, tile = blockIdx.x >> 3;
  const GenLds lp = gen_lds(S, L, a.hyperbolic, 2);
  const int ldS = lp.ldS;
  const GenWs gw = gen_ws(B, S, L);
  float* ws = a.ws + sig * a.ws_sig_stride;
  const float* pk = ws + a.pk_off;
  const GenPack gp = gen_pack(S, L, a.hyperbolic);
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  float* dzc = smem + lp.small;             // [16][LP] gradient of -mean(critic_z) w.r.t. the encoder output
  float* cw = smem + lp.cw; float* ct = smem + lp.ct;
  const uint32_t tick = (uint32_t)a.counters[3] + (uint32_t)(a.step_add > 0 ? a.step_add : 0);
  const int g0 = tile * 16;
  const float* mbase = a.masks ? a.masks + sig * a.mask_sig_stride : nullptr;
  const CriticLayout clz = cz_layout(L);
  const CriticPad cpz = critic_pad(L, L, 2);
  // (the gather first: loads return in order, and the weight prefetch -- cold lines, rewritten by the previous launch -- would
  // hold its two dependent round trips back)
  tile_load_rows(xs, ldS, a.x + sig * a.x_sig_stride, a.x_ld, a.row_index ? a.row_index + (int64_t)sig * a.ri_sig_stride : nullptr, g0, 16, S, 16);
  const LstmPre pre_enc = lstm_layer_prefetch<WSC1>(pk + gp.enc_g[0], pk + gp.enc_g[1], ENC_H, S);
  stage_critic_padded(cw, a.P.cz + (int64_t)sig * a.pcz, clz, L, cpz);
  __syncthreads();
  tile_store(ws + gw.xg + (int64_t)(B + g0) * S, S, xs, ldS, 16, S, 16);
  float* zin = zs + 16 * LP;
  float* gates = ws + gw.enc_g2 + (int64_t)g0 * 8 * ENC_H;
  encoder_fwd_tile_packed<true, WSC1>(xs, ldS, S, L, pk, gp, bufA, ENC_LDG, bufB, ENC_LDH, zin, gates,
                                ws + gw.enc_h + (int64_t)(B + g0) * 2 * ENC_H, 16, pre_enc);
  const PackedPre pre_edt = gemm_nt_prefetch<WSC1>(pk + gp.enc_d_t, L, 2 * ENC_H);
  // critic_z(encoder(x)) and its input gradient (frozen critic; -mean(fake_z), train.py:215-217)
  const DropSrc dz = drop_src(a, sig, mbase, RS_DROP_CRITIC + 8 * 0, tick, clz.p_drop);
  const float sum_crit = critic_tile_fwd_bwd(zin, LP, cw, clz, L, cpz, ct, -1.f / B, [&](int li, int r, int c) { return dz.get4(li, g0 + r, c, L); },
                                             [&](int li, int r, int c) { return dz.get(li, g0 + r, c, L); }, dzc, LP);
  tile_store(ws + gw.dzenc + (int64_t)(B + g0) * L, L, dzc, LP, 16, L, 16);
  // encoder backward of that gradient
  float* dP = bufA;
  {
    LstmCellBwdEpi epi{gates, ENC_H, dP, ENC_LDG, 16, 16, nullptr, 0, ws + gw.dgenc + (int64_t)(B + g0) * 6 * ENC_H, {}, {}, {}, {}, {}};
    gemm_nt_packed_epi<1, true, decltype(epi), WSC1>(dzc, LP, L, 2 * ENC_H, pk + gp.enc_d_t, nullptr, 0, pre_edt, epi);      // dH, cell backward in the epilogue
  }
  float* part_out = ws + gw.partial + tile * 4;
  if (threadIdx.x == 0) part_out[2] = sum_crit;
  if (tile == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    // optimizer step number and its bias corrections (double-precision powers): once per launch, for the dW + Adam launch --
    // here, on the chain with slack; they go to the workspace of the launch's first model (IterArgs.sig0)
    const int step = a.counters[a.opt] + 1 + (a.step_add > 0 ? a.step_add : 0);
    if (a.step_add < 0) a.counters[a.opt] = step;
    const AdamCoef c0 = adam_coef(a.lr, a.b1, a.b2, a.eps, a.wd, a.riemannian, a.stabilize, step);
    float* ac = a.ws + (int64_t)a.sig0 * a.ws_sig_stride + gw.adamc;
    ac[0] = c0.bc1; ac[1] = c0.bc2; ac[2] = c0.sqrt_bc2;
  }
}

// SC / LC / BC: window length, latent width, batch as compile-time constants (0 = from the arguments); see
// critic_fused.hip: every layer of the chain runs once per launch, so index arithmetic is never amortised.
static_assert(TB == 512, "gen_body deals rows over 8 waves");
template <bool HYPER, int SC, int LC, int BC, bool WSC1 = false>
__device__ __forceinline__ void gen_body(const IterArgs& a, float* smem) {
  const int S = SC ? SC : a.S, L = LC ? LC : a.L, B = BC ? BC : a.B;
  const int sig = blockIdx.y + a.sig0, tile = blockIdx.x >> 3, role = blockIdx.z;
  __builtin_amdgcn_s_setprio(2);            // (tile_gemm.h mfma_prio_*: the MFMA loops run below everything else)
  if (role == 2) { gen_role_z<SC, LC, BC, WSC1>(a, smem); return; }
  const GenLds lp = gen_lds(S, L, HYPER ? 1 : 0, role);
  const int ldS = lp.ldS;
  const EncLayout el = enc_layout(S, L);
  const DecLayout dl = dec_layout(S, L, HYPER ? 1 : 0);
  const GenWs gw = gen_ws(B, S, L);
  const float* PE = a.P.enc + (int64_t)sig * a.pe;
  const float* PD = a.P.dec + (int64_t)sig * a.pd;
  float* ws = a.ws + sig * a.ws_sig_stride;
  const float* pk = ws + a.pk_off;          // MFMA-native copies of the generator weights (pack_generator_kernel / dW kernel)
  const GenPack gp = gen_pack(S, L, HYPER ? 1 : 0);
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  float* small = smem + lp.small;
  float* red = small + 3 * 16 * LP;
  float* cw = smem + lp.cw; float* ct = smem + lp.ct;
  float* Ein = (HYPER && role == 1) ? smem + lp.hb : bufA;      // the Moebius head's input rows
  const uint32_t tick = (uint32_t)a.counters[3] + (uint32_t)(a.step_add > 0 ? a.step_add : 0);
  // (the optimizer step number and its bias corrections -- two double-precision powers on one lane, ~2 k cycles -- are taken by
  // chain Z's first workgroup, which ends at 40 k of the kernel's 88 k cycles: gen_role_z)
  const int g0 = tile * 16;                 // first batch row of this tile
  const int pass = role;                    // decoder pass carried by this workgroup
  const int64_t prow0 = (int64_t)pass * B + g0;          // its first operand row in the pass-major workspace arrays
  const float* mbase = a.masks ? a.masks + sig * a.mask_sig_stride : nullptr;
  const int64_t BL = (int64_t)B * L;
  const int lane = threadIdx.x & 63, wave = wave_id(), nw = blockDim.x >> 6;
  float sum_crit = 0.f, sum_aux = 0.f;
  constexpr int ldH = 2 * DEC_H + 4, ldG = 6 * DEC_H + 4, ldA0 = 52;
  GEN_STAMP(0);
  // (Rounds 1-4 started with an "L2 warm-up": the workgroups of chains G and R together touched every 128-byte line of the packed
  // weights once, so that the layers would find them in this XCD's L2.  Since the weights come through prefetched sc1 buffer loads a
  // stage ahead it only stood in front of the chains' first loads -- loads return in order: the window gather and the first layer's
  // weights waited behind ~0.4 MB of cold lines.  Without it: 2.749 against 2.766 ms per epoch, 36.0 against 36.15 us per launch.)
  float* zin;                               // decoder input rows [16][LP]
  PackedPre pre_d1;                         // first weights of the decoder's first layer, requested a stage ahead
  if (role == 0) {
    pre_d1 = gemm_nt_prefetch<WSC1>(pk + gp.d1, L, DEC_D1);
    const CriticLayout clx = cx_layout(S, L);
    stage_critic_padded(cw, a.P.cx + (int64_t)sig * a.pcx, clx, L, critic_pad(S, L, 4));
    load_z(a, sig, tile, tick, zs);
    zin = zs;
    __syncthreads();
  } else {
    // ---- encoder(x)  (critic_z(encoder(x)) and its way back through the encoder: chain Z)
    // the window gather is two dependent memory round trips (row index, then the row)
    GEN_STAMP(14);
    tile_load_rows(xs, ldS, a.x + sig * a.x_sig_stride, a.x_ld, a.row_index ? a.row_index + (int64_t)sig * a.ri_sig_stride : nullptr, g0, 16, S, 16);
    const LstmPre pre_enc = lstm_layer_prefetch<WSC1>(pk + gp.enc_g[0], pk + gp.enc_g[1], ENC_H, S);      // behind the gather (loads return in order)
    GEN_STAMP(15);
    GEN_STAMP(12);
    __syncthreads();
    GEN_STAMP(13);
    tile_store(ws + gw.xg + (int64_t)g0 * S, S, xs, ldS, 16, S, 16);
    if (HYPER) tile_store(ws + gw.ecat + (int64_t)(2 * B + g0) * S, S, xs, ldS, 16, S, 16);      // the head's second row block (pass 2)
    zin = zs + 16 * LP;
    encoder_fwd_tile_packed<true, WSC1>(xs, ldS, S, L, pk, gp, bufA, ENC_LDG, bufB, ENC_LDH, zin,
                                  ws + gw.enc_g + (int64_t)g0 * 8 * ENC_H, ws + gw.enc_h + (int64_t)g0 * 2 * ENC_H, 16, pre_enc);
    GEN_STAMP(1);
    pre_d1 = gemm_nt_prefetch<WSC1>(pk + gp.d1, L, DEC_D1);
  }
  GEN_STAMP(2);
  // ---- decoder trunk on this role's pass
  tile_store(ws + gw.zcat + prow0 * L, L, zin, LP, 16, L, 16);
  DecSave sv;
  sv.ps = 16;
  sv.a0 = ws + gw.a0 + prow0 * DEC_D1;
  sv.g0 = ws + gw.g0 + prow0 * 8 * DEC_H;
  sv.h0d = ws + gw.h0d + prow0 * 2 * DEC_H;
  sv.mask = ws + gw.mask + prow0 * 2 * DEC_H;
  sv.g1 = ws + gw.g1 + prow0 * 8 * DEC_H;
  sv.h1 = ws + gw.h1 + prow0 * 2 * DEC_H;
  sv.e = ws + gw.ecat + prow0 * S;
  sv.stamps = (a.stamps && blockIdx.x == 0 && blockIdx.y == 0) ? a.stamps + (int64_t)blockIdx.z * 48 * 8 : nullptr;
  // injected layout: critic_z 2x(B,L) | critic_x 4x(B,L) | decoder(z) (B,128) | decoder(enc(x)) (B,128): as a
  // (layer, batch, 128) array with "layer" = pass, the two decoder masks are rows [0,B) and [B,2B) of one block.
  const DropSrc dd = drop_src(a, sig, mbase ? mbase + 6 * BL : nullptr, RS_DROP_DEC0, tick, 0.2f);
  const int growp = pass * B + g0;
  const PackedPre pre_head = decoder_trunk_fwd_tile_packed<1, true, WSC1>(zin, L, S, pk, gp, bufA, bufB, ldS, dd, [growp](int r) { return growp + r; }, sv, 16,
                                                                    pre_d1, HYPER ? pk + gp.head : nullptr, S, S, Ein);
  GEN_STAMP(3);
  // E = tanh output in Ein[0..15]
  const int hrows = (HYPER && role == 1) ? 32 : 16;       // rows through the Moebius head: role R adds pass 2 = the real window
  float u_keep[16], hb_keep[16];            // this wave's head rows: pre-activations and bias (lane's elements), forward -> backward
  if (HYPER) {
    GEN_STAMP(20);
    if (role == 1) gemm_nt_packed<2, true, ActIdentity, WSC1>(Ein, ldS, S, S, pk + gp.head, nullptr, bufB, ldS, 0, 0, pre_head);
    else gemm_nt_packed<1, true, ActIdentity, WSC1>(Ein, ldS, S, S, pk + gp.head, nullptr, bufB, ldS, 0, 0, pre_head);
    __syncthreads();
    GEN_STAMP(21);
    GEN_STAMP(22);
    // the Moebius head, row-wise in place (head_rows_tile's row dealing: four rows per wave, one per 16-lane DPP row).  The wave
    // that carries a row forward carries it backward as well: u (the head's pre-activations) and the bias stay in its registers
    // until then -- the backward used to re-read u from the workspace, a memory round trip in the middle of the chain.
    epl16_dispatch(S, [&](auto tag) {
      constexpr int EPL = decltype(tag)::value;
      using R16 = RowT<16, EPL>;
      const R16 hb = row_load<R16>(PD + dl.head_b, S, lane);
      const int r = wave * 4 + (lane >> 4);
      if (wave * 4 < hrows) {
        const R16 u = row_load<R16>(bufB + r * ldS, S, lane);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { u_keep[e] = u.v[e]; hb_keep[e] = hb.v[e]; }
        row_store(bufB + r * ldS, head_row(u, hb), S, lane);
      }
    });
    __syncthreads();
  }
  float* R = HYPER ? bufB : bufA;           // decoder outputs (rows 0-15: this pass [, rows 16-31: hyper_x])
  float* dR = HYPER ? bufA : bufB;          // their gradients
  GEN_STAMP(4);
  if (role == 0) {
    // ---- critic_x(x_gen) and its input gradient (loss term -mean(fake_x), train.py:205-207)
    const CriticLayout clx = cx_layout(S, L);
    const DropSrc dx = drop_src(a, sig, mbase ? mbase + 2 * BL : nullptr, RS_DROP_CRITIC + 8 * 1, tick, clx.p_drop);
    sum_crit = critic_tile_fwd_bwd(R, ldS, cw, clx, L, critic_pad(S, L, 4), ct, -1.f / B,
                                   [&](int li, int r, int c) { return dx.get4(li, g0 + r, c, L); },
                                   [&](int li, int r, int c) { return dx.get(li, g0 + r, c, L); }, dR, ldS);
  }
  GEN_STAMP(5);
  if (HYPER) {
    if (role == 1) {
      // ---- hyperbolic reconstruction loss 10 * sum(dist)/B (train.py:226-234) and its gradients
      float part = 0.f;
      if (wave < 4) {                                 // 16 row pairs: four per wave, one per 16-lane DPP row
        epl16_dispatch(S, [&](auto tag) {
          using R16 = RowT<16, decltype(tag)::value>;
          const int r = wave * 4 + (lane >> 4);
          R16 du, dv;
          const float d = rowdist_row_bwd(row_load<R16>(R + r * ldS, S, lane), row_load<R16>(R + (16 + r) * ldS, S, lane), 10.f / B, du, dv);
          row_store(dR + r * ldS, du, S, lane);
          row_store(dR + (16 + r) * ldS, dv, S, lane);
          const float d0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 0)), d1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 16));
          const float d2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 32)), d3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 48));
          part = (d0 + d1) + (d2 + d3);
        });
      }
      if (lane == 0) red[16 + wave] = part;
      __syncthreads();
      for (int w = 0; w < nw; ++w) sum_aux += red[16 + w];
    }
    GEN_STAMP(23);
    const PackedPre pre_ht = gemm_nt_prefetch<WSC1>(pk + gp.head_t, S, S);        // the backward products' weights, each a stage ahead
    // ---- Moebius head backward, row-wise: dR -> dU (in place); this wave's share of the bias gradient in registers
    // (four rows per wave, one per 16-lane DPP row; the row's bias gradient goes to its own row of R: the head outputs are dead)
    epl16_dispatch(S, [&](auto tag) {
      constexpr int EPL = decltype(tag)::value;
      using R16 = RowT<16, EPL>;
      const int r = wave * 4 + (lane >> 4);
      if (wave * 4 < hrows) {
        R16 du, db, u, hb;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { u.v[e] = u_keep[e]; hb.v[e] = hb_keep[e]; }
        head_row_bwd(u, hb, row_load<R16>(dR + r * ldS, S, lane), du, db);
        row_store(dR + r * ldS, du, S, lane);
        row_store(ws + gw.du + (prow0 + prow(r, B)) * S, du, S, lane);       // the dW kernel's operand rows, straight from here
        // the wave's four bias-gradient rows are summed across its four 16-lane groups before they leave the registers: the
        // column sums below then add one row per wave instead of four
#pragma unroll
        for (int e = 0; e < R16::EPL; ++e) {
          float v = db.v[e];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          db.v[e] = v;
        }
        if (lane < 16) row_store(R + wave * ldS, db, S, lane);                // row `wave` of R: this wave's partial
      }
    });
    __syncthreads();
    GEN_STAMP(24);
    for (int c = threadIdx.x; c < S; c += blockDim.x) {
      float s = 0.f;
      for (int w = 0; w < hrows / 4; ++w) s += R[w * ldS + c];
      ws[gw.ballpart + ((int64_t)role * (B / 16) + tile) * S + c] = s;
    }
    __syncthreads();
    GEN_STAMP(25);
    // dE = dU W_h for the decoder pass, and d(pre-tanh) = dE * (1 - E^2) in its epilogue (E re-read from the workspace,
    // requested before the reduction)
    {
      struct TanhBwdEpi {
        float* Ys; int ldy; const float* E; int ldE; float* gout; float e[4]; int vo;
        __device__ __forceinline__ void prefetch(int n, int q, bool ok) {
          vo = (4 * q * ldE + (ok ? n : 0)) * 4;                          // (buffer addressing: tile_gemm.h GBuf)
          const GBuf eb(E);
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = ok ? eb.ld(vo, r * ldE * 4) : 0.f;
        }
        __device__ __forceinline__ void emit(int, int r, int row, int n, float v) {
          const float o = v * (1.f - e[r] * e[r]);
          Ys[row * ldy + n] = o; GBuf(gout).st(o, vo, r * ldE * 4);      // LDS tile + the dW kernel's operand rows
        }
      } epi{R, ldS, ws + gw.ecat + prow0 * S, S, ws + gw.dpre2 + prow0 * S, {}, 0};
      gemm_nt_packed_epi<1, true, decltype(epi), WSC1>(dR, ldS, S, S, pk + gp.head_t, nullptr, 0, pre_ht, epi);
    }
    __syncthreads();
    GEN_STAMP(26);
  } else {
    if (role == 1) {
      // ---- 10 * MSE(x, x_rec) (train.py:241-242): E in bufA (= R), gradients into bufB (= dR)
      float part = 0.f;
      tile_for(16, S, [&](int r, int c) {
        const float diff = R[r * ldS + c] - xs[r * ldS + c];
        part += diff * diff;
        dR[r * ldS + c] = 20.f * diff / ((float)B * (float)S);
      });
      sum_aux = block_sum(part, red);
      __syncthreads();
    }
    tile_for(16, S, [&](int r, int c) {
      const float e = R[r * ldS + c];
      dR[r * ldS + c] *= 1.f - e * e;
    });
    __syncthreads();
  }
  // The backward half ping-pongs between the two big LDS tiles X (holds d(pre-tanh)) and Y; every LSTM cell backward runs in
  // the epilogue of the product that makes its dH (LstmCellBwdEpi), so a layer is one stage: product -> gate deltas.
  float* X = HYPER ? R : dR;                // d(pre-tanh) [16][ldS]
  float* Y = HYPER ? dR : R;
  GEN_STAMP(6);
  const PackedPre pre_d2t = gemm_nt_prefetch<WSC1>(pk + gp.d2_t, S, 2 * DEC_H);
  if (!HYPER) tile_store(ws + gw.dpre2 + prow0 * S, S, X, ldS, 16, S, 16);        // (hyperbolic: written by the epilogue above)
  const PackedPre pre_l1t = gemm_nt_prefetch<WSC1>(pk + gp.l_t[1], 6 * DEC_H, 2 * DEC_H);
  // dH1 = dpre W2, layer 1 cell backward -> dG1 in Y
  {
    LstmCellBwdEpi epi{ws + gw.g1 + prow0 * 8 * DEC_H, DEC_H, Y, ldG, 16, 16, nullptr, 0, ws + gw.dg1 + prow0 * 6 * DEC_H, {}, {}, {}, {}, {}};
    gemm_nt_packed_epi<1, true, decltype(epi), WSC1>(X, ldS, S, 2 * DEC_H, pk + gp.d2_t, nullptr, 0, pre_d2t, epi);
  }
  GEN_STAMP(32);
  __syncthreads();
  GEN_STAMP(7);
  GEN_STAMP(33);
  const PackedPre pre_l0t = gemm_nt_prefetch<WSC1>(pk + gp.l_t[0], 6 * DEC_H, DEC_D1);
  GEN_STAMP(34);
  // dH0d = dG1 W_ih(l1) (both directions: one stacked reduction), x the inter-layer dropout mask, layer 0 cell backward -> dG0 in X
  {
    LstmCellBwdEpi epi{ws + gw.g0 + prow0 * 8 * DEC_H, DEC_H, X, ldG, 16, 16,
                       a.drop_mode != 0 ? ws + gw.mask + prow0 * 2 * DEC_H : nullptr, 2 * DEC_H, ws + gw.dg0 + prow0 * 6 * DEC_H, {}, {}, {}, {}, {}};
    gemm_nt_packed_epi<1, true, decltype(epi), WSC1>(Y, ldG, 6 * DEC_H, 2 * DEC_H, pk + gp.l_t[1], nullptr, 0, pre_l1t, epi);
  }
  GEN_STAMP(35);
  __syncthreads();
  GEN_STAMP(36);
  GEN_STAMP(8);
  GEN_STAMP(37);
  PackedPre pre_d1t{};
  if (role == 1) pre_d1t = gemm_nt_prefetch<WSC1>(pk + gp.d1_t, DEC_D1, L);
  GEN_STAMP(38);
  gemm_nt_packed<1, true, ActIdentity, WSC1>(X, ldG, 6 * DEC_H, DEC_D1, pk + gp.l_t[0], nullptr, Y, ldA0, 0, 0, pre_l0t, ActIdentity{}, nullptr, 0,
                          ws + gw.da0 + prow0 * DEC_D1, DEC_D1);                  // (+ the dW kernel's operand rows)
  GEN_STAMP(39);
  __syncthreads();
  GEN_STAMP(9);
  float* part_out = ws + gw.partial + tile * 4;
  if (role == 0) {
    if (threadIdx.x == 0) part_out[1] = sum_crit;
    GEN_STAMP(11);
    return;
  }
  // dZ = dA0 W1: the gradient reaching the encoder's output
  const PackedPre pre_edt = gemm_nt_prefetch<WSC1>(pk + gp.enc_d_t, L, 2 * ENC_H);
  gemm_nt_packed<1, true, ActIdentity, WSC1>(Y, ldA0, DEC_D1, L, pk + gp.d1_t, nullptr, X, LP, 0, 0, pre_d1t, ActIdentity{}, nullptr, 0,
                          ws + gw.dzenc + (int64_t)g0 * L, L);                     // (the critic_z part of dZ: chain Z)
  __syncthreads();
  GEN_STAMP(10);
  // ---- encoder backward: dH = dZ W_dense, cell backward -> dG in Y
  {
    LstmCellBwdEpi epi{ws + gw.enc_g + (int64_t)g0 * 8 * ENC_H, ENC_H, Y, ENC_LDG, 16, 16, nullptr, 0,
                       ws + gw.dgenc + (int64_t)g0 * 6 * ENC_H, {}, {}, {}, {}, {}};
    gemm_nt_packed_epi<1, true, decltype(epi), WSC1>(X, LP, L, 2 * ENC_H, pk + gp.enc_d_t, nullptr, 0, pre_edt, epi);
  }
  if (threadIdx.x == 0) part_out[0] = sum_aux;
  GEN_STAMP(11);
}

// ---- kernels: single iterations, and the critic_x || critic_z pair (blockIdx.z picks the critic)
__global__ __launch_bounds__(TB) void cx_pass_kernel(IterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cx_pass_body(a, smem);
}
__global__ __launch_bounds__(TB) void cz_pass_kernel(IterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cz_pass_body(a, smem);
}
__global__ __launch_bounds__(TB) void critic_pass_pair_kernel(IterArgs ax, IterArgs az) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.z == 0) cx_pass_body(ax, smem); else cz_pass_body(az, smem);
}
template <bool IS_X>
__global__ __launch_bounds__(TB) void critic_gp_kernel(IterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gp_body<IS_X>(a, smem);
}
__global__ __launch_bounds__(TB) void critic_gp_pair_kernel(IterArgs ax, IterArgs az) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.z == 0) gp_body<true>(ax, smem); else gp_body<false>(az, smem);
}
// Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, dispatch): blockIdx.x is stretched by 8 and
// only the blocks that land on XCD (signal mod 8) work, so the 2 * B/16 workgroups of one model share an L2 -- each
// generator weight is fetched from HBM once per launch instead of once per workgroup.  Placement is a speed matter
// only: results do not depend on it.
// (Measured and dropped in round 3: four extra workgroups per XCD that read the packed weights once, in the chains' order, to leave
// them in that XCD's L2 ahead of the chains' sc1 loads -- the weights were rewritten a launch ago on other XCDs.  No change:
// 40.14 vs 40.17 us per launch.  The chains' weight fetches are already hidden behind the previous stage; what a chain's 88 k
// cycles consist of is MFMA issue plus the epilogues' vector instructions, which do not overlap on one SIMD: DESIGN.md §4.)
template <bool HYPER, int SC, int LC, int BC>
__global__ __launch_bounds__(TB) void gen_kernel(IterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (a.guard && a.counters[4] != 0) return;         // fail-stop behind a resident critic launch that gave up (hypad_epoch_status)
  // chain Z (blockIdx.z == 2) goes to the neighbouring XCD: it reads only the encoder's weights and must not queue behind
  // chains G and R for the 32 CUs of theirs (batch 256: 16 tiles x 2 chains fill an XCD)
  if ((blockIdx.x & 7) != ((blockIdx.y + a.sig0 + (blockIdx.z == 2 ? 1 : 0)) & 7)) return;
  gen_body<HYPER, SC, LC, BC, true>(a, smem);        // (weights through sc1 buffer loads: tile_gemm.h WeightBlocks)
}

// ------------------------------------------------------------------------------------------------ dW + Adam
enum DwKind : int { DW_WEIGHT = 0, DW_BIAS = 1, DW_DECAY = 2, DW_BALL = 3 };
struct DwDesc {
  int16_t kind, net;
  int32_t p_off, p_ld, nrows, ncols;          // destination block (weights: nrows x ncols at row stride p_ld)
  int32_t left_off, left_ld, right_off, right_ld;
  int32_t red_rows;
  int32_t p_off2;                             // second destination with the same gradient (b_hh), or -1
  int32_t begin;                              // first work item
};
template <int CAP, int BLK>
struct DwTableT {
  static constexpr int cap = CAP, blk = BLK;
  int n, total_items;                         // n < 0: the table did not fit (host check)
  int finalize;                               // 0 none, 1 generator losses
  // Work items are dealt four to a workgroup (one per wave) and every descriptor starts on a workgroup boundary, so a
  // workgroup finds its descriptor with one byte lookup instead of a search (byte b of word b / 4).
  uint32_t block_desc[BLK / 4];
  DwDesc d[CAP];
};
using DwTable = DwTableT<60, 512>;            // generator (S = MAX_S needs ~470 workgroups)
using DwTableS = DwTableT<12, 64>;            // one critic

// Where a generator weight block / bias vector lives in the packed copies (layout.h GenPack), found from its arena offset:
// the dW + Adam kernel keeps the copies current so that an epoch never re-packs.
struct ShadowRef {
  int fwd, fwd_kg;       // forward copy: blocks of (compact rows, K), or -1
  int nbase;             // compact row of the block's row 0 (LSTM: 0 for the i gate rows, H for the g|o rows)
  int bwd, bwd_ng;       // backward-data copy (W^T), or -1; ng = k-groups of its reduction
  int mbase;             // reduction index of compact row 0 in the backward copy (direction * 3H)
  int bsum;              // summed-bias vector, or -1 (bias items)
  int gate_H;            // LSTM gate matrices: hidden size (forward copy rows are x * up16(H) + u); 0 otherwise
};
__device__ __forceinline__ ShadowRef shadow_ref(int net, int p_off, int S, int L, int hyper) {
  ShadowRef r{-1, 0, 0, -1, 0, 0, -1, 0};
  const GenPack gp = gen_pack(S, L, hyper);
  if (net == HYPAD_NET_ENCODER) {
    const EncLayout el = enc_layout(S, L);
    for (int d = 0; d < 2; ++d) {
      const int w = el.dir[d].w_ih, wg = w + 2 * ENC_H * S;
      if (p_off == w || p_off == wg) { r.fwd = gp.enc_g[d]; r.fwd_kg = (S + 15) >> 4; r.nbase = p_off == w ? 0 : ENC_H; r.gate_H = ENC_H; }
      const int b = el.dir[d].b_ih;
      if (p_off == b || p_off == b + 2 * ENC_H) { r.bsum = gp.enc_gb[d]; r.nbase = p_off == b ? 0 : ENC_H; r.gate_H = ENC_H; }
    }
    if (p_off == el.dense_w) { r.fwd = gp.enc_d; r.fwd_kg = (2 * ENC_H + 15) >> 4; r.bwd = gp.enc_d_t; r.bwd_ng = (L + 15) >> 4; }
    if (p_off == el.dense_b) r.bsum = gp.enc_db;
  } else if (net == HYPAD_NET_DECODER) {
    const DecLayout dl = dec_layout(S, L, hyper);
    if (p_off == dl.d1_w) { r.fwd = gp.d1; r.fwd_kg = (L + 15) >> 4; r.bwd = gp.d1_t; r.bwd_ng = (DEC_D1 + 15) >> 4; }
    if (p_off == dl.d1_b) r.bsum = gp.d1b;
    for (int l = 0; l < 2; ++l) {
      const int in = l == 0 ? DEC_D1 : 2 * DEC_H;
      for (int d = 0; d < 2; ++d) {
        const int w = dl.l[l][d].w_ih, wg = w + 2 * DEC_H * in;
        if (p_off == w || p_off == wg) {
          r.fwd = gp.l_g[l][d]; r.fwd_kg = (in + 15) >> 4; r.nbase = p_off == w ? 0 : DEC_H; r.gate_H = DEC_H;
          r.bwd = gp.l_t[l]; r.bwd_ng = (6 * DEC_H) >> 4; r.mbase = d * 3 * DEC_H;
        }
        const int b = dl.l[l][d].b_ih;
        if (p_off == b || p_off == b + 2 * DEC_H) { r.bsum = gp.l_gb[l][d]; r.nbase = p_off == b ? 0 : DEC_H; r.gate_H = DEC_H; }
      }
    }
    if (p_off == dl.d2_w) { r.fwd = gp.d2; r.fwd_kg = (2 * DEC_H + 15) >> 4; r.bwd = gp.d2_t; r.bwd_ng = (S + 15) >> 4; }
    if (p_off == dl.d2_b) r.bsum = gp.d2b;
    if (hyper && p_off == dl.head_w) { r.fwd = gp.head; r.fwd_kg = (S + 15) >> 4; r.bwd = gp.head_t; r.bwd_ng = (S + 15) >> 4; }
  }
  return r;
}

// SC / LC / BC: compile-time dims of the reference configuration (0 = run-time), which fold the layout arithmetic of
// shadow_ref() and gen_ws() into constants.
// (Measured and dropped in round 3: 32 x 32 weight tiles -- four accumulators per wave, half the operand loads per MFMA and per
// parameter -- for the many-signal launches: 157 registers (2 waves per SIMD instead of 3) and three round trips per item made
// the launch SLOWER at 8 / 16 / 32 signals per GPU: 33 -> 42, 56 -> 64, 101 -> 110 us.  The operand re-reads of the 16 x 16 tiles
// are L2 hits once a model's tiles share an XCD (COLOC); what the launch waits for is its scattered 64-byte store segments.)
// What the launch's time consists of at 32 signals per GPU (round 3 what-if builds, 85.6 us): without the packed copies' stores
// 69 us, without the moments' stores 83, without the operand loads 62, without the optimiser-state loads 74 -- i.e. bytes, not
// instructions: writing the packed copies as 16-byte stores (transposed copy as is, forward copy after a DPP quad transpose)
// instead of eight scattered 4-byte stores per lane changed nothing (84.5 us; 11.5 at one signal) and was dropped.
// KS: k-steps (groups of four reduction rows) a weight item keeps in flight: 48 covers B <= 64 in one memory round trip.
// (16 halves the registers and doubles the waves per SIMD; measured with 8 and 32 signals per GPU it changes nothing --
// with many signals the launch moves ~9 MB per signal and sits at ~3.5 TB/s of HBM traffic.)
#ifndef HYPAD_DW_KS_MANY
#define HYPAD_DW_KS_MANY 16                  // k-steps in flight in the many-signal (co-located) form of the dW + Adam launch: 61 registers, seven waves
                                             // per SIMD instead of three (126 registers at 48).  Round 3, same box, launch time at 8 / 16 / 32 signals
                                             // per GPU: 48 -> 30.5 / 49.0 / 86.4 us, 32 -> 29.4 / 45.2 / 77.7, 24 -> 27.2 / 41.7 / 70.3, 16 -> 26.7 / 41.1 /
                                             // 68.1 (same accumulation order: same bits).  The one-signal launch keeps 48: it is as long as its slowest
                                             // item, and an item with all its operand rows in flight makes ONE memory round trip
#endif
template <class Table, int SC = 0, int LC = 0, int BC = 0, int KS = 48>
__device__ __forceinline__ void dw_adam_body(const IterArgs& a, const Table& tab, const int bx = (int)blockIdx.x) {      // bx: the workgroup's index in the launch's work (dw_adam_kernel: co-location)
  const int S_ = SC ? SC : a.S, L_ = LC ? LC : a.L, B_ = BC ? BC : a.B;
  const int sig = blockIdx.y + a.sig0;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int j = lane & 15, q = lane >> 4;
  float* ws = a.ws + sig * a.ws_sig_stride;
  const int item = bx * (THREADS / 64) + wave;          // the launch covers total_items: one item per wave
  // Every scalar fetch that hangs off the kernel arguments leaves in one batch -- the workgroup's descriptor, the step
  // counter, the generator's bias corrections -- instead of one memory round trip each.
  const int di = (tab.block_desc[bx >> 2] >> (8 * (bx & 3))) & 0xff;
  const DwDesc d = tab.d[di];                    // by value: one load group, not one load per field
  const int step = a.counters[a.opt] + (a.step_add >= 0 ? a.step_add + 1 : 0);      // (step_add < 0: already incremented by the iteration's first kernel)
  const float* ac = a.ws + (tab.finalize == 1 ? (int64_t)a.sig0 * a.ws_sig_stride + gen_ws(B_, S_, L_).adamc : 0);
  const float ac0 = ac[0], ac1 = ac[1], ac2 = ac[2];
  AdamCoef co;
  if (tab.finalize == 1) {                     // generator: the first kernel left the bias corrections in the workspace
    co.lr = a.lr; co.b1 = a.b1; co.b2 = a.b2; co.eps = a.eps; co.wd = a.wd; co.riemannian = a.riemannian; co.stabilize = a.stabilize;
    co.step = step; co.bc1 = ac0; co.bc2 = ac1; co.sqrt_bc2 = ac2;
  } else {
    co = adam_coef(a.lr, a.b1, a.b2, a.eps, a.wd, a.riemannian, a.stabilize, step);
  }
  const int64_t arena = d.net == HYPAD_NET_ENCODER ? a.pe : d.net == HYPAD_NET_DECODER ? a.pd : d.net == HYPAD_NET_CRITIC_X ? a.pcx : a.pcz;
  auto pick = [&](const hypad_nets& n) __attribute__((always_inline)) {
    float* base = d.net == HYPAD_NET_ENCODER ? n.enc : d.net == HYPAD_NET_DECODER ? n.dec : d.net == HYPAD_NET_CRITIC_X ? n.cx : n.cz;
    return base + sig * arena;
  };
  if (item < tab.total_items) {
    const int local = item - d.begin;            // may lie in the padding behind the descriptor's last item (checked per kind)
    float* P = pick(a.P); float* M = pick(a.M); float* V = pick(a.V);
    if (d.kind == DW_WEIGHT && local < ((d.nrows + 15) >> 4) * ((d.ncols + 15) >> 4)) {
      const int tk = (d.ncols + 15) >> 4;
      const int n0 = (local / tk) * 16, k0 = (local % tk) * 16;
      // out-of-range columns are clamped (their results are dropped below); rows past red_rows contribute zeros
      const int nj = n0 + j < d.nrows ? n0 + j : d.nrows - 1, kj = k0 + j < d.ncols ? k0 + j : d.ncols - 1;
      // Operands through buffer loads: one per-lane byte offset (row q, column nj / kj) plus a scalar row offset per load,
      // so the ~100 loads of an item cost no vector address arithmetic (with flat addressing that arithmetic took longer
      // than the memory round trip).  Row offsets are clamped on the scalar side; masked rows contribute zeros below.
      const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws + d.left_off), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws + d.right_off), 0, 0x7fffffff, 0x00020000);
      // red_rows is a multiple of 16 (whole batches of B = 16 m rows): the reduction runs in groups of four k-steps, each
      // entirely in range, so nothing is clamped or masked.
      const int lvo = (q * d.left_ld + nj) * 4, rvo = (q * d.right_ld + kj) * 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // the optimiser state of this lane's four elements travels with the operand loads (one round trip, not two)
      int po[4]; float pp[4], pm[4], pvv[4];
      float* Pb = P + d.p_off; float* Mb = M + d.p_off; float* Vb = V + d.p_off;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * q + r, k = k0 + j;
        const bool ok = n < d.nrows && k < d.ncols;
        po[r] = ok ? n * d.p_ld + k : -1;
        const uint32_t oc = ok ? (uint32_t)po[r] : 0u;
        pp[r] = Pb[oc]; pm[r] = Mb[oc]; pvv[r] = Vb[oc];
      }
      for (int rc = 0; rc < d.red_rows; rc += 4 * KS) {    // up to KS k-steps in flight
        float la[KS], rb[KS];
#pragma unroll
        for (int c = 0; c < KS / 4; ++c)
          if (rc + 16 * c < d.red_rows) {                  // wave-uniform
#pragma unroll
            for (int u = 4 * c; u < 4 * c + 4; ++u) {
              la[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, lvo, (rc + 4 * u) * d.left_ld * 4, 0));
              rb[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, rvo, (rc + 4 * u) * d.right_ld * 4, 0));
            }
          }
#pragma unroll
        for (int c = 0; c < KS / 4; ++c)
          if (rc + 16 * c < d.red_rows) {
#pragma unroll
            for (int u = 4 * c; u < 4 * c + 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc, 0, 0, 0);
          }
      }
      ShadowRef sh{-1, 0, 0, -1, 0, 0, -1, 0};
      if (tab.finalize == 1) sh = shadow_ref(d.net, d.p_off, S_, L_, a.hyperbolic);
      float* pk = ws + a.pk_off;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = pp[r], m = pm[r], v = pvv[r];
        adam_update(p, m, v, acc[r], co);
        if (po[r] >= 0) {
          const uint32_t o = (uint32_t)po[r];
          Pb[o] = p; Mb[o] = m; Vb[o] = v;
          const int nc = sh.nbase + n0 + 4 * q + r, k = k0 + j;            // compact row, column
          if (sh.fwd >= 0) {
            int nf = nc;                                                     // forward copy row: gates padded to 16-row blocks
            if (sh.gate_H > 0) nf += ((nc >= sh.gate_H) + (nc >= 2 * sh.gate_H)) * (((sh.gate_H + 15) & ~15) - sh.gate_H);   // nc < 3H
            pk[sh.fwd + (((nf >> 4) * sh.fwd_kg + (k >> 4)) * 64 + (nf & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3)] = p;
          }
          if (sh.bwd >= 0) {
            const int mm = sh.mbase + nc;                                     // reduction index of the transposed copy
            pk[sh.bwd + (((k >> 4) * sh.bwd_ng + (mm >> 4)) * 64 + (k & 15) + 16 * ((mm & 15) >> 2)) * 4 + (mm & 3)] = p;
          }
        }
      }
    } else if (d.kind == DW_BIAS) {
      // 16 columns per item; lane (j, q) sums rows r = q (mod 4), then the four row classes are folded by shuffles
      const int n = local * 16 + j;
      const bool nv = n < d.nrows;
      const float* left = ws + d.left_off + (nv ? n : d.nrows - 1);
      const int rlast = d.red_rows - 1;
      float g = 0.f;
      for (int rc = 0; rc < d.red_rows; rc += 4 * KS) {    // KS rows per lane in flight
        float t[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          const int r = rc + 4 * u + q;
          t[u] = left[(int64_t)(r < rlast ? r : rlast) * d.left_ld];
        }
#pragma unroll
        for (int u = 0; u < KS; ++u) g += rc + 4 * u + q < d.red_rows ? t[u] : 0.f;
      }
      g += __shfl_xor(g, 16, WAVE);
      g += __shfl_xor(g, 32, WAVE);
      if (nv && q == 0) {
        int64_t o = d.p_off + n;
        float p = P[o], m = M[o], v = V[o];
        adam_update(p, m, v, g, co);
        P[o] = p; M[o] = m; V[o] = v;
        float bs = p;
        if (d.p_off2 >= 0) {
          o = d.p_off2 + n;
          p = P[o]; m = M[o]; v = V[o];
          adam_update(p, m, v, g, co);
          P[o] = p; M[o] = m; V[o] = v;
          bs += p;
        }
        if (tab.finalize == 1) {
          const ShadowRef sh = shadow_ref(d.net, d.p_off, S_, L_, a.hyperbolic);
          if (sh.bsum >= 0) {
            int nf = sh.nbase + n;
            if (sh.gate_H > 0) { const int x = nf / sh.gate_H; nf = x * ((sh.gate_H + 15) & ~15) + nf - x * sh.gate_H; }
            ws[a.pk_off + sh.bsum + nf] = bs;
          }
        }
      }
    } else if (d.kind == DW_DECAY) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = local * 256 + e * 64 + lane;
        if (n < d.nrows) {
          const int64_t o = d.p_off + n;
          float p = P[o], m = M[o], v = V[o];
          adam_update(p, m, v, 0.f, co);
          P[o] = p; M[o] = m; V[o] = v;
        }
      }
    } else if (d.kind == DW_BALL && local == 0) {   // hyperbolic_linear.bias; gradient = sum of the per-tile partial column sums
      const float* left = ws + d.left_off;
      RowVec g;
#pragma unroll
      for (int e = 0; e < MAX_EPL; ++e) g.v[e] = 0.f;
      for (int r0 = 0; r0 < d.red_rows; r0 += 8) {           // eight partial rows in flight per round trip (fixed order)
        float t[8][MAX_EPL];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u < d.red_rows ? r0 + u : d.red_rows - 1;
#pragma unroll
          for (int e = 0; e < MAX_EPL; ++e) {
            const int c = lane + 64 * e;
            t[u][e] = c < d.nrows ? left[(int64_t)r * d.left_ld + c] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < MAX_EPL; ++e) g.v[e] += r0 + u < d.red_rows ? t[u][e] : 0.f;
      }
      radam_ball_wave(P + d.p_off, M + d.p_off, V + d.p_off, g, d.nrows, lane, co);
    }
  }
#if HYPAD_DIAG
  // development aid (scripts/diag_dw_items.py): wall-clock end of every wave of the generator's dW launch + its descriptor kind
  if (a.stamps && tab.finalize == 1 && blockIdx.y == 0 && lane == 0 && item < 1024) {
    a.stamps[3 * 48 * 8 + 64 + 2 * item] = (long long)__builtin_amdgcn_s_memrealtime();
    a.stamps[3 * 48 * 8 + 64 + 2 * item + 1] = item < tab.total_items ? d.kind * 1000 + d.net * 100 + (d.red_rows >> 4) : -1;
  }
#endif
  if (bx == 0 && threadIdx.x == 0) {
    if (tab.finalize == 1) {        // generator losses (train.py:232-234, 243-244)
      const GenWs gw = gen_ws(B_, S_, L_);
      float aux = 0.f, fx = 0.f, fz = 0.f;
      for (int t = 0; t < B_ / 16; ++t) {
        const float* part = ws + gw.partial + t * 4;
        aux += part[0]; fx += part[1]; fz += part[2];
      }
      aux = a.hyperbolic ? aux / a.B : aux / ((float)a.B * (float)a.S);
      float* lo = a.losses + sig * a.loss_sig_stride;
      lo[0] = 10.f * aux - fx / a.B - fz / a.B;
      lo[1] = aux; lo[2] = fx / a.B; lo[3] = fz / a.B;
    }
    if (blockIdx.y == 0 && a.tick_owner && a.step_add < 0) a.counters[3] += 1;     // rng tick: nobody reads it inside this kernel
  }
}
// ---- the generator's dW + Adam launch from prepared work-item records.  A wave of that launch used to spend 2.2 of its 5.5 us before its
// first operand load left: the workgroup's descriptor byte and the 52-byte descriptor out of a 3 KB kernel-argument table (whose
// scalar registers the compiler spilled), the tile's (n0, k0) by integer division, clamps, and shadow_ref()'s search for the packed
// positions of the updated weights.  All of that depends on the dimensions only: dw_items_kernel writes it once per pack launch, 128
// bytes per item, and a wave fetches its record with two scalar loads.
struct DwItem {                                // 32 words (train_common.h DW_ITEM_WORDS)
  int32_t kind;                                // DwKind, or -1: nothing to do (padding behind a descriptor's last item)
  int32_t net;
  int32_t n0, k0;                              // weight tile origin; bias: n0 = first column; decay: n0 = first element
  int32_t nrows, ncols, p_off, p_ld, p_off2;
  int32_t left_off, left_ld, right_off, right_ld, red_rows;
  int32_t fwd, fwd_kg, nbase, bwd, bwd_ng, mbase, bsum, gate_H;      // ShadowRef
  int32_t table_sig;                           // dw_table_sig of the table the records were written from: a launch that expects another table refuses them
  int32_t pad[9];
};
// what the records of a workspace depend on: the table (item count, with / without the decay-only ranges) and the dimensions.  The launch
// that consumes them carries the signature it expects; records written for another table (a pack with other arguments, or a caller that
// used the workspace as scratch in between) poison the step's loss instead of silently applying the wrong item set.
__host__ __device__ inline int32_t dw_table_sig(int total_items, int S, int L, int hyper, int n_desc) {
  uint32_t h = 0x9E3779B9u * (uint32_t)(total_items + 1);
  h ^= (uint32_t)S << 20; h ^= (uint32_t)L << 10; h ^= (uint32_t)n_desc << 2; h ^= (uint32_t)(hyper ? 1 : 0);
  return (int32_t)(h | 0x40000000u);          // never 0 (a zeroed workspace is not a table)
}
static_assert(sizeof(DwItem) == 4 * DW_ITEM_WORDS, "DwItem is DW_ITEM_WORDS words");
__global__ __launch_bounds__(256) void dw_items_kernel(DwTable tab, DwItem* __restrict__ out, int S, int L, int hyper) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= DW_ITEM_CAP) return;
  DwItem it{};
  it.kind = -1;
  it.table_sig = dw_table_sig(tab.total_items, S, L, hyper, tab.n);
  if (item < tab.total_items) {
    const int bx = item >> 2;
    const int di = (tab.block_desc[bx >> 2] >> (8 * (bx & 3))) & 0xff;
    const DwDesc d = tab.d[di];
    const int local = item - d.begin;
    bool live = false;
    if (d.kind == DW_WEIGHT) {
      const int tk = (d.ncols + 15) >> 4;
      live = local < ((d.nrows + 15) >> 4) * tk;
      it.n0 = (local / tk) * 16; it.k0 = (local % tk) * 16;
    } else if (d.kind == DW_BIAS) { live = local * 16 < d.nrows; it.n0 = local * 16; }
    else if (d.kind == DW_DECAY) { live = local * 256 < d.nrows; it.n0 = local * 256; }
    else if (d.kind == DW_BALL) live = local == 0;
    if (live) {
      it.kind = d.kind; it.net = d.net; it.nrows = d.nrows; it.ncols = d.ncols; it.p_off = d.p_off; it.p_ld = d.p_ld; it.p_off2 = d.p_off2;
      it.left_off = d.left_off; it.left_ld = d.left_ld; it.right_off = d.right_off; it.right_ld = d.right_ld; it.red_rows = d.red_rows;
      const ShadowRef sh = shadow_ref(d.net, d.p_off, S, L, hyper);
      it.fwd = sh.fwd; it.fwd_kg = sh.fwd_kg; it.nbase = sh.nbase; it.bwd = sh.bwd; it.bwd_ng = sh.bwd_ng; it.mbase = sh.mbase; it.bsum = sh.bsum; it.gate_H = sh.gate_H;
    }
  }
  out[item] = it;
}

// chunk > 0 (the spread placement): the records are cut into eight contiguous chunks of `chunk` items and workgroup bx takes its four
// from chunk bx mod 8 -- the workgroups of one chunk land on one XCD (round-robin dispatch), a matrix's tiles are neighbours in the
// table, so its operand rows are fetched into one or two L2s instead of all eight.  Speed / traffic only: the items are independent.
template <int SC, int LC, int BC, int KS>
__device__ __forceinline__ void dw_adam_items_body(const IterArgs& a, const DwItem* __restrict__ items, int total_items, const int bx, const int32_t table_sig,
                                                   const int chunk = 0, const int model = (int)blockIdx.y) {      // model: index inside the launch (blockIdx.y, or the co-located form's own)
  const int S_ = SC ? SC : a.S, L_ = LC ? LC : a.L, B_ = BC ? BC : a.B;
  const int sig = model + a.sig0;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int j = lane & 15, q = lane >> 4;
  float* ws = a.ws + sig * a.ws_sig_stride;
  int item = bx * (THREADS / 64) + wave;                // the launch covers total_items: one item per wave
  if (chunk > 0) {
    const int c = bx & 7, in_chunk = (bx >> 3) * (THREADS / 64) + wave;
    item = in_chunk < chunk ? c * chunk + in_chunk : total_items;
  }
  // the record (wave-uniform address: scalar loads from the constant address space -- nothing writes the table during this launch) ...
  using CWord = const __attribute__((address_space(4))) int32_t;
  DwItem d;
  {
    CWord* src = (CWord*)(items + (item < total_items ? item : 0));
    int32_t* dst = reinterpret_cast<int32_t*>(&d);
#pragma unroll
    for (int i = 0; i < 23; ++i) dst[i] = src[i];          // (the 23 words in use)
  }
  if (d.table_sig != table_sig) {                           // (wave-uniform) records of another table: refuse them loudly -- a NaN loss for this model
    if (threadIdx.x == 0 && a.losses) { float* lo = a.losses + sig * a.loss_sig_stride; lo[0] = lo[1] = __builtin_nanf(""); }
    return;
  }
  // ... and, in the same batch of scalar fetches, the step counter and the bias corrections the generator launch left behind
  const int step = a.counters[a.opt] + (a.step_add >= 0 ? a.step_add + 1 : 0);      // (step_add < 0: already incremented by the iteration's first kernel)
  const float* ac = a.ws + (int64_t)a.sig0 * a.ws_sig_stride + gen_ws(B_, S_, L_).adamc;
  AdamCoef co;
  co.lr = a.lr; co.b1 = a.b1; co.b2 = a.b2; co.eps = a.eps; co.wd = a.wd; co.riemannian = a.riemannian; co.stabilize = a.stabilize;
  co.step = step; co.bc1 = ac[0]; co.bc2 = ac[1]; co.sqrt_bc2 = ac[2];
  const int64_t arena = d.net == HYPAD_NET_ENCODER ? a.pe : a.pd;
  float* P = (d.net == HYPAD_NET_ENCODER ? a.P.enc : a.P.dec) + sig * arena;
  float* M = (d.net == HYPAD_NET_ENCODER ? a.M.enc : a.M.dec) + sig * arena;
  float* V = (d.net == HYPAD_NET_ENCODER ? a.V.enc : a.V.dec) + sig * arena;
  if (item < total_items && d.kind >= 0) {
    if (d.kind == DW_WEIGHT) {
      const int n0 = d.n0, k0 = d.k0;
      // out-of-range columns are clamped (their results are dropped below); rows past red_rows contribute zeros
      const int nj = n0 + j < d.nrows ? n0 + j : d.nrows - 1, kj = k0 + j < d.ncols ? k0 + j : d.ncols - 1;
      const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws + d.left_off), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws + d.right_off), 0, 0x7fffffff, 0x00020000);
      const int lvo = (q * d.left_ld + nj) * 4, rvo = (q * d.right_ld + kj) * 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // the optimiser state of this lane's four elements travels with the operand loads (one round trip, not two)
      int po[4]; float pp[4], pm[4], pvv[4];
      float* Pb = P + d.p_off; float* Mb = M + d.p_off; float* Vb = V + d.p_off;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * q + r, k = k0 + j;
        const bool ok = n < d.nrows && k < d.ncols;
        po[r] = ok ? n * d.p_ld + k : -1;
      }
      // (requested BEHIND the first batch of operand rows -- loads return in order, and the products need only the rows)
      auto load_state = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t oc = po[r] >= 0 ? (uint32_t)po[r] : 0u;
          pp[r] = Pb[oc]; pm[r] = Mb[oc]; pvv[r] = Vb[oc];
        }
      };
      if (d.red_rows <= 0) load_state();
      for (int rc = 0; rc < d.red_rows; rc += 4 * KS) {    // up to KS k-steps in flight
        float la[KS], rb[KS];
#pragma unroll
        for (int c = 0; c < KS / 4; ++c)
          if (rc + 16 * c < d.red_rows) {                  // wave-uniform
#pragma unroll
            for (int u = 4 * c; u < 4 * c + 4; ++u) {
              la[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, lvo, (rc + 4 * u) * d.left_ld * 4, 0));
              rb[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, rvo, (rc + 4 * u) * d.right_ld * 4, 0));
            }
          }
        if (rc == 0) load_state();
#pragma unroll
        for (int c = 0; c < KS / 4; ++c)
          if (rc + 16 * c < d.red_rows) {
#pragma unroll
            for (int u = 4 * c; u < 4 * c + 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc, 0, 0, 0);
          }
      }
      float* pk = ws + a.pk_off;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = pp[r], m = pm[r], v = pvv[r];
        adam_update(p, m, v, acc[r], co);
        if (po[r] >= 0) {
          const uint32_t o = (uint32_t)po[r];
          Pb[o] = p; Mb[o] = m; Vb[o] = v;
          const int nc = d.nbase + n0 + 4 * q + r, k = k0 + j;             // compact row, column
          if (d.fwd >= 0) {
            int nf = nc;                                                     // forward copy row: gates padded to 16-row blocks
            if (d.gate_H > 0) nf += ((nc >= d.gate_H) + (nc >= 2 * d.gate_H)) * (((d.gate_H + 15) & ~15) - d.gate_H);   // nc < 3H
            pk[d.fwd + (((nf >> 4) * d.fwd_kg + (k >> 4)) * 64 + (nf & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3)] = p;
          }
          if (d.bwd >= 0) {
            const int mm = d.mbase + nc;                                      // reduction index of the transposed copy
            pk[d.bwd + (((k >> 4) * d.bwd_ng + (mm >> 4)) * 64 + (k & 15) + 16 * ((mm & 15) >> 2)) * 4 + (mm & 3)] = p;
          }
        }
      }
    } else if (d.kind == DW_BIAS) {
      // 16 columns per item; lane (j, q) sums rows r = q (mod 4), then the four row classes are folded by shuffles
      const int n = d.n0 + j;
      const bool nv = n < d.nrows;
      const float* left = ws + d.left_off + (nv ? n : d.nrows - 1);
      const int rlast = d.red_rows - 1;
      // the optimiser state with the rows (one round trip: as a second one behind the sums it made the bias items the launch's long pole)
      const bool own = nv && q == 0;
      const int64_t o1 = d.p_off + (own ? n : 0), o2 = (d.p_off2 >= 0 ? d.p_off2 : d.p_off) + (own ? n : 0);
      const float p1_ = P[o1], m1_ = M[o1], v1_ = V[o1], p2_ = P[o2], m2_ = M[o2], v2_ = V[o2];
      float g = 0.f;
      for (int rc = 0; rc < d.red_rows; rc += 4 * KS) {    // KS rows per lane in flight
        float t[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          const int r = rc + 4 * u + q;
          t[u] = left[(int64_t)(r < rlast ? r : rlast) * d.left_ld];
        }
#pragma unroll
        for (int u = 0; u < KS; ++u) g += rc + 4 * u + q < d.red_rows ? t[u] : 0.f;
      }
      g += __shfl_xor(g, 16, WAVE);
      g += __shfl_xor(g, 32, WAVE);
      if (own) {
        float p = p1_, m = m1_, v = v1_;
        adam_update(p, m, v, g, co);
        P[o1] = p; M[o1] = m; V[o1] = v;
        float bs = p;
        if (d.p_off2 >= 0) {
          p = p2_; m = m2_; v = v2_;
          adam_update(p, m, v, g, co);
          P[o2] = p; M[o2] = m; V[o2] = v;
          bs += p;
        }
        if (d.bsum >= 0) {
          int nf = d.nbase + n;
          if (d.gate_H > 0) { const int x = nf / d.gate_H; nf = x * ((d.gate_H + 15) & ~15) + nf - x * d.gate_H; }
          ws[a.pk_off + d.bsum + nf] = bs;
        }
      }
    } else if (d.kind == DW_DECAY) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = d.n0 + e * 64 + lane;
        if (n < d.nrows) {
          const int64_t o = d.p_off + n;
          float p = P[o], m = M[o], v = V[o];
          adam_update(p, m, v, 0.f, co);
          P[o] = p; M[o] = m; V[o] = v;
        }
      }
    } else if (d.kind == DW_BALL) {                // hyperbolic_linear.bias; gradient = sum of the per-tile partial column sums
      const float* left = ws + d.left_off;
      // (the ball rows with the gradient's parts: one round trip -- this item is the longest of the launch)
      const RowVec Pr = row_load(P + d.p_off, d.nrows, lane), Mr = row_load(M + d.p_off, d.nrows, lane), Vr = row_load(V + d.p_off, d.nrows, lane);
      RowVec g;
#pragma unroll
      for (int e = 0; e < MAX_EPL; ++e) g.v[e] = 0.f;
      for (int r0 = 0; r0 < d.red_rows; r0 += 8) {           // eight partial rows in flight per round trip (fixed order)
        float t[8][MAX_EPL];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u < d.red_rows ? r0 + u : d.red_rows - 1;
#pragma unroll
          for (int e = 0; e < MAX_EPL; ++e) {
            const int c = lane + 64 * e;
            t[u][e] = c < d.nrows ? left[(int64_t)r * d.left_ld + c] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < MAX_EPL; ++e) g.v[e] += r0 + u < d.red_rows ? t[u][e] : 0.f;
      }
      radam_ball_wave(P + d.p_off, M + d.p_off, V + d.p_off, Pr, Mr, Vr, g, d.nrows, lane, co);
    }
  }
#if HYPAD_DIAG
  if (a.stamps && model == 0 && lane == 0 && item < 1024) {      // (scripts/diag_dw_items.py)
    a.stamps[3 * 48 * 8 + 64 + 2 * item] = (long long)__builtin_amdgcn_s_memrealtime();
    a.stamps[3 * 48 * 8 + 64 + 2 * item + 1] = item < total_items && d.kind >= 0 ? d.kind * 1000 + d.net * 100 + (d.red_rows >> 4) : -1;
  }
#endif
  if (bx == 0 && threadIdx.x == 0) {               // generator losses (train.py:232-234, 243-244)
    const GenWs gw = gen_ws(B_, S_, L_);
    float aux = 0.f, fx = 0.f, fz = 0.f;
    for (int t = 0; t < B_ / 16; ++t) {
      const float* part = ws + gw.partial + t * 4;
      aux += part[0]; fx += part[1]; fz += part[2];
    }
    aux = a.hyperbolic ? aux / a.B : aux / ((float)a.B * (float)a.S);
    float* lo = a.losses + sig * a.loss_sig_stride;
    lo[0] = 10.f * aux - fx / a.B - fz / a.B;
    lo[1] = aux; lo[2] = fx / a.B; lo[3] = fz / a.B;
    if (model == 0 && a.tick_owner && a.step_add < 0) a.counters[3] += 1;     // rng tick: nobody reads it inside this kernel
  }
}
// COLOC: one model's weight tiles share an XCD's L2 -- its ~0.9 MB of operand rows are fetched from HBM once, not once per XCD.
// Workgroups are dealt round-robin over the 8 XCDs, so in a ONE-dimensional grid workgroup id lands on XCD id mod 8: the launch's
// models go in groups of eight, id mod 8 picks the model of the group and id / 8 its workgroup -- 8 x blocks x ceil(models / 8)
// workgroups, none of them empty while the model count is a multiple of eight.  (Rounds 2-4 stretched blockIdx.x by 8 under a
// (blocks, models) grid and let seven of eight workgroups leave at once: 10 500 workgroups at 8 models, 1.2 us of the launch's
// 18.8 for the dispatcher to deal them -- 3.212 -> 3.178 ms per epoch at 8 models.  What the launch takes at 8 models and more
// is its bytes: 6.4 MB per model in one burst of loads and one of stores through the model's XCD, ~3 TB/s over the chip.)  Used
// from 8 signals per GPU on.  Speed only: no result depends on it.
template <int SC, int LC, int BC, int KS, bool COLOC = false>
__global__ __launch_bounds__(THREADS) void dw_adam_kernel(IterArgs a, const DwItem* __restrict__ items, int total_items, int chunk, int32_t table_sig) {
  if (a.guard && a.counters[4] != 0) return;
  if constexpr (COLOC) {
    // chunk carries the launch's model count here
    const int blocks = (total_items + THREADS / 64 - 1) / (THREADS / 64);
    const int t = (int)(blockIdx.x >> 3), model = (t / blocks) * 8 + (int)(blockIdx.x & 7);
    if (model >= chunk) return;
    dw_adam_items_body<SC, LC, BC, KS>(a, items, total_items, t % blocks, table_sig, 0, model);
  } else {
    // (chunk > 0: the grid is 8 x chunk / 4 workgroups, chunk = ceil(total / 8) rounded up to a multiple of four)
    dw_adam_items_body<SC, LC, BC, KS>(a, items, total_items, (int)blockIdx.x, table_sig, chunk);
  }
}
__global__ __launch_bounds__(THREADS) void dw_adam_small_kernel(IterArgs a, DwTableS tab) { dw_adam_body(a, tab); }
__global__ __launch_bounds__(THREADS) void dw_adam_pair_kernel(IterArgs ax, DwTableS tx, IterArgs az, DwTableS tz) {
  if (blockIdx.z == 0) dw_adam_body(ax, tx); else dw_adam_body(az, tz);
}

// ------------------------------------------------------------------------------------------------ host: tables
template <class Table>
struct TableBuilder {
  Table t;
  bool overflow = false;
  TableBuilder() { t.n = 0; t.total_items = 0; t.finalize = 0; for (int i = 0; i < Table::blk / 4; ++i) t.block_desc[i] = 0; }
  void push(DwDesc d, int items) {
    if (items <= 0) return;
    const int start = (t.total_items + 3) & ~3;          // descriptors start on a workgroup boundary
    const int b0 = start / 4, b1 = (start + items + 3) / 4;
    if (t.n >= Table::cap || b1 > Table::blk) { overflow = true; return; }
    d.begin = start;
    for (int b = b0; b < b1; ++b) t.block_desc[b >> 2] |= (uint32_t)t.n << (8 * (b & 3));
    t.d[t.n++] = d;
    t.total_items = start + items;
  }
  Table done() { if (overflow) t.n = -1; return t; }
  void weight(int net, int p_off, int p_ld, int nrows, int ncols, int left_off, int left_ld, int right_off, int right_ld, int red) {
    DwDesc d{};
    d.kind = DW_WEIGHT; d.net = (int16_t)net; d.p_off = p_off; d.p_ld = p_ld; d.nrows = nrows; d.ncols = ncols;
    d.left_off = left_off; d.left_ld = left_ld; d.right_off = right_off; d.right_ld = right_ld; d.red_rows = red; d.p_off2 = -1;
    push(d, ((nrows + 15) / 16) * ((ncols + 15) / 16));
  }
  void bias(int net, int p_off, int p_off2, int n, int left_off, int left_ld, int red) {
    DwDesc d{};
    d.kind = DW_BIAS; d.net = (int16_t)net; d.p_off = p_off; d.p_off2 = p_off2; d.nrows = n; d.ncols = 1;
    d.left_off = left_off; d.left_ld = left_ld; d.red_rows = red;
    push(d, (n + 15) / 16);
  }
  bool skip_decay = false;                     // decay-only ranges are left to decay_steps_kernel (hypad_train_epoch)
  void decay(int net, int p_off, int n) {
    if (skip_decay) return;
    DwDesc d{};
    d.kind = DW_DECAY; d.net = (int16_t)net; d.p_off = p_off; d.nrows = n; d.p_off2 = -1;
    push(d, (n + 255) / 256);
  }
  void ball(int net, int p_off, int n, int left_off, int left_ld, int red) {
    DwDesc d{};
    d.kind = DW_BALL; d.net = (int16_t)net; d.p_off = p_off; d.nrows = n; d.left_off = left_off; d.left_ld = left_ld;
    d.red_rows = red; d.p_off2 = -1;
    push(d, 1);
  }
  // one LSTM direction: weight_ih rows [0,H) <- compact cols [c0, c0+H); rows [2H,4H) <- cols [c0+H, c0+3H)
  void lstm_dir(int net, const LstmDir& ld, int H, int K, int left_off, int left_ld, int c0, int right_off, int right_ld, int red,
                bool decay_too) {
    weight(net, ld.w_ih, K, H, K, left_off + c0, left_ld, right_off, right_ld, red);
    weight(net, ld.w_ih + 2 * H * K, K, 2 * H, K, left_off + c0 + H, left_ld, right_off, right_ld, red);
    bias(net, ld.b_ih, ld.b_hh, H, left_off + c0, left_ld, red);
    bias(net, ld.b_ih + 2 * H, ld.b_hh + 2 * H, 2 * H, left_off + c0 + H, left_ld, red);
    if (decay_too) {   // no data gradient: f-gate rows and W_hh still decay under RiemannianAdam
      decay(net, ld.w_ih + H * K, H * K);
      decay(net, ld.w_hh, 4 * H * H);
      decay(net, ld.b_ih + H, H);
      decay(net, ld.b_hh + H, H);
    }
  }
};

DwTableS critic_table(int net, const CriticLayout& cl, const CritWs& cw, int B, int L) {
  TableBuilder<DwTableS> tb;
  for (int li = 0; li <= cl.nh; ++li) {
    int k = li == 0 ? cl.in_dim : L, n = li == cl.nh ? 1 : L;
    int right = li == 0 ? cw.in_right : cw.act[li - 1];
    tb.weight(net, cl.w[li], k, n, k, cw.left[li], n, right, k, 3 * B);
    tb.bias(net, cl.b[li], -1, n, cw.left[li], n, 2 * B);      // the GP rows carry no bias gradient
  }
  return tb.done();
}

// Parameters that never receive a data gradient -- W_hh and the f-gate rows of W_ih / b_ih / b_hh of every LSTM direction (h0 =
// c0 = 0 at T = 1: SURVEY.md A.2) -- but still move under RiemannianAdam's weight decay (train.py:286): 119 k of the generator's
// 245 k floats at window 100, 45 % of the dW + Adam launch's optimiser traffic.  No kernel ever reads them, so inside an epoch
// they are advanced ONCE, after the last generator step, by all the epoch's steps at a time (decay_steps_kernel): the same
// arithmetic in the same order, element by element -- bit-identical -- with 1 / n_batches of the traffic.
struct DecayTable { int n, total; int net[24], off[24], len[24], begin[24]; };
DecayTable decay_table(const hypad_dims& dm) {
  DecayTable t{};
  const int S = dm.signal_shape, L = dm.latent_dim;
  const EncLayout el = enc_layout(S, L);
  const DecLayout dl = dec_layout(S, L, dm.hyperbolic);
  auto add = [&](int net, int off, int len) { t.net[t.n] = net; t.off[t.n] = off; t.len[t.n] = len; t.begin[t.n] = t.total; t.total += (len + 255) & ~255; ++t.n; };
  auto dir = [&](int net, const LstmDir& ld, int H, int K) { add(net, ld.w_ih + H * K, H * K); add(net, ld.w_hh, 4 * H * H); add(net, ld.b_ih + H, H); add(net, ld.b_hh + H, H); };
  for (int d = 0; d < 2; ++d) dir(HYPAD_NET_ENCODER, el.dir[d], ENC_H, S);
  for (int d = 0; d < 2; ++d) { dir(HYPAD_NET_DECODER, dl.l[0][d], DEC_H, DEC_D1); dir(HYPAD_NET_DECODER, dl.l[1][d], DEC_H, 2 * DEC_H); }
  return t;
}
// the last `nsteps` generator steps of the decay-only parameters; counters[opt] already holds the last step's number
__global__ __launch_bounds__(256) void decay_steps_kernel(IterArgs a, DecayTable tab, int nsteps) {
  __shared__ float bc[2 * 64];
  if (a.guard && a.counters[4] != 0) return;
  const int sig = blockIdx.y;
  const int last = a.counters[a.opt];
  const int e = blockIdx.x * 256 + threadIdx.x;
  int r = 0;
  for (int i = 1; i < tab.n; ++i) r = e >= tab.begin[i] ? i : r;          // (ranges start on 256-element boundaries: block-uniform)
  const int idx = e - tab.begin[r];
  const bool live = idx < tab.len[r];
  const int net = tab.net[r];
  const int64_t o = (int64_t)sig * (net == HYPAD_NET_ENCODER ? a.pe : a.pd) + tab.off[r] + (live ? idx : 0);
  float* P = (net == HYPAD_NET_ENCODER ? a.P.enc : a.P.dec) + o;
  float* M = (net == HYPAD_NET_ENCODER ? a.M.enc : a.M.dec) + o;
  float* V = (net == HYPAD_NET_ENCODER ? a.V.enc : a.V.dec) + o;
  float p = *P, m = *M, v = *V;
  AdamCoef co;
  co.lr = a.lr; co.b1 = a.b1; co.b2 = a.b2; co.eps = a.eps; co.wd = a.wd; co.riemannian = a.riemannian; co.stabilize = a.stabilize; co.step = 0;
  co.bc1 = 1.f; co.bc2 = 1.f; co.sqrt_bc2 = 1.f;
  for (int s0 = 0; s0 < nsteps; s0 += 64) {                               // bias corrections of 64 steps at a time (double-precision powers)
    __syncthreads();
    if (threadIdx.x < 64 && s0 + threadIdx.x < nsteps) {
      const AdamCoef c = adam_coef(a.lr, a.b1, a.b2, a.eps, a.wd, a.riemannian, a.stabilize, last - nsteps + 1 + s0 + threadIdx.x);
      bc[2 * threadIdx.x] = c.bc1; bc[2 * threadIdx.x + 1] = c.bc2;
    }
    __syncthreads();
    const int n = nsteps - s0 < 64 ? nsteps - s0 : 64;
    for (int k = 0; k < n; ++k) {
      co.bc1 = bc[2 * k]; co.bc2 = bc[2 * k + 1];
      adam_update(p, m, v, 0.f, co);
    }
  }
  if (live) { *P = p; *M = m; *V = v; }
}

DwTable gen_table(const hypad_dims& dm, bool with_decay = true) {
  const int B = dm.batch, S = dm.signal_shape, L = dm.latent_dim;
  const bool hyp = dm.hyperbolic != 0;
  const EncLayout el = enc_layout(S, L);
  const DecLayout dl = dec_layout(S, L, dm.hyperbolic);
  const GenWs gw = gen_ws(B, S, L);
  TableBuilder<DwTable> tb;
  tb.t.finalize = 1;
  tb.skip_decay = !with_decay;
  for (int d = 0; d < 2; ++d)
    tb.lstm_dir(HYPAD_NET_ENCODER, el.dir[d], ENC_H, S, gw.dgenc, 6 * ENC_H, d * 3 * ENC_H, gw.xg, S, 2 * B, hyp);     // chains R and Z
  tb.weight(HYPAD_NET_ENCODER, el.dense_w, 2 * ENC_H, L, 2 * ENC_H, gw.dzenc, L, gw.enc_h, 2 * ENC_H, 2 * B);
  tb.bias(HYPAD_NET_ENCODER, el.dense_b, -1, L, gw.dzenc, L, 2 * B);
  tb.weight(HYPAD_NET_DECODER, dl.d1_w, L, DEC_D1, L, gw.da0, DEC_D1, gw.zcat, L, 2 * B);
  tb.bias(HYPAD_NET_DECODER, dl.d1_b, -1, DEC_D1, gw.da0, DEC_D1, 2 * B);
  for (int d = 0; d < 2; ++d) {
    tb.lstm_dir(HYPAD_NET_DECODER, dl.l[0][d], DEC_H, DEC_D1, gw.dg0, 6 * DEC_H, d * 3 * DEC_H, gw.a0, DEC_D1, 2 * B, hyp);
    tb.lstm_dir(HYPAD_NET_DECODER, dl.l[1][d], DEC_H, 2 * DEC_H, gw.dg1, 6 * DEC_H, d * 3 * DEC_H, gw.h0d, 2 * DEC_H, 2 * B, hyp);
  }
  tb.weight(HYPAD_NET_DECODER, dl.d2_w, 2 * DEC_H, S, 2 * DEC_H, gw.dpre2, S, gw.h1, 2 * DEC_H, 2 * B);
  tb.bias(HYPAD_NET_DECODER, dl.d2_b, -1, S, gw.dpre2, S, 2 * B);
  if (hyp) {
    tb.weight(HYPAD_NET_DECODER, dl.head_w, S, S, S, gw.du, S, gw.ecat, S, 3 * B);
    tb.ball(HYPAD_NET_DECODER, dl.head_b, S, gw.ballpart, S, 2 * (B / 16));      // per (role, tile) partial column sums
  }
  return tb.done();
}

// ------------------------------------------------------------------------------------------------ host: launches
int check_dims(const hypad_dims* d) {
  if (!d || d->signal_shape <= 0 || d->latent_dim <= 0 || d->batch <= 0 || d->n_signals <= 0 || d->first_signal < 0) return HYPAD_EINVAL;
  if (d->batch % 16 != 0) return HYPAD_EINVAL;
  if (d->signal_shape > MAX_S || d->latent_dim > MAX_L) return HYPAD_EUNSUPPORTED;
  if (gen_table(*d).n < 0) return HYPAD_EUNSUPPORTED;      // cannot happen within MAX_S / MAX_L; the critic tables are far smaller
  // the closed forms device code uses for a critic's tensor offsets (layout.h CriticLayout::wof / bof) ARE the layout's table
  const CriticLayout cls[2] = {cx_layout(d->signal_shape, d->latent_dim), cz_layout(d->latent_dim)};
  for (const CriticLayout& cl : cls)
    for (int li = 0; li <= cl.nh; ++li)
      if (cl.wof(li) != cl.w[li] || cl.bof(li) != cl.b[li]) return HYPAD_EUNSUPPORTED;
  return HYPAD_OK;
}

hipError_t allow_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ------------------------------------------------------------------------------------------------ packed generator weights
// Builds the MFMA-native copies (layout.h GenPack) from the parameter arenas: one thread per float4 of a packed block.
struct PackDesc {
  int kind;            // 0 W, 1 W with LSTM gate rows compacted, 2 W^T, 3 [W_fwd; W_rev]^T with compacted gate rows, 4 summed biases, 5 critic_x padded image,
                       // 6 snapshot of the critics' parameters, moments and the counters (hypad_epoch_restore)
  int net;             // HYPAD_NET_ENCODER / HYPAD_NET_DECODER
  int dst, nout, kred; // packed matrix: nout output rows, kred reduction columns
  int src0, src1, ld, H;
};
struct PackTable { int n; int max_units; PackDesc d[32]; };
// Layout of the epoch snapshot (hypad_epoch_restore): per signal [P.cx | M.cx | V.cx | P.cz | M.cz | V.cz], then counters[0..3]
HD int64_t snapshot_floats(int pcx, int pcz, int n_signals) { return (int64_t)n_signals * 3 * (pcx + pcz) + 4; }
// one float4 of one signal's snapshot: to the snapshot (restore = false) or back to the arenas
__device__ __forceinline__ void snapshot_move(const IterArgs& a, float* snap, int sig, int u, bool restore) {
  const int qx = a.pcx / 4, qz = a.pcz / 4;
  if (u >= 3 * (qx + qz)) return;
  float4* sn = reinterpret_cast<float4*>(snap + (int64_t)sig * 3 * (a.pcx + a.pcz)) + u;
  const bool is_x = u < 3 * qx;
  const int v = is_x ? u : u - 3 * qx, q = is_x ? qx : qz;
  const int which = v / q, e = v - which * q;
  const hypad_nets& N = which == 0 ? a.P : (which == 1 ? a.M : a.V);
  float4* ar = reinterpret_cast<float4*>(is_x ? N.cx + (int64_t)sig * a.pcx : N.cz + (int64_t)sig * a.pcz) + e;
  if (restore) *ar = *sn; else *sn = *ar;
}
__global__ __launch_bounds__(256) void epoch_restore_kernel(IterArgs a, float* snap, int n_signals) {
  snapshot_move(a, snap, blockIdx.y, blockIdx.x * 256 + threadIdx.x, true);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8) {
    const int32_t* saved = reinterpret_cast<const int32_t*>(snap + (int64_t)n_signals * 3 * (a.pcx + a.pcz));
    a.counters[threadIdx.x] = threadIdx.x < 4 ? saved[threadIdx.x] : 0;
  }
}
__global__ __launch_bounds__(256) void pack_generator_kernel(IterArgs a, PackTable tab, unsigned* zero_ptr, int zero_words, float* snap) {
  if (a.guard && a.counters[4] != 0) return;   // fail-stop: in particular the snapshot of the state the failed epoch began from stays
  if (a.guard && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) a.counters[5] = 0;      // (placement census of the resident launch)
  if (zero_words) {                            // (launch_pack: one word per thread of the grid)
    const int64_t flat = (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    if (flat < zero_words) zero_ptr[flat] = 0u;
  }
  const PackDesc d = tab.d[blockIdx.y];
  const int sig = blockIdx.z;
  float* pk = a.ws + sig * a.ws_sig_stride + a.pk_off + d.dst;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (d.kind == 6) {
    snapshot_move(a, snap, sig, u, false);
    if (u < 4 && sig == 0) reinterpret_cast<int32_t*>(snap + (int64_t)gridDim.z * 3 * (a.pcx + a.pcz))[u] = a.counters[u];
    return;
  }
  if (d.kind == 5) {                           // critic_x as the padded image of critic_mfma.h (stage_critic_padded's arithmetic, once)
    const CriticLayout cl = cx_layout(a.S, a.L);
    const CriticPad cp = critic_pad(a.S, a.L, 4);
    if (u < cp.total) {
      const float* C = a.P.cx + (int64_t)sig * a.pcx;
      const int L = a.L, in_dim = cl.in_dim;
      float v;
      if (u < cp.wh) { const int n = u / cp.ldin, k = u - n * cp.ldin; v = k < in_dim ? C[cl.wof(0) + n * in_dim + k] : (k == in_dim ? C[cl.bof(0) + n] : 0.f); }
      else if (u < cp.wl) {
        const int i = u - cp.wh, li = 1 + i / (L * cp.LQ), rem = i - (li - 1) * L * cp.LQ, n = rem / cp.LQ, k = rem - n * cp.LQ;
        v = k < L ? C[cl.wof(li) + n * L + k] : (k == L ? C[cl.bof(li) + n] : 0.f);
      } else { const int k = u - cp.wl; v = k < L ? C[cl.wof(cl.nh) + k] : (k == L ? C[cl.bof(cl.nh)] : 0.f); }
      pk[u] = v;
    }
    return;
  }
  const float* P = d.net == HYPAD_NET_ENCODER ? a.P.enc + (int64_t)sig * a.pe : a.P.dec + (int64_t)sig * a.pd;
  auto gate_row = [&](int n) { return n < d.H ? n : n + d.H; };           // compact [i|g|o] -> PyTorch [i,f,g,o] row
  const int Hp = (d.H + 15) & ~15;
  // forward gate matrices: packed row x * Hp + u (gate x of unit u; layout.h gate_rows) -> PyTorch row, or -1 for padding
  auto padded_gate_row = [&](int n) { const int x = n / Hp, u = n - x * Hp; return u < d.H ? (x == 0 ? 0 : x == 1 ? 2 * d.H : 3 * d.H) + u : -1; };
  if (d.kind == 4) {
    if (u < ((d.nout + 15) & ~15)) {
      float v = 0.f;
      if (u < d.nout) {
        const int r = d.H > 0 ? padded_gate_row(u) : u;
        if (r >= 0) v = P[d.src0 + r] + (d.src1 >= 0 ? P[d.src1 + r] : 0.f);
      }
      pk[u] = v;
    }
    return;
  }
  const int kg = (d.kred + 15) >> 4, units = ((d.nout + 15) >> 4) * kg * 64;
  if (u >= units) return;
  const int blk = u >> 6, lane = u & 63, tn = blk / kg, g = blk - tn * kg;
  const int n = 16 * tn + (lane & 15), k0 = 16 * g + 4 * (lane >> 4);
  float v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = k0 + c;
    float x = 0.f;
    if (n < d.nout && k < d.kred) {
      if (d.kind == 0) x = P[d.src0 + n * d.ld + k];
      else if (d.kind == 1) { const int r = padded_gate_row(n); x = r >= 0 ? P[d.src0 + r * d.ld + k] : 0.f; }
      else if (d.kind == 2) x = P[d.src0 + k * d.ld + n];
      else { const int dir = k >= 3 * d.H ? 1 : 0, kc = k - dir * 3 * d.H; x = P[(dir ? d.src1 : d.src0) + gate_row(kc) * d.ld + n]; }
    }
    v[c] = x;
  }
  reinterpret_cast<float4*>(pk)[u] = make_float4(v[0], v[1], v[2], v[3]);
}
// where the scoring kernel's padded critic_x image sits in its workspace: behind the packed generator weights, 16-byte aligned
HD int score_critic_offset(int S, int L, int hyperbolic) { return (gen_pack(S, L, hyperbolic).total + 3) & ~3; }
// nets: bit 0 the encoder's copies, bit 1 the decoder's; forward_only: without the transposed copies of the backward products
PackTable pack_table(const hypad_dims& dm, bool with_critic = false, bool with_snapshot = false, int nets = 3, bool forward_only = false) {
  const int S = dm.signal_shape, L = dm.latent_dim;
  const EncLayout el = enc_layout(S, L);
  const DecLayout dl = dec_layout(S, L, dm.hyperbolic);
  const GenPack gp = gen_pack(S, L, dm.hyperbolic);
  PackTable t; t.n = 0; t.max_units = 0;
  auto push = [&](int kind, int net, int dst, int nout, int kred, int src0, int src1, int ld, int H) {
    if (!(nets & (net == HYPAD_NET_ENCODER ? 1 : 2)) || (forward_only && (kind == 2 || kind == 3))) return;
    PackDesc d{kind, net, dst, nout, kred, src0, src1, ld, H};
    t.d[t.n++] = d;
    const int units = kind == 4 ? ((nout + 15) & ~15) : ((nout + 15) >> 4) * ((kred + 15) >> 4) * 64;
    if (units > t.max_units) t.max_units = units;
  };
  const int E = HYPAD_NET_ENCODER, D = HYPAD_NET_DECODER;
  for (int d = 0; d < 2; ++d) {
    push(1, E, gp.enc_g[d], gate_rows(ENC_H), S, el.dir[d].w_ih, -1, S, ENC_H);
    push(4, E, gp.enc_gb[d], gate_rows(ENC_H), 0, el.dir[d].b_ih, el.dir[d].b_hh, 0, ENC_H);
  }
  push(0, E, gp.enc_d, L, 2 * ENC_H, el.dense_w, -1, 2 * ENC_H, 0);
  push(4, E, gp.enc_db, L, 0, el.dense_b, -1, 0, 0);
  push(0, D, gp.d1, DEC_D1, L, dl.d1_w, -1, L, 0);
  push(4, D, gp.d1b, DEC_D1, 0, dl.d1_b, -1, 0, 0);
  for (int l = 0; l < 2; ++l) {
    const int in = l == 0 ? DEC_D1 : 2 * DEC_H;
    for (int d = 0; d < 2; ++d) {
      push(1, D, gp.l_g[l][d], gate_rows(DEC_H), in, dl.l[l][d].w_ih, -1, in, DEC_H);
      push(4, D, gp.l_gb[l][d], gate_rows(DEC_H), 0, dl.l[l][d].b_ih, dl.l[l][d].b_hh, 0, DEC_H);
    }
    push(3, D, gp.l_t[l], in, 6 * DEC_H, dl.l[l][0].w_ih, dl.l[l][1].w_ih, in, DEC_H);
  }
  push(0, D, gp.d2, S, 2 * DEC_H, dl.d2_w, -1, 2 * DEC_H, 0);
  push(4, D, gp.d2b, S, 0, dl.d2_b, -1, 0, 0);
  if (dm.hyperbolic) {
    push(0, D, gp.head, S, S, dl.head_w, -1, S, 0);
    push(2, D, gp.head_t, S, S, dl.head_w, -1, S, 0);
  }
  push(2, E, gp.enc_d_t, 2 * ENC_H, L, el.dense_w, -1, 2 * ENC_H, 0);
  push(2, D, gp.d2_t, 2 * DEC_H, S, dl.d2_w, -1, 2 * DEC_H, 0);
  push(2, D, gp.d1_t, L, DEC_D1, dl.d1_w, -1, L, 0);
  if (with_critic) {                           // behind the generator's copies (hypad_score_workspace_bytes reserves it)
    const int units = critic_pad(S, L, 4).total;
    PackDesc d{5, HYPAD_NET_CRITIC_X, score_critic_offset(S, L, dm.hyperbolic), units, 0, 0, -1, 0, 0};
    t.d[t.n++] = d;
    if (units > t.max_units) t.max_units = units;
  }
  if (with_snapshot) {
    const int units = 3 * (cx_layout(S, L).total + cz_layout(L).total) / 4;
    PackDesc d{6, HYPAD_NET_CRITIC_X, 0, units, 0, 0, -1, 0, 0};
    t.d[t.n++] = d;
    if (units > t.max_units) t.max_units = units;
  }
  return t;
}
// zero_ptr / zero_words: a block the next launches need zeroed (the critic phase's epoch words and flags), one word per thread
// snap: where to snapshot the critics' state for hypad_epoch_restore, or null
int launch_pack(const IterArgs& a, const hypad_dims& dm, hipStream_t s, unsigned* zero_ptr = nullptr, int zero_words = 0, bool* zeroed = nullptr,
                bool with_critic = false, float* snap = nullptr, int nets = 3, bool forward_only = false) {
  const PackTable t = pack_table(dm, with_critic, snap != nullptr, nets, forward_only);
  const dim3 grid((t.max_units + 255) / 256, t.n, dm.n_signals);
  const bool z = zero_ptr && zero_words > 0 && (int64_t)zero_words <= (int64_t)grid.x * grid.y * grid.z * 256;
  if (zeroed) *zeroed = z;
  hipLaunchKernelGGL(pack_generator_kernel, grid, dim3(256), 0, s, a, t, z ? zero_ptr : nullptr, z ? zero_words : 0, snap);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

// ------------------------------------------------------------------------------------------------ scoring forward, packed
// The test-loop body (anomaly_detection.py:67-113, eval mode) on the training kernels' machinery: 512 threads per 16 rows,
// packed weights (no LDS re-shape), LSTM cells in the gate products' epilogues, the critic as LDS-resident MFMA layers, the
// row-wise ball math four rows per wave.  The critic value of the windows (anomaly_detection.py:96-101) comes from its own launch,
// critic_rows_kernel: inside this kernel it was six workgroup barriers and an 18 KB weight image per 16 windows for 1 % of the FLOPs
// (0.526 -> 0.479 ms per 125 000 windows without it; the launch below takes 0.02).
//
// critic_x over rows, eval mode: the padded weight image (critic_mfma.h CriticPad) goes into LDS once per workgroup, then every WAVE walks
// its own 16-row tiles -- rows into a wave-private LDS tile (the next tile's rows are requested into registers before the layers of
// the current one), four layers on wave_gemm_nt with nothing but wave-local fences between them, the last layer as 16 dot products.
// Same products in the same order as critic_tile_fwd: same bits.
constexpr int CR_WAVES = 8;                  // waves per workgroup (fewer where the wave-private tiles of a wide window would not fit the LDS)
HD int critic_rows_lds_floats(int S, int L, int waves) {
  const CriticPad cp = critic_pad(S, L, 4);
  return cp.total + waves * (16 * cp.ldin + 2 * 16 * cp.LQ);
}
template <int SC, int LC>
__global__ __launch_bounds__(64 * CR_WAVES) void critic_rows_kernel(const float* __restrict__ cxpad, const float* __restrict__ x, int64_t x_ld,
                                                                    float* __restrict__ out, int64_t rows, int S_, int L_) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = SC ? SC : S_, L = LC ? LC : L_;
  const CriticPad cp = critic_pad(S, L, 4);
  const int ldin = cp.ldin, LQ = cp.LQ, nh = 4;
  const int lane = threadIdx.x & 63, wave = wave_id();
  float* in = smem + cp.total + wave * (16 * ldin + 2 * 16 * LQ);
  float* act = in + 16 * ldin;
  stage_params(smem, cxpad, cp.total);
  // the tile's constant part: the ones column behind the window (the layer's bias sits in that column of the image), zero padding
  for (int i = lane; i < 16 * ldin; i += 64) { const int c = i % ldin; in[i] = c == S ? 1.f : 0.f; }
  for (int i = lane; i < 2 * 16 * LQ; i += 64) act[i] = 0.f;
  __syncthreads();
  const float* w0 = smem + cp.w0; const float* wh = smem + cp.wh; const float* wl = smem + cp.wl;
  const int nwaves = blockDim.x >> 6;
  const int64_t tiles = (rows + 15) >> 4, stride = (int64_t)gridDim.x * nwaves;
  constexpr int NV = SC ? (16 * SC + 63) / 64 : 1;     // floats of a tile per lane (compiled-in window; any other streams its rows)
  float xr[NV];
  auto fetch = [&](int64_t t) __attribute__((always_inline)) {
    const int64_t r0 = t * 16;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int i = lane + 64 * u, r = i / S, c = i - r * S;
      const int64_t row = r0 + r < rows ? r0 + r : rows - 1;
      xr[u] = i < 16 * S ? x[row * x_ld + c] : 0.f;
    }
  };
  int64_t t = (int64_t)blockIdx.x * nwaves + wave;
  if constexpr (SC != 0) { if (t < tiles) fetch(t); }
  for (; t < tiles; t += stride) {
    if constexpr (SC != 0) {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = lane + 64 * u, r = i / S, c = i - r * S;
        if (i < 16 * S) in[r * ldin + c] = xr[u];
      }
    } else {
      for (int i = lane; i < 16 * S; i += 64) {
        const int r = i / S, c = i - r * S;
        const int64_t row = t * 16 + r < rows ? t * 16 + r : rows - 1;
        in[r * ldin + c] = x[row * x_ld + c];
      }
    }
    wave_lds_fence();
    if constexpr (SC != 0) { if (t + stride < tiles) fetch(t + stride); }          // the next tile's rows arrive under this tile's layers
    for (int li = 0; li < nh; ++li) {
      const float* A = li == 0 ? in : act + ((li - 1) & 1) * 16 * LQ;
      const float* Wl = li == 0 ? w0 : wh + (li - 1) * L * LQ;
      float* ao = act + (li & 1) * 16 * LQ;
      wave_gemm_nt(A, li == 0 ? ldin : LQ, Wl, li == 0 ? ldin : LQ, L, L + 1, li == 0 ? cp.Kin : cp.Lp, lane, [&](int r, int c, float pre) {
        if (c < L) ao[r * LQ + c] = pre * leaky_slope(pre);
        else if (c == L) ao[r * LQ + c] = 1.f;
      });
      wave_lds_fence();
    }
    if (lane < 16) {
      const float* xa = act + ((nh - 1) & 1) * 16 * LQ + lane * LQ;
      float o = 0.f;
      for (int c = 0; c <= L; ++c) o += xa[c] * wl[c];
      if (t * 16 + lane < rows) out[t * 16 + lane] = o;
    }
    wave_lds_fence();
  }
}
int launch_critic_rows(const float* cxpad, const float* x, int64_t x_ld, float* out, int64_t rows, int S, int L, hipStream_t s) {
  int waves = CR_WAVES;
  while (waves > 1 && (size_t)critic_rows_lds_floats(S, L, waves) * sizeof(float) > 160 * 1024) waves >>= 1;
  const size_t lds = (size_t)critic_rows_lds_floats(S, L, waves) * sizeof(float);
  if (lds > 160 * 1024) return HYPAD_EUNSUPPORTED;
  const int64_t tiles = (rows + 15) / 16;
  // (its LDS plan puts one workgroup on a CU: 256 of them cover an MI355X, the tiles go round)
  const unsigned grid = (unsigned)std::min<int64_t>((tiles + waves - 1) / waves, 256);
  if (S == 100 && L == 20) {
    hipError_t e = allow_lds((const void*)critic_rows_kernel<100, 20>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((critic_rows_kernel<100, 20>), dim3(grid), dim3(64 * waves), lds, s, cxpad, x, x_ld, out, rows, S, L);
  } else {
    hipError_t e = allow_lds((const void*)critic_rows_kernel<0, 0>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((critic_rows_kernel<0, 0>), dim3(grid), dim3(64 * waves), lds, s, cxpad, x, x_ld, out, rows, S, L);
  }
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
struct ScoreArgs {
  const float* pk; const float* head_b; const float* x; int64_t x_ld;
  float* hyper; float* eucl; float* hyper_real; float* rowdist;
  int64_t rows; int S, L, hyperbolic;
};
struct ScoreLds { int xs, zs, bufA, bufB, total, ldS; };
#ifndef HYPAD_SCORE_WPE
#define HYPAD_SCORE_WPE 4
#endif
HD ScoreLds score_lds(int S, int L, int MT) {
  ScoreLds p; int o = 0;
  p.ldS = lds_stride(S);
  const int rows = 16 * MT;
  int buf = rows * (2 * DEC_H + 4) > rows * p.ldS ? rows * (2 * DEC_H + 4) : rows * p.ldS;       // h tiles / e and head tiles
  if (MT == 1 && buf < 32 * p.ldS) buf = 32 * p.ldS;                                             // (the 32-row head tile of the 16-window form)
  buf = (buf + 3) & ~3;
  p.xs = o; o += rows * p.ldS;
  p.zs = o; o += rows * LP;
  p.bufA = o; o += buf;
  p.bufB = o; o += buf;
  p.total = o;
  return p;
}
// MT = 1: 16 windows per workgroup (small calls: more workgroups).  MT = 2: 32 windows -- every weight block a wave fetches feeds two
// row tiles, the stages' barriers and per-tile set-up are paid once per 32 windows.  Same products in the same order per row: same bits.
template <int SC, int LC, int MT>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(HYPAD_SCORE_WPE, HYPAD_SCORE_WPE))) void score_forward_packed_kernel(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ROWS = 16 * MT;
  const int S = SC ? SC : a.S, L = LC ? LC : a.L;
  const ScoreLds lp = score_lds(S, L, MT);
  const int ldS = lp.ldS;
  const GenPack gp = gen_pack(S, L, a.hyperbolic);
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  const int64_t r0 = (int64_t)blockIdx.x * ROWS;
  const int valid = (int)(a.rows - r0 < ROWS ? a.rows - r0 : ROWS);
  const int lane = threadIdx.x & 63, wave = wave_id();
  tile_load_b(xs, ldS, a.x + r0 * a.x_ld, (int)a.x_ld, ROWS, S, valid);
  __syncthreads();
  encoder_fwd_tile_packed<false, false, MT>(xs, ldS, S, L, a.pk, gp, bufA, ENC_LDG, bufB, ENC_LDH, zs, nullptr, nullptr, valid);
  DecSave none{16, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  decoder_trunk_fwd_tile_packed<MT>(zs, L, S, a.pk, gp, bufA, bufB, ldS, no_drop(), [](int r) { return r; }, none, valid);
  if (a.eucl) tile_store_b(a.eucl + r0 * S, S, bufA, ldS, ROWS, S, valid);
  if (a.hyperbolic) {
    // the head on the reconstruction AND on the real windows (anomaly_detection.py:84-90): u rows of both, then the ball rows
    float* urec; float* ureal;
    if constexpr (MT == 1) {
      for (int i = threadIdx.x; i < 16 * ldS; i += blockDim.x) bufA[16 * ldS + i] = xs[i];      // rows 16-31: the real windows
      __syncthreads();
      gemm_nt_packed<2>(bufA, ldS, S, S, a.pk + gp.head, nullptr, bufB, ldS, 0);
      __syncthreads();
      head_rows_tile(bufB, ldS, 32, S, a.head_b);
      urec = bufB; ureal = bufB + 16 * ldS;
    } else {
      gemm_nt_packed<MT>(bufA, ldS, S, S, a.pk + gp.head, nullptr, bufB, ldS, 0);
      __syncthreads();                                       // (e has been read -- by the product and by the store above)
      gemm_nt_packed<MT>(xs, ldS, S, S, a.pk + gp.head, nullptr, bufA, ldS, 0);
      __syncthreads();
      head_rows_tile(bufB, ldS, ROWS, S, a.head_b);
      head_rows_tile(bufA, ldS, ROWS, S, a.head_b);
      urec = bufB; ureal = bufA;
    }
    __syncthreads();
    if (a.hyper) tile_store_b(a.hyper + r0 * S, S, urec, ldS, ROWS, S, valid);
    if (a.hyper_real) tile_store_b(a.hyper_real + r0 * S, S, ureal, ldS, ROWS, S, valid);
    if (a.rowdist && wave < 4 * MT) {
      // (pred = real window on the ball, true = reconstruction): anomaly_detection_utils.py:58-65; four rows per wave
      epl16_dispatch(S, [&](auto tag) {
        using R16 = RowT<16, decltype(tag)::value>;
        const int r = wave * 4 + (lane >> 4);
        const float d = rowdist_row(row_load<R16>(ureal + r * ldS, S, lane), row_load<R16>(urec + r * ldS, S, lane));
        if ((lane & 15) == 0 && r < valid) a.rowdist[r0 + r] = d;
      });
    }
  }
}

#if HYPAD_DIAG
long long* g_gen_stamps = nullptr;     // development builds only (libhypad_hip_dev.so)
#endif

struct IterCall {
  const float* x; int64_t x_sig_stride; int64_t x_row_stride; const int32_t* row_index; int64_t ri_sig_stride = 0;
  const float* z; const float* alpha;
  int train_mode; const float* masks; uint64_t seed;
  float* losses; int64_t loss_sig_stride;
  void* workspace; size_t workspace_bytes;
  int guard = 0;
  int flags = 0;                         // hypad_epoch_io.flags (HYPAD_EPOCH_DW_* select the dW + Adam launch's placement)
};

// opt: 0 critic_x, 1 critic_z, 2 generator
int fill_args(IterArgs& a, const hypad_dims* d, const hypad_train_state* st, const IterCall& io, int opt) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !st->params.enc || !st->params.dec || !st->params.cx || !st->params.cz || !st->counters) return HYPAD_EINVAL;
  if (!io.x || !io.losses) return HYPAD_EINVAL;
  const int64_t per = ws_floats_per_signal(*d);
  if (!io.workspace || io.workspace_bytes < (size_t)per * d->n_signals * sizeof(float)) return HYPAD_EWORKSPACE;
  a.S = d->signal_shape; a.L = d->latent_dim; a.B = d->batch; a.hyperbolic = d->hyperbolic;
  a.P = st->params; a.M = st->exp_avg; a.V = st->exp_avg_sq;
  a.pe = enc_layout(a.S, a.L).total; a.pd = dec_layout(a.S, a.L, a.hyperbolic).total;
  a.pcx = cx_layout(a.S, a.L).total; a.pcz = cz_layout(a.L).total;
  a.counters = st->counters;
  a.x = io.x; a.x_sig_stride = io.x_sig_stride; a.x_ld = io.x_row_stride > 0 ? io.x_row_stride : d->signal_shape; a.row_index = io.row_index;
  a.z = io.z; a.alpha = io.alpha;
  a.drop_mode = io.train_mode ? (io.masks ? 1 : 2) : 0;
  a.masks = io.masks; a.seed = io.seed;
  a.losses = io.losses; a.loss_sig_stride = io.loss_sig_stride;
  a.ws = (float*)io.workspace + (opt == 1 ? ws_cz_offset(*d) : 0);
  a.ws_sig_stride = per;
  a.pk_off = ws_pack_offset(*d) - (opt == 1 ? ws_cz_offset(*d) : 0);       // a.ws is shifted for critic_z
  a.lr = st->lr; a.b1 = st->beta1; a.b2 = st->beta2; a.eps = st->eps; a.wd = 0.f; a.stabilize = 0; a.riemannian = 0;
  a.opt = opt; a.tick_owner = 1; a.stamps = nullptr; a.guard = io.guard; a.sig0 = 0; a.step_add = -1;
  a.rng_sig0 = d->first_signal; a.ri_sig_stride = io.ri_sig_stride;
#if HYPAD_DIAG
  a.stamps = g_gen_stamps;
#endif
  if (opt == 0) {
    if (!st->exp_avg.cx || !st->exp_avg_sq.cx) return HYPAD_EINVAL;
    a.mask_sig_stride = (int64_t)12 * a.B * a.L + (int64_t)a.B * 2 * DEC_H;
  } else if (opt == 1) {
    if (!st->exp_avg.cz || !st->exp_avg_sq.cz) return HYPAD_EINVAL;
    a.mask_sig_stride = (int64_t)6 * a.B * a.L;
  } else {
    if (!st->exp_avg.enc || !st->exp_avg_sq.enc || !st->exp_avg.dec || !st->exp_avg_sq.dec) return HYPAD_EINVAL;
    a.mask_sig_stride = (int64_t)6 * a.B * a.L + (int64_t)2 * a.B * 2 * DEC_H;
    if (a.hyperbolic) { a.riemannian = 1; a.wd = st->gen_weight_decay; a.stabilize = st->gen_stabilize; }
  }
  return HYPAD_OK;
}

// Optional profiling marks: ev[k] is recorded on the stream after the k-th kernel of an iteration (ev[0] before the first).
#define HYPAD_MARK(ev, k, s) do { if (ev) (void)hipEventRecord((ev)[k], s); } while (0)

inline int dw_blocks(int items) { int b = (items + 3) / 4; return b < 1 ? 1 : b; }
inline size_t cx_pass_lds(const IterArgs& a) {
  return (size_t)lds_plan(a.S, 16, 16, cx_layout(a.S, a.L).total, critic_batch_lds_floats(48, a.L)).total * sizeof(float);
}
inline size_t cz_pass_lds(const IterArgs& a) {
  return (size_t)lds_plan(a.S, 16, 16, cz_layout(a.L).total, critic_batch_lds_floats(48, a.L)).total * sizeof(float);
}
inline size_t gp_x_lds(const IterArgs& a) { return (size_t)gp_lds_floats(a.S, cx_layout(a.S, a.L).total) * sizeof(float); }
inline size_t gp_z_lds(const IterArgs& a) { return (size_t)gp_lds_floats(a.L, cz_layout(a.L).total) * sizeof(float); }

int run_cx(const hypad_dims* d, const hypad_train_state* st, const IterCall& io, hipStream_t s, hipEvent_t* ev = nullptr) {
  IterArgs a;
  int rc = fill_args(a, d, st, io, 0);
  if (rc) return rc;
  const size_t lds = cx_pass_lds(a), lds2 = gp_x_lds(a);
  hipError_t e = allow_lds((const void*)cx_pass_kernel, lds);
  if (e == hipSuccess) e = allow_lds((const void*)critic_gp_kernel<true>, lds2);
  if (e != hipSuccess) return (int)e;
  dim3 grid(a.B / 16, d->n_signals);
  HYPAD_MARK(ev, 0, s);
  hipLaunchKernelGGL(cx_pass_kernel, grid, dim3(TB), lds, s, a);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 1, s);
  hipLaunchKernelGGL(critic_gp_kernel<true>, grid, dim3(TB), lds2, s, a);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 2, s);
  const DwTableS tab = critic_table(HYPAD_NET_CRITIC_X, cx_layout(a.S, a.L), crit_ws(a.B, a.S, a.L, 4), a.B, a.L);
  hipLaunchKernelGGL(dw_adam_small_kernel, dim3(dw_blocks(tab.total_items), d->n_signals), dim3(THREADS), 0, s, a, tab);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 3, s);
  return HYPAD_OK;
}

int run_cz(const hypad_dims* d, const hypad_train_state* st, const IterCall& io, hipStream_t s, hipEvent_t* ev = nullptr) {
  IterArgs a;
  int rc = fill_args(a, d, st, io, 1);
  if (rc) return rc;
  const size_t lds = cz_pass_lds(a), lds2 = gp_z_lds(a);
  hipError_t e = allow_lds((const void*)cz_pass_kernel, lds);
  if (e == hipSuccess) e = allow_lds((const void*)critic_gp_kernel<false>, lds2);
  if (e != hipSuccess) return (int)e;
  dim3 grid(a.B / 16, d->n_signals);
  HYPAD_MARK(ev, 0, s);
  hipLaunchKernelGGL(cz_pass_kernel, grid, dim3(TB), lds, s, a);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 1, s);
  hipLaunchKernelGGL(critic_gp_kernel<false>, grid, dim3(TB), lds2, s, a);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 2, s);
  const DwTableS tab = critic_table(HYPAD_NET_CRITIC_Z, cz_layout(a.L), crit_ws(a.B, a.L, a.L, 2), a.B, a.L);
  hipLaunchKernelGGL(dw_adam_small_kernel, dim3(dw_blocks(tab.total_items), d->n_signals), dim3(THREADS), 0, s, a, tab);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 3, s);
  return HYPAD_OK;
}

// critic_x_iteration and critic_z_iteration of one minibatch side by side (train.py:320-327: disjoint weights, frozen
// generator): three launches with blockIdx.z selecting the critic.  losses_x / losses_z: where each writes its 4 floats.
// io_z: critic_z's own call description when the two iterations take different injected planes (null: io for both).
int run_critic_pair(const hypad_dims* d, const hypad_train_state* st, const IterCall& io, float* losses_x, float* losses_z,
                    hipStream_t s, hipEvent_t* ev = nullptr, const IterCall* io_z = nullptr) {
  IterArgs ax, az;
  IterCall cx = io, cz = io_z ? *io_z : io;
  cx.losses = losses_x; cz.losses = losses_z;
  int rc = fill_args(ax, d, st, cx, 0);
  if (rc) return rc;
  rc = fill_args(az, d, st, cz, 1);
  if (rc) return rc;
  az.tick_owner = 0;                       // one rng tick per launch group; the two critics use distinct Philox streams
  az.seed = ax.seed ^ CRITIC_Z_SEED_XOR;
  size_t lds = cx_pass_lds(ax), l2 = cz_pass_lds(az);
  if (l2 > lds) lds = l2;
  size_t ldsg = gp_x_lds(ax), g2 = gp_z_lds(az);
  if (g2 > ldsg) ldsg = g2;
  hipError_t e = allow_lds((const void*)critic_pass_pair_kernel, lds);
  if (e == hipSuccess) e = allow_lds((const void*)critic_gp_pair_kernel, ldsg);
  if (e != hipSuccess) return (int)e;
  dim3 grid(ax.B / 16, d->n_signals, 2);
  HYPAD_MARK(ev, 0, s);
  hipLaunchKernelGGL(critic_pass_pair_kernel, grid, dim3(TB), lds, s, ax, az);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 1, s);
  hipLaunchKernelGGL(critic_gp_pair_kernel, grid, dim3(TB), ldsg, s, ax, az);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 2, s);
  const DwTableS tx = critic_table(HYPAD_NET_CRITIC_X, cx_layout(ax.S, ax.L), crit_ws(ax.B, ax.S, ax.L, 4), ax.B, ax.L);
  const DwTableS tz = critic_table(HYPAD_NET_CRITIC_Z, cz_layout(az.L), crit_ws(az.B, az.L, az.L, 2), az.B, az.L);
  const int items = tx.total_items > tz.total_items ? tx.total_items : tz.total_items;
  hipLaunchKernelGGL(dw_adam_pair_kernel, dim3(dw_blocks(items), d->n_signals, 2), dim3(THREADS), 0, s, ax, tx, az, tz);
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 3, s);
  return HYPAD_OK;
}

// the work-item records of the generator's dW + Adam launch (DwItem), into signal 0's workspace: behind every pack launch whose
// workspace generator steps will run in (run_gen with pack = true; hypad_train_epoch once per epoch)
int launch_dw_items(const IterArgs& a, const hypad_dims& d, bool with_decay, hipStream_t s) {
  const DwTable tab = gen_table(d, with_decay);
  if (tab.n < 0 || tab.total_items > DW_ITEM_CAP) return HYPAD_EUNSUPPORTED;
  DwItem* items = reinterpret_cast<DwItem*>(a.ws + ws_items_offset(d));
  hipLaunchKernelGGL(dw_items_kernel, dim3(DW_ITEM_CAP / 256), dim3(256), 0, s, tab, items, d.signal_shape, d.latent_dim, d.hyperbolic);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

// with_decay = false: the decay-only parameters are left alone (the caller advances them with run_decay_steps)
// sig0 / nsig / step_add: the models [sig0, sig0 + nsig) only, as generator iteration `step_add` of an epoch (IterArgs.step_add);
// defaults: all models, counters read and advanced by the launches themselves.
// reps > 1 (hypad_profile_iteration kind 5): each of the two kernels launched `reps` times back to back between its events -- an
// event pair around ONE short launch measures the event path too (10 - 20 % on a 9 us kernel)
int run_gen(const hypad_dims* d, const hypad_train_state* st, const IterCall& io, hipStream_t s, hipEvent_t* ev = nullptr, bool pack = true,
            bool with_decay = true, int sig0 = 0, int nsig = -1, int step_add = -1, int reps = 1) {
  IterArgs a;
  int rc = fill_args(a, d, st, io, 2);
  if (rc) return rc;
  if (nsig < 0) nsig = d->n_signals;
  a.sig0 = sig0; a.step_add = step_add;
  dim3 grid(8 * (a.B / 16), nsig, 3);                    // blockIdx.x >> 3: tile (see gen_kernel); blockIdx.z: role G / R / Z
  const int l0 = gen_lds(a.S, a.L, a.hyperbolic, 0).total, l1 = gen_lds(a.S, a.L, a.hyperbolic, 1).total;
  const size_t lds = (size_t)(l0 > l1 ? l0 : l1) * sizeof(float);
  if (lds > 160 * 1024) return HYPAD_EUNSUPPORTED;
  if (pack) {                             // the workspace is scratch between calls: rebuild the packed weights
    rc = launch_pack(a, *d, s);           // (inside an epoch the dW + Adam kernel keeps them current)
    if (!rc) rc = launch_dw_items(a, *d, with_decay, s);
    if (rc) return rc;
  }
  HYPAD_MARK(ev, 0, s);
  const bool ref_cfg = a.S == 100 && a.L == 20 && a.B == 64;          // BASELINE.json configs[0..2]
  const bool mv_cfg = a.S == 150 && a.L == 20 && a.B == 256;          // configs[3]: 5 channels x 30, batch 256
#define HYPAD_LAUNCH_GEN(...)                                                     \
  do {                                                                           \
    hipError_t e = allow_lds((const void*)gen_kernel<__VA_ARGS__>, lds);         \
    if (e != hipSuccess) return (int)e;                                          \
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gen_kernel<__VA_ARGS__>), grid, dim3(TB), lds, s, a);    \
  } while (0)
  const bool wadi_cfg = a.S == 123 && a.L == 20 && a.B == 64;         // the reference's configs/multivariate.yaml:5-7 as shipped (WADI)
  const bool swat_cfg = a.S == 51 && a.L == 20 && a.B == 64;          // ... and its SWAT alternative
  if (a.hyperbolic) {
    if (ref_cfg) HYPAD_LAUNCH_GEN(true, 100, 20, 64); else if (mv_cfg) HYPAD_LAUNCH_GEN(true, 150, 20, 256);
    else if (wadi_cfg) HYPAD_LAUNCH_GEN(true, 123, 20, 64); else if (swat_cfg) HYPAD_LAUNCH_GEN(true, 51, 20, 64); else HYPAD_LAUNCH_GEN(true, 0, 0, 0);
  }
  else { if (ref_cfg) HYPAD_LAUNCH_GEN(false, 100, 20, 64); else HYPAD_LAUNCH_GEN(false, 0, 0, 0); }
#undef HYPAD_LAUNCH_GEN
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 1, s);
  const DwTable tab = gen_table(*d, with_decay);           // (its item count; the records themselves were written behind the pack launch)
  const DwItem* items = reinterpret_cast<const DwItem*>(a.ws + ws_items_offset(*d));
  const int total_items = tab.total_items;
  const int32_t tsig = dw_table_sig(tab.total_items, d->signal_shape, d->latent_dim, d->hyperbolic, tab.n);      // (the records carry the signature of the table they were written from)
  const bool coloc = (io.flags & HYPAD_EPOCH_DW_COLOC) ? true : (io.flags & HYPAD_EPOCH_DW_SPREAD) ? false : d->n_signals >= 8;      // (measured: -2 % of the epoch at 8-32 signals, +8 % at 1-2: few signals' tiles want all of the chip's CUs)
  const int dw_chunk = ((tab.total_items + 7) / 8 + 3) & ~3;          // spread placement: eight chunks of the records, one per XCD (dw_adam_items_body)
  const dim3 dgrid = coloc ? dim3(8 * dw_blocks(tab.total_items) * ((nsig + 7) / 8)) : dim3(8 * (dw_chunk / 4), nsig);
  for (int r = 0; r < reps; ++r) {
    if (coloc) {
      if (ref_cfg) hipLaunchKernelGGL((dw_adam_kernel<100, 20, 64, HYPAD_DW_KS_MANY, true>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
      else if (mv_cfg) hipLaunchKernelGGL((dw_adam_kernel<150, 20, 256, HYPAD_DW_KS_MANY, true>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
      else hipLaunchKernelGGL((dw_adam_kernel<0, 0, 0, HYPAD_DW_KS_MANY, true>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
    } else if (ref_cfg) hipLaunchKernelGGL((dw_adam_kernel<100, 20, 64, 48>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
    else if (mv_cfg) hipLaunchKernelGGL((dw_adam_kernel<150, 20, 256, 48>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
    else if (wadi_cfg) hipLaunchKernelGGL((dw_adam_kernel<123, 20, 64, 48>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
    else if (swat_cfg) hipLaunchKernelGGL((dw_adam_kernel<51, 20, 64, 48>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
    else hipLaunchKernelGGL((dw_adam_kernel<0, 0, 0, 48>), dgrid, dim3(THREADS), 0, s, a, items, total_items, coloc ? nsig : dw_chunk, tsig);
  }
  HYPAD_CHECK_LAUNCH();
  HYPAD_MARK(ev, 2, s);
  return HYPAD_OK;
}

int run_decay_steps(const hypad_dims* d, const hypad_train_state* st, const IterCall& io, int nsteps, hipStream_t s) {
  if (!d->hyperbolic || nsteps <= 0) return HYPAD_OK;          // torch.optim.Adam (Euclidean generator) has no weight decay: nothing moves
  IterArgs a;
  int rc = fill_args(a, d, st, io, 2);
  if (rc) return rc;
  const DecayTable tab = decay_table(*d);
  hipLaunchKernelGGL(decay_steps_kernel, dim3(tab.total / 256, d->n_signals), dim3(256), 0, s, a, tab, nsteps);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

IterCall from_io(const hypad_iter_io* io) {
  IterCall c;
  c.x = io->x; c.x_sig_stride = io->x_signal_stride; c.x_row_stride = io->x_row_stride; c.row_index = io->row_index; c.z = io->z; c.alpha = io->alpha;
  c.train_mode = io->drop.train_mode; c.masks = io->drop.masks; c.seed = io->drop.seed;
  c.losses = io->losses; c.loss_sig_stride = 4; c.workspace = io->workspace; c.workspace_bytes = io->workspace_bytes;
  return c;
}

// ---- stand-alone optimizers
__global__ __launch_bounds__(THREADS) void adam_flat_kernel(float* p, const float* g, float* m, float* v, int64_t n, int step,
                                                             float lr, float b1, float b2, float eps, float wd, int riem,
                                                             int64_t ball_off, int ball_dim) {
  const AdamCoef co = adam_coef(lr, b1, b2, eps, wd, riem, 0, step);
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
    if (ball_dim > 0 && i >= ball_off && i < ball_off + ball_dim) continue;
    float pp = p[i], mm = m[i], vv = v[i];
    float gg = g[i];
    if (!riem) gg += wd * pp;     // torch.optim.Adam weight_decay: L2 into the gradient
    adam_update(pp, mm, vv, gg, co);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}
__global__ __launch_bounds__(64) void radam_ball_kernel(float* p, const float* g, float* m, float* v, int dim, int step, float lr,
                                                         float b1, float b2, float eps, float wd, int stabilize) {
  const AdamCoef co = adam_coef(lr, b1, b2, eps, wd, 1, stabilize, step);
  radam_ball_wave(p, m, v, row_load(g, dim, threadIdx.x), dim, threadIdx.x, co);
}

}  // namespace

// development aid (not declared in hypad.h): device buffer of 64 int64 stamped by the generator kernel, or null
#if HYPAD_DIAG
extern "C" __attribute__((visibility("default"))) void hypad_diag_set_gen_stamps(long long* p) { g_gen_stamps = p; }
#endif

extern "C" {

// Floats in front of the hoisted critic phase's area in an epoch workspace: the per-signal iteration workspaces, then the snapshot
// of the critics' state (hypad_epoch_restore), 64-byte aligned
static size_t epoch_base_floats(const hypad_dims& d) {
  const size_t snap = (size_t)snapshot_floats(cx_layout(d.signal_shape, d.latent_dim).total, cz_layout(d.latent_dim).total, d.n_signals);
  return (size_t)ws_floats_per_signal(d) * d.n_signals + ((snap + 15) & ~(size_t)15);
}
static float* epoch_snapshot_ptr(const hypad_dims& d, void* workspace) { return (float*)workspace + (size_t)ws_floats_per_signal(d) * d.n_signals; }

size_t hypad_train_workspace_bytes(const hypad_dims* d) {
  if (check_dims(d)) return 0;
  return (size_t)ws_floats_per_signal(*d) * d->n_signals * sizeof(float);
}

int hypad_pack_generator(const hypad_dims* d, const hypad_train_state* st, void* workspace, size_t workspace_bytes, hypad_stream_t s) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !st->params.enc || !st->params.dec || !workspace) return HYPAD_EINVAL;
  const int64_t per = ws_floats_per_signal(*d);
  if (workspace_bytes < (size_t)per * d->n_signals * sizeof(float)) return HYPAD_EWORKSPACE;
  IterArgs a{};
  a.S = d->signal_shape; a.L = d->latent_dim; a.B = d->batch; a.hyperbolic = d->hyperbolic;
  a.P = st->params;
  a.pe = enc_layout(a.S, a.L).total; a.pd = dec_layout(a.S, a.L, a.hyperbolic).total;
  a.ws = (float*)workspace; a.ws_sig_stride = per; a.pk_off = ws_pack_offset(*d);
  return launch_pack(a, *d, (hipStream_t)s);
}
size_t hypad_score_workspace_bytes(int S, int L, int hyperbolic) {
  if (S < 1 || S > MAX_S || L < 1 || L > MAX_L) return 0;
  return (size_t)(score_critic_offset(S, L, hyperbolic) + critic_pad(S, L, 4).total) * sizeof(float);      // packed generator + padded critic_x
}
int hypad_score_forward_packed(const float* enc, const float* dec, const float* cx, const float* x, int64_t x_row_stride, float* hyper,
                               float* eucl, float* hyper_real, float* critic, float* rowdist, int64_t rows, int S, int L,
                               int hyperbolic, void* workspace, size_t workspace_bytes, hypad_stream_t s) {
  if (S < 1 || S > MAX_S || L < 1 || L > MAX_L) return HYPAD_EUNSUPPORTED;
  if (!enc || !dec || !x || rows < 0 || (critic && !cx)) return HYPAD_EINVAL;
  if (x_row_stride > (1 << 24)) return HYPAD_EUNSUPPORTED;       // (a tile's rows are addressed with 32-bit byte offsets)
  if (!workspace || workspace_bytes < hypad_score_workspace_bytes(S, L, hyperbolic)) return HYPAD_EWORKSPACE;
  if (rows == 0) return HYPAD_OK;
  hypad_dims d; d.signal_shape = S; d.latent_dim = L; d.batch = 16; d.hyperbolic = hyperbolic; d.n_signals = 1; d.first_signal = 0;
  IterArgs pa{};
  pa.S = S; pa.L = L; pa.B = 16; pa.hyperbolic = hyperbolic;
  pa.P.enc = const_cast<float*>(enc); pa.P.dec = const_cast<float*>(dec);
  pa.pe = enc_layout(S, L).total; pa.pd = dec_layout(S, L, hyperbolic).total;
  pa.ws = (float*)workspace; pa.ws_sig_stride = 0; pa.pk_off = 0;
  pa.P.cx = const_cast<float*>(cx); pa.pcx = cx_layout(S, L).total;
  int rc = launch_pack(pa, d, (hipStream_t)s, nullptr, 0, nullptr, cx != nullptr);
  if (rc) return rc;
  if (critic) {          // (the image the pack launch wrote: critic_mfma.h CriticPad)
    rc = launch_critic_rows((const float*)workspace + score_critic_offset(S, L, hyperbolic), x, x_row_stride > 0 ? x_row_stride : S, critic, rows, S, L, (hipStream_t)s);
    if (rc) return rc;
  }
  if (!hyper && !eucl && !hyper_real && !rowdist) return HYPAD_OK;      // (only the critic value was asked for)
  ScoreArgs a;
  a.pk = (const float*)workspace; a.head_b = hyperbolic ? dec + dec_layout(S, L, 1).head_b : nullptr;
  a.x = x; a.x_ld = x_row_stride > 0 ? x_row_stride : S;
  a.hyper = hyper; a.eucl = eucl; a.hyper_real = hyper_real; a.rowdist = rowdist;
  a.rows = rows; a.S = S; a.L = L; a.hyperbolic = hyperbolic;
  // 32 windows per workgroup once that still leaves every CU several workgroups (two are resident on a CU at a time)
  // (the reference window only: the run-time-shape build of the 32-window form and the one for window 150 do not fit 128 registers)
  const bool ref_shape = S == 100 && L == 20;
  const int mt = ref_shape && rows >= (int64_t)32 * 2048 ? 2 : 1;
  const size_t lds = (size_t)score_lds(S, L, mt).total * sizeof(float);
  if (lds > 160 * 1024) return HYPAD_EUNSUPPORTED;
  const int64_t tiles = (rows + 16 * mt - 1) / (16 * mt);
  if (tiles > 0x7fffffff) return HYPAD_EINVAL;
  const void* fn = ref_shape ? (mt == 2 ? (const void*)score_forward_packed_kernel<100, 20, 2> : (const void*)score_forward_packed_kernel<100, 20, 1>)
                   : (const void*)score_forward_packed_kernel<0, 0, 1>;
  hipError_t e = allow_lds(fn, lds);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {&a};
  e = hipLaunchKernel(fn, dim3((unsigned)tiles), dim3(TB), kargs, lds, (hipStream_t)s);
  if (e != hipSuccess) return (int)e;
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_packed_region(const hypad_dims* d, int64_t* offset_floats, int64_t* signal_stride_floats, int64_t* count_floats) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (offset_floats) *offset_floats = ws_pack_offset(*d);
  if (signal_stride_floats) *signal_stride_floats = ws_floats_per_signal(*d);
  if (count_floats) *count_floats = gen_pack(d->signal_shape, d->latent_dim, d->hyperbolic).total;
  return HYPAD_OK;
}
// One critic's iteration as a ONE-ITERATION PHASE of the hoisted form (critic_fused.hip): pack the frozen generator half the critic
// looks at (forward copies only), record precompute (decoder(z) or encoder(x), noise / interpolation / dropout -> record), the
// iteration launch (register-resident MFMA chains -> gradient slabs) and its finalising launch (slab reduction, whole-batch norm,
// Adam, weights and moments back to the arenas, loss row) -- ~40 us of GPU time against 76 (critic_x) / 47 (critic_z) for the
// stand-alone pass / gradient-penalty / dW launches below, whose critic passes run on the vector ALU.  Taken when the caller's
// workspace has room for it (hypad_epoch_workspace_bytes(dims, 1, 1)) and the shape fits the iteration kernel; HYPAD_ITER_PHASE=0
// keeps the stand-alone launches.  Same arithmetic per row as the epoch's critic phase; against the stand-alone launches only the
// floating-point summation order (and the device dropout streams) differ.
static int run_critic_single(const hypad_dims* d, const hypad_train_state* st, const hypad_iter_io* io, int critic, hipStream_t s, bool* taken) {
  *taken = false;
  static const int enabled = HYPAD_TUNE_INT("HYPAD_ITER_PHASE", 1);
  if (!enabled || check_dims(d) || !critic_phase_supported(*d)) return HYPAD_OK;
  const size_t base = epoch_base_floats(*d);
  const size_t have = io->workspace_bytes / sizeof(float);
  if (!io->workspace || have < base + critic_phase_fixed_floats(*d) + critic_phase_floats_per_iter(*d)) return HYPAD_OK;
  IterCall c = from_io(io);
  IterArgs ax, az;
  int rc = fill_args(ax, d, st, c, 0);
  if (!rc) rc = fill_args(az, d, st, c, 1);
  if (rc) return rc;
  *taken = true;
  rc = launch_pack(critic == 0 ? ax : az, *d, s, nullptr, 0, nullptr, false, nullptr, critic == 0 ? 2 : 1, true);
  if (rc) return rc;
  hypad_epoch_noise nz{};
  const bool inj = io->drop.train_mode && io->drop.masks;
  if (critic == 0) { nz.z_cx = io->z; nz.alpha_cx = io->alpha; nz.masks_cx = inj ? io->drop.masks : nullptr; }
  else { nz.z_cz = io->z; nz.alpha_cz = io->alpha; nz.masks_cz = inj ? io->drop.masks : nullptr; }
  // (the phase files critic_x's loss row at row 0 and critic_z's at row 1 of an iteration: the caller's row is row `critic`)
  return run_critic_phase(ax, az, io->row_index, 1, io->losses - 4 * critic, (float*)io->workspace + base, have - base, d->n_signals, s, nullptr,
                          &nz, nullptr, nullptr, HYPAD_EPOCH_PER_ITERATION, critic);
}
int hypad_critic_x_iteration(const hypad_dims* d, const hypad_train_state* st, const hypad_iter_io* io, hypad_stream_t s) {
  if (!io) return HYPAD_EINVAL;
  bool taken = false;
  const int rc = run_critic_single(d, st, io, 0, (hipStream_t)s, &taken);
  if (taken || rc) return rc;
  return run_cx(d, st, from_io(io), (hipStream_t)s);
}
int hypad_critic_z_iteration(const hypad_dims* d, const hypad_train_state* st, const hypad_iter_io* io, hypad_stream_t s) {
  if (!io) return HYPAD_EINVAL;
  bool taken = false;
  const int rc = run_critic_single(d, st, io, 1, (hipStream_t)s, &taken);
  if (taken || rc) return rc;
  return run_cz(d, st, from_io(io), (hipStream_t)s);
}
int hypad_decoder_iteration(const hypad_dims* d, const hypad_train_state* st, const hypad_iter_io* io, hypad_stream_t s) {
  if (!io) return HYPAD_EINVAL;
  return run_gen(d, st, from_io(io), (hipStream_t)s);
}

// Profiling aid for bench.py: runs ONE iteration with HIP events between its kernels on `stream`, synchronises the
// stream and returns each kernel's duration in milliseconds (critic iterations: pass, gp, dw_adam; generator: gen,
// dw_adam).  Not capturable (creates events, synchronises).
int hypad_profile_iteration(int kind, const hypad_dims* d, const hypad_train_state* st, const hypad_iter_io* io, float* ms_out,
                            int n_out, hypad_stream_t s) {
  if (!io || !ms_out || kind < 0 || kind > 5) return HYPAD_EINVAL;
  const int nk = (kind == 2 || kind == 5) ? 2 : 3;
  constexpr int GEN_REPS = 64;
  if (n_out < nk) return HYPAD_EINVAL;
  if (kind >= 3 && !io->losses) return HYPAD_EINVAL;
  hipEvent_t ev[4];
  int kind4_div = 0;
  for (int i = 0; i <= nk; ++i) {
    hipError_t e = hipEventCreate(&ev[i]);
    if (e != hipSuccess) return (int)e;
  }
  int rc;
  if (kind == 0) rc = run_cx(d, st, from_io(io), (hipStream_t)s, ev);
  else if (kind == 1) rc = run_cz(d, st, from_io(io), (hipStream_t)s, ev);
  else if (kind == 2) rc = run_gen(d, st, from_io(io), (hipStream_t)s, ev);
  else if (kind == 5) rc = run_gen(d, st, from_io(io), (hipStream_t)s, ev, true, false, 0, -1, -1, GEN_REPS);      // (no decay-only tensors: the launch an epoch issues)
  else if (kind == 3) rc = run_critic_pair(d, st, from_io(io), io->losses, io->losses + 4 * (int64_t)d->n_signals, (hipStream_t)s, ev);
  else {
    IterArgs ax, az;
    IterCall c = from_io(io);
    constexpr int PROF_ITERS = 145;        // the critic phase of one configs[1] epoch (5 passes x 29 minibatches): ONE launch of 145
                                           // iterations in the persistent form -- the same launch hypad_train_epoch issues, so a
                                           // profiler's mean duration of that kernel is over like launches -- or one launch without
                                           // an Adam prologue + 144 steady-state launches
    c.loss_sig_stride = 2 * PROF_ITERS * 4;
    rc = fill_args(ax, d, st, c, 0);
    if (!rc) rc = fill_args(az, d, st, c, 1);
    const size_t base = epoch_base_floats(*d);
    if (!rc && !critic_phase_supported(*d)) rc = HYPAD_EUNSUPPORTED;
    if (!rc && io->workspace_bytes < (base + critic_phase_fixed_floats(*d) + PROF_ITERS * critic_phase_floats_per_iter(*d)) * sizeof(float)) rc = HYPAD_EWORKSPACE;
    if (!rc) rc = launch_pack(ax, *d, (hipStream_t)s);      // the precompute reads the packed generator weights
    int persistent = 0;
    if (!rc) rc = run_critic_phase(ax, az, nullptr, PROF_ITERS, io->losses, (float*)io->workspace + base,
                                   io->workspace_bytes / sizeof(float) - base, d->n_signals, (hipStream_t)s, ev, nullptr, &persistent);
    if (!rc) kind4_div = persistent ? PROF_ITERS : PROF_ITERS - 1;
  }
  if (rc == HYPAD_OK) {
    hipError_t e = hipStreamSynchronize((hipStream_t)s);
    if (e != hipSuccess) rc = (int)e;
  }
  for (int i = 0; i < nk && rc == HYPAD_OK; ++i) {
    hipError_t e = hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
    if (e != hipSuccess) rc = (int)e;
  }
  if (rc == HYPAD_OK && kind4_div > 0) ms_out[2] /= (float)kind4_div;      // mean of the steady-state launches
  if (rc == HYPAD_OK && kind == 5) { ms_out[0] /= (float)GEN_REPS; ms_out[1] /= (float)GEN_REPS; }
  for (int i = 0; i <= nk; ++i) (void)hipEventDestroy(ev[i]);
  return rc;
}

size_t hypad_epoch_workspace_bytes(const hypad_dims* d, int n_batches, int n_critics) {
  if (check_dims(d) || n_batches <= 0 || n_critics < 0) return 0;
  int64_t n = (int64_t)n_batches * n_critics;
  if (n > 512) n = 512;
  const size_t base = epoch_base_floats(*d);
  if (!critic_phase_supported(*d) || n == 0) return (size_t)ws_floats_per_signal(*d) * d->n_signals * sizeof(float);      // per-minibatch launch groups only
  return (base + critic_phase_fixed_floats(*d) + (size_t)n * critic_phase_floats_per_iter(*d)) * sizeof(float);
}

namespace {
// Fork / join events of the generator phase's model groups (hypad_epoch_io.aux_streams): created once per process on first use.
// (Event creation is not a stream operation: legal while a stream is being captured.  One epoch call at a time per process uses them.)
struct GenFork { hipEvent_t forked; hipEvent_t joined[7]; };
GenFork* gen_fork_events() {
  static GenFork f;
  static int state = 0;                    // 0 not tried, 1 ready, -1 failed
  if (state == 0) {
    bool ok = hipEventCreateWithFlags(&f.forked, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 7 && ok; ++i) ok = hipEventCreateWithFlags(&f.joined[i], hipEventDisableTiming) == hipSuccess;
    state = ok ? 1 : -1;
  }
  return state == 1 ? &f : nullptr;
}
__global__ void advance_counters_kernel(int32_t* counters, int opt, int n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (counters[4] != 0) return;          // fail-stop (hypad_epoch_status): the launches before this one were no-ops
    counters[opt] += n; counters[3] += n;
  }
}
} // namespace
int hypad_train_epoch(const hypad_dims* d, const hypad_train_state* st, const hypad_epoch_io* io, hypad_stream_t s) {
  if (!io || !io->row_index || io->n_batches <= 0 || io->n_critics < 0) return HYPAD_EINVAL;
  int rc = check_dims(d);
  if (rc) return rc;
  const hypad_epoch_noise* nz = io->noise;
  const bool inj_masks = io->train_mode && nz && (nz->masks_cx || nz->masks_cz || nz->masks_gen);
  if (inj_masks && !(nz->masks_cx && nz->masks_cz && nz->masks_gen)) return HYPAD_EINVAL;      // all three planes or none
  IterCall c;
  c.x = io->x; c.x_sig_stride = io->x_signal_stride; c.x_row_stride = io->x_row_stride; c.z = nullptr; c.alpha = nullptr;
  c.ri_sig_stride = io->row_index_signal_stride;
  c.train_mode = io->train_mode; c.masks = nullptr; c.seed = io->seed;
  c.workspace = io->workspace; c.workspace_bytes = io->workspace_bytes;
  c.guard = 1;                                          // every launch of the epoch stops behind a resident critic launch that gave up
  c.flags = io->flags;
  const int iters = (2 * io->n_critics + 1) * io->n_batches;
  c.loss_sig_stride = (int64_t)iters * 4;
  int it = 0;
  const int64_t pass_rows = (int64_t)io->n_batches * d->batch;
  const size_t base = epoch_base_floats(*d);
  const size_t have = io->workspace_bytes / sizeof(float);
  const int64_t B = d->batch, L = d->latent_dim, S = d->signal_shape, ns = d->n_signals;
  const int64_t mk_cx = 12 * B * L + B * 2 * DEC_H, mk_cz = 6 * B * L, mk_gen = 6 * B * L + 2 * B * 2 * DEC_H;   // hypad_iter_io.drop layouts
  const bool hoisted = io->n_critics > 0 && critic_phase_supported(*d) &&
                       have >= base + critic_phase_fixed_floats(*d) + critic_phase_floats_per_iter(*d) && !(io->flags & HYPAD_EPOCH_PER_MINIBATCH);
  unsigned* zero_ptr = nullptr;                        // the resident critic launch's epoch words / flags: zeroed by the pack launch
  int zero_words = 0;
  bool zeroed = false;
  if (hoisted) critic_phase_zero_block(*d, (float*)io->workspace + base, have - base, io->n_critics * io->n_batches, &zero_ptr, &zero_words, io->flags);
  // the resident critic launch can give up (bounded waits): the state it began from is kept for hypad_epoch_restore
  float* snap = hoisted && critic_phase_persistent(*d, io->flags) ? epoch_snapshot_ptr(*d, io->workspace) : nullptr;
  {                                                    // packed generator weights: built once, then kept current by the dW kernel
    IterArgs ag;
    c.row_index = io->row_index; c.losses = io->losses;
    rc = fill_args(ag, d, st, c, 2);
    if (!rc) rc = launch_pack(ag, *d, (hipStream_t)s, zero_ptr, zero_words, &zeroed, false, snap);
    if (!rc) rc = launch_dw_items(ag, *d, false, (hipStream_t)s);      // (the epoch's generator steps leave the decay-only tensors to decay_steps_kernel)
    if (rc) return rc;
  }
  if (hoisted) {                                       // train.py:315-328, generator forwards hoisted (critic_fused.hip)
    IterArgs ax, az;
    c.row_index = io->row_index; c.losses = io->losses;
    c.masks = inj_masks ? nz->masks_cx : nullptr;      // (non-null selects the injected-mask mode; the planes travel in `nz`)
    rc = fill_args(ax, d, st, c, 0);
    c.masks = inj_masks ? nz->masks_cz : nullptr;
    if (!rc) rc = fill_args(az, d, st, c, 1);
    c.masks = nullptr;
    if (rc) return rc;
    const int n = io->n_critics * io->n_batches;
    if (io->enc_table && io->enc_table_rows <= 0) return HYPAD_EINVAL;
    rc = run_critic_phase(ax, az, io->row_index, n, io->losses, (float*)io->workspace + base, have - base, d->n_signals,
                          (hipStream_t)s, nullptr, nz, nullptr, zeroed ? zero_ptr : nullptr, io->flags, -1, io->enc_table, io->enc_table_rows);
    if (rc) return rc;
    it = 2 * n;
  } else {
    for (int k = 0; k < io->n_critics; ++k) {          // train.py:315-328, one launch group per minibatch
      for (int b = 0; b < io->n_batches; ++b) {
        const int64_t ci = (int64_t)k * io->n_batches + b;          // critic iteration index of the injected planes
        c.row_index = io->row_index + k * pass_rows + (int64_t)b * d->batch;
        c.losses = io->losses;             // (validated by fill_args; the pair writes to the two pointers below)
        IterCall cz = c;
        if (nz) {
          c.z = nz->z_cx ? nz->z_cx + ci * ns * B * L : nullptr; c.alpha = nz->alpha_cx ? nz->alpha_cx + ci * ns * B * S : nullptr;
          cz.z = nz->z_cz ? nz->z_cz + ci * ns * B * L : nullptr; cz.alpha = nz->alpha_cz ? nz->alpha_cz + ci * ns * B * L : nullptr;
          if (inj_masks) { c.masks = nz->masks_cx + ci * ns * mk_cx; cz.masks = nz->masks_cz + ci * ns * mk_cz; }
        }
        rc = run_critic_pair(d, st, c, io->losses + (int64_t)it * 4, io->losses + (int64_t)(it + 1) * 4, (hipStream_t)s, nullptr, &cz);
        it += 2;
        if (rc) return rc;
      }
    }
    c.z = nullptr; c.alpha = nullptr; c.masks = nullptr;
  }
  // train.py:347-352.  The models are independent of each other: with auxiliary streams they run in groups, one chain of
  // (generator launch, dW + Adam launch) x n_batches per group and stream, so a group's optimizer launch overlaps another group's
  // generator launch (a generator launch keeps 12 workgroups per model busy for ~37 us, the optimizer launch that follows is
  // latency- or bandwidth-bound: in lock step the chip idles through both).  Every launch then carries its own step number and
  // rng tick (IterArgs.step_add) and nobody advances the counters until the groups have joined.
  int groups = 1 + (io->aux_streams ? (io->n_aux_streams < 0 ? 0 : io->n_aux_streams > 7 ? 7 : io->n_aux_streams) : 0);
  if (groups > d->n_signals) groups = d->n_signals;
  if (io->n_batches < 1) groups = 1;
  GenFork* fork = nullptr;
  if (groups > 1) {
    fork = gen_fork_events();
    if (!fork) groups = 1;
  }
  if (groups > 1) {
    hipError_t e = hipEventRecord(fork->forked, (hipStream_t)s);
    for (int g = 1; g < groups && e == hipSuccess; ++g) e = hipStreamWaitEvent((hipStream_t)io->aux_streams[g - 1], fork->forked, 0);
    if (e != hipSuccess) return (int)e;
  }
  const int it_gen0 = it;
  for (int g = 0; g < groups; ++g) {
    const int s0 = (int)((int64_t)d->n_signals * g / groups), s1 = (int)((int64_t)d->n_signals * (g + 1) / groups);
    hipStream_t gs = g == 0 ? (hipStream_t)s : (hipStream_t)io->aux_streams[g - 1];
    for (int b = 0; b < io->n_batches; ++b) {
      c.row_index = io->row_index + io->n_critics * pass_rows + (int64_t)b * d->batch;
      c.losses = io->losses + (int64_t)(it_gen0 + b) * 4;
      if (nz) {
        c.z = nz->z_gen ? nz->z_gen + (int64_t)b * ns * B * L : nullptr;
        c.masks = inj_masks ? nz->masks_gen + (int64_t)b * ns * mk_gen : nullptr;
      }
      rc = run_gen(d, st, c, gs, nullptr, false, false, s0, s1 - s0, groups > 1 ? b : -1);
      if (rc) return rc;
    }
  }
  it = it_gen0 + io->n_batches;
  if (groups > 1) {
    hipError_t e = hipSuccess;
    for (int g = 1; g < groups && e == hipSuccess; ++g) {
      e = hipEventRecord(fork->joined[g - 1], (hipStream_t)io->aux_streams[g - 1]);
      if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)s, fork->joined[g - 1], 0);
    }
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(advance_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, st->counters, 2, io->n_batches);      // generator steps (counters[2]) and rng ticks
    HYPAD_CHECK_LAUNCH();
  }
  // W_hh and the f-gate rows: all of the epoch's weight-decay steps at once (see decay_table)
  c.losses = io->losses;
  return run_decay_steps(d, st, c, io->n_batches, (hipStream_t)s);
}

int hypad_critic_phase_persistent(const hypad_dims* d) {
  if (check_dims(d) || !critic_phase_supported(*d)) return 0;
  return critic_phase_persistent(*d) ? 1 : 0;
}

int hypad_critic_phase_producers(const hypad_dims* d, int n_iters) {
  if (check_dims(d)) return 0;
  return critic_phase_producers(*d, n_iters) ? 1 : 0;
}

int hypad_epoch_record_info(const hypad_dims* d, int n_batches, int n_critics, int critic, hypad_record_info* out) {
  if (check_dims(d) || n_batches <= 0 || n_critics <= 0 || !out) return HYPAD_EINVAL;
  const int64_t n = (int64_t)n_batches * n_critics;
  if (n > 512) return HYPAD_EUNSUPPORTED;
  int rc = critic_phase_record_info(*d, (int)n, critic, out);
  if (rc) return rc;
  out->offset_floats += (int64_t)epoch_base_floats(*d);
  return HYPAD_OK;
}

int hypad_epoch_status(const hypad_train_state* st, int* status_host, hypad_stream_t s) {
  if (!st || !st->counters || !status_host) return HYPAD_EINVAL;
  int32_t v = 0;
  hipError_t e = hipMemcpyAsync(&v, st->counters + 4, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)s);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)s);
  if (e != hipSuccess) return (int)e;
  *status_host = (int)v;
  return HYPAD_OK;
}

int hypad_epoch_restore(const hypad_dims* d, const hypad_train_state* st, void* workspace, size_t workspace_bytes, hypad_stream_t s) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!st || !st->counters || !st->params.cx || !st->params.cz || !st->exp_avg.cx || !st->exp_avg.cz || !st->exp_avg_sq.cx || !st->exp_avg_sq.cz)
    return HYPAD_EINVAL;
  if (!workspace || workspace_bytes < epoch_base_floats(*d) * sizeof(float)) return HYPAD_EWORKSPACE;
  IterArgs a{};
  a.P = st->params; a.M = st->exp_avg; a.V = st->exp_avg_sq; a.counters = st->counters;
  a.pcx = cx_layout(d->signal_shape, d->latent_dim).total; a.pcz = cz_layout(d->latent_dim).total;
  const int units = 3 * (a.pcx + a.pcz) / 4;
  hipLaunchKernelGGL(epoch_restore_kernel, dim3((units + 255) / 256, d->n_signals), dim3(256), 0, (hipStream_t)s, a, epoch_snapshot_ptr(*d, workspace),
                     d->n_signals);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2, float eps,
                    float wd, hypad_stream_t s) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return HYPAD_EINVAL;
  if (n == 0) return HYPAD_OK;
  int blocks = (int)((n + THREADS - 1) / THREADS);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(THREADS), 0, (hipStream_t)s, p, g, m, v, n, step, lr, b1, b2, eps, wd, 0,
                     (int64_t)0, 0);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_radam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t ball_off, int ball_dim, int step, float lr,
                     float b1, float b2, float eps, float wd, int stabilize, hypad_stream_t s) {
  if (!p || !g || !m || !v || n < 0 || step < 1 || ball_dim < 0 || ball_dim > 64 * MAX_EPL) return HYPAD_EINVAL;
  if (ball_dim > 0 && (ball_off < 0 || ball_off + ball_dim > n)) return HYPAD_EINVAL;
  if (n == 0) return HYPAD_OK;
  int blocks = (int)((n + THREADS - 1) / THREADS);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(THREADS), 0, (hipStream_t)s, p, g, m, v, n, step, lr, b1, b2, eps, wd, 1,
                     ball_off, ball_dim);
  HYPAD_CHECK_LAUNCH();
  if (ball_dim > 0) {
    hipLaunchKernelGGL(radam_ball_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, p + ball_off, g + ball_off, m + ball_off,
                       v + ball_off, ball_dim, step, lr, b1, b2, eps, wd, stabilize);
    HYPAD_CHECK_LAUNCH();
  }
  return HYPAD_OK;
}

}  // extern "C"
