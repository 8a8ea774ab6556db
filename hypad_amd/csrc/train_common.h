// Shared definitions of the training kernels (train_iters.hip, critic_fused.hip): workspace layouts, kernel arguments,
// LDS plans, Adam / Riemannian-Adam update rules.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hypad.h"
#include "critic_valu.h"
#include "nets.h"

namespace hypad {
namespace train {

constexpr int THREADS = 256;      // dW + Adam kernel
constexpr int TB = 512;           // row-tile kernels: 16 waves share one tile (the layers are latency-bound: more waves
                                  // = more weight tiles in flight per layer)

// ------------------------------------------------------------------------------------------------ workspace
struct CritWs {
  int in_right, act[4], left[5], dm[4], partial, total;
};
HD CritWs crit_ws(int B, int in_dim, int L, int nh) {
  CritWs w; int o = 0;
  w.in_right = o; o += pad4(3 * B * in_dim);
  for (int i = 0; i < 4; ++i) { w.act[i] = o; if (i < nh) o += pad4(3 * B * L); }
  for (int i = 0; i < 5; ++i) { w.left[i] = o; if (i < nh) o += pad4(3 * B * L); else if (i == nh) o += pad4(3 * B); }
  for (int i = 0; i < 4; ++i) { w.dm[i] = o; if (i < nh) o += pad4(B * L); }
  w.partial = o; o += pad4((B / 16) * 4);
  w.total = o;
  return w;
}
struct GenWs {
  int xg, enc_g, enc_g2, enc_h, zcat, a0, g0, h0d, mask, g1, h1, ecat, u, du, ballpart, dpre2, dg1, dg0, da0, dzenc, dgenc, partial, adamc, total;
};
HD GenWs gen_ws(int B, int S, int L) {
  GenWs w; int o = 0;
  // the encoder's operand rows come twice: rows [0, B) from chain R (gradient arriving through the decoder), rows [B, 2B)
  // from chain Z (gradient arriving through critic_z); the weight-gradient reduction runs over all 2B rows
  w.xg = o; o += pad4(2 * B * S);
  w.enc_g = o; o += pad4(B * 8 * ENC_H);
  w.enc_g2 = o; o += pad4(B * 8 * ENC_H);            // chain Z's own copy of the encoder gates
  w.enc_h = o; o += pad4(2 * B * 2 * ENC_H);
  w.zcat = o; o += pad4(2 * B * L);
  w.a0 = o; o += pad4(2 * B * DEC_D1);
  w.g0 = o; o += pad4(2 * B * 8 * DEC_H);
  w.h0d = o; o += pad4(2 * B * 2 * DEC_H);
  w.mask = o; o += pad4(2 * B * 2 * DEC_H);
  w.g1 = o; o += pad4(2 * B * 8 * DEC_H);
  w.h1 = o; o += pad4(2 * B * 2 * DEC_H);
  w.ecat = o; o += pad4(3 * B * S);
  w.u = o; o += pad4(3 * B * S);
  w.du = o; o += pad4(3 * B * S);
  w.ballpart = o; o += pad4(2 * (B / 16) * S);      // per (role, tile) column sums of the head-bias gradient rows
  w.dpre2 = o; o += pad4(2 * B * S);
  w.dg1 = o; o += pad4(2 * B * 6 * DEC_H);
  w.dg0 = o; o += pad4(2 * B * 6 * DEC_H);
  w.da0 = o; o += pad4(2 * B * DEC_D1);
  w.dzenc = o; o += pad4(2 * B * L);
  w.dgenc = o; o += pad4(2 * B * 6 * ENC_H);
  w.partial = o; o += pad4((B / 16) * 4);
  w.adamc = o; o += 4;                              // Adam bias corrections of this step {1 - b1^t, 1 - b2^t, sqrt(1 - b2^t)} (the launch's first model's workspace)
  w.total = o;
  return w;
}
// critic_x and critic_z iterations of one minibatch may run side by side (train.py:320-327 touch disjoint weights):
// their workspaces are disjoint, the generator's overlays both.
inline int64_t ws_cz_offset(const hypad_dims& d) { return crit_ws(d.batch, d.signal_shape, d.latent_dim, 4).total; }
// the MFMA-native packed copies of the generator's weights (layout.h GenPack) follow the iteration scratch
inline int64_t ws_pack_offset(const hypad_dims& d) {
  int64_t a = ws_cz_offset(d) + crit_ws(d.batch, d.latent_dim, d.latent_dim, 2).total;
  int64_t c = gen_ws(d.batch, d.signal_shape, d.latent_dim).total;
  return ((a > c ? a : c) + 63) & ~(int64_t)63;
}
// ... and the generator's dW + Adam work-item records follow the packed copies (train_iters.hip DwItem: 32 words per item, room for
// DW_ITEM_CAP items; the launches read signal 0's copy): what a wave of that launch needs to know, prepared once per pack launch
constexpr int DW_ITEM_WORDS = 32, DW_ITEM_CAP = 2048;
inline int64_t ws_items_offset(const hypad_dims& d) {
  return (ws_pack_offset(d) + gen_pack(d.signal_shape, d.latent_dim, d.hyperbolic).total + 63) & ~(int64_t)63;
}
inline int64_t ws_floats_per_signal(const hypad_dims& d) {
  return ws_items_offset(d) + (int64_t)DW_ITEM_WORDS * DW_ITEM_CAP;
}

// ------------------------------------------------------------------------------------------------ kernel arguments
struct IterArgs {
  int S, L, B, hyperbolic;
  hypad_nets P, M, V;
  int pe, pd, pcx, pcz;            // floats per signal in each arena
  int32_t* counters;
  const float* x; int64_t x_sig_stride; int64_t x_ld; const int32_t* row_index;   // x_ld: floats between window rows (S, or 1 = series view)
  const float* z; const float* alpha;
  int drop_mode;                   // 0 eval, 1 injected, 2 Philox
  const float* masks; int64_t mask_sig_stride;
  uint64_t seed;
  float* losses; int64_t loss_sig_stride;
  float* ws; int64_t ws_sig_stride;
  int64_t pk_off;                  // packed generator weights of signal s at ws + s * ws_sig_stride + pk_off
  float lr, b1, b2, eps, wd; int stabilize; int riemannian;
  int opt;                         // counters index of the optimizer stepped by this iteration
  int tick_owner;                  // 1: this iteration's dW kernel advances the rng tick (one owner per launch group)
  int sig0;                        // first model of this launch (blockIdx.y + sig0 = the model's index): the generator phase of an epoch may run
                                   // the models in groups, one chain of launches per group on a stream of its own (hypad_epoch_io.aux_streams)
  int step_add;                    // < 0: the launch reads the step / rng tick counters and advances them (one launch group at a time on one
                                   // stream); >= 0: it belongs to generator iteration `step_add` of an epoch -- step = counters[opt] + step_add + 1,
                                   // tick = counters[3] + step_add, nothing is advanced (one closing launch adds the epoch's iterations)
  int rng_sig0;                    // hypad_dims.first_signal: model `sig` draws its device random streams as stream rng_sig0 + sig
  int64_t ri_sig_stride;           // int32 elements between the models' row_index planes (0: one plane for all)
  int guard;                       // 1 (hypad_train_epoch): a launch is a no-op while counters[4] -- the resident critic launch's
                                   // status word -- is non-zero (fail-stop, see hypad_epoch_status in hypad.h)
  long long* stamps;               // development aid: shader-clock stamps [role][48 marks][8 waves] of workgroup (0, 0) of the generator kernel, or null
};
#if HYPAD_DIAG
#define GEN_STAMP(k) do { if (a.stamps && (threadIdx.x & 63) == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.stamps[(blockIdx.z * 48 + (k)) * 8 + (threadIdx.x >> 6)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEN_STAMP(k) do { } while (0)
#endif

struct LdsPlan {
  int xs, zs, bufA, bufB, crit, small, wst, cparams, total, ldS, bufFloats;
  int stage;      // 1: the critics' weights are staged in LDS (fits the 160 KiB budget)
};
constexpr int LDS_LIMIT_FLOATS = 160 * 1024 / 4;
HD LdsPlan lds_plan(int S, int rows_lstm, int rows_head, int critic_floats, int crit_scratch = CRITIC_LDS_FLOATS) {
  LdsPlan p;
  p.ldS = lds_stride(S);
  int a = rows_lstm * (6 * DEC_H + 4), b = rows_head * p.ldS;
  p.bufFloats = a > b ? a : b;
  int o = 0;
  p.xs = o; o += 16 * p.ldS;
  p.zs = o; o += 32 * LP;
  p.bufA = o; o += p.bufFloats;
  p.bufB = o; o += p.bufFloats;
  p.crit = o; o += crit_scratch;
  p.small = o; o += 3 * 16 * LP + 64;     // three (16, LP) scratch tiles + 64 reduction slots
  p.wst = o; o += (TB / 64) * WSTAGE_FLOATS;   // wave-private weight slabs of gemm_nt
  p.cparams = o;
  p.stage = (o + critic_floats <= LDS_LIMIT_FLOATS) ? 1 : 0;
  if (p.stage) o += critic_floats;
  p.total = o;
  return p;
}
// reduction slots inside `small`: [0,16) block_sum, [16,32) per-wave partials, [32,36) pass sums

__device__ __forceinline__ DropSrc drop_src(const IterArgs& a, int sig, const float* ptr, uint32_t stream, uint32_t tick, float p) {
  DropSrc s;
  s.mode = a.drop_mode; s.ptr = ptr; s.batch = a.B; s.seed = a.seed; s.tick = tick; s.stream = stream; s.sig = (uint32_t)(sig + a.rng_sig0); s.p = p;
  return s;
}

// block-wide sum of per-thread values; result valid in every thread.  red: LDS >= 16 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}


struct AdamCoef {
  float lr, b1, b2, eps, wd, bc1, bc2, sqrt_bc2;
  int riemannian, step, stabilize;
};
// Division and square root through v_rcp_f32 / v_sqrt_f32 (1 ulp each): the update differs from IEEE division by at most
// a few ulp of a step that is itself <= lr -- 1e-10 absolute, far below anything the parity tests (or fp32 training)
// resolve -- and the update sits on the critical path of every iteration (4x fewer instructions).
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, const AdamCoef& c) {
  if (c.riemannian) {     // oracle/radam.py, Euclidean branch
    g += c.wd * p;
    m = c.b1 * m + (1.f - c.b1) * g;
    v = c.b2 * v + (1.f - c.b2) * g * g;
    const float den = __builtin_amdgcn_sqrtf(v * __builtin_amdgcn_rcpf(c.bc2)) + c.eps;
    p -= c.lr * (m * __builtin_amdgcn_rcpf(c.bc1)) * __builtin_amdgcn_rcpf(den);
  } else {                // torch.optim.Adam (single-tensor rule)
    m = c.b1 * m + (1.f - c.b1) * g;
    v = c.b2 * v + (1.f - c.b2) * g * g;
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(c.sqrt_bc2) + c.eps;
    p -= (c.lr * __builtin_amdgcn_rcpf(c.bc1)) * (m * __builtin_amdgcn_rcpf(denom));
  }
}
// torch.optim.Adam's rule of adam_update with the two reciprocals that do not depend on the element -- 1 / sqrt(bc2) and lr / bc1 -- taken
// once by the caller (isb2 = rcp(c.sqrt_bc2), lrb1 = c.lr * rcp(c.bc1)): the same operations on the same values, so the same bits; the
// compiler did not hoist them out of the per-row branches itself (3 rcp + 1 sqrt per element, each a quarter-rate instruction).
__device__ __forceinline__ void adam_update_hoisted(float& p, float& m, float& v, float g, const AdamCoef& c, float isb2, float lrb1) {
  m = c.b1 * m + (1.f - c.b1) * g;
  v = c.b2 * v + (1.f - c.b2) * g * g;
  const float denom = __builtin_amdgcn_sqrtf(v) * isb2 + c.eps;
  p -= lrb1 * (m * __builtin_amdgcn_rcpf(denom));
}
__device__ __forceinline__ AdamCoef adam_coef(float lr, float b1, float b2, float eps, float wd, int riem, int stab, int step) {
  AdamCoef c;
  c.lr = lr; c.b1 = b1; c.b2 = b2; c.eps = eps; c.wd = wd; c.riemannian = riem; c.stabilize = stab; c.step = step;
  double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  c.bc1 = (float)bc1; c.bc2 = (float)bc2; c.sqrt_bc2 = (float)sqrt(bc2);
  return c;
}

// Riemannian Adam on one ball-valued vector held by a wave (oracle/manual.py radam_ball_step)
// (P, Mv, V: the rows at p, m, v -- loaded by the caller, so that they can be requested together with the gradient's parts)
__device__ __forceinline__ void radam_ball_wave(float* p, float* m, float* v, RowVec P, RowVec Mv, RowVec V, RowVec g, int dim, int lane, const AdamCoef& c) {
#pragma unroll
  for (int e = 0; e < MAX_EPL; ++e) g.v[e] += c.wd * P.v[e];
  float lam = 2.f / fmaxf(1.f - row_dot(P, P), MIN_NORM);
  // (per-row reciprocals and the hardware rcp / sqrt, as in adam_update: this item is the longest of the dW + Adam launch)
  const float ilam2 = 1.f / (lam * lam), ibc1 = 1.f / c.bc1, ibc2 = 1.f / c.bc2;
  RowVec rg;
#pragma unroll
  for (int e = 0; e < MAX_EPL; ++e) rg.v[e] = g.v[e] * ilam2;
  float inner = lam * lam * row_dot(rg, rg);
  RowVec np;
#pragma unroll
  for (int e = 0; e < MAX_EPL; ++e) {
    Mv.v[e] = c.b1 * Mv.v[e] + (1.f - c.b1) * rg.v[e];
    V.v[e] = c.b2 * V.v[e] + (1.f - c.b2) * inner;
    float den = __builtin_amdgcn_sqrtf(V.v[e] * ibc2) + c.eps;
    np.v[e] = P.v[e] - c.lr * (Mv.v[e] * ibc1) * __builtin_amdgcn_rcpf(den);
  }
  np = project_row(np);
  // parallel transport of the first moment: gyr[np, -p] m * lambda_p / lambda_np (math_.py:1738-1746, 656-676)
  RowVec nb;
#pragma unroll
  for (int e = 0; e < MAX_EPL; ++e) nb.v[e] = -P.v[e];
  float u2 = row_dot(np, np), v2 = row_dot(nb, nb), uv = row_dot(np, nb), uw = row_dot(np, Mv), vw = row_dot(nb, Mv);
  float ca = -uw * v2 + vw + 2.f * uv * vw;
  float cb = -vw * u2 - uw;
  float d = fmaxf(1.f + 2.f * uv + u2 * v2, MIN_NORM);
  float lam_n = 2.f / fmaxf(1.f - u2, MIN_NORM);
  const float id2 = 2.f / d, sc = lam / lam_n;
#pragma unroll
  for (int e = 0; e < MAX_EPL; ++e) Mv.v[e] = (Mv.v[e] + (ca * np.v[e] + cb * nb.v[e]) * id2) * sc;
  if (c.stabilize > 0 && c.step % c.stabilize == 0) np = project_row(np);
  row_store(p, np, dim, lane);
  row_store(m, Mv, dim, lane);
  row_store(v, V, dim, lane);
}
__device__ __forceinline__ void radam_ball_wave(float* p, float* m, float* v, RowVec g, int dim, int lane, const AdamCoef& c) {
  radam_ball_wave(p, m, v, row_load(p, dim, lane), row_load(m, dim, lane), row_load(v, dim, lane), g, dim, lane, c);
}

// critic_fused.hip: the critic phase of an epoch with the frozen generator's forwards hoisted out of the chain
bool critic_phase_supported(const hypad_dims& d);          // LDS / register budget of the iteration kernel
size_t critic_phase_fixed_floats(const hypad_dims& d);     // double-buffered optimiser state + gradient slabs
size_t critic_phase_floats_per_iter(const hypad_dims& d);  // precomputed records of one (critic_x || critic_z) iteration
int run_critic_phase(IterArgs ax, IterArgs az, const int32_t* row_index, int n_iters, float* losses, float* extra, size_t extra_floats,
                     int n_signals, hipStream_t s, hipEvent_t* ev, const hypad_epoch_noise* noise = nullptr, int* persistent_used = nullptr,
                     const unsigned* zeroed = nullptr, int flags = 0, int only = -1,       // only: 0 / 1 = critic_x / critic_z alone
                     float* enc_table = nullptr, int64_t enc_rows = 0);                    // hypad_epoch_io.enc_table
// the block the resident form needs zero in front of its first launch (null: none) -- the caller's previous launch may zero it
void critic_phase_zero_block(const hypad_dims& d, float* extra, size_t extra_floats, int n_iters, unsigned** ptr, int* words, int flags = 0);
bool critic_phase_producers(const hypad_dims& d, int n_iters);      // ... and that launch produces the records itself
bool critic_phase_persistent(const hypad_dims& d, int flags = 0);   // the phase runs as ONE resident launch (critic_persistent_kernel)
int critic_phase_record_info(const hypad_dims& d, int n_iters, int critic, hypad_record_info* out);

}  // namespace train
}  // namespace hypad
