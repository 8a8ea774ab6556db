// Critic phase of an epoch (train.py:306-328) restructured for the GPU.
//
// During the n_critics passes the generator is frozen (train.py:306-309), so nothing the phase updates feeds
// decoder(z_i), encoder(x_i), the noise, the interpolation weights or the dropout masks of ANY of its iterations.
// `critic_phase_precompute_kernel` hoists all of it out of the sequential chain: one wide launch (iterations x
// row tiles x signals x {critic_x side, critic_z side} workgroups -- 1 160 workgroups for 29 batches x 5 passes) writes,
// per iteration and 16-row chunk, a *record*: the real | fake | interpolated rows exactly as the critic's first layer
// wants them in LDS (zero padded, with the constant-one column that carries the biases) and the dropout scales of every
// layer and pass.
//
// What remains sequential is the critic arithmetic itself: ~0.9 MFLOP per iteration, a chain of ~14 dependent small
// matrix products.  `critic_iteration_kernel` gives each 16-row chunk of the minibatch to one workgroup (B / 16
// workgroups per critic, blockIdx.z picks critic_x / critic_z): forward of the 48 rows (3 passes), first backward,
// g = d out / d interpolated, the second-order chain (unscaled: it is linear in coef * g, and coef = 20 (||g|| - 1) /
// ||g|| needs the WHOLE batch), and the chunk's share of every weight gradient, all on
// v_mfma_f32_16x16x4_f32 out of LDS.  The shares go to a slab in HBM; the NEXT launch's prologue (every workgroup,
// redundantly and in a fixed order, so the replicas stay bit-identical) sums the slabs, forms coef, applies Adam and
// rebuilds the weights in LDS.  The kernel boundary is the only inter-workgroup synchronisation; state and slabs are
// double-buffered by iteration parity so no workgroup overwrites what a sibling of the same launch still reads.
//
// Biases ride in the matrix products: every layer input has a constant-one column at index K, the weight rows carry
// the bias there, and the bias gradient is column K of the weight-gradient tile.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/hypad.h"
#include "critic_mfma.h"
#include "train_common.h"

using namespace hypad;
using namespace hypad::train;

namespace {

constexpr int FT = 512;                      // threads of the critic iteration kernel (8 waves)
constexpr int NW = FT / 64;
typedef unsigned int u32x4_t __attribute__((vector_size(16)));
static_assert(NW == CM_NW, "critic_mfma.h deals tiles over 8 waves");
constexpr int MAXT = 5;                      // weight tiles per wave
constexpr int MAX_ROW4 = 4, MAX_MASK4 = 3;   // float4 record loads per thread (rows / masks)
constexpr int NITEM = 3;                     // accumulator quads per thread in the reduction prologue
constexpr int XS = 3;                        // log2 of the blockIdx.x stretch that co-locates a model's workgroups on one XCD (0: off; measured +2 %)

// Geometry of one critic inside the phase: padded lengths, record and slab sizes, weight-tile census
struct CritGeom {
  int in_dim, L, nh, params;
  int Kin, Lp, ldin, LQ, L4;
  int rec_rows4, rec_mask4, rec_floats;      // record = [48][Kin] rows, [nh][48][L4] dropout scales, then a tail of 32 floats:
                                             // Adam's {1 - beta1^t, sqrt(1 - beta2^t)} of the step the NEXT iteration's prologue applies, then zeros
  int tk0, tn, tkh, tiles0, tilesh, ntiles;
  int slab_floats;                           // [ntiles][2][256] accumulator images + 4 scalars
};
HD CritGeom crit_geom(int in_dim, int L, int nh, int params) {
  CritGeom g;
  g.in_dim = in_dim; g.L = L; g.nh = nh; g.params = params;
  g.Kin = up16(in_dim + 1); g.Lp = up16(L + 1);
  g.ldin = g.Kin + 4; g.LQ = g.Lp + 4; g.L4 = pad4(L);
  g.rec_rows4 = 12 * g.Kin; g.rec_mask4 = 12 * nh * g.L4;
  g.rec_floats = 4 * (g.rec_rows4 + g.rec_mask4) + 32;          // (a 128-byte tail: records stay cache-line aligned)
  g.tk0 = g.Kin >> 4; g.tn = (L + 15) >> 4; g.tkh = g.Lp >> 4;
  g.tiles0 = g.tn * g.tk0; g.tilesh = g.tn * g.tkh;
  g.ntiles = g.tiles0 + (nh - 1) * g.tilesh + g.tkh;
  g.slab_floats = g.ntiles * 512 + 4;
  return g;
}
HD CritGeom cx_geom(int S, int L) { return crit_geom(S, L, 4, cx_layout(S, L).total); }
HD CritGeom cz_geom(int L) { return crit_geom(L, L, 2, cz_layout(L).total); }
HD bool geom_supported(const CritGeom& g) {
  const int Q = (g.L + 3) >> 2;
  const int nitems = Q * (g.in_dim + 1) + (g.nh - 1) * Q * (g.L + 1) + g.L + 1;      // valid quads (critic_iteration_body)
  // (the layer-0 quads take NITEM - 1 register slots of all threads; the other layers' quads loop on waves 3-7)
  (void)nitems;
  return g.ntiles <= MAXT * NW && g.rec_rows4 <= MAX_ROW4 * FT && g.rec_mask4 <= MAX_MASK4 * FT && Q * (g.in_dim + 1) <= (NITEM - 1) * FT;
}

// LDS plan of the iteration kernel (floats).  Row strides are (multiple of 16) + 4: operand fetches are ds_read_b128
// (lane (i, q) supplies k = 16 g + 4 q + s to the s-th MFMA of k-group g -- the reduction index may be permuted as long
// as A and B agree), which needs 16-byte aligned rows and zero padding up to the next multiple of 16 columns.
struct IterLds {
  int in0;      // [48][ldin]  rows 0-15 real, 16-31 fake, 32-47 interpolated; after the first backward rows 32-47 hold g
  int act;      // [nh][48][LQ] layer outputs (+ ones column); rows 32-47 are overwritten by the second-order chain ep_li
  int dl;       // [nh+1][48][LQ] first-order deltas of every layer (dl[nh]: column 0 = d loss / d out)   (follows act)
  int dm;       // [nh][48][LQ] leaky'(pre) * dropout scale
  int w0;       // [L][ldin]      weights of layer 0, bias in column in_dim
  int wh;       // [nh-1][L][LQ]  hidden layers, bias in column L
  int wl;       // [LQ]           output layer, bias at index L
  int gram;     // [Lp][LQ]     G0 = W0 W0^T over the input columns (bias column excluded), zero-padded
  int whT;      // [nh-1][Lp][LQ] hidden weights transposed ([in feature][out feature], no bias), zero-padded: backward chain
  int red;      // [64]
  int total;
};
HD IterLds iter_lds(const CritGeom& g) {
  IterLds f; int o = 0;
  f.in0 = o; o += 48 * g.ldin;
  f.act = o; o += g.nh * 48 * g.LQ;
  f.dl = o; o += (g.nh + 1) * 48 * g.LQ;
  f.dm = o; o += g.nh * 48 * g.LQ;
  f.w0 = o; o += g.L * g.ldin;
  f.wh = o; o += (g.nh - 1) * g.L * g.LQ;
  f.wl = o; o += g.LQ;
  f.gram = o; o += g.Lp * g.LQ;
  f.whT = o; o += (g.nh - 1) * g.Lp * g.LQ;
  f.red = o; o += 64;
  f.total = o;
  return f;
}

struct PhaseArgs {
  float* rec_x;             // (n_signals, n_iters, B/16, cx record)
  float* rec_z;             // (n_signals, n_iters, B/16, cz record)
  float* state_x;           // (n_signals, 2, 3, cx params)   P | exp_avg | exp_avg_sq, double-buffered by iteration parity
  float* state_z;
  float* slab_x;            // (n_signals, 2, B/16, cx slab)
  float* slab_z;
  const int32_t* row_index; // (n_iters, B)
  float* losses;            // first loss row of this phase chunk (row 2 it: critic_x, 2 it + 1: critic_z)
  float* bias_corr;         // (2 critics, n_iters + 1, 2): Adam's {1 - beta1^t, sqrt(1 - beta2^t)} of the step applied by launch `it`
  int n_iters;
  int it;                   // iteration index of an iteration launch; == n_iters: finalise only
  long long* stamps;        // development aid: shader-clock stamps of workgroup (0, 0, z) (64 per critic), or null
  // injected randomness (hypad_epoch_noise; null = device Philox), already advanced to this phase chunk's first iteration:
  // (iteration, signal, per-iteration layout of hypad_iter_io)
  const float* inj_z_x; const float* inj_al_x; const float* inj_mk_x;
  const float* inj_z_z; const float* inj_al_z; const float* inj_mk_z;
  // exchange buffers of the persistent form (critic_persistent_kernel): see PersistPlan
  float* xslab_x; float* xslab_z;              // (n_signals, 2 parities, B/16, nitems x 4) merged gradient shares, compact valid quads
  unsigned long long* gran_x;                  // (n_signals, 2 parities, B/16, 4) granules {epoch << 32 | float bits}: sum g^2, sum real out, sum fake out
  unsigned long long* gran_z;
  unsigned* flags;                             // (2 critics, n_signals, B/16) epoch counters: value k = "slab of iteration k - 1 is published"
  unsigned* err;                               // first word: set when a bounded wait gave up
  // launch folding (persistent form): the precompute launch zeroes the [epoch words | error word | granules] block for the launch
  // that follows it (it was a memset node + its dependency gaps: ~8 us per epoch), and that launch advances the step counters and
  // the rng tick itself when it ends (it was a one-thread kernel: another ~8 us)
  unsigned* zero_ptr; int zero_words; int advance;
  // records produced INSIDE the resident launch (blockIdx.z >= 2 of critic_persistent_kernel): one word per record, set (write-
  // through, after the record's write-through stores have drained) by the workgroup that wrote it; null = the records come from
  // the precompute launch in front
  unsigned* rec_flags;                         // (2 critics, n_signals, n_iters, B/16)
  int n_signals;
  int clear_each;                              // 1: zero the activation / delta tiles after every iteration (round 2; HYPAD_CRITIC_CLEAR=1).  Every element
                                               // an iteration reads is written by that iteration first (the chains' epilogues store whole 16-column
                                               // groups incl. the ones column and zero padding, the record fills in0 / dm, d loss / d out is set per
                                               // iteration), so the sweep -- 62 KB of LDS stores, ~1 k cycles -- is only needed once, before the loop
  int xcd_stretch;                             // 1: the critics' workgroup ids are stretched by 8 (one critic's chunks on one XCD); 0: dealt in id order
  int only;                                    // -1: both critics; 0 / 1: critic_x / critic_z alone (the per-iteration entry points run one critic's
                                               // iteration as a one-iteration phase: launch grids carry that critic only)
  int fault_it;                                // tests (HYPAD_EPOCH_TEST_GIVE_UP_SHIFT): > 0 = critic_x chunk 0 of signal 0 behaves as if its
                                               // wait for the siblings' shares had timed out at that iteration
  // hypad_epoch_io.enc_table (ABI 6): encoder(x) of every window row, (n_signals, enc_rows, L), filled by encoder_table_kernel in
  // front of the phase -- the encoder is frozen through it (train.py:306-309) and every pass shuffles the same windows: critic_z's
  // records gather their 16 rows from it instead of running the encoder on them once per pass.  Null: they run it.
  const float* enc_table; int64_t enc_rows;
};
// (HYPAD_DIAG: development builds only -- libhypad_hip_dev.so, `python -m hypad_amd.build --dev`; the product library carries
// neither the stamps nor their setter)
#if HYPAD_DIAG
long long* g_stamps = nullptr;
#define STAMP(k) do { if (ph.stamps && ph.it == 1 && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) ph.stamps[blockIdx.z * 64 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// four consecutive uniforms of a stream: the numbers rng_uniform gives for idx = 4 group + e, from one Philox evaluation
__device__ __forceinline__ float4 rng_uniform4(uint64_t seed, uint32_t tick, uint32_t stream, uint32_t sig, uint32_t group) {
  Philox ph(seed);
  const uint4 r = ph(group, stream, tick, sig);
  return make_float4(u32_to_unit(r.x), u32_to_unit(r.y), u32_to_unit(r.z), u32_to_unit(r.w));
}

// ---------------------------------------------------------------------------------------------- precompute
// LDS of a precompute workgroup: the window rows, the latent rows and two layer buffers -- 28 KB at S = 100, so the
// register file, not LDS, bounds the workgroups per CU (the kernel is one wide launch of 1 160 workgroups per epoch).
struct PreLds { int xs, zs, bufA, bufB, total, ldS; };
HD PreLds pre_lds(int S) {
  PreLds p; int o = 0;
  p.ldS = lds_stride(S);
  const int a = 16 * (2 * DEC_H + 4), b = 16 * p.ldS;      // h tiles (the gate tiles never reach LDS: fused LSTM layers)
  const int buf = a > b ? a : b;
  p.xs = o; o += 16 * p.ldS;
  p.zs = o; o += 32 * LP;
  p.bufA = o; o += buf;
  p.bufB = o; o += buf;
  p.total = o;
  return p;
}
// Writes one record: rows real | fake | interpolated ([48][Kin], ones column at in_dim, zeros after) and the dropout
// scales [nh][48][L4] (pass order real, fake, interpolated).  real / fake: LDS tiles of 16 rows.
template <bool IS_X>
__device__ __forceinline__ void emit_record(const IterArgs& a, const CritGeom& g, float* rec, const float* real, int ldr, const float* fake,
                                            int ldf, int sig, int g0, uint32_t tick, float p_drop, const float* inj_alpha, const float* inj_masks) {
  // inj_alpha: this (iteration, signal)'s (B, in_dim) interpolation weights or null; inj_masks: its keep-scales, pass blocks of
  // [nh][B][L] in the order hypad_iter_io.drop states (critic_x: valid, fake, interpolated; critic_z: fake, valid, interpolated)
  const int in_dim = g.in_dim, Kin = g.Kin, L = g.L, nh = g.nh;
  tile_for(16, Kin, [&](int r, int c) {
    const float pad = c == in_dim ? 1.f : 0.f;
    rec[r * Kin + c] = c < in_dim ? real[r * ldr + c] : pad;
    rec[(16 + r) * Kin + c] = c < in_dim ? fake[r * ldf + c] : pad;
    if (c >= in_dim) rec[(32 + r) * Kin + c] = pad;
  });
  for (int gi = threadIdx.x; gi < 4 * in_dim; gi += blockDim.x) {      // interpolation (train.py:64-69 / 149-154)
    float4 al = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!inj_alpha) al = rng_uniform4(a.seed, tick, RS_ALPHA, (uint32_t)(sig + a.rng_sig0), (uint32_t)(g0 * in_dim) / 4 + gi);
    const float alv[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = 4 * gi + e;
      const int r = f / in_dim, c = f - r * in_dim;
      const float w = inj_alpha ? inj_alpha[(int64_t)(g0 + r) * in_dim + c] : alv[e];
      rec[(32 + r) * Kin + c] = w * real[r * ldr + c] + (1.f - w) * fake[r * ldf + c];
    }
  }
  float* rm = rec + 48 * Kin;
  const float keep = 1.f / (1.f - p_drop);
  const int per = 4 * L;                                               // groups per (pass, layer)
  for (int w = threadIdx.x; w < 3 * nh * per; w += blockDim.x) {
    const int pl = w / per, gi = w - pl * per;
    const int p = pl / nh, li = pl - p * nh;
    float v[4] = {1.f, 1.f, 1.f, 1.f};
    const int pass = p == 2 ? 2 : (IS_X ? p : 1 - p);                  // stream / mask-block numbering of the per-iteration entry points
    if (a.drop_mode == 1) {
      const float* mb = inj_masks + ((int64_t)(pass * nh + li) * a.B + g0) * L;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mb[4 * gi + e];               // (row r, column c) of the tile = flat element r * L + c
    } else if (a.drop_mode == 2) {
      const float4 uu = rng_uniform4(a.seed, tick, RS_DROP_CRITIC + 8 * pass + li, (uint32_t)(sig + a.rng_sig0), (uint32_t)(g0 * L) / 4 + gi);
      v[0] = uu.x >= p_drop ? keep : 0.f; v[1] = uu.y >= p_drop ? keep : 0.f;
      v[2] = uu.z >= p_drop ? keep : 0.f; v[3] = uu.w >= p_drop ? keep : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = 4 * gi + e;
      const int r = f / L, c = f - r * L;
      rm[(li * 48 + p * 16 + r) * g.L4 + c] = v[e];
    }
  }
}

// One (iteration, critic, signal, 16-row tile) of the critic phase's critic-independent work -> its record.  FUSED: the workgroup
// is a producer inside the resident launch -- the record is assembled in LDS, leaves as 16-byte write-through stores, and its
// flag word follows once every storing wave has drained (cdna_hip_programming.md Guideline 16 R1: the consumer loads it sc1).
// SC / LC: window length and latent width as compile-time constants (0 = read them from the arguments), as for the iteration
// kernels below: with run-time dimensions the tile functions' loops and index arithmetic cost 9.9 vector instructions per MFMA
// (SQ counters, profiles/r03_sq_counters.json), with the reference configuration folded in 2.9 (the scorers' forward).
template <bool FUSED, int SC = 0, int LC = 0>
__device__ __forceinline__ void precompute_body(const IterArgs& ax, const IterArgs& az, const PhaseArgs& ph, float* smem, int tile, int sig, int it,
                                                int role, int n_signals) {
  const int S = SC ? SC : ax.S, L = LC ? LC : ax.L, B = ax.B;
  const PreLds lp = pre_lds(S);
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  const uint32_t tick = (uint32_t)ax.counters[3] + (uint32_t)it;
  const int g0 = tile * 16, nchunks = B / 16;
  if (!FUSED && ph.zero_words) {                      // (see PhaseArgs: one word per thread, the first workgroups of the grid)
    const int64_t flat = (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * TB + threadIdx.x;
    if (flat < ph.zero_words) ph.zero_ptr[flat] = 0u;
  }
  // Adam bias corrections of the step that iteration it + 1's prologue applies (train.py:274-281): into the record's tail (the
  // resident launch reads them there) and, from one workgroup, into the table the per-iteration launches read
  float bc1 = 0.f, bc2s = 0.f;
  if (threadIdx.x == 0) {
    const IterArgs& c = role == 0 ? ax : az;
    const AdamCoef co = adam_coef(c.lr, c.b1, c.b2, c.eps, 0.f, 0, 0, c.counters[c.opt] + it + 1);
    bc1 = co.bc1; bc2s = co.sqrt_bc2;
    if (!FUSED && tile == 0 && sig == 0) {
      float* bc = ph.bias_corr + ((int64_t)role * (ph.n_iters + 1) + it + 1) * 2;
      bc[0] = co.bc1; bc[1] = co.sqrt_bc2;
    }
  }
  // where the record goes: straight to its place, or (FUSED) to an LDS image behind the layer buffers first
  float* stage = smem + ((lp.total + 3) & ~3);
  const CritGeom grec = role == 0 ? cx_geom(S, L) : cz_geom(L);
  float* rec_g = (role == 0 ? ph.rec_x : ph.rec_z) + (((int64_t)sig * ph.n_iters + it) * nchunks + tile) * grec.rec_floats;
  float* rec_w = FUSED ? stage : rec_g;
  if (threadIdx.x == 0) {
    float* tail = rec_w + 4 * (grec.rec_rows4 + grec.rec_mask4);
    tail[0] = bc1; tail[1] = bc2s;
  }
  if (threadIdx.x >= 2 && threadIdx.x < 32) rec_w[4 * (grec.rec_rows4 + grec.rec_mask4) + threadIdx.x] = 0.f;
  const int32_t* ridx = ph.row_index ? ph.row_index + (int64_t)sig * ax.ri_sig_stride + (int64_t)it * B : nullptr;
  const bool from_table = role == 1 && ph.enc_table != nullptr;
  if (!from_table) tile_load_rows(xs, lp.ldS, ax.x + sig * ax.x_sig_stride, ax.x_ld, ridx, g0, 16, S, 16);
  if (role == 0) {          // critic_x side: x_ = decoder(z), train-mode dropout (train.py:24-33)
    const DecLayout dl = dec_layout(S, L, ax.hyperbolic);
    const float* PD = ax.P.dec + (int64_t)sig * ax.pd;
    const int64_t isl = (int64_t)it * n_signals + sig;                 // (iteration, signal) slice of an injected plane
    const float* zin = ph.inj_z_x ? ph.inj_z_x + (isl * B + g0) * L : nullptr;
    tile_for(16, L, [&](int r, int c) { zs[r * LP + c] = zin ? zin[r * L + c] : rng_normal(ax.seed, tick, RS_Z, (uint32_t)(sig + ax.rng_sig0), (uint32_t)((g0 + r) * L + c)); });
    __syncthreads();
    const float* mk = ph.inj_mk_x ? ph.inj_mk_x + isl * ax.mask_sig_stride : nullptr;
    DropSrc ddrop = drop_src(ax, sig, mk ? mk + (int64_t)12 * B * L : nullptr, RS_DROP_DEC0, tick, 0.2f);
    DecSave none{16, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const float* pk = ax.ws + sig * ax.ws_sig_stride + ax.pk_off;       // packed generator weights (hypad_train_epoch builds them first)
    const GenPack gp = gen_pack(S, L, ax.hyperbolic);
    decoder_trunk_fwd_tile_packed<1>(zs, L, S, pk, gp, bufA, bufB, lp.ldS, ddrop, [g0](int r) { return g0 + r; }, none, 16);
    float* gen = bufA;
    if (ax.hyperbolic) {
      gemm_nt_packed<1>(bufA, lp.ldS, S, S, pk + gp.head, nullptr, bufB, lp.ldS, 0);
      __syncthreads();
      head_rows_tile(bufB, lp.ldS, 16, S, PD + dl.head_b);
      __syncthreads();
      gen = bufB;
    }
    const CritGeom g = cx_geom(S, L);
    emit_record<true>(ax, g, rec_w, xs, lp.ldS, gen, lp.ldS, sig, g0, tick, 0.25f, ph.inj_al_x ? ph.inj_al_x + isl * B * S : nullptr, mk);
  } else {                  // critic_z side: z_ = encoder(x), z ~ N(0, 1)  (train.py:111-116)
    const EncLayout el = enc_layout(S, L);
    const float* PE = az.P.enc + (int64_t)sig * az.pe;
    const int64_t isl = (int64_t)it * n_signals + sig;
    const float* zin = ph.inj_z_z ? ph.inj_z_z + (isl * B + g0) * L : nullptr;
    tile_for(16, L, [&](int r, int c) { zs[r * LP + c] = zin ? zin[r * L + c] : rng_normal(az.seed, tick, RS_Z, (uint32_t)(sig + az.rng_sig0), (uint32_t)((g0 + r) * L + c)); });
    __syncthreads();
    float* zenc = zs + 16 * LP;
    if (from_table) {       // the rows' encoder outputs, computed once per window in front of the phase (same function of the row: same bits)
      const float* tab = ph.enc_table + (int64_t)sig * ph.enc_rows * L;
      tile_for(16, L, [&](int r, int c) { zenc[r * LP + c] = tab[(int64_t)(ridx ? ridx[g0 + r] : g0 + r) * L + c]; });
    } else {
      const float* pk = az.ws + sig * az.ws_sig_stride + az.pk_off;
      encoder_fwd_tile_packed(xs, lp.ldS, S, L, pk, gen_pack(S, L, az.hyperbolic), bufA, ENC_LDG, bufB, ENC_LDH, zenc, nullptr, nullptr, 16);
    }
    __syncthreads();
    const CritGeom g = cz_geom(L);
    emit_record<false>(az, g, rec_w, zs, LP, zenc, LP, sig, g0, tick, 0.2f, ph.inj_al_z ? ph.inj_al_z + isl * B * L : nullptr,
                       ph.inj_mk_z ? ph.inj_mk_z + isl * az.mask_sig_stride : nullptr);
  }
  if constexpr (FUSED) {
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rec_g, 0, 0x7fffffff, 0x00020000);
    for (int i = threadIdx.x; i < grec.rec_floats / 4; i += TB) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * i);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, i * 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
    __syncthreads();                                      // ... before ONE lane signals for all of them
    if (threadIdx.x == 0)
      __hip_atomic_store(ph.rec_flags + ((((int64_t)role * n_signals + sig) * ph.n_iters + it) * nchunks + tile), 1u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
}
#ifndef HYPAD_PRE_WPE
#define HYPAD_PRE_WPE 4
#endif
template <int SC, int LC>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(HYPAD_PRE_WPE, HYPAD_PRE_WPE))) void critic_phase_precompute_kernel(IterArgs ax, IterArgs az, PhaseArgs ph) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  precompute_body<false, SC, LC>(ax, az, ph, smem, blockIdx.x, blockIdx.y, ph.only < 0 ? blockIdx.z >> 1 : blockIdx.z, ph.only < 0 ? blockIdx.z & 1 : ph.only,
                                 gridDim.y);
}
// encoder(x) of every window row of every model -> hypad_epoch_io.enc_table (PhaseArgs.enc_table): one workgroup per 16 rows
template <int SC, int LC>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(HYPAD_PRE_WPE, HYPAD_PRE_WPE))) void encoder_table_kernel(IterArgs az, float* table, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = SC ? SC : az.S, L = LC ? LC : az.L;
  const PreLds lp = pre_lds(S);
  float* xs = smem + lp.xs; float* zs = smem + lp.zs; float* bufA = smem + lp.bufA; float* bufB = smem + lp.bufB;
  const int sig = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)(rows - r0 < 16 ? rows - r0 : 16);
  tile_load(xs, lp.ldS, az.x + sig * az.x_sig_stride + r0 * az.x_ld, az.x_ld, 16, S, valid);
  __syncthreads();
  const float* pk = az.ws + sig * az.ws_sig_stride + az.pk_off;
  encoder_fwd_tile_packed(xs, lp.ldS, S, L, pk, gen_pack(S, L, az.hyperbolic), bufA, ENC_LDG, bufB, ENC_LDH, zs, nullptr, nullptr, 16);
  __syncthreads();
  float* out = table + ((int64_t)sig * rows + r0) * L;
  tile_for(16, L, [&](int r, int c) { if (r < valid) out[r * L + c] = zs[r * LP + c]; });
}
typedef void (*PreKernel)(IterArgs, IterArgs, PhaseArgs);
inline PreKernel precompute_kernel(int S, int L) {
  if (L == 20 && S == 100) return critic_phase_precompute_kernel<100, 20>;
  if (L == 20 && S == 123) return critic_phase_precompute_kernel<123, 20>;          // configs/multivariate.yaml: WADI
  if (L == 20 && S == 51) return critic_phase_precompute_kernel<51, 20>;            // ... SWAT
  return critic_phase_precompute_kernel<0, 0>;
}

// ---------------------------------------------------------------------------------------------- iteration kernel
// SC / LC / BC: window length, latent width and batch as compile-time constants (0 = read them from the arguments): the
// kernel runs every stage exactly once, so its index arithmetic is not amortised by any loop -- folding the strides and
// tile counts of the reference configuration (100, 20, 64) removes a third of its instructions.
template <bool IS_X, int SC, int LC, int BC>
__device__ __forceinline__ void critic_iteration_body(const IterArgs& a, const PhaseArgs& ph, float* smem) {
  const int L = LC ? LC : a.L, B = BC ? BC : a.B, S = SC ? SC : a.S;
  const int sig = blockIdx.y, chunk = blockIdx.x >> XS, nchunks = B / 16;
  constexpr int nh = IS_X ? 4 : 2;
  const CriticLayout cl = IS_X ? cx_layout(S, L) : cz_layout(L);
  const CritGeom g = IS_X ? cx_geom(S, L) : cz_geom(L);
  const IterLds fl = iter_lds(g);
  const int in_dim = g.in_dim, ldin = g.ldin, LQ = g.LQ, Kin = g.Kin, Lp = g.Lp;
  float* in0 = smem + fl.in0; float* act = smem + fl.act; float* dm = smem + fl.dm; float* dl = smem + fl.dl;
  float* w0 = smem + fl.w0; float* wh = smem + fl.wh; float* wl = smem + fl.wl; float* red = smem + fl.red;
  float* gram = smem + fl.gram; float* whT = smem + fl.whT;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int it = ph.it;
  const bool fin = it == ph.n_iters;
  const float invB = 1.f / B;
  STAMP(0);

  // ---- this iteration's record -> registers (HBM latency overlaps the reduction below); clear act | dl (padding)
  const float* rec = (IS_X ? ph.rec_x : ph.rec_z) + (((int64_t)sig * ph.n_iters + it) * nchunks + chunk) * g.rec_floats;
  float4 rrow[MAX_ROW4], rmask[MAX_MASK4];
  if (!fin) {
#pragma unroll
    for (int u = 0; u < MAX_ROW4; ++u) {
      const int i = threadIdx.x + u * FT;
      rrow[u] = i < g.rec_rows4 ? reinterpret_cast<const float4*>(rec)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < MAX_MASK4; ++u) {
      const int i = threadIdx.x + u * FT;
      rmask[u] = i < g.rec_mask4 ? reinterpret_cast<const float4*>(rec + 4 * g.rec_rows4)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = threadIdx.x * 4; i < (2 * nh + 1) * 48 * LQ; i += FT * 4)
      *reinterpret_cast<float4*>(act + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- prologue: weights of this iteration = Adam(previous weights, gradient of iteration it - 1 summed over the chunks)
  float* arena_p = (IS_X ? a.P.cx + (int64_t)sig * a.pcx : a.P.cz + (int64_t)sig * a.pcz);
  float* arena_m = (IS_X ? a.M.cx + (int64_t)sig * a.pcx : a.M.cz + (int64_t)sig * a.pcz);
  float* arena_v = (IS_X ? a.V.cx + (int64_t)sig * a.pcx : a.V.cz + (int64_t)sig * a.pcz);
  float* state = (IS_X ? ph.state_x : ph.state_z) + (int64_t)sig * 6 * g.params;
  float* slabs = (IS_X ? ph.slab_x : ph.slab_z) + (int64_t)sig * 2 * nchunks * g.slab_floats;
  const float* __restrict__ src_p = it == 0 ? arena_p : state + ((it - 1) & 1) * 3 * g.params;
  const float* __restrict__ src_m = it == 0 ? arena_m : src_p + g.params;
  const float* __restrict__ src_v = it == 0 ? arena_v : src_p + 2 * g.params;
  float* __restrict__ dst_p = fin ? arena_p : state + (it & 1) * 3 * g.params;
  float* __restrict__ dst_m = fin ? arena_m : dst_p + g.params;
  float* __restrict__ dst_v = fin ? arena_v : dst_p + 2 * g.params;
  const float* __restrict__ prev = slabs + (int64_t)((it - 1) & 1) * nchunks * g.slab_floats;
  const bool writer = chunk == 0;
  AdamCoef co;
  co.lr = a.lr; co.b1 = a.b1; co.b2 = a.b2; co.eps = a.eps; co.wd = 0.f; co.riemannian = 0; co.stabilize = 0; co.step = 0;
  {
    const float* bc = ph.bias_corr + ((int64_t)(IS_X ? 0 : 1) * (ph.n_iters + 1) + it) * 2;   // written by the precompute launch
    co.bc1 = it > 0 ? bc[0] : 1.f; co.sqrt_bc2 = it > 0 ? bc[1] : 1.f; co.bc2 = 1.f;
  }
  STAMP(50);

  auto tile_desc = [&](int t, int& li, int& n0, int& k0) __attribute__((always_inline)) {
    if (t < g.tiles0) { li = 0; n0 = (t / g.tk0) * 16; k0 = (t % g.tk0) * 16; return; }
    t -= g.tiles0;
    li = 1 + t / g.tilesh;
    t -= (li - 1) * g.tilesh;
    n0 = (t / g.tkh) * 16; k0 = (t % g.tkh) * 16;
  };
  // Reduction + Adam over *valid accumulator quads* -- a lane's 4 rows of one weight tile at a column that holds real
  // parameters (weight columns and the bias column; the tiles' padding columns are not visited: 841 quads instead of 1072
  // at S = 100, L = 20, two rounds of the 512 threads instead of three) -- dealt round-robin to the threads; every load is
  // issued before the first use (one HBM round trip for the previous launch's slabs and the optimiser state together):
  // clamped indices instead of branches around the loads, rounds beyond the last quad skipped block-wide.
  const int Q = (L + 3) >> 2;                              // valid quad rows of an L-row layer
  const int C0 = in_dim + 1, Ch = L + 1;                   // valid columns of layer 0 / of a hidden layer, bias column included
  const int I0 = Q * C0, Ih = Q * Ch, nitems = I0 + (nh - 1) * Ih + Ch;
  float4 sx[NITEM][4], sy[NITEM][4];
  float pv[NITEM][4], mv[NITEM][4], vv[NITEM][4];
  int off[NITEM][4], i_li[NITEM], i_n[NITEM], i_k[NITEM], i_so[NITEM];
  // the weight tiles' padding columns (beyond the bias column) are read by the products: zero them once, here
  for (int i = threadIdx.x; i < L * (ldin - C0); i += FT) { const int n = i / (ldin - C0), c = i - n * (ldin - C0); w0[n * ldin + C0 + c] = 0.f; }
  for (int i = threadIdx.x; i < (nh - 1) * Lp * LQ; i += FT) {          // everything outside the L x L blocks
    const int rr = i / LQ, k = i - rr * LQ, m = rr % Lp;
    if (m >= L || k >= L) whT[i] = 0.f;
  }
  for (int i = threadIdx.x; i < ((nh - 1) * L + 1) * (LQ - Ch); i += FT) {
    const int n = i / (LQ - Ch), c = i - n * (LQ - Ch);
    wh[n * LQ + Ch + c] = 0.f;                             // wl follows wh: row (nh - 1) L of this loop is wl
  }
  // the chunks' scalars {sum g^2, sum real out, sum fake out} are requested first: loads return in order, and the penalty
  // coefficient they give is needed by the first quad's update, which can then start while the later quads are still in flight
  float gsum = 0.f, sreal = 0.f, sfake = 0.f;
#pragma unroll 4
  for (int w = 0; w < nchunks; ++w) {                   // fixed order: every workgroup gets the same bits
    const float* sc = prev + (int64_t)w * g.slab_floats + g.ntiles * 512;
    gsum += sc[0]; sreal += sc[1]; sfake += sc[2];
  }
  // Two phases.  A: the quads of layer 0 (505 of the 841), dealt to all 512 threads -- the first chain layer needs nothing
  // else, so the chains start after ONE round.  B: the quads of the other layers, dealt to the 320 threads of waves 3-7,
  // which have ~10 k idle cycles while waves 0-2 carry the chains; their results (hidden weights, transposed copies, output
  // layer) are published through an LDS counter the chain waves check before the second layer.  All loads of A and of B's
  // first round are issued up front.
  constexpr int BW0 = 3, BT = (NW - BW0) * 64;                       // phase-B waves, threads
  const int NA = (I0 + FT - 1) / FT;                                  // phase-A rounds (1 at S = 100)
  const bool bthread = wave >= BW0;
  const int btid = threadIdx.x - BW0 * 64;
  auto issue = [&](int u, int e0, bool live) __attribute__((always_inline)) {
    const int e = live ? e0 : 0;
    const bool first = e < I0;
    const int e1 = e - I0;
    const int lh = first ? 0 : e1 / Ih;
    const bool last = !first && lh >= nh - 1;
    const int li = first ? 0 : (last ? nh : 1 + lh);
    const int er = first ? e : (last ? e1 - (nh - 1) * Ih : e1 - lh * Ih);
    const int tk = first ? g.tk0 : g.tkh, cols = first ? C0 : Ch;
    const int base = first ? 0 : g.tiles0 + (last ? nh - 1 : lh) * g.tilesh;
    const int qq = last ? 0 : er / cols, k = er - qq * cols;               // quad row, column (k == K: the bias)
    const int kt = k >> 4, jj = k & 15;
    const int t = base + (qq >> 2) * tk + kt;
    const int so = t * 512 + ((qq & 3) * 16 + jj) * 4;
    const int N = last ? 1 : L, K = first ? in_dim : L;                   // K = index of the bias column
    const int n = 4 * qq;
    // critic_layout (layout.h) in closed form (a table indexed by li becomes a scratch array)
    const int wof = li == 0 ? 0 : pad4(L * in_dim) + pad4(L) + (li - 1) * (pad4(L * L) + pad4(L));
    const int bof = li == 0 ? pad4(L * in_dim) : (li == nh ? wof + pad4(L) : wof + pad4(L * L));
    i_li[u] = live ? li : -1; i_n[u] = n; i_k[u] = k; i_so[u] = so;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* sl = prev + (int64_t)(w < nchunks ? w : 0) * g.slab_floats + so;
      sx[u][w] = *reinterpret_cast<const float4*>(sl); sy[u][w] = *reinterpret_cast<const float4*>(sl + 256);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = live && n + r < N;
      const int o = k < K ? wof + (n + r) * K + k : bof + n + r;
      off[u][r] = ok ? o : -1;
      const int oc = ok ? o : 0;
      pv[u][r] = src_p[oc]; mv[u][r] = src_m[oc]; vv[u][r] = src_v[oc];
    }
  };
  // register slots 0 .. NA-1: phase A; the remaining ones: phase B's first rounds (waves 3-7 only)
#pragma unroll
  for (int u = 0; u < NITEM; ++u) {            // straight-line code: selects, no branches, so all loads leave together
    if (u < NA) issue(u, threadIdx.x + u * FT, threadIdx.x + u * FT < I0);
    else if (bthread && u < NITEM && I0 + (u - NA) * BT < nitems) issue(u, I0 + (u - NA) * BT + btid, I0 + (u - NA) * BT + btid < nitems);
  }
  constexpr int UB = NITEM - 1;                // the slot any further phase-B rounds re-use
  float coef = 0.f;
  {
    const float nrm = sqrtf(gsum + 1e-12f);             // train.py:90, whole batch
    const float gp = (nrm - 1.f) * (nrm - 1.f);
    coef = it > 0 ? 20.f * (nrm - 1.f) / nrm : 0.f;     // d(10 gp) / d g = coef * g
    if (it > 0 && writer && threadIdx.x == 0) {
      float* lo = ph.losses + sig * a.loss_sig_stride + (int64_t)(2 * (it - 1) + (IS_X ? 0 : 1)) * 4;
      lo[0] = sfake * invB - sreal * invB + 10.f * gp;  // train.py:98-99
      lo[1] = gp; lo[2] = sreal * invB; lo[3] = sfake * invB;
    }
  }
  STAMP(51);
  const bool upd = it > 0;
  // a slot's quad: sum over the chunks in chunk order (chunks 0-3 came with the batch; batches above 64 rows bring more, four at
  // a time), Adam, state and LDS weight image
  auto finish = [&](int u) __attribute__((always_inline)) {
    f32x4 grf = {0.f, 0.f, 0.f, 0.f}, ggp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float on = w < nchunks ? 1.f : 0.f;
      grf[0] += on * sx[u][w].x; grf[1] += on * sx[u][w].y; grf[2] += on * sx[u][w].z; grf[3] += on * sx[u][w].w;
      ggp[0] += on * sy[u][w].x; ggp[1] += on * sy[u][w].y; ggp[2] += on * sy[u][w].z; ggp[3] += on * sy[u][w].w;
    }
    for (int w0c = 4; w0c < nchunks; w0c += 4) {
      float4 x[4], y[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float* sl = prev + (int64_t)(w0c + w < nchunks ? w0c + w : 0) * g.slab_floats + i_so[u];
        x[w] = *reinterpret_cast<const float4*>(sl); y[w] = *reinterpret_cast<const float4*>(sl + 256);
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float on = w0c + w < nchunks ? 1.f : 0.f;
        grf[0] += on * x[w].x; grf[1] += on * x[w].y; grf[2] += on * x[w].z; grf[3] += on * x[w].w;
        ggp[0] += on * y[w].x; ggp[1] += on * y[w].y; ggp[2] += on * y[w].z; ggp[3] += on * y[w].w;
      }
    }
    const int li = i_li[u];
    const int N = li == nh ? 1 : L;
    const int n = i_n[u], k = i_k[u];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = off[u][r];
      float p = pv[u][r], m = mv[u][r], v = vv[u][r];
      float pn = p, mn = m, vn = v;
      adam_update(pn, mn, vn, grf[r] + coef * ggp[r], co);
      p = upd ? pn : p; m = upd ? mn : m; v = upd ? vn : v;
      if (writer && o >= 0) { dst_p[o] = p; dst_m[o] = m; dst_v[o] = v; }
      if (li >= 0 && n + r < N) {
        const float pw = o >= 0 ? p : 0.f;
        float* wdst = li == 0 ? w0 + (n + r) * ldin + k : (li < nh ? wh + ((li - 1) * L + n + r) * LQ + k : wl + k);
        *wdst = pw;
        if (li > 0 && li < nh && k < L) whT[((li - 1) * Lp + k) * LQ + n + r] = pw;
      }
    }
  };
#pragma unroll
  for (int u = 0; u < NITEM; ++u)
    if (u < NA) finish(u);
  // phase B (waves 3-7): the slot loaded up front, then any further rounds one after the other (off the critical path)
  int* bdone = reinterpret_cast<int*>(red + 40);
  auto phase_b = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NITEM; ++u)
      if (u >= NA && I0 + (u - NA) * BT < nitems) finish(u);
    for (int eb = I0 + (NITEM - NA) * BT + btid; eb - btid < nitems; eb += BT) {     // (uniform trip count; none at S = 100)
      issue(UB, eb, eb < nitems);
      finish(UB);
    }
  };
  if (fin) { if (bthread) phase_b(); return; }
  STAMP(1);

  // ---- record -> LDS; constant d loss / d out (column 0 of dl[nh]: the left operand of the output layer's weight gradient)
#pragma unroll
  for (int u = 0; u < MAX_ROW4; ++u) {
    const int i = threadIdx.x + u * FT;
    if (i < g.rec_rows4) { const int r = i / (Kin / 4), c4 = i - r * (Kin / 4); *reinterpret_cast<float4*>(in0 + r * ldin + 4 * c4) = rrow[u]; }
  }
#pragma unroll
  for (int u = 0; u < MAX_MASK4; ++u) {
    const int i = threadIdx.x + u * FT;
    if (i < g.rec_mask4) { const int r = i / (g.L4 / 4), c4 = i - r * (g.L4 / 4); *reinterpret_cast<float4*>(dm + r * LQ + 4 * c4) = rmask[u]; }
  }
  if (threadIdx.x < 48) { const int p = threadIdx.x >> 4; dl[(nh * 48 + threadIdx.x) * LQ] = p == 0 ? -invB : p == 1 ? invB : 1.f; }
  if (threadIdx.x == 0) *bdone = 0;
  // LDS-only barrier: what crosses it is LDS data (record, layer-0 weights); __syncthreads() would also make waves 3-7 wait
  // for their phase-B loads (vmcnt(0)), which are still in flight and not needed yet
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  STAMP(2);
  int sk = 3;

  // ---- forward and first backward, 48 rows: three register-resident chains.  Wave p (0 real, 1 fake, 2 interpolated) carries
  // its 16 rows through every layer in the *transposed* form out^T = W in^T: the MFMA A operand is a weight row block
  // (ds_read_b128 from LDS, independent of the data), the B operand is the previous layer's output -- and the accumulator
  // layout of v_mfma_f32_16x16x4 (lane (j, q), register r = out^T[feature 4 q + r][row j]) is exactly the B layout the next
  // product wants for k = 16 g + 4 q + s.  So a layer's result never leaves the registers: no LDS round trip, no barrier, no
  // flag between dependent layers; activations, scales and deltas are written to LDS on the side, for the weight-gradient
  // tiles.  A 20-wide layer is 16 MFMAs (two feature tiles x two k-groups), ~0.6 k cycles; it was ~1.5 k as a block-wide stage.
  // Meanwhile waves 3 and 4 form G0 = W0 W0^T over the input columns: the second-order chain starts with
  // ep_0 = ((delta_0 W0) W0^T) * dm_0 = (delta_0 G0) * dm_0, a 20-wide product that does not wait for g = delta_0 W0 -- g is
  // still needed, for its norm and as a dW operand, but not on the critical path.
  constexpr int MF = 3, MAXNH = 4;                               // feature tiles of a layer (Lp <= 48), hidden layers
  const int NT = Lp >> 4;
  f32x4 DD[MAXNH][MF], DL[MF];                                   // leaky' * dropout scale of every layer; current delta^T
  const int myrow = 16 * wave + j;                               // the chain waves' batch row (of 48)
  auto as4 = [](const f32x4& v) __attribute__((always_inline)) { return make_float4(v[0], v[1], v[2], v[3]); };
  if (wave < 3) {
    f32x4 T[MF];                                                 // current activation^T tiles
#pragma unroll
    for (int li = 0; li < MAXNH; ++li)
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        DD[li][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (li < nh && t < NT) { const float4 v = *reinterpret_cast<const float4*>(dm + (li * 48 + myrow) * LQ + 16 * t + 4 * q); DD[li][t] = f32x4{v.x, v.y, v.z, v.w}; }
      }
    auto epilogue = [&](int li) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        if (t >= NT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * q + r;
          const float pre = T[t][r];
          const float dd = f < L ? leaky_slope(pre) * DD[li][t][r] : 0.f;
          DD[li][t][r] = dd;
          T[t][r] = f < L ? pre * dd : (f == L ? 1.f : 0.f);     // feature L: the ones row that carries the next layer's bias
        }
        *reinterpret_cast<float4*>(act + (li * 48 + myrow) * LQ + 16 * t + 4 * q) = as4(T[t]);
      }
    };
    // weight operands of the layer about to run; requested one layer ahead (they do not depend on the data), before the
    // current layer's epilogue, so that a layer starts with its A operands in registers
    float4 Aw[MF][MF];
    auto load_fwd = [&](int li) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < MF; ++t)
#pragma unroll
        for (int gg = 0; gg < MF; ++gg)
          if (t < NT && gg < NT) { const int m = 16 * t + j; Aw[t][gg] = *reinterpret_cast<const float4*>(wh + ((li - 1) * L + (m < L ? m : L - 1)) * LQ + 4 * q + 16 * gg); }
    };
    auto load_bwd = [&](int li) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < MF; ++t)
#pragma unroll
        for (int gg = 0; gg < MF; ++gg)
          if (t < NT && gg < NT) Aw[t][gg] = *reinterpret_cast<const float4*>(whT + (li * Lp + 16 * t + j) * LQ + 4 * q + 16 * gg);
    };
    // layer 0: the input rows come from LDS
#pragma unroll
    for (int t = 0; t < MF; ++t) {
      if (t >= NT) continue;
      const int m = 16 * t + j;
      const float* ap = w0 + (m < L ? m : L - 1) * ldin + 4 * q;
      const float* bp = in0 + myrow * ldin + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      for (int g16 = 0; g16 < Kin; g16 += 32) {
        acc = mfma4(*reinterpret_cast<const float4*>(ap + g16), *reinterpret_cast<const float4*>(bp + g16), acc);
        if (g16 + 16 < Kin) acc2 = mfma4(*reinterpret_cast<const float4*>(ap + g16 + 16), *reinterpret_cast<const float4*>(bp + g16 + 16), acc2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) T[t][r] = acc[r] + acc2[r];
    }
    STAMP(20);
    epilogue(0);
    STAMP(21);
    // the other layers' weights come from phase B
    while (__hip_atomic_load(bdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NW - BW0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (nh > 1) load_fwd(1);
    for (int li = 1; li < nh; ++li) {
      f32x4 N[MF];
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        if (t >= NT) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < MF; ++gg)
          if (gg < NT) acc = mfma4(Aw[t][gg], as4(T[gg]), acc);
        N[t] = acc;
      }
#pragma unroll
      for (int t = 0; t < MF; ++t) T[t] = N[t];
      if (li + 1 < nh) load_fwd(li + 1); else if (nh > 1) load_bwd(nh - 2);
#pragma unroll
      for (int x = 0; x < MAXNH; ++x) if (x == li) epilogue(x);   // (keeps DD's first index a compile-time constant)
      STAMP(21 + li);
    }
    // critic outputs (loss terms) and the top delta: d loss / d out = -1/B (real), +1/B (fake), 1 (interpolated: the penalty's
    // gradient is taken of the plain output)
    const float doutp = wave == 0 ? -invB : (wave == 1 ? invB : 1.f);
    float o = 0.f;
#pragma unroll
    for (int t = 0; t < MF; ++t) {
      DL[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t >= NT) continue;
      const float4 wv = *reinterpret_cast<const float4*>(wl + 16 * t + 4 * q);
      const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o += T[t][r] * wq[r];                                      // features past L are zero, feature L is 1 x bias
        float ddtop = 0.f;
#pragma unroll
        for (int x = 0; x < MAXNH; ++x) ddtop = x == nh - 1 ? DD[x][t][r] : ddtop;
        DL[t][r] = doutp * wq[r] * ddtop;                          // zero past L (ddtop is)
      }
      *reinterpret_cast<float4*>(dl + ((nh - 1) * 48 + myrow) * LQ + 16 * t + 4 * q) = as4(DL[t]);
    }
    STAMP(26);
    // first-order backward chain: delta_li^T = dm_li * (W_{li+1}^T delta_{li+1}^T), A operand from the transposed copies
    for (int li = nh - 2; li >= 0; --li) {
      f32x4 N[MF];
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        N[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t >= NT) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < MF; ++gg)
          if (gg < NT) acc = mfma4(Aw[t][gg], as4(DL[gg]), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float ddl = 0.f;
#pragma unroll
          for (int x = 0; x < MAXNH; ++x) ddl = x == li ? DD[x][t][r] : ddl;
          N[t][r] = acc[r] * ddl;
        }
      }
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        DL[t] = N[t];
        if (t < NT) *reinterpret_cast<float4*>(dl + (li * 48 + myrow) * LQ + 16 * t + 4 * q) = as4(DL[t]);
      }
      if (li > 0) load_bwd(li - 1);
      STAMP(27 + li);
    }
    // (off the chain: the passes' output sums)
    o += __shfl_xor(o, 16, 64);
    o += __shfl_xor(o, 32, 64);                                    // row j's output, in every lane of column j
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) o += __shfl_xor(o, off, 64);
    if (wave < 2 && lane == 0) red[32 + wave] = o;                 // sum over the real / the fake rows
  } else {
    phase_b();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(bdone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (wave == 3 || wave == 7) {                                    // (SIMD 3: no chain wave there)
    const int CTg = Lp >> 4;                                       // Gram tiles per side
    for (int t = wave == 3 ? 0 : 1; t < CTg * CTg; t += 2) {
      const int mt = t / CTg, nt = t - mt * CTg;
      const int m = mt * 16 + j, n = nt * 16 + j;
      const float* ap = w0 + (m < L ? m : L - 1) * ldin + 4 * q;
      const float* bp = w0 + (n < L ? n : L - 1) * ldin + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int g16 = 0; g16 < Kin; g16 += 16) {
        float4 av = *reinterpret_cast<const float4*>(ap + g16);
        const float4 bv = *reinterpret_cast<const float4*>(bp + g16);
        const int k = g16 + 4 * q;                                  // the bias sits in column in_dim: not part of W0
        av.x = k == in_dim ? 0.f : av.x; av.y = k + 1 == in_dim ? 0.f : av.y; av.z = k + 2 == in_dim ? 0.f : av.z; av.w = k + 3 == in_dim ? 0.f : av.w;
        acc = mfma4(av, bv, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = mt * 16 + 4 * q + r;
        gram[mm * LQ + n] = (mm < L && n < L) ? acc[r] : 0.f;
      }
    }
  }
  __syncthreads();
  STAMP(sk++);
  float gsq = 0.f;
  // ---- weight-gradient tiles: dW += left^T right over the chunk's rows; rows 0-31 (real, fake) -> acc_rf, GP rows -> acc_gp
  f32x4 acc_rf[MAXT], acc_gp[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) { acc_rf[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_gp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto dw_tile = [&](int i, bool rf, bool gpp) __attribute__((always_inline)) {
    const int t = wave + NW * i;
    if (t < g.ntiles) {
      int li, n0, k0;
      tile_desc(t, li, n0, k0);
      const int N = li == nh ? 1 : L;
      const int nj = n0 + j < N ? n0 + j : N - 1;                        // clamped; dropped at the update
      const float* left = dl + li * 48 * LQ + nj;
      const float* right = li == 0 ? in0 + k0 + j : act + (li - 1) * 48 * LQ + k0 + j;
      const int ldr = li == 0 ? ldin : LQ;
      if (rf) {
        float la[8], rb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { la[u] = left[(4 * u + q) * LQ]; rb[u] = right[(4 * u + q) * ldr]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc_rf[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc_rf[i], 0, 0, 0);
      }
      if (gpp) {
        float la[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { la[u] = left[(32 + 4 * u + q) * LQ]; rb[u] = right[(32 + 4 * u + q) * ldr]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc_gp[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc_gp[i], 0, 0, 0);
      }
    }
  };
  // ---- unscaled second-order chain on the interpolated rows' wave, still in registers: ep_0^T = dm_0 * (G0 delta_0^T),
  // ep_li^T = dm_li * (W_li ep_{li-1}^T)  -> act rows 32-47 (feature L stays zero: the GP rows carry no bias term).  Meanwhile
  // the other seven waves compute g = delta_0 W_0 on the interpolated rows (unscaled) -> in0 rows 32-47 (ones column cleared)
  // with its sum of squares, then the real / fake part of their weight-gradient tiles.
  if (wave == 2) {
    f32x4 E[MF];
    for (int li = 0; li < nh; ++li) {
      f32x4 N[MF];
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        N[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t >= NT) continue;
        const int m = 16 * t + j;
        const float* ap = (li == 0 ? gram + m * LQ : wh + ((li - 1) * L + (m < L ? m : L - 1)) * LQ) + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < MF; ++gg)
          if (gg < NT) acc = mfma4(*reinterpret_cast<const float4*>(ap + 16 * gg), li == 0 ? as4(DL[gg]) : as4(E[gg]), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float ddl = 0.f;
#pragma unroll
          for (int x = 0; x < MAXNH; ++x) ddl = x == li ? DD[x][t][r] : ddl;
          N[t][r] = acc[r] * ddl;                                  // zero from feature L on
        }
      }
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        E[t] = N[t];
        if (t < NT) *reinterpret_cast<float4*>(act + (li * 48 + 32 + j) * LQ + 16 * t + 4 * q) = as4(E[t]);
      }
    }
  } else {
    // g's column tiles over the seven other waves
    const int CT = (in_dim + 1 + 15) >> 4, slot = wave < 2 ? wave : wave - 1;
    const float* a = dl + (32 + j) * LQ + 4 * q;
    for (int ct = slot; ct < CT; ct += NW - 1) {
      int c = ct * 16 + j; c = c < in_dim ? c : in_dim - 1;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int g16 = 0; g16 < Lp; g16 += 16) {
        const int o = g16 + 4 * q;
        float4 bv;
        bv.x = w0[(o < L ? o : L - 1) * ldin + c];
        bv.y = w0[(o + 1 < L ? o + 1 : L - 1) * ldin + c];
        bv.z = w0[(o + 2 < L ? o + 2 : L - 1) * ldin + c];
        bv.w = w0[(o + 3 < L ? o + 3 : L - 1) * ldin + c];
        acc = mfma4(*reinterpret_cast<const float4*>(a + g16), bv, acc);
      }
      const int cc = ct * 16 + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        if (cc < in_dim) { in0[(32 + row) * ldin + cc] = acc[r]; gsq += acc[r] * acc[r]; }
        else if (cc == in_dim) in0[(32 + row) * ldin + cc] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) dw_tile(i, true, false);
  }
  __syncthreads();
  STAMP(sk++);
  if (wave == 2) {
#pragma unroll
    for (int i = 0; i < MAXT; ++i) dw_tile(i, true, true);          // the chain's wave does both parts of its tiles now
  } else {
#pragma unroll
    for (int i = 0; i < MAXT; ++i) dw_tile(i, false, true);
  }
  STAMP(sk++);

  // ---- publish the chunk's shares: accumulator images + {sum g^2, sum real out, sum fake out}
  float* mine = slabs + ((int64_t)(it & 1) * nchunks + chunk) * g.slab_floats;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + NW * i;
    if (t < g.ntiles) {
      *reinterpret_cast<float4*>(mine + t * 512 + lane * 4) = make_float4(acc_rf[i][0], acc_rf[i][1], acc_rf[i][2], acc_rf[i][3]);
      *reinterpret_cast<float4*>(mine + t * 512 + 256 + lane * 4) = make_float4(acc_gp[i][0], acc_gp[i][1], acc_gp[i][2], acc_gp[i][3]);
    }
  }
  const float tot = wave_sum(gsq);
  if (lane == 0) red[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    float gs = 0.f;
    for (int w = 0; w < NW; ++w) gs += red[w];
    float* sc = mine + g.ntiles * 512;
    sc[0] = gs; sc[1] = red[32]; sc[2] = red[33]; sc[3] = 0.f;
  }
  STAMP(sk++);
  STAMP(40);
}

// blockIdx.x is stretched by 8 and only the blocks dealt to XCD (signal mod 8) work (MI355X_MICROARCH.md, dispatch:
// round-robin over the XCDs): the workgroups of one model -- which all read the same slabs and optimiser state --
// share an L2.  A speed matter only.
template <int SC, int LC, int BC>
__global__ __launch_bounds__(FT) void critic_iteration_kernel(IterArgs ax, IterArgs az, PhaseArgs ph) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (XS && (blockIdx.x & 7) != (blockIdx.y & 7)) return;
  if ((ph.only < 0 ? (int)blockIdx.z : ph.only) == 0) critic_iteration_body<true, SC, LC, BC>(ax, ph, smem); else critic_iteration_body<false, SC, LC, BC>(az, ph, smem);
}


// ---------------------------------------------------------------------------------------------- persistent form
// The whole critic phase as ONE launch: every (signal, critic, 16-row chunk) workgroup stays resident for all iterations and
// keeps, across iterations, what the per-iteration launches reload each time:
//   * the critic's weights in LDS and its Adam state (p, exp_avg, exp_avg_sq) in registers -- each thread owns the same
//     accumulator quads every iteration (layer 0: one quad per thread; the other layers: waves 3-7, as in the prologue of
//     critic_iteration_body), so the optimiser state never travels (40 KB read + 40 KB written per launch before);
//   * the chunk's own share of nothing else: what crosses between the B / 16 workgroups of a critic per iteration is
//       - three 8-byte granules {epoch, value} per chunk (sum g^2, the two output sums), published as soon as g exists, so that
//         every chunk knows the whole-batch norm, hence the penalty's coefficient, BEFORE it publishes, and
//       - ONE merged gradient share per chunk, dW(real, fake) + coef * dW(penalty rows), as compact valid quads (13.5 KB for
//         critic_x; the launches exchanged 57 KB of padded tile images per chunk, two parts).
// Hand-off (cdna_hip_programming.md Guideline 16, R1 with sc1 loads; MI355X_MICROARCH.md, valid forms, first table row): the
// share is stored write-through (buffer_store ... sc1), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at
// a barrier, ONE lane stores the chunk's epoch word (sc1); a consumer polls the B / 16 epoch words with relaxed agent-scope
// loads from one wave, meets its workgroup at a barrier and reads the shares with buffer_load ... sc1 (every load of
// handed-off bytes is such a load).  Granules are R2: the data is the flag.  Shares and granules are double-buffered by
// iteration parity: a chunk can run at most one iteration ahead of its slowest sibling (it needs the sibling's share of
// iteration k to start k + 1), so parity k + 1's buffers are never rewritten while a sibling still reads parity k + 1 ... k.
// Every wait is bounded: on a timeout the workgroup sets ph.err, poisons its loss row with NaN and leaves; siblings follow.
// Residency: one workgroup per CU (the launcher asks for the full LDS plan for both critics and checks
// workgroups <= CUs before choosing this form; otherwise the per-iteration launches run).
#if HYPAD_DIAG
#define PSTAMP(k) do { if (ph.stamps && it == 5 && (threadIdx.x & 63) == 0 && chunk == 0 && sig == 0) ph.stamps[(((IS_X ? 0 : 1)) * 32 + (k)) * 8 + (threadIdx.x >> 6)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
#ifndef HYPAD_R6_BATCH
#define HYPAD_R6_BATCH 1
#endif
#ifndef HYPAD_R6_G
#define HYPAD_R6_G 1
#endif
#ifndef HYPAD_R6_STAGE
#define HYPAD_R6_STAGE 1
#endif
#ifndef HYPAD_R6_EOFF
#define HYPAD_R6_EOFF 1
#endif
#ifndef HYPAD_R6_TOFF
#define HYPAD_R6_TOFF 1
#endif
#ifndef HYPAD_R6_HOIST
#define HYPAD_R6_HOIST 1
#endif
#ifndef HYPAD_R6_FWD0P
#define HYPAD_R6_FWD0P 1
#endif
constexpr int PSLOT = 4;                     // Adam-state quads per thread
constexpr int MAXCH = 21;                    // chunks whose granules one wave sweeps in one pass (3 x 21 <= 64 lanes)
constexpr unsigned SPIN_LIMIT = 1u << 21;    // bounded waits: ~1 s of polling
// the 16-byte payload type of the raw-buffer builtins.  hipcc (ROCm 7.2) pitfall: indexing a result of
// __builtin_amdgcn_raw_buffer_load_b128 element by element (v[1], v[2] ...) is narrowed to ONE buffer_load_dword whose value
// stands for all four elements; bit-cast the whole vector to f32x4 first (and build stores the same way round).

// Adam-state quads ("slots") a thread of the resident kernel owns: layer 0's I0 quads take NA rounds of all 512 threads (phase A);
// at the compile-time shapes the last round's free threads take other-layer quads from the END of the list (`tail` of them) where
// that saves a slot; what is left of the other layers takes rounds of the five helper waves (phase B).  Window 150: 755 + 336 quads
// = 2 rounds + (336 - 269 = 67 quads ->) 1 round = 3 slots (4 before the tail rode along: 12 more state registers per thread, part of
// the 89 that instantiation spilled); window 100: 3 slots without a tail; critic_z: 2, never the kernel's register bound.
constexpr int persist_tail(int in_dim, int L, int nh) {
  const int Q = (L + 3) / 4, I0 = Q * (in_dim + 1), other = (nh - 1) * Q * (L + 1) + L + 1, NA = (I0 + FT - 1) / FT, BT = (NW - 3) * 64;
  const int tail = NA * FT - I0 < other ? NA * FT - I0 : other;
  return (nh == 4 && (other - tail + BT - 1) / BT < (other + BT - 1) / BT) ? tail : 0;
}
// A second way to shorten phase B: where its last round holds only a few quads (window 100: 336 = 320 + 16), the three CHAIN waves'
// first threads own those as one more phase-A slot -- requested with their layer-0 shares at the loop top, updated before the chains
// start (the chain body has 45 registers to spare; the quads are the last layer's, needed at the top of the forward chain only) --
// and the helper waves' phase B is ONE round of shares instead of two (the chains waited 1.8 k cycles for it).
constexpr int persist_ctail(int in_dim, int L, int nh) {
  const int Q = (L + 3) / 4, other = (nh - 1) * Q * (L + 1) + L + 1, BT = (NW - 3) * 64;
  const int left = other - persist_tail(in_dim, L, nh), rest = left % BT;
  return (nh == 4 && left > BT && rest > 0 && rest <= 64) ? rest : 0;
}
constexpr int persist_slots(int in_dim, int L, int nh) {
  const int Q = (L + 3) / 4, I0 = Q * (in_dim + 1), other = (nh - 1) * Q * (L + 1) + L + 1, NA = (I0 + FT - 1) / FT, BT = (NW - 3) * 64;
  const int b = (other - persist_tail(in_dim, L, nh) - persist_ctail(in_dim, L, nh) + BT - 1) / BT;
  return NA + (b > 0 ? b : (persist_ctail(in_dim, L, nh) > 0 ? 1 : 0));
}
HD bool persist_geom_supported(const CritGeom& g, int nchunks) {      // (run-time shapes: no tail, PSLOT slots)
  const int Q = (g.L + 3) >> 2;
  const int I0 = Q * (g.in_dim + 1), nitems = I0 + (g.nh - 1) * Q * (g.L + 1) + g.L + 1;
  const int NA = (I0 + FT - 1) / FT, BT = (NW - 3) * 64;
  return geom_supported(g) && g.ntiles <= MAXT * (NW - 1) && nchunks <= MAXCH && NA + (nitems - I0 + BT - 1) / BT <= PSLOT;
}
constexpr int persist_tiles(int in_dim, int L, int nh) {       // == crit_geom(...).ntiles
  return ((L + 15) / 16) * (((in_dim + 1 + 15) & ~15) / 16) + (nh - 1) * ((L + 15) / 16) * (((L + 1 + 15) & ~15) / 16) + ((L + 1 + 15) & ~15) / 16;
}
HD int persist_items(const CritGeom& g) {
  const int Q = (g.L + 3) >> 2;
  return Q * (g.in_dim + 1) + (g.nh - 1) * Q * (g.L + 1) + g.L + 1;
}

template <bool IS_X, int SC, int LC, int BC>
__device__ __forceinline__ void critic_persistent_body(const IterArgs& a, const PhaseArgs& ph, float* smem, const int sig, const int chunk,
                                                       const int n_signals) {
  const int L = LC ? LC : a.L, B = BC ? BC : a.B, S = SC ? SC : a.S;
  const int nchunks = B / 16;
  constexpr int nh = IS_X ? 4 : 2;
  // weight-gradient tiles per wave (seven waves share them): exact at the compile-time shapes, MAXT otherwise
  constexpr int TPW = (SC && LC) ? (persist_tiles(IS_X ? SC : LC, LC, nh) + NW - 2) / (NW - 1) : MAXT;
  // state quads per thread: window 100 needs 1 (layer 0) + 2 (the other layers over waves 3-7); critic_z fewer still
  constexpr int PS = (SC && LC) ? persist_slots(IS_X ? SC : LC, LC, nh) : (IS_X ? PSLOT : 2);
  const CriticLayout cl = IS_X ? cx_layout(S, L) : cz_layout(L);
  const CritGeom g = IS_X ? cx_geom(S, L) : cz_geom(L);
  const IterLds fl = iter_lds(g);
  const int in_dim = g.in_dim, ldin = g.ldin, LQ = g.LQ, Kin = g.Kin, Lp = g.Lp;
  float* in0 = smem + fl.in0; float* act = smem + fl.act; float* dm = smem + fl.dm; float* dl = smem + fl.dl;
  float* w0 = smem + fl.w0; float* wh = smem + fl.wh; float* wl = smem + fl.wl; float* red = smem + fl.red;
  float* gram = smem + fl.gram; float* whT = smem + fl.whT;
  float* xsc = smem + fl.total;                            // [3][MAXCH] the chunks' scalars of this iteration (+ 4 control words)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int n_iters = ph.n_iters;
  const float invB = 1.f / B;

  float* arena_p = (IS_X ? a.P.cx + (int64_t)sig * a.pcx : a.P.cz + (int64_t)sig * a.pcz);
  float* arena_m = (IS_X ? a.M.cx + (int64_t)sig * a.pcx : a.M.cz + (int64_t)sig * a.pcz);
  float* arena_v = (IS_X ? a.V.cx + (int64_t)sig * a.pcx : a.V.cz + (int64_t)sig * a.pcz);
  const bool writer = chunk == 0;
  AdamCoef co;
  co.lr = a.lr; co.b1 = a.b1; co.b2 = a.b2; co.eps = a.eps; co.wd = 0.f; co.riemannian = 0; co.stabilize = 0; co.step = 0;
  co.bc1 = 1.f; co.sqrt_bc2 = 1.f; co.bc2 = 1.f;

  const int Q = (L + 3) >> 2;
  const int C0 = in_dim + 1, Ch = L + 1;
  const int I0 = Q * C0, Ih = Q * Ch, nitems = I0 + (nh - 1) * Ih + Ch;
  constexpr int BW0 = 3, BT = (NW - BW0) * 64;
  const int NA = (I0 + FT - 1) / FT;
  constexpr int TAILI = (SC && LC) ? persist_tail(IS_X ? SC : LC, LC, nh) : 0;      // other-layer quads (the list's last ones) in phase A's free threads
  constexpr int CTAIL = (SC && LC) ? persist_ctail(IS_X ? SC : LC, LC, nh) : 0;     // ... and in slot NA of the chain waves' first threads (persist_ctail)
  const bool bthread = wave >= BW0;
  const int btid = threadIdx.x - BW0 * 64;
  const int slabf = nitems * 4;
  float* xslab = (IS_X ? ph.xslab_x : ph.xslab_z) + (int64_t)sig * 2 * nchunks * slabf;
  unsigned long long* gran = (IS_X ? ph.gran_x : ph.gran_z) + (int64_t)sig * 2 * nchunks * 4;
  unsigned* flags = ph.flags + ((int64_t)(IS_X ? 0 : 1) * n_signals + sig) * nchunks;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xslab, 0, 0x7fffffff, 0x00020000);
  float* lo_base = ph.losses + sig * a.loss_sig_stride + (IS_X ? 0 : 4);
  // Which XCD this workgroup runs on (MI355X_MICROARCH.md, dispatch: a hardware register, not an assumption about placement).  It
  // travels in the top byte of the chunk's epoch word; once a chunk has seen that ALL chunks of its critic share its XCD -- the
  // launcher deals them that way, see critic_persistent_kernel -- their exchange stays inside that XCD's L2: shares, granules and
  // epoch words are stored WITHOUT the write-through bit (the line stays in L2; every sibling reads it there with the sc1 --
  // L1-bypassing -- loads it uses anyway) instead of being pushed to the memory side and fetched back from there by every
  // sibling.  Any other placement keeps the write-through forms: correctness never depends on where a workgroup landed.
  const unsigned my_xcc = (__builtin_amdgcn_s_getreg(6164) & 0xfu) + 1u;            // hwreg(HW_REG_XCC_ID, 0, 4) + 1
  bool same_xcd = false;                                                            // decided at the first wait (iteration 1)

  auto tile_desc = [&](int t, int& li, int& n0, int& k0) __attribute__((always_inline)) {
    if (t < g.tiles0) { li = 0; n0 = (t / g.tk0) * 16; k0 = (t % g.tk0) * 16; return; }
    t -= g.tiles0;
    li = 1 + t / g.tilesh;
    t -= (li - 1) * g.tilesh;
    n0 = (t / g.tkh) * 16; k0 = (t % g.tkh) * 16;
  };

  // ---- once: zero the weight images' padding, take this thread's quads of the optimiser state into registers
  for (int i = threadIdx.x; i < L * (ldin - C0); i += FT) { const int n = i / (ldin - C0), c = i - n * (ldin - C0); w0[n * ldin + C0 + c] = 0.f; }
  for (int i = threadIdx.x; i < (nh - 1) * Lp * LQ; i += FT) {
    const int rr = i / LQ, k = i - rr * LQ, m = rr % Lp;
    if (m >= L || k >= L) whT[i] = 0.f;
  }
  for (int i = threadIdx.x; i < ((nh - 1) * L + 1) * (LQ - Ch); i += FT) {
    const int n = i / (LQ - Ch), c = i - n * (LQ - Ch);
    wh[n * LQ + Ch + c] = 0.f;
  }
  int i_li[PS], i_n[PS], i_k[PS], i_e[PS];
  float pv[PS][4], mv[PS][4], vv[PS][4];
  // arena offset of row r of slot u (or -1); recomputed where needed (setup, final write) instead of held in registers
  auto arena_off = [&](int u, int r) __attribute__((always_inline)) {
    const int li = i_li[u];
    if (li < 0) return -1;
    const int N = li == nh ? 1 : L, K = li == 0 ? in_dim : L;
    if (i_n[u] + r >= N) return -1;
    // critic_layout (layout.h) in closed form -- a table indexed by li became a scratch array (64 bytes of private segment per lane)
    const int wof = li == 0 ? 0 : pad4(L * in_dim) + pad4(L) + (li - 1) * (pad4(L * L) + pad4(L));
    const int bof = li == 0 ? pad4(L * in_dim) : (li == nh ? wof + pad4(L) : wof + pad4(L * L));
    return i_k[u] < K ? wof + (i_n[u] + r) * K + i_k[u] : bof + i_n[u] + r;
  };
#pragma unroll
  for (int u = 0; u < PS; ++u) {
    int e = -1;
    if (u < NA) {
      const int e0 = threadIdx.x + u * FT;
      if constexpr (TAILI == 0) e = e0 < I0 ? e0 : -1;
      else e = e0 < I0 ? e0 : (e0 - I0 < TAILI ? nitems - TAILI + (e0 - I0) : -1);
    } else if (bthread) {
      const int e0 = I0 + (u - NA) * BT + btid;
      if constexpr (TAILI == 0 && CTAIL == 0) e = e0 < nitems ? e0 : -1;
      else e = e0 < nitems - TAILI - CTAIL ? e0 : -1;
    } else if (CTAIL > 0 && u == NA) {
      e = (int)threadIdx.x < CTAIL ? nitems - TAILI - CTAIL + (int)threadIdx.x : -1;      // chain waves: the quads phase B's last round would hold
    }
    i_e[u] = e;
    const int ee = e < 0 ? 0 : e;
    const bool first = ee < I0;
    const int e1 = ee - I0;
    const int lh = first ? 0 : e1 / Ih;
    const bool last = !first && lh >= nh - 1;
    const int li = first ? 0 : (last ? nh : 1 + lh);
    const int er = first ? ee : (last ? e1 - (nh - 1) * Ih : e1 - lh * Ih);
    const int cols = first ? C0 : Ch;
    const int qq = last ? 0 : er / cols, k = er - qq * cols;
    i_li[u] = e < 0 ? -1 : li; i_n[u] = 4 * qq; i_k[u] = k;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = arena_off(u, r);
      pv[u][r] = o >= 0 ? arena_p[o] : 0.f; mv[u][r] = o >= 0 ? arena_m[o] : 0.f; vv[u][r] = o >= 0 ? arena_v[o] : 0.f;
    }
  }
  if (threadIdx.x < 4) reinterpret_cast<int*>(xsc)[3 * MAXCH + threadIdx.x] = 0;      // control words: [0] "give up"
  auto clear_tiles = [&]() __attribute__((always_inline)) {         // act | dl: zero padding columns, fresh accumulation targets
    for (int i = threadIdx.x * 4; i < (2 * nh + 1) * 48 * LQ; i += FT * 4)
      *reinterpret_cast<float4*>(act + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  clear_tiles();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (the first iteration writes d loss / d out into the cleared tiles)

  // record of iteration `it` in registers (requested one hand-off ahead)
  int* ctl = reinterpret_cast<int*>(xsc) + 3 * MAXCH;
  auto give_up = [&](unsigned code) __attribute__((always_inline)) {   // called by one lane: tell the workgroup and the siblings
    __hip_atomic_store(ph.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.guard) {                                                     // ... and the host: the first code sticks (hypad_epoch_status)
      int expected = 0;
      __hip_atomic_compare_exchange_strong(a.counters + 4, &expected, (int)code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ctl[0] = 1;
  };
  // Records come from the precompute launch in front of this one, or (ph.rec_flags) from producer workgroups of THIS launch
  // (blockIdx.z >= 2): then one wave waits for the record's flag word -- the producers run far ahead of the critics: a poll
  // that finds it set costs one memory round trip on a wave with slack -- and the workgroup loads behind a barrier that wave
  // joins.  Every record load is a 16-byte sc1 buffer load either way (Guideline 16 R1).
  auto await_record = [&](int it) __attribute__((always_inline)) {     // one wave calls it; a workgroup barrier follows
    if (!ph.rec_flags) return;
    const unsigned* f = ph.rec_flags + ((((int64_t)(IS_X ? 0 : 1) * n_signals + sig) * n_iters + it) * nchunks + chunk);
    bool ok = false;
    for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
      ok = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      if (ok) break;
      if ((spins & 1023) == 1023 && __hip_atomic_load(ph.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok && lane == 0) give_up(0x300u + (unsigned)it);
  };
  // (record registers stay in the load's own type: the LDS image is written from them as they are.  Lanes past the record's
  // rows / scales point outside the descriptor's range -- 1 GB: run_critic_phase keeps the records below it -- and get zeros
  // without a memory access.)
  u32x4_t rrow[MAX_ROW4], rmask[MAX_MASK4];
  const __amdgpu_buffer_rsrc_t rrs1 = __builtin_amdgcn_make_buffer_rsrc(IS_X ? ph.rec_x : ph.rec_z, 0, 0x40000000, 0x00020000);
  auto request_record = [&](int it) __attribute__((always_inline)) {
    const int rec_off = (int)(((((int64_t)sig * n_iters + it) * nchunks + chunk) * g.rec_floats) * 4);       // bytes (< 2^31: run_critic_phase)
#pragma unroll
    for (int u = 0; u < MAX_ROW4; ++u) {
      const int i = threadIdx.x + u * FT;
      rrow[u] = __builtin_amdgcn_raw_buffer_load_b128(rrs1, i < g.rec_rows4 ? i * 16 : 0x7ffffff0, rec_off, 16);
    }
#pragma unroll
    for (int u = 0; u < MAX_MASK4; ++u) {
      const int i = threadIdx.x + u * FT;
      rmask[u] = __builtin_amdgcn_raw_buffer_load_b128(rrs1, i < g.rec_mask4 ? (g.rec_rows4 + i) * 16 : 0x7ffffff0, rec_off, 16);
    }
  };
  if (n_iters > 0) {
    if (wave == 0) await_record(0);
    __syncthreads();
    if (ctl[0]) { if (writer && threadIdx.x == 0) lo_base[0] = __builtin_nanf(""); return; }
    request_record(0);
  }

#if HYPAD_R6_EOFF
  // Byte offset of this lane's quad of every tile it owns inside a merged share (or an offset past the descriptor's range: the hardware
  // drops that store -- lanes outside the tile's valid quads, tiles past the list, wave 2).  Once per launch: inside the loop the tile
  // arithmetic, its exec masks and the spilled scalars behind them cost ~400 cycles per tile and iteration (round 6, shader-clock stamps).
  int eoff[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = (wave < 2 ? wave : wave - 1) + (NW - 1) * i;
    int li, n0, k0;
    tile_desc(t < g.ntiles ? t : 0, li, n0, k0);
    const int N = li == nh ? 1 : L, K = li == 0 ? in_dim : L;
    const int n = n0 + 4 * q, k = k0 + j, qq = n >> 2;
    const int e = li == 0 ? qq * C0 + k : (li < nh ? I0 + (li - 1) * Ih + qq * Ch + k : I0 + (nh - 1) * Ih + k);
    eoff[i] = (t < g.ntiles && wave != 2 && n < N && k <= K) ? e * 16 : (int)0x80000000u;      // (past the 2 GB range whatever is added)
  }
#endif
  // ... and the LDS offsets (floats from `smem`) of its operand rows in every tile: left = the layer's deltas at column n0 + j, right = the
  // layer's input rows at column k0 + j (row 0 of the tile; the k-steps add row strides).  Windows up to 100 only: eight more registers
  // make the window-123 build spill 4 (it is 1.3 % faster with them all the same) and cost the window-150 build 5 % (41 spilled).
  constexpr bool TOFF = HYPAD_R6_TOFF && SC != 0 && SC <= 100;
  int lofs[TOFF ? TPW : 1], rofs[TOFF ? TPW : 1];
#pragma unroll
  for (int i = 0; i < (TOFF ? TPW : 0); ++i) {
    const int t = (wave < 2 ? wave : wave - 1) + (NW - 1) * i;
    int li, n0, k0;
    tile_desc(t < g.ntiles ? t : 0, li, n0, k0);
    const int N = li == nh ? 1 : L;
    const int nj = n0 + j < N ? n0 + j : N - 1;
    lofs[i] = fl.dl + li * 48 * LQ + nj;
    rofs[i] = li == 0 ? fl.in0 + k0 + j : fl.act + (li - 1) * 48 * LQ + k0 + j;
  }
  const int j_ = j, q_ = q, lane_ = lane;
  // Wave specialisation: the loop is instantiated twice -- for the three chain waves (which carry the register-resident
  // forward / backward chains and, of the optimiser state, only their layer-0 quads) and for the five helper waves (which carry
  // the other layers' state, the Gram matrix and no chain registers).  Both instantiations meet at the same barriers; the
  // register allocation is the larger of the two, not their sum (it spilled 110 registers as one body).
  auto run = [&](auto chain_tag) __attribute__((always_inline)) {
  constexpr bool CHAIN = decltype(chain_tag)::value;
  for (int it = 0;; ++it) {
    // (the loop body runs ~30 k cycles: nothing is gained by hoisting its address arithmetic out of the loop, and the hoisted
    // values would occupy registers across it -- 260 spilled registers before this; re-derive the lane indices per iteration)
    int j = j_, q = q_, lane = lane_;
    asm volatile("" : "+v"(j), "+v"(q), "+v"(lane));
    PSTAMP(0);
    const bool fin = it == n_iters;
    const int ppar = (it - 1) & 1;
    // layer-0 slots whose first four shares are requested at the loop top: all of them, except at window 150 (two slots: the second
    // one's sixteen registers were spilled; it loads inside finish() like the other layers' quads)
    constexpr int NPRE_C = (SC == 150) ? 1 : PS;
    const int NA_BODY = NA + ((CHAIN && CTAIL > 0) ? 1 : 0);             // phase-A slots of this body (the chain waves' extra one: CTAIL)
    // (measured and dropped in round 5, again: with phase B down to one round the helper waves requesting its shares at the loop top as
    // well -- 251 registers, none spilled this time -- 2.796-2.805 vs 2.799-2.818 ms per epoch in four alternations: nothing)
    const int NPRE = NA_BODY < NPRE_C ? NA_BODY : NPRE_C;
    u32x4_t x0[NPRE_C][4];
    int obase = 0;
    if (it > 0) {
      // ---- every chunk's share of iteration it - 1
      if (CHAIN && wave == 0) {
        bool ok = false;
        const bool fault = IS_X && ph.fault_it > 0 && it == ph.fault_it && sig == 0 && chunk == 0;      // (tests: a timed-out wait)
        for (unsigned spins = fault ? SPIN_LIMIT : 0u; spins < SPIN_LIMIT; ++spins) {
          const unsigned f = lane < nchunks ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)it | (my_xcc << 24);
          ok = __all((int)((f & 0xffffffu) >= (unsigned)it));
          if (ok) {
            const bool together = __all((int)((f >> 24) == my_xcc));             // every chunk of this critic on this workgroup's XCD
            if (it == 1 && lane == 0) ctl[1] = together ? 1 : 0;
            break;
          }
          if ((spins & 1023) == 1023 && __hip_atomic_load(ph.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0) give_up(0x100u + (unsigned)it);
      }
      __syncthreads();
      if (ctl[0]) { if (writer && threadIdx.x == 0 && it <= n_iters) lo_base[(int64_t)(2 * (it - 1)) * 4] = __builtin_nanf(""); return; }
      if (it == 1) {
        same_xcd = ctl[1] != 0;
        // (observability: counters[5] = critics of this launch whose chunks all share an XCD -- zeroed by the epoch's first launch)
        if (same_xcd && a.guard && chunk == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(a.counters + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      PSTAMP(1);                                                         // siblings' shares are there
      obase = ppar * nchunks * slabf;
      // the first four chunks' shares of this thread's layer-0 quads: requested together, before anything is waited for (clamped
      // addresses, no branches).  The other layers' quads (helper waves) are requested inside phase B, which has the chains'
      // whole first layer to hide them: a helper wave's phase A does not wait for them, and they occupy no registers until
      // then.  Fixed chunk order: every sibling sums the same bits.
#pragma unroll
      for (int u = 0; u < PS; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          // (measured and dropped in round 3: the helper waves requesting their other-layer quads here too, so that phase B starts
          // with its shares in registers -- the chains wait 1.8 k cycles for phase B: +28 spilled registers, epoch +0.015 ms)
          if (u >= NPRE) continue;                                       // (compile-time for the reference shapes: NA is)
          x0[u][w] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (obase + (w < nchunks ? w : 0) * slabf + (i_e[u] < 0 ? 0 : i_e[u]) * 4) * 4, 0, 16);
        }
      // Adam's bias corrections of this step: the tail of the previous iteration's record (whose flag this workgroup has seen:
      // sc1 loads, requested with the shares above)
      const int prev_off = (int)(((((int64_t)sig * n_iters + (it - 1)) * nchunks + chunk) * g.rec_floats) * 4);
      const int tv = 16 * (g.rec_rows4 + g.rec_mask4);
      co.bc1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs1, tv, prev_off, 16));
      co.sqrt_bc2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs1, tv + 4, prev_off, 16));
    }
    // ---- Adam on this thread's quads, weight images in LDS.  Phase A (layer 0, every thread) first: the chains need nothing else
    // to start; phase B (the other layers, waves 3-7) runs beside the chains' first layer.
    // Round 6: the step's two element-independent reciprocals -- 1 / sqrt(bc2) and lr / bc1 -- once per thread and iteration
    // (adam_update_hoisted: the compiler left them inside the per-row branches, 3 rcp + 1 sqrt per element, each a quarter-rate
    // instruction, in the phase every wave runs before the chains can start): 2.593 -> 2.575 ms per epoch, the same bits.
    float isb2 = 1.f, lrb1 = 0.f;
    auto finish = [&](int u) __attribute__((always_inline)) {
      const int li = i_li[u];
      f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
      const bool pre = u < NPRE;                                                     // the first four chunks' shares are in x0 already
      if (it > 0) {
        if (pre) {
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float on = w < nchunks ? 1.f : 0.f;
            const f32x4 xf = __builtin_bit_cast(f32x4, x0[pre ? u : 0][w]);
#pragma unroll
            for (int r = 0; r < 4; ++r) gsum[r] += on * xf[r];
          }
        }
#pragma unroll 1                                                                    // (batch 256 compiled in: unrolled, its loads were hoisted and spilled)
        for (int w0c = pre ? 4 : 0; w0c < nchunks; w0c += 4) {                        // batches above 64 rows: four more chunks at a time
          u32x4_t x[4];
#pragma unroll
          for (int w = 0; w < 4; ++w)
            x[w] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (obase + (w0c + w < nchunks ? w0c + w : 0) * slabf + (i_e[u] < 0 ? 0 : i_e[u]) * 4) * 4, 0, 16);
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float on = w0c + w < nchunks ? 1.f : 0.f;
            const f32x4 xf = __builtin_bit_cast(f32x4, x[w]);
#pragma unroll
            for (int r = 0; r < 4; ++r) gsum[r] += on * xf[r];
          }
        }
      }
      if (li < 0) return;
      const int N = li == nh ? 1 : L;
      const int n = i_n[u], k = i_k[u];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= N) continue;
#if HYPAD_R6_HOIST
        if (it > 0) adam_update_hoisted(pv[u][r], mv[u][r], vv[u][r], gsum[r], co, isb2, lrb1);
#else
        if (it > 0) adam_update(pv[u][r], mv[u][r], vv[u][r], gsum[r], co);
#endif
        const float pw = pv[u][r];
        float* wdst = li == 0 ? w0 + (n + r) * ldin + k : (li < nh ? wh + ((li - 1) * L + n + r) * LQ + k : wl + k);
        *wdst = pw;
        if (li > 0 && li < nh && k < L) whT[((li - 1) * Lp + k) * LQ + n + r] = pw;
      }
    };
    if (HYPAD_R6_HOIST && it > 0) { isb2 = __builtin_amdgcn_rcpf(co.sqrt_bc2); lrb1 = co.lr * __builtin_amdgcn_rcpf(co.bc1); }
#pragma unroll
    for (int u = 0; u < PS; ++u)
      if (u < NA_BODY) finish(u);
    PSTAMP(2);                                                           // phase A: shares loaded, Adam, layer-0 image
    auto phase_b = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < PS; ++u)
        if (u >= NA_BODY) finish(u);
    };
    if (fin) {
      if (!CHAIN) phase_b();
      if (IS_X && ph.advance && chunk == 0 && sig == 0 && threadIdx.x == 0) {      // step counters and rng tick: nobody reads
        a.counters[0] += n_iters; a.counters[1] += n_iters; a.counters[3] += n_iters;           // them inside this launch
      }
      if (writer) {                                                   // the phase's last state -> arenas
#pragma unroll
        for (int u = 0; u < PS; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (CHAIN && u >= NA_BODY) continue;                      // (a chain wave owns its phase-A slots only: the other slots' registers are dead in this body)
            const int o = arena_off(u, r);
            if (o >= 0) { arena_p[o] = pv[u][r]; arena_m[o] = mv[u][r]; arena_v[o] = vv[u][r]; }
          }
      }
      return;
    }

    // ---- record -> LDS; constant d loss / d out.  Round 6: the record of iteration it + 1 is staged at the END of iteration it, between
    // the share stores and their drain (nothing reads in0 / dm behind the barrier in front of the stores; the LDS writes fill the wait
    // for the write-through stores) instead of here, between the optimizer step and the chains; only iteration 0's is staged here.
    auto stage_record = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < MAX_ROW4; ++u) {
        const int i = threadIdx.x + u * FT;
        if (i < g.rec_rows4) { const int r = i / (Kin / 4), c4 = i - r * (Kin / 4); *reinterpret_cast<u32x4_t*>(in0 + r * ldin + 4 * c4) = rrow[u]; }
      }
#pragma unroll
      for (int u = 0; u < MAX_MASK4; ++u) {
        const int i = threadIdx.x + u * FT;
        if (i < g.rec_mask4) { const int r = i / (g.L4 / 4), c4 = i - r * (g.L4 / 4); *reinterpret_cast<u32x4_t*>(dm + r * LQ + 4 * c4) = rmask[u]; }
      }
      if (threadIdx.x < 48) { const int p = threadIdx.x >> 4; dl[(nh * 48 + threadIdx.x) * LQ] = p == 0 ? -invB : p == 1 ? invB : 1.f; }
    };
    if (!HYPAD_R6_STAGE || it == 0) stage_record();
    int* bdone = reinterpret_cast<int*>(red + 40);
    if (threadIdx.x == 0) *bdone = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PSTAMP(3);                                                           // record staged

    // ---- forward and first backward: three register-resident chains (critic_iteration_body has the commentary)
    constexpr int MF = 3, MAXNH = 4;
    const int NT = Lp >> 4;
    f32x4 DD[MAXNH][MF], DL[MF];
    const int myrow = 16 * wave + j;
    auto as4 = [](const f32x4& v) __attribute__((always_inline)) { return make_float4(v[0], v[1], v[2], v[3]); };
    if constexpr (CHAIN) {
      f32x4 T[MF];
      constexpr bool FWD0P = HYPAD_R6_FWD0P && SC != 0 && SC <= 100;
      auto load_dd = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int li = 0; li < MAXNH; ++li)
#pragma unroll
          for (int t = 0; t < MF; ++t) {
            DD[li][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (li < nh && t < NT) { const float4 v = *reinterpret_cast<const float4*>(dm + (li * 48 + myrow) * LQ + 16 * t + 4 * q); DD[li][t] = f32x4{v.x, v.y, v.z, v.w}; }
          }
      };
      if constexpr (!FWD0P) load_dd();
      auto epilogue = [&](int li) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          if (t >= NT) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * t + 4 * q + r;
            const float pre = T[t][r];
            const float dd = f < L ? leaky_slope(pre) * DD[li][t][r] : 0.f;
            DD[li][t][r] = dd;
            T[t][r] = f < L ? pre * dd : (f == L ? 1.f : 0.f);
          }
          *reinterpret_cast<float4*>(act + (li * 48 + myrow) * LQ + 16 * t + 4 * q) = as4(T[t]);
        }
      };
      float4 Aw[MF][MF];
      auto load_fwd = [&](int li) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MF; ++t)
#pragma unroll
          for (int gg = 0; gg < MF; ++gg)
            if (t < NT && gg < NT) { const int m = 16 * t + j; Aw[t][gg] = *reinterpret_cast<const float4*>(wh + ((li - 1) * L + (m < L ? m : L - 1)) * LQ + 4 * q + 16 * gg); }
      };
      auto load_bwd = [&](int li) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MF; ++t)
#pragma unroll
          for (int gg = 0; gg < MF; ++gg)
            if (t < NT && gg < NT) Aw[t][gg] = *reinterpret_cast<const float4*>(whT + (li * Lp + 16 * t + j) * LQ + 4 * q + 16 * gg);
      };
      if constexpr (FWD0P) {
        // Layer 0, software-pipelined by hand (windows up to 100): the operands of the next 16 input columns -- the window's rows ONCE for
        // both row blocks of the weights -- are requested before the 8 products of the current ones, the first ones in front of the
        // masks (LDS answers in order).  The compiler's own order was load -> wait -> 4 products, ten times over.  Every accumulator
        // still takes its groups in the same order: the same bits.  2.575 -> 2.570 ms per epoch.
        const float* bp = in0 + myrow * ldin + 4 * q;
        const float* ap[MF];
        f32x4 accA[MF], accB[MF];
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          const int m = 16 * t + j;
          ap[t] = w0 + (m < L ? m : L - 1) * ldin + 4 * q;
          accA[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accB[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float4 cb, ca[MF];
        auto ld = [&](int g0, float4& b, float4 (&a)[MF]) __attribute__((always_inline)) {
          b = *reinterpret_cast<const float4*>(bp + g0);
#pragma unroll
          for (int t = 0; t < MF; ++t)
            if (t < NT) a[t] = *reinterpret_cast<const float4*>(ap[t] + g0);
        };
        ld(0, cb, ca);
        __builtin_amdgcn_sched_barrier(0);
        load_dd();
#pragma unroll
        for (int g16 = 0; g16 < Kin; g16 += 16) {
          float4 nb, na[MF];
          if (g16 + 16 < Kin) ld(g16 + 16, nb, na);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < MF; ++t) {
            if (t >= NT) continue;
            if ((g16 & 16) == 0) accA[t] = mfma4(ca[t], cb, accA[t]); else accB[t] = mfma4(ca[t], cb, accB[t]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g16 + 16 < Kin) {
            cb = nb;
#pragma unroll
            for (int t = 0; t < MF; ++t) ca[t] = na[t];
          }
        }
#pragma unroll
        for (int t = 0; t < MF; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) T[t][r] = accA[t][r] + accB[t][r];
      } else
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        if (t >= NT) continue;
        const int m = 16 * t + j;
        const float* ap = w0 + (m < L ? m : L - 1) * ldin + 4 * q;
        const float* bp = in0 + myrow * ldin + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        for (int g16 = 0; g16 < Kin; g16 += 32) {
          acc = mfma4(*reinterpret_cast<const float4*>(ap + g16), *reinterpret_cast<const float4*>(bp + g16), acc);
          if (g16 + 16 < Kin) acc2 = mfma4(*reinterpret_cast<const float4*>(ap + g16 + 16), *reinterpret_cast<const float4*>(bp + g16 + 16), acc2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) T[t][r] = acc[r] + acc2[r];
      }
      epilogue(0);
      PSTAMP(4);                                                         // layer 0 forward
      // (measured and dropped in round 6: a "done" word per helper wave, phase B's quads dealt in the order 3, 7, 4, 5, 6 -- the second
      // layer's quads on the two waves of SIMD 3, which finish 1.5 k cycles before the ones that share a SIMD with a chain -- and the chains
      // waiting layer by layer only for the waves that hold that layer: by the shader-clock stamps the second layer would start 1 k cycles
      // earlier; on one box 2.771 against 2.744 ms per epoch -- its MFMAs then run beside waves 4-6's phase B and slow THAT down, and the
      // five-word poll alone costs 0.014 ms.  A SIMD's issue slots are what the stage is short of, not its order.)
      while (__hip_atomic_load(bdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NW - BW0) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      PSTAMP(5);                                                         // phase B's weights seen
      if (nh > 1) load_fwd(1);
      for (int li = 1; li < nh; ++li) {
        f32x4 N[MF];
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          if (t >= NT) continue;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int gg = 0; gg < MF; ++gg)
            if (gg < NT) acc = mfma4(Aw[t][gg], as4(T[gg]), acc);
          N[t] = acc;
        }
#pragma unroll
        for (int t = 0; t < MF; ++t) T[t] = N[t];
        if (li + 1 < nh) load_fwd(li + 1); else if (nh > 1) load_bwd(nh - 2);
#pragma unroll
        for (int x = 0; x < MAXNH; ++x) if (x == li) epilogue(x);
      }
      PSTAMP(6);                                                         // forward done
      const float doutp = wave == 0 ? -invB : (wave == 1 ? invB : 1.f);
      float o = 0.f;
#pragma unroll
      for (int t = 0; t < MF; ++t) {
        DL[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t >= NT) continue;
        const float4 wv = *reinterpret_cast<const float4*>(wl + 16 * t + 4 * q);
        const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o += T[t][r] * wq[r];
          float ddtop = 0.f;
#pragma unroll
          for (int x = 0; x < MAXNH; ++x) ddtop = x == nh - 1 ? DD[x][t][r] : ddtop;
          DL[t][r] = doutp * wq[r] * ddtop;
        }
        *reinterpret_cast<float4*>(dl + ((nh - 1) * 48 + myrow) * LQ + 16 * t + 4 * q) = as4(DL[t]);
      }
      for (int li = nh - 2; li >= 0; --li) {
        f32x4 N[MF];
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          N[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (t >= NT) continue;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int gg = 0; gg < MF; ++gg)
            if (gg < NT) acc = mfma4(Aw[t][gg], as4(DL[gg]), acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float ddl = 0.f;
#pragma unroll
            for (int x = 0; x < MAXNH; ++x) ddl = x == li ? DD[x][t][r] : ddl;
            N[t][r] = acc[r] * ddl;
          }
        }
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          DL[t] = N[t];
          if (t < NT) *reinterpret_cast<float4*>(dl + (li * 48 + myrow) * LQ + 16 * t + 4 * q) = as4(DL[t]);
        }
        if (li > 0) load_bwd(li - 1);
      }
      o += __shfl_xor(o, 16, 64);
      o += __shfl_xor(o, 32, 64);
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) o += __shfl_xor(o, off, 64);
      if (wave < 2 && lane == 0) red[32 + wave] = o;
      PSTAMP(7);                                                         // backward done
    } else {
      // (waves 4-6 share their SIMDs with the chain waves' back-to-back MFMAs and run this at half the speed of waves 3 and 7;
      // the chains wait ~1 k cycles for it before their second layer.  Raising these waves' priority for it did not help.)
      phase_b();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(bdone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      PSTAMP(4);                                                         // phase B done (helper waves)
    }
    if (!CHAIN && (wave == 3 || wave == 7)) {
      const int CTg = Lp >> 4;
      for (int t = wave == 3 ? 0 : 1; t < CTg * CTg; t += 2) {
        const int mt = t / CTg, nt = t - mt * CTg;
        const int m = mt * 16 + j, n = nt * 16 + j;
        const float* ap = w0 + (m < L ? m : L - 1) * ldin + 4 * q;
        const float* bp = w0 + (n < L ? n : L - 1) * ldin + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int g16 = 0; g16 < Kin; g16 += 16) {
          float4 av = *reinterpret_cast<const float4*>(ap + g16);
          const float4 bv = *reinterpret_cast<const float4*>(bp + g16);
          const int k = g16 + 4 * q;
          av.x = k == in_dim ? 0.f : av.x; av.y = k + 1 == in_dim ? 0.f : av.y; av.z = k + 2 == in_dim ? 0.f : av.z; av.w = k + 3 == in_dim ? 0.f : av.w;
          acc = mfma4(av, bv, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = mt * 16 + 4 * q + r;
          gram[mm * LQ + n] = (mm < L && n < L) ? acc[r] : 0.f;
        }
      }
    }
    if (!CHAIN && wave == BW0 + 1 && it + 1 < n_iters) await_record(it + 1);     // (a helper wave with ~6 k idle cycles here: the chains run)
    PSTAMP(8);                                                           // (helpers: Gram done)
    __syncthreads();
    PSTAMP(9);
    float gsq = 0.f;
    f32x4 acc_rf[TPW], acc_gp[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) { acc_rf[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_gp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // tiles are dealt over the seven waves that do NOT carry the second-order chain (wave 2, the interpolated rows' chain wave):
    // 28 tiles = 7 x 4 at window 100.  Wave 2 did both parts of its tiles after its chain and finished 2.4 k cycles behind.
    const int tslot = wave < 2 ? wave : wave - 1;
    auto dw_tile = [&](int i, bool rf, bool gpp) __attribute__((always_inline)) {
      const int t = tslot + (NW - 1) * i;
      if (t < g.ntiles) {
        const float* left; const float* right; int ldr;
        if constexpr (TOFF) {
          left = smem + lofs[i]; right = smem + rofs[i]; ldr = t < g.tiles0 ? ldin : LQ;
        } else {
          int li, n0, k0;
          tile_desc(t, li, n0, k0);
          const int N = li == nh ? 1 : L;
          const int nj = n0 + j < N ? n0 + j : N - 1;
          left = dl + li * 48 * LQ + nj;
          right = li == 0 ? in0 + k0 + j : act + (li - 1) * 48 * LQ + k0 + j;
          ldr = li == 0 ? ldin : LQ;
        }
        // (measured and dropped in round 5: which four rows make a k-step is free, and rows r, r + 4, r + 8, r + 12 instead of four
        // consecutive ones make these ds_read_b32 -- the kernel's only conflicting LDS accesses by counter: the lane groups q and q + 1 of
        // a 32-lane half sit 36 / 116 dwords = 4 / 20 banks apart -- conflict-free (144 / 464 dwords = 16 banks).  Same time, 2.802-2.813
        // vs 2.807-2.814 ms per epoch: the stage waits on dependent latencies, not on LDS cycles.)
        auto krow = [&](int base, int u) __attribute__((always_inline)) { return base + 4 * u + q; };
        if (rf) {
          float la[8], rb[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { la[u] = left[krow(0, u) * LQ]; rb[u] = right[krow(0, u) * ldr]; }
#if HYPAD_R6_BATCH
          __builtin_amdgcn_sched_barrier(0);               // all operands requested before the first product: ONE exposed LDS latency per tile
#endif
#pragma unroll
          for (int u = 0; u < 8; ++u) acc_rf[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc_rf[i], 0, 0, 0);
        }
        if (gpp) {
          float la[4], rb[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { la[u] = left[krow(32, u) * LQ]; rb[u] = right[krow(32, u) * ldr]; }
#if HYPAD_R6_BATCH
          __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
          for (int u = 0; u < 4; ++u) acc_gp[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(la[u], rb[u], acc_gp[i], 0, 0, 0);
        }
      }
    };
    // (measured and dropped in round 6: the tiles ONE AHEAD -- tile i + 1's operands requested under tile i's products: 2.689 against
    // 2.672 ms per epoch, 250 registers; the partner wave of the SIMD already fills the one round trip that is left)
    // (measured and dropped: requesting the operands of all of a wave's tiles first and running the k-steps tile-interleaved --
    // independent accumulators, no dependent-latency stalls -- left this stage at the same 6 k cycles and cost 54 spilled
    // registers: the stage is not bound by the MFMAs' dependent latency)
    if (CHAIN && wave == 2) {
      f32x4 E[MF];
      // (measured and dropped in round 6: this chain's weight rows requested one layer ahead, as the forward / backward chains do: +0.007 ms
      // per epoch -- wave 2 is not the stage's long pole, and the extra registers cost the other body)
      for (int li = 0; li < nh; ++li) {
        f32x4 N[MF];
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          N[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (t >= NT) continue;
          const int m = 16 * t + j;
          const float* ap = (li == 0 ? gram + m * LQ : wh + ((li - 1) * L + (m < L ? m : L - 1)) * LQ) + 4 * q;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int gg = 0; gg < MF; ++gg)
            if (gg < NT) acc = mfma4(*reinterpret_cast<const float4*>(ap + 16 * gg), li == 0 ? as4(DL[gg]) : as4(E[gg]), acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float ddl = 0.f;
#pragma unroll
            for (int x = 0; x < MAXNH; ++x) ddl = x == li ? DD[x][t][r] : ddl;
            N[t][r] = acc[r] * ddl;
          }
        }
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          E[t] = N[t];
          if (t < NT) *reinterpret_cast<float4*>(act + (li * 48 + 32 + j) * LQ + 16 * t + 4 * q) = as4(E[t]);
        }
      }
    } else {
      const int CT = (in_dim + 1 + 15) >> 4, slot = wave < 2 ? wave : wave - 1;
      const float* av = dl + (32 + j) * LQ + 4 * q;
      for (int ct = slot; ct < CT; ct += NW - 1) {
        int c = ct * 16 + j; c = c < in_dim ? c : in_dim - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#if HYPAD_R6_G
        if (Lp == 32) {                                     // (latent 17 .. 32: both k-groups' operands requested before the first product)
          float4 bv[2], avv[2];
#pragma unroll
          for (int gi = 0; gi < 2; ++gi) {
            const int o = 16 * gi + 4 * q;
            bv[gi].x = w0[(o < L ? o : L - 1) * ldin + c];
            bv[gi].y = w0[(o + 1 < L ? o + 1 : L - 1) * ldin + c];
            bv[gi].z = w0[(o + 2 < L ? o + 2 : L - 1) * ldin + c];
            bv[gi].w = w0[(o + 3 < L ? o + 3 : L - 1) * ldin + c];
            avv[gi] = *reinterpret_cast<const float4*>(av + 16 * gi);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc = mfma4(avv[0], bv[0], acc);
          acc = mfma4(avv[1], bv[1], acc);
        } else
#endif
#pragma unroll 2
        for (int g16 = 0; g16 < Lp; g16 += 16) {
          const int o = g16 + 4 * q;
          float4 bv;
          bv.x = w0[(o < L ? o : L - 1) * ldin + c];
          bv.y = w0[(o + 1 < L ? o + 1 : L - 1) * ldin + c];
          bv.z = w0[(o + 2 < L ? o + 2 : L - 1) * ldin + c];
          bv.w = w0[(o + 3 < L ? o + 3 : L - 1) * ldin + c];
          acc = mfma4(*reinterpret_cast<const float4*>(av + g16), bv, acc);
        }
        const int cc = ct * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * q + r;
          if (cc < in_dim) { in0[(32 + row) * ldin + cc] = acc[r]; gsq += acc[r] * acc[r]; }
          else if (cc == in_dim) in0[(32 + row) * ldin + cc] = 0.f;
        }
      }
      // (measured and dropped in round 5: at window 100 a wave's first two tiles are the layer-0 tiles whose k-tile is ITS g tile
      // (t = slot + 7 i, t mod 7 = slot), so their penalty-row products need nothing another wave writes and can run here, beside wave
      // 2's second-order chain, instead of behind the barrier.  Bit-equal, and SLOWER: 2.820-2.823 vs 2.810-2.817 ms per epoch in four
      // alternations -- the tile waves, not wave 2, are this stage's long pole, and the stage behind the barrier is as long as the
      // siblings' scalars take to arrive, not as its tiles.)
#pragma unroll
      for (int i = 0; i < TPW; ++i) dw_tile(i, true, false);
    }
    {
      const float tot = wave_sum(gsq);
      if (lane == 0) red[wave] = tot;
    }
    PSTAMP(10);                                                          // second-order chain (wave 2) | g + dW(real, fake)
    __syncthreads();
    PSTAMP(11);
    // ---- this chunk's scalars -> its siblings (granules: the data is the flag), as early as they exist
    const unsigned epoch = (unsigned)it + 1u;
    unsigned long long* gr = gran + (int64_t)(it & 1) * nchunks * 4;
    if (threadIdx.x < 3) {
      float v;
      if (threadIdx.x == 0) { v = 0.f; for (int w = 0; w < NW; ++w) v += red[w]; }
      else v = red[31 + threadIdx.x];
      const unsigned long long word = ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
      if (same_xcd) __hip_atomic_store(gr + chunk * 4 + threadIdx.x, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // stays in this XCD's L2
      else __hip_atomic_store(gr + chunk * 4 + threadIdx.x, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (it + 1 < n_iters) request_record(it + 1);         // lands during the tiles below and the waits that follow
    if (!(CHAIN && wave == 2)) {
#pragma unroll
      for (int i = 0; i < TPW; ++i) dw_tile(i, false, true);
    }
    PSTAMP(12);                                                          // dW(penalty rows)
    // ---- the siblings' scalars: whole-batch norm, penalty coefficient, this iteration's loss
    if (CHAIN && wave == 2) {                               // (wave 2 has no tiles: it waits for the scalars meanwhile)
      bool ok = false;
      const int c = lane / 3, f = lane - 3 * c;
      for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        unsigned long long x = (unsigned long long)epoch << 32;
        if (c < nchunks) x = __hip_atomic_load(gr + c * 4 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = __all((int)((unsigned)(x >> 32) == epoch));
        if (ok) { if (c < nchunks) xsc[f * MAXCH + c] = __uint_as_float((unsigned)x); break; }
        if ((spins & 1023) == 1023 && __hip_atomic_load(ph.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok && lane == 0) give_up(0x200u + (unsigned)it);
      // The whole-batch norm, the penalty's coefficient and this iteration's loss ONCE, on this wave (it has nothing else to do until the
      // barrier).  Until round 6 all eight waves did it behind the barrier -- four dependent LDS round trips, a square root and a division
      // at the head of the store stage: 2.747 -> 2.719 ms per epoch on one box (scripts/ab_variants.sh).  The same additions in the same
      // fixed order as before, so every chunk of the critic still gets the same bits.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float gs = 0.f, sreal = 0.f, sfake = 0.f;
      for (int w = 0; w < nchunks; ++w) { gs += xsc[w]; sreal += xsc[MAXCH + w]; sfake += xsc[2 * MAXCH + w]; }   // fixed order
      const float nrm = sqrtf(gs + 1e-12f);               // train.py:90, whole batch
      if (lane == 0) xsc[3 * MAXCH + 2] = 20.f * (nrm - 1.f) / nrm;      // d(10 gp) / d g = coef * g  (control word [2])
      if (writer && lane == 0) {
        const float gp = (nrm - 1.f) * (nrm - 1.f);
        float* lo = lo_base + (int64_t)(2 * it) * 4;
        lo[0] = sfake * invB - sreal * invB + 10.f * gp;  // train.py:98-99
        lo[1] = gp; lo[2] = sreal * invB; lo[3] = sfake * invB;
      }
    }
    __syncthreads();
    if (ctl[0]) { if (writer && threadIdx.x == 0) lo_base[(int64_t)(2 * it) * 4] = __builtin_nanf(""); return; }
    PSTAMP(13);                                                          // siblings' scalars are there
    if (ph.clear_each) clear_tiles();                     // (every read of this iteration's tiles is behind the barrier above; one
                                                          // wave -- wave 2 has no share to store -- clearing alone took 5 k cycles)
    const float coef = xsc[3 * MAXCH + 2];
    PSTAMP(17);
    // ---- merged share -> compact valid quads, write-through; then the epoch word
    // (measured and dropped in round 3: publishing dW(real, fake) and dW(penalty rows) as two unscaled parts the moment each exists --
    // the consumers apply the coefficient -- so that the epoch word need not wait for the siblings' scalars: twice the share
    // stores and loads cost more than the shorter dependency saves, 2.86 -> 3.03 ms per epoch)
    {
      const int obase = ((it & 1) * nchunks + chunk) * slabf;
#if HYPAD_R6_EOFF
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc_rf[i][r] + coef * acc_gp[i][r];
        // (16-byte stores take no scalar offset: GBuf::st4's hazard note)
        if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), xrs, eoff[i] + obase * 4, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), xrs, eoff[i] + obase * 4, 0, 16);
        PSTAMP(18 + (i < 4 ? i : 3));
      }
#else
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int t = tslot + (NW - 1) * i;
        if (t < g.ntiles && !(CHAIN && wave == 2)) {
          int li, n0, k0;
          tile_desc(t, li, n0, k0);
          const int N = li == nh ? 1 : L, K = li == 0 ? in_dim : L;
          const int n = n0 + 4 * q, k = k0 + j;
          if (n < N && k <= K) {
            const int qq = n >> 2;
            const int e = li == 0 ? qq * C0 + k : (li < nh ? I0 + (li - 1) * Ih + qq * Ch + k : I0 + (nh - 1) * Ih + k);
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc_rf[i][r] + coef * acc_gp[i][r];
            if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), xrs, (obase + e * 4) * 4, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), xrs, (obase + e * 4) * 4, 0, 16);
          }
        }
        PSTAMP(18 + (i < 4 ? i : 3));
      }
#endif
    }
    if (HYPAD_R6_STAGE && it + 1 < n_iters) stage_record();
    PSTAMP(22);      // the next iteration's record (requested behind barrier 2), under the stores' drain
    PSTAMP(14);                                                          // share stored
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
    __syncthreads();                                      // ... before ONE lane signals for all of them
    if (threadIdx.x == 0) {
      if (same_xcd) __hip_atomic_store(flags + chunk, epoch | (my_xcc << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(flags + chunk, epoch | (my_xcc << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PSTAMP(15);                                                          // drained, epoch word out
    PSTAMP(16);
  }
  };
  if (wave < BW0) run(std::true_type{}); else run(std::false_type{});
}

// Placement.  The grid is one-dimensional.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, dispatch), so the
// critic part of the grid is stretched by 8: id = 8 * slot + x runs on the XCD group x, and critic c = 2 * signal + {0: critic_x,
// 1: critic_z} takes x = c mod 8 with slots (c / 8) * nchunks + chunk -- ALL chunk workgroups of one critic on one XCD (they
// exchange a gradient share with each other every iteration; critic_x and critic_z of a signal on neighbouring XCDs), at most
// 32 of them per XCD for any signal count the resident form accepts.  Ids whose (x, slot) names no critic exit at once.  The
// record producers follow in iteration order.  A speed matter only: the body reads the XCD it really runs on from the
// hardware and picks its store flavour from that (see `same_xcd`).
template <int SC, int LC, int BC>
__global__ __launch_bounds__(FT) void critic_persistent_kernel(IterArgs ax, IterArgs az, PhaseArgs ph) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (ax.guard && ax.counters[4] != 0) return;     // fail-stop: an earlier resident launch on this state gave up (hypad_epoch_status)
  const int nchunks = (BC ? BC : ax.B) / 16, ns = ph.n_signals;
  const int crit_ids = ph.xcd_stretch ? 8 * nchunks * ((2 * ns + 7) >> 3) : 2 * ns * nchunks;
  const int id = blockIdx.x;
  if (id >= crit_ids) {             // producer workgroups (ph.rec_flags): the records, in iteration order behind the resident critics
    const int q = id - crit_ids;
    const int tile = q % nchunks, rest = q / nchunks;
    const int sig = rest % ns, zz = rest / ns;
    precompute_body<true, SC, LC>(ax, az, ph, smem, tile, sig, zz >> 1, zz & 1, ns);
    return;
  }
  int c, chunk;
  if (ph.xcd_stretch) { const int slot = id >> 3; c = (slot / nchunks) * 8 + (id & 7); chunk = slot % nchunks; }
  else { chunk = id % nchunks; const int r = id / nchunks; c = 2 * (r % ns) + r / ns; }      // (HYPAD_CRITIC_XCD=0: the round-2 order)
  if (c >= 2 * ns) return;
  if ((c & 1) == 0) critic_persistent_body<true, SC, LC, BC>(ax, ph, smem, c >> 1, chunk, ns);
  else critic_persistent_body<false, SC, LC, BC>(az, ph, smem, c >> 1, chunk, ns);
}

__global__ void advance_counters_kernel(int32_t* counters, int n, int only) {
  if (threadIdx.x == 0) {
    if (only != 1) counters[0] += n;
    if (only != 0) counters[1] += n;
    counters[3] += n;
  }
}

}  // namespace

namespace hypad {
namespace train {

bool critic_phase_supported(const hypad_dims& d) {
  const CritGeom gx = cx_geom(d.signal_shape, d.latent_dim), gz = cz_geom(d.latent_dim);
  if (!geom_supported(gx) || !geom_supported(gz)) return false;
  const int lx = iter_lds(gx).total, lz = iter_lds(gz).total;
  return (size_t)(lx > lz ? lx : lz) * sizeof(float) <= 160 * 1024;
}
// double-buffered optimiser state and gradient slabs
size_t critic_phase_fixed_floats(const hypad_dims& d) {
  const CritGeom gx = cx_geom(d.signal_shape, d.latent_dim), gz = cz_geom(d.latent_dim);
  const size_t nchunks = d.batch / 16;
  return (size_t)d.n_signals * (6 * (size_t)(gx.params + gz.params) + 2 * nchunks * (size_t)(gx.slab_floats + gz.slab_floats)) + 8;
}
// records of one iteration
size_t critic_phase_floats_per_iter(const hypad_dims& d) {
  const CritGeom gx = cx_geom(d.signal_shape, d.latent_dim), gz = cz_geom(d.latent_dim);
  return (size_t)d.n_signals * (d.batch / 16) * (size_t)(gx.rec_floats + gz.rec_floats) + 4;      // + Adam bias corrections
}

// Runs the critic phase of an epoch (train.py:315-328) for n_iters = n_critics * n_batches (critic_x || critic_z)
// iterations.  ax / az: arguments of the two iterations (loss_sig_stride set; row_index and losses are taken from the
// arguments here).  extra: `extra_floats` floats of scratch; the phase is cut into chunks of as many iterations as fit.
// losses: iteration `it` writes rows 2*it (critic_x) and 2*it+1 (critic_z) of each signal's loss table.  ev (optional,
// 4 events, profiling): recorded before the precompute, after it, after the first iteration launch (no Adam prologue)
// and after the last one (n_iters - 1 steady-state launches back to back: event overhead amortised).
// CUs of the current device (cached): the persistent form needs every workgroup resident, one per CU
static int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus[dev] = n > 0 ? n : -1;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}
using IterKernel = void (*)(IterArgs, IterArgs, PhaseArgs);
// compile-time shapes: BASELINE.json configs[0..2] (univariate) and configs[3] (5 channels x 30 = 150 wide, batch 256)
// and the two shapes of the reference's shipped configs/multivariate.yaml:5-7 (WADI: 123 wide, SWAT: 51 wide; batch 64)
static int shape_slot(int S, int L, int B) {
  return (S == 100 && L == 20 && B == 64) ? 0 : (S == 150 && L == 20 && B == 256) ? 1 : (S == 123 && L == 20 && B == 64) ? 2 : (S == 51 && L == 20 && B == 64) ? 3 : 4;
}
static IterKernel phase_kernel(int S, int L, int B, bool persistent) {
  switch (shape_slot(S, L, B)) {
    case 0: return persistent ? critic_persistent_kernel<100, 20, 64> : critic_iteration_kernel<100, 20, 64>;
    case 1: return persistent ? critic_persistent_kernel<150, 20, 256> : critic_iteration_kernel<150, 20, 0>;
    case 2: return persistent ? critic_persistent_kernel<123, 20, 64> : critic_iteration_kernel<123, 20, 64>;
    case 3: return persistent ? critic_persistent_kernel<51, 20, 64> : critic_iteration_kernel<51, 20, 64>;
    default: return persistent ? critic_persistent_kernel<0, 0, 0> : critic_iteration_kernel<0, 0, 0>;
  }
}
// Does the runtime place at least one workgroup of the resident kernel on a CU (registers + the full LDS plan)?  Asked once
// per kernel instantiation and device.  (Residency itself comes from the grid: workgroups <= CUs, one per CU.)
static bool persistent_kernel_fits(int S, int L, int B, size_t lds) {
  static int cache[64][5] = {{0}};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  const int slot = shape_slot(S, L, B);
  if (cache[dev][slot] == 0 || slot == 4) {
    const void* kfn = (const void*)phase_kernel(S, L, B, true);
    int blocks = 0;
    bool ok = true;
    if (lds > 64 * 1024) ok = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (ok) ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kfn, FT, lds) == hipSuccess && blocks >= 1;
    cache[dev][slot] = ok ? 1 : -1;
  }
  return cache[dev][slot] > 0;
}
bool critic_phase_persistent(const hypad_dims& d, int flags) {
  if (flags & HYPAD_EPOCH_PER_ITERATION) return false;
  const CritGeom gx = cx_geom(d.signal_shape, d.latent_dim), gz = cz_geom(d.latent_dim);
  const int nchunks = d.batch / 16;
  if (!persist_geom_supported(gx, nchunks) || !persist_geom_supported(gz, nchunks)) return false;
  const int lx = iter_lds(gx).total, lz = iter_lds(gz).total;
  const size_t lds = (size_t)((lx > lz ? lx : lz) + 3 * MAXCH + 4) * sizeof(float);
  if (lds > 160 * 1024) return false;
  if ((long long)nchunks * d.n_signals * 2 > device_cus()) return false;       // all resident, one workgroup per CU
  return persistent_kernel_fits(d.signal_shape, d.latent_dim, d.batch, lds);
}

// How a phase of n_iters iterations runs in `extra`: iterations per slice, and whether the resident launch carries its own record
// producers (no precompute launch in front; HYPAD_CRITIC_PRODUCERS=0 turns it off: A/B timing and the equality test)
struct PhasePlan { bool ok, persistent, fused; int cap; size_t fixed, per_iter; };
static PhasePlan plan_phase(const hypad_dims& d, size_t extra_floats, int n_iters, int flags = 0) {
  PhasePlan p{};
  p.fixed = critic_phase_fixed_floats(d); p.per_iter = critic_phase_floats_per_iter(d);
  if (!critic_phase_supported(d) || extra_floats < p.fixed + p.per_iter) return p;
  p.ok = true;
  p.cap = (int)((extra_floats - p.fixed) / p.per_iter);
  if (p.cap > n_iters) p.cap = n_iters;
  const CritGeom gx = cx_geom(d.signal_shape, d.latent_dim), gz = cz_geom(d.latent_dim);
  p.persistent = critic_phase_persistent(d, flags);
  // Producers share the chip with the resident critics, one workgroup per CU either way: they must find free CUs (every CU taken
  // by a critic that waits for its record would be a deadlock -- the bounded polls would end it with an error) and enough of
  // them to keep up: the critics may hold half of the CUs at most (measured: 12 signals 4.67 -> 3.94 ms per epoch, 16 signals --
  // half the chip -- 5.02 -> 4.95; beyond that the precompute launch in front is the better form).
  const long long critics = (long long)(d.batch / 16) * d.n_signals * 2;
  p.fused = p.persistent && !(flags & HYPAD_EPOCH_NO_PRODUCERS) && 2 * critics <= device_cus() &&
            (size_t)d.n_signals * n_iters * (d.batch / 16) * (size_t)(gx.rec_floats > gz.rec_floats ? gx.rec_floats : gz.rec_floats) * 4 < ((size_t)1 << 30);
  return p;
}
bool critic_phase_producers(const hypad_dims& d, int n_iters) {
  if (!critic_phase_supported(d) || n_iters <= 0) return false;
  return plan_phase(d, critic_phase_fixed_floats(d) + (size_t)n_iters * critic_phase_floats_per_iter(d), n_iters).fused;
}
// [epoch words | error word | granules | record flags]: ONE block at the (16-byte aligned) start of `extra`, zero before every
// resident launch (a word left by an earlier launch looks valid: Guideline 16, "re-initialise every call")
static size_t sync_block_bytes(const hypad_dims& d, const PhasePlan& p) {
  const size_t nchunks = d.batch / 16, ns = d.n_signals;
  const size_t flag_bytes = ((2 * ns * nchunks + 4) * sizeof(unsigned) + 15) & ~(size_t)15;
  size_t b = flag_bytes + 2 * ns * 2 * nchunks * 4 * sizeof(unsigned long long);
  if (p.fused) b += ((2 * ns * p.cap * nchunks) * sizeof(unsigned) + 15) & ~(size_t)15;
  return b;
}
// What hypad_train_epoch's first launch (the weight pack) can zero for the phase: the first slice's block, or nothing
void critic_phase_zero_block(const hypad_dims& d, float* extra, size_t extra_floats, int n_iters, unsigned** ptr, int* words, int flags) {
  *ptr = nullptr; *words = 0;
  const PhasePlan p = plan_phase(d, extra_floats, n_iters, flags);
  if (!p.ok || !p.persistent) return;
  *ptr = (unsigned*)(((uintptr_t)extra + 15) & ~(uintptr_t)15);
  *words = (int)(sync_block_bytes(d, p) / 4);
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, int words) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < words) p[i] = 0u;
}

int run_critic_phase(IterArgs ax, IterArgs az, const int32_t* row_index, int n_iters, float* losses, float* extra, size_t extra_floats,
                     int n_signals, hipStream_t s, hipEvent_t* ev, const hypad_epoch_noise* noise, int* persistent_used,
                     const unsigned* zeroed, int flags, int only, float* enc_table, int64_t enc_rows) {
  hypad_dims d; d.signal_shape = ax.S; d.latent_dim = ax.L; d.batch = ax.B; d.hyperbolic = ax.hyperbolic; d.n_signals = n_signals; d.first_signal = ax.rng_sig0;
  if (!critic_phase_supported(d)) return HYPAD_EUNSUPPORTED;
  const PhasePlan plan = plan_phase(d, extra_floats, n_iters, flags);
  if (!plan.ok) return HYPAD_EWORKSPACE;
  const size_t fixed = plan.fixed;
  const int cap = plan.cap;
  const CritGeom gx = cx_geom(ax.S, ax.L), gz = cz_geom(ax.L);
  const int nchunks = ax.B / 16;
  const bool persistent = plan.persistent, fused = plan.fused;
  if (persistent_used) *persistent_used = persistent ? 1 : 0;
  // the generator-side randomness (z, decoder dropout) keeps the critic_x seed; critic_z draws from its own
  az.seed = ax.seed ^ CRITIC_Z_SEED_XOR;
  const size_t lds_pre = (size_t)pre_lds(ax.S).total * sizeof(float);
  if (lds_pre > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)precompute_kernel(ax.S, ax.L), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pre);
    if (e != hipSuccess) return (int)e;
  }
  const int lx = iter_lds(gx).total, lz = iter_lds(gz).total;
  // (persistent form: the same -- full -- LDS request for both critics keeps it at one workgroup per CU)
  size_t lds = (size_t)((lx > lz ? lx : lz) + (persistent ? 3 * MAXCH + 4 : 0)) * sizeof(float);
  if (fused) {                                            // producer workgroups: layer buffers + the record's LDS image
    const size_t need = (size_t)(pre_lds(ax.S).total + 4 + (gx.rec_floats > gz.rec_floats ? gx.rec_floats : gz.rec_floats)) * sizeof(float);
    if (need > lds) lds = need;
  }
  const IterKernel kern = phase_kernel(ax.S, ax.L, ax.B, persistent);
  const void* kfn = (const void*)kern;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  PhaseArgs ph{};
  ph.fault_it = (flags >> HYPAD_EPOCH_TEST_GIVE_UP_SHIFT) & 0xff;
  ph.n_signals = n_signals;
  ph.only = only;
  ph.xcd_stretch = (flags & HYPAD_EPOCH_ID_ORDER) ? 0 : 1;
  ph.clear_each = (flags & HYPAD_EPOCH_CLEAR_TILES) ? 1 : 0;
  // the fixed area: optimiser state + gradient slabs of the per-iteration launches, or -- carved out of the same floats -- the
  // persistent form's exchange buffers: [epoch words | error word] (zeroed before every launch), granules, merged shares
  float* p = extra;
  size_t sync_bytes = 0;
  if (persistent) {
    // [epoch words | error word | granules]: ONE block, zeroed before every launch (a granule left by an earlier launch
    // carries a valid-looking epoch tag: Guideline 16, "re-initialise every call")
    uintptr_t up = ((uintptr_t)p + 15) & ~(uintptr_t)15;
    ph.flags = (unsigned*)up;
    const size_t nflags = (size_t)2 * n_signals * nchunks;
    const size_t flag_bytes = ((nflags + 4) * sizeof(unsigned) + 15) & ~(size_t)15;
    ph.err = ph.flags + nflags;
    ph.gran_x = (unsigned long long*)(up + flag_bytes);
    ph.gran_z = ph.gran_x + (size_t)n_signals * 2 * nchunks * 4;
    sync_bytes = sync_block_bytes(d, plan);
    unsigned* after_gran = (unsigned*)(ph.gran_z + (size_t)n_signals * 2 * nchunks * 4);
    if (fused) { ph.rec_flags = after_gran; after_gran = (unsigned*)((char*)up + sync_bytes); }
    ph.xslab_x = (float*)after_gran;
    ph.xslab_z = ph.xslab_x + (size_t)n_signals * 2 * nchunks * persist_items(gx) * 4;
    if ((size_t)((ph.xslab_z + (size_t)n_signals * 2 * nchunks * persist_items(gz) * 4) - extra) > fixed) return HYPAD_EWORKSPACE;
    p = extra + fixed - 8;
  } else {
    ph.state_x = p; p += (size_t)n_signals * 6 * gx.params;
    ph.state_z = p; p += (size_t)n_signals * 6 * gz.params;
    ph.slab_x = p; p += (size_t)n_signals * 2 * nchunks * gx.slab_floats;
    ph.slab_z = p; p += (size_t)n_signals * 2 * nchunks * gz.slab_floats;
  }
  float* bcorr = p; p += pad4(4 * (cap + 1));
  float* recs = p;
#if HYPAD_DIAG
  ph.stamps = g_stamps;
#else
  ph.stamps = nullptr;
#endif
  if (enc_table && enc_rows > 0 && only != 0) {            // encoder(x) once per window row, for every slice of the phase
    const bool cfg100 = ax.S == 100 && ax.L == 20;
    const void* tfn = cfg100 ? (const void*)encoder_table_kernel<100, 20> : (const void*)encoder_table_kernel<0, 0>;
    if (lds_pre > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(tfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pre);
      if (e != hipSuccess) return (int)e;
    }
    const dim3 tgrid((unsigned)((enc_rows + 15) / 16), n_signals);
    if (cfg100) hipLaunchKernelGGL((encoder_table_kernel<100, 20>), tgrid, dim3(TB), lds_pre, s, az, enc_table, enc_rows);
    else hipLaunchKernelGGL((encoder_table_kernel<0, 0>), tgrid, dim3(TB), lds_pre, s, az, enc_table, enc_rows);
    HYPAD_CHECK_LAUNCH();
    ph.enc_table = enc_table; ph.enc_rows = enc_rows;
  }
  for (int it0 = 0; it0 < n_iters; it0 += cap) {
    const int n = n_iters - it0 < cap ? n_iters - it0 : cap;
    ph.n_iters = n;
    ph.row_index = row_index ? row_index + (int64_t)it0 * ax.B : nullptr;
    ph.losses = losses + (int64_t)(2 * it0) * 4;
    ph.rec_x = recs;
    ph.rec_z = recs + (size_t)n_signals * n * nchunks * gx.rec_floats;
    ph.bias_corr = bcorr;
    ph.it = 0;
    {
      const int64_t per = (int64_t)it0 * n_signals;                  // (iteration, signal) slices ahead of this chunk
      auto adv = [&](const float* p, int64_t slice) { return p ? p + per * slice : nullptr; };
      const int64_t B = ax.B;
      ph.inj_z_x = adv(noise ? noise->z_cx : nullptr, B * ax.L);
      ph.inj_al_x = adv(noise ? noise->alpha_cx : nullptr, B * ax.S);
      ph.inj_z_z = adv(noise ? noise->z_cz : nullptr, B * ax.L);
      ph.inj_al_z = adv(noise ? noise->alpha_cz : nullptr, B * ax.L);
      ph.inj_mk_x = adv(noise && ax.drop_mode == 1 ? noise->masks_cx : nullptr, ax.mask_sig_stride);
      ph.inj_mk_z = adv(noise && az.drop_mode == 1 ? noise->masks_cz : nullptr, az.mask_sig_stride);
    }
    if (ev) (void)hipEventRecord(ev[0], s);
    // epoch words, error word, granules (and record flags): zero before EVERY resident launch -- by the precompute launch when
    // there is one and its grid has a thread per word, by a memset node otherwise
    const bool fold = persistent && !fused && sync_bytes / 4 <= (size_t)nchunks * n_signals * 2 * n * TB;
    ph.zero_ptr = fold ? ph.flags : nullptr; ph.zero_words = fold ? (int)(sync_bytes / 4) : 0; ph.advance = persistent ? 1 : 0;
    if (!fused) {
      hipLaunchKernelGGL(precompute_kernel(ax.S, ax.L), dim3(nchunks, n_signals, (only < 0 ? 2 : 1) * n), dim3(TB), lds_pre, s, ax, az, ph);
      HYPAD_CHECK_LAUNCH();
    }
    if (ev) (void)hipEventRecord(ev[1], s);
    if (persistent) {
      if (!fold && !(it0 == 0 && zeroed == ph.flags)) {   // (the first slice's block may come zeroed from the caller's previous launch)
        // a KERNEL, not hipMemsetAsync: inside a captured epoch a memset NODE between two resident launches (phases of more than
        // 512 iterations: one per slice) was not reliably ordered against its neighbours on ROCm 7.2 -- queued replays of a sliced
        // phase went non-finite (round 6: window 100 and 123 at 20 480 windows, 1 600 iterations; eager launches and
        // one-replay-at-a-time never did) -- a kernel node is.
        const int words = (int)(sync_bytes / 4);
        hipLaunchKernelGGL(zero_words_kernel, dim3((words + 255) / 256), dim3(256), 0, s, ph.flags, words);
        HYPAD_CHECK_LAUNCH();
      }
      if (ev) (void)hipEventRecord(ev[2], s);
      // (one-dimensional grid: the critics' ids stretched by 8 for XCD placement, then the producers -- critic_persistent_kernel)
      const unsigned crit_ids = ph.xcd_stretch ? 8u * nchunks * ((2u * n_signals + 7u) >> 3) : 2u * n_signals * nchunks;
      hipLaunchKernelGGL(kern, dim3(crit_ids + (fused ? 2u * n * nchunks * n_signals : 0u)), dim3(FT), lds, s, ax, az, ph);
      HYPAD_CHECK_LAUNCH();
      if (ev) (void)hipEventRecord(ev[3], s);
    } else {
      for (int it = 0; it <= n; ++it) {                  // launch n: finalise (last Adam step -> arenas)
        ph.it = it;
        const dim3 grid((1 << XS) * (it == n ? 1 : nchunks), n_signals, only < 0 ? 2 : 1);
        hipLaunchKernelGGL(kern, grid, dim3(FT), lds, s, ax, az, ph);
        HYPAD_CHECK_LAUNCH();
        if (ev && (it == 0 || it == n - 1)) (void)hipEventRecord(ev[it == 0 ? 2 : 3], s);
      }
    }
    if (!persistent) {
      hipLaunchKernelGGL(advance_counters_kernel, dim3(1), dim3(64), 0, s, ax.counters, n, only);
      HYPAD_CHECK_LAUNCH();
    }
  }
  return HYPAD_OK;
}

// hypad_epoch_record_info: where the records of the last phase chunk sit inside `extra` and how one is laid out
int critic_phase_record_info(const hypad_dims& d, int n_iters, int critic, hypad_record_info* out) {
  if (!critic_phase_supported(d) || n_iters <= 0 || n_iters > 512 || critic < 0 || critic > 1 || !out) return HYPAD_EINVAL;
  const CritGeom gx = cx_geom(d.signal_shape, d.latent_dim), gz = cz_geom(d.latent_dim);
  const size_t nchunks = d.batch / 16;
  size_t o = critic_phase_fixed_floats(d) - 8;            // the fixed area (either form of the phase)
  o += pad4(4 * (n_iters + 1));
  if (critic == 1) o += (size_t)d.n_signals * n_iters * nchunks * gx.rec_floats;
  const CritGeom& g = critic == 0 ? gx : gz;
  out->offset_floats = (int64_t)o;
  out->record_floats = g.rec_floats; out->row_stride = g.Kin; out->mask_offset_floats = 48 * g.Kin; out->mask_row_stride = g.L4;
  out->n_layers = g.nh; out->in_dim = g.in_dim;
  return HYPAD_OK;
}

}  // namespace train
}  // namespace hypad

#if HYPAD_DIAG
// development aid (dev library only): device buffer of 128 int64 that the next critic launches stamp, or null
extern "C" __attribute__((visibility("default"))) void hypad_diag_set_fused_stamps(long long* p) { g_stamps = p; }
#endif
