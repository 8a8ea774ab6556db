// Network forwards and the dense building blocks, as row-tile kernels:
// one workgroup = 16 (or 32) window rows carried through every layer with activations in LDS.
#include <hip/hip_runtime.h>

#include "../../include/hypad.h"
#include "nets.h"

using namespace hypad;

namespace {

constexpr int THREADS = 256;
constexpr int WST = (THREADS / 64) * WSTAGE_FLOATS;     // wave-private weight slabs of gemm_nt

__host__ __device__ inline int ld_of(int n) { return pad4(n) + 4; }

hipError_t allow_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

__device__ __forceinline__ DropSrc make_drop(const hypad_dropout& d, int batch, uint32_t stream, float p) {
  DropSrc s;
  s.mode = d.train_mode ? (d.masks ? 1 : 2) : 0;
  s.ptr = d.masks; s.batch = batch; s.seed = d.seed; s.tick = (uint32_t)d.offset; s.stream = stream; s.sig = 0; s.p = p;
  return s;
}

// ------------------------------------------------------------------------------------------ generic linear
__global__ __launch_bounds__(THREADS) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              int64_t rows, int K, int N, int act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = ld_of(K), ldy = ld_of(N);
  float* xs = smem;
  float* ys = xs + 16 * ldx;
  float* wst = ys + 16 * ldy;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  tile_load(xs, ldx, x + r0 * K, K, 16, K, valid);
  __syncthreads();
  gemm_nt<1>(xs, ldx, w, K, K, N, identity_map(), b, nullptr, ys, ldy, 0, wst);
  __syncthreads();
  for (int i = threadIdx.x; i < valid * N; i += THREADS) {
    int r = i / N, c = i - r * N;
    float v = ys[r * ldy + c];
    if (act == HYPAD_ACT_TANH) v = tanhf(v);
    else if (act == HYPAD_ACT_LEAKY02) v = v > 0.f ? v : LEAK * v;
    y[(r0 + r) * N + c] = v;
  }
}

// grad_pre = grad_y * act'(y) -> scratch (rows, N) ; grad_x = grad_pre W
__global__ __launch_bounds__(THREADS) void linear_bwd_data_kernel(const float* __restrict__ w, const float* __restrict__ y,
                                                                   const float* __restrict__ gy, float* __restrict__ gpre,
                                                                   float* __restrict__ gx, int64_t rows, int K, int N, int act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldd = ld_of(N), ldx = ld_of(K);
  float* ds = smem;
  float* xs = ds + 16 * ldd;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  for (int i = threadIdx.x; i < 16 * N; i += THREADS) {
    int r = i / N, c = i - r * N;
    float v = 0.f;
    if (r < valid) {
      v = gy[(r0 + r) * N + c];
      float o = y ? y[(r0 + r) * N + c] : 0.f;
      if (act == HYPAD_ACT_TANH) v *= 1.f - o * o;
      else if (act == HYPAD_ACT_LEAKY02) v *= o > 0.f ? 1.f : LEAK;
      if (gpre) gpre[(r0 + r) * N + c] = v;
    }
    ds[r * ldd + c] = v;
  }
  __syncthreads();
  if (gx) {
    gemm_nn<1>(ds, ldd, 0, w, K, N, identity_map(), K, xs, ldx, false);
    __syncthreads();
    tile_store(gx + r0 * K, K, xs, ldx, 16, K, valid);
  }
}

// grad_w[n][k] = sum_r left[r][n] * right[r][k]; 16x16 tile per wave on MFMA; optional bias column sums.
__global__ __launch_bounds__(THREADS) void outer_sum_kernel(const float* __restrict__ left, int ldl, const float* __restrict__ right,
                                                             int ldr, float* __restrict__ gw, float* __restrict__ gb,
                                                             int64_t rows, int N, int K) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int j = lane & 15, q = lane >> 4;
  const int tk = (K + 15) >> 4, tn = (N + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile < tn * tk) {
    const int n0 = (tile / tk) * 16, k0 = (tile % tk) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t rr = 0; rr < rows; rr += 4) {
      const int64_t r = rr + q;
      const float a = (r < rows && n0 + j < N) ? left[r * ldl + n0 + j] : 0.f;
      const float b = (r < rows && k0 + j < K) ? right[r * ldr + k0 + j] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n = n0 + 4 * q + r, k = k0 + j;
      if (n < N && k < K) gw[(size_t)n * K + k] = acc[r];
    }
  }
  if (gb) {
    int n = blockIdx.x * THREADS + threadIdx.x;
    if (n < N) {
      float s = 0.f;
      for (int64_t r = 0; r < rows; ++r) s += left[r * ldl + n];
      gb[n] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------ generic BiLSTM (T=1)
__global__ __launch_bounds__(THREADS) void lstm_fwd_kernel(const float* __restrict__ x, const float* wf, const float* bif,
                                                            const float* bhf, const float* wr, const float* bir,
                                                            const float* bhr, float* __restrict__ out,
                                                            float* __restrict__ gates_save, int64_t rows, int K, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = ld_of(K), ldg = ld_of(6 * H), ldh = ld_of(2 * H);
  float* xs = smem;
  float* gs = xs + 16 * ldx;
  float* hs = gs + 16 * ldg;
  float* wst = hs + 16 * ldh;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  tile_load(xs, ldx, x + r0 * K, K, 16, K, valid);
  __syncthreads();
  gemm_nt<1>(xs, ldx, wf, K, K, 3 * H, lstm_gate_map(H), bif, bhf, gs, ldg, 0, wst);
  gemm_nt<1>(xs, ldx, wr, K, K, 3 * H, lstm_gate_map(H), bir, bhr, gs, ldg, 3 * H, wst);
  __syncthreads();
  lstm_cell_tile(gs, ldg, H, 16, hs, ldh, gates_save ? gates_save + r0 * 8 * H : nullptr, valid);
  __syncthreads();
  tile_store(out + r0 * 2 * H, 2 * H, hs, ldh, 16, 2 * H, valid);
}

// ---- the same layer for MANY rows (round 3): weights stationary in LDS.  lstm_fwd_kernel above re-reads the layer's weights from L2
// for every 16-row tile and re-shapes them through a wave-private LDS slab (gemm_nt: 25-28 % of the SIMD-cycles are matrix-pipe
// cycles by counter, profiles/r03_mfma_util.json).  Here a 512-thread workgroup owns ONE direction: its three gate blocks of W_ih
// (i, g, o: the f gate meets c0 = 0) sit in LDS for the workgroup's lifetime -- [gate][unit][k], row stride kpad + 4 (= 4 x odd:
// the 16 lanes of a ds_read_b128 phase hit 16 different 4-bank groups) -- and every wave walks its own 16-row tiles: the tile's x
// rows go straight from global memory into the MFMA A layout (lane (row, kq) holds x[row][16 g + 4 kq .. + 3]; the four MFMAs of a
// k-group use the components in turn, so A and B agree on a permuted k order), one 16-unit block at a time takes its three
// accumulators through the whole K, and the cell runs on the accumulators (lane (unit, q), register r = row 4 q + r: the three
// gates of a unit on one lane).  No barrier after the weights are staged.
template <int KG, int NW>              // k-groups of 16: in_dim <= 16 KG; NW waves per workgroup (16 where the registers allow: more waves to put
                                       // under the layer's stores)
__global__ __launch_bounds__(64 * NW) void lstm_fwd_lds_kernel(const float* __restrict__ x, const float* wf, const float* bif, const float* bhf,
                                                            const float* wr, const float* bir, const float* bhr, float* __restrict__ out,
                                                            float* __restrict__ gates_save, int64_t rows, int K, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int dir = blockIdx.x & 1, slice = blockIdx.x >> 1, nslices = gridDim.x >> 1;
  const float* __restrict__ w = dir ? wr : wf;
  const float* __restrict__ b1 = dir ? bir : bif;
  const float* __restrict__ b2 = dir ? bhr : bhf;
  const int Hp = (H + 15) & ~15, nub = Hp >> 4;
  constexpr int ld = KG * 16 + 4;
  for (int idx = threadIdx.x; idx < 3 * Hp * ld; idx += 64 * NW) {
    const int g3 = idx / (Hp * ld), rem = idx - g3 * Hp * ld, n = rem / ld, k = rem - n * ld;
    smem[idx] = (n < H && k < K) ? w[(int64_t)((g3 == 0 ? 0 : g3 + 1) * H + n) * K + k] : 0.f;      // PyTorch gate order i, f, g, o
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), j = lane & 15, q = lane >> 4;
  const int64_t ntiles = (rows + 15) >> 4;
  const bool vec = (K & 3) == 0 && ((uintptr_t)x & 15) == 0;
  const bool nt_ok = (H & 15) == 0 && gates_save && ((uintptr_t)gates_save & 63) == 0;
  // A tile: rows past the end repeat the last one (their results are not stored)
  auto load_a = [&](float4 (&a)[KG], int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile << 4;
    const int64_t row = r0 + j < rows ? r0 + j : rows - 1;
    const float* xr = x + row * K;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const int k0 = 16 * g + 4 * q;
      if (vec && k0 + 3 < K) a[g] = *reinterpret_cast<const float4*>(xr + k0);
      else a[g] = make_float4(k0 < K ? xr[k0] : 0.f, k0 + 1 < K ? xr[k0 + 1] : 0.f, k0 + 2 < K ? xr[k0 + 2] : 0.f, k0 + 3 < K ? xr[k0 + 3] : 0.f);
    }
  };
  const int64_t tstep = (int64_t)nslices * NW;
  float4 a[KG], an[KG];
  int64_t tile = (int64_t)slice * NW + wave;
  if (tile < ntiles) load_a(a, tile);
  for (; tile < ntiles; tile += tstep) {
    const int64_t r0 = tile << 4;
    // the next tile's rows are requested before this tile's results are stored: memory operations of a wave return in order, so
    // loads issued behind the epilogue's ~80 stores would wait for all of them
    if (tile + tstep < ntiles) load_a(an, tile + tstep);
    // two 16-unit blocks at a time: their stores land next to each other in time, so the two 64-byte halves of a 128-byte line of the
    // saved gates meet in L2 instead of leaving it one by one (the output is 2 KB per row: the launch is bound by its writes)
    for (int ub = 0; ub < nub; ub += 2) {
      const bool two = ub + 1 < nub;
      f32x4 ai = {0.f, 0.f, 0.f, 0.f}, ag = ai, ao = ai, ci = ai, cg = ai, co = ai;
      const float* wb = smem + (ub * 16 + j) * ld + 4 * q;
      const float* wc = wb + (two ? 16 * ld : 0);
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const float4 bi = *reinterpret_cast<const float4*>(wb + 16 * g);
        const float4 bg = *reinterpret_cast<const float4*>(wb + Hp * ld + 16 * g);
        const float4 bo = *reinterpret_cast<const float4*>(wb + 2 * Hp * ld + 16 * g);
        const float4 di = *reinterpret_cast<const float4*>(wc + 16 * g);
        const float4 dg = *reinterpret_cast<const float4*>(wc + Hp * ld + 16 * g);
        const float4 dO = *reinterpret_cast<const float4*>(wc + 2 * Hp * ld + 16 * g);
#define HYPAD_LSTM_STEP(c)                                                   \
        ai = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].c, bi.c, ai, 0, 0, 0); \
        ag = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].c, bg.c, ag, 0, 0, 0); \
        ao = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].c, bo.c, ao, 0, 0, 0); \
        if (two) {                                                           \
          ci = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].c, di.c, ci, 0, 0, 0); \
          cg = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].c, dg.c, cg, 0, 0, 0); \
          co = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].c, dO.c, co, 0, 0, 0); \
        }
        HYPAD_LSTM_STEP(x) HYPAD_LSTM_STEP(y) HYPAD_LSTM_STEP(z) HYPAD_LSTM_STEP(w)
#undef HYPAD_LSTM_STEP
      }
      // stores through buffer addressing: the tile's base in the descriptor, the lane's (row 4 q, unit) once as a 32-bit offset,
      // row and gate as scalar offsets -- per-element 64-bit address arithmetic was a third of this epilogue's instructions
      const GBuf ob(out + r0 * 2 * H), gb(gates_save ? gates_save + r0 * 8 * H : out);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int unit = (ub + half) * 16 + j;
        if (half == 1 && !two) break;
        if (unit >= H) continue;
        const f32x4 pi = half ? ci : ai, pg = half ? cg : ag, po = half ? co : ao;
        const float bsi = b1[unit] + b2[unit], bsg = b1[2 * H + unit] + b2[2 * H + unit], bso = b1[3 * H + unit] + b2[3 * H + unit];
        const int vo_h = (4 * q * 2 * H + dir * H + unit) * 4, vo_g = (4 * q * 8 * H + dir * 4 * H + unit) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r0 + 4 * q + r >= rows) continue;
          const float gi = sigmoidf_(pi[r] + bsi), gg = tanhf_(pg[r] + bsg), go = sigmoidf_(po[r] + bso);
          const float tc = tanhf_(gi * gg);
          ob.st(go * tc, vo_h, r * 2 * H * 4);
          if (gates_save) {
            // whole, aligned 64-byte segments (hidden a multiple of 16): past the caches (non-temporal) -- 323 -> 288 us at 128 -> 2 x 64;
            // with 200-byte gate rows (hidden 50) the same hint makes partial lines and costs 100 us, so it is not given there
            if (nt_ok) {
              auto nt = [&](float v, int so) __attribute__((always_inline)) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), gb.rs, vo_g, so, 2); };
              nt(gi, r * 8 * H * 4); nt(gg, (r * 8 * H + H) * 4); nt(go, (r * 8 * H + 2 * H) * 4); nt(tc, (r * 8 * H + 3 * H) * 4);
            } else {
              gb.st(gi, vo_g, r * 8 * H * 4); gb.st(gg, vo_g, (r * 8 * H + H) * 4);
              gb.st(go, vo_g, (r * 8 * H + 2 * H) * 4); gb.st(tc, vo_g, (r * 8 * H + 3 * H) * 4);
            }
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) a[g] = an[g];
  }
}

// ---- round 4: the same layer with the reference's shapes folded in.  What the kernel above loses (rocprofv3 + what-if builds, round 4:
// 349 us per 200 000 rows at 100 -> 2 x 50 with the gates saved, 217 us with every store removed, against 109 us of MFMA issue): its
// run-time "second unit block" branch splits the K loop into basic blocks, the compiler -- at the 128-register budget of four waves
// per SIMD -- reloads the second block's B operands dword by dword right in front of each MFMA (ds_read_b32, s_waitcnt lgkmcnt(0),
// v_mfma: the LDS latency exposed three times per k-step), and a quarter of its MFMAs multiply padding (hidden 50 in blocks of 64,
// K = 100 in groups of 16).  Here H and the k-groups are template arguments:
//  * column tiles: gate i / g / o of every FULL 16-unit block (H / 16 of them), then ONE remainder tile that holds the last H % 16
//    units of all three gates side by side (columns [i .. | g .. | o ..], H % 16 <= 5): 10 tiles at H = 50 instead of 12;
//  * K: full groups of 16 exactly as above (lane (row, q) holds x[row][16 g + 4 q ..+ 3], MFMA c takes component c), the last
//    partial group in the order k = 16 g + 4 c + q, so it costs ceil(rest / 4) MFMAs instead of four: 25 per tile at K = 100, not 28;
//  * two passes per row tile (two unit blocks = six accumulators; at H = 50 the second pass has four), every pass a straight-line K
//    loop; the NEXT tile's x rows are requested inside the last pass, group by group, into the registers the MFMAs have just read
//    (no second register set: the kernel stays below the budget without spilling);
//  * the summed biases sit in LDS.
// Any other shape keeps the kernel above.
template <int K, int H, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_fwd_lds2_kernel(const float* __restrict__ x, const float* wf, const float* bif, const float* bhf,
                                                             const float* wr, const float* bir, const float* bhr, float* __restrict__ out,
                                                             float* __restrict__ gates_save, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KG = (K + 15) / 16;
  constexpr int NFB = H / 16, REM = H % 16, NT = 3 * NFB + (REM ? 1 : 0), ld = KG * 16 + 4;
  static_assert(REM <= 5, "the remainder tile holds 3 x (H % 16) columns");
  static_assert(NFB >= 1 && NFB <= 4, "hidden <= 64");
  const int dir = blockIdx.x & 1, slice = blockIdx.x >> 1, nslices = gridDim.x >> 1;
  const float* __restrict__ w = dir ? wr : wf;
  const float* __restrict__ b1 = dir ? bir : bif;
  const float* __restrict__ b2 = dir ? bhr : bhf;
  float* bias_s = smem + NT * 16 * ld;                       // [3][H]: b_ih + b_hh of gates i, g, o
  // column c of tile ct -> (PyTorch gate block, unit)
  for (int idx = threadIdx.x; idx < NT * 16 * ld; idx += 64 * NW) {
    const int ct = idx / (16 * ld), rem = idx - ct * 16 * ld, n = rem / ld, k = rem - n * ld;
    int gate, unit;
    if (ct < 3 * NFB) { gate = ct % 3; unit = (ct / 3) * 16 + n; }
    else { gate = REM ? n / (REM ? REM : 1) : 3; unit = NFB * 16 + (REM ? n % (REM ? REM : 1) : 0); }
    smem[idx] = (gate < 3 && unit < H && k < K) ? w[(int64_t)((gate == 0 ? 0 : gate + 1) * H + unit) * K + k] : 0.f;      // i, f, g, o -> i, g, o
  }
  for (int idx = threadIdx.x; idx < 3 * H; idx += 64 * NW) {
    const int gate = idx / H, unit = idx - gate * H, pg = (gate == 0 ? 0 : gate + 1) * H + unit;
    bias_s[idx] = b1[pg] + b2[pg];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), j = lane & 15, q = lane >> 4;
  const int64_t ntiles = (rows + 15) >> 4;
  constexpr int gfull = K >> 4;                               // full k-groups; the rest (K & 15) in `tq` quads
  constexpr int tq = ((K & 15) + 3) >> 2;
  const bool vec = (K & 3) == 0 && ((uintptr_t)x & 15) == 0;
  const bool nt_ok = (H & 15) == 0 && gates_save && ((uintptr_t)gates_save & 63) == 0;
  float4 a[gfull > 0 ? gfull : 1];                            // full groups: x[row][16 g + 4 q ..+ 3]
  float at[3] = {0.f, 0.f, 0.f};                              // tail: x[row][16 gfull + 4 c + q]
  auto load_group = [&](int64_t tile, int g) __attribute__((always_inline)) {
    const int64_t r0 = tile << 4;
    const int64_t row = r0 + j < rows ? r0 + j : rows - 1;
    const float* xr = x + row * K;
    const int k0 = 16 * g + 4 * q;
    if (vec) a[g] = *reinterpret_cast<const float4*>(xr + k0);
    else a[g] = make_float4(xr[k0], xr[k0 + 1], xr[k0 + 2], xr[k0 + 3]);
  };
  auto load_tail = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile << 4;
    const int64_t row = r0 + j < rows ? r0 + j : rows - 1;
    const float* xr = x + row * K;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int k = 16 * gfull + 4 * c + q;
      at[c] = (c < tq && k < K) ? xr[k] : 0.f;
    }
  };
  const int64_t tstep = (int64_t)nslices * NW;
  int64_t tile = (int64_t)slice * NW + wave;
  if (tile < ntiles) {
#pragma unroll
    for (int g = 0; g < gfull; ++g) load_group(tile, g);
    load_tail(tile);
  }
  // one pass: NTP column tiles starting at ct0 through the whole K; FETCH: the next tile's rows replace a[g] as soon as group g is
  // done (`next` is always a valid tile: the last one repeats itself -- no branch inside the loop)
  auto pass = [&](auto ntp_tag, auto fetch_tag, int ct0, f32x4* acc, int64_t next) __attribute__((always_inline)) {
    constexpr int NTP = decltype(ntp_tag)::value;
    constexpr bool fetch_next = decltype(fetch_tag)::value;
#pragma unroll
    for (int t = 0; t < NTP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wb = smem + (ct0 * 16 + j) * ld + 4 * q;
#pragma unroll
    for (int g = 0; g < gfull; ++g) {
      {
        float4 b[NTP];
#pragma unroll
        for (int t = 0; t < NTP; ++t) b[t] = *reinterpret_cast<const float4*>(wb + t * 16 * ld + 16 * g);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].w, b[t].w, acc[t], 0, 0, 0);
        if constexpr (fetch_next) load_group(next, g);
      }
    }
    // the partial group: lane (n, q) of B holds W[n][16 gfull + 4 c + q]
    const float* wt = smem + (ct0 * 16 + j) * ld + 16 * gfull + q;
#pragma unroll
    for (int c = 0; c < tq; ++c) {
      float bt[NTP];
#pragma unroll
      for (int t = 0; t < NTP; ++t) bt[t] = wt[t * 16 * ld + 4 * c];
#pragma unroll
      for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[c], bt[t], acc[t], 0, 0, 0);
    }
    if constexpr (fetch_next) load_tail(next);
  };
  for (; tile < ntiles; tile += tstep) {
    // (the weights in LDS are loop-invariant: without this the compiler hoists their reads out of the tile loop and spills 236 registers)
    asm volatile("" ::: "memory");
    const int64_t r0 = tile << 4;
    const int64_t next = tile + tstep < ntiles ? tile + tstep : tile;
    const GBuf ob(out + r0 * 2 * H), gb(gates_save ? gates_save + r0 * 8 * H : out);
    // the cell on one lane's (i, g, o) of unit `unit`, rows 4 q + r
    auto cell = [&](const f32x4& pi, const f32x4& pg, const f32x4& po, int unit) __attribute__((always_inline)) {
      const float bsi = bias_s[unit], bsg = bias_s[H + unit], bso = bias_s[2 * H + unit];
      constexpr int HS = H;
      const int vo_h = (4 * q * 2 * H + dir * H + unit) * 4, vo_g = (4 * q * 8 * HS + dir * 4 * HS + unit) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r0 + 4 * q + r >= rows) continue;
        const float gi = sigmoidf_(pi[r] + bsi), gg = tanhf_(pg[r] + bsg), go = sigmoidf_(po[r] + bso);
        const float tc = tanhf_(gi * gg);
        ob.st(go * tc, vo_h, r * 2 * H * 4);
        if (gates_save) {
          if (nt_ok) {
            auto nt = [&](float v, int so) __attribute__((always_inline)) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), gb.rs, vo_g, so, 2); };
            nt(gi, r * 8 * HS * 4); nt(gg, (r * 8 * HS + HS) * 4); nt(go, (r * 8 * HS + 2 * HS) * 4); nt(tc, (r * 8 * HS + 3 * HS) * 4);
          } else {
            gb.st(gi, vo_g, r * 8 * HS * 4); gb.st(gg, vo_g, (r * 8 * HS + HS) * 4);
            gb.st(go, vo_g, (r * 8 * HS + 2 * HS) * 4); gb.st(tc, vo_g, (r * 8 * HS + 3 * HS) * 4);
          }
        }
      }
    };
    // the remainder tile: columns [i x REM | g x REM | o x REM]: g and o of a unit come from the lanes REM and 2 REM further on
    auto cell_rem = [&](const f32x4& p) __attribute__((always_inline)) {
      f32x4 pg, po;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pg[r] = __shfl_down(p[r], REM ? REM : 1); po[r] = __shfl_down(p[r], REM ? 2 * REM : 1); }
      if (j < REM) cell(p, pg, po, NFB * 16 + j);
    };
    constexpr int FIRST = NFB >= 2 ? 6 : 3;                  // first pass: two full blocks (or the only one)
    constexpr int SECOND = NT - FIRST;                       // second pass: what is left (0 .. 6 tiles)
    f32x4 acc[6];
    pass(std::integral_constant<int, FIRST>{}, std::integral_constant<bool, SECOND == 0>{}, 0, acc, next);
    cell(acc[0], acc[1], acc[2], j);
    if constexpr (FIRST == 6) cell(acc[3], acc[4], acc[5], 16 + j);
    if constexpr (SECOND > 0) {
      pass(std::integral_constant<int, SECOND>{}, std::true_type{}, FIRST, acc, next);
      constexpr int fb2 = NFB - 2;                           // full blocks in the second pass (NFB >= 2 here)
      if constexpr (fb2 >= 1) cell(acc[0], acc[1], acc[2], 32 + j);
      if constexpr (fb2 >= 2) cell(acc[3], acc[4], acc[5], 48 + j);
      if constexpr (REM > 0) cell_rem(acc[3 * fb2]);
    }
  }
}

// ---- the same with the layer's stores staged through LDS (hidden not a multiple of 16: the reference's encoder, 2 x 50): see the
// comment at "pass A" below.  Column tiles in the order the passes take them.
template <int K, int H, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_fwd_lds3_kernel(const float* __restrict__ x, const float* wf, const float* bif, const float* bhf,
                                                             const float* wr, const float* bir, const float* bhr, float* __restrict__ out,
                                                             float* __restrict__ gates_save, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KG = (K + 15) / 16;
  constexpr int NFB = H / 16, REM = H % 16, NT = 3 * NFB + (REM ? 1 : 0), ld = KG * 16 + 4;
  static_assert(REM <= 5, "the remainder tile holds 3 x (H % 16) columns");
  static_assert(NFB >= 1 && NFB <= 4, "hidden <= 64");
  const int dir = blockIdx.x & 1, slice = blockIdx.x >> 1, nslices = gridDim.x >> 1;
  const float* __restrict__ w = dir ? wr : wf;
  const float* __restrict__ b1 = dir ? bir : bif;
  const float* __restrict__ b2 = dir ? bhr : bhf;
  float* bias_s = smem + NT * 16 * ld;                       // [3][H]: b_ih + b_hh of gates i, g, o
  // column c of tile ct -> (PyTorch gate block, unit)
  for (int idx = threadIdx.x; idx < NT * 16 * ld; idx += 64 * NW) {
    const int ct = idx / (16 * ld), rem = idx - ct * 16 * ld, n = rem / ld, k = rem - n * ld;
    int gate, unit;                                          // tile order: (i, ub), (g, ub) for every full block; the remainder tile; (o, ub)
    if (ct < 2 * NFB) { gate = ct & 1; unit = (ct >> 1) * 16 + n; }
    else if (REM && ct == 2 * NFB) { gate = n / (REM ? REM : 1); unit = NFB * 16 + n % (REM ? REM : 1); }
    else { gate = 2; unit = (ct - 2 * NFB - (REM ? 1 : 0)) * 16 + n; }
    smem[idx] = (gate < 3 && unit < H && k < K) ? w[(int64_t)((gate == 0 ? 0 : gate + 1) * H + unit) * K + k] : 0.f;      // i, f, g, o -> i, g, o
  }
  for (int idx = threadIdx.x; idx < 3 * H; idx += 64 * NW) {
    const int gate = idx / H, unit = idx - gate * H, pg = (gate == 0 ? 0 : gate + 1) * H + unit;
    bias_s[idx] = b1[pg] + b2[pg];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), j = lane & 15, q = lane >> 4;
  const int64_t ntiles = (rows + 15) >> 4;
  constexpr int gfull = K >> 4;                               // full k-groups; the rest (K & 15) in `tq` quads
  constexpr int tq = ((K & 15) + 3) >> 2;
  const bool vec = (K & 3) == 0 && ((uintptr_t)x & 15) == 0;
  const bool nt_ok = (H & 15) == 0 && gates_save && ((uintptr_t)gates_save & 63) == 0;
  float4 a[gfull > 0 ? gfull : 1];                            // full groups: x[row][16 g + 4 q ..+ 3]
  float at[3] = {0.f, 0.f, 0.f};                              // tail: x[row][16 gfull + 4 c + q]
  auto load_group = [&](int64_t tile, int g) __attribute__((always_inline)) {
    const int64_t r0 = tile << 4;
    const int64_t row = r0 + j < rows ? r0 + j : rows - 1;
    const float* xr = x + row * K;
    const int k0 = 16 * g + 4 * q;
    if (vec) a[g] = *reinterpret_cast<const float4*>(xr + k0);
    else a[g] = make_float4(xr[k0], xr[k0 + 1], xr[k0 + 2], xr[k0 + 3]);
  };
  auto load_tail = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile << 4;
    const int64_t row = r0 + j < rows ? r0 + j : rows - 1;
    const float* xr = x + row * K;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int k = 16 * gfull + 4 * c + q;
      at[c] = (c < tq && k < K) ? xr[k] : 0.f;
    }
  };
  const int64_t tstep = (int64_t)nslices * NW;
  int64_t tile = (int64_t)slice * NW + wave;
  if (tile < ntiles) {
#pragma unroll
    for (int g = 0; g < gfull; ++g) load_group(tile, g);
    load_tail(tile);
  }
  // one pass: NTP column tiles starting at ct0 through the whole K; FETCH: the next tile's rows replace a[g] as soon as group g is
  // done (`next` is always a valid tile: the last one repeats itself -- no branch inside the loop)
  auto pass = [&](auto ntp_tag, auto fetch_tag, int ct0, f32x4* acc, int64_t next) __attribute__((always_inline)) {
    constexpr int NTP = decltype(ntp_tag)::value;
    constexpr bool fetch_next = decltype(fetch_tag)::value;
#pragma unroll
    for (int t = 0; t < NTP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wb = smem + (ct0 * 16 + j) * ld + 4 * q;
#pragma unroll
    for (int g = 0; g < gfull; ++g) {
      {
        float4 b[NTP];
#pragma unroll
        for (int t = 0; t < NTP; ++t) b[t] = *reinterpret_cast<const float4*>(wb + t * 16 * ld + 16 * g);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].w, b[t].w, acc[t], 0, 0, 0);
        if constexpr (fetch_next) load_group(next, g);
      }
    }
    // the partial group: lane (n, q) of B holds W[n][16 gfull + 4 c + q]
    const float* wt = smem + (ct0 * 16 + j) * ld + 16 * gfull + q;
#pragma unroll
    for (int c = 0; c < tq; ++c) {
      float bt[NTP];
#pragma unroll
      for (int t = 0; t < NTP; ++t) bt[t] = wt[t * 16 * ld + 4 * c];
#pragma unroll
      for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[c], bt[t], acc[t], 0, 0, 0);
    }
    if constexpr (fetch_next) load_tail(next);
  };
  // ---- pass A: gates i and g of every unit (+ the remainder tile, which also carries o of the last H % 16 units); pass B: gate o
  // of the full blocks.  tc = tanh(sigma(i) tanh(g)) needs pass A only and waits in registers for o.  What a pass produces goes
  // through a wave-private LDS slab, eight rows at a time, and leaves as 16-byte stores of whole row pieces: [i | g] = floats
  // [0, 2 H) of the row's (direction) block after pass A, [o | tanh c] = floats [2 H, 4 H) and the h row after pass B.  With 4-byte
  // stores from the accumulator layout (16 units = 64 bytes per row and instruction, at 200-byte gate rows never sector-aligned)
  // nearly every sector was written in pieces: 305 us with the gates against 163 without, 238 with the gate rows padded to 64
  // floats (what-if, round 4); whole sectors also take the non-temporal hint.
  constexpr int NA = 2 * NFB + (REM ? 1 : 0), NB = NFB;       // column tiles of the two passes
  constexpr int SROW = 3 * H + 2;                             // slab row: [o | tc | h] (pass B) or [i | g] (pass A); even, not a multiple of 32
  float* slab = smem + NT * 16 * ld + 3 * H + wave * (8 * SROW);
  const bool g16 = gates_save && (((uintptr_t)gates_save | (uintptr_t)(4 * H * 4)) & 15) == 0;
  for (; tile < ntiles; tile += tstep) {
    // (the weights in LDS are loop-invariant: without this the compiler hoists their reads out of the tile loop and spills 236 registers)
    asm volatile("" ::: "memory");
    const int64_t r0 = tile << 4;
    const int64_t next = tile + tstep < ntiles ? tile + tstep : tile;
    const int nvalid = rows - r0 < 16 ? (int)(rows - r0) : 16;
    // buffer descriptors sized to the tile's valid rows: stores past the end are dropped by the hardware (no per-row branches)
    const __amdgpu_buffer_rsrc_t gsr = __builtin_amdgcn_make_buffer_rsrc(gates_save ? gates_save + (r0 * 2 + dir) * 4 * H : out, 0,
                                                                          gates_save ? (nvalid * 8 * H - dir * 4 * H) * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t osr = __builtin_amdgcn_make_buffer_rsrc(out + r0 * 2 * H + dir * H, 0, (nvalid * 2 * H - dir * H) * 4, 0x00020000);
    f32x4 acc[NA > NB ? NA : NB];
    float tck[NFB + 1][4], gok[4];                            // tanh(c) of every block's rows 4 q + r; sigma(o) of the remainder units
    const int lrow = 4 * (q & 1);                             // this lane's first row inside its half of the tile
    // ---- pass A
    pass(std::integral_constant<int, NA>{}, std::false_type{}, 0, acc, next);
    float gi_[NFB + 1][4], gg_[NFB + 1][4];
#pragma unroll
    for (int ub = 0; ub < NFB; ++ub) {
      const int unit = 16 * ub + j;
      const float bsi = bias_s[unit], bsg = bias_s[H + unit];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi_[ub][r] = sigmoidf_(acc[2 * ub][r] + bsi); gg_[ub][r] = tanhf_(acc[2 * ub + 1][r] + bsg);
        tck[ub][r] = tanhf_(gi_[ub][r] * gg_[ub][r]);
      }
    }
    if constexpr (REM > 0) {
      const int unit = 16 * NFB + (j < REM ? j : 0);
      const float bsi = bias_s[unit], bsg = bias_s[H + unit], bso = bias_s[2 * H + unit];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = acc[2 * NFB][r];
        const float pg = __shfl_down(p, REM), po = __shfl_down(p, 2 * REM);
        gi_[NFB][r] = sigmoidf_(p + bsi); gg_[NFB][r] = tanhf_(pg + bsg); gok[r] = sigmoidf_(po + bso);
        tck[NFB][r] = tanhf_(gi_[NFB][r] * gg_[NFB][r]);
      }
    }
    if (gates_save) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if ((q >> 1) == half) {
#pragma unroll
          for (int ub = 0; ub < NFB; ++ub)
#pragma unroll
            for (int r = 0; r < 4; ++r) { slab[(lrow + r) * SROW + 16 * ub + j] = gi_[ub][r]; slab[(lrow + r) * SROW + H + 16 * ub + j] = gg_[ub][r]; }
          if constexpr (REM > 0) {
            if (j < REM) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { slab[(lrow + r) * SROW + 16 * NFB + j] = gi_[NFB][r]; slab[(lrow + r) * SROW + H + 16 * NFB + j] = gg_[NFB][r]; }
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // rows 8 half .. + 7, floats [0, 2 H) of the (row, direction) block: 2 H / 4 pieces of 16 bytes per row
        constexpr int PPR = 2 * H / 4;
        for (int p = lane; p < 8 * PPR; p += 64) {
          const int rr = p / PPR, c4 = p - rr * PPR;
          const float* src = slab + rr * SROW + 4 * c4;
          const float4 v = make_float4(src[0], src[1], src[2], src[3]);
          const int off = ((8 * half + rr) * 8 * H + 4 * c4) * 4;
          if (g16) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_u32x4, v), gsr, off, 0, 2);
          else { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x), gsr, off, 0, 0); __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.y), gsr, off + 4, 0, 0);
                 __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.z), gsr, off + 8, 0, 0); __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.w), gsr, off + 12, 0, 0); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the slab is rewritten by the other half / by pass B)
      }
    }
    // ---- pass B: o of the full blocks; the next tile's x rows arrive under it
    pass(std::integral_constant<int, NB>{}, std::true_type{}, NA, acc, next);
    float go_[NFB + 1][4];
#pragma unroll
    for (int ub = 0; ub < NFB; ++ub) {
      const float bso = bias_s[2 * H + 16 * ub + j];
#pragma unroll
      for (int r = 0; r < 4; ++r) go_[ub][r] = sigmoidf_(acc[ub][r] + bso);
    }
    if constexpr (REM > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) go_[NFB][r] = gok[r];
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((q >> 1) == half) {
#pragma unroll
        for (int ub = 0; ub < NFB + (REM ? 1 : 0); ++ub) {
          if (ub < NFB || j < REM) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* row = slab + (lrow + r) * SROW + 16 * ub + j;
              row[0] = go_[ub][r]; row[H] = tck[ub][r]; row[2 * H] = go_[ub][r] * tck[ub][r];
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (gates_save) {
        constexpr int PPR = 2 * H / 4;
        for (int p = lane; p < 8 * PPR; p += 64) {
          const int rr = p / PPR, c4 = p - rr * PPR;
          const float* src = slab + rr * SROW + 4 * c4;
          const float4 v = make_float4(src[0], src[1], src[2], src[3]);
          const int off = ((8 * half + rr) * 8 * H + 2 * H + 4 * c4) * 4;
          if (g16 && (H % 2) == 0 && ((2 * H * 4) & 15) == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_u32x4, v), gsr, off, 0, 2);
          else { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x), gsr, off, 0, 0); __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.y), gsr, off + 4, 0, 0);
                 __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.z), gsr, off + 8, 0, 0); __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.w), gsr, off + 12, 0, 0); }
        }
      }
      {                                                         // the h rows: H floats per (row, direction), pieces of 8 bytes
        constexpr int PPR2 = H / 2;
        for (int p = lane; p < 8 * PPR2; p += 64) {
          const int rr = p / PPR2, c2 = p - rr * PPR2;
          const float* src = slab + rr * SROW + 2 * H + 2 * c2;
          const int off = ((8 * half + rr) * 2 * H + 2 * c2) * 4;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wl_u32x2, make_float2(src[0], src[1])), osr, off, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

__global__ __launch_bounds__(THREADS) void lstm_bwd_kernel(const float* wf, const float* wr, const float* __restrict__ gates_saved,
                                                            const float* __restrict__ gout, float* __restrict__ ggates,
                                                            float* __restrict__ gx, int64_t rows, int K, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = ld_of(K), ldg = ld_of(6 * H), ldh = ld_of(2 * H);
  float* dhs = smem;
  float* dgs = dhs + 16 * ldh;
  float* xs = dgs + 16 * ldg;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  tile_load(dhs, ldh, gout + r0 * 2 * H, 2 * H, 16, 2 * H, valid);
  __syncthreads();
  lstm_cell_bwd_tile(dhs, ldh, gates_saved + r0 * 8 * H, H, 16, dgs, ldg, valid);
  __syncthreads();
  if (ggates) {   // (rows, 2, 4H) in PyTorch gate order, f block zero
    for (int i = threadIdx.x; i < valid * 8 * H; i += THREADS) {
      int r = i / (8 * H), c = i - r * 8 * H;
      int d = c / (4 * H), g = (c - d * 4 * H) / H, jj = c - d * 4 * H - g * H;
      float v = 0.f;
      if (g != 1) v = dgs[r * ldg + d * 3 * H + (g == 0 ? 0 : g - 1) * H + jj];
      ggates[(r0 + r) * 8 * H + c] = v;
    }
  }
  if (gx) {
    gemm_nn<1>(dgs, ldg, 0, wf, K, 3 * H, lstm_gate_map(H), K, xs, ldx, false);
    gemm_nn<1>(dgs, ldg, 3 * H, wr, K, 3 * H, lstm_gate_map(H), K, xs, ldx, true);
    __syncthreads();
    tile_store(gx + r0 * K, K, xs, ldx, 16, K, valid);
  }
}

// ------------------------------------------------------------------------------------------ networks
struct NetLds {          // float offsets into dynamic LDS for the network kernels
  int xs, zs, bufA, bufB, crit, wst, total;
  int ldS, bufFloats;
};
__host__ __device__ inline NetLds net_lds(int S, int MT) {
  NetLds n;
  n.ldS = lds_stride(S);
  int rows = MT * 16;
  int per_row = n.ldS > 6 * DEC_H + 4 ? n.ldS : 6 * DEC_H + 4;
  n.bufFloats = rows * per_row;
  int o = 0;
  n.xs = o; o += 16 * n.ldS;
  n.zs = o; o += rows * LP;
  n.bufA = o; o += n.bufFloats;
  n.bufB = o; o += n.bufFloats;
  n.crit = o; o += CRITIC_LDS_FLOATS;
  n.wst = o; o += WST;
  n.total = o;
  return n;
}

__global__ __launch_bounds__(THREADS) void encoder_fwd_kernel(const float* __restrict__ P, const float* __restrict__ x,
                                                               float* __restrict__ out, int64_t rows, int S, int L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetLds nl = net_lds(S, 1);
  const EncLayout el = enc_layout(S, L);
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  tile_load(smem + nl.xs, nl.ldS, x + r0 * S, S, 16, S, valid);
  __syncthreads();
  encoder_fwd_tile(smem + nl.xs, nl.ldS, S, L, P, el, smem + nl.bufA, ENC_LDG, smem + nl.bufB, ENC_LDH,
                   smem + nl.zs, nullptr, nullptr, valid, smem + nl.wst);
  tile_store(out + r0 * L, L, smem + nl.zs, LP, 16, L, valid);
}

template <int MT>
__global__ __launch_bounds__(THREADS) void decoder_fwd_kernel(const float* __restrict__ P, const float* __restrict__ z,
                                                               float* __restrict__ hyper, float* __restrict__ eucl,
                                                               int64_t rows, int S, int L, int hyperbolic, hypad_dropout dp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = MT * 16;
  const NetLds nl = net_lds(S, MT);
  const DecLayout dl = dec_layout(S, L, hyperbolic);
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int valid = (int)min((int64_t)R, rows - r0);
  float* zs = smem + nl.zs; float* bufA = smem + nl.bufA; float* bufB = smem + nl.bufB;
  tile_load(zs, LP, z + r0 * L, L, R, L, valid);
  __syncthreads();
  DropSrc drop = make_drop(dp, (int)rows, RS_DROP_DEC0, 0.2f);
  DecSave none{16, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const int base_row = (int)r0;
  decoder_trunk_fwd_tile<MT>(zs, L, S, P, dl, bufA, bufB, nl.ldS, drop, [base_row](int r) { return base_row + r; }, none, valid,
                             smem + nl.wst);
  if (eucl) tile_store(eucl + r0 * S, S, bufA, nl.ldS, R, S, valid);
  if (hyperbolic && hyper) {
    gemm_nt<MT>(bufA, nl.ldS, P + dl.head_w, S, S, S, identity_map(), nullptr, nullptr, bufB, nl.ldS, 0, smem + nl.wst);
    __syncthreads();
    head_rows_tile(bufB, nl.ldS, R, S, P + dl.head_b);
    __syncthreads();
    tile_store(hyper + r0 * S, S, bufB, nl.ldS, R, S, valid);
  }
}

__global__ __launch_bounds__(THREADS) void critic_fwd_kernel(const float* __restrict__ P, const float* __restrict__ x,
                                                              float* __restrict__ out, int64_t rows, int in_dim, int L,
                                                              int nh, float p, hypad_dropout dp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = pad4(in_dim) + 4;
  float* xs = smem;
  CriticLds cs = critic_lds(smem + 16 * ldx);
  float* wst = smem + 16 * ldx + CRITIC_LDS_FLOATS;
  // (only the scalar fields: critic_fwd_tile takes its offsets from the closed forms wof / bof -- filling the w[] / b[] tables with a
  // run-time layer count put the struct into 64 bytes of scratch per lane)
  CriticLayout cl; cl.nh = nh; cl.in_dim = in_dim; cl.L = L; cl.p_drop = p; cl.total = 0;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  tile_load(xs, ldx, x + r0 * in_dim, in_dim, 16, in_dim, valid);
  __syncthreads();
  DropSrc drop = make_drop(dp, (int)rows, RS_DROP_CRITIC, p);
  critic_fwd_tile(xs, ldx, P, cl, L, cs, drop, (int)r0, wst);
  if (threadIdx.x < valid) out[r0 + threadIdx.x] = cs.out[threadIdx.x];
}

// mobius_linear forward: u = x W^T (saved) ; out = head(u)
__global__ __launch_bounds__(THREADS) void mobius_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     float* __restrict__ u_save, int64_t rows, int K, int N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = ld_of(K), ldy = ld_of(N);
  float* xs = smem;
  float* ys = xs + 16 * ldx;
  float* wst = ys + 16 * ldy;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  tile_load(xs, ldx, x + r0 * K, K, 16, K, valid);
  __syncthreads();
  gemm_nt<1>(xs, ldx, w, K, K, N, identity_map(), nullptr, nullptr, ys, ldy, 0, wst);
  __syncthreads();
  if (u_save) tile_store(u_save + r0 * N, N, ys, ldy, 16, N, valid);
  __syncthreads();
  head_rows_tile(ys, ldy, 16, N, bias);
  __syncthreads();
  tile_store(out + r0 * N, N, ys, ldy, 16, N, valid);
}

// fused scoring forward (anomaly_detection.py:67-113 + utils/anomaly_detection_utils.py:58-66)
__global__ __launch_bounds__(THREADS) void score_forward_kernel(const float* __restrict__ PE, const float* __restrict__ PD,
                                                                 const float* __restrict__ PC, const float* __restrict__ x,
                                                                 float* __restrict__ hyper, float* __restrict__ eucl,
                                                                 float* __restrict__ hyper_real, float* __restrict__ critic,
                                                                 float* __restrict__ rowdist, int64_t rows, int S, int L,
                                                                 int hyperbolic) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetLds nl = net_lds(S, 2);      // bufA/bufB sized for 32 rows: decoder rows 0..15, head(x) rows 16..31
  const EncLayout el = enc_layout(S, L);
  const DecLayout dl = dec_layout(S, L, hyperbolic);
  const CriticLayout cl = cx_layout(S, L);
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int valid = (int)min((int64_t)16, rows - r0);
  float* xs = smem + nl.xs; float* zs = smem + nl.zs; float* bufA = smem + nl.bufA; float* bufB = smem + nl.bufB;
  CriticLds cs = critic_lds(smem + nl.crit);
  tile_load(xs, nl.ldS, x + r0 * S, S, 16, S, valid);
  __syncthreads();
  if (critic) {
    critic_fwd_tile(xs, nl.ldS, PC, cl, L, cs, no_drop(), 0, smem + nl.wst);
    if (threadIdx.x < valid) critic[r0 + threadIdx.x] = cs.out[threadIdx.x];
  }
  encoder_fwd_tile(xs, nl.ldS, S, L, PE, el, bufA, ENC_LDG, bufB, ENC_LDH, zs, nullptr, nullptr, valid, smem + nl.wst);
  DecSave none{16, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  decoder_trunk_fwd_tile<1>(zs, L, S, PD, dl, bufA, bufB, nl.ldS, no_drop(), [](int r) { return r; }, none, valid, smem + nl.wst);
  if (eucl) tile_store(eucl + r0 * S, S, bufA, nl.ldS, 16, S, valid);
  if (hyperbolic) {
    // rows 16..31 of bufA <- x: one head GEMM serves decoder output and the real window
    for (int i = threadIdx.x; i < 16 * nl.ldS; i += THREADS) bufA[16 * nl.ldS + i] = xs[i];
    __syncthreads();
    gemm_nt<2>(bufA, nl.ldS, PD + dl.head_w, S, S, S, identity_map(), nullptr, nullptr, bufB, nl.ldS, 0, smem + nl.wst);
    __syncthreads();
    head_rows_tile(bufB, nl.ldS, 32, S, PD + dl.head_b);
    __syncthreads();
    if (hyper) tile_store(hyper + r0 * S, S, bufB, nl.ldS, 16, S, valid);
    if (hyper_real) tile_store(hyper_real + r0 * S, S, bufB + 16 * nl.ldS, nl.ldS, 16, S, valid);
    if (rowdist) {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (int r = wave; r < valid; r += 4) {
        // (pred = real window on the ball, true = reconstruction): anomaly_detection_utils.py:58-65
        float d = rowdist_row(row_load(bufB + (16 + r) * nl.ldS, S, lane), row_load(bufB + r * nl.ldS, S, lane));
        if (lane == 0) rowdist[r0 + r] = d;
      }
    }
  }
}

inline int tiles16(int64_t rows) { return (int)((rows + 15) / 16); }

}  // namespace

extern "C" {

int hypad_linear_act_fwd(const float* x, const float* w, const float* b, float* y, int64_t rows, int K, int N, int act,
                         hypad_stream_t s) {
  if (!x || !w || !y || rows < 0 || K <= 0 || N <= 0) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  size_t lds = (size_t)(16 * (ld_of(K) + ld_of(N)) + WST) * sizeof(float);
  if (lds > 150 * 1024) return HYPAD_EUNSUPPORTED;
  hipError_t e = allow_lds((const void*)linear_fwd_kernel, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, x, w, b, y, rows, K, N, act);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_linear_act_bwd(const float* x, const float* w, const float* y, const float* gy, float* gx, float* gw, float* gb,
                         float* gpre, int64_t rows, int K, int N, int act, hypad_stream_t s) {
  if (!w || !gy || rows < 0 || K <= 0 || N <= 0) return HYPAD_EINVAL;
  if (act != HYPAD_ACT_NONE && !y) return HYPAD_EINVAL;
  if ((gw || gb) && (!gpre || !x)) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  size_t lds = (size_t)16 * (ld_of(K) + ld_of(N)) * sizeof(float);
  if (lds > 150 * 1024) return HYPAD_EUNSUPPORTED;
  hipError_t e = allow_lds((const void*)linear_bwd_data_kernel, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, w, y, gy, gpre, gx,
                     rows, K, N, act);
  HYPAD_CHECK_LAUNCH();
  if (gw || gb) {
    int tiles = ((N + 15) / 16) * ((K + 15) / 16);
    int blocks = (tiles + 3) / 4;
    int bb = (N + THREADS - 1) / THREADS;
    if (bb > blocks) blocks = bb;
    if (!gw) return HYPAD_EINVAL;
    hipLaunchKernelGGL(outer_sum_kernel, dim3(blocks), dim3(THREADS), 0, (hipStream_t)s, gpre, N, x, K, gw, gb, rows, N, K);
    HYPAD_CHECK_LAUNCH();
  }
  return HYPAD_OK;
}

int hypad_lstm_bidir_fwd(const float* x, const float* wf, const float* bif, const float* bhf, const float* wr,
                         const float* bir, const float* bhr, float* out, float* gates_save, int64_t rows, int K, int H,
                         hypad_stream_t s) {
  if (!x || !wf || !bif || !bhf || !wr || !bir || !bhr || !out || rows < 0 || K <= 0 || H <= 0) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  // many rows: the weights-stationary form (one direction's W_ih in LDS per workgroup); HYPAD_LSTM_LDS=0 keeps the streamed form
  static const int lds_form = HYPAD_TUNE_INT("HYPAD_LSTM_LDS", 1);
  if (lds_form && rows >= 2048 && ((H == 50 && K == 100) || (H == 64 && (K == 128 || K == 50)))) {
    // the reference's three layer shapes (encoder 100 -> 2 x 50; decoder 50 -> 2 x 64 and 128 -> 2 x 64): lstm_fwd_lds2_kernel
    const int64_t ntiles = (rows + 15) >> 4;
#ifndef HYPAD_LSTM_NW
#define HYPAD_LSTM_NW 16
#endif
    constexpr int nw = HYPAD_LSTM_NW;
    // one workgroup per CU and direction slice; NWC waves each (16, except 50 -> 2 x 64 with saved gates: eight waves, 132 -> 117 us per
    // 200 000 rows -- round 6 sweep of waves x slices, scripts/time_lstm.py; every other shape is fastest at sixteen)
    auto slices = [&](int nwc) { int n = (int)((ntiles + nwc - 1) / nwc); return n > 128 ? 128 : n; };
#define HYPAD_LSTM_LDS2_LAUNCH(KC, HC, NWC)                                                                                      \
    do {                                                                                                                         \
      constexpr int NTC = 3 * (HC / 16) + ((HC % 16) ? 1 : 0);                                                                   \
      const size_t lw2 = (size_t)(NTC * 16 * (((KC + 15) / 16) * 16 + 4) + 3 * HC) * sizeof(float);                              \
      hipError_t e2 = allow_lds((const void*)lstm_fwd_lds2_kernel<KC, HC, NWC>, lw2);                                            \
      if (e2 != hipSuccess) return (int)e2;                                                                                      \
      hipLaunchKernelGGL((lstm_fwd_lds2_kernel<KC, HC, NWC>), dim3(2 * slices(NWC)), dim3(64 * NWC), lw2, (hipStream_t)s, x, wf, bif, bhf, wr, bir, bhr, out, \
                         gates_save, rows);                                                                                      \
    } while (0)
    if (H == 50 && !gates_save) HYPAD_LSTM_LDS2_LAUNCH(100, 50, nw);      // (forward only: 163 us per 200 000 rows, 171 through the slab)
    else if (H == 50) {
      constexpr int KC = 100, HC = 50, NTC = 3 * (HC / 16) + 1;
      const size_t lw3 = (size_t)(NTC * 16 * (((KC + 15) / 16) * 16 + 4) + 3 * HC + nw * 8 * (3 * HC + 2)) * sizeof(float);
      hipError_t e3 = allow_lds((const void*)lstm_fwd_lds3_kernel<KC, HC, nw>, lw3);
      if (e3 != hipSuccess) return (int)e3;
      hipLaunchKernelGGL((lstm_fwd_lds3_kernel<KC, HC, nw>), dim3(2 * slices(nw)), dim3(64 * nw), lw3, (hipStream_t)s, x, wf, bif, bhf, wr, bir, bhr, out, gates_save, rows);
    }
    else if (K == 128) HYPAD_LSTM_LDS2_LAUNCH(128, 64, nw);
    else if (gates_save && nw == 16) HYPAD_LSTM_LDS2_LAUNCH(50, 64, 8);
    else HYPAD_LSTM_LDS2_LAUNCH(50, 64, nw);
#undef HYPAD_LSTM_LDS2_LAUNCH
    HYPAD_CHECK_LAUNCH();
    return HYPAD_OK;
  }
  if (lds_form && rows >= 2048 && H <= 64 && K <= 128) {
    const int KG = (K + 15) >> 4, Hp = (H + 15) & ~15;
    const size_t lw = (size_t)3 * Hp * (KG * 16 + 4) * sizeof(float);
    const int64_t ntiles = (rows + 15) >> 4;
    static const int nw_env = HYPAD_TUNE_INT("HYPAD_LSTM_WAVES", 0);
    // (16 waves: 357 -> 349 us at 100 -> 2 x 50, 356 -> 321 at 128 -> 2 x 64, gates saved, 200 000 rows -- for inputs up to 96 wide: at seven and
    // eight k-groups the 128-register budget of sixteen waves spills 2 / 10 registers, so those widths run eight waves, spill-free)
    const int nw = (nw_env == 8 || KG >= 7) ? 8 : 16;
    int nslices = (int)((ntiles + nw - 1) / nw);
    if (nslices > 128) nslices = 128;
    const dim3 grid(2 * nslices);
#define HYPAD_LSTM_LDS_LAUNCH(KGC)                                                                                               \
    do {                                                                                                                         \
      if constexpr (KGC < 7) if (nw == 16) {                                                                                     \
        hipError_t e2 = allow_lds((const void*)lstm_fwd_lds_kernel<KGC, 16>, lw);                                                \
        if (e2 != hipSuccess) return (int)e2;                                                                                    \
        hipLaunchKernelGGL((lstm_fwd_lds_kernel<KGC, 16>), grid, dim3(1024), lw, (hipStream_t)s, x, wf, bif, bhf, wr, bir, bhr, out, \
                           gates_save, rows, K, H);                                                                              \
        break;                                                                                                                   \
      }                                                                                                                          \
      {                                                                                                                          \
        hipError_t e2 = allow_lds((const void*)lstm_fwd_lds_kernel<KGC, 8>, lw);                                                 \
        if (e2 != hipSuccess) return (int)e2;                                                                                    \
        hipLaunchKernelGGL((lstm_fwd_lds_kernel<KGC, 8>), grid, dim3(512), lw, (hipStream_t)s, x, wf, bif, bhf, wr, bir, bhr, out, \
                           gates_save, rows, K, H);                                                                              \
      }                                                                                                                          \
    } while (0)
    switch (KG) {
      case 1: HYPAD_LSTM_LDS_LAUNCH(1); break; case 2: HYPAD_LSTM_LDS_LAUNCH(2); break; case 3: HYPAD_LSTM_LDS_LAUNCH(3); break;
      case 4: HYPAD_LSTM_LDS_LAUNCH(4); break; case 5: HYPAD_LSTM_LDS_LAUNCH(5); break; case 6: HYPAD_LSTM_LDS_LAUNCH(6); break;
      case 7: HYPAD_LSTM_LDS_LAUNCH(7); break; default: HYPAD_LSTM_LDS_LAUNCH(8); break;
    }
#undef HYPAD_LSTM_LDS_LAUNCH
    HYPAD_CHECK_LAUNCH();
    return HYPAD_OK;
  }
  size_t lds = (size_t)(16 * (ld_of(K) + ld_of(6 * H) + ld_of(2 * H)) + WST) * sizeof(float);
  if (lds > 150 * 1024) return HYPAD_EUNSUPPORTED;
  hipError_t e = allow_lds((const void*)lstm_fwd_kernel, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, x, wf, bif, bhf, wr, bir, bhr,
                     out, gates_save, rows, K, H);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_lstm_bidir_bwd(const float* wf, const float* wr, const float* gates_saved, const float* gout, float* ggates,
                         float* gx, int64_t rows, int K, int H, hypad_stream_t s) {
  if (!wf || !wr || !gates_saved || !gout || rows < 0 || K <= 0 || H <= 0) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  size_t lds = (size_t)16 * (ld_of(K) + ld_of(6 * H) + ld_of(2 * H)) * sizeof(float);
  if (lds > 150 * 1024) return HYPAD_EUNSUPPORTED;
  hipError_t e = allow_lds((const void*)lstm_bwd_kernel, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, wf, wr, gates_saved, gout,
                     ggates, gx, rows, K, H);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

static int check_net(int S, int L) {
  if (S <= 0 || L <= 0) return HYPAD_EINVAL;
  if (S > MAX_S || L > MAX_L) return HYPAD_EUNSUPPORTED;
  return HYPAD_OK;
}

int hypad_encoder_fwd(const float* P, const float* x, float* out, int64_t rows, int S, int L, hypad_stream_t s) {
  int rc = check_net(S, L);
  if (rc) return rc;
  if (!P || !x || !out || rows < 0) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  size_t lds = (size_t)net_lds(S, 1).total * sizeof(float);
  hipError_t e = allow_lds((const void*)encoder_fwd_kernel, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(encoder_fwd_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, P, x, out, rows, S, L);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_decoder_fwd(const float* P, const float* z, float* hyper, float* eucl, int64_t rows, int S, int L, int hyperbolic,
                      const hypad_dropout* drop, hypad_stream_t s) {
  int rc = check_net(S, L);
  if (rc) return rc;
  if (!P || !z || rows < 0 || (!hyper && !eucl)) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  hypad_dropout dp = drop ? *drop : hypad_dropout{0, nullptr, 0, 0};
  if (rows >= 32 * 512) {
    size_t lds = (size_t)net_lds(S, 2).total * sizeof(float);
    hipError_t e = allow_lds((const void*)decoder_fwd_kernel<2>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(decoder_fwd_kernel<2>, dim3((int)((rows + 31) / 32)), dim3(THREADS), lds, (hipStream_t)s, P, z, hyper,
                       eucl, rows, S, L, hyperbolic, dp);
  } else {
    size_t lds = (size_t)net_lds(S, 1).total * sizeof(float);
    hipError_t e = allow_lds((const void*)decoder_fwd_kernel<1>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(decoder_fwd_kernel<1>, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, P, z, hyper, eucl,
                       rows, S, L, hyperbolic, dp);
  }
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

static int critic_fwd(const float* P, const float* x, float* out, int64_t rows, int in_dim, int L, int nh, float p,
                      const hypad_dropout* drop, hypad_stream_t s) {
  if (!P || !x || !out || rows < 0) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  hypad_dropout dp = drop ? *drop : hypad_dropout{0, nullptr, 0, 0};
  size_t lds = (size_t)(16 * (pad4(in_dim) + 4) + CRITIC_LDS_FLOATS + WST) * sizeof(float);
  hipLaunchKernelGGL(critic_fwd_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, P, x, out, rows, in_dim, L,
                     nh, p, dp);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_critic_x_fwd(const float* P, const float* x, float* out, int64_t rows, int S, int L, const hypad_dropout* drop,
                       hypad_stream_t s) {
  int rc = check_net(S, L);
  if (rc) return rc;
  return critic_fwd(P, x, out, rows, S, L, 4, 0.25f, drop, s);
}
int hypad_critic_z_fwd(const float* P, const float* z, float* out, int64_t rows, int L, const hypad_dropout* drop,
                       hypad_stream_t s) {
  int rc = check_net(1, L);
  if (rc) return rc;
  return critic_fwd(P, z, out, rows, L, L, 2, 0.2f, drop, s);
}

size_t hypad_mobius_linear_workspace_bytes(int64_t rows, int out_dim) {
  return (size_t)rows * out_dim * 2 * sizeof(float);   // grad_u and per-row bias gradients
}
int hypad_mobius_linear_fwd(const float* x, const float* w, const float* bias, float* out, float* u_save, int64_t rows,
                            int K, int N, hypad_stream_t s) {
  if (!x || !w || !bias || !out || rows < 0 || K <= 0 || N <= 0) return HYPAD_EINVAL;
  if (N > 64 * MAX_EPL) return HYPAD_EUNSUPPORTED;
  if (rows == 0) return HYPAD_OK;
  size_t lds = (size_t)(16 * (ld_of(K) + ld_of(N)) + WST) * sizeof(float);
  if (lds > 150 * 1024) return HYPAD_EUNSUPPORTED;
  hipError_t e = allow_lds((const void*)mobius_linear_fwd_kernel, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(mobius_linear_fwd_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, x, w, bias, out,
                     u_save, rows, K, N);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_mobius_linear_bwd(const float* x, const float* w, const float* bias, const float* u_saved, const float* go,
                            float* gx, float* gw, float* gbias, void* workspace, size_t workspace_bytes, int64_t rows,
                            int K, int N, hypad_stream_t s) {
  if (!x || !w || !bias || !u_saved || !go || rows < 0 || K <= 0 || N <= 0) return HYPAD_EINVAL;
  if (!workspace || workspace_bytes < hypad_mobius_linear_workspace_bytes(rows, N)) return HYPAD_EWORKSPACE;
  if (rows == 0) return HYPAD_OK;
  float* gu = (float*)workspace;
  float* gb_rows = gu + (size_t)rows * N;
  int rc = hypad_mobius_head_bwd(u_saved, bias, go, gu, gb_rows, rows, N, s);
  if (rc) return rc;
  if (gbias) {
    rc = hypad_column_sum(gb_rows, gbias, rows, N, s);
    if (rc) return rc;
  }
  // grad_x = gu W ; grad_w = gu^T x
  return hypad_linear_act_bwd(x, w, nullptr, gu, gx, gw, nullptr, gw ? gb_rows /*scratch for grad_pre*/ : nullptr, rows, K, N,
                              HYPAD_ACT_NONE, s);
}

int hypad_score_forward(const float* enc, const float* dec, const float* cx, const float* x, float* hyper, float* eucl,
                        float* hyper_real, float* critic, float* rowdist, int64_t rows, int S, int L, int hyperbolic,
                        hypad_stream_t s) {
  int rc = check_net(S, L);
  if (rc) return rc;
  if (!enc || !dec || !x || rows < 0 || (critic && !cx)) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  size_t lds = (size_t)net_lds(S, 2).total * sizeof(float);
  hipError_t e = allow_lds((const void*)score_forward_kernel, lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(score_forward_kernel, dim3(tiles16(rows)), dim3(THREADS), lds, (hipStream_t)s, enc, dec, cx, x, hyper,
                     eucl, hyper_real, critic, rowdist, rows, S, L, hyperbolic);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

}  // extern "C"
