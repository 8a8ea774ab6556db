// One bidirectional LSTM layer over T time steps: torch.nn.LSTM(in, H, num_layers=1, bidirectional=True) semantics with
// (T, rows, in) input (models/tadgan.py:15-20, :35-38 build exactly such layers; the reference only ever drives them with T = 1,
// SURVEY.md D2 -- this is the general form BASELINE.json's north_star describes).
//
// Two kernels.  (1) The input projection x_t W_ih^T + b_ih of ALL time steps is one dense GEMM over T * rows rows -- the only
// part of an LSTM that is a real dense contraction -- on the library's row-tile MFMA GEMM (hypad_linear_act_fwd).  (2) The
// recurrence is a PERSISTENT kernel: one workgroup per (16-row tile, direction) stays resident for all T steps with
//   * W_hh in LDS (zero-padded MFMA-friendly rows: 4 x Hp x (Hp + 4) floats, 70 KB at H = 64), loaded once;
//   * the hidden tile h_t in LDS (double-buffered: step t + 1's A operand), the cell state c_t in registers;
//   * the four gates of one hidden unit on ONE lane: wave u owns units [16 u, 16 u + 16) of i, f, g and o -- four 16 x 16
//     accumulators whose (row, unit) layouts coincide -- so the cell update needs no cross-lane traffic at all;
//   * one workgroup barrier per time step; the next step's pre-activations are requested before the recurrent MFMAs.
// Backward (round 6: back-propagation through time, so that the layer can TRAIN at any T like the nn.LSTM modules of
// models/tadgan.py:15-27, 35-38 under autograd): the training form of the forward saves, per (step, row, direction), the gate
// activations and the cell state ([i | f | g | o | c], 5 H floats); lstm_seq_bwd_kernel walks the steps in reverse, persistent like
// the forward -- W_hh TRANSPOSED in LDS, the carried dh / dc in registers, the pre-activation gradient tile of a step through a
// double-buffered LDS tile into the MFMA A layout for `dh_prev = da . W_hh` (wave u produces dh of ITS 16 units: the accumulator
// layout is again the elementwise layout, one barrier per step) -- and writes the pre-activation gradients of all steps; the
// parameter and input gradients are then dense contractions over T * rows rows on the library's linear backward
// (hypad_linear_act_bwd: dX = da W_ih, dW_ih = da^T x, db = colsum da; dW_hh = da^T h_prev with h_prev gathered from `out` / h0).
#include <hip/hip_runtime.h>

#include "../../include/hypad.h"
#include "device_utils.h"

using namespace hypad;

namespace {

constexpr int TS = 256;                                    // 4 waves
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct SeqArgs {
  const float* pre[2];                                     // (T * rows, 4 H) per direction: x_t W_ih^T + b_ih
  const float* whh[2]; const float* bhh[2];
  const float* h0; const float* c0;                        // (2, rows, H) or null
  float* out; float* hn; float* cn;                        // (T, rows, 2 H); (2, rows, H) or null
  float* saved;                                            // (T, rows, 2, 5, H) = [i | f | g | o | c] per (step, row, direction), or null
  int T; int64_t rows; int H;
};

template <int HP>
__global__ __launch_bounds__(TS) void lstm_seq_kernel(SeqArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDW = HP + 4;
  float* Wl = smem;                                        // [4][HP][LDW]
  float* hs = Wl + 4 * HP * LDW;                           // [2][16][LDW]
  const int dir = blockIdx.y, H = a.H;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int j = lane & 15, q = lane >> 4;
  const float* whh = a.whh[dir];
  for (int i = threadIdx.x; i < 4 * HP * HP; i += TS) {
    const int g = i / (HP * HP), rem = i - g * HP * HP, n = rem / HP, k = rem - n * HP;
    Wl[(g * HP + n) * LDW + k] = (n < H && k < H) ? whh[(size_t)(g * H + n) * H + k] : 0.f;
  }
  for (int i = threadIdx.x; i < 2 * 16 * LDW; i += TS) hs[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * HP; i += TS) {
    const int r = i / HP, k = i - r * HP;
    if (a.h0 && r0 + r < a.rows && k < H) hs[r * LDW + k] = a.h0[((size_t)dir * a.rows + r0 + r) * H + k];
  }
  const int unit = 16 * wave + j;
  const bool active = 16 * wave < HP, uok = unit < H;
  float c[4], bh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = r0 + 4 * q + r;
    c[r] = (a.c0 && uok && row < a.rows) ? a.c0[((size_t)dir * a.rows + row) * H + unit] : 0.f;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = uok ? a.bhh[dir][g * H + unit] : 0.f;
  float hlast[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const float* pre = a.pre[dir];
  auto load_pre = [&](int t, float (&p)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = r0 + 4 * q + r;
        p[g][r] = (uok && row < a.rows) ? pre[((size_t)t * a.rows + row) * 4 * H + g * H + unit] : 0.f;
      }
  };
  float pcur[4][4];
  if (a.T > 0) load_pre(dir ? a.T - 1 : 0, pcur);
  for (int step = 0; step < a.T; ++step) {
    const int t = dir ? a.T - 1 - step : step;
    const float* hc = hs + (step & 1) * 16 * LDW;
    float* hn_ = hs + ((step + 1) & 1) * 16 * LDW;
    float pnext[4][4];
    if (step + 1 < a.T) load_pre(dir ? t - 1 : t + 1, pnext);          // lands under the MFMAs below
    if (active) {
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kg = 0; kg < HP / 16; ++kg) {
        const float4 av = *reinterpret_cast<const float4*>(hc + j * LDW + 16 * kg + 4 * q);
        const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(Wl + (g * HP + unit) * LDW + 16 * kg + 4 * q);
          const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], b4[i], acc[g], 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                       // gate order of torch.nn.LSTM: i, f, g, o
        const float gi = sigmoidf_(acc[0][r] + pcur[0][r] + bh[0]), gf = sigmoidf_(acc[1][r] + pcur[1][r] + bh[1]);
        const float gg = tanhf_(acc[2][r] + pcur[2][r] + bh[2]), go = sigmoidf_(acc[3][r] + pcur[3][r] + bh[3]);
        c[r] = gf * c[r] + gi * gg;
        const float h = uok ? go * tanhf_(c[r]) : 0.f;
        hlast[r] = h;
        hn_[(4 * q + r) * LDW + unit] = h;
        const int64_t row = r0 + 4 * q + r;
        if (uok && row < a.rows) {
          a.out[((size_t)t * a.rows + row) * 2 * H + dir * H + unit] = h;
          if (a.saved) {
            float* sv = a.saved + (((size_t)t * a.rows + row) * 2 + dir) * 5 * H + unit;
            sv[0] = gi; sv[H] = gf; sv[2 * H] = gg; sv[3 * H] = go; sv[4 * H] = c[r];
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) pcur[g][r] = pnext[g][r];
    __syncthreads();
  }
  if (active && uok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = r0 + 4 * q + r;
      if (row >= a.rows) continue;
      float hv = hlast[r];
      if (a.T == 0) hv = a.h0 ? a.h0[((size_t)dir * a.rows + row) * H + unit] : 0.f;
      if (a.hn) a.hn[((size_t)dir * a.rows + row) * H + unit] = hv;
      if (a.cn) a.cn[((size_t)dir * a.rows + row) * H + unit] = c[r];
    }
  }
}


// ---------------------------------------------------------------------------------------------- back-propagation through time
struct SeqBwdArgs {
  const float* saved;                                      // (T, rows, 2, 5, H) from the training forward
  const float* whh[2];
  const float* c0;                                         // (2, rows, H) or null
  const float* gout; const float* ghn; const float* gcn;   // (T, rows, 2 H) or null; (2, rows, H) or null
  float* dpre[2];                                          // (T * rows, 4 H) per direction: d loss / d pre-activations [i | f | g | o]
  float* gh0; float* gc0;                                  // (2, rows, H) or null
  int T; int64_t rows; int H;
};

template <int HP>
__global__ __launch_bounds__(TS) void lstm_seq_bwd_kernel(SeqBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDT = 4 * HP + 4;                          // (4 x odd: conflict-free 16-byte reads along a row)
  float* WT = smem;                                        // [HP][LDT]: WT[k][g * HP + n] = W_hh[g * H + n][k]
  float* das = WT + HP * LDT;                              // [2][16][LDT]: a step's pre-activation gradients, rows x (gate, unit)
  const int dir = blockIdx.y, H = a.H;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int j = lane & 15, q = lane >> 4;
  const float* whh = a.whh[dir];
  for (int i = threadIdx.x; i < HP * 4 * HP; i += TS) {
    const int k = i / (4 * HP), rem = i - k * 4 * HP, g = rem / HP, n = rem - g * HP;
    WT[k * LDT + rem] = (n < H && k < H) ? whh[(size_t)(g * H + n) * H + k] : 0.f;
  }
  for (int i = threadIdx.x; i < 2 * 16 * LDT; i += TS) das[i] = 0.f;
  const int unit = 16 * wave + j;
  const bool active = 16 * wave < HP, uok = unit < H;
  float dh[4], dc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = r0 + 4 * q + r;
    const bool ok = uok && row < a.rows;
    dh[r] = (ok && a.ghn) ? a.ghn[((size_t)dir * a.rows + row) * H + unit] : 0.f;
    dc[r] = (ok && a.gcn) ? a.gcn[((size_t)dir * a.rows + row) * H + unit] : 0.f;
  }
  __syncthreads();
  // what a step reads: its saved activations and cell state, the cell state of the step BEFORE it in time order, dL/dout
  struct StepIn { float g[4][4]; float c[4], cp[4], go[4]; };
  auto load_step = [&](int t, StepIn& s) __attribute__((always_inline)) {
    const int tp = dir ? t + 1 : t - 1;                    // the step whose cell state entered step t (the reverse direction runs T-1 .. 0)
    const bool first = dir ? t == a.T - 1 : t == 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = r0 + 4 * q + r;
      const bool ok = uok && row < a.rows;
      const float* sv = a.saved + (((size_t)t * a.rows + row) * 2 + dir) * 5 * H + unit;
#pragma unroll
      for (int g = 0; g < 4; ++g) s.g[g][r] = ok ? sv[g * H] : 0.f;
      s.c[r] = ok ? sv[4 * H] : 0.f;
      if (first) s.cp[r] = (ok && a.c0) ? a.c0[((size_t)dir * a.rows + row) * H + unit] : 0.f;
      else s.cp[r] = ok ? a.saved[(((size_t)tp * a.rows + row) * 2 + dir) * 5 * H + 4 * H + unit] : 0.f;
      s.go[r] = (ok && a.gout) ? a.gout[((size_t)t * a.rows + row) * 2 * H + dir * H + unit] : 0.f;
    }
  };
  StepIn cur;
  if (a.T > 0) load_step(dir ? 0 : a.T - 1, cur);
  for (int step = 0; step < a.T; ++step) {
    const int t = dir ? step : a.T - 1 - step;             // reverse of the forward's processing order
    float* dt = das + (step & 1) * 16 * LDT;
    StepIn nxt;
    if (step + 1 < a.T) load_step(dir ? t + 1 : t - 1, nxt);           // lands under this step's arithmetic and MFMAs
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gi = cur.g[0][r], gf = cur.g[1][r], gg = cur.g[2][r], go = cur.g[3][r];
        const float tc = tanhf_(cur.c[r]);
        const float dhr = dh[r] + cur.go[r];
        const float dct = dc[r] + dhr * go * (1.f - tc * tc);
        const float da_i = dct * gg * gi * (1.f - gi), da_f = dct * cur.cp[r] * gf * (1.f - gf);
        const float da_g = dct * gi * (1.f - gg * gg), da_o = dhr * tc * go * (1.f - go);
        dc[r] = dct * gf;
        float* drow = dt + (4 * q + r) * LDT + unit;       // (padding units: every factor above is zero there)
        drow[0] = da_i; drow[HP] = da_f; drow[2 * HP] = da_g; drow[3 * HP] = da_o;
        const int64_t row = r0 + 4 * q + r;
        if (uok && row < a.rows) {
          float* dp = a.dpre[dir] + ((size_t)t * a.rows + row) * 4 * H + unit;
          dp[0] = da_i; dp[H] = da_f; dp[2 * H] = da_g; dp[3 * H] = da_o;
        }
      }
    }
    __syncthreads();                                        // the tile is whole: every wave reads all of it (the other buffer is next step's)
    if (active) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int ng = 0; ng < 4 * HP / 16; ++ng) {
        const float4 av = *reinterpret_cast<const float4*>(dt + j * LDT + 16 * ng + 4 * q);
        const float4 bv = *reinterpret_cast<const float4*>(WT + unit * LDT + 16 * ng + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = acc[r];           // dL/dh of the step before, rows 4q + r, unit 16 wave + j: this lane's own
    }
    cur = nxt;
  }
  if (active && uok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = r0 + 4 * q + r;
      if (row >= a.rows) continue;
      if (a.gh0) a.gh0[((size_t)dir * a.rows + row) * H + unit] = dh[r];
      if (a.gc0) a.gc0[((size_t)dir * a.rows + row) * H + unit] = dc[r];
    }
  }
}

// h_prev of every step, contiguous (T * rows, H), for one direction: out[t -+ 1][:, dir * H ...] or h0 (zeros without it)
__global__ __launch_bounds__(256) void lstm_seq_hprev_kernel(const float* __restrict__ out, const float* __restrict__ h0, float* __restrict__ hp, int T,
                                                             int64_t rows, int H, int dir) {
  const int64_t n = (int64_t)T * rows * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % H);
    const int64_t tr = i / H, row = tr % rows;
    const int t = (int)(tr / rows);
    const int tp = dir ? t + 1 : t - 1;
    float v;
    if (tp < 0 || tp >= T) v = h0 ? h0[((size_t)dir * rows + row) * H + k] : 0.f;
    else v = out[((size_t)tp * rows + row) * 2 * H + dir * H + k];
    hp[i] = v;
  }
}
__global__ __launch_bounds__(256) void lstm_seq_add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] += src[i];
}

}  // namespace

extern "C" {

size_t hypad_lstm_seq_workspace_bytes(int seq_len, int64_t rows, int hidden) {
  if (seq_len <= 0 || rows <= 0 || hidden <= 0) return 0;
  return (size_t)2 * seq_len * rows * 4 * hidden * sizeof(float);
}

int hypad_lstm_bidir_seq_fwd(const float* x, const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                             const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, const float* h0,
                             const float* c0, float* out, float* hn, float* cn, int seq_len, int64_t rows, int in_dim, int hidden,
                             void* workspace, size_t workspace_bytes, hypad_stream_t s) {
  return hypad_lstm_bidir_seq_fwd_train(x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, h0, c0, out, hn, cn, nullptr, seq_len, rows,
                                        in_dim, hidden, workspace, workspace_bytes, s);
}

int hypad_lstm_bidir_seq_fwd_train(const float* x, const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                                   const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, const float* h0,
                                   const float* c0, float* out, float* hn, float* cn, float* saved, int seq_len, int64_t rows, int in_dim,
                                   int hidden, void* workspace, size_t workspace_bytes, hypad_stream_t s) {
  if (!x || !w_ih_f || !w_hh_f || !b_ih_f || !b_hh_f || !w_ih_r || !w_hh_r || !b_ih_r || !b_hh_r || !out || seq_len <= 0 || rows <= 0 ||
      in_dim <= 0 || hidden <= 0)
    return HYPAD_EINVAL;
  if (hidden > 64) return HYPAD_EUNSUPPORTED;               // four waves x 16 units; W_hh in LDS
  if (!workspace || workspace_bytes < hypad_lstm_seq_workspace_bytes(seq_len, rows, hidden)) return HYPAD_EWORKSPACE;
  if ((int64_t)seq_len * rows > 0x7fffffff) return HYPAD_EINVAL;
  float* pre_f = (float*)workspace;
  float* pre_r = pre_f + (size_t)seq_len * rows * 4 * hidden;
  int rc = hypad_linear_act_fwd(x, w_ih_f, b_ih_f, pre_f, (int64_t)seq_len * rows, in_dim, 4 * hidden, HYPAD_ACT_NONE, s);
  if (rc) return rc;
  rc = hypad_linear_act_fwd(x, w_ih_r, b_ih_r, pre_r, (int64_t)seq_len * rows, in_dim, 4 * hidden, HYPAD_ACT_NONE, s);
  if (rc) return rc;
  SeqArgs a;
  a.pre[0] = pre_f; a.pre[1] = pre_r; a.whh[0] = w_hh_f; a.whh[1] = w_hh_r; a.bhh[0] = b_hh_f; a.bhh[1] = b_hh_r;
  a.h0 = h0; a.c0 = c0; a.out = out; a.hn = hn; a.cn = cn; a.saved = saved; a.T = seq_len; a.rows = rows; a.H = hidden;
  const int hp = (hidden + 15) & ~15;
  const size_t lds = (size_t)(4 * hp * (hp + 4) + 2 * 16 * (hp + 4)) * sizeof(float);
  const dim3 grid((unsigned)((rows + 15) / 16), 2), block(TS);
#define HYPAD_SEQ(HP)                                                                                                          \
  do {                                                                                                                        \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)lstm_seq_kernel<HP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return HYPAD_EUNSUPPORTED;                                                                                               \
    hipLaunchKernelGGL(lstm_seq_kernel<HP>, grid, block, lds, (hipStream_t)s, a);                                              \
  } while (0)
  if (hp == 16) HYPAD_SEQ(16); else if (hp == 32) HYPAD_SEQ(32); else if (hp == 48) HYPAD_SEQ(48); else HYPAD_SEQ(64);
#undef HYPAD_SEQ
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}


// workspace of the backward, in floats: [dpre_f | dpre_r] (2 x T rows 4H), the linear backward's scratch copy (T rows 4H), h_prev of one
// direction at a time (T rows H), the second direction's input gradient (T rows in), one 4H bias row nobody asked for
size_t hypad_lstm_seq_bwd_workspace_bytes(int seq_len, int64_t rows, int in_dim, int hidden) {
  if (seq_len <= 0 || rows <= 0 || in_dim <= 0 || hidden <= 0) return 0;
  const size_t tr = (size_t)seq_len * rows;
  return (tr * (size_t)(12 * hidden + hidden + in_dim) + 4 * (size_t)hidden + 64) * sizeof(float);
}

int hypad_lstm_bidir_seq_bwd(const float* x, const float* w_ih_f, const float* w_hh_f, const float* w_ih_r, const float* w_hh_r,
                             const float* h0, const float* c0, const float* out, const float* saved, const float* grad_out,
                             const float* grad_hn, const float* grad_cn, float* grad_x, float* grad_w_ih_f, float* grad_w_hh_f,
                             float* grad_b_f, float* grad_w_ih_r, float* grad_w_hh_r, float* grad_b_r, float* grad_h0, float* grad_c0,
                             int seq_len, int64_t rows, int in_dim, int hidden, void* workspace, size_t workspace_bytes, hypad_stream_t s) {
  if (!x || !w_ih_f || !w_hh_f || !w_ih_r || !w_hh_r || !out || !saved || seq_len <= 0 || rows <= 0 || in_dim <= 0 || hidden <= 0) return HYPAD_EINVAL;
  if (!grad_out && !grad_hn && !grad_cn) return HYPAD_EINVAL;
  if (!grad_x || !grad_w_ih_f || !grad_w_hh_f || !grad_b_f || !grad_w_ih_r || !grad_w_hh_r || !grad_b_r) return HYPAD_EINVAL;
  if (hidden > 64) return HYPAD_EUNSUPPORTED;
  if (!workspace || workspace_bytes < hypad_lstm_seq_bwd_workspace_bytes(seq_len, rows, in_dim, hidden)) return HYPAD_EWORKSPACE;
  if ((int64_t)seq_len * rows > 0x7fffffff) return HYPAD_EINVAL;
  const size_t tr = (size_t)seq_len * rows;
  const int H = hidden;
  float* dpre_f = (float*)workspace;
  float* dpre_r = dpre_f + tr * 4 * H;
  float* scratch = dpre_r + tr * 4 * H;
  float* hprev = scratch + tr * 4 * H;
  float* gx_r = hprev + tr * H;
  float* gb_tmp = gx_r + tr * in_dim;
  SeqBwdArgs a;
  a.saved = saved; a.whh[0] = w_hh_f; a.whh[1] = w_hh_r; a.c0 = c0; a.gout = grad_out; a.ghn = grad_hn; a.gcn = grad_cn;
  a.dpre[0] = dpre_f; a.dpre[1] = dpre_r; a.gh0 = grad_h0; a.gc0 = grad_c0; a.T = seq_len; a.rows = rows; a.H = H;
  const int hp = (H + 15) & ~15;
  const size_t lds = (size_t)(hp * (4 * hp + 4) + 2 * 16 * (4 * hp + 4)) * sizeof(float);
  const dim3 grid((unsigned)((rows + 15) / 16), 2), block(TS);
#define HYPAD_SEQ_BWD(HP)                                                                                                      \
  do {                                                                                                                        \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel<HP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return HYPAD_EUNSUPPORTED;                                                                                               \
    hipLaunchKernelGGL(lstm_seq_bwd_kernel<HP>, grid, block, lds, (hipStream_t)s, a);                                          \
  } while (0)
  if (hp == 16) HYPAD_SEQ_BWD(16); else if (hp == 32) HYPAD_SEQ_BWD(32); else if (hp == 48) HYPAD_SEQ_BWD(48); else HYPAD_SEQ_BWD(64);
#undef HYPAD_SEQ_BWD
  HYPAD_CHECK_LAUNCH();
  // the dense contractions over all T * rows rows: dX, dW_ih, db from the input projection; dW_hh from the recurrent one
  int rc = hypad_linear_act_bwd(x, w_ih_f, nullptr, dpre_f, grad_x, grad_w_ih_f, grad_b_f, scratch, (int64_t)tr, in_dim, 4 * H, HYPAD_ACT_NONE, s);
  if (rc) return rc;
  rc = hypad_linear_act_bwd(x, w_ih_r, nullptr, dpre_r, gx_r, grad_w_ih_r, grad_b_r, scratch, (int64_t)tr, in_dim, 4 * H, HYPAD_ACT_NONE, s);
  if (rc) return rc;
  const int64_t nx = (int64_t)tr * in_dim;
  hipLaunchKernelGGL(lstm_seq_add_kernel, dim3((unsigned)((nx + 255) / 256 > 4096 ? 4096 : (nx + 255) / 256)), dim3(256), 0, (hipStream_t)s, grad_x, gx_r, nx);
  HYPAD_CHECK_LAUNCH();
  for (int dir = 0; dir < 2; ++dir) {
    const int64_t nh = (int64_t)tr * H;
    hipLaunchKernelGGL(lstm_seq_hprev_kernel, dim3((unsigned)((nh + 255) / 256 > 4096 ? 4096 : (nh + 255) / 256)), dim3(256), 0, (hipStream_t)s, out, h0, hprev,
                       seq_len, rows, H, dir);
    HYPAD_CHECK_LAUNCH();
    rc = hypad_linear_act_bwd(hprev, dir ? w_hh_r : w_hh_f, nullptr, dir ? dpre_r : dpre_f, nullptr, dir ? grad_w_hh_r : grad_w_hh_f, gb_tmp, scratch,
                              (int64_t)tr, H, 4 * H, HYPAD_ACT_NONE, s);
    if (rc) return rc;
  }
  return HYPAD_OK;
}

}  // extern "C"
