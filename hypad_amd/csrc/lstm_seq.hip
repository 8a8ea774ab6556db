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
// Forward only: the reference never back-propagates through time (T = 1).
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
        if (uok && row < a.rows) a.out[((size_t)t * a.rows + row) * 2 * H + dir * H + unit] = h;
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
  a.h0 = h0; a.c0 = c0; a.out = out; a.hn = hn; a.cn = cn; a.T = seq_len; a.rows = rows; a.H = hidden;
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

}  // extern "C"
