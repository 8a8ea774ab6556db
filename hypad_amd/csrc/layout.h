// Parameter-arena layouts of the four TadGAN networks (host + device).
// Tensor order, names and shapes follow the reference's state_dict (models/tadgan.py:11-21,31-56,71-89,
// 110-121); every tensor starts at a multiple of 4 floats (16 B).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace hypad {

constexpr int ENC_H = 50;      // Encoder LSTM hidden size      models/tadgan.py:17
constexpr int DEC_H = 64;      // Decoder LSTM hidden size      models/tadgan.py:36
constexpr int DEC_D1 = 50;     // Decoder dense1 out_features   models/tadgan.py:34
constexpr int ENC_LDG = 6 * ENC_H + 8;   // LDS row stride of the encoder's gate tile [16][3 x 2 x 50]: 308 = 4 x 77 (lds_stride)
constexpr int ENC_LDH = 2 * ENC_H + 8;   // ... of its hidden tile [16][2 x 50]: 108 = 4 x 27
constexpr int MAX_S = 256;     // fused kernels: signal_shape limit (LDS budget)
constexpr int MAX_L = 32;      // latent / critic width limit

HD int pad4(int n) { return (n + 3) & ~3; }
// LDS row stride (floats) of a 16-row tile: a multiple of four floats that is 4 x ODD.  Both ways a tile is touched are then free of
// bank conflicts (MI355X_MICROARCH.md, LDS: banks = dword address mod 64 for ds_read_b128 within a 16-lane group, mod 32 for 4-byte
// accesses within a 32-lane half): the MFMA A-operand reads (lane (row j, group q) takes 16 bytes of row j: sixteen rows at stride
// 4 (2 m + 1) dwords fall on sixteen different 16-byte slots of the 256-byte bank row) and the epilogues' accumulator writes (lane
// (j, q) writes rows 4 q + r: lane groups q and q + 1 are 16 (2 m + 1) dwords = 16 banks apart).  pad4(100) + 4 = 104 = 4 x 26 -- the
// reference window of all things -- cost two LDS cycles for every such access: a third of the generator kernel's LDS cycles by counter.
HD int lds_stride(int cols) { const int n = pad4(cols) + 4; return ((n >> 2) & 1) ? n : n + 4; }

struct LstmDir { int w_ih, w_hh, b_ih, b_hh; };

struct EncLayout {
  LstmDir dir[2];
  int dense_w, dense_b, total;
};
HD EncLayout enc_layout(int S, int L) {
  EncLayout e; int o = 0;
  for (int d = 0; d < 2; ++d) {
    e.dir[d].w_ih = o; o += pad4(4 * ENC_H * S);
    e.dir[d].w_hh = o; o += pad4(4 * ENC_H * ENC_H);
    e.dir[d].b_ih = o; o += pad4(4 * ENC_H);
    e.dir[d].b_hh = o; o += pad4(4 * ENC_H);
  }
  e.dense_w = o; o += pad4(L * 2 * ENC_H);
  e.dense_b = o; o += pad4(L);
  e.total = o;
  return e;
}

struct DecLayout {
  int d1_w, d1_b;
  LstmDir l[2][2];     // [layer][direction]
  int d2_w, d2_b, head_w, head_b, total;
};
HD DecLayout dec_layout(int S, int L, int hyperbolic) {
  DecLayout d; int o = 0;
  d.d1_w = o; o += pad4(DEC_D1 * L);
  d.d1_b = o; o += pad4(DEC_D1);
  for (int layer = 0; layer < 2; ++layer) {
    int in = layer == 0 ? DEC_D1 : 2 * DEC_H;
    for (int dir = 0; dir < 2; ++dir) {
      d.l[layer][dir].w_ih = o; o += pad4(4 * DEC_H * in);
      d.l[layer][dir].w_hh = o; o += pad4(4 * DEC_H * DEC_H);
      d.l[layer][dir].b_ih = o; o += pad4(4 * DEC_H);
      d.l[layer][dir].b_hh = o; o += pad4(4 * DEC_H);
    }
  }
  d.d2_w = o; o += pad4(S * 2 * DEC_H);
  d.d2_b = o; o += pad4(S);
  d.head_w = d.head_b = -1;
  if (hyperbolic) {
    d.head_w = o; o += pad4(S * S);
    d.head_b = o; o += pad4(S);
  }
  d.total = o;
  return d;
}

// ---- MFMA-native packed copies of the generator's weights (tile_gemm.h gemm_nt_packed): a matrix of N output rows and K
// reduction columns is stored as blocks [ceil(N/16)][ceil(K/16)][64 lanes][4], lane (j, q) of block (tn, g) holding
// M[16 tn + j][16 g + 4 q .. + 3], zero padded.  Forward products use M = W with the LSTM gate rows compacted to [i|g|o];
// backward-data products use M = W^T (for an LSTM layer both directions stacked along the reduction), so that both are
// "NT" products against fully coalesced 1 KB blocks.  The copies live in the training workspace and are rebuilt / updated
// by the library (pack kernel, dW + Adam kernel); the parameter arena stays the PyTorch layout.
HD int packed_floats(int N, int K) { return ((N + 15) >> 4) * ((K + 15) >> 4) * 256; }
// In the packed gate matrices every gate owns a whole number of 16-row blocks: gate x of hidden unit u is packed row
// x * up16(H) + u (rows H .. up16(H)-1 of a gate are zero).  A wave can then own the i, g and o tiles of the same 16
// hidden units and apply the LSTM cell straight from its accumulators (nets.h lstm_layer_fwd_packed).
HD int gate_rows(int H) { return 3 * ((H + 15) & ~15); }
struct GenPack {
  // forward: W blocks, then summed biases (padded to 16)
  int enc_g[2], enc_gb[2];       // encoder gates, direction d: (3 ENC_H, S)
  int enc_d, enc_db;             // encoder dense (L, 2 ENC_H)
  int d1, d1b;                   // decoder dense1 (DEC_D1, L)
  int l_g[2][2], l_gb[2][2];     // decoder layer l, direction d gates (3 DEC_H, in_l)
  int d2, d2b;                   // decoder dense2 (S, 2 DEC_H)
  int head;                      // hyperbolic_linear.weight (S, S), hyperbolic only (else -1)
  // backward data: W^T blocks
  int enc_d_t;                   // (2 ENC_H out, L red)
  int head_t;                    // (S, S)
  int d2_t;                      // (2 DEC_H out, S red)
  int l_t[2];                    // layer l: (in_l out, 6 DEC_H red)
  int d1_t;                      // (L out, DEC_D1 red)
  int total;
};
HD GenPack gen_pack(int S, int L, int hyperbolic) {
  GenPack g; int o = 0;
  auto vec = [](int n) { return (n + 15) & ~15; };
  for (int d = 0; d < 2; ++d) { g.enc_g[d] = o; o += packed_floats(gate_rows(ENC_H), S); g.enc_gb[d] = o; o += gate_rows(ENC_H); }
  g.enc_d = o; o += packed_floats(L, 2 * ENC_H); g.enc_db = o; o += vec(L);
  g.d1 = o; o += packed_floats(DEC_D1, L); g.d1b = o; o += vec(DEC_D1);
  for (int l = 0; l < 2; ++l) {
    const int in = l == 0 ? DEC_D1 : 2 * DEC_H;
    for (int d = 0; d < 2; ++d) { g.l_g[l][d] = o; o += packed_floats(gate_rows(DEC_H), in); g.l_gb[l][d] = o; o += gate_rows(DEC_H); }
  }
  g.d2 = o; o += packed_floats(S, 2 * DEC_H); g.d2b = o; o += vec(S);
  g.head = -1;
  if (hyperbolic) { g.head = o; o += packed_floats(S, S); }
  g.enc_d_t = o; o += packed_floats(2 * ENC_H, L);
  g.head_t = -1;
  if (hyperbolic) { g.head_t = o; o += packed_floats(S, S); }
  g.d2_t = o; o += packed_floats(2 * DEC_H, S);
  for (int l = 0; l < 2; ++l) { g.l_t[l] = o; o += packed_floats(l == 0 ? DEC_D1 : 2 * DEC_H, 6 * DEC_H); }
  g.d1_t = o; o += packed_floats(L, DEC_D1);
  g.total = o;
  return g;
}

// Critics: nh hidden Linear(.,L)+LeakyReLU+Dropout blocks, then Linear(L,1).
struct CriticLayout {
  int nh;              // 4 (CriticX) or 2 (CriticZ)
  int in_dim;          // S or L
  int w[5], b[5];      // layer li: weight (L, in_dim | L) ... last (1, L)
  float p_drop;
  int total;
  int L;               // hidden width
  // w[li] / b[li] in closed form, for device code that indexes with a run-time layer number: indexing the arrays there puts the
  // whole struct into scratch memory (60 bytes of private segment per lane in every kernel that did)
  HD int wof(int li) const { return li == 0 ? 0 : pad4(L * in_dim) + pad4(L) + (li - 1) * (pad4(L * L) + pad4(L)); }
  HD int bof(int li) const { return li == 0 ? pad4(L * in_dim) : wof(li) + (li == nh ? pad4(L) : pad4(L * L)); }
};
HD CriticLayout critic_layout(int in_dim, int L, int nh, float p_drop) {
  CriticLayout c; c.nh = nh; c.in_dim = in_dim; c.p_drop = p_drop; c.L = L; int o = 0;
  for (int i = 0; i <= nh; ++i) {
    int k = i == 0 ? in_dim : L;
    int n = i == nh ? 1 : L;
    c.w[i] = o; o += pad4(n * k);
    c.b[i] = o; o += pad4(n);
  }
  for (int i = nh + 1; i < 5; ++i) c.w[i] = c.b[i] = -1;
  c.total = o;
  return c;
}
HD CriticLayout cx_layout(int S, int L) { return critic_layout(S, L, 4, 0.25f); }   // models/tadgan.py:75
HD CriticLayout cz_layout(int L) { return critic_layout(L, L, 2, 0.2f); }          // models/tadgan.py:120

}  // namespace hypad
