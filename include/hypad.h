/*
 * hypad.h -- C ABI of libhypad_hip.so: the HypAD / TadGAN train + score hot path on MI355X (gfx950).
 *
 * The reference (aleflabo/HypAD, pure Python) has no FFI: its hot path sits behind Python callables.
 * Each entry point below names the reference code it replaces (file:line under /root/reference), so a
 * maintainer can bind it (ctypes stub in INTEGRATION.md) where that code runs today.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every data pointer is a DEVICE pointer owned by the caller.
 *   - `stream` is a hipStream_t passed as void*.  All work is enqueued on it; nothing synchronises,
 *     allocates or frees, so every call may be captured into a hipGraph.
 *   - return 0 on success, a negative HYPAD_E* code for argument errors, a positive hipError_t otherwise.
 *   - fp32 arithmetic throughout unless a function says fp64 (scoring post-processing follows NumPy's fp64).
 *   - matrices are dense row-major; weights keep PyTorch's (out_features, in_features) layout.
 *   - curvature is fixed at k = -1 (the only value the reference uses: hyperspace/hyrnn_nets.py:20,166).
 */
#ifndef HYPAD_H_
#define HYPAD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations below are its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define HYPAD_ABI_VERSION 7

enum {
  HYPAD_OK = 0,
  HYPAD_EINVAL = -1,       /* bad argument (null pointer, non-positive size, unsupported shape) */
  HYPAD_EWORKSPACE = -2,   /* workspace missing or too small */
  HYPAD_EUNSUPPORTED = -3  /* shape outside the fused path's limits (see hypad_limits) */
};

typedef void* hypad_stream_t; /* hipStream_t */

int hypad_abi_version(void);
const char* hypad_error_string(int code);
/* signal_shape <= max_signal_shape, latent_dim <= max_latent for the fused network kernels */
void hypad_limits(int* max_signal_shape, int* max_latent);

/* ------------------------------------------------------------------------------------------------
 * Parameter arenas.  One flat fp32 buffer per network; tensors sit at fixed, 16-byte aligned offsets
 * and carry the reference's state_dict names and shapes (models/tadgan.py:11-21,31-56,71-89,110-121;
 * SURVEY.md A.1).  The host side builds its torch views from these queries.
 * ---------------------------------------------------------------------------------------------- */
enum { HYPAD_NET_ENCODER = 0, HYPAD_NET_DECODER = 1, HYPAD_NET_CRITIC_X = 2, HYPAD_NET_CRITIC_Z = 3 };

int hypad_param_count(int net, int signal_shape, int latent_dim, int hyperbolic);   /* floats, padded */
int hypad_param_tensors(int net, int hyperbolic);
int hypad_param_info(int net, int signal_shape, int latent_dim, int hyperbolic, int index,
                     char* name, int name_cap, int* offset, int* rows, int* cols);   /* cols = 0 for 1-D */

/* ------------------------------------------------------------------------------------------------
 * Poincare-ball ops (geoopt.manifolds.stereographic.math, vendored as /root/reference/math_.py).
 * Row-wise over (rows, dim) matrices.
 * ---------------------------------------------------------------------------------------------- */
/* gmath.expmap0  math_.py:1132-1136 (+ tan_k :217-238, tanh :51-53) */
int hypad_expmap0_fwd(const float* u, float* out, int64_t rows, int dim, hypad_stream_t stream);
int hypad_expmap0_bwd(const float* u, const float* grad_out, float* grad_u, int64_t rows, int dim, hypad_stream_t stream);
/* gmath.logmap0  math_.py:1267-1270 (+ artan_k :241-262, artanh :56-59) */
int hypad_logmap0_fwd(const float* y, float* out, int64_t rows, int dim, hypad_stream_t stream);
int hypad_logmap0_bwd(const float* y, const float* grad_out, float* grad_y, int64_t rows, int dim, hypad_stream_t stream);
/* gmath.mobius_add  math_.py:536-555.  y_rows == 1 broadcasts y over the rows of x (hyrnn_nets.py:31);
 * then grad_y is the (rows, dim) matrix of per-row contributions, to be column-summed by the caller
 * (hypad_column_sum). */
int hypad_mobius_add_fwd(const float* x, const float* y, float* out, int64_t rows, int dim, int64_t y_rows, hypad_stream_t stream);
int hypad_mobius_add_bwd(const float* x, const float* y, const float* grad_out, float* grad_x, float* grad_y,
                         int64_t rows, int dim, int64_t y_rows, hypad_stream_t stream);
/* gmath.project  math_.py:340-352 (fp32 eps 4e-3) */
int hypad_project_fwd(const float* x, float* out, int64_t rows, int dim, hypad_stream_t stream);
int hypad_project_bwd(const float* x, const float* grad_out, float* grad_x, int64_t rows, int dim, hypad_stream_t stream);
/* fused expmap0 -> mobius_add(bias) -> project: the tail of mobius_linear, hyperspace/hyrnn_nets.py:27-34 */
int hypad_mobius_head_fwd(const float* u, const float* bias, float* out, int64_t rows, int dim, hypad_stream_t stream);
int hypad_mobius_head_bwd(const float* u, const float* bias, const float* grad_out, float* grad_u,
                          float* grad_bias_rows, int64_t rows, int dim, hypad_stream_t stream);
/* MobiusLinear.forward / mobius_linear  hyperspace/hyrnn_nets.py:186-200, :13-35
 * (hyperbolic_input=False, hyperbolic_bias=True, nonlin=None).  weight (out_dim, in_dim), bias (out_dim).
 * workspace: hypad_mobius_linear_workspace_bytes(rows, out_dim) -- holds u = x W^T for the backward. */
size_t hypad_mobius_linear_workspace_bytes(int64_t rows, int out_dim);
int hypad_mobius_linear_fwd(const float* x, const float* weight, const float* bias, float* out, float* u_save,
                            int64_t rows, int in_dim, int out_dim, hypad_stream_t stream);
int hypad_mobius_linear_bwd(const float* x, const float* weight, const float* bias, const float* u_saved,
                            const float* grad_out, float* grad_x, float* grad_weight, float* grad_bias,
                            void* workspace, size_t workspace_bytes,
                            int64_t rows, int in_dim, int out_dim, hypad_stream_t stream);
/* inline row-wise Poincare distance  train.py:226-230; utils/anomaly_detection_utils.py:58-66,167-175 */
int hypad_poincare_rowdist_fwd(const float* u, const float* v, float* dist, int64_t rows, int dim, hypad_stream_t stream);
int hypad_poincare_rowdist_bwd(const float* u, const float* v, const float* grad_dist, float* grad_u, float* grad_v,
                               int64_t rows, int dim, hypad_stream_t stream);
/* hyperbolic reconstruction loss  train.py:226-232: loss[0] = sum_r dist(u_r, v_r) / batch.
 * bwd: gradients of (grad_loss * loss) w.r.t. u and v. */
int hypad_hyper_loss_fwd(const float* u, const float* v, float* loss, int64_t rows, int dim, int batch, hypad_stream_t stream);
int hypad_hyper_loss_bwd(const float* u, const float* v, float grad_loss, float* grad_u, float* grad_v,
                         int64_t rows, int dim, int batch, hypad_stream_t stream);
/* pair-wise distance  hyperspace/poincare_distance.py:5-16 (+ :19-25, :28-48): out (n, m) */
int hypad_poincare_pairdist_fwd(const float* pred, const float* gt, float* out, int n, int m, int dim, hypad_stream_t stream);
/* out[c] = sum_r in[r][c] */
int hypad_column_sum(const float* in, float* out, int64_t rows, int dim, hypad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense building blocks (torch ATen calls of the reference: F.linear, nn.LSTM at T=1).
 * ---------------------------------------------------------------------------------------------- */
enum { HYPAD_ACT_NONE = 0, HYPAD_ACT_TANH = 1, HYPAD_ACT_LEAKY02 = 2 };
/* y = act(x W^T + b): nn.Linear + nn.Tanh / nn.LeakyReLU(0.2)  models/tadgan.py:21,34,39,40,77-89 */
int hypad_linear_act_fwd(const float* x, const float* weight, const float* bias, float* y,
                         int64_t rows, int in_dim, int out_dim, int act, hypad_stream_t stream);
/* grad_x = (grad_y * act'(y)) W ; grad_w = (grad_y * act')^T x ; grad_b = colsum(grad_y * act').  y = forward output. */
int hypad_linear_act_bwd(const float* x, const float* weight, const float* y, const float* grad_y,
                         float* grad_x, float* grad_weight, float* grad_bias, float* grad_pre_scratch,
                         int64_t rows, int in_dim, int out_dim, int act, hypad_stream_t stream);
/* One bidirectional LSTM layer at seq_len 1 with h0 = c0 = 0 (models/tadgan.py:15-20,35-37 as driven by
 * :24-25,59-60): out (rows, 2*hidden) = [h_fwd | h_rev].  w_ih_* (4*hidden, in_dim) with
 * PyTorch gate order [i,f,g,o]; W_hh cannot influence the result and is not read.
 * gates_save (rows, 2, 4, hidden) receives (i, g, o, tanh(c)) for the backward (may be NULL). */
int hypad_lstm_bidir_fwd(const float* x, const float* w_ih_f, const float* b_ih_f, const float* b_hh_f,
                         const float* w_ih_r, const float* b_ih_r, const float* b_hh_r,
                         float* out, float* gates_save, int64_t rows, int in_dim, int hidden, hypad_stream_t stream);
/* grad_gates (rows, 2, 4*hidden) pre-activation gradients in PyTorch gate order (f block zero);
 * grad_x (rows, in_dim); parameter gradients follow as grad_gates^T x (hypad_linear weight rule). */
int hypad_lstm_bidir_bwd(const float* w_ih_f, const float* w_ih_r, const float* gates_saved, const float* grad_out,
                         float* grad_gates, float* grad_x, int64_t rows, int in_dim, int hidden, hypad_stream_t stream);

/* The same layer over seq_len time steps -- torch.nn.LSTM(in_dim, hidden, num_layers=1, bidirectional=True) on a (seq_len, rows,
 * in_dim) input, with W_hh and optional initial states (the module models/tadgan.py:15-20, :35-38 builds; the reference itself
 * always feeds it seq_len = 1, where hypad_lstm_bidir_fwd is the faster form).  x (seq_len, rows, in_dim); w_ih_* (4*hidden,
 * in_dim), w_hh_* (4*hidden, hidden), biases (4*hidden), PyTorch gate order [i,f,g,o]; h0 / c0 (2, rows, hidden) or NULL = zeros;
 * out (seq_len, rows, 2*hidden) = [h_fwd(t) | h_rev(t)]; hn / cn (2, rows, hidden) final states, may be NULL.  The input
 * projections of all steps run as one MFMA GEMM; the recurrence is a persistent kernel per (16-row tile, direction): W_hh in
 * LDS, h_t in LDS, c_t in registers, the four gates of a unit on one lane, one barrier per step.  hidden <= 64.
 * workspace: hypad_lstm_seq_workspace_bytes(seq_len, rows, hidden). */
size_t hypad_lstm_seq_workspace_bytes(int seq_len, int64_t rows, int hidden);
int hypad_lstm_bidir_seq_fwd(const float* x, const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                             const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                             const float* h0, const float* c0, float* out, float* hn, float* cn, int seq_len, int64_t rows,
                             int in_dim, int hidden, void* workspace, size_t workspace_bytes, hypad_stream_t stream);
/* The training form and its back-propagation through time (ABI 7) -- what autograd gives the nn.LSTM modules of
 * models/tadgan.py:15-27, 35-38 at any seq_len.  fwd_train = the forward above that also fills
 *   saved (seq_len, rows, 2, 5, hidden) = per (step, row, direction) [i | f | g | o | c]: gate activations and cell state.
 * bwd: grad_out (seq_len, rows, 2*hidden), grad_hn / grad_cn (2, rows, hidden) -- any of the three may be NULL (= zeros), not all;
 *   grad_x (seq_len, rows, in_dim); grad_w_ih_* (4*hidden, in_dim); grad_w_hh_* (4*hidden, hidden); grad_b_* (4*hidden) = the
 *   gradient of b_ih_* AND of b_hh_* (they enter as a sum); grad_h0 / grad_c0 (2, rows, hidden) may be NULL.  h0 / c0 / out as
 *   the forward took / returned them.  A persistent kernel per (16-row tile, direction) walks the steps in reverse (W_hh^T in
 *   LDS, carried dh / dc in registers, one barrier per step) and writes the pre-activation gradients of every step; the
 *   parameter / input gradients are dense contractions over seq_len * rows rows (hypad_linear_act_bwd).  hidden <= 64.
 * workspace: hypad_lstm_seq_bwd_workspace_bytes(seq_len, rows, in_dim, hidden). */
int hypad_lstm_bidir_seq_fwd_train(const float* x, const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                                   const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                                   const float* h0, const float* c0, float* out, float* hn, float* cn, float* saved, int seq_len,
                                   int64_t rows, int in_dim, int hidden, void* workspace, size_t workspace_bytes, hypad_stream_t stream);
size_t hypad_lstm_seq_bwd_workspace_bytes(int seq_len, int64_t rows, int in_dim, int hidden);
int hypad_lstm_bidir_seq_bwd(const float* x, const float* w_ih_f, const float* w_hh_f, const float* w_ih_r, const float* w_hh_r,
                             const float* h0, const float* c0, const float* out, const float* saved, const float* grad_out,
                             const float* grad_hn, const float* grad_cn, float* grad_x, float* grad_w_ih_f, float* grad_w_hh_f,
                             float* grad_b_f, float* grad_w_ih_r, float* grad_w_hh_r, float* grad_b_r, float* grad_h0, float* grad_c0,
                             int seq_len, int64_t rows, int in_dim, int hidden, void* workspace, size_t workspace_bytes,
                             hypad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Networks, forward (eval or train-mode dropout).  `params` = arena of that network.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hypad_dropout {
  int train_mode;        /* 0: eval (no dropout).  1: dropout active as after .train() */
  const float* masks;    /* optional injected masks holding 0 or 1/(1-p); layout stated per function */
  uint64_t seed;         /* used when train_mode && !masks: device Philox4x32-10 */
  uint64_t offset;       /* Philox stream offset (e.g. an iteration counter) */
} hypad_dropout;

/* Encoder.forward  models/tadgan.py:23-27: x (rows, S) -> (rows, L) */
int hypad_encoder_fwd(const float* params, const float* x, float* out, int64_t rows, int signal_shape,
                      int latent_dim, hypad_stream_t stream);
/* Decoder.forward  models/tadgan.py:58-67: z (rows, L) -> eucl (rows, S) and, if hyperbolic, hyper (rows, S).
 * dropout mask layout: (rows, 128) inter-layer LSTM mask (p = 0.2). */
int hypad_decoder_fwd(const float* params, const float* z, float* hyper_out, float* eucl_out, int64_t rows,
                      int signal_shape, int latent_dim, int hyperbolic, const hypad_dropout* drop, hypad_stream_t stream);
/* CriticX.forward  models/tadgan.py:91-106: x (rows, S) -> (rows,).  masks: 4 x (rows, L), p = 0.25 */
int hypad_critic_x_fwd(const float* params, const float* x, float* out, int64_t rows, int signal_shape,
                       int latent_dim, const hypad_dropout* drop, hypad_stream_t stream);
/* CriticZ.forward  models/tadgan.py:123-132: z (rows, L) -> (rows,).  masks: 2 x (rows, L), p = 0.2 */
int hypad_critic_z_fwd(const float* params, const float* z, float* out, int64_t rows, int latent_dim,
                       const hypad_dropout* drop, hypad_stream_t stream);
/* test_tadgan batch body  anomaly_detection.py:67-113 (eval mode), fused: for every window row
 *   lat = Encoder(x); (hyper, eucl) = Decoder(lat); hyper_real = hyperbolic_linear(x); critic = CriticX(x);
 *   rowdist = poincare distance(hyper_real, hyper)  (utils/anomaly_detection_utils.py:58-66).
 * Any output pointer may be NULL (that output is then not written).  Euclidean mode: recon = eucl only. */
/* The same on the training kernels' machinery (MFMA-native packed weights built into `workspace` by the call, LSTM cells in
 * the gate products' epilogues, 512 threads per 16 windows): ~4x the rate of hypad_score_forward at large `rows`.
 * x_row_stride: 0 / S = window matrix, 1 = x is the scaled series and window n is x[n .. n+S) (no matrix at all); any other
 *   stride in floats up to 2^24 (a tile's rows are addressed with 32-bit byte offsets: 16 rows x 2^24 floats x 4 bytes): a larger
 *   one returns HYPAD_EUNSUPPORTED (hypad_score_forward takes contiguous rows only; gather such rows first).
 * Which tile form runs (16 or 32 windows per workgroup; the latter from 65 536 windows on, window 100 / latent 20 only) depends on
 *   the arguments alone and changes no result bit.
 * workspace: hypad_score_workspace_bytes(S, L, hyperbolic). */
size_t hypad_score_workspace_bytes(int signal_shape, int latent_dim, int hyperbolic);
int hypad_score_forward_packed(const float* enc, const float* dec, const float* cx, const float* x, int64_t x_row_stride,
                               float* hyper, float* eucl, float* hyper_real, float* critic, float* rowdist, int64_t rows,
                               int signal_shape, int latent_dim, int hyperbolic, void* workspace, size_t workspace_bytes,
                               hypad_stream_t stream);
int hypad_score_forward(const float* enc, const float* dec, const float* cx, const float* x,
                        float* hyper, float* eucl, float* hyper_real, float* critic, float* rowdist,
                        int64_t rows, int signal_shape, int latent_dim, int hyperbolic, hypad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Training iterations (train.py:18-104, :107-186, :189-249) with the optimizer step fused in
 * (torch.optim.Adam train.py:274-281; geoopt RiemannianAdam train.py:282-288).
 * ---------------------------------------------------------------------------------------------- */
typedef struct hypad_dims {
  int signal_shape;  /* params.signal_shape */
  int latent_dim;    /* params.latent_space_dim (train.py:413) */
  int batch;         /* params.batch_size; multiple of 16 */
  int hyperbolic;    /* params.hyperbolic */
  int n_signals;     /* independent models trained side by side (one per signal); >= 1 */
  int first_signal;  /* ABI 5: model s of this call draws its device random streams (latent vectors, interpolation weights, dropout)
                        as stream first_signal + s.  A signal's training then does not depend on which models share its launches:
                        model k of a group whose first_signal is f == a single model trained with first_signal = f + k, bit for bit
                        (hypad_amd.train.train_signals_resident: one model per signal, train.py:428-437).  0 = streams 0 .. n_signals-1 */
} hypad_dims;

typedef struct hypad_nets { float *enc, *dec, *cx, *cz; } hypad_nets;

typedef struct hypad_train_state {
  hypad_nets params;       /* signal s of a net at base + s * hypad_param_count(net) */
  hypad_nets exp_avg;      /* Adam first moment, same layout */
  hypad_nets exp_avg_sq;   /* Adam second moment */
  int32_t* counters;       /* device int32[8]: [0..2] optimizer steps taken by {critic_x, critic_z, generator}, [3] rng ticks,
                              [4] STATUS of hypad_train_epoch's resident critic launch (0 = fine; otherwise the code of the first
                              bounded wait that gave up, see hypad_epoch_status), [5] how many critics of the last hypad_train_epoch's resident
                              launch had all their chunk workgroups on one XCD (a speed matter only), [6..7] reserved (zero) */
  float lr, beta1, beta2, eps;
  float gen_weight_decay;  /* hyperbolic generator optimizer: 1e-5 (train.py:286); ignored otherwise */
  int gen_stabilize;       /* 10 (train.py:287) */
} hypad_train_state;

typedef struct hypad_iter_io {
  const float* x;            /* window matrix resident in HBM: (n_signals, n_windows, S) fp32 */
  int64_t x_signal_stride;   /* floats between signals */
  int64_t x_row_stride;      /* floats between consecutive window rows of x: 0 or signal_shape = a dense (n_windows,
                                signal_shape) matrix; 1 = x is the scaled series itself and window n is x[n .. n + signal_shape)
                                (utils/dataloader.py:139-222 materialises exactly that sliding view) */
  const int32_t* row_index;  /* (batch) rows of x forming this minibatch (shared by all signals); NULL = 0..batch-1 */
  const float* z;            /* injected N(0,1) latent draw (n_signals, batch, L) or NULL = device Philox (train.py:24,118,205) */
  const float* alpha;        /* injected U(0,1) interpolation weights (n_signals, batch, S or L) or NULL (train.py:64,149) */
  hypad_dropout drop;        /* masks layout per signal:  critic_x_iteration: valid 4x(B,L) | fake 4x(B,L) | interpolated 4x(B,L) | decoder (B,128)
                                                          critic_z_iteration: fake 2x(B,L) | valid 2x(B,L) | interpolated 2x(B,L)
                                                          decoder_iteration : critic_z 2x(B,L) | critic_x 4x(B,L) | decoder(z) (B,128) | decoder(enc(x)) (B,128) */
  float* losses;             /* device out (n_signals, 4): [loss, aux (hyper_loss or mse), mean critic term a, mean critic term b] */
  void* workspace;
  size_t workspace_bytes;    /* >= hypad_train_workspace_bytes(dims) */
} hypad_iter_io;

size_t hypad_train_workspace_bytes(const hypad_dims* dims);
/* The training workspace also holds MFMA-native packed copies of the generator's weights (blocks of 16 output rows x 16
 * reduction columns laid out as the matrix cores consume them; forward and transposed).  The library builds them itself
 * (every hypad_decoder_iteration call; once per hypad_train_epoch, whose dW + Adam kernel then keeps them current), so
 * callers never need these two: hypad_pack_generator rebuilds the copies of every signal from the parameter arenas,
 * hypad_packed_region reports where they are (floats from the start of a signal's workspace slice, slice stride and
 * count) -- for tests and tools. */
int hypad_pack_generator(const hypad_dims* dims, const hypad_train_state* st, void* workspace, size_t workspace_bytes,
                         hypad_stream_t stream);
int hypad_packed_region(const hypad_dims* dims, int64_t* offset_floats, int64_t* signal_stride_floats, int64_t* count_floats);
/* critic_x_iteration train.py:18-104 / critic_z_iteration :107-186 / decoder_iteration :189-249 -- one optimizer step each.
 * The two critic entry points have two forms with the same results up to floating-point summation order (and the device dropout
 * streams, where the masks are not injected): three stand-alone launches (pass, gradient penalty, dW + Adam), or -- when
 * io->workspace_bytes >= hypad_epoch_workspace_bytes(dims, 1, 1) and the shape fits the epoch's critic kernels -- a one-iteration
 * phase of the epoch's hoisted form (pack of the frozen generator half, record precompute, iteration launch, finalising launch,
 * counter advance): 45 / 29 us of GPU time per call instead of 66 / 39 at the reference configuration, five launches instead of
 * three.  HYPAD_ITER_PHASE=0 in the environment keeps the stand-alone launches whatever the workspace. */
int hypad_critic_x_iteration(const hypad_dims* dims, const hypad_train_state* st, const hypad_iter_io* io, hypad_stream_t stream);
int hypad_critic_z_iteration(const hypad_dims* dims, const hypad_train_state* st, const hypad_iter_io* io, hypad_stream_t stream);
int hypad_decoder_iteration(const hypad_dims* dims, const hypad_train_state* st, const hypad_iter_io* io, hypad_stream_t stream);

/* One epoch of train_tadgan's loops (train.py:299-356): n_critics passes of (critic_x, critic_z) over every
 * minibatch, then one generator pass.  row_index: ((n_critics + 1), n_batches * batch) int32 window rows (the
 * DataLoader's shuffles); noise and dropout come from device Philox (seed).  losses: (n_signals,
 * (2 * n_critics + 1) * n_batches, 4) in launch order. */
/* Optional injected randomness of a whole epoch (parity runs; a NULL plane = device Philox keyed by `seed`).  Planes are
 * iteration-major, so one iteration's slice has exactly the layout hypad_iter_io states for that iteration:
 * critic iteration it = pass * n_batches + batch (launch order), generator iteration b = batch. */
typedef struct hypad_epoch_noise {
  const float* z_cx;      /* (n_critics * n_batches, n_signals, batch, L)  N(0,1), decoder input of critic_x_iteration  train.py:24 */
  const float* alpha_cx;  /* (n_critics * n_batches, n_signals, batch, S)  U[0,1)                                        train.py:64 */
  const float* z_cz;      /* (n_critics * n_batches, n_signals, batch, L)  N(0,1), `valid` latent of critic_z_iteration   train.py:118 */
  const float* alpha_cz;  /* (n_critics * n_batches, n_signals, batch, L)                                                train.py:149 */
  const float* z_gen;     /* (n_batches, n_signals, batch, L)              decoder_iteration                              train.py:205 */
  /* dropout keep-scales (0 or 1/(1-p)), read when train_mode != 0: all three planes or none.  Per (iteration, signal)
   * the layouts of hypad_iter_io.drop for critic_x_iteration / critic_z_iteration / decoder_iteration. */
  const float* masks_cx;
  const float* masks_cz;
  const float* masks_gen;
} hypad_epoch_noise;

typedef struct hypad_epoch_io {
  const float* x; int64_t x_signal_stride;
  int64_t x_row_stride;      /* as in hypad_iter_io */
  const int32_t* row_index;  /* (n_critics + 1, n_batches * batch): the passes' shuffles */
  int n_batches, n_critics;
  int train_mode; uint64_t seed;
  float* losses;
  void* workspace; size_t workspace_bytes;   /* >= hypad_train_workspace_bytes(dims); with >= hypad_epoch_workspace_bytes(...)
                                                the critic phase runs in its hoisted form (see below) */
  const hypad_epoch_noise* noise;            /* NULL: all randomness from device Philox(seed) */
  int flags;                                 /* HYPAD_EPOCH_* bits, 0 = defaults */
  /* ABI 4.  Optional: n_aux_streams (<= 7) further streams of the same device.  With more than one signal (model) per call the
   * generator phase (train.py:347-352: 29 x [decoder_iteration, its optimizer step] per model, the models independent of each other)
   * then runs the models in n_aux_streams + 1 groups, each group's chain of launches on a stream of its own -- forked from `stream`
   * by an event after the critic phase, joined into it before the call returns its last launches -- so one group's optimizer
   * launch overlaps another group's generator launch.  Results are the same bits with any number of streams (the step number and
   * rng tick of every launch are its own arguments; the counters advance once, after the join).  Capturable like everything else:
   * the groups become parallel branches of the captured graph.  NULL / 0: everything on `stream`.  Calls that pass auxiliary streams
   * share one process-wide set of fork / join events: issue them from one host thread at a time. */
  hypad_stream_t const* aux_streams; int n_aux_streams;
  /* ABI 5.  int32 elements between the row_index planes of consecutive signals: 0 = ONE plane shared by all signals (every model sees
   * the same shuffles); (n_critics + 1) * n_batches * batch or more = a plane per signal -- signals of different lengths (each with
   * permutations of its OWN window count, hypad_epoch_shuffles_signals) trained by the same launches. */
  int64_t row_index_signal_stride;
  /* ABI 6.  Optional scratch, n_signals x enc_table_rows x latent_dim floats, enc_table_rows = number of window rows of `x` a row
   * index may name (every model: rows 0 .. enc_table_rows - 1 of its x must be readable).  The encoder is frozen through the critic
   * phase (train.py:306-309) and every pass shuffles the same windows, so encoder(x) -- critic_z's fake input, train.py:111-116 --
   * is then evaluated ONCE per window row in front of the phase (one launch) and gathered by the record producers, instead of once
   * per pass (n_critics times): worth it from 14 models per call on, where the producers bound the phase (16 models: 4.25 -> 3.97 ms
   * per epoch, 32: 5.99 -> 5.76); the same bits either way (the encoder output of a row does not depend on the other rows of its tile).  NULL: off. */
  float* enc_table; int64_t enc_table_rows;
} hypad_epoch_io;
enum {
  HYPAD_EPOCH_PER_ITERATION = 1,             /* run the critic phase as one launch per iteration even where the resident form fits */
  /* A/B switches (tests, timing).  The library reads NO environment variable: which kernels a call launches depends on its
   * arguments alone.  Every combination gives the same results bit for bit, except PER_ITERATION / PER_MINIBATCH (another
   * summation order of the critics' gradient shares). */
  HYPAD_EPOCH_NO_PRODUCERS = 2,              /* resident critic launch behind a precompute launch instead of with its own record producers */
  HYPAD_EPOCH_ID_ORDER = 4,                  /* resident critic workgroups in id order (no per-XCD placement: shares go through memory) */
  HYPAD_EPOCH_CLEAR_TILES = 8,               /* sweep the activation / delta tiles after every resident iteration (round-2 behaviour) */
  HYPAD_EPOCH_PER_MINIBATCH = 16,            /* no hoisting at all: one launch group per minibatch (critic_x || critic_z pass, penalty, dW) */
  HYPAD_EPOCH_DW_COLOC = 32,                 /* dW + Adam workgroups co-located per model and XCD (the default from 8 models per call on) */
  HYPAD_EPOCH_DW_SPREAD = 64,                /* ... spread over the chip (the default below 8 models) */
  HYPAD_EPOCH_TEST_GIVE_UP_SHIFT = 8         /* tests only: bits 8..15 = k > 0 makes the resident launch behave as if its wait for the
                                                siblings' shares had timed out at critic iteration k (signal 0, critic_x) */
};
/* The DataLoader's shuffles of one epoch (main.py:38: shuffle=True, drop_last=True; train.py:315-351 iterates the loader once per
 * pass) drawn on the device: row_index (n_passes, take) int32 <- for every pass the first `take` = n_batches * batch entries of an
 * independent uniform random permutation of [0, n_windows) (argsort of Philox keys, keyed by seed, pass and the rng tick
 * counters[3] -- NULL: tick 0 -- so every epoch of a replayed graph is shuffled afresh).  Capturable: an epoch including its
 * shuffles is then a fixed launch sequence with no host work at all.  n_windows <= 4096 (one workgroup sorts a pass in LDS);
 * HYPAD_EUNSUPPORTED beyond: draw the permutations with any other generator and pass them to hypad_train_epoch as before. */
int hypad_epoch_shuffles(int32_t* row_index, int n_passes, int take, int n_windows, uint64_t seed, const int32_t* counters,
                         hypad_stream_t stream);
/* The same for n_signals models of different lengths in one launch: signal s gets its own plane row_index + s * signal_stride
 * (n_passes, take) of permutations of [0, n_windows[s]) (n_windows: DEVICE int32[n_signals], each in [take, 4096]), keyed by
 * (seed, first_signal + s, pass, tick): the plane of signal s == hypad_epoch_shuffles' for one model with that seed and
 * first_signal + s (the key folds the stream number in; stream 0 is the key of hypad_epoch_shuffles itself). */
int hypad_epoch_shuffles_signals(int32_t* row_index, int64_t signal_stride, int n_signals, int first_signal, const int32_t* n_windows,
                                 int n_passes, int take, uint64_t seed, const int32_t* counters, hypad_stream_t stream);

/* HOST helper (the one entry point whose pointers are host pointers; no device work, no stream): the latent draws of a whole epoch
 * of the reference's loop -- np.random.normal(size=(1, batch, L)) on NumPy's GLOBAL generator, once per iteration, train.py:24,118,205
 * -- continued from that generator's state (np.random.get_state(): MT19937 key[624], pos, has_gauss, cached_gaussian; all updated in
 * place: hand them back with np.random.set_state()) and written as float32 (torch.Tensor(float64 array), train.py:24) into the pinned
 * planes hypad_epoch_noise is uploaded from.  Draw order: for r in [0, rounds): for k in [0, n_outs): `chunk` values -> outs[k] + r*chunk
 * (critic phase: outs = {z_cx, z_cz}, chunk = batch * L, rounds = n_critics * n_batches: critic_x_iteration draws before
 * critic_z_iteration, train.py:320-327; generator pass: outs = {z_gen}).  NumPy's legacy polar Box-Muller bit for bit
 * (tests/test_host_rng.py); unlike np.random.normal the call does not hold the interpreter lock. */
int hypad_host_mt19937_normal(uint32_t* key, int* pos, int* has_gauss, double* cached_gaussian, float* const* outs, int n_outs,
                              int64_t chunk, int64_t rounds);
/* hypad_host_mt19937_normal with `threads` helper threads: the calling thread runs the generator and the polar method's rejection test
 * (sequential), the helpers take the accepted pairs' transforms -- sqrt(-2 log(r2) / r2), two products, the float32 stores; independent
 * per pair -- block by block as they become ready.  Same values, same final state; threads < 1 = hypad_host_mt19937_normal.  Worth it
 * from about a million values per call on (configs[3]: 4.1 M per epoch). */
int hypad_host_mt19937_normal_mt(uint32_t* key, int* pos, int* has_gauss, double* cached_gaussian, float* const* outs, int n_outs,
                                 int64_t chunk, int64_t rounds, int threads);
/* The interpolation weights of train.py:64,149 -- torch.rand(...) on torch's default CPU generator -- for a whole pass in one call:
 * `state` = the bytes of torch.get_rng_state() (5 056: at::CPUGeneratorImplState, an at::mt19937 inside), advanced in place as n draws
 * leave it; out[i] = the i-th float32 torch.rand would have returned.  HOST pointers, like hypad_host_mt19937_normal. */
int hypad_host_torch_mt19937_uniform(void* state, size_t state_bytes, float* out, int64_t n);

/* Workspace that lets hypad_train_epoch hoist the frozen generator's forwards (decoder(z_i), encoder(x_i) of every
 * critic iteration, train.py:306-328) out of the sequential critic chain: hypad_train_workspace_bytes plus room for up
 * to 512 iterations of precomputed rows (longer phases are processed in chunks).  Same random streams and the same
 * arithmetic per row as the per-iteration entry points; only floating-point summation order differs. */
size_t hypad_epoch_workspace_bytes(const hypad_dims* dims, int n_batches, int n_critics);
/* 1 when hypad_train_epoch (given that workspace) runs the critic phase of these dimensions as ONE resident launch -- every
 * (signal, critic, 16-row chunk) workgroup stays on its CU for all iterations, weights in LDS, Adam state in registers, gradient
 * shares exchanged through write-through stores and epoch words -- instead of one launch per iteration.  Needs
 * 2 * n_signals * batch / 16 <= CUs of the device; HYPAD_CRITIC_PERSISTENT=0 in the environment selects the launches. */
int hypad_critic_phase_persistent(const hypad_dims* dims);
/* 1 when that resident launch also produces the phase's records itself (extra workgroups behind the resident ones: no separate
 * precompute launch) for a phase of n_iters = n_critics * n_batches iterations: where the resident critics hold at most half of
 * the device's CUs (the producers need the others); hypad_epoch_io.flags & HYPAD_EPOCH_NO_PRODUCERS turns it off. */
int hypad_critic_phase_producers(const hypad_dims* dims, int n_iters);
int hypad_train_epoch(const hypad_dims* dims, const hypad_train_state* st, const hypad_epoch_io* io, hypad_stream_t stream);

/* Status channel of the resident critic launch.  That launch needs every one of its critic workgroups co-resident (one per CU:
 * the launcher checks the grid against the device's CU count and the kernel's occupancy, but a CU mask, a partitioned or shared
 * device can still withhold CUs) and all its waits are bounded: a wait that gives up stores its code -- 0x100 + it: a sibling's
 * gradient share of critic iteration `it` never arrived; 0x200 + it: a sibling's scalars; 0x300 + it: a record -- into
 * counters[4] (first code wins), writes NaN into that iteration's loss row, and the launch ends.  From then on every training
 * launch of hypad_train_epoch on this state is a no-op (fail-stop: the generator is never stepped against half-updated
 * critics, the snapshot below is never overwritten) until hypad_epoch_restore.
 *   hypad_epoch_status : copies counters[4] to *status_host and SYNCHRONISES the stream (not capturable).
 *   hypad_epoch_restore: puts both critics' parameters and moments and counters[0..3] back to what they were when the failed
 *                        hypad_train_epoch call began (it snapshots them into its workspace first: 50 KB per signal) and clears
 *                        counters[4]; the caller then repeats that epoch with HYPAD_EPOCH_PER_ITERATION -- same random streams:
 *                        bit for bit the epoch a healthy run in that form produces (the two forms differ only in floating-point
 *                        summation order).  Capturable; workspace = the failed call's. */
int hypad_epoch_status(const hypad_train_state* st, int* status_host, hypad_stream_t stream);
int hypad_epoch_restore(const hypad_dims* dims, const hypad_train_state* st, void* workspace, size_t workspace_bytes,
                        hypad_stream_t stream);

/* Where hypad_train_epoch's hoisted critic phase left its precomputed records (tests and tools; valid after a call with
 * n_critics * n_batches <= 512 iterations): records of critic `critic` (0 = critic_x, 1 = critic_z) start `offset_floats` floats
 * into the workspace and are indexed (signal, iteration, batch / 16 chunks); one record is `record_floats` floats:
 * [48][row_stride] input rows (16 real | 16 fake | 16 interpolated; the input, a constant-one column, zero padding) followed,
 * at `mask_offset_floats`, by the dropout keep-scales [n_layers][48][mask_row_stride] (pass order real, fake, interpolated) and
 * a tail of 32 floats (Adam's bias corrections of the step the next iteration applies, then zeros: records stay 128-byte
 * aligned).  With the resident form of the phase the records are written by producer workgroups of that launch itself
 * (HYPAD_CRITIC_PRODUCERS=0 in the environment: by a launch in front of it) -- same layout, same bits. */
typedef struct hypad_record_info {
  int64_t offset_floats;
  int record_floats, row_stride, mask_offset_floats, mask_row_stride, n_layers, in_dim;
} hypad_record_info;
int hypad_epoch_record_info(const hypad_dims* dims, int n_batches, int n_critics, int critic, hypad_record_info* out);

/* The device random streams the training kernels draw from when no plane is injected -- Philox4x32-10 keyed by (seed, tick,
 * stream, signal), element index = position in the (batch, width) matrix -- exported for distribution tests.
 * kind: 0 = N(0,1) (z), 1 = U[0,1) (alpha), 2 = dropout keep-scale 0 | 1/(1-p_drop).  tick = counters[3] at the iteration.
 * streams: HYPAD_STREAM_*; a critic's dropout stream is HYPAD_STREAM_DROP_CRITIC + 8 * pass + layer (pass: critic_x 0 valid,
 * 1 fake, 2 interpolated; critic_z 0 fake, 1 valid, 2 interpolated).  Inside hypad_train_epoch the critic_z side draws with
 * hypad_critic_z_seed(seed). */
enum { HYPAD_RNG_NORMAL = 0, HYPAD_RNG_UNIFORM = 1, HYPAD_RNG_DROPOUT = 2 };
enum { HYPAD_STREAM_Z = 1, HYPAD_STREAM_ALPHA = 2, HYPAD_STREAM_DROP_DEC0 = 3, HYPAD_STREAM_DROP_DEC1 = 4, HYPAD_STREAM_DROP_CRITIC = 16 };
int hypad_rng_fill(int kind, uint64_t seed, uint32_t tick, uint32_t rng_stream, uint32_t signal, float p_drop, float* out,
                   int64_t n, hypad_stream_t stream);
uint64_t hypad_critic_z_seed(uint64_t seed);

/* Measurement aid (bench.py): run ONE iteration (kind 0 = critic_x, 1 = critic_z, 2 = decoder, 3 = the critic_x ||
 * critic_z pair of the per-iteration path) or, kind 4, 145 iterations of the hoisted critic phase of hypad_train_epoch
 * over rows 0 .. batch-1 (row_index is ignored), with HIP events recorded on `stream` between the kernels; synchronise;
 * return the per-kernel durations in ms: critic iterations -> {pass kernel, gradient-penalty kernel, dW+Adam};
 * decoder -> {generator kernel, dW+Adam}; kind 4 -> {precompute kernel (the 145 iterations' records), X, mean time of one
 * critic_x || critic_z iteration}: when the phase runs as one resident launch (hypad_critic_phase_persistent) X is the
 * re-initialisation of its epoch words and the mean is that launch's duration / 145; otherwise X is the first iteration launch
 * (no Adam in its prologue) and the mean is over the 144 steady-state launches that follow back to back.  losses: room for
 * 2 * n_signals * 4 floats (kind 3) / 290 * n_signals * 4 floats (kind 4); workspace for kind 4:
 * hypad_epoch_workspace_bytes(dims, 145, 1).  kind 5 = the decoder iteration's two kernels as hypad_train_epoch launches them (the decay-only tensors left to the epoch's
 * decay launch), each launched 64 times back to back
 * between its events -> {mean generator kernel, mean dW+Adam}: an event pair around ONE launch of a 9 - 40 us kernel also measures
 * the event path (the sum of such figures exceeded the epoch they were taken from); the 64 dW+Adam launches are 64 optimizer steps
 * on the same gradient (the weights move: measure after, not inside, a run whose results matter).
 * Not capturable into a graph. */
int hypad_profile_iteration(int kind, const hypad_dims* dims, const hypad_train_state* st, const hypad_iter_io* io,
                            float* ms_out, int n_out, hypad_stream_t stream);

/* Stand-alone optimizer steps over a flat arena given its gradient arena.
 * torch.optim.Adam (train.py:274-281): step_index = 1-based step number. */
int hypad_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                    int step_index, float lr, float beta1, float beta2, float eps, float weight_decay,
                    hypad_stream_t stream);
/* geoopt.optim.RiemannianAdam (train.py:282-288; geoopt==0.5.0, restated -- see oracle/radam.py):
 * Euclidean rule on [0, n) except the ball-valued vector [ball_offset, ball_offset + ball_dim) (ball_dim = 0: none). */
int hypad_radam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                     int64_t ball_offset, int ball_dim, int step_index, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int stabilize, hypad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Window scoring (utils/anomaly_detection_utils.py).  fp64 where NumPy computes in fp64.
 * ---------------------------------------------------------------------------------------------- */
/* reconstruction_errors un-roll, :918-935: for t in [0, n + S - 1) gather y_hat[t - j, j] over valid j;
 * median (T,) in the input precision (np.median of float32) and, if summary != NULL, (T, 5) fp64
 * [min, p25, p50, p75, max] with NumPy's linear-interpolated percentiles. */
int hypad_unroll_median(const float* y_hat, float* median, double* summary, int64_t n, int window, hypad_stream_t stream);
/* :908-910 -- true[t] = y[t][0] (t < n), y[n-1][t-n+1] otherwise; y (n, S) fp64 */
int hypad_unroll_true(const double* y, double* out, int64_t n, int window, hypad_stream_t stream);
/* the same from the fp32 window matrix the forward reads (row n at y + n * row_stride: row_stride = S for a matrix, 1 for the
 * scaled series itself) -- no fp64 copy of the (n, S) matrix for the sake of n + S - 1 of its values */
int hypad_unroll_true_f32(const float* y, int64_t row_stride, double* out, int64_t n, int window, hypad_stream_t stream);
/* _point_wise_error :761-777 */
int hypad_point_error(const double* y, const float* y_hat, double* out, int64_t t, hypad_stream_t stream);
/* _area_error :780-812 (centred rolling trapezoid, window score_window, min_periods score_window/2) */
int hypad_area_error(const double* y, const float* y_hat, double* out, int64_t t, int score_window, hypad_stream_t stream);
/* _dtw_error :815-863 (pyts.metrics.dtw classic, squared cost, sqrt of the accumulated cost) */
int hypad_dtw_error(const double* y, const float* y_hat, double* out, int64_t t, int score_window, hypad_stream_t stream);
/* pandas rolling(window, center=True, min_periods=window/2).mean()  :953-961, :325-330 -- of in[], or, with sub != NULL, of the
 * point-wise error |in[i] - sub[i]| (:761-777 fused).  Windows wider than 32 are summed from two levels of pre-summed chunks
 * (16 and 256 elements) aligned to the absolute index origin + i, in one canonical order: O(60 + window / 256) additions per
 * output instead of O(window), and a caller that smooths a slice [origin, origin + t) of a longer series gets, for every
 * timestep whose window lies inside the slice, the bits the whole series gives (sharded scoring).
 * workspace: hypad_rolling_workspace_bytes(t) (may be NULL for window <= 32). */
size_t hypad_rolling_workspace_bytes(int64_t t);
int hypad_rolling_mean(const double* in, const float* sub, double* out, int64_t t, int window, int64_t origin, void* workspace,
                       size_t workspace_bytes, hypad_stream_t stream);
/* workspace of hypad_zscore_clip / hypad_critic_zscore: the per-slice partial statistics of their first launch */
#define HYPAD_STATS_WORKSPACE_BYTES (256 * 5 * 8)
/* stats.zscore -> clip(min=0) + 1  :523-524,542-543.  workspace: HYPAD_STATS_WORKSPACE_BYTES */
int hypad_zscore_clip(const double* in, double* out, int64_t t, void* workspace, size_t workspace_bytes, hypad_stream_t stream);
/* final_critic_scores :365-404 (also :470-504), KDE step: modes (n + window - 1) fp64 -- for every un-rolled timestep
 * the window-critic value at which scipy.stats.gaussian_kde of the covering windows' values peaks (median fallback). */
int hypad_kde_mode(const float* critic, double* modes, int64_t n, int window, hypad_stream_t stream);
/* _compute_critic_score :307-322: out = |x - mean(x within [q25, q75])| / std(x) + 1 (quantiles supplied by the caller;
 * the rolling mean of :325-330 is hypad_rolling_mean).  workspace: HYPAD_STATS_WORKSPACE_BYTES */
int hypad_critic_zscore(const double* in, double q25, double q75, double* out, int64_t t, void* workspace,
                        size_t workspace_bytes, hypad_stream_t stream);
/* np.quantile(in, q) (method "linear") for nq <= 2 quantiles of n fp64 values, out (nq) fp64 ON THE DEVICE: exact order
 * statistics by radix selection on the keys (no sort, no host round trip), numpy's interpolation; any NaN -> NaN.
 * :319-320 (`np.quantile(critics, 0.25)`, `np.quantile(critics, 0.75)`).  workspace: hypad_quantile_workspace_bytes() */
size_t hypad_quantile_workspace_bytes(void);
int hypad_quantiles(const double* in, int64_t n, const double* q, int nq, double* out, void* workspace, size_t workspace_bytes,
                    hypad_stream_t stream);
/* _compute_critic_score :307-322 with the two quantiles taken on the device (hypad_quantiles), then as hypad_critic_zscore.
 * workspace: hypad_critic_score_workspace_bytes() */
size_t hypad_critic_score_workspace_bytes(void);
int hypad_critic_score(const double* in, double* out, int64_t t, void* workspace, size_t workspace_bytes, hypad_stream_t stream);
/* np.linalg.norm(recons, axis=1)  :341,347,350,359 */
int hypad_row_norms(const float* x, double* out, int64_t rows, int dim, hypad_stream_t stream);
/* combine_scores :336-362 */
enum { HYPAD_COMB_SUM = 0, HYPAD_COMB_MULT = 1, HYPAD_COMB_UNCERTAINTY = 2, HYPAD_COMB_CRITIC = 3,
       HYPAD_COMB_CRITIC_UNCERTAINTY = 4, HYPAD_COMB_SUM_UNCERTAINTY = 5, HYPAD_COMB_REC = 6,
       HYPAD_COMB_REC_UNCERTAINTY = 7,
       /* score_anomalies tail :553-570 */
       HYPAD_COMB_EUCL_MULT = 8, HYPAD_COMB_EUCL_SUM = 9 };
int hypad_combine_scores(int combination, const double* critic_scores, const double* rec_scores,
                         const double* uncertainty, double* out, int64_t n, hypad_stream_t stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HYPAD_H_ */
