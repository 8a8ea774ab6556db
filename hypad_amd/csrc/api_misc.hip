// ABI book-keeping: version, error strings, limits and the parameter-arena catalogue the host side builds
// its tensor views from (names/shapes = the reference's state_dict).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/hypad.h"
#include "device_utils.h"
#include "layout.h"

using namespace hypad;

namespace {

struct TensorInfo {
  char name[64];
  int offset, rows, cols;
};

void add(std::vector<TensorInfo>& v, const char* name, int off, int rows, int cols) {
  TensorInfo t;
  std::snprintf(t.name, sizeof(t.name), "%s", name);
  t.offset = off; t.rows = rows; t.cols = cols;
  v.push_back(t);
}
void add_lstm(std::vector<TensorInfo>& v, const LstmDir& d, int layer, bool reverse, int H, int in) {
  char n[64];
  const char* sfx = reverse ? "_reverse" : "";
  std::snprintf(n, sizeof(n), "lstm.weight_ih_l%d%s", layer, sfx); add(v, n, d.w_ih, 4 * H, in);
  std::snprintf(n, sizeof(n), "lstm.weight_hh_l%d%s", layer, sfx); add(v, n, d.w_hh, 4 * H, H);
  std::snprintf(n, sizeof(n), "lstm.bias_ih_l%d%s", layer, sfx); add(v, n, d.b_ih, 4 * H, 0);
  std::snprintf(n, sizeof(n), "lstm.bias_hh_l%d%s", layer, sfx); add(v, n, d.b_hh, 4 * H, 0);
}

bool catalogue(int net, int S, int L, int hyperbolic, std::vector<TensorInfo>& v, int& total) {
  if (S <= 0 || L <= 0) return false;
  switch (net) {
    case HYPAD_NET_ENCODER: {
      EncLayout e = enc_layout(S, L);
      add_lstm(v, e.dir[0], 0, false, ENC_H, S);
      add_lstm(v, e.dir[1], 0, true, ENC_H, S);
      add(v, "dense.weight", e.dense_w, L, 2 * ENC_H);
      add(v, "dense.bias", e.dense_b, L, 0);
      total = e.total;
      return true;
    }
    case HYPAD_NET_DECODER: {
      DecLayout d = dec_layout(S, L, hyperbolic);
      add(v, "dense1.weight", d.d1_w, DEC_D1, L);
      add(v, "dense1.bias", d.d1_b, DEC_D1, 0);
      for (int layer = 0; layer < 2; ++layer)
        for (int dir = 0; dir < 2; ++dir) add_lstm(v, d.l[layer][dir], layer, dir == 1, DEC_H, layer == 0 ? DEC_D1 : 2 * DEC_H);
      add(v, "dense2.weight", d.d2_w, S, 2 * DEC_H);
      add(v, "dense2.bias", d.d2_b, S, 0);
      if (hyperbolic) {
        add(v, "hyperbolic_linear.weight", d.head_w, S, S);
        add(v, "hyperbolic_linear.bias", d.head_b, S, 0);
      }
      total = d.total;
      return true;
    }
    case HYPAD_NET_CRITIC_X:
    case HYPAD_NET_CRITIC_Z: {
      CriticLayout c = net == HYPAD_NET_CRITIC_X ? cx_layout(S, L) : cz_layout(L);
      for (int i = 0; i <= c.nh; ++i) {
        char n[64];
        std::snprintf(n, sizeof(n), "dense%d.weight", i + 1); add(v, n, c.w[i], i == c.nh ? 1 : L, i == 0 ? c.in_dim : L);
        std::snprintf(n, sizeof(n), "dense%d.bias", i + 1); add(v, n, c.b[i], i == c.nh ? 1 : L, 0);
      }
      total = c.total;
      return true;
    }
  }
  return false;
}

// hypad_rng_fill: the training kernels' own draw functions (device_utils.h), one element per thread
__global__ void rng_fill_kernel(int kind, uint64_t seed, uint32_t tick, uint32_t stream, uint32_t sig, float p, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t idx = (uint32_t)i;
    out[i] = kind == HYPAD_RNG_NORMAL    ? rng_normal(seed, tick, stream, sig, idx)
             : kind == HYPAD_RNG_UNIFORM ? rng_uniform(seed, tick, stream, sig, idx)
                                         : rng_dropout(seed, tick, stream, sig, idx, p);
  }
}

// hypad_epoch_shuffles: the DataLoader's shuffle (main.py:38 shuffle=True, drop_last=True) of every pass of an epoch, on the device
// and capturable: pass p = the first `take` entries of a uniform random permutation of [0, n_windows) -- argsort of independent
// Philox keys (64-bit key << 32 | index: no ties), one workgroup per pass, bitonic sort in LDS.  Keyed by (seed, tick, pass) with
// the tick read from the device counters, so a replayed graph draws fresh permutations every epoch.
constexpr int SHUF_MAX = 4096, SHUF_THREADS = 1024, RS_SHUFFLE = 5;
// blockIdx.y = model: its own plane (out + y * signal_stride), window count (n_per_signal[y], or `n` for all) and key -- the seed
// with the model's stream number first_signal + y folded in (stream 0: the seed itself)
__global__ __launch_bounds__(SHUF_THREADS) void epoch_shuffle_kernel(int32_t* __restrict__ out, int take, int n, int npow2, uint64_t seed,
                                                                     const int32_t* __restrict__ counters, int64_t signal_stride = 0,
                                                                     const int32_t* __restrict__ n_per_signal = nullptr, int first_signal = 0) {
  __shared__ unsigned long long keys[SHUF_MAX];
  const uint32_t tick = counters ? (uint32_t)counters[3] : 0u;
  out += (int64_t)blockIdx.y * signal_stride;
  if (n_per_signal) {
    n = n_per_signal[blockIdx.y];
    n = n < take ? take : n > SHUF_MAX ? SHUF_MAX : n;                    // (validated on the host side where the counts are known; never out of LDS)
    npow2 = 2;
    while (npow2 < n) npow2 <<= 1;
  }
  const Philox ph(seed ^ ((uint64_t)(uint32_t)(first_signal + (int)blockIdx.y) * 0x9E3779B97F4A7C15ULL));
  for (int i = threadIdx.x; i < npow2; i += SHUF_THREADS) {
    unsigned long long k = ~0ull;                                        // padding sorts last
    if (i < n) {
      const uint4 r = ph((uint32_t)i >> 2, (uint32_t)RS_SHUFFLE, tick, (uint32_t)blockIdx.x);
      k = ((unsigned long long)pick(r, i & 3) << 32) | (unsigned)i;
    }
    keys[i] = k;
  }
  __syncthreads();
  // 66 compare-exchange stages at 2 048 keys.  As LDS round trips with a barrier each they were the kernel: 0.3 us per stage, 20.6 us per
  // launch.  A stage with stride <= 64 stays inside a block of 128 keys, and a wave holds such a block in registers -- lane l the keys
  // l and 64 + l of the block: stride 64 is the lane's own pair, strides 32 .. 1 exchange with lane l ^ stride (two 32-bit shuffles per
  // key) -- so only the stages with stride >= 128 go through LDS (10 of the 66 at 2 048 keys): the first seven sizes (2 .. 128) are one
  // register run, every later size is its wide stages in LDS and then one register run.  Any correct sort of these keys (all different:
  // the index is their low word) gives the same permutation.
  if (npow2 >= 128) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto xchg = [&](unsigned long long& k, int e, int stride, int size) __attribute__((always_inline)) {      // e: the key's index
      const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)k, stride, 64), hi32 = (unsigned)__shfl_xor((int)(unsigned)(k >> 32), stride, 64);
      const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;
      const bool up = ((e & ~stride) & size) == 0, is_lo = (e & stride) == 0;
      const bool want_min = is_lo == up;
      k = (k < other) == want_min ? k : other;
    };
    auto reg_run = [&](int base, int size_from, int size_to, bool first_strides_only) __attribute__((always_inline)) {
      // sizes size_from .. size_to (each: its strides <= 64) on the block [base, base + 128); first_strides_only: one size, strides 64 .. 1
      unsigned long long ka = keys[base + lane], kb = keys[base + 64 + lane];
      for (int size = size_from; size <= size_to; size <<= 1) {
        int stride = size >> 1;
        if (stride > 64) stride = 64;
        if (stride == 64) {                              // the lane's own pair (lo = base + lane)
          const bool up = ((base + lane) & size) == 0;
          if ((ka > kb) == up) { const unsigned long long t = ka; ka = kb; kb = t; }
          stride = 32;
        }
        for (; stride > 0; stride >>= 1) { xchg(ka, base + lane, stride, size); xchg(kb, base + 64 + lane, stride, size); }
      }
      (void)first_strides_only;
      keys[base + lane] = ka; keys[base + 64 + lane] = kb;
    };
    for (int base = wave * 128; base < npow2; base += (SHUF_THREADS / 64) * 128) reg_run(base, 2, 128, false);
    for (int size = 256; size <= npow2; size <<= 1) {
      for (int stride = size >> 1; stride >= 128; stride >>= 1) {
        __syncthreads();
        for (int t = threadIdx.x; t < npow2 / 2; t += SHUF_THREADS) {
          const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
          const bool up = (lo & size) == 0;
          const unsigned long long a = keys[lo], b = keys[hi];
          if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
        }
      }
      __syncthreads();
      for (int base = wave * 128; base < npow2; base += (SHUF_THREADS / 64) * 128) reg_run(base, size, size, true);
    }
  } else {
  bool wide = true;                                                       // the previous stage exchanged across waves (or the fill did)
  for (int size = 2; size <= npow2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (wide || stride >= 64) __syncthreads();
      else { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f); }
      wide = stride >= 64;
      for (int t = threadIdx.x; t < npow2 / 2; t += SHUF_THREADS) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;     // the pair (lo, lo + stride) of this compare-exchange
        const bool up = (lo & size) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < take; i += SHUF_THREADS) out[(int64_t)blockIdx.x * take + i] = (int32_t)(unsigned)keys[i];
}

}  // namespace

extern "C" {

int hypad_epoch_shuffles(int32_t* row_index, int n_passes, int take, int n_windows, uint64_t seed, const int32_t* counters,
                         hypad_stream_t stream) {
  if (!row_index || n_passes <= 0 || take <= 0 || n_windows < take) return HYPAD_EINVAL;
  if (n_windows > SHUF_MAX) return HYPAD_EUNSUPPORTED;
  int npow2 = 2;
  while (npow2 < n_windows) npow2 <<= 1;
  hipLaunchKernelGGL(epoch_shuffle_kernel, dim3(n_passes), dim3(SHUF_THREADS), 0, (hipStream_t)stream, row_index, take, n_windows, npow2, seed,
                     counters);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_epoch_shuffles_signals(int32_t* row_index, int64_t signal_stride, int n_signals, int first_signal, const int32_t* n_windows,
                                 int n_passes, int take, uint64_t seed, const int32_t* counters, hypad_stream_t stream) {
  if (!row_index || !n_windows || n_signals <= 0 || first_signal < 0 || n_passes <= 0 || take <= 0 || take > SHUF_MAX) return HYPAD_EINVAL;
  if (n_signals > 1 && signal_stride < (int64_t)n_passes * take) return HYPAD_EINVAL;
  hipLaunchKernelGGL(epoch_shuffle_kernel, dim3(n_passes, n_signals), dim3(SHUF_THREADS), 0, (hipStream_t)stream, row_index, take, take, 2, seed,
                     counters, signal_stride, n_windows, first_signal);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_rng_fill(int kind, uint64_t seed, uint32_t tick, uint32_t rng_stream, uint32_t signal, float p_drop, float* out, int64_t n,
                   hypad_stream_t stream) {
  if (kind < 0 || kind > 2 || !out || n < 0 || n > 0xffffffffLL || (kind == 2 && !(p_drop >= 0.f && p_drop < 1.f))) return HYPAD_EINVAL;
  if (n == 0) return HYPAD_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rng_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, kind, seed, tick, rng_stream, signal, p_drop, out, n);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
uint64_t hypad_critic_z_seed(uint64_t seed) { return seed ^ CRITIC_Z_SEED_XOR; }

int hypad_abi_version(void) { return HYPAD_ABI_VERSION; }

const char* hypad_error_string(int code) {
  switch (code) {
    case HYPAD_OK: return "ok";
    case HYPAD_EINVAL: return "invalid argument";
    case HYPAD_EWORKSPACE: return "workspace missing or too small";
    case HYPAD_EUNSUPPORTED: return "shape outside the fused kernels' limits";
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

void hypad_limits(int* max_signal_shape, int* max_latent) {
  if (max_signal_shape) *max_signal_shape = MAX_S;
  if (max_latent) *max_latent = MAX_L;
}

int hypad_param_count(int net, int S, int L, int hyperbolic) {
  std::vector<TensorInfo> v; int total = 0;
  return catalogue(net, S, L, hyperbolic, v, total) ? total : HYPAD_EINVAL;
}
int hypad_param_tensors(int net, int hyperbolic) {
  std::vector<TensorInfo> v; int total = 0;
  return catalogue(net, 100, 20, hyperbolic, v, total) ? (int)v.size() : HYPAD_EINVAL;
}
int hypad_param_info(int net, int S, int L, int hyperbolic, int index, char* name, int name_cap, int* offset, int* rows, int* cols) {
  std::vector<TensorInfo> v; int total = 0;
  if (!catalogue(net, S, L, hyperbolic, v, total) || index < 0 || index >= (int)v.size()) return HYPAD_EINVAL;
  if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s", v[index].name);
  if (offset) *offset = v[index].offset;
  if (rows) *rows = v[index].rows;
  if (cols) *cols = v[index].cols;
  return HYPAD_OK;
}

}  // extern "C"
