// Device helpers: 64-lane wave reductions, Philox4x32-10, activations, launch-error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hypad {

constexpr int WAVE = 64;
// The wave's index in its workgroup as a SCALAR: task / tile selection and weight base addresses derived from it stay on the
// scalar unit (free next to the vector stream: every VALU instruction of the latency-chain kernels costs SIMD time the fp32
// MFMAs cannot overlap, scripts/diag_coissue.py).
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
constexpr float MIN_NORM = 1e-15f;          // math_.py:1134,1269,349
constexpr float TANH_CLAMP = 15.0f;         // math_.py:53
constexpr float ARTANH_EPS = 1e-7f;         // math_.py:58
constexpr float BALL_MAXNORM = 1.0f - 4e-3f;// math_.py:343-349 (fp32), k = -1
constexpr float LEAK = 0.2f;                // models/tadgan.py:76,121

// Sum over the 64 lanes, result in every lane.  Four DPP butterfly steps inside each row of 16 lanes (quad swaps, then
// the half-row and row mirrors), then the four row sums through v_readlane: ~12 VALU instructions and no LDS crossbar
// round trip, against six ds_bpermute hops for the shuffle form (the Moebius head rows are chains of 5-14 of these).
template <int CTRL>
__device__ __forceinline__ float dpp_move_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move_<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_move_<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_move_<0x141>(v);       // row_half_mirror
  v += dpp_move_<0x140>(v);       // row_mirror
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (a + b) + (c + d);
}
// Maximum / minimum over the 64 lanes, result in every lane (same DPP butterfly; six ds_bpermute hops as shuffles)
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move_<0xB1>(v));
  v = fmaxf(v, dpp_move_<0x4E>(v));
  v = fmaxf(v, dpp_move_<0x141>(v));
  v = fmaxf(v, dpp_move_<0x140>(v));
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }
// Sum over each aligned group of 16 lanes (one DPP row), result in every lane of the group: the first four steps above.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move_<0xB1>(v);
  v += dpp_move_<0x4E>(v);
  v += dpp_move_<0x141>(v);
  v += dpp_move_<0x140>(v);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}
// sum over aligned groups of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

// Hardware-transcendental forms (v_exp_f32 / v_rcp_f32): absolute error of a few 1e-7, far inside the 1e-4 parity
// budget, and ~10x cheaper than libm's tanhf/expf in the latency-bound cell epilogues.
// v_rcp_f32 (1 ulp) instead of a correctly rounded reciprocal: the IEEE division sequence is ~10 instructions, and the LSTM
// cells evaluate 16 of these per lane and layer on the generator step's critical chain; the difference (<= 1.2e-7 relative)
// is far below the exp approximation's own error and the 1e-4 parity bar
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// rows x cols sweep of a tile: waves over rows, lanes over columns (no integer division)
template <class F>
__device__ __forceinline__ void tile_for(int rows, int cols, F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < rows; r += nw)
    for (int c = lane; c < cols; c += 64) f(r, c);
}
__device__ __forceinline__ float leaky_slope(float pre) { return pre > 0.0f ? 1.0f : LEAK; }

// ---------------------------------------------------------------------------------------- Philox4x32-10
struct Philox {
  uint32_t key[2];
  __device__ __forceinline__ Philox(uint64_t seed) { key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32); }
  __device__ __forceinline__ uint4 operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const {
    uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
      uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
      uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
      c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
  }
};
__device__ __forceinline__ float u32_to_unit_open(uint32_t x) {   // (0, 1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) {        // [0, 1)
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ uint32_t pick(const uint4& r, int i) {
  return i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w;
}
// critic_z draws from its own key when it runs beside critic_x in one launch group (hypad_critic_z_seed)
constexpr uint64_t CRITIC_Z_SEED_XOR = 0x5851F42D4C957F2DULL;
// Streams (c1): which random tensor of an iteration
enum RngStream : uint32_t {
  RS_Z = 1, RS_ALPHA = 2, RS_DROP_DEC0 = 3, RS_DROP_DEC1 = 4, RS_DROP_CRITIC = 16 /* + pass*8 + layer */
};
// uniform [0,1) for element `idx` of stream `stream` at tick `tick` of model `sig`
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t tick, uint32_t stream, uint32_t sig, uint32_t idx) {
  Philox ph(seed);
  uint4 r = ph(idx >> 2, stream, tick, sig);
  return u32_to_unit(pick(r, idx & 3));
}
__device__ __forceinline__ float rng_normal(uint64_t seed, uint32_t tick, uint32_t stream, uint32_t sig, uint32_t idx) {
  Philox ph(seed);
  uint4 r = ph(idx >> 1, stream, tick, sig);
  float u1 = u32_to_unit_open((idx & 1) ? r.z : r.x), u2 = u32_to_unit((idx & 1) ? r.w : r.y);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}
// dropout keep-scale: 0 or 1/(1-p)
__device__ __forceinline__ float rng_dropout(uint64_t seed, uint32_t tick, uint32_t stream, uint32_t sig, uint32_t idx, float p) {
  return rng_uniform(seed, tick, stream, sig, idx) >= p ? 1.0f / (1.0f - p) : 0.0f;
}

}  // namespace hypad

// Tuning switches of the DEVELOPMENT library only (libhypad_hip_dev.so, -DHYPAD_DIAG=1: scripts/diag_*.py, A/B timing): the product
// library reads no environment variable -- what a C-ABI call launches depends on its arguments alone (include/hypad.h; the
// A/B switches of the training epoch are hypad_epoch_io.flags bits).
#if HYPAD_DIAG
#include <cstdlib>
#define HYPAD_TUNE_INT(name, dflt) (std::getenv(name) ? std::atoi(std::getenv(name)) : (dflt))
#else
#define HYPAD_TUNE_INT(name, dflt) (dflt)
#endif

#define HYPAD_CHECK_LAUNCH()                          \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)
