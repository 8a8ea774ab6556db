// HOST helper of the drop-in training loop: the latent draws of train.py:24,118,205 -- np.random.normal(size=(1, B, L)) on
// NumPy's global generator, one call per iteration -- for a whole epoch in one call, written as float32 straight into the
// pinned planes hypad_epoch_noise is uploaded from.  NumPy's global generator is RandomState(MT19937) with the legacy polar
// Box-Muller (numpy/random/src/legacy/legacy-distributions.c: legacy_gauss over mt19937_next_double); the caller hands over
// np.random.get_state() and puts the advanced state back with np.random.set_state(), so the process-wide stream continues
// exactly where a reference-style loop would have left it.  Same numbers bit for bit (tests/test_host_rng.py); the point is
// that ctypes releases the interpreter lock for the call (np.random.normal keeps it) and that no float64 temporaries are
// built: the epoch's 408 320 draws overlap the loader iteration and the torch.rand draws of the main thread.
// No device code in this file: compiled with g++ -O2 -ffp-contract=off (hypad_amd/build.py; clang's -O3 code for this loop is 1.9x slower)
// and linked into libhypad_hip.so.
#include <cmath>
#include <cstdint>

#include "../../include/hypad.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

inline void mt_refill(uint32_t* mt) {
  constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  int kk = 0;
  uint32_t y;
  for (; kk < MT_N - MT_M; ++kk) {
    y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
    mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  for (; kk < MT_N - 1; ++kk) {
    y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
    mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  y = (mt[MT_N - 1] & UPPER) | (mt[0] & LOWER);
  mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
}

struct Mt {
  uint32_t* key;
  int pos;
  inline uint32_t next() {
    if (pos == MT_N) { mt_refill(key); pos = 0; }
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  // mt19937_next_double: 53 random bits from two words
  inline double next_double() {
    const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
};

}  // namespace

extern "C" int hypad_host_mt19937_normal(uint32_t* key, int* pos, int* has_gauss, double* cached_gaussian, float* const* outs, int n_outs,
                                         int64_t chunk, int64_t rounds) {
  if (!key || !pos || !has_gauss || !cached_gaussian || !outs || n_outs <= 0 || chunk < 0 || rounds < 0 || *pos < 0 || *pos > MT_N)
    return HYPAD_EINVAL;
  Mt g{key, *pos};
  int have = *has_gauss;
  double spare = *cached_gaussian;
  for (int64_t r = 0; r < rounds; ++r)
    for (int k = 0; k < n_outs; ++k) {
      float* out = outs[k] + r * chunk;
      for (int64_t i = 0; i < chunk; ++i) {
        double v;
        if (have) {
          v = spare; have = 0; spare = 0.0;
        } else {
          double x1, x2, r2;
          do {
            x1 = 2.0 * g.next_double() - 1.0;
            x2 = 2.0 * g.next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
          } while (r2 >= 1.0 || r2 == 0.0);
          const double f = std::sqrt(-2.0 * std::log(r2) / r2);
          spare = f * x1; have = 1;
          v = f * x2;
        }
        out[i] = (float)(0.0 + 1.0 * v);      // legacy_normal(loc = 0, scale = 1), then torch.Tensor(float64 array): round to nearest
      }
    }
  *pos = g.pos; *has_gauss = have; *cached_gaussian = spare;
  return HYPAD_OK;
}
