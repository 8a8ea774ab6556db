// HOST helper of the drop-in training loop: the latent draws of train.py:24,118,205 -- np.random.normal(size=(1, B, L)) on
// NumPy's global generator, one call per iteration -- for a whole epoch in one call, written as float32 straight into the
// pinned planes hypad_epoch_noise is uploaded from.  NumPy's global generator is RandomState(MT19937) with the legacy polar
// Box-Muller (numpy/random/src/legacy/legacy-distributions.c: legacy_gauss over mt19937_next_double); the caller hands over
// np.random.get_state() and puts the advanced state back with np.random.set_state(), so the process-wide stream continues
// exactly where a reference-style loop would have left it.  Same numbers bit for bit (tests/test_host_rng.py); the point is
// that ctypes releases the interpreter lock for the call (np.random.normal keeps it) and that no float64 temporaries are
// built: the epoch's 408 320 draws overlap the loader iteration and the torch.rand draws of the main thread.
//
// No device code in this file: compiled with g++ -O2 -ffp-contract=off (hypad_amd/build.py; clang's -O3 code for this loop is 1.9x slower)
// and linked into libhypad_hip.so.
#include <cmath>
#include <cstdint>
#include <vector>

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

// The stream of normals as legacy_gauss returns them: per accepted pair first f * x2, then (cached) f * x1.
// Two phases per call.  (1) ONE thread runs the generator and the rejection test for all pairs the call needs.  Candidate pairs are
// consecutive pairs of doubles of the generator -- a rejected pair consumes exactly two doubles like an accepted one -- so a block
// of NC candidates is data-parallel: 4 NC tempered words -> 2 NC doubles -> (x1, x2, r2) -> the accepted ones compacted.  A block is
// only taken whole while at least NC accepted pairs are still needed (it cannot yield more): the generator then ends exactly where
// NumPy's would; the last < NC pairs come from the pair-by-pair loop.  (2) f = sqrt(-2 log(r2) / r2), the two products and the
// float32 stores, independent per pair: a loop whose iterations the core overlaps (NumPy's loop serialises the log / divide /
// square-root latencies behind its rejection branch).  Each pair sees the same operations in the same order: same bits.
struct Gen {
  Mt g;
  static constexpr int NC = 512;                    // candidate pairs per block
  uint32_t w[4 * NC];
  double d[2 * NC];

  void words(uint32_t* out, int n) {                // the next n tempered outputs
    while (n > 0) {
      if (g.pos == MT_N) { mt_refill(g.key); g.pos = 0; }
      int take = MT_N - g.pos;
      if (take > n) take = n;
      const uint32_t* src = g.key + g.pos;
      for (int i = 0; i < take; ++i) {
        uint32_t y = src[i];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        out[i] = y;
      }
      out += take; n -= take; g.pos += take;
    }
  }
  // exactly `pairs` accepted (x1, x2, r2) triples
  void accepted(int64_t pairs, double* x1, double* x2, double* r2) {
    int64_t k = 0;
    while (pairs - k >= NC) {
      words(w, 4 * NC);
      for (int i = 0; i < 2 * NC; ++i) {
        const int32_t a = (int32_t)(w[2 * i] >> 5), b = (int32_t)(w[2 * i + 1] >> 6);
        d[i] = (a * 67108864.0 + b) / 9007199254740992.0;
      }
      for (int i = 0; i < NC; ++i) {
        const double a = 2.0 * d[2 * i] - 1.0, b = 2.0 * d[2 * i + 1] - 1.0, s2 = a * a + b * b;
        x1[k] = a; x2[k] = b; r2[k] = s2;
        k += (s2 < 1.0 && s2 != 0.0) ? 1 : 0;       // (branch-free compaction: a rejected candidate is overwritten by the next one;
      }                                             //  the arrays have NC entries of slack for the last block's overhang)
    }
    for (; k < pairs; ++k) {                        // the tail: pair by pair, never a candidate too many
      double a, b, s2;
      do {
        a = 2.0 * g.next_double() - 1.0;
        b = 2.0 * g.next_double() - 1.0;
        s2 = a * a + b * b;
      } while (s2 >= 1.0 || s2 == 0.0);
      x1[k] = a; x2[k] = b; r2[k] = s2;
    }
  }
};

struct Dest {                                       // value index -> its float: outs[(i / chunk) % n_outs][(i / (chunk * n_outs)) * chunk + i % chunk]
  float* const* outs; int n_outs; int64_t chunk;
  inline float* at(int64_t i) const {
    const int64_t c = i / chunk, pos = i - c * chunk;
    return outs[c % n_outs] + (c / n_outs) * chunk + pos;
  }
};

// pairs [p0, p1): value `first + 2 p` = f x2, `first + 2 p + 1` = f x1 (when below `total`; else it is the value left cached)
void transform_range(const double* x1, const double* x2, const double* r2, int64_t p0, int64_t p1, int64_t first, int64_t total, const Dest& dst,
                     double* cached_out) {
  for (int64_t p = p0; p < p1; ++p) {
    const double f = std::sqrt(-2.0 * std::log(r2[p]) / r2[p]);
    const double a = f * x2[p], b = f * x1[p];
    const int64_t i = first + 2 * p;
    *dst.at(i) = (float)(0.0 + 1.0 * a);            // legacy_normal(loc 0, scale 1), then float32 (torch.Tensor(float64 array))
    if (i + 1 < total) *dst.at(i + 1) = (float)(0.0 + 1.0 * b);
    else *cached_out = b;
  }
}

}  // namespace

extern "C" int hypad_host_mt19937_normal(uint32_t* key, int* pos, int* has_gauss, double* cached_gaussian, float* const* outs, int n_outs,
                                         int64_t chunk, int64_t rounds) {
  if (!key || !pos || !has_gauss || !cached_gaussian || !outs || n_outs <= 0 || chunk < 0 || rounds < 0 || *pos < 0 || *pos > MT_N)
    return HYPAD_EINVAL;
  const int64_t total = rounds * n_outs * chunk;
  if (total == 0) return HYPAD_OK;
  const Dest dst{outs, n_outs, chunk};
  int64_t first = 0;                                // value index of the first pair's first value
  if (*has_gauss) {                                 // the value a previous call (or NumPy itself) left cached comes first
    *dst.at(0) = (float)(0.0 + 1.0 * *cached_gaussian);
    *has_gauss = 0; *cached_gaussian = 0.0;
    first = 1;
  }
  const int64_t pairs = (total - first + 1) / 2;
  if (pairs == 0) return HYPAD_OK;
  static thread_local std::vector<double> buf;
  buf.resize((size_t)(3 * (pairs + Gen::NC)));
  double* x1 = buf.data(); double* x2 = x1 + pairs + Gen::NC; double* r2 = x2 + pairs + Gen::NC;
  static thread_local Gen gen;
  gen.g.key = key; gen.g.pos = *pos;
  gen.accepted(pairs, x1, x2, r2);
  *pos = gen.g.pos;
  double cached = 0.0;
  const bool odd = ((total - first) & 1) != 0;      // an odd count leaves the second value of the last pair cached, as legacy_gauss would
  // (one thread: splitting the transform over four std::threads made the call SLOWER inside a process with torch and the HIP
  // runtime loaded -- 2.6 ms against 1.7 -- creating a thread there initialises every loaded library's thread-local block)
  transform_range(x1, x2, r2, 0, pairs, first, total, dst, &cached);
  if (odd) { *has_gauss = 1; *cached_gaussian = cached; }
  return HYPAD_OK;
}
