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
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/hypad.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

// The hot loops below are plain integer / IEEE double arithmetic over arrays (no contraction: -ffp-contract=off): compiled three
// times -- baseline x86-64, AVX2, AVX-512 -- and picked at load time for the host the library runs on (it is built in another
// container than it runs in, so -march=native is not an option).  Same bits in every clone.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(HYPAD_NOCLONE)
#define HYPAD_CLONES __attribute__((target_clones("default", "avx2", "avx512f")))
#else
#define HYPAD_CLONES
#endif

HYPAD_CLONES void mt_refill(uint32_t* mt) {
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

HYPAD_CLONES void temper_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, int n) {
  for (int i = 0; i < n; ++i) {
    uint32_t y = src[i];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    out[i] = y;
  }
}
// n candidate pairs from 4 n tempered words: (a, b) = 2 * double(two words) - 1 each, s2 = a^2 + b^2
HYPAD_CLONES void candidates(const uint32_t* __restrict__ w, double* __restrict__ ca, double* __restrict__ cb, double* __restrict__ cs, int n) {
  for (int i = 0; i < n; ++i) {
    const int32_t a0 = (int32_t)(w[4 * i] >> 5), b0 = (int32_t)(w[4 * i + 1] >> 6), a1 = (int32_t)(w[4 * i + 2] >> 5), b1 = (int32_t)(w[4 * i + 3] >> 6);
    const double d0 = (a0 * 67108864.0 + b0) / 9007199254740992.0, d1 = (a1 * 67108864.0 + b1) / 9007199254740992.0;
    const double a = 2.0 * d0 - 1.0, b = 2.0 * d1 - 1.0;
    ca[i] = a; cb[i] = b; cs[i] = a * a + b * b;
  }
}
HYPAD_CLONES void words_to_unit_floats(const uint32_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    uint32_t y = src[i];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    dst[i] = (float)(y & 0xffffffu) * 5.9604644775390625e-8f;      // 2^-24: exact in float32 (24 bits)
  }
}

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
  double ca[NC], cb[NC], cs[NC];

  void words(uint32_t* out, int n) {                // the next n tempered outputs
    while (n > 0) {
      if (g.pos == MT_N) { mt_refill(g.key); g.pos = 0; }
      int take = MT_N - g.pos;
      if (take > n) take = n;
      temper_words(g.key + g.pos, out, take);
      out += take; n -= take; g.pos += take;
    }
  }
  // exactly `pairs` accepted (x1, x2, r2) triples
  void accepted(int64_t pairs, double* x1, double* x2, double* r2) {
    int64_t k = 0;
    while (pairs - k >= NC) {
      words(w, 4 * NC);
      candidates(w, ca, cb, cs, NC);
      for (int i = 0; i < NC; ++i) {
        const double s2 = cs[i];
        x1[k] = ca[i]; x2[k] = cb[i]; r2[k] = s2;
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

// The accepted pairs of a call: ONE buffer per process, grown and never shrunk, held for the call (concurrent callers take turns: the
// generator they would advance is one and the same anyway).  A thread-local buffer looked free and was not: the drop-in loop draws
// every epoch on a fresh helper thread, so every epoch mapped, zero-filled and page-faulted its 5 MB (configs[1]) / 49 MB (configs[3])
// anew -- 1 of the 2.2 ms and 14 of the 19 ms those draws took inside train_tadgan, against 1.4 / 5.2 ms in a loop on one thread.
struct PairBuffer {
  static std::mutex& mu() { static std::mutex m; return m; }
  static std::vector<double>& store() { static std::vector<double> v; return v; }
  std::unique_lock<std::mutex> lock;
  explicit PairBuffer(size_t n) : lock(mu()) { if (store().size() < n) store().resize(n); }
  double* data() { return store().data(); }
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
  PairBuffer pb((size_t)(3 * (pairs + Gen::NC)));
  double* x1 = pb.data(); double* x2 = x1 + pairs + Gen::NC; double* r2 = x2 + pairs + Gen::NC;
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

// hypad_host_mt19937_normal for draws large enough to be worth several threads (configs[3]: 4.5 M values per epoch, 33 ms in one
// thread on the GPU box's host): the calling thread runs phase 1 -- the generator and the rejection test are sequential -- in blocks
// of 32 768 accepted pairs and publishes how far it is; `threads` helper threads take the blocks' transforms (independent per pair)
// as they become ready.  Same values, same final generator state.  (Starting a thread costs ~0.3 ms in a process with torch and the
// HIP runtime loaded: worth it from about a million values on; the caller decides.)
extern "C" int hypad_host_mt19937_normal_mt(uint32_t* key, int* pos, int* has_gauss, double* cached_gaussian, float* const* outs, int n_outs,
                                            int64_t chunk, int64_t rounds, int threads) {
  if (threads < 1) return hypad_host_mt19937_normal(key, pos, has_gauss, cached_gaussian, outs, n_outs, chunk, rounds);
  if (!key || !pos || !has_gauss || !cached_gaussian || !outs || n_outs <= 0 || chunk < 0 || rounds < 0 || *pos < 0 || *pos > MT_N)
    return HYPAD_EINVAL;
  const int64_t total = rounds * n_outs * chunk;
  if (total == 0) return HYPAD_OK;
  const Dest dst{outs, n_outs, chunk};
  int64_t first = 0;
  if (*has_gauss) {
    *dst.at(0) = (float)(0.0 + 1.0 * *cached_gaussian);
    *has_gauss = 0; *cached_gaussian = 0.0;
    first = 1;
  }
  const int64_t pairs = (total - first + 1) / 2;
  if (pairs == 0) return HYPAD_OK;
  constexpr int64_t BLK = 32768;
  PairBuffer pb((size_t)(3 * (pairs + Gen::NC)));
  double* x1 = pb.data(); double* x2 = x1 + pairs + Gen::NC; double* r2 = x2 + pairs + Gen::NC;
  std::atomic<int64_t> produced{0}, next_block{0};
  double cached = 0.0;
  const int64_t nblocks = (pairs + BLK - 1) / BLK;
  auto worker = [&]() {
    for (;;) {
      const int64_t b = next_block.fetch_add(1, std::memory_order_relaxed);
      if (b >= nblocks) return;
      const int64_t p0 = b * BLK, p1 = p0 + BLK < pairs ? p0 + BLK : pairs;
      // (a helper that is ahead of the generator sleeps: three helpers spinning on sched_yield slowed the whole call down to the
      // single-thread time on the 256-core host -- 15.7 ms against 11.8 with ONE helper, which never waits)
      while (produced.load(std::memory_order_acquire) < p1) std::this_thread::sleep_for(std::chrono::microseconds(20));
      transform_range(x1, x2, r2, p0, p1, first, total, dst, &cached);      // (only the block that holds the last pair writes `cached`)
    }
  };
  std::vector<std::thread> pool;
  pool.reserve((size_t)threads);
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
#if defined(__linux__)
  {
    // keep the helpers next to the generator: on the GPU box's two-socket host the scheduler spread them over both sockets and three
    // helpers were SLOWER than one (15.7 against 11.9 ms for 4.1 M values: every accepted pair crossing the socket link).  Held on the
    // eight cores around the caller's (one core complex on that host: consecutive core numbers) 5.2 ms; on the caller's socket 8.3.
    // A mask the kernel refuses is simply not applied.
    // Only the CPUs the caller may run on count (a cpuset / taskset that overlaps the eight partly would otherwise squeeze the helpers
    // onto one or two cores while the generator keeps producing): with fewer than threads + 1 of them left, no pinning at all.
    const int cpu = sched_getcpu();
    cpu_set_t allowed;
    if (cpu >= 0 && sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
      cpu_set_t set;
      CPU_ZERO(&set);
      const int base = cpu & ~7;
      int n_set = 0;
      for (int c = base; c < base + 8 && c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &set); ++n_set; }
      if (n_set >= threads + 1)
        for (auto& t : pool) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
    }
  }
#endif
  static thread_local Gen gen;
  gen.g.key = key; gen.g.pos = *pos;
  for (int64_t p0 = 0; p0 < pairs; p0 += BLK) {
    const int64_t n = p0 + BLK < pairs ? BLK : pairs - p0;
    // (a block's compaction may overhang its end by < NC entries: they are the next block's first entries, rewritten by it -- and
    // never read before `produced` passes them)
    gen.accepted(n, x1 + p0, x2 + p0, r2 + p0);
    produced.store(p0 + n, std::memory_order_release);
  }
  *pos = gen.g.pos;
  for (auto& t : pool) t.join();
  if ((total - first) & 1) { *has_gauss = 1; *cached_gaussian = cached; }
  return HYPAD_OK;
}

// The interpolation weights of train.py:64,149 -- torch.rand on torch's default CPU generator: at::mt19937 (the same recurrence and
// tempering as above), one 32-bit output per float32 element, value = (word & (2^24 - 1)) * 2^-24 (ATen/core/TransformationHelper.h
// uniform_real; the CPU kernel is serial: element i takes word i).  `state`: the 5 056 bytes of torch.get_rng_state() -- at::
// CPUGeneratorImplState = {uint64 seed; int left; int seeded; uint64 next; uint64 state[624]; double normal_x, normal_y, normal_rho;
// int normal_is_valid; float next_float_normal_sample; bool valid} (checked against the live generator by tests/test_host_rng.py) --
// advanced in place exactly as n calls of the engine would leave it (left + next == 625 after every call).  torch.rand generates
// 0.57 floats per ns on the GPU box's host; this loop ~2: an epoch of configs[3] draws 17.4 M of them.
extern "C" int hypad_host_torch_mt19937_uniform(void* state, size_t state_bytes, float* out, int64_t n) {
  if (!state || state_bytes != 5056 || n < 0 || (n > 0 && !out)) return HYPAD_EINVAL;
  unsigned char* sb = (unsigned char*)state;
  int32_t left, seeded; uint64_t next;
  std::memcpy(&left, sb + 8, 4); std::memcpy(&seeded, sb + 12, 4); std::memcpy(&next, sb + 16, 8);
  if (!seeded || left < 1 || left > MT_N || next > (uint64_t)MT_N) return HYPAD_EINVAL;
  if (n == 0) return HYPAD_OK;
  uint64_t* st64 = (uint64_t*)(sb + 24);                // (8-byte aligned inside a torch tensor's storage)
  uint32_t key[MT_N];
  for (int i = 0; i < MT_N; ++i) key[i] = (uint32_t)st64[i];
  int pos = MT_N + 1 - left;                            // left == 1: the block is used up (also the freshly seeded state, next == 0)
  int64_t done = 0;
  while (done < n) {
    if (pos == MT_N) { mt_refill(key); pos = 0; }
    int64_t take = MT_N - pos;
    if (take > n - done) take = n - done;
    words_to_unit_floats(key + pos, out + done, take);
    done += take; pos += (int)take;
  }
  for (int i = 0; i < MT_N; ++i) st64[i] = key[i];
  left = MT_N + 1 - pos; next = (uint64_t)pos;
  std::memcpy(sb + 8, &left, 4); std::memcpy(sb + 16, &next, 8);
  return HYPAD_OK;
}

