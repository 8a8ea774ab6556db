// Window-scoring kernels: utils/anomaly_detection_utils.py on the GPU.
// Post-processing arithmetic is fp64 because the reference computes it in NumPy/pandas fp64.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/hypad.h"
#include "device_utils.h"

using namespace hypad;

namespace {

constexpr int THREADS = 256;
constexpr int MAX_WINDOW = 256;

inline int grid_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- S1: anti-diagonal un-roll (:918-935).  One wave per output timestep: gather <= window values, rank them
// by counting (ties broken by position), scatter into sorted order in LDS, read the order statistics.
__device__ __forceinline__ float np_lerp(float a, float b, float t) {
  // numpy.lib._function_base_impl._lerp, evaluated in the data's precision (float32) as NumPy does
  float diff = b - a;
  float r = a + diff * t;
  if (t >= 0.5f) r = b - diff * (1.0f - t);
  return r;
}
// EPL: anti-diagonal values per lane (window <= 64 EPL); the rank-by-counting loop is the kernel's cost (window^2 / 64
// compares per timestep and lane), so it is instantiated for the window class and reads the broadcast values four at a time.
template <int EPL>
__global__ __launch_bounds__(THREADS) void unroll_median_kernel(const float* __restrict__ y_hat, float* __restrict__ median,
                                                                 double* __restrict__ summary, int64_t n, int W) {
  __shared__ __attribute__((aligned(16))) float vals[THREADS / 64][MAX_WINDOW];
  __shared__ float sorted[THREADS / 64][MAX_WINDOW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t T = n + W - 1;
  float* v = vals[wave];
  float* s = sorted[wave];
  for (int64_t t = (int64_t)blockIdx.x * (THREADS / 64) + wave; t < T; t += (int64_t)gridDim.x * (THREADS / 64)) {
    const int j0 = (int)(t - n + 1 > 0 ? t - n + 1 : 0);
    const int j1 = (int)(t + 1 < W ? t + 1 : W);
    const int cnt = j1 - j0;
    float mine[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      int i = lane + 64 * e;
      mine[e] = 0.f;
      if (i < cnt) {
        int j = j0 + i;
        mine[e] = y_hat[(t - j) * W + j];
        v[i] = mine[e];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave landed
    // pad the slab to a multiple of 4 with +inf (never below or equal to a finite value)
    if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) v[cnt + lane] = __int_as_float(0x7f800000);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // rank = #{k : v[k] < mine} + #{k < i : v[k] == mine}.  Fast pass: count "less" only (one compare + add-carry per value).
    // Without ties those counts are a permutation of 0 .. cnt-1, with ties two values share a count and the counts' sum falls
    // short of cnt (cnt - 1) / 2: only then is the ordered tie count needed.  (The sum is exact in fp32: < 2^15 at window 256.)
    int rank[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) rank[e] = 0;
    for (int k0 = 0; k0 < cnt; k0 += 4) {
      const float4 q = *reinterpret_cast<const float4*>(v + k0);
      const float vk[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < EPL; ++e) rank[e] += vk[u] < mine[e] ? 1 : 0;
    }
    float rsum = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) rsum += lane + 64 * e < cnt ? (float)rank[e] : 0.f;
    const bool ties = hypad::wave_sum(rsum) != 0.5f * (float)cnt * (float)(cnt - 1);
    if (ties) {                                      // wave-uniform
#pragma unroll
      for (int e = 0; e < EPL; ++e) rank[e] = 0;
      for (int k = 0; k < cnt; ++k) {
        const float vk = v[k];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int i = lane + 64 * e;
          rank[e] += (vk < mine[e]) || (vk == mine[e] && k < i);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
      if (lane + 64 * e < cnt) s[rank[e]] = mine[e];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) {
      const float lo = s[(cnt - 1) >> 1], hi = s[cnt >> 1];
      median[t] = (cnt & 1) ? lo : (lo + hi) * 0.5f;     // np.median of float32 stays float32
      if (summary) {
        double* o = summary + t * 5;
        o[0] = (double)s[0];
        const double qs[3] = {0.25, 0.5, 0.75};
        for (int qi = 0; qi < 3; ++qi) {
          double pos = qs[qi] * (double)(cnt - 1);
          int a = (int)floor(pos);
          int b = a + 1 < cnt ? a + 1 : cnt - 1;
          o[1 + qi] = (double)np_lerp(s[a], s[b], (float)(pos - (double)a));
        }
        o[4] = (double)s[cnt - 1];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(THREADS) void unroll_true_kernel(const double* __restrict__ y, double* __restrict__ out, int64_t n, int W) {
  const int64_t T = n + W - 1;
  for (int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x; t < T; t += (int64_t)gridDim.x * THREADS)
    out[t] = t < n ? y[t * W] : y[(n - 1) * W + (t - n + 1)];
}

__global__ __launch_bounds__(THREADS) void point_error_kernel(const double* __restrict__ y, const float* __restrict__ yh,
                                                               double* __restrict__ out, int64_t T) {
  for (int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x; t < T; t += (int64_t)gridDim.x * THREADS)
    out[t] = fabs(y[t] - (double)yh[t]);
}

// pandas centred window of size w at i: [i + off - w + 1, i + off], off = (w - 1) / 2, clipped to the array
__device__ __forceinline__ void centred_window(int64_t i, int w, int64_t T, int64_t& lo, int64_t& hi) {
  const int off = (w - 1) / 2;
  lo = i + off - w + 1; hi = i + off;
  if (lo < 0) lo = 0;
  if (hi > T - 1) hi = T - 1;
}

__global__ __launch_bounds__(THREADS) void area_error_kernel(const double* __restrict__ y, const float* __restrict__ yh,
                                                              double* __restrict__ out, int64_t T, int w) {
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < T; i += (int64_t)gridDim.x * THREADS) {
    int64_t lo, hi;
    centred_window(i, w, T, lo, hi);
    if (hi - lo + 1 < w / 2) { out[i] = NAN; continue; }
    double a = 0.0, b = 0.0;
    for (int64_t k = lo; k < hi; ++k) {
      a += (y[k] + y[k + 1]) * 0.5;
      b += ((double)yh[k] + (double)yh[k + 1]) * 0.5;
    }
    out[i] = fabs(a - b);
  }
}

template <int LEN>
__global__ __launch_bounds__(THREADS) void dtw_error_kernel(const double* __restrict__ y, const float* __restrict__ yh,
                                                             double* __restrict__ out, int64_t T) {
  constexpr int HALF = LEN / 2;
  for (int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x; p < T; p += (int64_t)gridDim.x * THREADS) {
    const int64_t i = p - HALF;                 // window start in padded coordinates
    if (i < 0 || i >= T - LEN) { out[p] = 0.0; continue; }
    double a[LEN], b[LEN], row[LEN];
#pragma unroll
    for (int k = 0; k < LEN; ++k) {
      int64_t src = i + k - HALF;               // y_pad[i + k] = y[i + k - HALF]
      bool ok = src >= 0 && src < T;
      a[k] = ok ? y[src] : 0.0;
      b[k] = ok ? (double)yh[src] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < LEN; ++j) { double d = a[0] - b[j]; acc += d * d; row[j] = acc; }
#pragma unroll
    for (int r = 1; r < LEN; ++r) {
      double diag = row[0];
      double d0 = a[r] - b[0];
      row[0] = row[0] + d0 * d0;
#pragma unroll
      for (int j = 1; j < LEN; ++j) {
        double up = row[j];
        double d = a[r] - b[j];
        double m = fmin(fmin(up, row[j - 1]), diag);
        row[j] = d * d + m;
        diag = up;
      }
    }
    out[p] = sqrt(row[LEN - 1]);
  }
}

__global__ __launch_bounds__(THREADS) void rolling_mean_kernel(const double* __restrict__ in, double* __restrict__ out, int64_t T, int w) {
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < T; i += (int64_t)gridDim.x * THREADS) {
    int64_t lo, hi;
    centred_window(i, w, T, lo, hi);
    int cnt = 0;
    double s = 0.0;
    for (int64_t k = lo; k <= hi; ++k) {
      double v = in[k];
      if (v == v) { s += v; ++cnt; }            // pandas skips NaN
    }
    out[i] = cnt >= w / 2 && cnt > 0 ? s / (double)cnt : NAN;
  }
}

// stats[0] = mean, stats[1] = population std (scipy.stats.zscore, ddof = 0)
__global__ __launch_bounds__(1024) void zscore_stats_kernel(const double* __restrict__ in, double* __restrict__ stats, int64_t T) {
  __shared__ double part[16];
  __shared__ double mean_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < T; i += 1024) s += in[i];
  s = wave_sum(s);
  if (lane == 0) part[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += part[k];
    mean_s = t / (double)T;
  }
  __syncthreads();
  const double mean = mean_s;
  double q = 0.0;
  for (int64_t i = threadIdx.x; i < T; i += 1024) { double d = in[i] - mean; q += d * d; }
  q = wave_sum(q);
  __syncthreads();
  if (lane == 0) part[wave] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += part[k];
    stats[0] = mean;
    stats[1] = sqrt(t / (double)T);
  }
}
__global__ __launch_bounds__(THREADS) void zscore_apply_kernel(const double* __restrict__ in, const double* __restrict__ stats,
                                                                double* __restrict__ out, int64_t T) {
  const double mean = stats[0], sd = stats[1];
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < T; i += (int64_t)gridDim.x * THREADS) {
    double z = (in[i] - mean) / sd;
    out[i] = (z != z) ? z : fmax(z, 0.0) + 1.0;  // np.clip keeps NaN
  }
}

// ---- critic smoothing: final_critic_scores :365-404.  Timestep t sees the critic values of the
// windows covering it, critic[t - j] for the valid j (each window's score repeated along the window, un-rolled along
// anti-diagonals).  Its score is the sample at which a Scott-bandwidth Gaussian KDE of those values is largest
// (scipy.stats.gaussian_kde(v)(v), first arg-max), the median when fewer than two values or a singular covariance.
constexpr int KDE_CB = 4;                    // candidates per pass-2 batch
//
// Selection in two passes.  The result is a SAMPLE (the arg-max's value), so only the arg-max must be exact, not the densities:
// pass 1 evaluates every density in fp32 (v_exp_f32: cnt^2 = 10^4 exponentials per timestep at window 100 -- in fp64 this pass
// was 97 % of the kernel, 1.7 ms for 125 000 windows); pass 2 re-evaluates in fp64, exactly as before, only the samples whose
// fp32 density lies within 1e-3 of the fp32 maximum -- a superset of the true arg-max set (the fp32 density's relative error is
// below 3e-5: arguments carry <= 4 ulp, |argument| < 88 wherever the term is not 0, v_exp_f32 adds 2 ulp) -- with the same
// first-maximum tie rule.  Clustered samples (many near-equal densities) simply put more candidates into pass 2.
__global__ __launch_bounds__(THREADS) void kde_mode_kernel(const float* __restrict__ critic, double* __restrict__ modes,
                                                            int64_t n, int W) {
  __shared__ double vals[THREADS / 64][MAX_WINDOW];
  __shared__ __attribute__((aligned(16))) float vals32[THREADS / 64][MAX_WINDOW + 4];
  __shared__ double terms[THREADS / 64][KDE_CB * MAX_WINDOW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t T = n + W - 1;
  double* v = vals[wave];
  float* vf = vals32[wave];
  constexpr int KPL = MAX_WINDOW / 64;                     // samples per lane
  const double scottW = pow((double)W, -0.4);
  for (int64_t t = (int64_t)blockIdx.x * (THREADS / 64) + wave; t < T; t += (int64_t)gridDim.x * (THREADS / 64)) {
    const int j0 = (int)(t - n + 1 > 0 ? t - n + 1 : 0);
    const int j1 = (int)(t + 1 < W ? t + 1 : W);
    const int cnt = j1 - j0;
    double s = 0.0;
    for (int k = lane; k < cnt; k += 64) {
      const float xf = critic[t - (j0 + k)];
      v[k] = (double)xf;
      vf[k] = xf;
      s += (double)xf;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const double mean = wave_sum(s) / (double)cnt;
    double q = 0.0;
    for (int k = lane; k < cnt; k += 64) { const double d = v[k] - mean; q += d * d; }
    const double var = cnt > 1 ? wave_sum(q) / (double)(cnt - 1) : 0.0;          // np.cov: ddof = 1
    // Scott: factor = n^(-1/5), squared.  (All but the 2 (W - 1) edge timesteps have cnt == W: that power is taken once per
    // thread, not once per timestep -- a double-precision pow is ~200 instructions.)
    const double cov = var * (cnt == W ? scottW : pow((double)cnt, -0.4));
    double out;
    if (cnt > 1 && cov > 0.0 && cov == cov) {
      const double inv = 0.5 / cov;
      // pass 1: fp32 densities of this lane's samples.  exp(-d^2 inv) = exp2(-(c d)^2) with c = sqrt(inv log2 e): the samples are
      // centred and rescaled once (pass 2 reads the fp64 copies), so a pair costs a subtract, a multiply, an exp2 and an add; the
      // slab is padded with +inf to a multiple of four (a padded pair contributes exp2(-inf) = 0) and read four values at a
      // time, every value once for all of the lane's samples.
      // The samples are CENTRED first, in fp64 (densities depend on differences only): rescaling the raw values would leave the
      // fp32 copies with an absolute error of |value| 2^-24 c, which at |mean| / bandwidth beyond ~1e4 exceeds the screen's margin.
      const double c64 = sqrt(inv * 1.44269504088896341);
      for (int k = lane; k < cnt; k += 64) vf[k] = (float)((v[k] - mean) * c64);
      if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) vf[cnt + lane] = __int_as_float(0x7f800000);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      float d32[KPL], xs[KPL];
#pragma unroll
      for (int u = 0; u < KPL; ++u) { const int k = lane + 64 * u; xs[u] = vf[k < cnt ? k : 0]; d32[u] = 0.f; }
      const int nu = (cnt + 63) >> 6;                                             // sample slots in use (wave-uniform)
      for (int m = 0; m < cnt; m += 4) {
        const float4 q4 = *reinterpret_cast<const float4*>(vf + m);
        const float vm[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int u = 0; u < KPL; ++u) {
          if (u >= nu) continue;
#pragma unroll
          for (int c = 0; c < 4; ++c) { const float d = xs[u] - vm[c]; d32[u] += __builtin_amdgcn_exp2f(-(d * d)); }
        }
      }
      float mx = -1.f;
#pragma unroll
      for (int u = 0; u < KPL; ++u) {
        if (lane + 64 * u >= cnt) d32[u] = -1.f;
        mx = fmaxf(mx, d32[u]);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, WAVE));
      // (error of an fp32 density: the exponent's argument carries ~4e-6 absolute at the terms that matter, the 100-term sum
      // ~6e-6 relative: 1.5e-5 in all.  Everything within 2e-4 of the fp32 maximum goes to pass 2.)
      const float thr = mx * (1.f - 2e-4f);
      // pass 2: fp64 densities of the candidates, in ascending sample order (the first maximum is kept); of every sample if
      // pass 1 produced no candidate (a bandwidth so small that its reciprocal leaves the fp32 range makes the screen NaN).
      // A wave pays for a sequential sum as if all 64 lanes ran it, so a candidate's sum is NOT given to one lane with its
      // exponentials: the lanes compute a candidate's cnt exponentials side by side into LDS (two per lane at window 100), four
      // candidates per batch, then lane c adds candidate c's terms in index order -- the same additions in the same order as
      // the one-lane loop, hence the same bits, at 1/20 of its cycles.
      double best = -1.0;
      int besti = 0x7fffffff;
      double* tm = terms[wave];
      {
        // one candidate only: the screen has decided (its margin is far above the fp32 pass's error), no fp64 sum is needed
        int ncand = 0, first = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < KPL; ++u) {
          const unsigned long long mk = __ballot(lane + 64 * u < cnt && d32[u] >= thr);
          ncand += __builtin_popcountll(mk);
          if (mk && first == 0x7fffffff) first = __builtin_ctzll(mk) + 64 * u;
        }
        if (ncand == 1) besti = first;
      }
      for (int round = 0; round < 2 && besti == 0x7fffffff; ++round) {
#pragma unroll
        for (int u = 0; u < KPL; ++u) {
          const int ku = lane + 64 * u;
          unsigned long long mask = __ballot(ku < cnt && (round == 1 || d32[u] >= thr));
          while (mask) {                                                          // wave-uniform
            int kc[KDE_CB];
            int nb = 0;
#pragma unroll
            for (int c = 0; c < KDE_CB; ++c) {
              kc[c] = -1;
              if (mask) { kc[c] = __builtin_ctzll(mask) + 64 * u; mask &= mask - 1; nb = c + 1; }
            }
#pragma unroll
            for (int c = 0; c < KDE_CB; ++c) {
              if (kc[c] < 0) continue;
              const double xk = v[kc[c]];
              for (int m = lane; m < cnt; m += 64) { const double d = xk - v[m]; tm[c * MAX_WINDOW + m] = exp(-d * d * inv); }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            double dens = -1.0;
            if (lane < nb) {
              dens = 0.0;
              const double* tp = tm + lane * MAX_WINDOW;
              int m = 0;
              for (; m + 8 <= cnt; m += 8) {                 // (the terms of eight steps requested together; added in index order)
                double t8[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) t8[x] = tp[m + x];
#pragma unroll
                for (int x = 0; x < 8; ++x) dens += t8[x];
              }
              for (; m < cnt; ++m) dens += tp[m];
            }
#pragma unroll
            for (int c = 0; c < KDE_CB; ++c) {
              const double dc = __shfl(dens, c, WAVE);
              if (c < nb && dc > best) { best = dc; besti = kc[c]; }
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
      out = v[besti];
    } else {
      // median by rank counting (cnt <= 256)
      double lo = 0.0, hi = 0.0;
      for (int k = lane; k < cnt; k += 64) {
        const double xk = v[k];
        int rank = 0;
        for (int m = 0; m < cnt; ++m) rank += (v[m] < xk) || (v[m] == xk && m < k);
        if (rank == (cnt - 1) / 2) lo = xk;
        if (rank == cnt / 2) hi = xk;
      }
      out = 0.5 * (wave_sum(lo) + wave_sum(hi));
    }
    if (lane == 0) modes[t] = out;
    __builtin_amdgcn_wave_barrier();
  }
}

// _compute_critic_score :307-333 without the rolling mean: stats[0] = mean of the values inside [lo, hi],
// stats[1] = population std of all values; out = |x - mean| / std + 1.
__global__ __launch_bounds__(1024) void critic_stats_kernel(const double* __restrict__ in, double lo, double hi,
                                                             double* __restrict__ stats, int64_t T) {
  __shared__ double part[3][16];
  __shared__ double mean_all;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0, sr = 0.0, cr = 0.0;
  for (int64_t i = threadIdx.x; i < T; i += 1024) {
    const double x = in[i];
    s += x;
    if (x >= lo && x <= hi) { sr += x; cr += 1.0; }
  }
  s = wave_sum(s); sr = wave_sum(sr); cr = wave_sum(cr);
  if (lane == 0) { part[0][wave] = s; part[1][wave] = sr; part[2][wave] = cr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = 0; k < 16; ++k) { a += part[0][k]; b += part[1][k]; c += part[2][k]; }
    mean_all = a / (double)T;
    stats[0] = b / c;
  }
  __syncthreads();
  const double m = mean_all;
  double q = 0.0;
  for (int64_t i = threadIdx.x; i < T; i += 1024) { const double d = in[i] - m; q += d * d; }
  q = wave_sum(q);
  __syncthreads();
  if (lane == 0) part[0][wave] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int k = 0; k < 16; ++k) a += part[0][k];
    stats[1] = sqrt(a / (double)T);
  }
}
__global__ __launch_bounds__(THREADS) void critic_apply_kernel(const double* __restrict__ in, const double* __restrict__ stats,
                                                                double* __restrict__ out, int64_t T) {
  const double mean = stats[0], sd = stats[1];
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < T; i += (int64_t)gridDim.x * THREADS)
    out[i] = fabs((in[i] - mean) / sd) + 1.0;
}

__global__ __launch_bounds__(THREADS) void row_norms_kernel(const float* __restrict__ x, double* __restrict__ out, int64_t rows, int dim) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * (THREADS / 64)) {
    float s = 0.f;    // np.linalg.norm on float32 reduces in float32
    for (int c = lane; c < dim; c += 64) { float v = x[r * dim + c]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) out[r] = (double)sqrtf(s);
  }
}

__global__ __launch_bounds__(THREADS) void combine_kernel(int mode, const double* __restrict__ c, const double* __restrict__ r,
                                                           const double* __restrict__ u, double* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
    double cv = c ? c[i] : 0.0, rv = r ? r[i] : 0.0, uv = u ? u[i] : 0.0, o;
    switch (mode) {
      case HYPAD_COMB_SUM: o = 0.2 * cv + 0.8 * rv; break;
      case HYPAD_COMB_MULT: o = cv * rv; break;
      case HYPAD_COMB_UNCERTAINTY: o = cv * rv * uv; break;
      case HYPAD_COMB_CRITIC: o = cv; break;
      case HYPAD_COMB_CRITIC_UNCERTAINTY: o = cv * uv; break;
      case HYPAD_COMB_SUM_UNCERTAINTY: o = 0.5 * cv * uv + 0.5 * rv * uv; break;
      case HYPAD_COMB_REC: o = rv; break;
      case HYPAD_COMB_REC_UNCERTAINTY: o = rv * uv; break;
      case HYPAD_COMB_EUCL_MULT: o = cv * rv; break;
      default: o = 0.5 * (cv - 1.0) + 0.5 * (rv - 1.0); break;   // HYPAD_COMB_EUCL_SUM, lambda_rec = 0.5
    }
    out[i] = o;
  }
}

}  // namespace

extern "C" {

int hypad_unroll_median(const float* y_hat, float* median, double* summary, int64_t n, int window, hypad_stream_t s) {
  if (!y_hat || !median || n <= 0 || window <= 0) return HYPAD_EINVAL;
  if (window > MAX_WINDOW) return HYPAD_EUNSUPPORTED;
  const int64_t T = n + window - 1;
  if (window <= 64) hipLaunchKernelGGL(unroll_median_kernel<1>, dim3(grid_for(T, THREADS / 64)), dim3(THREADS), 0, (hipStream_t)s, y_hat, median, summary, n, window);
  else if (window <= 128) hipLaunchKernelGGL(unroll_median_kernel<2>, dim3(grid_for(T, THREADS / 64)), dim3(THREADS), 0, (hipStream_t)s, y_hat, median, summary, n, window);
  else hipLaunchKernelGGL(unroll_median_kernel<4>, dim3(grid_for(T, THREADS / 64)), dim3(THREADS), 0, (hipStream_t)s, y_hat, median, summary, n, window);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_unroll_true(const double* y, double* out, int64_t n, int window, hypad_stream_t s) {
  if (!y || !out || n <= 0 || window <= 0) return HYPAD_EINVAL;
  hipLaunchKernelGGL(unroll_true_kernel, dim3(grid_for(n + window - 1, THREADS)), dim3(THREADS), 0, (hipStream_t)s, y, out, n, window);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_point_error(const double* y, const float* yh, double* out, int64_t t, hypad_stream_t s) {
  if (!y || !yh || !out || t < 0) return HYPAD_EINVAL;
  if (t == 0) return HYPAD_OK;
  hipLaunchKernelGGL(point_error_kernel, dim3(grid_for(t, THREADS)), dim3(THREADS), 0, (hipStream_t)s, y, yh, out, t);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_area_error(const double* y, const float* yh, double* out, int64_t t, int score_window, hypad_stream_t s) {
  if (!y || !yh || !out || t < 0 || score_window < 2) return HYPAD_EINVAL;
  if (t == 0) return HYPAD_OK;
  hipLaunchKernelGGL(area_error_kernel, dim3(grid_for(t, THREADS)), dim3(THREADS), 0, (hipStream_t)s, y, yh, out, t, score_window);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_dtw_error(const double* y, const float* yh, double* out, int64_t t, int score_window, hypad_stream_t s) {
  if (!y || !yh || !out || t < 0 || score_window < 2) return HYPAD_EINVAL;
  if (t == 0) return HYPAD_OK;
  const int len = (score_window / 2) * 2 + 1;
  dim3 g(grid_for(t, THREADS)), b(THREADS);
  switch (len) {
    case 3: hipLaunchKernelGGL(dtw_error_kernel<3>, g, b, 0, (hipStream_t)s, y, yh, out, t); break;
    case 5: hipLaunchKernelGGL(dtw_error_kernel<5>, g, b, 0, (hipStream_t)s, y, yh, out, t); break;
    case 7: hipLaunchKernelGGL(dtw_error_kernel<7>, g, b, 0, (hipStream_t)s, y, yh, out, t); break;
    case 9: hipLaunchKernelGGL(dtw_error_kernel<9>, g, b, 0, (hipStream_t)s, y, yh, out, t); break;
    case 11: hipLaunchKernelGGL(dtw_error_kernel<11>, g, b, 0, (hipStream_t)s, y, yh, out, t); break;   // reference default
    case 21: hipLaunchKernelGGL(dtw_error_kernel<21>, g, b, 0, (hipStream_t)s, y, yh, out, t); break;
    default: return HYPAD_EUNSUPPORTED;
  }
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_rolling_mean(const double* in, double* out, int64_t t, int window, hypad_stream_t s) {
  if (!in || !out || t < 0 || window < 1) return HYPAD_EINVAL;
  if (t == 0) return HYPAD_OK;
  hipLaunchKernelGGL(rolling_mean_kernel, dim3(grid_for(t, THREADS)), dim3(THREADS), 0, (hipStream_t)s, in, out, t, window);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_zscore_clip(const double* in, double* out, int64_t t, void* workspace, size_t workspace_bytes, hypad_stream_t s) {
  if (!in || !out || t <= 0) return HYPAD_EINVAL;
  if (!workspace || workspace_bytes < 4 * sizeof(double)) return HYPAD_EWORKSPACE;
  hipLaunchKernelGGL(zscore_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, in, (double*)workspace, t);
  HYPAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(zscore_apply_kernel, dim3(grid_for(t, THREADS)), dim3(THREADS), 0, (hipStream_t)s, in, (const double*)workspace, out, t);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_kde_mode(const float* critic, double* modes, int64_t n, int window, hypad_stream_t s) {
  if (!critic || !modes || n <= 0 || window <= 0) return HYPAD_EINVAL;
  if (window > MAX_WINDOW) return HYPAD_EUNSUPPORTED;
  hipLaunchKernelGGL(kde_mode_kernel, dim3(grid_for(n + window - 1, THREADS / 64)), dim3(THREADS), 0, (hipStream_t)s, critic, modes, n, window);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_critic_zscore(const double* in, double q25, double q75, double* out, int64_t t, void* workspace, size_t workspace_bytes,
                        hypad_stream_t s) {
  if (!in || !out || t <= 0) return HYPAD_EINVAL;
  if (!workspace || workspace_bytes < 4 * sizeof(double)) return HYPAD_EWORKSPACE;
  hipLaunchKernelGGL(critic_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, in, q25, q75, (double*)workspace, t);
  HYPAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(critic_apply_kernel, dim3(grid_for(t, THREADS)), dim3(THREADS), 0, (hipStream_t)s, in, (const double*)workspace, out, t);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_row_norms(const float* x, double* out, int64_t rows, int dim, hypad_stream_t s) {
  if (!x || !out || rows < 0 || dim <= 0) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  hipLaunchKernelGGL(row_norms_kernel, dim3(grid_for(rows, THREADS / 64)), dim3(THREADS), 0, (hipStream_t)s, x, out, rows, dim);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_combine_scores(int mode, const double* c, const double* r, const double* u, double* out, int64_t n, hypad_stream_t s) {
  if (!out || n < 0 || mode < 0 || mode > HYPAD_COMB_EUCL_SUM) return HYPAD_EINVAL;
  if (n == 0) return HYPAD_OK;
  hipLaunchKernelGGL(combine_kernel, dim3(grid_for(n, THREADS)), dim3(THREADS), 0, (hipStream_t)s, mode, c, r, u, out, n);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

}  // extern "C"
