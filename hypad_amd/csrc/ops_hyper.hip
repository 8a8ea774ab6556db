// Stand-alone Poincare-ball ops: HBM-bound row-wise kernels, one 64-lane wave per row,
// grid-stride over rows; plus the pair-wise distance (MFMA x y^T with a fused acosh epilogue).
#include <hip/hip_runtime.h>

#include "../../include/hypad.h"
#include "rowops.h"
#include "tile_gemm.h"

using namespace hypad;

namespace {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;

inline int row_grid(int64_t rows) {
  int64_t blocks = (rows + WAVES * 4 - 1) / (WAVES * 4);
  if (blocks > 4096) blocks = 4096;   // >> 256 CUs; the rest grid-strides
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

enum UnaryOp { OP_EXPMAP0, OP_EXPMAP0_BWD, OP_LOGMAP0, OP_LOGMAP0_BWD, OP_PROJECT, OP_PROJECT_BWD };

// Rows per wave and iteration: a row is only 4*dim bytes, so each wave keeps RPW consecutive rows in flight (all loads
// issued before the first reduction, branch-free, clamped row index) -- the kernels are HBM-bound and one 400-byte row
// per wave leaves the memory pipeline mostly empty.
constexpr int RPW = 4;

// RPW consecutive rows are RPW*dim contiguous floats starting on a 16-byte boundary (r % 4 == 0): they move between HBM and
// a wave-private LDS slab as `dim` float4s (16 B per lane per instruction, the shape that reaches HBM rate) and between the
// slab and the row registers with the usual lane + 64 e pattern.  LDS instructions of a wave execute in order; the fence
// only keeps the compiler from reordering a lane's accesses around other lanes' accesses of the same wave.
__device__ __forceinline__ void slab_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Cache hints (round 3, measured at 2 000 000 rows of 100): result rows are written once and not read again by the kernel --
// non-temporal stores take expmap0 / logmap0 / project / mobius_add from 5.5-5.6 to 5.8-5.9 TB/s; operand rows as non-temporal
// loads help only the kernel that writes almost nothing (row distance: +1.5 % on the same box) and cost the others 1-2 %.
typedef float rows_f4 __attribute__((ext_vector_type(4)));
template <bool NT = false>
__device__ __forceinline__ void slab_in(float* slab, const float* __restrict__ src, int dim, int lane) {
  for (int k = lane; k < dim; k += 64) {
    if constexpr (NT) reinterpret_cast<rows_f4*>(slab)[k] = __builtin_nontemporal_load(reinterpret_cast<const rows_f4*>(src) + k);
    else reinterpret_cast<float4*>(slab)[k] = reinterpret_cast<const float4*>(src)[k];
  }
}
__device__ __forceinline__ void slab_out(float* __restrict__ dst, const float* slab, int dim, int lane) {
  for (int k = lane; k < dim; k += 64) __builtin_nontemporal_store(reinterpret_cast<const rows_f4*>(slab)[k], reinterpret_cast<rows_f4*>(dst) + k);
}
inline size_t slab_bytes(int dim, int nbuf) { return (size_t)WAVES * nbuf * RPW * dim * sizeof(float); }
// elements per lane of the 16-lanes-per-row layout, by row length
#define HYPAD_EPL_DISPATCH(dim, ...)                        \
  do {                                                      \
    if ((dim) <= 64) { constexpr int EPL = 4; __VA_ARGS__; }        \
    else if ((dim) <= 128) { constexpr int EPL = 8; __VA_ARGS__; }  \
    else { constexpr int EPL = 16; __VA_ARGS__; }           \
  } while (0)

// Four rows per wave, one per DPP row of 16 lanes (RowT<16, EPL>, rowops.h).  Row r0 + (lane >> 4) belongs to the lane.
template <int OP, int EPL>
__global__ __launch_bounds__(THREADS) void unary_rows(const float* __restrict__ a, const float* __restrict__ g,
                                                       float* __restrict__ out, int64_t rows, int dim) {
  using R = RowT<16, EPL>;
  const int lane = threadIdx.x & 63, sub = lane >> 4;
  const int64_t wave0 = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * RPW;
  const int64_t stride = (int64_t)gridDim.x * WAVES * RPW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool BWD = OP == OP_EXPMAP0_BWD || OP == OP_LOGMAP0_BWD || OP == OP_PROJECT_BWD;
  float* sa = smem + (threadIdx.x >> 6) * 3 * RPW * dim;
  float* sg = sa + RPW * dim;
  float* so = sg + RPW * dim;
  const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(out) | (BWD ? reinterpret_cast<uintptr_t>(g) : 0)) & 15) == 0;
  for (int64_t r = wave0; r < rows; r += stride) {
    const bool full = al && r + RPW <= rows;
    const int64_t rr = r + sub < rows ? r + sub : rows - 1;
    R x, go, o;
    if (full) {
      slab_in(sa, a + r * dim, dim, lane);
      if (BWD) slab_in(sg, g + r * dim, dim, lane);
      slab_fence();
      x = row_load<R>(sa + sub * dim, dim, lane);
      if (BWD) go = row_load<R>(sg + sub * dim, dim, lane);
    } else {
      x = row_load<R>(a + rr * dim, dim, lane);
      if (BWD) go = row_load<R>(g + rr * dim, dim, lane);
    }
    if (OP == OP_EXPMAP0) o = expmap0_row(x);
    else if (OP == OP_LOGMAP0) o = logmap0_row(x);
    else if (OP == OP_PROJECT) o = project_row(x);
    else if (OP == OP_EXPMAP0_BWD) o = expmap0_row_bwd(x, go);
    else if (OP == OP_LOGMAP0_BWD) o = logmap0_row_bwd(x, go);
    else o = project_row_bwd(x, go);
    if (full) {
      row_store(so + sub * dim, o, dim, lane);
      slab_fence();
      slab_out(out + r * dim, so, dim, lane);
      slab_fence();
    } else if (r + sub < rows) {
      row_store(out + (r + sub) * dim, o, dim, lane);
    }
  }
}

template <int EPL>
__global__ __launch_bounds__(THREADS) void mobius_add_rows(const float* __restrict__ x, const float* __restrict__ y,
                                                            float* __restrict__ out, int64_t rows, int dim, int ybc) {
  using R = RowT<16, EPL>;
  const int lane = threadIdx.x & 63, sub = lane >> 4;
  const int64_t wave0 = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * RPW;
  const int64_t stride = (int64_t)gridDim.x * WAVES * RPW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sa = smem + (threadIdx.x >> 6) * 3 * RPW * dim;
  float* sb = sa + RPW * dim;
  float* so = sb + RPW * dim;
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | (ybc ? 0 : reinterpret_cast<uintptr_t>(y))) & 15) == 0;
  for (int64_t r = wave0; r < rows; r += stride) {
    const bool full = al && r + RPW <= rows;
    const int64_t rr = r + sub < rows ? r + sub : rows - 1;
    R a, b;
    if (full) {
      slab_in(sa, x + r * dim, dim, lane);
      if (!ybc) slab_in(sb, y + r * dim, dim, lane);
      slab_fence();
      a = row_load<R>(sa + sub * dim, dim, lane);
      b = ybc ? row_load<R>(y, dim, lane) : row_load<R>(sb + sub * dim, dim, lane);
    } else {
      a = row_load<R>(x + rr * dim, dim, lane);
      b = row_load<R>(y + (ybc ? 0 : rr * dim), dim, lane);
    }
    const R o = mobius_add_row(a, b);
    if (full) {
      row_store(so + sub * dim, o, dim, lane);
      slab_fence();
      slab_out(out + r * dim, so, dim, lane);
      slab_fence();
    } else if (r + sub < rows) {
      row_store(out + (r + sub) * dim, o, dim, lane);
    }
  }
}
__global__ __launch_bounds__(THREADS) void mobius_add_rows_bwd(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ go, float* __restrict__ gx,
                                                                float* __restrict__ gy, int64_t rows, int dim, int ybc) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  for (int64_t r = wave0; r < rows; r += stride) {
    RowVec a = row_load(x + r * dim, dim, lane);
    RowVec b = row_load(y + (ybc ? 0 : r * dim), dim, lane);
    RowVec g = row_load(go + r * dim, dim, lane);
    RowVec da, db;
    mobius_add_row_bwd(a, b, g, da, db);
    row_store(gx + r * dim, da, dim, lane);
    row_store(gy + r * dim, db, dim, lane);
  }
}

template <int EPL>
__global__ __launch_bounds__(THREADS) void head_rows(const float* __restrict__ u, const float* __restrict__ bias,
                                                      float* __restrict__ out, int64_t rows, int dim) {
  using R = RowT<16, EPL>;
  const int lane = threadIdx.x & 63, sub = lane >> 4;
  const int64_t wave0 = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * RPW;
  const int64_t stride = (int64_t)gridDim.x * WAVES * RPW;
  const R b = row_load<R>(bias, dim, lane);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sa = smem + (threadIdx.x >> 6) * 2 * RPW * dim;
  float* so = sa + RPW * dim;
  const bool al = ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  for (int64_t r = wave0; r < rows; r += stride) {
    const bool full = al && r + RPW <= rows;
    R x;
    if (full) { slab_in(sa, u + r * dim, dim, lane); slab_fence(); x = row_load<R>(sa + sub * dim, dim, lane); }
    else x = row_load<R>(u + (r + sub < rows ? r + sub : rows - 1) * dim, dim, lane);
    x = head_row(x, b);
    if (full) {
      row_store(so + sub * dim, x, dim, lane);
      slab_fence();
      slab_out(out + r * dim, so, dim, lane);
      slab_fence();
    } else if (r + sub < rows) {
      row_store(out + (r + sub) * dim, x, dim, lane);
    }
  }
}
__global__ __launch_bounds__(THREADS) void head_rows_bwd(const float* __restrict__ u, const float* __restrict__ bias,
                                                          const float* __restrict__ go, float* __restrict__ gu,
                                                          float* __restrict__ gb_rows, int64_t rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const RowVec b = row_load(bias, dim, lane);
  for (int64_t r = wave0; r < rows; r += stride) {
    RowVec du, db;
    head_row_bwd(row_load(u + r * dim, dim, lane), b, row_load(go + r * dim, dim, lane), du, db);
    row_store(gu + r * dim, du, dim, lane);
    if (gb_rows) row_store(gb_rows + r * dim, db, dim, lane);
  }
}

template <int EPL>
__global__ __launch_bounds__(THREADS) void rowdist_rows(const float* __restrict__ u, const float* __restrict__ v,
                                                         float* __restrict__ dist, int64_t rows, int dim) {
  using R = RowT<16, EPL>;
  const int lane = threadIdx.x & 63, sub = lane >> 4;
  const int64_t wave0 = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * RPW;
  const int64_t stride = (int64_t)gridDim.x * WAVES * RPW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sa = smem + (threadIdx.x >> 6) * 2 * RPW * dim;
  float* sb = sa + RPW * dim;
  const bool al = ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  for (int64_t r = wave0; r < rows; r += stride) {
    const bool full = al && r + RPW <= rows;
    const int64_t rr = r + sub < rows ? r + sub : rows - 1;
    R a, b;
    if (full) {
      slab_in<true>(sa, u + r * dim, dim, lane);
      slab_in<true>(sb, v + r * dim, dim, lane);
      slab_fence();
      a = row_load<R>(sa + sub * dim, dim, lane);
      b = row_load<R>(sb + sub * dim, dim, lane);
    } else {
      a = row_load<R>(u + rr * dim, dim, lane);
      b = row_load<R>(v + rr * dim, dim, lane);
    }
    const float d = rowdist_row(a, b);
    if ((lane & 15) == 0 && r + sub < rows) dist[r + sub] = d;
    slab_fence();
  }
}
// gd_scalar used when gd == nullptr (hyperbolic loss: every row gets grad_loss / batch)
__global__ __launch_bounds__(THREADS) void rowdist_rows_bwd(const float* __restrict__ u, const float* __restrict__ v,
                                                             const float* __restrict__ gd, float gd_scalar,
                                                             float* __restrict__ gu, float* __restrict__ gv,
                                                             int64_t rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  for (int64_t r = wave0; r < rows; r += stride) {
    RowVec du, dv;
    rowdist_row_bwd(row_load(u + r * dim, dim, lane), row_load(v + r * dim, dim, lane), gd ? gd[r] : gd_scalar, du, dv);
    row_store(gu + r * dim, du, dim, lane);
    row_store(gv + r * dim, dv, dim, lane);
  }
}

// loss[0] = sum_r dist_r / batch: one workgroup, fixed summation order (deterministic)
__global__ __launch_bounds__(1024) void hyper_loss_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                           float* __restrict__ loss, int64_t rows, int dim, int batch) {
  __shared__ float part[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int64_t r = wave; r < rows; r += 16) acc += rowdist_row(row_load(u + r * dim, dim, lane), row_load(v + r * dim, dim, lane));
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += part[w];
    loss[0] = s / (float)batch;
  }
}

__global__ __launch_bounds__(THREADS) void column_sum_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int64_t rows, int dim) {
  int c = blockIdx.x * THREADS + threadIdx.x;
  if (c >= dim) return;
  float s = 0.f;
  for (int64_t r = 0; r < rows; ++r) s += in[r * dim + c];
  out[c] = s;
}

// ---- pair-wise distance (hyperspace/poincare_distance.py:5-16): 64x64 output tile per workgroup,
// x y^T on fp32 MFMA, clamps 1e-5 (norms) / 1e-7 (squared distance) in the epilogue.
constexpr int PT = 64;
__global__ __launch_bounds__(THREADS) void pairdist_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                            float* __restrict__ out, int n, int m, int dim) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ld = ((dim + 3) & ~3) + 4;
  float* xs = smem;                 // [64][ld]
  float* ys = xs + PT * ld;         // [64][ld]
  float* xn = ys + PT * ld;         // [64] raw squared norms
  float* yn = xn + PT;
  const int r0 = blockIdx.y * PT, c0 = blockIdx.x * PT;
  for (int i = threadIdx.x; i < PT * ld; i += THREADS) {
    int r = i / ld, c = i % ld;
    xs[i] = (r0 + r < n && c < dim) ? X[(size_t)(r0 + r) * dim + c] : 0.f;
    ys[i] = (c0 + r < m && c < dim) ? Y[(size_t)(c0 + r) * dim + c] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 2 * PT) {
    const float* p = (threadIdx.x < PT ? xs + threadIdx.x * ld : ys + (threadIdx.x - PT) * ld);
    float s = 0.f;
    for (int c = 0; c < dim; ++c) s += p[c] * p[c];
    (threadIdx.x < PT ? xn : yn)[threadIdx.x & (PT - 1)] = s;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  // wave w owns output rows [16w, 16w+16) x all 64 columns (4 accumulators)
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < dim; k0 += 4) {
    const int k = k0 + q;
    const float a = k < dim ? xs[(wave * 16 + j) * ld + k] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = k < dim ? ys[(t * 16 + j) * ld + k] : 0.f;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = wave * 16 + 4 * q + r, lc = t * 16 + j;
      const int gr = r0 + lr, gc = c0 + lc;
      if (gr < n && gc < m) {
        float d = fmaxf(xn[lr] + yn[lc] - 2.0f * acc[t][r], 1e-7f);
        float a = 1.f - fmaxf(xn[lr], 1e-5f), b = 1.f - fmaxf(yn[lc], 1e-5f);
        out[(size_t)gr * m + gc] = acoshf(1.f + 2.f * d / (a * b));
      }
    }
}

inline int check_rows(const void* a, const void* b, int64_t rows, int dim) {
  if (!a || !b || rows < 0 || dim <= 0) return HYPAD_EINVAL;
  if (dim > 64 * MAX_EPL) return HYPAD_EUNSUPPORTED;
  return HYPAD_OK;
}

template <int OP>
int launch_unary(const float* a, const float* g, float* out, int64_t rows, int dim, hypad_stream_t stream) {
  int rc = check_rows(a, out, rows, dim);
  if (rc) return rc;
  if (rows == 0) return HYPAD_OK;
  HYPAD_EPL_DISPATCH(dim, hipLaunchKernelGGL((unary_rows<OP, EPL>), dim3(row_grid(rows)), dim3(THREADS), slab_bytes(dim, 3), (hipStream_t)stream, a, g, out, rows, dim));
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

}  // namespace

extern "C" {

int hypad_expmap0_fwd(const float* u, float* out, int64_t rows, int dim, hypad_stream_t s) {
  return launch_unary<OP_EXPMAP0>(u, nullptr, out, rows, dim, s);
}
int hypad_expmap0_bwd(const float* u, const float* go, float* gu, int64_t rows, int dim, hypad_stream_t s) {
  if (!go) return HYPAD_EINVAL;
  return launch_unary<OP_EXPMAP0_BWD>(u, go, gu, rows, dim, s);
}
int hypad_logmap0_fwd(const float* y, float* out, int64_t rows, int dim, hypad_stream_t s) {
  return launch_unary<OP_LOGMAP0>(y, nullptr, out, rows, dim, s);
}
int hypad_logmap0_bwd(const float* y, const float* go, float* gy, int64_t rows, int dim, hypad_stream_t s) {
  if (!go) return HYPAD_EINVAL;
  return launch_unary<OP_LOGMAP0_BWD>(y, go, gy, rows, dim, s);
}
int hypad_project_fwd(const float* x, float* out, int64_t rows, int dim, hypad_stream_t s) {
  return launch_unary<OP_PROJECT>(x, nullptr, out, rows, dim, s);
}
int hypad_project_bwd(const float* x, const float* go, float* gx, int64_t rows, int dim, hypad_stream_t s) {
  if (!go) return HYPAD_EINVAL;
  return launch_unary<OP_PROJECT_BWD>(x, go, gx, rows, dim, s);
}

int hypad_mobius_add_fwd(const float* x, const float* y, float* out, int64_t rows, int dim, int64_t y_rows, hypad_stream_t s) {
  int rc = check_rows(x, out, rows, dim);
  if (rc) return rc;
  if (!y || (y_rows != 1 && y_rows != rows)) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  HYPAD_EPL_DISPATCH(dim, hipLaunchKernelGGL((mobius_add_rows<EPL>), dim3(row_grid(rows)), dim3(THREADS), slab_bytes(dim, 3), (hipStream_t)s, x, y, out, rows, dim,
                     (int)(y_rows == 1 && rows != 1)));
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_mobius_add_bwd(const float* x, const float* y, const float* go, float* gx, float* gy, int64_t rows, int dim,
                         int64_t y_rows, hypad_stream_t s) {
  int rc = check_rows(x, gx, rows, dim);
  if (rc) return rc;
  if (!y || !go || !gy || (y_rows != 1 && y_rows != rows)) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  hipLaunchKernelGGL(mobius_add_rows_bwd, dim3(row_grid(rows)), dim3(THREADS), 0, (hipStream_t)s, x, y, go, gx, gy, rows,
                     dim, (int)(y_rows == 1 && rows != 1));
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_mobius_head_fwd(const float* u, const float* bias, float* out, int64_t rows, int dim, hypad_stream_t s) {
  int rc = check_rows(u, out, rows, dim);
  if (rc) return rc;
  if (!bias) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  HYPAD_EPL_DISPATCH(dim, hipLaunchKernelGGL((head_rows<EPL>), dim3(row_grid(rows)), dim3(THREADS), slab_bytes(dim, 2), (hipStream_t)s, u, bias, out, rows, dim));
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_mobius_head_bwd(const float* u, const float* bias, const float* go, float* gu, float* gb_rows, int64_t rows,
                          int dim, hypad_stream_t s) {
  int rc = check_rows(u, gu, rows, dim);
  if (rc) return rc;
  if (!bias || !go) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  hipLaunchKernelGGL(head_rows_bwd, dim3(row_grid(rows)), dim3(THREADS), 0, (hipStream_t)s, u, bias, go, gu, gb_rows, rows, dim);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_poincare_rowdist_fwd(const float* u, const float* v, float* dist, int64_t rows, int dim, hypad_stream_t s) {
  int rc = check_rows(u, v, rows, dim);
  if (rc) return rc;
  if (!dist) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  HYPAD_EPL_DISPATCH(dim, hipLaunchKernelGGL((rowdist_rows<EPL>), dim3(row_grid(rows)), dim3(THREADS), slab_bytes(dim, 2), (hipStream_t)s, u, v, dist, rows, dim));
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_poincare_rowdist_bwd(const float* u, const float* v, const float* gd, float* gu, float* gv, int64_t rows,
                               int dim, hypad_stream_t s) {
  int rc = check_rows(u, v, rows, dim);
  if (rc) return rc;
  if (!gd || !gu || !gv) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  hipLaunchKernelGGL(rowdist_rows_bwd, dim3(row_grid(rows)), dim3(THREADS), 0, (hipStream_t)s, u, v, gd, 0.f, gu, gv, rows, dim);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_hyper_loss_fwd(const float* u, const float* v, float* loss, int64_t rows, int dim, int batch, hypad_stream_t s) {
  int rc = check_rows(u, v, rows, dim);
  if (rc) return rc;
  if (!loss || batch <= 0) return HYPAD_EINVAL;
  hipLaunchKernelGGL(hyper_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, u, v, loss, rows, dim, batch);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}
int hypad_hyper_loss_bwd(const float* u, const float* v, float grad_loss, float* gu, float* gv, int64_t rows, int dim,
                         int batch, hypad_stream_t s) {
  int rc = check_rows(u, v, rows, dim);
  if (rc) return rc;
  if (!gu || !gv || batch <= 0) return HYPAD_EINVAL;
  if (rows == 0) return HYPAD_OK;
  hipLaunchKernelGGL(rowdist_rows_bwd, dim3(row_grid(rows)), dim3(THREADS), 0, (hipStream_t)s, u, v,
                     (const float*)nullptr, grad_loss / (float)batch, gu, gv, rows, dim);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_poincare_pairdist_fwd(const float* pred, const float* gt, float* out, int n, int m, int dim, hypad_stream_t s) {
  if (!pred || !gt || !out || n < 0 || m < 0 || dim <= 0) return HYPAD_EINVAL;
  if (n == 0 || m == 0) return HYPAD_OK;
  const int ld = ((dim + 3) & ~3) + 4;
  size_t lds = (size_t)(2 * PT * ld + 2 * PT) * sizeof(float);
  if (lds > 160 * 1024) return HYPAD_EUNSUPPORTED;
  hipLaunchKernelGGL(pairdist_kernel, dim3((m + PT - 1) / PT, (n + PT - 1) / PT), dim3(THREADS), lds, (hipStream_t)s,
                     pred, gt, out, n, m, dim);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

int hypad_column_sum(const float* in, float* out, int64_t rows, int dim, hypad_stream_t s) {
  if (!in || !out || rows < 0 || dim <= 0) return HYPAD_EINVAL;
  hipLaunchKernelGGL(column_sum_kernel, dim3((dim + THREADS - 1) / THREADS), dim3(THREADS), 0, (hipStream_t)s, in, out, rows, dim);
  HYPAD_CHECK_LAUNCH();
  return HYPAD_OK;
}

}  // extern "C"
