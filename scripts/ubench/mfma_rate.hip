// Issue cost of the fp32 matrix instructions on one SIMD (shader clock, one wave / two waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/mfma_rate scripts/ubench/mfma_rate.hip ; (GPU box) ./ab_libs/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int CHAINS>
__global__ void k(float* o, long long* t, int n) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      else acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int c = 1; c < CHAINS; ++c) s += acc[c];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { t[2 * (threadIdx.x >> 6)] = t0; t[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int KIND, int CHAINS>
void run(const char* name, int threads) {
  float* o; long long* t; hipMalloc(&o, 4 * 1024); hipMalloc(&t, 8 * 32);
  const int n = 1024;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(1), dim3(threads), 0, 0, o, t, n);
  long long h[32]; hipMemcpy(h, t, 8 * 32, hipMemcpyDeviceToHost);
  long long lo = h[0], hi = h[1];
  for (int w = 0; w < threads / 64; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
  printf("%-8s chains %d waves/SIMD %d : wave 0 alone %.2f ticks per instruction; all waves' span / (instructions per SIMD) %.2f\n", name, CHAINS, threads / 256,
         (double)(h[1] - h[0]) / (n * CHAINS), (double)(hi - lo) / (n * CHAINS * (threads / 256)));
  hipFree(o); hipFree(t);
}
int main() {
  run<0, 1>("16x16x4", 256); run<0, 2>("16x16x4", 256); run<0, 2>("16x16x4", 512);
  run<1, 1>("4x4x1", 256);  run<1, 2>("4x4x1", 256);  run<1, 4>("4x4x1", 256);  run<1, 8>("4x4x1", 256);  run<1, 4>("4x4x1", 512);  run<1, 8>("4x4x1", 512);
  return 0;
}
