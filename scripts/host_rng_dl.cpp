#include <chrono>
#include <cstdio>
#include <cstdint>
#include <dlfcn.h>
#include <vector>
typedef int (*fn_t)(uint32_t*, int*, int*, double*, float* const*, int, int64_t, int64_t);
int main(int argc, char** argv) {
  void* h = dlopen(argv[1], RTLD_NOW); if (!h) { printf("%s\n", dlerror()); return 1; }
  fn_t f = (fn_t)dlsym(h, "hypad_host_mt19937_normal");
  static uint32_t key[624]; for (int i = 0; i < 624; ++i) key[i] = i * 2654435761u + 1;
  std::vector<float> zx(145 * 1280), zz(145 * 1280); float* const o2[2] = {zx.data(), zz.data()};
  int pos = 624, has = 0; double cg = 0;
  for (int rep = 0; rep < 5; ++rep) {
    auto a0 = std::chrono::steady_clock::now();
    f(key, &pos, &has, &cg, o2, 2, 1280, 145);
    auto a1 = std::chrono::steady_clock::now();
    printf("dlopen'd entry point: critic planes %.3f ms\n", std::chrono::duration<double, std::milli>(a1 - a0).count());
  }
}
