// Phase timing of hypad_host_mt19937_normal's pieces on this host: g++ -O3 -ffp-contract=off -pthread scripts/host_rng_bench.cpp -o /tmp/hrb && /tmp/hrb
#include <chrono>
#include <cstdio>
#define main_guard
#include "../hypad_amd/csrc/host_rng.cpp"
int main() {
  static uint32_t key[624]; for (int i = 0; i < 624; ++i) key[i] = i * 2654435761u + 1;
  const int64_t pairs = 204160;
  std::vector<double> x1(pairs + 512), x2(pairs + 512), r2(pairs + 512);
  static Gen gen; gen.g.key = key; gen.g.pos = 624;
  static uint32_t w[1 << 20];
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    gen.words(w, 1 << 20);
    auto t1 = std::chrono::steady_clock::now();
    gen.accepted(pairs, x1.data(), x2.data(), r2.data());
    auto t2 = std::chrono::steady_clock::now();
    std::vector<float> o(2 * pairs); float* op = o.data(); float* const outs[1] = {op}; Dest dst{outs, 1, 2 * pairs}; double c;
    transform_range(x1.data(), x2.data(), r2.data(), 0, pairs, 0, 2 * pairs, dst, &c);
    auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    {
      static std::vector<float> zx(145 * 1280), zz(145 * 1280), zg(29 * 1280);
      float* const o2[2] = {zx.data(), zz.data()}; float* const o1[1] = {zg.data()};
      int pos = gen.g.pos, has = 0; double cg = 0;
      auto a0 = std::chrono::steady_clock::now();
      hypad_host_mt19937_normal(key, &pos, &has, &cg, o2, 2, 1280, 145);
      auto a1 = std::chrono::steady_clock::now();
      hypad_host_mt19937_normal(key, &pos, &has, &cg, o1, 1, 1280, 29);
      auto a2 = std::chrono::steady_clock::now();
      gen.g.pos = pos;
      printf("entry point: critic planes %.3f ms, generator plane %.3f ms | ", std::chrono::duration<double, std::milli>(a1 - a0).count(), std::chrono::duration<double, std::milli>(a2 - a1).count());
    }
    printf("1M words %.3f ms | accepted(204160 pairs) %.3f ms | transform (1 thread) %.3f ms  (%g)\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), (double)o[7]);
  }
}
