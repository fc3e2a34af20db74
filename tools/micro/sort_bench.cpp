// Timing of mimosa_amd/csrc/exact_sort.hpp on a detectFeatures-like sequence (31 k packed words, 8-bit keys, exponential
// key distribution): std::sort, the restated introsort on 1 / 2 / 4 / 8 threads (persistent helpers), one partition step in
// its scanning and its list form.   g++ -O2 -std=c++17 -pthread [-DMH_LIST_PARTITION_MIN=n] tools/micro/sort_bench.cpp
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <random>

#include "../../mimosa_amd/csrc/exact_sort.hpp"

int main()
{
  std::mt19937 rng(7);
  const int n = 31187;
  std::vector<uint32_t> base(n);
  std::exponential_distribution<double> ex(1.0 / 12.0);
  for (int i = 0; i < n; ++i) base[i] = (std::min(255u, 10u + static_cast<uint32_t>(ex(rng))) << 24) | static_cast<uint32_t>(i);
  auto cmp = [](uint32_t a, uint32_t b) { return (a >> 24) > (b >> 24); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  std::vector<uint32_t> ref = base;
  std::sort(ref.begin(), ref.end(), cmp);
  for (int th : {0, 1, 2, 4, 8}) {
    mh::exact_sort::Pool pool(th > 1 ? th - 1 : 0);
    double best = 1e9;
    for (int r = 0; r < 50; ++r) {
      auto v = base;
      const auto t0 = std::chrono::steady_clock::now();
      if (th == 0) std::sort(v.begin(), v.end(), cmp);
      else if (th == 1) mh::exact_sort::sort_sequential(v.data(), v.data() + n, cmp);
      else mh::exact_sort::sort_parallel(v.data(), v.data() + n, cmp, th, 4096, &pool);
      best = std::min(best, us(t0, std::chrono::steady_clock::now()));
      if (v != ref) { std::printf("MISMATCH\n"); return 1; }
    }
    std::printf("%s %d: %.0f us\n", th ? "threads" : "std::sort", th, best);
  }
  std::vector<uint32_t> idx(2 * n);
  double bc = 1e9, bl = 1e9;
  for (int r = 0; r < 100; ++r) {
    auto v = base, w = base;
    const auto t0 = std::chrono::steady_clock::now();
    mh::exact_sort::partition_pivot(v.data(), v.data() + n, cmp);
    const auto t1 = std::chrono::steady_clock::now();
    mh::exact_sort::partition_pivot_lists(w.data(), w.data() + n, cmp, idx.data());
    const auto t2 = std::chrono::steady_clock::now();
    bc = std::min(bc, us(t0, t1));
    bl = std::min(bl, us(t1, t2));
    if (v != w) { std::printf("MISMATCH (partition)\n"); return 1; }
  }
  std::printf("one partition of %d: scanning %.1f us, lists %.1f us (MH_LIST_PARTITION_MIN %d)\n", n, bc, bl, static_cast<int>(mh::exact_sort::kListPartitionMin));
  return 0;
}
