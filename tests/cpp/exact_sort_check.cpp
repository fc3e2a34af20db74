// CPU check of mimosa_amd/csrc/exact_sort.hpp: the restated introsort (one thread and several) must leave every sequence
// exactly as this toolchain's std::sort leaves it — including the order of equal keys, which is what detectFeatures'
// non-maximum suppression depends on.  Prints "OK <cases>"; tests/test_capi_cpu.py runs it.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "../../mimosa_amd/csrc/exact_sort.hpp"

int main()
{
  std::mt19937_64 rng(12345);
  auto by_gradient = [](uint32_t a, uint32_t b) { return (a >> 24) > (b >> 24); };  // photo_api.hip's comparator
  long cases = 0;
  double t_std = 0, t_par = 0;
  mh::exact_sort::Pool pool(3);  // every other case runs on the persistent helpers
  for (int rep = 0; rep < 400; ++rep) {
    const size_t n = rep < 40 ? static_cast<size_t>(rep) : (rep % 7 == 0 ? 30000 + rng() % 40000 : rng() % 9000);
    const int distinct = 1 + static_cast<int>(rng() % (rep % 3 == 0 ? 3 : 246));  // few keys: thousands of ties
    std::vector<uint32_t> v(n);
    const int pattern = rep % 6;
    for (size_t i = 0; i < n; ++i) {
      uint32_t key;
      switch (pattern) {
        case 0: key = static_cast<uint32_t>(rng() % distinct); break;
        case 1: key = static_cast<uint32_t>(i * distinct / (n ? n : 1)); break;                       // ascending
        case 2: key = static_cast<uint32_t>((n - i) * distinct / (n ? n : 1)); break;                 // descending
        case 3: key = static_cast<uint32_t>((i < n / 2 ? i : n - i) * 2 * distinct / (n ? n : 1)); break;  // organ pipe
        case 4: key = 7; break;                                                                      // all equal
        default: key = static_cast<uint32_t>((rng() % 100 < 90) ? 10 : rng() % distinct); break;     // one dominant key
      }
      v[i] = (std::min<uint32_t>(key, 255u) << 24) | static_cast<uint32_t>(i & 0xFFFFFFu);            // payload = position: ties are visible
    }
    std::vector<uint32_t> a = v, b = v, c = v;
    auto t0 = std::chrono::steady_clock::now();
    std::sort(a.begin(), a.end(), by_gradient);
    auto t1 = std::chrono::steady_clock::now();
    mh::exact_sort::sort_sequential(b.data(), b.data() + b.size(), by_gradient);
    auto t2 = std::chrono::steady_clock::now();
    mh::exact_sort::sort_parallel(c.data(), c.data() + c.size(), by_gradient, 2 + rep % 5, rep % 2 ? 4096 : 600, rep % 4 < 2 ? &pool : nullptr);
    auto t3 = std::chrono::steady_clock::now();
    if (n >= 30000) {
      t_std += std::chrono::duration<double, std::micro>(t1 - t0).count();
      t_par += std::chrono::duration<double, std::micro>(t3 - t2).count();
    }
    if (a != b || a != c) {
      std::printf("MISMATCH rep %d n %zu pattern %d distinct %d (sequential %d, parallel %d)\n", rep, n, pattern, distinct, a == b, a == c);
      return 1;
    }
    ++cases;
  }
  // one partition step, scanning form against list form: same array afterwards, same cut
  for (int rep = 0; rep < 300; ++rep) {
    const size_t n = 17 + rng() % (rep % 5 == 0 ? 40000 : 3000);
    const int distinct = 1 + static_cast<int>(rng() % (rep % 3 == 0 ? 2 : 200));
    std::vector<uint32_t> v(n);
    for (size_t i = 0; i < n; ++i) {
      const uint32_t key = rep % 7 == 0 ? static_cast<uint32_t>(i * distinct / n) : static_cast<uint32_t>(rng() % distinct);
      v[i] = (std::min<uint32_t>(key, 255u) << 24) | static_cast<uint32_t>(i & 0xFFFFFFu);
    }
    std::vector<uint32_t> a = v, b = v, idx(2 * n);
    const auto ca = mh::exact_sort::partition_pivot(a.data(), a.data() + n, by_gradient) - a.data();
    const auto cb = mh::exact_sort::partition_pivot_lists(b.data(), b.data() + n, by_gradient, idx.data()) - b.data();
    if (a != b || ca != cb) {
      std::printf("MISMATCH (partition) rep %d n %zu distinct %d cut %ld vs %ld\n", rep, n, distinct, static_cast<long>(ca), static_cast<long>(cb));
      return 1;
    }
    ++cases;
  }
  // a full-word comparator (no ties) and a depth-exhausting input for the heap-sort branch: median-of-three killer
  for (int rep = 0; rep < 20; ++rep) {
    const size_t n = 5000 + 997 * rep;
    std::vector<uint32_t> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = static_cast<uint32_t>(rng());
    if (rep % 2) {  // Musser's adversary, approximately: pairs arranged so that the medians are always poor
      for (size_t i = 0; i < n; ++i) v[i] = static_cast<uint32_t>(i % 2 ? i / 2 : n - i / 2);
    }
    std::vector<uint32_t> a = v, b = v, c = v;
    std::sort(a.begin(), a.end());
    mh::exact_sort::sort_sequential(b.data(), b.data() + n, std::less<uint32_t>());
    mh::exact_sort::sort_parallel(c.data(), c.data() + n, std::less<uint32_t>(), 4, 512, rep % 3 ? &pool : nullptr);
    if (a != b || a != c) {
      std::printf("MISMATCH (plain) rep %d\n", rep);
      return 1;
    }
    ++cases;
  }
  std::printf("OK %ld  (>= 30000 elements: std::sort %.0f us, parallel %.0f us in total)\n", cases, t_std, t_par);
  return 0;
}
