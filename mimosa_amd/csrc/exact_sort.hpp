// std::sort's result, computed on several host threads.
//
// Photometric::detectFeatures (src/lidar/photometric.cpp:556-560) sorts ~30 000 (gradient, pixel) pairs with a comparator
// that looks at the 8-bit gradient only.  std::sort is not stable: which of several thousand equal-gradient pixels comes
// first is decided by libstdc++'s introsort (median-of-three pivot moved to the front, unguarded Hoare partition, recursion
// on the right part, a final insertion sort), and that order decides the greedy non-maximum suppression that follows — so
// the feature set of the reference is only reproduced by running THAT algorithm on THAT sequence.  It is the largest single
// item of a frame's feature bookkeeping (0.49 ms of 0.79 ms).
//
// The algorithm's structure allows more than one thread: after a partition step the two parts are disjoint and what happens
// to each depends on its contents and the remaining depth budget alone.  This header restates the introsort loop (own code,
// written from the algorithm's description: GCC's bits/stl_algo.h, std::__introsort_loop / __unguarded_partition_pivot /
// __final_insertion_sort, unchanged since GCC 4.x apart from the pivot landing at `first`) and runs disjoint parts on
// worker threads; the depth-exhausted fallback calls std::partial_sort(first, last, last), which IS the library's own heap
// sort.  tests/cpp/exact_sort_check.cpp compares it element for element with std::sort on random, duplicate-heavy,
// sorted, reversed and organ-pipe sequences; the photometric parity tests compare the features against an oracle that
// calls std::sort.
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace mh
{
namespace exact_sort
{
constexpr std::ptrdiff_t kThreshold = 16;  // _S_threshold
#ifndef MH_LIST_PARTITION_MIN
#define MH_LIST_PARTITION_MIN 1024
#endif
constexpr std::ptrdiff_t kListPartitionMin = MH_LIST_PARTITION_MIN;  // ranges at least this long are partitioned by partition_pivot_lists

inline long floor_log2(std::ptrdiff_t n)
{
  long l = 0;
  while (n > 1) {
    n >>= 1;
    ++l;
  }
  return l;
}

template <class T, class Comp>
inline void median_to_first(T * result, T * a, T * b, T * c, Comp comp)
{
  if (comp(*a, *b)) {
    if (comp(*b, *c))
      std::iter_swap(result, b);
    else if (comp(*a, *c))
      std::iter_swap(result, c);
    else
      std::iter_swap(result, a);
  } else if (comp(*a, *c))
    std::iter_swap(result, a);
  else if (comp(*b, *c))
    std::iter_swap(result, c);
  else
    std::iter_swap(result, b);
}

template <class T, class Comp>
inline T * partition_pivot(T * first, T * last, Comp comp)
{
  T * mid = first + (last - first) / 2;
  median_to_first(first, first + 1, mid, last - 1, comp);
  T * pivot = first;
  T * lo = first + 1;
  T * hi = last;
  while (true) {
    while (comp(*lo, *pivot)) ++lo;
    --hi;
    while (comp(*pivot, *hi)) --hi;
    if (!(lo < hi)) return lo;
    std::iter_swap(lo, hi);
    ++lo;
  }
}

// The same partition — same swaps in the same order, same cut — without the data-dependent branches of the two scans (on
// random 8-bit keys every second one is mispredicted): the scans only ever stop at "left stoppers" (!comp(x, pivot)) and
// "right stoppers" (!comp(pivot, x)), so one branch-free pass lists both kinds of positions, and the algorithm's k-th swap is
// (k-th left stopper from the left, k-th right stopper from the right) for as long as the former lies left of the latter: between
// the two last swapped positions the array is still the original one, and the elements swapped INTO them are stoppers of the
// right kind, which is where a scan ends when it finds no original stopper first — hence the cut below.
// `idx` = scratch for 2 * (last - first) positions.
template <class T, class Comp>
inline T * partition_pivot_lists(T * first, T * last, Comp comp, uint32_t * idx)
{
  T * mid = first + (last - first) / 2;
  median_to_first(first, first + 1, mid, last - 1, comp);
  const T pivot = *first;
  const uint32_t n = static_cast<uint32_t>(last - first);
  uint32_t * A = idx;      // left stoppers, ascending positions (relative to first), from [1, n)
  uint32_t * B = idx + n;  // right stoppers, ascending positions, from [0, n): position 0 (the pivot itself) is the sentinel
  uint32_t na = 0, nb = 0;
  B[nb++] = 0;
  for (uint32_t p = 1; p < n; ++p) {
    const T x = first[p];
    A[na] = p;
    na += comp(x, pivot) ? 0u : 1u;
    B[nb] = p;
    nb += comp(pivot, x) ? 0u : 1u;
  }
  // k-th right stopper from the right = B[nb - 1 - k]
  uint32_t k = 0;
  while (k < na && k < nb && A[k] < B[nb - 1 - k]) {
    std::iter_swap(first + A[k], first + B[nb - 1 - k]);
    ++k;
  }
  // the scan from the left ends at its next original stopper or, failing one before it, at the position of the last swap's
  // right partner (which now holds a left stopper)
  if (k == 0) return first + A[0];  // (exists: the median of three left an element that is not before the pivot)
  const uint32_t prev_right = B[nb - k];
  const uint32_t next_left = k < na ? A[k] : n;
  return first + (next_left < prev_right ? next_left : prev_right);
}

// one partition step: the list form for long ranges (per-thread scratch), the scanning form for short ones
template <class T, class Comp>
inline T * partition_auto(T * first, T * last, Comp comp)
{
  if (last - first < kListPartitionMin) return partition_pivot(first, last, comp);
  static thread_local std::vector<uint32_t> scratch;
  const size_t need = 2 * static_cast<size_t>(last - first);
  if (scratch.size() < need) scratch.resize(need);
  return partition_pivot_lists(first, last, comp, scratch.data());
}

// the loop of std::sort on [first, last) with `depth` partition levels left; ranges of <= 16 elements stay unsorted
template <class T, class Comp>
inline void introsort_loop(T * first, T * last, long depth, Comp comp)
{
  while (last - first > kThreshold) {
    if (depth == 0) {
      std::partial_sort(first, last, last, comp);  // the library's heap sort, as std::sort itself falls back to
      return;
    }
    --depth;
    T * cut = partition_auto(first, last, comp);
    introsort_loop(cut, last, depth, comp);
    last = cut;
  }
}

template <class T, class Comp>
inline void linear_insert_unguarded(T * last, Comp comp)
{
  T val = std::move(*last);
  T * next = last - 1;
  while (comp(val, *next)) {
    *last = std::move(*next);
    last = next;
    --next;
  }
  *last = std::move(val);
}

template <class T, class Comp>
inline void insertion_sort(T * first, T * last, Comp comp)
{
  if (first == last) return;
  for (T * i = first + 1; i != last; ++i) {
    if (comp(*i, *first)) {
      T val = std::move(*i);
      std::move_backward(first, i, i + 1);
      *first = std::move(val);
    } else {
      linear_insert_unguarded(i, comp);
    }
  }
}

template <class T, class Comp>
inline void final_insertion_sort(T * first, T * last, Comp comp)
{
  if (last - first > kThreshold) {
    insertion_sort(first, first + kThreshold, comp);
    for (T * i = first + kThreshold; i != last; ++i) linear_insert_unguarded(i, comp);
  } else {
    insertion_sort(first, last, comp);
  }
}

// std::sort(first, last, comp), one thread
template <class T, class Comp>
inline void sort_sequential(T * first, T * last, Comp comp)
{
  if (first == last) return;
  introsort_loop(first, last, floor_log2(last - first) * 2, comp);
  final_insertion_sort(first, last, comp);
}

// A few host threads that stay around between sorts (starting three threads costs ~60 us, a fifth of the sort they are
// started for).  One job at a time: a caller that finds the pool busy sorts on its own thread.
class Pool
{
public:
  explicit Pool(int helpers)
  {
    try {
      for (int t = 0; t < helpers; ++t) threads_.emplace_back([this] { loop(); });
    } catch (...) {
      // fewer helpers than asked for: the caller and the ones that started do the work
    }
  }
  ~Pool()
  {
    {
      std::lock_guard<std::mutex> g(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto & t : threads_) t.join();
  }
  Pool(const Pool &) = delete;
  Pool & operator=(const Pool &) = delete;
  int helpers() const { return static_cast<int>(threads_.size()); }
  // runs `fn` on every helper and on the caller, returns when all have returned; false (nothing run) when the pool is busy
  template <class Fn>
  bool run(Fn & fn)
  {
    std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
    if (!job.owns_lock()) return false;
    {
      std::lock_guard<std::mutex> g(mu_);
      call_ = [](void * f) { (*static_cast<Fn *>(f))(); };
      arg_ = &fn;
      pending_ = helpers();
      ++generation_;
    }
    cv_.notify_all();
    fn();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    return true;
  }

private:
  void loop()
  {
    unsigned long seen = 0;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
      if (quit_) return;
      seen = generation_;
      void (*call)(void *) = call_;
      void * arg = arg_;
      lk.unlock();
      call(arg);
      lk.lock();
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::mutex mu_, job_mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;
  void (*call_)(void *) = nullptr;
  void * arg_ = nullptr;
  int pending_ = 0;
  unsigned long generation_ = 0;
  bool quit_ = false;
};

// std::sort(first, last, comp) with the partition tree spread over `threads` host threads (the caller is one of them;
// the others come from `pool` when one is given and free, else they are started for this call).
// Parts larger than `leaf` are split further by whoever takes them; smaller ones are finished in place — including their
// share of the final insertion sort: a partition cut is a border no element crosses in that pass (everything right of it is
// not before anything left of it, so the backward scan of an insertion stops there at the latest), hence the pass over the
// whole array equals the passes over the parts, whoever runs them.
template <class T, class Comp>
inline void sort_parallel(T * first, T * last, Comp comp, int threads, std::ptrdiff_t leaf = 4096, Pool * pool = nullptr)
{
  if (first == last) return;
  if (threads < 2 || last - first <= 2 * leaf) {
    sort_sequential(first, last, comp);
    return;
  }
  struct Part
  {
    T * first;
    T * last;
    long depth;
  };
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Part> queue;
  int busy = 0;  // parts being worked on: the tree is finished when the queue is empty and nobody is busy
  queue.push_back({first, last, floor_log2(last - first) * 2});
  auto worker = [&]() {
    std::unique_lock<std::mutex> lk(mu);
    while (true) {
      while (queue.empty() && busy > 0) cv.wait(lk);
      if (queue.empty()) return;  // busy == 0: done
      Part p = queue.back();
      queue.pop_back();
      ++busy;
      lk.unlock();
      // the introsort loop on this part: big right halves go to the queue instead of the call stack
      bool sorted = false;
      while (p.last - p.first > kThreshold) {
        if (p.last - p.first <= leaf) {
          introsort_loop(p.first, p.last, p.depth, comp);
          break;
        }
        if (p.depth == 0) {
          std::partial_sort(p.first, p.last, p.last, comp);
          sorted = true;  // nothing left for the insertion pass to move
          break;
        }
        --p.depth;
        T * cut = partition_auto(p.first, p.last, comp);
        if (p.last - cut > kThreshold) {
          lk.lock();
          queue.push_back({cut, p.last, p.depth});
          lk.unlock();
          cv.notify_one();
        } else {
          insertion_sort(cut, p.last, comp);  // a right part the loop leaves to the final pass
        }
        p.last = cut;
      }
      if (!sorted) insertion_sort(p.first, p.last, comp);
      lk.lock();
      --busy;
      if (queue.empty() && busy == 0) cv.notify_all();
    }
  };
  if (pool && pool->helpers() > 0 && pool->run(worker)) return;
  std::vector<std::thread> spawned;
  spawned.reserve(static_cast<size_t>(threads));
  try {
    for (int t = 1; t < threads; ++t) spawned.emplace_back(worker);
  } catch (...) {
    // no more threads to be had: the ones that started and the caller finish the tree
  }
  worker();
  for (auto & t : spawned) t.join();
}

}  // namespace exact_sort
}  // namespace mh
