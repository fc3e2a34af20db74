// Internals shared by the translation units that implement the C ABI (mh_api.hip, photo_api.hip): the context,
// error plumbing, the device / pinned allocation cache and the growable device buffer.  Not installed, not part of
// the boundary.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <set>
#include <new>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/mimosa_hip.h"
#include <atomic>

#include "icp_device.hpp"
#include "map_device.hpp"
#include "scan_device.hpp"
#include "shard_device.hpp"

inline thread_local std::string g_mh_err;
constexpr int kMaxPending = 64;
constexpr int kMaxBatch = 64;  // factors per mh_icp_linearize_batch call

struct mh_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = true;  // false: adopted from the caller (mh_init_on_stream)
  std::string err;
  int profiling = 0;  // 0 off; n: HIP events around the kernels of every n-th linearize call of a factor
  hipEvent_t timer[2] = {nullptr, nullptr};
  void * h_batch = nullptr;  // pinned staging of mh_icp_linearize_batch's argument blocks
  void * d_batch = nullptr;  // ... and the device copy the batched kernels read
  void * d_scratch = nullptr;  // stream-ordered scratch of factor creation (source ordering): reused, never freed per call
  size_t d_scratch_cap = 0;
  hipStream_t copy_stream = nullptr;  // mh_scan_prefetch: uploads beside the compute stream (created on first use: an HSA queue costs ~1 ms)
  std::mutex copy_mu;
};

// The stream of the context the calling thread is working for (set by mh_enter at every entry point): what the allocation
// cache orders its hand-overs by.
inline thread_local hipStream_t g_mh_stream = nullptr;
inline hipError_t mh_enter(const mh_ctx * ctx)
{
  g_mh_stream = ctx->stream;
  return hipSetDevice(ctx->device);
}

inline int fail(const mh_ctx * ctx, int code, const std::string & msg)
{
  g_mh_err = msg;
  if (ctx) const_cast<mh_ctx *>(ctx)->err = msg;
  return code;
}
inline int hip_fail(const mh_ctx * ctx, hipError_t e, const char * what)
{
  const int code = (e == hipErrorOutOfMemory) ? MH_ERR_OOM : (e == hipErrorNoDevice ? MH_ERR_NO_DEVICE : MH_ERR_HIP);
  return fail(ctx, code, std::string(what) + ": " + hipGetErrorString(e));
}
// Nothing throws across the C boundary (include/mimosa_hip.h): every extern "C" body runs inside this.
template <typename E>
inline int guarded(const mh_ctx * ctx, const char * what, E && body)
{
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return fail(ctx, MH_ERR_OOM, std::string(what) + ": host allocation failed");
  } catch (const std::exception & e) {
    return fail(ctx, MH_ERR_HIP, std::string(what) + ": " + e.what());
  } catch (...) {
    return fail(ctx, MH_ERR_HIP, std::string(what) + ": unknown exception");
  }
}
#define MH_HIP(ctx, call)                                   \
  do {                                                      \
    const hipError_t e_ = (call);                           \
    if (e_ != hipSuccess) return hip_fail(ctx, e_, #call);  \
  } while (0)

// Device / pinned-host allocation cache.  A scan creates a factor (a dozen device buffers, a pinned result
// ring, sort temporaries) and destroys it a few hundred milliseconds later; hipMalloc / hipFree / hipHostMalloc
// cost 10-200 us each and hipFree synchronises the device.  Freed blocks of 4 KiB .. 64 MiB are kept per
// (device, rounded size) and handed out again; the cache holds at most kMaxCachedBytes per device.
// Hand-over is STREAM-ORDERED, there is no device-wide drain: a block freed while its last user may still be running
// carries an event recorded on the freeing thread's stream (g_mh_stream: every piece of work on the block was enqueued there
// before the free); whoever takes the block next on the same stream simply queues behind, on another stream waits for the
// event on the device.  (Round 2 drained the whole device per free, which stalled the photometric stream behind the
// geometric one and vice versa.)
// MH_ALLOC_CHECK=1 (environment, diagnostic): the rule above is CHECKED.  Every block that goes back to the cache is
// poisoned on the freeing stream (a fill behind everything that stream enqueued on it, in front of the release event), and
// every cached block that is handed out again is verified on the taker's stream behind the event: a word that is not the
// poison pattern any more was written by work that was NOT ordered in front of the free — a user on another stream the
// freeing thread did not wait for.  Reads of that kind see the pattern and fail the parity suites.  Violations are counted
// in mapped host memory (mh_alloc_check_stats) and reported when the last context of the process is shut down.
namespace mh
{
hipError_t launch_alloc_verify(const void * p, size_t bytes, unsigned long long * violations, hipStream_t stream);  // mh_api.hip
}
constexpr unsigned int kAllocPoison = 0xA5C3A5C3u;

class AllocCache
{
public:
  static bool checking()
  {
    static const bool on = [] {
      const char * e = std::getenv("MH_ALLOC_CHECK");
      return e && *e && *e != '0';
    }();
    return on;
  }
  // [0] blocks verified at hand-out, [1] words found overwritten; mapped pinned (the verify kernel adds to [1])
  static unsigned long long * check_counters()
  {
    static unsigned long long * c = [] {
      void * p = nullptr;
      if (hipHostMalloc(&p, 2 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) return static_cast<unsigned long long *>(nullptr);
      std::memset(p, 0, 2 * sizeof(unsigned long long));
      return static_cast<unsigned long long *>(p);
    }();
    return c;
  }
  static hipError_t alloc(void ** out, size_t bytes)
  {
    const size_t cls = size_class(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (cls) {
      Cached c{};
      if (take(dev, cls, true, c)) {
        hand_over(c, dev, cls);
        *out = c.p;
        return hipSuccess;
      }
    }
    static const bool trace = std::getenv("MH_ALLOC_TRACE") != nullptr;  // every trip to the runtime, with its cost, on stderr
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, cls ? cls : bytes);
    if (trace)
      std::fprintf(stderr, "AllocCache: hipMalloc(%zu) %.0f us (stream %p)\n", cls ? cls : bytes,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), static_cast<void *>(g_mh_stream));
    if (e != hipSuccess && cls) {
      // out of memory: a cached block that is still in flight on another stream is better than none
      (void)hipGetLastError();
      Cached c{};
      if (take(dev, cls, false, c)) {
        hand_over(c, dev, cls);
        *out = c.p;
        return hipSuccess;
      }
      return e;
    }
    if (e == hipSuccess && cls) {
      std::lock_guard<std::mutex> g(mu());
      live()[*out] = Block{cls, dev};
    }
    return e;
  }
  // drained: the caller has already waited for every stream that ever touched the block (a factor's or a scan's own
  // buffers after a synchronisation of their context's stream — and of the copy stream where one wrote them): it can be
  // handed out at once
  static void free(void * p, bool drained = false)
  {
    if (!p) return;
    Block b{0, 0};
    {
      std::lock_guard<std::mutex> g(mu());
      auto it = live().find(p);
      if (it != live().end()) {
        b = it->second;
        live().erase(it);
      }
    }
    if (b.cls) {
      Cached c{p, nullptr, nullptr, false};
      int cur = 0;
      (void)hipGetDevice(&cur);
      const bool same_dev = cur == b.dev;
      if (checking() && same_dev && g_mh_stream) {
        // poison behind the freeing stream's work on the block (for a drained block: behind nothing — it is idle)
        c.poisoned = hipMemsetD32Async(p, static_cast<int>(kAllocPoison), b.cls / 4, g_mh_stream) == hipSuccess;
        if (drained && c.poisoned) drained = false;  // the fill itself is in flight now: hand over behind an event
      }
      if (!drained) {
        if (g_mh_stream && same_dev) {  // (an event of one device cannot be recorded on a stream of another)
          {
            std::lock_guard<std::mutex> g(mu());
            auto & pool = event_pool()[b.dev];
            if (!pool.empty()) {
              c.ev = pool.back();
              pool.pop_back();
            }
          }
          if (!c.ev && hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) c.ev = nullptr;
          if (c.ev && hipEventRecord(c.ev, g_mh_stream) == hipSuccess) {
            c.stream = g_mh_stream;
          } else {
            if (c.ev) (void)hipEventDestroy(c.ev);
            c.ev = nullptr;
          }
        }
        if (!c.ev) drain_device(b.dev);  // no stream known for this thread: the old way, on the block's own device
      }
      std::lock_guard<std::mutex> g(mu());
      if (cached_bytes()[b.dev] + b.cls <= kMaxCachedBytes) {
        free_list()[key(b.dev, b.cls)].push_back(c);
        cached_bytes()[b.dev] += b.cls;
        return;
      }
      if (c.ev) event_pool()[b.dev].push_back(c.ev);  // (the block itself goes back to the runtime below: hipFree waits for the device)
    }
    (void)hipFree(p);
  }
  // pinned, mapped result rings (one fixed size)
  static hipError_t alloc_pinned(void ** out, size_t bytes)
  {
    {
      std::lock_guard<std::mutex> g(mu());
      auto & v = pinned()[bytes];
      if (!v.empty()) {
        *out = v.back();
        v.pop_back();
        return hipSuccess;
      }
    }
    return hipHostMalloc(out, bytes, hipHostMallocMapped);
  }
  static void free_pinned(void * p, size_t bytes)
  {
    if (!p) return;
    {
      std::lock_guard<std::mutex> g(mu());
      auto & v = pinned()[bytes];
      if (v.size() < 64) {
        v.push_back(p);
        return;
      }
    }
    (void)hipHostFree(p);
  }
  // context bookkeeping (mh_init / mh_shutdown): per device, under the cache mutex
  static void context_created(int dev)
  {
    std::lock_guard<std::mutex> g(mu());
    contexts_of()[dev]++;
  }
  // the device's last context gives its cached blocks back to the runtime (the pinned ones go when no context is left on any
  // device)
  static void context_destroyed(int dev)
  {
    bool last_of_device = false;
    {
      std::lock_guard<std::mutex> g(mu());
      int & n = contexts_of()[dev];
      if (n > 0) --n;
      last_of_device = n == 0;
    }
    if (last_of_device) trim(dev);
  }
  static void trim(int dev)
  {
    std::vector<void *> dead, dead_pinned;
    bool none_left = false;
    {
      std::lock_guard<std::mutex> g(mu());
      for (auto & kv : free_list())
        if (static_cast<int>(kv.first >> 56) == dev) {
          for (const Cached & c : kv.second) {
            dead.push_back(c.p);
            if (c.ev) event_pool()[dev].push_back(c.ev);
          }
          kv.second.clear();
        }
      cached_bytes()[dev] = 0;
      int total = 0;
      for (const auto & kv : contexts_of()) total += kv.second;
      none_left = total == 0;
      if (none_left)
        for (auto & kv : pinned()) {
          dead_pinned.insert(dead_pinned.end(), kv.second.begin(), kv.second.end());
          kv.second.clear();
        }
    }
    for (void * p : dead) (void)hipFree(p);
    for (void * p : dead_pinned) (void)hipHostFree(p);
    if (none_left && checking() && check_counters()) {
      (void)hipDeviceSynchronize();
      std::fprintf(stderr, "MH_ALLOC_CHECK: %llu cached blocks verified at hand-out, %llu overwritten words found\n", check_counters()[0], check_counters()[1]);
      if (check_counters()[1]) std::abort();  // diagnostic mode: a process that broke the rule does not exit quietly
    }
  }

private:
  static std::unordered_map<int, int> & contexts_of()
  {
    static std::unordered_map<int, int> m;
    return m;
  }
  struct Block
  {
    size_t cls;
    int dev;
  };
  struct Cached
  {
    void * p;
    hipStream_t stream;  // where the free was ordered; null with ev == null: nobody is using the block
    hipEvent_t ev;
    bool poisoned;       // MH_ALLOC_CHECK: filled with kAllocPoison behind its last use
  };
  static void drain_device(int dev)
  {
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    if (cur != dev) (void)hipSetDevice(cur);
  }
  // a cached block of the class, preferring (prefer_idle) one nobody can still be using or whose last use was ordered on
  // THIS stream: taking another stream's block makes this stream wait (on the device) for that stream to get there
  static bool take(int dev, size_t cls, bool prefer_idle, Cached & c)
  {
    std::lock_guard<std::mutex> g(mu());
    auto & v = free_list()[key(dev, cls)];
    if (v.empty()) return false;
    size_t pick = v.size() - 1;
    if (prefer_idle)
      for (size_t i = v.size(); i-- > 0;)
        if (!v[i].ev || v[i].stream == g_mh_stream) {
          pick = i;
          break;
        }
    c = v[pick];
    v[pick] = v.back();
    v.pop_back();
    cached_bytes()[dev] -= cls;
    live()[c.p] = Block{cls, dev};
    return true;
  }
  // order the taker behind the block's release; a wait that cannot be set up (a stale stream after mh_shutdown, an event the
  // runtime rejects) falls back to draining the block's device — never hands the block out unordered
  static void hand_over(Cached & c, int dev, size_t cls)
  {
    if (c.ev) {
      hipError_t e = hipSuccess;
      if (!g_mh_stream)
        e = hipEventSynchronize(c.ev);
      else if (c.stream != g_mh_stream)
        e = hipStreamWaitEvent(g_mh_stream, c.ev, 0);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (hipEventSynchronize(c.ev) != hipSuccess) {
          (void)hipGetLastError();
          drain_device(dev);
        }
      }
      std::lock_guard<std::mutex> g(mu());
      event_pool()[dev].push_back(c.ev);
      c.ev = nullptr;
    }
    if (c.poisoned && checking() && check_counters()) {
      if (g_mh_stream) {
        __atomic_fetch_add(&check_counters()[0], 1ull, __ATOMIC_RELAXED);
        (void)mh::launch_alloc_verify(c.p, cls, check_counters() + 1, g_mh_stream);
      }
      c.poisoned = false;
    }
  }
  // events are per DEVICE: one recorded on another device's stream fails, and every free would fall back to draining the
  // device
  static std::unordered_map<int, std::vector<hipEvent_t>> & event_pool()
  {
    static std::unordered_map<int, std::vector<hipEvent_t>> v;
    return v;
  }
  static constexpr size_t kMaxCachedBytes = size_t(2) << 30;
  static size_t size_class(size_t bytes)
  {
    if (bytes > (size_t(64) << 20)) return 0;  // big (map-sized) blocks are not cached
    size_t c = 4096;
    while (c < bytes) c += c / 2 >= 4096 ? (c / 4) : c;  // 4K, 8K, 16K, 20K, 25K, ... (~25 % steps)
    return c;
  }
  static uint64_t key(int dev, size_t cls) { return (static_cast<uint64_t>(dev) << 56) | cls; }
  static std::mutex & mu()
  {
    static std::mutex m;
    return m;
  }
  static std::unordered_map<uint64_t, std::vector<Cached>> & free_list()
  {
    static std::unordered_map<uint64_t, std::vector<Cached>> m;
    return m;
  }
  static std::unordered_map<void *, Block> & live()
  {
    static std::unordered_map<void *, Block> m;
    return m;
  }
  static std::unordered_map<int, size_t> & cached_bytes()
  {
    static std::unordered_map<int, size_t> m;
    return m;
  }
  static std::unordered_map<size_t, std::vector<void *>> & pinned()
  {
    static std::unordered_map<size_t, std::vector<void *>> m;
    return m;
  }
};

template <typename T>
inline hipError_t dev_alloc(T ** out, size_t bytes)
{
  void * p = nullptr;
  const hipError_t e = AllocCache::alloc(&p, bytes);
  *out = static_cast<T *>(p);
  return e;
}
inline void dev_free(void * p) { AllocCache::free(p); }

// Scoped device temporary: released (back to the cache) on every exit path, including the early returns of MH_HIP.
template <typename T>
struct DevTemp
{
  T * p = nullptr;
  DevTemp() = default;
  DevTemp(const DevTemp &) = delete;
  DevTemp & operator=(const DevTemp &) = delete;
  ~DevTemp() { dev_free(p); }
  hipError_t alloc(size_t bytes) { return dev_alloc(&p, bytes ? bytes : 16); }
  operator T *() const { return p; }
};

// Growable device buffer
struct DevBuf
{
  void * p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes, hipStream_t stream, bool keep)
  {
    if (bytes <= cap) return hipSuccess;
    size_t ncap = cap ? cap : 4096;
    while (ncap < bytes) ncap += ncap / 2 + 4096;
    void * np = nullptr;
    hipError_t e = AllocCache::alloc(&np, ncap);
    if (e != hipSuccess) return e;
    if (keep && p && cap) {
      e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) {
        AllocCache::free(np);
        return e;
      }
    }
    if (p) AllocCache::free(p);
    p = np;
    cap = ncap;
    return hipSuccess;
  }
  void release(bool drained = false)
  {
    if (p) AllocCache::free(p, drained);
    p = nullptr;
    cap = 0;
  }
};

// device-resident scan front end (mh_scan_*): state of one scan
struct mh_scan
{
  mh_ctx * ctx;
  DevBuf d_raw, d_full, d_geo_idx, d_unique, d_body, d_ds, d_kept_idx, d_counters, d_rt;
  DevBuf d_full_raw;  // points_raw_ (lidar/manager.cpp:376-380): points_full_ as it was before deskewing, kept for the photometric path
  bool keep_raw = false, raw_valid = false;
  DevBuf d_sensor;       // other sensors (mh_scan_prepare_input_layout): raw records, organise-by-ring scratch
  DevBuf d_prep, d_vox;  // scratch of launch_prepare_input / launch_preprocess (scan_device.hpp: prepare_layout, voxel_layout)
  mh::ScanCounters * h_c = nullptr;  // pinned landing buffer: the device counters, then the first kUniqueCached distinct timestamps
  static constexpr size_t kUniqueCached = 4096;
  size_t n_unique_cached = 0;        // how many of unique_ns_ sit behind h_c (0 = ask the device)
  mh::ScanCounters c{};
  size_t n_in = 0, n_body = 0;
  bool prepared = false, preprocessed = false;
  // mh_scan_prefetch: the NEXT cloud staged (pinned buffer -> d_raw on a copy stream of its own) while another scan is processed
  hipEvent_t copy_done = nullptr;
  hipEvent_t compute_mark = nullptr;  // recorded on the compute stream at every prefetch: the copy stream queues behind it
  void * h_stage = nullptr;
  size_t h_stage_cap = 0, n_prefetched = 0;
  bool prefetch_valid = false;
  // mh_scan_deskew: the poses travel through a pinned block of the scan's own, so the call neither touches the caller's
  // (pageable) buffer from the stream nor waits for it
  void * h_rt = nullptr;
  size_t h_rt_cap = 0;
  hipEvent_t rt_done = nullptr;
};

// IncrementalVoxelMapPCL counterpart: the device-resident voxel map (map_device.hpp / map_kernels.hip).  The device
// arrays ARE the map; the host keeps counters only (refreshed from the mapped state after every mutation).
struct mh_map
{
  mh_ctx * ctx = nullptr;
  std::atomic<int> refs{1};
  mh_map_config cfg{};
  double inv_leaf = 0, min_sq = 0;
  DevBuf d_table, d_cells, d_buckets, d_qbuckets, d_vox, d_lru;
  size_t table_cap = 0, vox_cap = 0, block_cap = 0;  // capacities in entries (table slots, voxels, blocks)
  mh::MapState * d_state = nullptr;                  // device counters the kernels maintain
  mh::MapState * h_state = nullptr;                  // pinned host copy
  uint32_t n_voxels = 0, n_blocks = 0;
  uint64_t n_points = 0, lru_counter = 0;
  // insert scratch (grown on demand, reused)
  DevBuf s_in, s_pts, s_group, s_seg_vid, s_seg_added, s_blk_new, s_flags, s_pos, s_temp, s_rt, s_shard;
  void * h_in = nullptr;  // pinned staging of a host batch
  size_t h_in_cap = 0;
  int64_t inserts = 0, upload_bytes = 0, purges = 0;
  int n_off = 0;
  int8_t off[27][3];
  // contexts (= HIP streams) that hold factors on this map: a mutation waits for THEIR streams, not for the whole device
  std::mutex readers_mu;
  std::vector<std::pair<mh_ctx *, int>> readers;
  bool poisoned = false;  // a mutation failed half way (LRU purge): the device arrays and the counters disagree; every later call fails
};

inline void map_add_reader(mh_map * m, mh_ctx * c)
{
  std::lock_guard<std::mutex> g(m->readers_mu);
  for (auto & r : m->readers)
    if (r.first == c) {
      ++r.second;
      return;
    }
  m->readers.emplace_back(c, 1);
}
inline void map_remove_reader(mh_map * m, mh_ctx * c)
{
  std::lock_guard<std::mutex> g(m->readers_mu);
  for (auto & r : m->readers)
    if (r.first == c && r.second > 0) --r.second;
}
// Before a mutation: wait for the streams of the OTHER contexts that hold factors on this map (their kernels may be reading
// it); work on the map's own stream is ordered by the stream.  (Round 2 drained the whole device here.)
inline hipError_t map_wait_readers(mh_map * m)
{
  std::vector<hipStream_t> streams;
  {
    std::lock_guard<std::mutex> g(m->readers_mu);
    for (const auto & r : m->readers)
      if (r.second > 0 && r.first != m->ctx) streams.push_back(r.first->stream);
  }
  for (hipStream_t st : streams) {
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

inline mh::MapView map_view(const mh_map * m)
{
  mh::MapView v;
  v.table = static_cast<const int4 *>(m->d_table.p);
  v.cells = static_cast<const uint32_t *>(m->d_cells.p);
  v.buckets = static_cast<const float4 *>(m->d_buckets.p);
  v.qbuckets = static_cast<const uint4 *>(m->d_qbuckets.p);
  v.inv_leaf = m->inv_leaf;
  v.mask = static_cast<uint32_t>(m->table_cap - 1);
  v.n_off = m->n_off;
  v.mode_idx = m->n_off == 1 ? 0 : (m->n_off == 7 ? 1 : (m->n_off == 19 ? 2 : 3));
  return v;
}

// ---- ICP factor handle (mh_api.hip; shared with shard_api.hip) --------------------------------------------------
struct PendingCall
{
  mh_icp_result * out;
  double R[9];     // delta rotation (for reg_4_dof's local_z)
  double gz[3];    // global_z = -g_unit
  int parity;
  int linearize_count;
  unsigned int seq;  // what the call's last kernel publishes to the host slot when it is complete (0: nothing was launched)
  bool components;   // K4 ran for this call: loc_*_comp / status_hist are meaningful
  bool seq_has_basis = false;  // ... and wrote the eigenbases it projected on into the call's result slot
  int loc_blocks = 0;          // plain factors: K4's workgroups of this call = rows of flagged words the host folds
  bool launched_k4 = false;    // plain factors: the call's sums come from K4's workgroup 0 (else from K3's last block)
  hipEvent_t ev[3];
};

struct mh_icp
{
  mh_ctx * ctx;
  mh_map * map;
  size_t n;
  mh_reg_config cfg;
  bool binary;
  DevBuf d_src, d_qda, d_mean, d_normal, d_status, d_partials, d_ticket, d_result, d_dbg, d_perm, d_eig;
  // plain factors: what K4 reads of a call is the record K3 wrote for it (icp_device.hpp: IcpArgs::rec) and K3's partial rows;
  // K4 follows its K3 on the context's stream, so one of each serves every call of the factor
  DevBuf d_rec;
  bool ordered = false;  // d_src / per-point state are in Morton order, d_perm maps back
  mh::DeviceResult * h_results = nullptr;    // pinned, mapped ring: the device writes results here directly (two-phase callers)
  mh::DeviceResult * d_h_results = nullptr;  // its device-side address
  uint4 * h_ll = nullptr;     // pinned, mapped ring of flagged-word slots (icp_device.hpp): what a plain call's kernels publish
  uint4 * d_h_ll = nullptr;   // its device-side address
  size_t ll_words = 0;        // 16-byte words per slot
  PendingCall pending[kMaxPending];
  int n_pending = 0;
  int parity = 0;
  bool cold = true;
  bool components = true;  // mh_icp_set_components: run K4 (component localizabilities + status histogram) in every linearize
  int linearize_count = 0;
  hipEvent_t events[kMaxPending][3];
  bool events_ready = false;
  unsigned int seq_counter = 0;
  // map-sharded use (mh_icp_shard_*): points migrate between ranks with their association state
  bool no_order = false;  // keep the caller's point order (no Morton re-ordering)
  size_t cap_n = 0;       // per-point arrays are sized for this many points (>= n): room for arrivals of the sharded path
  DevBuf d_origin, x_src, x_qda, x_mean, x_normal, x_status, x_origin;  // origin ids; the second set of arrays pack compacts into
  DevBuf s_keys_a, s_keys_b, s_idx_a, s_idx_b, s_counts, s_temp, d_sums;
  uint32_t * h_counts = nullptr;  // pinned
  bool origin_ready = false, plan_open = false;
  uint32_t n_movers = 0;
};

// mh_api.hip internals used by shard_api.hip
namespace mhi
{
void pose_delta(const double * Rs, const double * ts, const double * Rt, const double * tt, double * R, double * t);
void finish(const mh_icp * icp, const mh::DeviceResult & d, const PendingCall & pc, mh_icp_result * out);
int icp_create(mh_ctx * ctx, mh_map * map, const mh_point32 * source, const mh_point32 * d_source, size_t n, size_t capacity,
               const mh_reg_config * cfg, int is_binary, mh_icp ** out, bool no_order);
// argument blocks of one linearize call in pending slot n_pending (claimed); no launch
int prepare(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt, const double * t_tgt, const double g_unit[3],
            mh_icp_result * out, bool want_flag, mh::IcpArgs & a, mh::LocArgs & l);
// shard_api.hip: called by mh_shutdown(ctx) so that communicators whose rounds ran on ctx let go of it
void shard_ctx_gone(mh_ctx * ctx);
}  // namespace mhi
