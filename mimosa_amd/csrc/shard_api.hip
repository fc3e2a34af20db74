// Native map-sharded scan-to-map factor (BASELINE configs[2]): C ABI `mh_shard_*` of include/mimosa_hip.h.
//
// No reference counterpart — the reference is single-process; what must hold is that the sharded factor equals the
// unsharded ICPFactor::linearize (include/mimosa/lidar/geometric_factor.hpp:231-562) point for point.  One process per
// GPU.  The map is partitioned into shard blocks of 2^block_log2 voxels per axis, block b owned by
// XORVector3iHash(b) mod world (include/mimosa/lidar/utils.hpp:228-238); every rank also stores the one-voxel halo of its
// blocks (mh_map_insert_shard), so the 1/7/19/27 neighbourhood of a query in an owned block is complete locally, and
// every source point is linearized on the owner of the centre voxel of its CURRENT position.
//
// One linearize = one chain of enqueues on the factor's stream and ONE wait at its end:
//   route    owner of every local point at this pose; movers' 112-byte records (point + data-association state) into
//            fixed-size per-peer segments [count | records], tombstones behind them                    (2 kernels)
//   C1       all-to-all of the segments: RCCL ncclAllToAll over xGMI — fixed size, so no counts cross the host
//   append   arrivals behind the last slot; the slot count stays on the device                         (1 kernel)
//   K3       icp_linearize_kernel over the slots (n read from the device, tombstones skipped); its last block
//            writes the Hessian sums into the all-reduce vector
//   C3a      all-reduce(sum) of that vector: sums, counters, and one slot per rank carrying its mover maximum
//   K4       component localizabilities in the eigenbasis of the GLOBAL sums (only when asked for) + C3b all-reduce
//   publish  results + this rank's counters to mapped host memory, completion flag                     (1 kernel)
// The segment capacity adapts to the traffic (a cold first call moves (P-1)/P of the cloud, later calls a few points that
// crossed a block face); a call whose movers did not fit — a global fact, every rank reads the same maxima — is repeated
// with larger segments: points that were processed hit their data-association cache, the rest is sent and processed.
// With one rank and no forced collectives there is nothing to exchange and the call IS mh_icp_linearize.
//
// Throughput forms (graph::Manager re-linearizes EVERY live factor per update, src/graph/manager.cpp:585-588): a protocol
// ROUND carries any number of factors — their movers interleaved per peer in ONE send buffer and ONE ncclAllToAll, one
// route / append launch each (the factors' argument blocks ride in the kernel-argument segment), one K3b (+ K4b) launch
// per kernel instantiation, ONE ncclAllReduce of B x 168 doubles (+ one of B x 16), one publish launch — and up to
// kShardRing rounds are in flight per communicator: mh_shard_icp_linearize_async / _batch_async enqueue, mh_shard_icp_wait
// completes them in order.  mh_shard_icp_linearize is a round of one that is completed at once.
//
// Transports: RCCL (librccl resolved at run time, so the library has no link-time dependency on it) and an in-process
// group (ranks = host threads of one process on one device) that the tests use to run the protocol at world > 1 on a
// one-GPU box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <vector>

#include "../../include/mimosa_hip.h"
#include "icp_device.hpp"
#include "mh_internal.hpp"
#include "shard_device.hpp"

namespace
{
// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------
struct RcclApi
{
  void * lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllToAll) AllToAll = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;  // optional
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // optional
  std::string err;
};
RcclApi & rccl()
{
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already in the process (PyTorch ships its own) is the one to use: two RCCL instances in one
    // process would each bring up their own transports
    const char * env = std::getenv("MH_RCCL_LIB");
    const char * names[] = {"librccl.so", "librccl.so.1"};
    if (env && *env) api.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (const char * n : names)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char * load[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char * n : load)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) {
      api.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
      return;
    }
#define MH_SYM(field, name)                                              \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name)); \
  if (!api.field) api.err += std::string(" missing ") + name;
    MH_SYM(GetUniqueId, "ncclGetUniqueId")
    MH_SYM(CommInitRank, "ncclCommInitRank")
    MH_SYM(CommDestroy, "ncclCommDestroy")
    MH_SYM(AllReduce, "ncclAllReduce")
    MH_SYM(AllToAll, "ncclAllToAll")
    MH_SYM(GetErrorString, "ncclGetErrorString")
    MH_SYM(GetVersion, "ncclGetVersion")
    api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(dlsym(api.lib, "ncclCommGetAsyncError"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
#undef MH_SYM
  });
  return api;
}

// ---- in-process group: ranks are host threads of one process (test transport) ----------------------------------------
struct LocalGroup
{
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long gen = 0;
  int refs = 0;
  std::vector<const char *> send_ptr;
  std::vector<std::vector<double>> ar_host;
  bool broken = false;  // a rank gave up waiting: every later barrier fails at once instead of hanging the others
  std::vector<char> here;  // who stands at the current barrier
  std::string missing;     // set by the rank that gave up: the ranks it was waiting for
  // false: a peer did not arrive within the time limit (it failed, or a collective was not entered by all ranks)
  bool barrier(int rank)
  {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    if (here.size() != static_cast<size_t>(world)) here.assign(static_cast<size_t>(world), 0);
    const unsigned long long g = gen;
    here[static_cast<size_t>(rank)] = 1;
    if (++arrived == world) {
      arrived = 0;
      ++gen;
      std::fill(here.begin(), here.end(), 0);
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || broken; }) || broken) {
      if (!broken) {
        missing.clear();
        for (int r = 0; r < world; ++r)
          if (!here[static_cast<size_t>(r)]) missing += (missing.empty() ? "" : ", ") + std::to_string(r);
      }
      broken = true;
      cv.notify_all();
      return false;
    }
    return true;
  }
};
}  // namespace

struct mh_shard_icp;
namespace
{
constexpr int kShardRing = 32;  // protocol rounds in flight per communicator
// one factor's part of a protocol round
struct ShardCall
{
  mh_shard_icp * S = nullptr;  // null: the factor was destroyed while the round was in flight (its result is dropped)
  mh_icp_result * out = nullptr;
  PendingCall pc{};
  uint32_t cap = 0;             // per-peer segment capacity of this call
  uint32_t arrivals_bound = 0;  // what the call added to the factor's slot bound
  double R_src[9], t_src[3], R_tgt[9], t_tgt[3], g_unit[3];  // kept for a repeat (segment overflow)
  bool has_tgt = false;
  double carried[4] = {0, 0, 0, 0};  // k-NN counters of earlier attempts of this call (their points hit the cache in the repeat)
  int attempts = 0;
  unsigned long long round_index = 0;  // (repair list) the round the call overflowed in
  bool cold = false;                   // the call was the first after mh_shard_icp_reset: a repeat of it starts from zero state again
};
// A segment-capacity decision read out of a completed round (complete_front).  Rounds complete at rank-local times (a rank
// that needs an idle factor drains early), but the capacity sizes the NEXT round's exchange layout, which every rank must
// agree on: the decision is parked here and takes effect at the next point all ranks reach with the same history
// (apply_cap_updates: settle_rounds for the rounds it covers, run_repairs once everything has been completed).
struct CapUpdate
{
  unsigned long long round_index;
  mh_shard_icp * S;
  uint32_t value;
  bool grow_only;  // an overflow: never below the current capacity
};
struct ShardRound
{
  std::vector<ShardCall> calls;
  mh::ShardPublish * h_pub = nullptr;  // calls.size() entries of the communicator's publish ring
  unsigned int seq = 0;
  unsigned long long index = 0;  // position in the communicator's sequence of rounds: the same number on every rank
  bool any_components = false;
};
}  // namespace

struct mh_shard_comm
{
  int world = 1, rank = 0;
  bool is_rccl = false;
  int device = 0;
  ncclComm_t nccl = nullptr;
  LocalGroup * grp = nullptr;
  std::string err;
  long long n_all_to_all = 0, n_all_reduce = 0;
  // protocol rounds in flight (FIFO: one stream, so they complete in order) and what they share: the exchange buffers are
  // reused in stream order, the publish slots form a ring in mapped pinned memory (kShardRing x ring_width entries)
  mh_ctx * ws_ctx = nullptr;
  std::deque<ShardRound> rounds;
  DevBuf ws_send, ws_recv, ws_ar, ws_loc;
  mh::ShardPublish * h_ring = nullptr;
  mh::ShardPublish * d_h_ring = nullptr;
  size_t ring_width = 0;
  int ring_pos = 0;
  unsigned int seq = 0;
  unsigned long long n_rounds = 0;   // rounds enqueued so far
  std::vector<ShardCall> repairs;    // calls whose segments overflowed, waiting to be repeated (settle_rounds)
  std::vector<CapUpdate> cap_updates;  // in round order (rounds complete in order)
  bool repairing = false;              // run_repairs is on the stack: its own enqueues must not start another one
  std::vector<mh_shard_icp *> plain_pending;  // one rank, no protocol: the factors with mh_icp_linearize_async calls to collect
  std::vector<mh_shard_icp *> factors;        // every live factor of this communicator (a communicator destroyed first orphans them)

  int peer_missing()
  {
    err = "in-process group: a rank did not reach the collective within 120 s";
    {
      std::lock_guard<std::mutex> g(grp->mu);
      if (!grp->missing.empty()) err += " (rank " + std::to_string(rank) + " waited; missing: " + grp->missing + ")";
    }
    g_mh_err = err;
    return MH_ERR_HIP;
  }
  int nccl_fail(ncclResult_t r, const char * what)
  {
    err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
    g_mh_err = err;
    return MH_ERR_HIP;
  }
  // every rank sends every rank `bytes` (segment p of `send` goes to rank p, lands as segment `rank` of its `recv`)
  int all_to_all(const void * send, void * recv, size_t bytes, hipStream_t stream)
  {
    ++n_all_to_all;
    if (is_rccl) {
      const ncclResult_t r = rccl().AllToAll(send, recv, bytes, ncclInt8, nccl, stream);
      return r == ncclSuccess ? MH_OK : nccl_fail(r, "ncclAllToAll");
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return MH_ERR_HIP;
    grp->send_ptr[rank] = static_cast<const char *>(send);
    if (!grp->barrier(rank)) return peer_missing();
    hipError_t e = hipSuccess;
    for (int p = 0; p < world && e == hipSuccess; ++p)
      e = hipMemcpyAsync(static_cast<char *>(recv) + static_cast<size_t>(p) * bytes, grp->send_ptr[p] + static_cast<size_t>(rank) * bytes, bytes,
                         hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (!grp->barrier(rank)) return peer_missing();  // nobody refills a send buffer a peer is still reading
    return e == hipSuccess ? MH_OK : MH_ERR_HIP;
  }
  int all_reduce(double * buf, size_t n, hipStream_t stream)
  {
    ++n_all_reduce;
    if (is_rccl) {
      const ncclResult_t r = rccl().AllReduce(buf, buf, n, ncclDouble, ncclSum, nccl, stream);
      return r == ncclSuccess ? MH_OK : nccl_fail(r, "ncclAllReduce");
    }
    std::vector<double> & mine = grp->ar_host[rank];
    mine.resize(n);
    if (hipMemcpyAsync(mine.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) return MH_ERR_HIP;
    if (hipStreamSynchronize(stream) != hipSuccess) return MH_ERR_HIP;
    if (!grp->barrier(rank)) return peer_missing();
    std::vector<double> sum(n, 0.0);
    for (int p = 0; p < world; ++p)  // rank order on every rank: identical bits everywhere
      for (size_t i = 0; i < n; ++i) sum[i] += grp->ar_host[p][i];
    if (!grp->barrier(rank)) return peer_missing();
    if (hipMemcpyAsync(buf, sum.data(), n * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) return MH_ERR_HIP;
    return hipStreamSynchronize(stream) == hipSuccess ? MH_OK : MH_ERR_HIP;
  }
};

struct mh_shard_icp
{
  mh_ctx * ctx = nullptr;
  mh_shard_comm * comm = nullptr;
  mh_icp * icp = nullptr;
  bool collective = false;
  int block_log2 = 3;
  // device
  mh::ShardState * d_state = nullptr;
  DevBuf d_dest, d_hist, d_ar, d_flags, d_pos, d_temp;
  // host knowledge.  n_slots / n_live are exact as of the last COMPLETED call (the publish kernel reports the counters);
  // slots_bound is an upper bound of the slot count behind every call enqueued so far (what sizes grids while calls are in
  // flight); cur is the ping-pong entry the NEXT enqueued call reads
  int cur = 0;
  uint32_t n_slots = 0, n_live = 0, slot_capacity = 0, slots_bound = 0;
  int inflight = 0;  // calls enqueued and not completed
  bool cold_pending = false;  // mh_shard_icp_reset since the last enqueued call: the next one reads the association state as zero
  bool replay = false;  // a call overflowed its segments and waits to be repeated: the calls made after it are repeated behind it
  uint32_t seg_cap = 0, seg_cap_max = 0;
  uint64_t n_total = 0;
  bool broken = false;  // a collective call failed half way (records sent, slots tombstoned): the factor's state is not to be trusted
  bool ctx_gone = false;  // mh_shutdown of the factor's context ran first (shard_ctx_gone): ctx and icp are null, the handle can only be destroyed
  int last_linearize_count = 0;  // ... what mh_shard_icp_stats still reports then
  mh_shard_stats stats{};
};

namespace
{
int ensure_ring(mh_ctx * ctx, mh_shard_comm * comm, size_t width);
mh::ShardArrays arrays_of(mh_icp * icp, bool alt)
{
  mh::ShardArrays a;
  a.src = static_cast<float4 *>((alt ? icp->x_src : icp->d_src).p);
  a.q_da = static_cast<double *>((alt ? icp->x_qda : icp->d_qda).p);
  a.mean = static_cast<double *>((alt ? icp->x_mean : icp->d_mean).p);
  a.normal = static_cast<double *>((alt ? icp->x_normal : icp->d_normal).p);
  a.status = static_cast<int32_t *>((alt ? icp->x_status : icp->d_status).p);
  a.origin = static_cast<unsigned long long *>((alt ? icp->x_origin : icp->d_origin).p);
  return a;
}
uint32_t pow2_at_least(uint32_t v)
{
  uint32_t p = 1;
  while (p < v && p < 0x40000000u) p <<= 1;
  return p;
}
// A dead peer is an error, not a hang: past 50 ms the wait polls the stream and RCCL's asynchronous error state, and gives
// up altogether after MH_SHARD_TIMEOUT_S seconds (default 120, like the in-process group's barrier).
double shard_timeout_s()
{
  static const double v = [] {
    const char * e = std::getenv("MH_SHARD_TIMEOUT_S");
    const double x = e ? std::atof(e) : 0.0;
    return x > 0.0 ? x : 120.0;
  }();
  return v;
}
int wait_publish(mh_ctx * ctx, mh_shard_comm * comm, const unsigned int * flag_word, unsigned int seq)
{
  const volatile unsigned int * flag = flag_word;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  auto elapsed_ns = [&]() {
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec);
  };
  for (unsigned spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 1023u) != 1023u) continue;
    if (elapsed_ns() < 50000000L) continue;
    // 50 ms: something is slow (a first call's channel set-up) or wrong (a peer is gone).  Stop burning the core; poll the
    // stream and, over RCCL, the communicator's asynchronous error state
    for (;;) {
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) MH_HIP(ctx, q);
      if (comm->is_rccl && rccl().CommGetAsyncError) {
        ncclResult_t async = ncclSuccess;
        const ncclResult_t r = rccl().CommGetAsyncError(comm->nccl, &async);
        if (r != ncclSuccess) return comm->nccl_fail(r, "ncclCommGetAsyncError");
        if (async != ncclSuccess && async != ncclInProgress) return comm->nccl_fail(async, "a collective failed asynchronously");
      }
      if (static_cast<double>(elapsed_ns()) * 1e-9 > shard_timeout_s()) {
        comm->err = "a collective call did not complete within " + std::to_string(static_cast<int>(shard_timeout_s())) + " s (a peer is missing?)";
        g_mh_err = comm->err;
        return MH_ERR_HIP;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
      comm->err = "the stream drained without the completion flag";
      g_mh_err = comm->err;
      return MH_ERR_HIP;
    }
    break;
  }
  return MH_OK;
}

int shard_compact(mh_shard_icp * S)
{
  mh_ctx * ctx = S->ctx;
  mh_icp * icp = S->icp;
  const size_t k = S->slot_capacity ? S->slot_capacity : 1;
  MH_HIP(ctx, icp->x_src.reserve(k * sizeof(float4), ctx->stream, false));
  MH_HIP(ctx, icp->x_qda.reserve(k * 3 * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, icp->x_mean.reserve(k * 3 * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, icp->x_normal.reserve(k * 3 * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, icp->x_status.reserve(k * sizeof(int32_t), ctx->stream, false));
  MH_HIP(ctx, icp->x_origin.reserve(k * sizeof(unsigned long long), ctx->stream, false));
  MH_HIP(ctx, S->d_flags.reserve(k * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, S->d_pos.reserve(k * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, S->d_temp.reserve(mh::shard_temp_bytes(k), ctx->stream, false));
  // (only ever called on an idle factor: the host knows n_slots / n_live exactly)
  MH_HIP(ctx, mh::launch_shard_compact(arrays_of(icp, false), arrays_of(icp, true), S->d_state, S->cur, S->n_slots, static_cast<uint32_t *>(S->d_flags.p),
                                       static_cast<uint32_t *>(S->d_pos.p), S->d_temp.p, S->d_temp.cap, ctx->stream));
  if (S->n_slots) {
    std::swap(icp->d_src, icp->x_src);
    std::swap(icp->d_qda, icp->x_qda);
    std::swap(icp->d_mean, icp->x_mean);
    std::swap(icp->d_normal, icp->x_normal);
    std::swap(icp->d_status, icp->x_status);
    std::swap(icp->d_origin, icp->x_origin);
    S->n_slots = S->n_live;  // the host has known n_live exactly since the last call
    S->slots_bound = S->n_slots;
  }
  S->stats.compactions_total++;
  return MH_OK;
}

// the global result from the all-reduced vector: localizabilities of the GLOBAL H (geometric_factor.hpp:405-411), Schur
// degeneracy info, 4-DoF projection and the degeneracy branch (:413-428, :464-557) — once, on the global sums
void global_result(mh_shard_icp * S, const mh::ShardPublish & p, const PendingCall & pc, bool components, mh_icp_result * out)
{
  const int nent = S->icp->binary ? 91 : 28;
  mh::DeviceResult d;
  std::memset(&d, 0, sizeof(d));
  for (int i = 0; i < nent; ++i) d.sums[i] = p.ar[i];
  d.n_knn = static_cast<unsigned long long>(p.ar[nent]);
  d.n_cand = static_cast<unsigned long long>(p.ar[nent + 1]);
  d.n_fallback = static_cast<unsigned long long>(p.ar[nent + 2]);
  d.n_scanned = static_cast<unsigned long long>(p.ar[nent + 3]);
  for (int i = 0; i < 6; ++i) d.loc_comp[i] = p.loc[i];
  for (int i = 0; i < 9; ++i) d.status_hist[i] = static_cast<unsigned int>(p.loc[6 + i]);
  PendingCall c = pc;
  c.components = components;
  mhi::finish(S->icp, d, c, out);
  out->gpu_ms_linearize = out->gpu_ms_localizability = -1.0f;
}
}  // namespace

// Every live communicator, so that a context that goes away first (mh_shutdown) can take back what the communicator holds on it.
static std::mutex g_comm_mu;
static std::vector<mh_shard_comm *> g_comms;

static void comm_register(mh_shard_comm * c)
{
  std::lock_guard<std::mutex> g(g_comm_mu);
  g_comms.push_back(c);
}

// Detach a communicator from the context its rounds run on: rounds nobody waited for are dropped (their factors can
// only be destroyed afterwards), the exchange buffers and the publish ring go back once the stream has drained.
static void comm_release_ctx(mh_shard_comm * comm)
{
  if (!comm->ws_ctx) return;
  (void)mh_enter(comm->ws_ctx);
  (void)hipStreamSynchronize(comm->ws_ctx->stream);
  for (ShardRound & r : comm->rounds)
    for (ShardCall & c : r.calls)
      if (c.S) {
        c.S->inflight = 0;
        c.S->broken = true;
        c.S->comm = nullptr;
      }
  comm->rounds.clear();
  comm->cap_updates.clear();
  for (DevBuf * b : {&comm->ws_send, &comm->ws_recv, &comm->ws_ar, &comm->ws_loc}) b->release(true);
  if (comm->h_ring) AllocCache::free_pinned(comm->h_ring, sizeof(mh::ShardPublish) * comm->ring_width * kShardRing);
  comm->h_ring = comm->d_h_ring = nullptr;
  comm->ring_width = 0;
  comm->ws_ctx = nullptr;
}

// Everything of a factor that lives on its context: device buffers, the inner plain factor.  The context must still be alive.
static void shard_release_device(mh_shard_icp * S)
{
  for (DevBuf * b : {&S->d_dest, &S->d_hist, &S->d_ar, &S->d_flags, &S->d_pos, &S->d_temp}) b->release(true);
  if (S->d_state) AllocCache::free(S->d_state, true);
  S->d_state = nullptr;
  if (S->icp) {
    S->last_linearize_count = S->icp->linearize_count;
    mh_icp_destroy(S->icp);
  }
  S->icp = nullptr;
}

namespace mhi
{
// mh_shutdown(ctx), before the stream goes: communicators whose rounds ran on ctx let go of it (a later
// mh_shard_comm_destroy must not touch a dead context; a later mh_shard_icp_create may bind a new one), and the sharded
// factors of ctx give their device memory back NOW, while the context is alive — afterwards such a handle holds no pointer
// into the dead context (ctx, icp null; ctx_gone set): every entry point refuses it and mh_shard_icp_destroy just deletes it.
void shard_ctx_gone(mh_ctx * ctx)
{
  std::lock_guard<std::mutex> g(g_comm_mu);
  for (mh_shard_comm * c : g_comms) {
    bool any = c->ws_ctx == ctx;
    for (mh_shard_icp * S : c->factors) any = any || S->ctx == ctx;
    if (!any) continue;
    if (c->ws_ctx == ctx) comm_release_ctx(c);  // (synchronises the stream; drops the rounds)
    for (mh_shard_icp * S : c->factors)
      if (S->ctx == ctx) {
        S->broken = true;
        S->inflight = 0;
        for (ShardCall & q : c->repairs)
          if (q.S == S) q.S = nullptr;
        for (CapUpdate & u : c->cap_updates)
          if (u.S == S) u.S = nullptr;
        auto & pend = c->plain_pending;
        pend.erase(std::remove(pend.begin(), pend.end(), S), pend.end());
        (void)mh_enter(ctx);
        (void)hipStreamSynchronize(ctx->stream);
        shard_release_device(S);
        S->ctx = nullptr;
        S->ctx_gone = true;
      }
  }
}
}  // namespace mhi

// every entry point but destroy: a factor whose context was shut down first
#define MH_SHARD_ALIVE(S, who)                                                                                                      \
  do {                                                                                                                              \
    if ((S) && (S)->ctx_gone) return fail(nullptr, MH_ERR_INVALID_ARG, std::string(who) + ": the factor's context was shut down; the handle can only be destroyed"); \
  } while (0)

extern "C" {

int mh_shard_unique_id(void * id128)
{
  return guarded(nullptr, "mh_shard_unique_id", [&]() -> int {
    if (!id128) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_unique_id: NULL argument");
    RcclApi & api = rccl();
    if (!api.lib || !api.err.empty()) return fail(nullptr, MH_ERR_UNSUPPORTED, "mh_shard_unique_id: " + api.err);
    static_assert(sizeof(ncclUniqueId) == MH_SHARD_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, MH_ERR_HIP, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
    std::memcpy(id128, &id, sizeof(id));
    return MH_OK;
  });
}

int mh_shard_comm_init_rccl(mh_ctx * ctx, const void * id128, int world, int rank, mh_shard_comm ** out)
{
  return guarded(ctx, "mh_shard_comm_init_rccl", [&]() -> int {
    if (!ctx || !id128 || !out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_comm_init_rccl: NULL argument");
    *out = nullptr;
    if (world < 1 || world > mh::kShardMaxWorld || rank < 0 || rank >= world)
      return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_comm_init_rccl: world in 1..64, 0 <= rank < world");
    RcclApi & api = rccl();
    if (!api.lib || !api.err.empty()) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_shard_comm_init_rccl: " + api.err);
    MH_HIP(ctx, mh_enter(ctx));
    mh_shard_comm * c = new mh_shard_comm;
    c->world = world;
    c->rank = rank;
    c->is_rccl = true;
    c->device = ctx->device;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = api.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
      const std::string msg = std::string("ncclCommInitRank: ") + api.GetErrorString(r);
      delete c;
      return fail(ctx, MH_ERR_HIP, msg);
    }
    comm_register(c);
    *out = c;
    return MH_OK;
  });
}

int mh_shard_comm_init_local(int world, mh_shard_comm ** out_array)
{
  return guarded(nullptr, "mh_shard_comm_init_local", [&]() -> int {
    if (!out_array) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_comm_init_local: NULL argument");
    if (world < 1 || world > mh::kShardMaxWorld) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_comm_init_local: world in 1..64");
    LocalGroup * g = new LocalGroup;
    g->world = world;
    g->refs = world;
    g->send_ptr.assign(world, nullptr);
    g->ar_host.resize(world);
    for (int r = 0; r < world; ++r) {
      mh_shard_comm * c = new mh_shard_comm;
      c->world = world;
      c->rank = r;
      c->grp = g;
      comm_register(c);
      out_array[r] = c;
    }
    return MH_OK;
  });
}

void mh_shard_comm_destroy(mh_shard_comm * comm)
{
  if (!comm) return;
  {
    std::lock_guard<std::mutex> g(g_comm_mu);
    g_comms.erase(std::remove(g_comms.begin(), g_comms.end(), comm), g_comms.end());
  }
  comm_release_ctx(comm);
  for (mh_shard_icp * S : comm->factors) {  // factors that outlive their communicator can only be destroyed
    S->comm = nullptr;
    S->broken = true;
  }
  if (comm->is_rccl && comm->nccl) {
    (void)hipSetDevice(comm->device);
    (void)rccl().CommDestroy(comm->nccl);
  }
  if (comm->grp) {
    bool last = false;
    {
      std::lock_guard<std::mutex> g(comm->grp->mu);
      last = --comm->grp->refs == 0;
    }
    if (last) delete comm->grp;
  }
  delete comm;
}

int mh_shard_comm_world(const mh_shard_comm * comm) { return comm ? comm->world : 0; }
int mh_shard_comm_rank(const mh_shard_comm * comm) { return comm ? comm->rank : -1; }
const char * mh_shard_comm_backend(const mh_shard_comm * comm) { return !comm ? "" : (comm->is_rccl ? "rccl" : "local"); }
int mh_shard_comm_info(const mh_shard_comm * comm, int * ranks_in_communicator, int * rccl_version)
{
  if (!comm) return MH_ERR_INVALID_ARG;
  int n = comm->world, v = 0;
  if (comm->is_rccl) {
    // what RCCL itself says about the communicator (not what the caller passed in): a driver can see that N ranks really joined
    RcclApi & api = rccl();
    if (api.CommCount && comm->nccl && api.CommCount(comm->nccl, &n) != ncclSuccess) n = -1;
    if (api.GetVersion && api.GetVersion(&v) != ncclSuccess) v = 0;
  }
  if (ranks_in_communicator) *ranks_in_communicator = n;
  if (rccl_version) *rccl_version = v;
  return MH_OK;
}

void mh_shard_icp_destroy(mh_shard_icp * S)
{
  if (!S) return;
  if (S->ctx) {
    (void)mh_enter(S->ctx);
    (void)hipStreamSynchronize(S->ctx->stream);
  }
  if (S->comm) {
    auto & all = S->comm->factors;
    all.erase(std::remove(all.begin(), all.end(), S), all.end());
  }
  if (S->comm) {  // calls nobody waited for: the rounds stay (other factors' results are in them), this factor's part is dropped
    for (ShardRound & r : S->comm->rounds)
      for (ShardCall & c : r.calls)
        if (c.S == S) c.S = nullptr;
    for (ShardCall & c : S->comm->repairs)
      if (c.S == S) c.S = nullptr;
    // capacity decisions parked for this factor (complete_front) must not be applied to freed memory later
    for (CapUpdate & u : S->comm->cap_updates)
      if (u.S == S) u.S = nullptr;
    auto & pend = S->comm->plain_pending;
    pend.erase(std::remove(pend.begin(), pend.end(), S), pend.end());
  }
  if (S->ctx) shard_release_device(S);  // (a factor whose context went first gave everything back in shard_ctx_gone)
  delete S;
}

static int shard_icp_create_impl(mh_ctx * ctx, mh_shard_comm * comm, mh_map * map, const mh_point32 * points, size_t n_local, int points_on_device,
                                 const mh_reg_config * cfg, int is_binary, const mh_shard_config * scfg, mh_shard_icp ** out)
{
  if (!ctx || !comm || !map || !cfg || !out || (!points && n_local)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_create: NULL argument");
  *out = nullptr;
  const int log2 = scfg ? scfg->block_log2 : 3;
  if (log2 < 0 || log2 > 10) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_create: block_log2 in 0..10");
  if (n_local > 0x1fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_shard_icp_create: cloud too large");
  if (comm->is_rccl && comm->device != ctx->device) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_create: communicator lives on another device");
  MH_HIP(ctx, mh_enter(ctx));
  mh_shard_icp * S = new mh_shard_icp;
  S->ctx = ctx;
  S->comm = comm;
  comm->factors.push_back(S);
  S->block_log2 = log2;
  S->collective = comm->world > 1 || (scfg && scfg->force_collectives);
  S->stats.world = comm->world;
  S->stats.rank = comm->rank;
  auto bail = [&](int rc) {
    mh_shard_icp_destroy(S);
    return rc;
  };
  if (!S->collective) {
    // one rank, nothing to exchange: the plain factor (Morton-ordered source, cold first call without state reads)
    const int rc = mhi::icp_create(ctx, map, points_on_device ? nullptr : points, points_on_device ? points : nullptr, n_local, 0, cfg, is_binary, &S->icp, false);
    if (rc != MH_OK) return bail(rc);
    S->n_slots = S->n_live = static_cast<uint32_t>(n_local);
    S->n_total = n_local;
    S->slot_capacity = static_cast<uint32_t>(n_local);
    *out = S;
    return MH_OK;
  }
  // every rank learns every rank's share: one slot per rank in an all-reduce vector (creation is collective)
  MH_HIP(ctx, S->d_ar.reserve(mh::kShardArLen * sizeof(double), ctx->stream, false));
  std::vector<double> share(mh::kShardArLen, 0.0);
  share[mh::kShardSums + comm->rank] = static_cast<double>(n_local);
  MH_HIP(ctx, hipMemcpyAsync(S->d_ar.p, share.data(), share.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int rc = comm->all_reduce(static_cast<double *>(S->d_ar.p), mh::kShardArLen, ctx->stream);
  if (rc != MH_OK) return bail(fail(ctx, rc, "mh_shard_icp_create: all-reduce of the shares failed: " + comm->err));
  MH_HIP(ctx, hipMemcpyAsync(share.data(), S->d_ar.p, share.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t total = 0, biggest = 0;
  for (int r = 0; r < comm->world; ++r) {
    const uint64_t v = static_cast<uint64_t>(share[mh::kShardSums + r]);
    total += v;
    biggest = std::max(biggest, v);
  }
  if (2 * total + 4096 > 0x3fffffffu) return bail(fail(ctx, MH_ERR_UNSUPPORTED, "mh_shard_icp_create: cloud too large"));
  S->n_total = total;
  // a rank may end up owning the whole cloud; tombstones take slots until the next compaction
  S->slot_capacity = static_cast<uint32_t>(2 * total + 4096);
  S->seg_cap_max = static_cast<uint32_t>(std::max<uint64_t>(total, 1));
  S->seg_cap = static_cast<uint32_t>(std::max<uint64_t>(biggest, 64));  // the first call may move a rank's whole share to one peer
  rc = mhi::icp_create(ctx, map, points_on_device ? nullptr : points, points_on_device ? points : nullptr, n_local, S->slot_capacity, cfg, is_binary, &S->icp, true);
  if (rc != MH_OK) return bail(rc);
  mh_icp * icp = S->icp;
  icp->cold = false;  // the state arrays are explicit (zeroed at creation): tombstones and migrated state live in them
  MH_HIP(ctx, icp->d_origin.reserve(static_cast<size_t>(S->slot_capacity) * sizeof(unsigned long long), ctx->stream, false));
  MH_HIP(ctx, mh::launch_shard_origin(static_cast<unsigned long long *>(icp->d_origin.p), static_cast<uint32_t>(n_local), static_cast<uint32_t>(comm->rank), ctx->stream));
  void * st = nullptr;
  MH_HIP(ctx, AllocCache::alloc(&st, sizeof(mh::ShardState)));
  S->d_state = static_cast<mh::ShardState *>(st);
  MH_HIP(ctx, mh::launch_shard_state_init(S->d_state, static_cast<uint32_t>(n_local), ctx->stream));
  if (!comm->rounds.empty() && comm->ws_ctx != ctx) return bail(fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_create: the communicator has rounds in flight on another context"));
  if (comm->rounds.empty()) {
    comm->ws_ctx = ctx;
    rc = ensure_ring(ctx, comm, 1);
    if (rc != MH_OK) return bail(rc);
  }
  MH_HIP(ctx, hipMemsetAsync(S->d_ar.p, 0, mh::kShardArLen * sizeof(double), ctx->stream));
  const size_t slots_rounded = (static_cast<size_t>(S->slot_capacity) + 255) & ~size_t(255);
  MH_HIP(ctx, S->d_dest.reserve(slots_rounded, ctx->stream, false));
  MH_HIP(ctx, S->d_hist.reserve((slots_rounded / 256 + 1) * comm->world * sizeof(uint32_t), ctx->stream, false));
  // partial rows for the largest grid a call can use
  MH_HIP(ctx, icp->d_partials.reserve(static_cast<size_t>(mh::linearize_grid_max(static_cast<int>(S->slot_capacity))) * mh::kPartialStride * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  S->n_slots = S->n_live = S->slots_bound = static_cast<uint32_t>(n_local);
  *out = S;
  return MH_OK;
}
int mh_shard_icp_create(mh_ctx * ctx, mh_shard_comm * comm, mh_map * map, const mh_point32 * points, size_t n_local, int points_on_device,
                        const mh_reg_config * cfg, int is_binary, const mh_shard_config * scfg, mh_shard_icp ** out)
{
  return guarded(ctx, "mh_shard_icp_create", [&]() -> int { return shard_icp_create_impl(ctx, comm, map, points, n_local, points_on_device, cfg, is_binary, scfg, out); });
}

// ---- protocol rounds ------------------------------------------------------------------------------------------------
namespace
{
// what a call can add to the slot count at most: a full segment from every peer, never more than the points of all ranks
uint32_t arrivals_bound_of(const mh_shard_icp * S, uint32_t world, uint32_t cap)
{
  return static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(world - 1) * cap, S->n_total));
}
// ... at an IDLE factor the host knows n_live exactly: never more than the points held elsewhere
uint64_t arrivals_max_idle(const mh_shard_icp * S, uint32_t world, uint32_t cap)
{
  return std::min<uint64_t>(static_cast<uint64_t>(world - 1) * cap, S->n_total - std::min<uint64_t>(S->n_total, S->n_live));
}
// once a round failed nothing in flight can be trusted: every factor with a pending call is in an unknown state
int fail_rounds(mh_shard_comm * comm, int rc)
{
  for (ShardRound & r : comm->rounds)
    for (ShardCall & c : r.calls)
      if (c.S) {
        c.S->broken = true;
        c.S->inflight = 0;
      }
  for (ShardCall & c : comm->repairs)
    if (c.S) c.S->broken = true;
  comm->rounds.clear();
  comm->repairs.clear();
  return rc;
}

int ensure_ring(mh_ctx * ctx, mh_shard_comm * comm, size_t width)  // only with no round in flight
{
  if (width <= comm->ring_width) return MH_OK;
  if (comm->h_ring) AllocCache::free_pinned(comm->h_ring, sizeof(mh::ShardPublish) * comm->ring_width * kShardRing);
  comm->h_ring = comm->d_h_ring = nullptr;
  comm->ring_width = 0;
  size_t w = 1;
  while (w < width) w *= 2;
  MH_HIP(ctx, AllocCache::alloc_pinned(reinterpret_cast<void **>(&comm->h_ring), sizeof(mh::ShardPublish) * w * kShardRing));
  std::memset(comm->h_ring, 0, sizeof(mh::ShardPublish) * w * kShardRing);
  MH_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&comm->d_h_ring), comm->h_ring, 0));
  comm->ring_width = w;
  comm->ring_pos = 0;
  return MH_OK;
}

struct RoundSpec  // what the caller asked for, one entry per factor
{
  mh_shard_icp * S;
  const double *R_src, *t_src, *R_tgt, *t_tgt, *g_unit;
  mh_icp_result * out;
};

// Complete the OLDEST round in flight: wait for its publication, bring every factor's host-side counters up to date,
// assemble the results.  A call whose movers did not fit its segments has processed only part of its points; that is a
// GLOBAL fact (every rank reads the same per-rank maxima out of the all-reduced vector), and the call goes on the
// communicator's repair list — repeated later, at a point every rank reaches with the same list (settle_rounds).
int complete_front(mh_shard_comm * comm)
{
  ShardRound & r = comm->rounds.front();
  mh_ctx * ctx = comm->ws_ctx;
  const uint32_t world = static_cast<uint32_t>(comm->world);
  for (size_t f = 0; f < r.calls.size(); ++f) {
    const int rc = wait_publish(ctx, comm, &r.h_pub[f].seq, r.seq);
    if (rc != MH_OK) return fail_rounds(comm, fail(ctx, rc, "mh_shard_icp_wait: " + comm->err));
  }
  int rc_all = MH_OK;
  std::vector<mh_shard_icp *> touched;
  for (size_t f = 0; f < r.calls.size(); ++f) {
    ShardCall & c = r.calls[f];
    mh_shard_icp * S = c.S;
    if (!S) continue;
    touched.push_back(S);
    const mh::ShardPublish & p = r.h_pub[f];
    const int nent = S->icp->binary ? 91 : 28;
    S->n_slots = p.n_slots;
    S->n_live = p.n_live;
    S->inflight--;
    if (p.error) {
      S->broken = true;
      rc_all = fail(ctx, MH_ERR_UNSUPPORTED, "mh_shard_icp_linearize: arrivals exceeded the slot capacity");
      continue;
    }
    uint32_t max_movers = 0;
    for (uint32_t q = 0; q < world; ++q) max_movers = std::max(max_movers, static_cast<uint32_t>(p.ar[mh::kShardSums + q]));
    S->stats.last_max_movers = max_movers;
    const bool overflow = max_movers > c.cap;
    if (overflow || (S->replay && c.attempts == 0)) {
      // some rank could not send everything: larger segments, same pose — processed points hit their data-association
      // cache in the repeat, the held-back ones travel then; the k-NN counters of this attempt are carried over.  Calls
      // of the factor that were enqueued BEHIND an overflowed one are repeated behind its repeat, whether they fitted or
      // not: the calls of a factor take effect — results, association cache — in the order they were made.
      if (overflow) {
        comm->cap_updates.push_back({r.index, S, std::min(S->seg_cap_max, pow2_at_least(max_movers)), true});
        S->stats.retries_total++;
        S->stats.retries_last++;
        S->replay = true;
      }
      ShardCall again = c;
      // (a cold call is repeated cold: every point is associated again, nothing of the failed attempt is carried over)
      for (int i = 0; i < 4; ++i) again.carried[i] = (overflow && !c.cold) ? c.carried[i] + p.ar[nent + i] : 0.0;
      again.round_index = r.index;
      comm->repairs.push_back(again);
      continue;
    }
    // next call: room for four times what moved now (a pose step of centimetres moves a few points across block faces)
    comm->cap_updates.push_back({r.index, S, std::min(S->seg_cap_max, std::max<uint32_t>(256u, pow2_at_least(4u * max_movers))), false});
    if (c.out) {
      mh::ShardPublish q = p;
      for (int i = 0; i < 4; ++i) q.ar[nent + i] += c.carried[i];
      global_result(S, q, c.pc, c.pc.components, c.out);
    }
  }
  comm->rounds.pop_front();
  // slot bounds of the factors of this round: the exact count + what their calls still in flight may add
  for (mh_shard_icp * S : touched) {
    uint64_t b = S->n_slots;
    for (const ShardRound & q : comm->rounds)
      for (const ShardCall & c : q.calls)
        if (c.S == S) b += c.arrivals_bound;
    S->slots_bound = static_cast<uint32_t>(std::min<uint64_t>(b, S->slot_capacity));
    S->icp->n = S->slots_bound;
  }
  return rc_all;
}

// The parked capacity decisions of rounds up to `upto` take effect, in round order.  Only called where every rank has completed
// exactly the same rounds <= upto (the decisions are read from all-reduced values, so the lists agree).
void apply_cap_updates(mh_shard_comm * comm, unsigned long long upto)
{
  size_t done = 0;
  for (const CapUpdate & u : comm->cap_updates) {
    if (u.round_index > upto) break;
    ++done;
    if (!u.S) continue;  // the factor was destroyed meanwhile
    u.S->seg_cap = u.grow_only ? std::max(u.S->seg_cap, u.value) : u.value;
  }
  comm->cap_updates.erase(comm->cap_updates.begin(), comm->cap_updates.begin() + static_cast<long>(done));
}

int drain_rounds(mh_shard_comm * comm)  // local: completes what is in flight, repeats nothing
{
  int rc_all = MH_OK;
  while (!comm->rounds.empty()) {
    const int rc = complete_front(comm);
    if (rc != MH_OK) rc_all = rc;
  }
  return rc_all;
}

int enqueue_round(const RoundSpec * spec, size_t B);

// The repeats of calls whose segments overflowed: one factor at a time, synchronously, with growing segments until the
// movers fit.  Every rank runs the same list in the same order (the overflow is read from all-reduced values).  A repeat
// reports the ORIGINAL call's linearize count and does not count as a linearize itself.
int run_repairs(mh_shard_comm * comm)
{
  struct Busy
  {
    mh_shard_comm * c;
    explicit Busy(mh_shard_comm * cc) : c(cc) { c->repairing = true; }
    ~Busy() { c->repairing = false; }
  } busy(comm);
  int rc_all = drain_rounds(comm);
  apply_cap_updates(comm, ~0ull);  // every rank is here with everything completed: all parked decisions take effect
  std::vector<mh_shard_icp *> seen;
  struct Clear
  {
    std::vector<mh_shard_icp *> & v;
    mh_shard_comm * comm;
    ~Clear()
    {
      for (mh_shard_icp * S : v) {  // (a factor destroyed meanwhile is not on any list any more: skip what is not ours)
        bool listed = false;
        for (const ShardCall & c : comm->repairs) listed = listed || c.S == S;
        if (!listed) S->replay = false;
      }
    }
  } clear{seen, comm};
  while (!comm->repairs.empty()) {
    ShardCall c = comm->repairs.front();
    comm->repairs.erase(comm->repairs.begin());
    mh_shard_icp * S = c.S;
    if (!S || S->broken) continue;
    if (std::find(seen.begin(), seen.end(), S) == seen.end()) seen.push_back(S);
    if (c.attempts >= 8) {
      S->broken = true;
      rc_all = fail(S->ctx, MH_ERR_HIP, "mh_shard_icp_linearize: segment capacity did not converge");
      continue;
    }
    const int saved_count = S->icp->linearize_count;
    const uint32_t retries_last = S->stats.retries_last;
    RoundSpec sp{S, c.R_src, c.t_src, c.has_tgt ? c.R_tgt : nullptr, c.has_tgt ? c.t_tgt : nullptr, c.g_unit, c.out};
    // a reset that a later call may have announced meanwhile belongs to THAT call (still ahead): set aside over the repeat
    const bool later_reset = S->cold_pending;
    S->cold_pending = c.cold;
    int rc = enqueue_round(&sp, 1);
    S->cold_pending = later_reset;
    if (rc != MH_OK) {
      rc_all = rc;
      continue;
    }
    S->icp->linearize_count = saved_count;
    S->stats.retries_last = retries_last;
    ShardCall & q = comm->rounds.back().calls[0];
    q.pc.linearize_count = c.pc.linearize_count;
    for (int i = 0; i < 4; ++i) q.carried[i] = c.carried[i];
    q.attempts = c.attempts + 1;
    const size_t listed = comm->repairs.size();
    rc = drain_rounds(comm);  // a further overflow puts the call back on the list, with this attempt's counters added ...
    if (rc != MH_OK) rc_all = rc;
    apply_cap_updates(comm, ~0ull);
    // ... and it is repeated at once, BEFORE the repeats of later calls: the calls of a factor end in the order they were made
    if (comm->repairs.size() > listed) std::rotate(comm->repairs.begin(), comm->repairs.begin() + static_cast<long>(listed), comm->repairs.end());
  }
  return rc_all;
}

// A point of the program every rank reaches with the same history (a blocking call or wait; the enqueue of round number n,
// for the rounds that left the ring by then): rounds up to `upto` are completed, and if one of them overflowed, everything
// in flight is completed and the repeats run.  Because overflow is a global fact and the points are defined by the call
// sequence alone, all ranks enter the repeats' collectives in the same order — whatever each rank completed earlier for
// local reasons (a compaction, a growing publish ring).
int settle_rounds(mh_shard_comm * comm, unsigned long long upto)
{
  int rc_all = MH_OK;
  while (!comm->rounds.empty() && comm->rounds.front().index <= upto) {
    const int rc = complete_front(comm);
    if (rc != MH_OK) rc_all = rc;
  }
  apply_cap_updates(comm, upto);
  bool due = false;
  for (const ShardCall & c : comm->repairs) due = due || c.round_index <= upto;
  if (due && !comm->repairing) {  // (the repeats' own enqueues pass through here: the list is already being worked off)
    const int rc = run_repairs(comm);
    if (rc != MH_OK) rc_all = rc;
  }
  return rc_all;
}

// Enqueue ONE protocol round over the given factors (same communicator, same context).  Nothing is waited for unless the
// ring is full, a factor needs compacting or the publish ring must grow (then rounds in flight are completed first).
int enqueue_round(const RoundSpec * spec, size_t B)
{
  mh_shard_icp * S0 = spec[0].S;
  mh_ctx * ctx = S0->ctx;
  mh_shard_comm * comm = S0->comm;
  MH_HIP(ctx, mh_enter(ctx));
  if (comm->ws_ctx && comm->ws_ctx != ctx && !comm->rounds.empty())
    return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_linearize: the communicator has rounds in flight on another context");
  comm->ws_ctx = ctx;
  const uint32_t world = static_cast<uint32_t>(comm->world), rank = static_cast<uint32_t>(comm->rank);
  const unsigned long long index = comm->n_rounds;
  const long long coll_before = comm->n_all_to_all + comm->n_all_reduce;
  if (index >= static_cast<unsigned long long>(kShardRing)) {
    const int rc = settle_rounds(comm, index - kShardRing);
    if (rc != MH_OK) return rc;
  }
  for (size_t f = 0; f < B; ++f)
    if (spec[f].S->broken) return fail(ctx, MH_ERR_HIP, "mh_shard_icp_linearize: an earlier call on this factor failed half way; create a new factor");
  // tombstones are dropped when the arrivals of this call might not fit behind them, or when they outnumber the points
  // (after a compaction n_slots == n_live, and n_live + everything held elsewhere == n_total always fits).  Compaction
  // needs the exact counts, i.e. an idle factor: a local matter (no collective), so each rank decides for itself.
  auto wants_compaction = [&](const mh_shard_icp * S) {
    return S->n_slots + arrivals_max_idle(S, world, S->seg_cap) > S->slot_capacity || S->n_slots > S->n_live + S->n_live / 2 + 16384;
  };
  bool need_idle = B > comm->ring_width;
  for (size_t f = 0; f < B && !need_idle; ++f) {
    const mh_shard_icp * S = spec[f].S;
    need_idle = S->inflight ? static_cast<uint64_t>(S->slots_bound) + arrivals_bound_of(S, world, S->seg_cap) > S->slot_capacity : wants_compaction(S);
  }
  if (need_idle) {
    int rc = drain_rounds(comm);
    if (rc != MH_OK) return rc;
    rc = ensure_ring(ctx, comm, B);
    if (rc != MH_OK) return rc;
    for (size_t f = 0; f < B; ++f)
      if (wants_compaction(spec[f].S)) {
        rc = shard_compact(spec[f].S);
        if (rc != MH_OK) return rc;
      }
  }
  struct Guard  // anything that leaves this function other than by success has left points in flight or tombstoned
  {
    const RoundSpec * spec;
    size_t B;
    bool ok = false;
    ~Guard()
    {
      if (!ok)
        for (size_t f = 0; f < B; ++f) spec[f].S->broken = true;
    }
  } guard{spec, B};

  // ---- layout of the round's exchange buffers: peer p's block = the factors' segments one after the other
  std::vector<size_t> seg_off(B);
  size_t peer_stride = 0;
  for (size_t f = 0; f < B; ++f) {
    seg_off[f] = peer_stride;
    peer_stride += mh::shard_segment_bytes(spec[f].S->seg_cap);
  }
  MH_HIP(ctx, comm->ws_send.reserve(peer_stride * world, ctx->stream, false));
  MH_HIP(ctx, comm->ws_recv.reserve(peer_stride * world, ctx->stream, false));
  {
    const size_t ar_bytes = B * mh::kShardArLen * sizeof(double), loc_bytes = B * 16 * sizeof(double);
    if (comm->ws_ar.cap < ar_bytes) {  // entries no kernel writes (the padding between the sums and the per-rank slots) stay zero
      MH_HIP(ctx, comm->ws_ar.reserve(ar_bytes, ctx->stream, false));
      MH_HIP(ctx, hipMemsetAsync(comm->ws_ar.p, 0, comm->ws_ar.cap, ctx->stream));
    }
    if (comm->ws_loc.cap < loc_bytes) {
      MH_HIP(ctx, comm->ws_loc.reserve(loc_bytes, ctx->stream, false));
      MH_HIP(ctx, hipMemsetAsync(comm->ws_loc.p, 0, comm->ws_loc.cap, ctx->stream));
    }
  }
  double * ar = static_cast<double *>(comm->ws_ar.p);
  double * loc = static_cast<double *>(comm->ws_loc.p);

  ShardRound round;
  round.index = index;
  round.calls.resize(B);
  std::vector<mh::ShardFactorArgs> fa(B);
  std::vector<uint32_t> route_bound(B);
  for (size_t f = 0; f < B; ++f) {
    mh_shard_icp * S = spec[f].S;
    mh_icp * icp = S->icp;
    ShardCall & c = round.calls[f];
    c.S = S;
    c.out = spec[f].out;
    c.cap = S->seg_cap;
    c.arrivals_bound = S->inflight ? arrivals_bound_of(S, world, c.cap) : static_cast<uint32_t>(arrivals_max_idle(S, world, c.cap));
    std::memcpy(c.R_src, spec[f].R_src, sizeof(c.R_src));
    std::memcpy(c.t_src, spec[f].t_src, sizeof(c.t_src));
    std::memcpy(c.g_unit, spec[f].g_unit, sizeof(c.g_unit));
    c.has_tgt = icp->binary;
    if (c.has_tgt) {
      std::memcpy(c.R_tgt, spec[f].R_tgt, sizeof(c.R_tgt));
      std::memcpy(c.t_tgt, spec[f].t_tgt, sizeof(c.t_tgt));
    }
    mh::ShardFactorArgs & a = fa[f];
    mhi::pose_delta(c.R_src, c.t_src, c.has_tgt ? c.R_tgt : nullptr, c.has_tgt ? c.t_tgt : nullptr, a.P.R, a.P.t);
    a.a = arrays_of(icp, false);
    a.st = S->d_state;
    a.inv_leaf = 1.0 / icp->map->cfg.leaf_size;
    a.dest = static_cast<uint8_t *>(S->d_dest.p);
    a.hist = static_cast<uint32_t *>(S->d_hist.p);
    a.send = static_cast<char *>(comm->ws_send.p) + seg_off[f];
    a.recv = static_cast<const char *>(comm->ws_recv.p) + seg_off[f];
    a.peer_stride = peer_stride;
    a.ar_slots = ar + f * mh::kShardArLen + mh::kShardSums;
    a.cap = c.cap;
    a.slot_capacity = S->slot_capacity;
    a.cur = S->cur;
    a.log2 = S->block_log2;
    route_bound[f] = S->slots_bound;
  }
  auto batch_of = [&](size_t f0) {
    mh::ShardBatch blk;
    std::memset(static_cast<void *>(&blk), 0, sizeof(blk));
    blk.n = static_cast<int>(std::min<size_t>(mh::kShardBatchMax, B - f0));
    blk.world = world;
    blk.rank = rank;
    for (int i = 0; i < blk.n; ++i) blk.f[i] = fa[f0 + i];
    mh::shard_batch_grids(blk, route_bound.data() + f0);
    return blk;
  };
  // ---- route -> C1 -> append
  if (B == 1) {
    MH_HIP(ctx, mh::launch_shard_route(fa[0], route_bound[0], world, rank, ctx->stream));
  } else {
    for (size_t f0 = 0; f0 < B; f0 += mh::kShardBatchMax) MH_HIP(ctx, mh::launch_shard_route_batch(batch_of(f0), ctx->stream));
  }
  int rc = comm->all_to_all(comm->ws_send.p, comm->ws_recv.p, peer_stride, ctx->stream);  // C1
  if (rc != MH_OK) return fail(ctx, rc, "mh_shard_icp_linearize: all-to-all failed: " + comm->err);
  if (B == 1) {
    MH_HIP(ctx, mh::launch_shard_append(fa[0], world, ctx->stream));
  } else {
    for (size_t f0 = 0; f0 < B; f0 += mh::kShardBatchMax) MH_HIP(ctx, mh::launch_shard_append_batch(batch_of(f0), ctx->stream));
  }
  // ---- K3 over the slots: the grid covers what the slot count can be at most, the kernel reads the count itself
  std::vector<mh::IcpArgs> ia(B);
  std::vector<mh::LocArgs> la(B);
  static thread_local mh_icp_result scratch;  // filled by nobody: the results are assembled from the global sums
  for (size_t f = 0; f < B; ++f) {
    mh_shard_icp * S = spec[f].S;
    mh_icp * icp = S->icp;
    ShardCall & c = round.calls[f];
    const uint32_t bound = static_cast<uint32_t>(std::min<uint64_t>(S->slot_capacity, static_cast<uint64_t>(S->slots_bound) + c.arrivals_bound));
    icp->n = bound ? bound : 1;
    rc = mhi::prepare(icp, c.R_src, c.t_src, c.has_tgt ? c.R_tgt : nullptr, c.has_tgt ? c.t_tgt : nullptr, c.g_unit, &scratch, false, ia[f], la[f]);
    if (rc != MH_OK) return rc;
    c.pc = icp->pending[0];
    icp->n_pending = 0;  // the pending slot is not used: the call lives in the round
    mh::IcpArgs & a = ia[f];
    mh::LocArgs & l = la[f];
    a.n_dev = l.n_dev = &S->d_state->n_slots[S->cur ^ 1];
    a.cold = S->cold_pending ? 1 : 0;  // (K3 still reads the status words: tombstones and held-back movers are marked there)
    round.calls[f].cold = S->cold_pending;
    S->cold_pending = false;
    a.host_result = nullptr;
    a.seq = 0;
    a.shard_out = ar + f * mh::kShardArLen;
    l.host_result = nullptr;
    l.seq = 0;
    l.sums = ar + f * mh::kShardArLen;  // the eigenbases of the GLOBAL H_rr / H_tt
    l.shard_out = loc + f * 16;
    round.any_components = round.any_components || c.pc.components;
  }
  // launch groups of the batched form: the factors that share a kernel instantiation (as mh_icp_linearize_batch), at most
  // kBatchInline per launch — the argument blocks ride in the kernel-argument segment, nothing is staged
  struct Group
  {
    int tpb, k, n_off;
    bool binary;
    std::vector<size_t> members;
  };
  std::vector<Group> groups;
  if (B > 1) {
    for (size_t f = 0; f < B; ++f) {
      const mh_icp * icp = spec[f].S->icp;
      const int tpb = mh::linearize_class(ia[f].n, ia[f].k, true), k = icp->cfg.num_corres_points == 5 ? 5 : 8, n_off = icp->map->n_off;
      Group * g = nullptr;
      for (Group & q : groups)
        if (q.tpb == tpb && q.k == k && q.n_off == n_off && q.binary == icp->binary && static_cast<int>(q.members.size()) < mh::kBatchInline) g = &q;
      if (!g) {
        groups.push_back(Group{tpb, k, n_off, icp->binary, {}});
        g = &groups.back();
      }
      g->members.push_back(f);
    }
  }
  if (B == 1) {
    const hipError_t e = mh::launch_linearize(ia[0], spec[0].S->icp->binary, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch_linearize");
  } else {
    for (const Group & g : groups) {
      mh::BatchInline<mh::IcpArgs> blk;
      std::memset(static_cast<void *>(&blk), 0, sizeof(blk));
      int acc = 0;
      for (size_t i = 0; i < g.members.size(); ++i) {
        blk.a[i] = ia[g.members[i]];
        blk.start[i] = acc;
        acc += mh::class_grid(ia[g.members[i]].n, g.tpb);
      }
      blk.start[g.members.size()] = acc;
      blk.n = static_cast<int>(g.members.size());
      const hipError_t e = mh::launch_linearize_batch_inline(blk, acc, g.tpb, g.k, g.n_off, g.binary, ctx->stream, true);
      if (e != hipSuccess) return hip_fail(ctx, e, "launch_linearize_batch_inline");
    }
  }
  rc = comm->all_reduce(ar, B * mh::kShardArLen, ctx->stream);  // C3a
  if (rc != MH_OK) return fail(ctx, rc, "mh_shard_icp_linearize: all-reduce failed: " + comm->err);
  if (round.any_components) {  // (a mixed round runs K4 for all; the factors that did not ask report NaN all the same)
    if (B == 1) {
      const hipError_t e = mh::launch_localizability(la[0], ctx->stream);
      if (e != hipSuccess) return hip_fail(ctx, e, "launch_localizability");
    } else {
      for (const Group & g : groups) {
        mh::BatchInline<mh::LocArgs> blk;
        std::memset(static_cast<void *>(&blk), 0, sizeof(blk));
        int acc = 0;
        for (size_t i = 0; i < g.members.size(); ++i) {
          blk.a[i] = la[g.members[i]];
          blk.start[i] = acc;
          acc += mh::class_grid(ia[g.members[i]].n, g.tpb);
        }
        blk.start[g.members.size()] = acc;
        blk.n = static_cast<int>(g.members.size());
        const hipError_t e = mh::launch_localizability_batch_inline(blk, acc, g.tpb, ctx->stream, true);
        if (e != hipSuccess) return hip_fail(ctx, e, "launch_localizability_batch_inline");
      }
    }
    rc = comm->all_reduce(loc, B * 16, ctx->stream);  // C3b
    if (rc != MH_OK) return fail(ctx, rc, "mh_shard_icp_linearize: all-reduce failed: " + comm->err);
  }
  // ---- publish into the next slot of the ring
  if (++comm->seq == 0) ++comm->seq;
  round.seq = comm->seq;
  const size_t slot = static_cast<size_t>(comm->ring_pos) * comm->ring_width;
  comm->ring_pos = (comm->ring_pos + 1) % kShardRing;
  round.h_pub = comm->h_ring + slot;
  mh::ShardPublish * d_pub = comm->d_h_ring + slot;
  if (B == 1) {
    const hipError_t e = mh::launch_shard_publish(ar, round.any_components ? loc : nullptr, spec[0].S->d_state, spec[0].S->cur ^ 1, d_pub, round.seq, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch_shard_publish");
  } else {
    for (size_t f0 = 0; f0 < B; f0 += mh::kShardBatchMax) {
      mh::ShardPublishBatch pb;
      std::memset(static_cast<void *>(&pb), 0, sizeof(pb));
      pb.n = static_cast<int>(std::min<size_t>(mh::kShardBatchMax, B - f0));
      for (int i = 0; i < pb.n; ++i) {
        pb.st[i] = spec[f0 + i].S->d_state;
        pb.next[i] = spec[f0 + i].S->cur ^ 1;
      }
      pb.host = d_pub + f0;
      pb.ar = ar + f0 * mh::kShardArLen;
      pb.loc = round.any_components ? loc + f0 * 16 : nullptr;
      pb.seq = round.seq;
      const hipError_t e = mh::launch_shard_publish_batch(pb, ctx->stream);
      if (e != hipSuccess) return hip_fail(ctx, e, "launch_shard_publish_batch");
    }
  }
  for (size_t f = 0; f < B; ++f) {
    mh_shard_icp * S = spec[f].S;
    S->cur ^= 1;
    S->slots_bound = static_cast<uint32_t>(std::min<uint64_t>(S->slot_capacity, static_cast<uint64_t>(S->slots_bound) + round.calls[f].arrivals_bound));
    S->inflight++;
    S->icp->n = S->slots_bound;
    S->stats.collectives_last = static_cast<uint32_t>(comm->n_all_to_all + comm->n_all_reduce - coll_before);  // (a blocking call adds its repeats')
  }
  comm->rounds.push_back(std::move(round));
  comm->n_rounds++;
  guard.ok = true;
  return MH_OK;
}

// argument checks shared by the entry points; the factors of a round share communicator and context
int check_round(mh_shard_icp * const * Ss, size_t B, const double * R_src, const double * t_src, const double * R_tgt, const double * t_tgt,
                const double * g_unit, const mh_icp_result * out, const char * who)
{
  if (!Ss || !B || !R_src || !t_src || !g_unit || !out) return fail(nullptr, MH_ERR_INVALID_ARG, std::string(who) + ": NULL argument");
  if (B > static_cast<size_t>(kMaxBatch)) return fail(nullptr, MH_ERR_UNSUPPORTED, std::string(who) + ": at most 64 factors per call");
  for (size_t f = 0; f < B; ++f)
    if (!Ss[f]) return fail(nullptr, MH_ERR_INVALID_ARG, std::string(who) + ": NULL factor");
  for (size_t f = 0; f < B; ++f) MH_SHARD_ALIVE(Ss[f], who);
  mh_ctx * ctx = Ss[0]->ctx;
  for (size_t f = 0; f < B; ++f)
    if (!Ss[f]->comm) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": the factor's communicator was destroyed");
  for (size_t f = 0; f < B; ++f) {
    const mh_shard_icp * S = Ss[f];
    if (S->ctx != ctx || S->comm != Ss[0]->comm) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": factors of different contexts / communicators");
    if (S->collective != Ss[0]->collective) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": factors with and without the exchange protocol in one call");
    if (S->icp->binary && (!R_tgt || !t_tgt)) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": binary factor needs the target pose");
    if (S->broken) return fail(ctx, MH_ERR_HIP, std::string(who) + ": an earlier call on this factor failed half way; create a new factor");
    for (size_t g = 0; g < f; ++g)
      if (Ss[g] == S) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": the same factor twice");
  }
  return MH_OK;
}

int round_from_arrays(mh_shard_icp * const * Ss, size_t B, const double * R_src, const double * t_src, const double * R_tgt, const double * t_tgt,
                      const double * g_unit, mh_icp_result * out)
{
  std::vector<RoundSpec> spec(B);
  for (size_t f = 0; f < B; ++f) {
    spec[f] = RoundSpec{Ss[f], R_src + 9 * f, t_src + 3 * f, R_tgt ? R_tgt + 9 * f : nullptr, t_tgt ? t_tgt + 3 * f : nullptr, g_unit + 3 * f, out + f};
    Ss[f]->stats.retries_last = 0;
  }
  return enqueue_round(spec.data(), B);
}
}  // namespace

static int shard_icp_linearize_impl(mh_shard_icp * S, const double R_src[9], const double t_src[3], const double * R_tgt, const double * t_tgt,
                                    const double g_unit[3], mh_icp_result * out)
{
  int rc = check_round(&S, S ? 1 : 0, R_src, t_src, R_tgt, t_tgt, g_unit, out, "mh_shard_icp_linearize");
  if (rc != MH_OK) return rc;
  if (!S->collective) {
    rc = mh_icp_linearize(S->icp, R_src, t_src, R_tgt, t_tgt, g_unit, out);
    S->stats.n_live = S->stats.n_slots = S->icp->n;
    return rc;
  }
  mh_shard_comm * comm = S->comm;
  const long long coll0 = comm->n_all_to_all + comm->n_all_reduce;
  rc = round_from_arrays(&S, 1, R_src, t_src, R_tgt, t_tgt, g_unit, out);
  if (rc != MH_OK) return rc;
  rc = settle_rounds(comm, ~0ull);
  S->stats.collectives_last = static_cast<uint32_t>(comm->n_all_to_all + comm->n_all_reduce - coll0);
  return rc;
}
int mh_shard_icp_linearize(mh_shard_icp * S, const double R_src[9], const double t_src[3], const double * R_tgt, const double * t_tgt, const double g_unit[3],
                           mh_icp_result * out)
{
  return guarded(S ? S->ctx : nullptr, "mh_shard_icp_linearize", [&]() -> int { return shard_icp_linearize_impl(S, R_src, t_src, R_tgt, t_tgt, g_unit, out); });
}

int mh_shard_icp_linearize_async(mh_shard_icp * S, const double R_src[9], const double t_src[3], const double * R_tgt, const double * t_tgt, const double g_unit[3],
                                 mh_icp_result * out)
{
  return guarded(S ? S->ctx : nullptr, "mh_shard_icp_linearize_async", [&]() -> int {
    const int rc = check_round(&S, S ? 1 : 0, R_src, t_src, R_tgt, t_tgt, g_unit, out, "mh_shard_icp_linearize_async");
    if (rc != MH_OK) return rc;
    if (!S->collective) {
      if (std::find(S->comm->plain_pending.begin(), S->comm->plain_pending.end(), S) == S->comm->plain_pending.end()) S->comm->plain_pending.push_back(S);
      return mh_icp_linearize_async(S->icp, R_src, t_src, R_tgt, t_tgt, g_unit, out);
    }
    return round_from_arrays(&S, 1, R_src, t_src, R_tgt, t_tgt, g_unit, out);
  });
}

int mh_shard_icp_linearize_batch_async(mh_shard_icp * const * Ss, size_t n_factors, const double * R_src, const double * t_src, const double * R_tgt,
                                       const double * t_tgt, const double * g_unit, mh_icp_result * out)
{
  return guarded((Ss && n_factors && Ss[0]) ? Ss[0]->ctx : nullptr, "mh_shard_icp_linearize_batch_async", [&]() -> int {
    const int rc = check_round(Ss, n_factors, R_src, t_src, R_tgt, t_tgt, g_unit, out, "mh_shard_icp_linearize_batch_async");
    if (rc != MH_OK) return rc;
    if (!Ss[0]->collective) {  // one rank, nothing to exchange: the plain factors, enqueued one behind the other
      for (size_t f = 0; f < n_factors; ++f) {
        auto & pend = Ss[f]->comm->plain_pending;
        if (std::find(pend.begin(), pend.end(), Ss[f]) == pend.end()) pend.push_back(Ss[f]);
        const int r2 = mh_icp_linearize_async(Ss[f]->icp, R_src + 9 * f, t_src + 3 * f, R_tgt ? R_tgt + 9 * f : nullptr, t_tgt ? t_tgt + 3 * f : nullptr, g_unit + 3 * f, out + f);
        if (r2 != MH_OK) return r2;
      }
      return MH_OK;
    }
    return round_from_arrays(Ss, n_factors, R_src, t_src, R_tgt, t_tgt, g_unit, out);
  });
}

int mh_shard_icp_linearize_batch(mh_shard_icp * const * Ss, size_t n_factors, const double * R_src, const double * t_src, const double * R_tgt,
                                 const double * t_tgt, const double * g_unit, mh_icp_result * out)
{
  return guarded((Ss && n_factors && Ss[0]) ? Ss[0]->ctx : nullptr, "mh_shard_icp_linearize_batch", [&]() -> int {
    int rc = check_round(Ss, n_factors, R_src, t_src, R_tgt, t_tgt, g_unit, out, "mh_shard_icp_linearize_batch");
    if (rc != MH_OK) return rc;
    if (!Ss[0]->collective) {  // one rank, nothing to exchange: the window batch of the plain factors
      for (size_t f = 0; f < n_factors; ++f)
        if (Ss[f]->icp->n_pending) return fail(Ss[0]->ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_linearize_batch: calls in flight");
      std::vector<mh_icp *> icps(n_factors);
      for (size_t f = 0; f < n_factors; ++f) icps[f] = Ss[f]->icp;
      return mh_icp_linearize_batch(icps.data(), n_factors, R_src, t_src, R_tgt, t_tgt, g_unit, out);
    }
    mh_shard_comm * comm = Ss[0]->comm;
    const long long coll0 = comm->n_all_to_all + comm->n_all_reduce;
    rc = round_from_arrays(Ss, n_factors, R_src, t_src, R_tgt, t_tgt, g_unit, out);
    if (rc != MH_OK) return rc;
    rc = settle_rounds(comm, ~0ull);
    for (size_t f = 0; f < n_factors; ++f) Ss[f]->stats.collectives_last = static_cast<uint32_t>(comm->n_all_to_all + comm->n_all_reduce - coll0);
    return rc;
  });
}

int mh_shard_icp_wait(mh_shard_icp * S)
{
  return guarded(S ? S->ctx : nullptr, "mh_shard_icp_wait", [&]() -> int {
    if (!S) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_wait: NULL argument");
    MH_SHARD_ALIVE(S, "mh_shard_icp_wait");
    if (!S->comm) return fail(S->ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_wait: the factor's communicator was destroyed");
    if (!S->collective) {  // every factor of the communicator with enqueued calls, like the rounds of the collective form
      int rc_all = mh_icp_wait(S->icp);
      for (mh_shard_icp * q : S->comm->plain_pending)
        if (q != S) {
          const int rc = mh_icp_wait(q->icp);
          if (rc != MH_OK) rc_all = rc;
        }
      S->comm->plain_pending.clear();
      return rc_all;
    }
    MH_HIP(S->ctx, mh_enter(S->ctx));
    return settle_rounds(S->comm, ~0ull);
  });
}

int mh_shard_icp_reset(mh_shard_icp * S)
{
  return guarded(S ? S->ctx : nullptr, "mh_shard_icp_reset", [&]() -> int {
    if (!S) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_reset: NULL argument");
    MH_SHARD_ALIVE(S, "mh_shard_icp_reset");
    if (!S->collective) return mh_icp_reset(S->icp);
    mh_ctx * ctx = S->ctx;
    MH_HIP(ctx, mh_enter(ctx));
    // No kernel: like mh_icp_reset, the next enqueued call treats the association state of every point as freshly
    // constructed (K3's `cold` launch reads it as zero and writes it for every slot it processes; movers' records carry
    // stale state that the receiving rank's cold K3 ignores the same way).  Ordered like a stream operation: it applies to
    // the calls enqueued after it.
    (void)ctx;
    S->cold_pending = true;
    return MH_OK;
  });
}

int mh_shard_icp_set_components(mh_shard_icp * S, int enabled)
{
  if (!S) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_set_components: NULL argument");
  MH_SHARD_ALIVE(S, "mh_shard_icp_set_components");
  return mh_icp_set_components(S->icp, enabled);
}

static int shard_icp_get_state_impl(mh_shard_icp * S, uint64_t * origin, int32_t * status, double * means, double * normals, size_t capacity, size_t * n_out)
{
  if (!S || !n_out) return fail(S ? S->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_get_state: NULL argument");
  MH_SHARD_ALIVE(S, "mh_shard_icp_get_state");
  mh_ctx * ctx = S->ctx;
  mh_icp * icp = S->icp;
  if (S->collective && S->inflight) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_get_state: calls in flight (mh_shard_icp_wait first)");
  *n_out = S->n_live;
  if (!origin && !status && !means && !normals) return MH_OK;
  if (capacity < S->n_live) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_get_state: capacity too small");
  if (!S->collective) {
    if (origin)
      for (size_t i = 0; i < S->n_live; ++i) origin[i] = (static_cast<uint64_t>(S->stats.rank) << 32) | i;
    return mh_icp_get_state(icp, status, means, normals);
  }
  MH_HIP(ctx, mh_enter(ctx));
  const size_t n = S->n_slots;
  std::vector<uint64_t> o(n);
  std::vector<int32_t> st(n);
  std::vector<double> m(3 * n), nr(3 * n);
  if (n) {
    MH_HIP(ctx, hipMemcpyAsync(o.data(), icp->d_origin.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(st.data(), icp->d_status.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(m.data(), icp->d_mean.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(nr.data(), icp->d_normal.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) {
    if (o[i] == mh::kShardTomb) continue;
    if (k >= capacity) return fail(ctx, MH_ERR_HIP, "mh_shard_icp_get_state: live count mismatch");
    if (origin) origin[k] = o[i];
    if (status) status[k] = st[i] & ~mh::kShardSkip;
    if (means) std::memcpy(means + 3 * k, m.data() + 3 * i, 24);
    if (normals) std::memcpy(normals + 3 * k, nr.data() + 3 * i, 24);
    ++k;
  }
  *n_out = k;
  return MH_OK;
}
int mh_shard_icp_get_state(mh_shard_icp * S, uint64_t * origin, int32_t * status, double * means, double * normals, size_t capacity, size_t * n_out)
{
  return guarded(S ? S->ctx : nullptr, "mh_shard_icp_get_state", [&]() -> int { return shard_icp_get_state_impl(S, origin, status, means, normals, capacity, n_out); });
}

int mh_shard_owner_of_block(int bx, int by, int bz, int world)
{
  // the host twin of shard_kernels.hip's owner_of_block (same tables, same arithmetic)
  static const uint8_t A[mh::kShardMaxWorld + 1] = MH_SHARD_OWNER_A, B[mh::kShardMaxWorld + 1] = MH_SHARD_OWNER_B;
  if (world < 1 || world > mh::kShardMaxWorld) return -1;
  int r = (bx % world + static_cast<int>(A[world]) * (by % world) + static_cast<int>(B[world]) * (bz % world)) % world;
  return r < 0 ? r + world : r;
}

int mh_shard_icp_stats(const mh_shard_icp * S, mh_shard_stats * out)
{
  if (!S || !out) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_stats: NULL argument");
  *out = S->stats;
  out->n_live = S->n_live;
  out->n_slots = S->n_slots;
  out->slot_capacity = S->slot_capacity;
  out->n_total = S->n_total;
  out->segment_records = S->seg_cap;
  out->world = S->comm ? S->comm->world : S->stats.world;
  out->rank = S->comm ? S->comm->rank : S->stats.rank;
  out->collective = S->collective ? 1 : 0;
  out->linearize_count = S->icp ? S->icp->linearize_count : S->last_linearize_count;
  return MH_OK;
}

}  // extern "C"
