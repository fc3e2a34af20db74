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

struct mh_shard_comm
{
  int world = 1, rank = 0;
  bool is_rccl = false;
  int device = 0;
  ncclComm_t nccl = nullptr;
  LocalGroup * grp = nullptr;
  std::string err;
  long long n_all_to_all = 0, n_all_reduce = 0;

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
  DevBuf d_dest, d_hist, d_send, d_recv, d_ar, d_loc, d_flags, d_pos, d_temp;
  mh::ShardPublish * h_pub = nullptr;    // mapped pinned
  mh::ShardPublish * d_h_pub = nullptr;  // its device address
  // host knowledge (exact after every call: the publish kernel reports the counters)
  int cur = 0;
  uint32_t n_slots = 0, n_live = 0, slot_capacity = 0;
  uint32_t seg_cap = 0, seg_cap_max = 0;
  uint64_t n_total = 0;
  unsigned int seq = 0;
  bool broken = false;  // a collective call failed half way (records sent, slots tombstoned): the factor's state is not to be trusted
  mh_shard_stats stats{};
};

namespace
{
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
int wait_publish(mh_shard_icp * S, unsigned int seq)
{
  mh_ctx * ctx = S->ctx;
  mh_shard_comm * comm = S->comm;
  const volatile unsigned int * flag = &S->h_pub->seq;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 1023u) != 1023u) continue;
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) < 50000000L) continue;
    // 50 ms: something is slow (a first call's channel set-up) or wrong (a peer is gone).  Stop burning the core; poll the
    // stream and, over RCCL, the communicator's asynchronous error state, so that a dead peer is an error and not a hang
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
      std::this_thread::sleep_for(std::chrono::microseconds(200));
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
  }
  S->stats.compactions_total++;
  return MH_OK;
}

// the global result from the all-reduced vector: localizabilities of the GLOBAL H (geometric_factor.hpp:405-411), Schur
// degeneracy info, 4-DoF projection and the degeneracy branch (:413-428, :464-557) — once, on the global sums
void global_result(mh_shard_icp * S, const PendingCall & pc, bool components, mh_icp_result * out)
{
  const mh::ShardPublish & p = *S->h_pub;
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
      out_array[r] = c;
    }
    return MH_OK;
  });
}

void mh_shard_comm_destroy(mh_shard_comm * comm)
{
  if (!comm) return;
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

void mh_shard_icp_destroy(mh_shard_icp * S)
{
  if (!S) return;
  if (S->ctx) {
    (void)mh_enter(S->ctx);
    (void)hipStreamSynchronize(S->ctx->stream);
  }
  for (DevBuf * b : {&S->d_dest, &S->d_hist, &S->d_send, &S->d_recv, &S->d_ar, &S->d_loc, &S->d_flags, &S->d_pos, &S->d_temp}) b->release(true);
  if (S->d_state) AllocCache::free(S->d_state, true);
  if (S->h_pub) AllocCache::free_pinned(S->h_pub, sizeof(mh::ShardPublish));
  if (S->icp) mh_icp_destroy(S->icp);
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
  MH_HIP(ctx, S->d_loc.reserve(16 * sizeof(double), ctx->stream, false));
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
  MH_HIP(ctx, AllocCache::alloc_pinned(reinterpret_cast<void **>(&S->h_pub), sizeof(mh::ShardPublish)));
  std::memset(S->h_pub, 0, sizeof(mh::ShardPublish));
  MH_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&S->d_h_pub), S->h_pub, 0));
  MH_HIP(ctx, hipMemsetAsync(S->d_ar.p, 0, mh::kShardArLen * sizeof(double), ctx->stream));
  MH_HIP(ctx, hipMemsetAsync(S->d_loc.p, 0, 16 * sizeof(double), ctx->stream));
  const size_t slots_rounded = (static_cast<size_t>(S->slot_capacity) + 255) & ~size_t(255);
  MH_HIP(ctx, S->d_dest.reserve(slots_rounded, ctx->stream, false));
  MH_HIP(ctx, S->d_hist.reserve((slots_rounded / 256 + 1) * comm->world * sizeof(uint32_t), ctx->stream, false));
  // partial rows for the largest grid a call can use
  MH_HIP(ctx, icp->d_partials.reserve(static_cast<size_t>(mh::linearize_grid(static_cast<int>(S->slot_capacity))) * mh::kPartialStride * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  S->n_slots = S->n_live = static_cast<uint32_t>(n_local);
  *out = S;
  return MH_OK;
}
int mh_shard_icp_create(mh_ctx * ctx, mh_shard_comm * comm, mh_map * map, const mh_point32 * points, size_t n_local, int points_on_device,
                        const mh_reg_config * cfg, int is_binary, const mh_shard_config * scfg, mh_shard_icp ** out)
{
  return guarded(ctx, "mh_shard_icp_create", [&]() -> int { return shard_icp_create_impl(ctx, comm, map, points, n_local, points_on_device, cfg, is_binary, scfg, out); });
}

static int shard_icp_linearize_impl(mh_shard_icp * S, const double R_src[9], const double t_src[3], const double * R_tgt, const double * t_tgt,
                                    const double g_unit[3], mh_icp_result * out)
{
  if (!S || !R_src || !t_src || !g_unit || !out) return fail(S ? S->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_linearize: NULL argument");
  mh_ctx * ctx = S->ctx;
  mh_icp * icp = S->icp;
  mh_shard_comm * comm = S->comm;
  S->stats.collectives_last = 0;
  S->stats.retries_last = 0;
  if (!S->collective) {
    const int rc = mh_icp_linearize(icp, R_src, t_src, R_tgt, t_tgt, g_unit, out);
    S->stats.n_live = S->stats.n_slots = icp->n;
    return rc;
  }
  if (icp->binary && (!R_tgt || !t_tgt)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_linearize: binary factor needs the target pose");
  if (icp->n_pending) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_linearize: calls in flight");
  if (S->broken) return fail(ctx, MH_ERR_HIP, "mh_shard_icp_linearize: an earlier call on this factor failed half way; create a new factor");
  struct Guard  // anything that leaves this function other than by success has left points in flight or tombstoned
  {
    mh_shard_icp * s;
    bool ok = false;
    ~Guard()
    {
      if (!ok) s->broken = true;
    }
  } guard{S};
  MH_HIP(ctx, mh_enter(ctx));
  const uint32_t world = static_cast<uint32_t>(comm->world), rank = static_cast<uint32_t>(comm->rank);
  const long long coll0 = comm->n_all_to_all + comm->n_all_reduce;
  mh::ShardPose P;
  mhi::pose_delta(R_src, t_src, icp->binary ? R_tgt : nullptr, icp->binary ? t_tgt : nullptr, P.R, P.t);
  const bool components = icp->components;
  const int saved_count = icp->linearize_count;
  double carried[4] = {0, 0, 0, 0};  // k-NN counters of repeated attempts (their points hit the cache in the repeat)
  for (int attempt = 0;; ++attempt) {
    if (attempt > 8) return fail(ctx, MH_ERR_HIP, "mh_shard_icp_linearize: segment capacity did not converge");
    const uint32_t cap = S->seg_cap;
    // what can arrive at most: a full segment from every peer, and never more than the points held elsewhere
    auto arrivals_max = [&]() { return std::min<uint64_t>(static_cast<uint64_t>(world - 1) * cap, S->n_total - std::min<uint64_t>(S->n_total, S->n_live)); };
    // tombstones are dropped when the arrivals of this call might not fit behind them, or when they outnumber the points
    // (after a compaction n_slots == n_live, and n_live + everything held elsewhere == n_total always fits)
    if (S->n_slots + arrivals_max() > S->slot_capacity || S->n_slots > S->n_live + S->n_live / 2 + 16384) {
      const int rc = shard_compact(S);
      if (rc != MH_OK) return rc;
    }
    const size_t seg = mh::shard_segment_bytes(cap);
    MH_HIP(ctx, S->d_send.reserve(seg * world, ctx->stream, false));
    MH_HIP(ctx, S->d_recv.reserve(seg * world, ctx->stream, false));
    const int cur = S->cur;
    MH_HIP(ctx, mh::launch_shard_route(P, arrays_of(icp, false), S->d_state, cur, S->n_slots, 1.0 / icp->map->cfg.leaf_size, world, rank, S->block_log2,
                                       static_cast<uint8_t *>(S->d_dest.p), static_cast<uint32_t *>(S->d_hist.p), cap, static_cast<char *>(S->d_send.p),
                                       static_cast<double *>(S->d_ar.p) + mh::kShardSums, ctx->stream));
    int rc = comm->all_to_all(S->d_send.p, S->d_recv.p, seg, ctx->stream);  // C1
    if (rc != MH_OK) return fail(ctx, rc, "mh_shard_icp_linearize: all-to-all failed: " + comm->err);
    MH_HIP(ctx, mh::launch_shard_append(arrays_of(icp, false), S->d_state, cur, world, cap, static_cast<const char *>(S->d_recv.p), S->slot_capacity, ctx->stream));
    // K3 over the slots: the grid covers what the slot count can be at most, the kernel reads the count itself
    const uint32_t bound = static_cast<uint32_t>(std::min<uint64_t>(S->slot_capacity, S->n_slots + arrivals_max()));
    mh::IcpArgs a;
    mh::LocArgs l;
    static thread_local mh_icp_result scratch;  // filled by nobody: the result is assembled from the global sums below
    icp->n = bound ? bound : 1;
    rc = mhi::prepare(icp, R_src, t_src, R_tgt, t_tgt, g_unit, &scratch, false, a, l);
    if (rc != MH_OK) return rc;
    const PendingCall pc = icp->pending[0];
    a.n_dev = l.n_dev = &S->d_state->n_slots[cur ^ 1];
    a.cold = 0;
    a.host_result = nullptr;
    a.seq = 0;
    a.shard_out = static_cast<double *>(S->d_ar.p);
    auto unwind = [&](int code) {
      icp->n_pending = 0;
      return code;
    };
    {
      const hipError_t e = mh::launch_linearize(a, icp->binary, ctx->stream);
      if (e != hipSuccess) return unwind(hip_fail(ctx, e, "launch_linearize"));
    }
    rc = comm->all_reduce(static_cast<double *>(S->d_ar.p), mh::kShardArLen, ctx->stream);  // C3a
    if (rc != MH_OK) return unwind(fail(ctx, rc, "mh_shard_icp_linearize: all-reduce failed: " + comm->err));
    if (components) {
      l.host_result = nullptr;
      l.seq = 0;
      l.sums = static_cast<const double *>(S->d_ar.p);  // the eigenbases of the GLOBAL H_rr / H_tt
      l.shard_out = static_cast<double *>(S->d_loc.p);
      const hipError_t e = mh::launch_localizability(l, ctx->stream);
      if (e != hipSuccess) return unwind(hip_fail(ctx, e, "launch_localizability"));
      rc = comm->all_reduce(static_cast<double *>(S->d_loc.p), 16, ctx->stream);  // C3b
      if (rc != MH_OK) return unwind(fail(ctx, rc, "mh_shard_icp_linearize: all-reduce failed: " + comm->err));
    }
    if (++S->seq == 0) ++S->seq;
    {
      const hipError_t e = mh::launch_shard_publish(static_cast<const double *>(S->d_ar.p), components ? static_cast<const double *>(S->d_loc.p) : nullptr, S->d_state,
                                                    cur ^ 1, S->d_h_pub, S->seq, ctx->stream);
      if (e != hipSuccess) return unwind(hip_fail(ctx, e, "launch_shard_publish"));
    }
    rc = wait_publish(S, S->seq);
    icp->n_pending = 0;
    if (rc != MH_OK) return rc;
    S->cur ^= 1;
    const mh::ShardPublish & p = *S->h_pub;
    S->n_slots = p.n_slots;
    S->n_live = p.n_live;
    icp->n = S->n_slots;
    if (p.error) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_shard_icp_linearize: arrivals exceeded the slot capacity");
    uint32_t max_movers = 0;
    for (uint32_t r = 0; r < world; ++r) max_movers = std::max(max_movers, static_cast<uint32_t>(p.ar[mh::kShardSums + r]));
    S->stats.last_max_movers = max_movers;
    if (max_movers > cap) {
      // some rank could not send everything (every rank reads the same maxima, so every rank repeats): larger segments,
      // same pose — processed points hit their data-association cache, the held-back ones travel now
      S->seg_cap = std::min(S->seg_cap_max, pow2_at_least(max_movers));
      S->stats.retries_total++;
      S->stats.retries_last++;
      icp->linearize_count = saved_count;
      for (int i = 0; i < 4; ++i) carried[i] += p.ar[(icp->binary ? 91 : 28) + i];
      continue;
    }
    // next call: room for four times what moved now (a pose step of centimetres moves a few points across block faces)
    S->seg_cap = std::min(S->seg_cap_max, std::max<uint32_t>(256u, pow2_at_least(4u * max_movers)));
    for (int i = 0; i < 4; ++i) S->h_pub->ar[(icp->binary ? 91 : 28) + i] += carried[i];
    global_result(S, pc, components, out);
    break;
  }
  S->stats.collectives_last = static_cast<uint32_t>(comm->n_all_to_all + comm->n_all_reduce - coll0);
  guard.ok = true;
  return MH_OK;
}
int mh_shard_icp_linearize(mh_shard_icp * S, const double R_src[9], const double t_src[3], const double * R_tgt, const double * t_tgt, const double g_unit[3],
                           mh_icp_result * out)
{
  return guarded(S ? S->ctx : nullptr, "mh_shard_icp_linearize", [&]() -> int { return shard_icp_linearize_impl(S, R_src, t_src, R_tgt, t_tgt, g_unit, out); });
}

int mh_shard_icp_reset(mh_shard_icp * S)
{
  return guarded(S ? S->ctx : nullptr, "mh_shard_icp_reset", [&]() -> int {
    if (!S) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_reset: NULL argument");
    if (!S->collective) return mh_icp_reset(S->icp);
    mh_ctx * ctx = S->ctx;
    MH_HIP(ctx, mh_enter(ctx));
    MH_HIP(ctx, mh::launch_shard_reset(arrays_of(S->icp, false), S->d_state, S->cur, S->n_slots, ctx->stream));
    return MH_OK;
  });
}

int mh_shard_icp_set_components(mh_shard_icp * S, int enabled)
{
  if (!S) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_set_components: NULL argument");
  return mh_icp_set_components(S->icp, enabled);
}

static int shard_icp_get_state_impl(mh_shard_icp * S, uint64_t * origin, int32_t * status, double * means, double * normals, size_t capacity, size_t * n_out)
{
  if (!S || !n_out) return fail(S ? S->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_get_state: NULL argument");
  mh_ctx * ctx = S->ctx;
  mh_icp * icp = S->icp;
  *n_out = S->n_live;
  if (!origin && !status && !means && !normals) return MH_OK;
  if (capacity < S->n_live) return fail(ctx, MH_ERR_INVALID_ARG, "mh_shard_icp_get_state: capacity too small");
  if (!S->collective) {
    if (origin)
      for (size_t i = 0; i < S->n_live; ++i) origin[i] = (static_cast<uint64_t>(S->comm->rank) << 32) | i;
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

int mh_shard_icp_stats(const mh_shard_icp * S, mh_shard_stats * out)
{
  if (!S || !out) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_shard_icp_stats: NULL argument");
  *out = S->stats;
  out->n_live = S->n_live;
  out->n_slots = S->n_slots;
  out->slot_capacity = S->slot_capacity;
  out->n_total = S->n_total;
  out->segment_records = S->seg_cap;
  out->world = S->comm->world;
  out->rank = S->comm->rank;
  out->collective = S->collective ? 1 : 0;
  out->linearize_count = S->icp->linearize_count;
  return MH_OK;
}

}  // extern "C"
