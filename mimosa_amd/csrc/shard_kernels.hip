// HIP kernels of the map-sharded scan-to-map factor (BASELINE configs[2]): the map is partitioned by
// spatial hash across the GPUs of a node, every source point is linearized on the rank that owns its centre voxel.
// The reference is single-process (no counterpart); what these kernels must preserve is that the SHARDED result equals
// the unsharded ICPFactor::linearize (include/mimosa/lidar/geometric_factor.hpp:231-562) bit for bit per point.
//
// Partition: shard blocks of 2^log2 voxels per axis, owner = a lattice colouring of the block grid (owner_of_block below;
// mh_shard_owner_of_block is its host twin).  Every rank also stores the one-voxel halo of its blocks, so the 1/7/19/27
// neighbourhood of a query in an owned block is complete locally.
//   shard_filter   which points of an insert batch this rank keeps (owned blocks + halo), order preserved
//   route / append / compact_slots / publish: the native protocol's stages (further down, shard_device.hpp)
// Streaming over <= 131 072 records per rank: HBM- and launch-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "math3.hpp"
#include "shard_device.hpp"
#include "voxel_map.hpp"

namespace mh
{
namespace
{
constexpr int kT = 256;
int grid_for(uint32_t n) { return static_cast<int>(max(1u, min((n + kT - 1) / kT, 8192u))); }

// Owner of a block: rank = (bx + A[P] by + B[P] bz) mod P — a lattice colouring of the block grid (tools/lattice_table.py has
// the rule the two tables come from and prints them).  Neighbouring blocks never share a rank and a sheet of blocks (a wall, a
// floor) spreads over all ranks as evenly as a static function can, so the few blocks next to the sensor that hold most of a
// scan's points do too: on the configs[1] world the fullest of 8 ranks gets 1.30 x its fair share of the queries, against the
// 1.65 x of the XOR hash of the block coordinates that rounds 3-4 used (neighbouring heavy blocks met on one rank at random).
// Results do not depend on the owner function; the storage balance stays within 2 percent.
__constant__ uint8_t kOwnerA[kShardMaxWorld + 1] = MH_SHARD_OWNER_A;
__constant__ uint8_t kOwnerB[kShardMaxWorld + 1] = MH_SHARD_OWNER_B;
__device__ __forceinline__ uint32_t owner_of_block(int bx, int by, int bz, uint32_t world)
{
  const int w = static_cast<int>(world);  // the terms are reduced first: no overflow for any block coordinate
  int r = (bx % w + static_cast<int>(kOwnerA[world]) * (by % w) + static_cast<int>(kOwnerB[world]) * (bz % w)) % w;
  r = r < 0 ? r + w : r;
  return static_cast<uint32_t>(r);
}

__global__ __launch_bounds__(kT) void shard_filter_kernel(const float * xyz, uint32_t n, uint32_t stride, double inv_leaf, uint32_t world,
                                                           uint32_t rank, int log2, uint32_t * flags)
{
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) {
    const int cx = fast_floor(static_cast<double>(xyz[static_cast<size_t>(i) * stride]) * inv_leaf),
              cy = fast_floor(static_cast<double>(xyz[static_cast<size_t>(i) * stride + 1]) * inv_leaf),
              cz = fast_floor(static_cast<double>(xyz[static_cast<size_t>(i) * stride + 2]) * inv_leaf);
    bool need = false;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dz = -1; dz <= 1; ++dz) need |= owner_of_block((cx + dx) >> log2, (cy + dy) >> log2, (cz + dz) >> log2, world) == rank;
    flags[i] = need ? 1u : 0u;
  }
}
__global__ __launch_bounds__(kT) void shard_compact_kernel(const float * xyz, uint32_t n, uint32_t stride, const uint32_t * flags,
                                                            const uint32_t * pos, float * out, uint32_t * n_out)
{
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) {
    if (flags[i]) {
      const size_t o = static_cast<size_t>(pos[i]) * 3;
      out[o] = xyz[static_cast<size_t>(i) * stride];
      out[o + 1] = xyz[static_cast<size_t>(i) * stride + 1];
      out[o + 2] = xyz[static_cast<size_t>(i) * stride + 2];
    }
    if (i == n - 1) *n_out = pos[i] + flags[i];
  }
}


__global__ __launch_bounds__(kT) void shard_origin_kernel(unsigned long long * origin, uint32_t n, uint32_t rank)
{
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) origin[i] = (static_cast<unsigned long long>(rank) << 32) | i;
}




// ---- native sharded factor: slots, tombstones, fixed-capacity segments (shard_device.hpp) ----------------------------
__global__ void shard_state_init_kernel(ShardState * st, uint32_t n)
{
  if (threadIdx.x == 0) {
    st->n_slots[0] = st->n_slots[1] = n;
    st->n_live[0] = st->n_live[1] = n;
    st->sent_total = st->max_total = st->error = st->pad = 0u;
  }
}

// ---- one factor of a protocol round (shard_device.hpp: ShardFactorArgs); the single-factor kernels and the batched ones
// (all factors of a window in ONE launch each: route count, route pack, append, publish) run the same bodies --------------
// Pass 1 of the routing: destination of every live slot (0xFF: stays, tombstone or past the end) and the block's count
// per destination.  A slot that could not be sent in the previous call (segment overflow) loses its skip flag here and is
// looked at afresh.  blk = this workgroup's index within the factor's grid.
__device__ __forceinline__ void route_count_body(const ShardFactorArgs & f, const uint32_t world, const uint32_t rank, const uint32_t blk)
{
  __shared__ uint32_t s_h[kShardMaxWorld];
  if (threadIdx.x < kShardMaxWorld) s_h[threadIdx.x] = 0u;
  __syncthreads();
  const ShardArrays & a = f.a;
  const uint32_t n = f.st->n_slots[f.cur];
  const uint32_t i = blk * kT + threadIdx.x;
  uint32_t d = 0xFFu;
  if (i < n && a.origin[i] != kShardTomb) {
    const int32_t s = a.status[i];
    if (s & kShardSkip) a.status[i] = s & ~kShardSkip;
    const float4 sp = a.src[i];
    const double px = sp.x, py = sp.y, pz = sp.z;
    const double q0 = (f.P.R[0] * px + (f.P.R[1] * py + f.P.R[2] * pz)) + f.P.t[0];
    const double q1 = (f.P.R[3] * px + (f.P.R[4] * py + f.P.R[5] * pz)) + f.P.t[1];
    const double q2 = (f.P.R[6] * px + (f.P.R[7] * py + f.P.R[8] * pz)) + f.P.t[2];
    const uint32_t o = owner_of_block(fast_floor(q0 * f.inv_leaf) >> f.log2, fast_floor(q1 * f.inv_leaf) >> f.log2, fast_floor(q2 * f.inv_leaf) >> f.log2, world);
    if (o != rank) {
      d = o;
      atomicAdd(&s_h[o], 1u);
    }
  }
  f.dest[i] = static_cast<uint8_t>(d);
  __syncthreads();
  if (threadIdx.x < world) f.hist[static_cast<size_t>(blk) * world + threadIdx.x] = s_h[threadIdx.x];
}

// Pass 2: position of every mover inside its destination's segment = movers of earlier blocks + earlier movers of this
// block (slot order: stable, so the arrival order — and with it every sum — is reproducible run to run).  The segment of
// destination d starts at send + d * peer_stride (a round of several factors interleaves their segments per peer).
__device__ __forceinline__ void route_pack_body(const ShardFactorArgs & f, const uint32_t world, const uint32_t rank, const uint32_t blk, const uint32_t nblk)
{
  constexpr int NW = kT / 64;
  __shared__ uint32_t s_base[kShardMaxWorld];
  __shared__ uint32_t s_wc[NW][kShardMaxWorld];
  __shared__ uint32_t s_tot[kShardMaxWorld];
  const ShardArrays & a = f.a;
  const uint32_t cap = f.cap;
  if (threadIdx.x < kShardMaxWorld) {
    s_base[threadIdx.x] = 0u;
    for (int w = 0; w < NW; ++w) s_wc[w][threadIdx.x] = 0u;
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < blk; b += kT)
    for (uint32_t d = 0; d < world; ++d) {
      const uint32_t c = f.hist[static_cast<size_t>(b) * world + d];
      if (c) atomicAdd(&s_base[d], c);
    }
  const uint32_t i = blk * kT + threadIdx.x;
  const uint32_t d = f.dest[i];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t my_rank = 0u;
  {
    unsigned long long todo = __ballot(d != 0xFFu);
    while (todo) {
      const int l0 = __builtin_ctzll(todo);
      const uint32_t d0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(d), l0));
      const unsigned long long m = __ballot(d == d0);
      if (d == d0) my_rank = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
      if (static_cast<int>(lane) == l0) s_wc[wv][d0] = static_cast<uint32_t>(__popcll(m));
      todo &= ~m;
    }
  }
  __syncthreads();
  if (d != 0xFFu) {
    uint32_t pos = s_base[d] + my_rank;
    for (uint32_t w = 0; w < wv; ++w) pos += s_wc[w][d];
    if (pos < cap) {
      ShardRecord r;
      r.src = a.src[i];
      for (int k = 0; k < 3; ++k) {
        r.q_da[k] = a.q_da[3 * static_cast<size_t>(i) + k];
        r.mean[k] = a.mean[3 * static_cast<size_t>(i) + k];
        r.normal[k] = a.normal[3 * static_cast<size_t>(i) + k];
      }
      r.status = a.status[i];
      r.pad = 0;
      r.origin = a.origin[i];
      *reinterpret_cast<ShardRecord *>(f.send + static_cast<size_t>(d) * f.peer_stride + sizeof(ShardHdr) + static_cast<size_t>(pos) * sizeof(ShardRecord)) = r;
      a.origin[i] = kShardTomb;
      a.status[i] = -1;  // carries kShardSkip
    } else {
      a.status[i] |= kShardSkip;  // the segment is full: stays here unprocessed, the caller repeats the call with larger segments
    }
  }
  if (blk == nblk - 1) {
    __syncthreads();
    if (threadIdx.x < kShardMaxWorld) {
      uint32_t t = 0u;
      if (threadIdx.x < world) {
        t = s_base[threadIdx.x];
        for (int w = 0; w < NW; ++w) t += s_wc[w][threadIdx.x];
        ShardHdr h;
        h.sent = t < cap ? t : cap;
        h.total = t;
        h.pad0 = h.pad1 = 0u;
        *reinterpret_cast<ShardHdr *>(f.send + static_cast<size_t>(threadIdx.x) * f.peer_stride) = h;
      }
      s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t sent = 0u, mx = 0u;
      for (uint32_t r = 0; r < world; ++r) {
        sent += s_tot[r] < cap ? s_tot[r] : cap;
        mx = s_tot[r] > mx ? s_tot[r] : mx;
      }
      f.st->sent_total = sent;
      f.st->max_total = mx;
    }
    if (threadIdx.x < kShardMaxWorld) {
      uint32_t mx = 0u;
      for (uint32_t r = 0; r < world; ++r) mx = s_tot[r] > mx ? s_tot[r] : mx;
      f.ar_slots[threadIdx.x] = threadIdx.x == rank ? static_cast<double>(mx) : 0.0;
    }
  }
}

// Arrivals of all peers appended behind the last slot in (peer, record) order; thread t of the factor's append grid takes
// record t % cap of peer t / cap; thread 0 writes the next ping-pong entries of the slot / live counters.
__device__ __forceinline__ void append_body(const ShardFactorArgs & f, const uint32_t world, const uint32_t t)
{
  const ShardArrays & a = f.a;
  const uint32_t cap = f.cap;
  const uint32_t peer = t / cap, j = t - peer * cap;
  const uint32_t n0 = f.st->n_slots[f.cur];
  if (peer < world) {
    uint32_t before = 0u;
    for (uint32_t p = 0; p < peer; ++p) before += reinterpret_cast<const ShardHdr *>(f.recv + p * f.peer_stride)->sent;
    const uint32_t cnt = reinterpret_cast<const ShardHdr *>(f.recv + peer * f.peer_stride)->sent;
    const size_t d = static_cast<size_t>(n0) + before + j;
    if (j < cnt && d < f.slot_capacity) {
      const ShardRecord r = *reinterpret_cast<const ShardRecord *>(f.recv + peer * f.peer_stride + sizeof(ShardHdr) + static_cast<size_t>(j) * sizeof(ShardRecord));
      a.src[d] = r.src;
      for (int k = 0; k < 3; ++k) {
        a.q_da[3 * d + k] = r.q_da[k];
        a.mean[3 * d + k] = r.mean[k];
        a.normal[3 * d + k] = r.normal[k];
      }
      a.status[d] = r.status;
      a.origin[d] = r.origin;
    }
  }
  if (t == 0) {
    uint32_t total = 0u;
    for (uint32_t p = 0; p < world; ++p) total += reinterpret_cast<const ShardHdr *>(f.recv + p * f.peer_stride)->sent;
    const uint32_t room = f.slot_capacity - n0;
    const uint32_t taken = total < room ? total : room;
    if (taken < total) f.st->error |= 1u;
    f.st->n_slots[f.cur ^ 1] = n0 + taken;
    f.st->n_live[f.cur ^ 1] = f.st->n_live[f.cur] - f.st->sent_total + taken;
  }
}

__global__ __launch_bounds__(kT) void shard_route_count_kernel(const ShardFactorArgs f, uint32_t world, uint32_t rank)
{
  route_count_body(f, world, rank, blockIdx.x);
}
__global__ __launch_bounds__(kT) void shard_route_pack_kernel(const ShardFactorArgs f, uint32_t world, uint32_t rank)
{
  route_pack_body(f, world, rank, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(kT) void shard_append_kernel(const ShardFactorArgs f, uint32_t world)
{
  append_body(f, world, blockIdx.x * kT + threadIdx.x);
}

// The batched forms: the argument blocks of up to kShardBatchMax factors ride in the kernel-argument segment (read through
// the segment pointer with a wave-uniform index: scalar loads; indexing the by-value parameter would copy it to scratch);
// a workgroup finds its factor in the prefix table of the per-factor grids.
__device__ __forceinline__ int shard_batch_factor_of(const int * start, int n, int b)
{
  int f = 0;
  for (int i = 1; i < n; ++i) f += (b >= start[i]) ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(f);
}
__global__ __launch_bounds__(kT) void shard_route_count_batch_kernel(const ShardBatch blk)
{
  (void)blk;
  const auto * p = (const ShardBatch *)__builtin_amdgcn_kernarg_segment_ptr();
  const int b = static_cast<int>(blockIdx.x);
  const int f = shard_batch_factor_of(p->route_start, p->n, b);
  route_count_body(p->f[f], p->world, p->rank, static_cast<uint32_t>(b - p->route_start[f]));
}
__global__ __launch_bounds__(kT) void shard_route_pack_batch_kernel(const ShardBatch blk)
{
  (void)blk;
  const auto * p = (const ShardBatch *)__builtin_amdgcn_kernarg_segment_ptr();
  const int b = static_cast<int>(blockIdx.x);
  const int f = shard_batch_factor_of(p->route_start, p->n, b);
  route_pack_body(p->f[f], p->world, p->rank, static_cast<uint32_t>(b - p->route_start[f]), static_cast<uint32_t>(p->route_start[f + 1] - p->route_start[f]));
}
__global__ __launch_bounds__(kT) void shard_append_batch_kernel(const ShardBatch blk)
{
  (void)blk;
  const auto * p = (const ShardBatch *)__builtin_amdgcn_kernarg_segment_ptr();
  const int b = static_cast<int>(blockIdx.x);
  const int f = shard_batch_factor_of(p->append_start, p->n, b);
  append_body(p->f[f], p->world, static_cast<uint32_t>(b - p->append_start[f]) * kT + threadIdx.x);
}

__global__ __launch_bounds__(kT) void shard_live_flags_kernel(const ShardArrays a, const ShardState * st, int cur, uint32_t n_bound, uint32_t * flags)
{
  const uint32_t n = st->n_slots[cur];
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n_bound; i += gridDim.x * kT) flags[i] = (i < n && a.origin[i] != kShardTomb) ? 1u : 0u;
}
__global__ __launch_bounds__(kT) void shard_compact_slots_kernel(const ShardArrays in, const ShardArrays out, ShardState * st, uint32_t n_bound, const uint32_t * flags,
                                                                  const uint32_t * pos)
{
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n_bound; i += gridDim.x * kT) {
    if (flags[i]) {
      const size_t d = pos[i];
      out.src[d] = in.src[i];
      for (int k = 0; k < 3; ++k) {
        out.q_da[3 * d + k] = in.q_da[3 * static_cast<size_t>(i) + k];
        out.mean[3 * d + k] = in.mean[3 * static_cast<size_t>(i) + k];
        out.normal[3 * d + k] = in.normal[3 * static_cast<size_t>(i) + k];
      }
      out.status[d] = in.status[i];
      out.origin[d] = in.origin[i];
    }
    if (i == n_bound - 1) {
      const uint32_t live = pos[i] + flags[i];
      st->n_slots[0] = st->n_slots[1] = live;
      st->n_live[0] = st->n_live[1] = live;
    }
  }
}

__global__ __launch_bounds__(kT) void shard_reset_kernel(const ShardArrays a, const ShardState * st, int cur, uint32_t n_bound)
{
  const uint32_t n = st->n_slots[cur];
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n_bound && i < n; i += gridDim.x * kT) {
    if (a.origin[i] == kShardTomb) continue;
    a.status[i] = 0;
    for (int k = 0; k < 3; ++k) a.q_da[3 * static_cast<size_t>(i) + k] = a.mean[3 * static_cast<size_t>(i) + k] = a.normal[3 * static_cast<size_t>(i) + k] = 0.0;
  }
}

// One workgroup per factor of the round: the factor's slice of the all-reduced vectors, its counters, then ITS completion
// flag (the host waits for every factor's flag of the round).
__device__ __forceinline__ void publish_body(const double * ar, const double * loc, const ShardState * st, int next, ShardPublish * host, uint32_t seq)
{
  const int t = threadIdx.x;
  if (t < kShardArLen) host->ar[t] = ar[t];
  if (t < 16) host->loc[t] = loc ? loc[t] : 0.0;
  if (t == 0) {
    host->n_slots = st->n_slots[next];
    host->n_live = st->n_live[next];
    host->error = st->error;
    host->max_total = st->max_total;
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void shard_publish_kernel(const double * ar, const double * loc, const ShardState * st, int next, ShardPublish * host, uint32_t seq)
{
  publish_body(ar, loc, st, next, host, seq);
}
__global__ void shard_publish_batch_kernel(const ShardPublishBatch blk)
{
  (void)blk;
  const auto * p = (const ShardPublishBatch *)__builtin_amdgcn_kernarg_segment_ptr();
  const int f = static_cast<int>(blockIdx.x);
  publish_body(p->ar + static_cast<size_t>(f) * kShardArLen, p->loc ? p->loc + static_cast<size_t>(f) * 16 : nullptr, p->st[f], p->next[f], p->host + f, p->seq);
}
}  // namespace

hipError_t launch_shard_state_init(ShardState * st, uint32_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_state_init_kernel, dim3(1), dim3(64), 0, stream, st, n);
  return hipGetLastError();
}
static uint32_t route_blocks(uint32_t n_bound) { return n_bound ? (n_bound + kT - 1) / kT : 1u; }  // at least one block: the headers must be written
hipError_t launch_shard_route(const ShardFactorArgs & f, uint32_t n_bound, uint32_t world, uint32_t rank, hipStream_t stream)
{
  const uint32_t nblk = route_blocks(n_bound);
  hipLaunchKernelGGL(shard_route_count_kernel, dim3(nblk), dim3(kT), 0, stream, f, world, rank);
  hipLaunchKernelGGL(shard_route_pack_kernel, dim3(nblk), dim3(kT), 0, stream, f, world, rank);
  return hipGetLastError();
}
hipError_t launch_shard_append(const ShardFactorArgs & f, uint32_t world, hipStream_t stream)
{
  const uint32_t threads = world * f.cap;
  hipLaunchKernelGGL(shard_append_kernel, dim3((threads + kT - 1) / kT), dim3(kT), 0, stream, f, world);
  return hipGetLastError();
}
void shard_batch_grids(ShardBatch & b, const uint32_t * n_bound)
{
  int r = 0, a = 0;
  for (int i = 0; i < b.n; ++i) {
    b.route_start[i] = r;
    b.append_start[i] = a;
    r += static_cast<int>(route_blocks(n_bound[i]));
    a += static_cast<int>((b.world * b.f[i].cap + kT - 1) / kT);
  }
  b.route_start[b.n] = r;
  b.append_start[b.n] = a;
}
hipError_t launch_shard_route_batch(const ShardBatch & b, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_route_count_batch_kernel, dim3(b.route_start[b.n]), dim3(kT), 0, stream, b);
  hipLaunchKernelGGL(shard_route_pack_batch_kernel, dim3(b.route_start[b.n]), dim3(kT), 0, stream, b);
  return hipGetLastError();
}
hipError_t launch_shard_append_batch(const ShardBatch & b, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_append_batch_kernel, dim3(b.append_start[b.n]), dim3(kT), 0, stream, b);
  return hipGetLastError();
}
hipError_t launch_shard_compact(const ShardArrays & in, const ShardArrays & out, ShardState * st, int cur, uint32_t n_bound, uint32_t * flags, uint32_t * pos,
                                void * temp, size_t temp_bytes, hipStream_t stream)
{
  if (!n_bound) return hipSuccess;
  hipLaunchKernelGGL(shard_live_flags_kernel, dim3(grid_for(n_bound)), dim3(kT), 0, stream, in, st, cur, n_bound, flags);
  size_t tb = temp_bytes;
  const hipError_t e = rocprim::exclusive_scan(temp, tb, flags, pos, 0u, static_cast<size_t>(n_bound), rocprim::plus<uint32_t>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(shard_compact_slots_kernel, dim3(grid_for(n_bound)), dim3(kT), 0, stream, in, out, st, n_bound, flags, pos);
  return hipGetLastError();
}
hipError_t launch_shard_reset(const ShardArrays & a, const ShardState * st, int cur, uint32_t n_bound, hipStream_t stream)
{
  if (n_bound) hipLaunchKernelGGL(shard_reset_kernel, dim3(grid_for(n_bound)), dim3(kT), 0, stream, a, st, cur, n_bound);
  return hipGetLastError();
}
hipError_t launch_shard_publish(const double * ar, const double * loc, const ShardState * st, int next, ShardPublish * host, uint32_t seq, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_publish_kernel, dim3(1), dim3(192), 0, stream, ar, loc, st, next, host, seq);
  return hipGetLastError();
}
hipError_t launch_shard_publish_batch(const ShardPublishBatch & b, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_publish_batch_kernel, dim3(b.n), dim3(192), 0, stream, b);
  return hipGetLastError();
}

size_t shard_temp_bytes(size_t n)
{
  if (n == 0) n = 1;
  size_t a = 0, b = 0;
  uint32_t * k = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, a, k, k, k, k, n, 0, 8, hipStream_t(nullptr));
  (void)rocprim::exclusive_scan(nullptr, b, k, k, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(nullptr));
  return (a > b ? a : b) + 256;
}

hipError_t launch_shard_filter(const float * xyz, uint32_t n, uint32_t stride, double inv_leaf, uint32_t world, uint32_t rank, int log2,
                               uint32_t * flags, uint32_t * pos, float * out, uint32_t * n_out, void * temp, size_t temp_bytes, hipStream_t stream)
{
  if (!n) return hipMemsetAsync(n_out, 0, sizeof(uint32_t), stream);
  hipLaunchKernelGGL(shard_filter_kernel, dim3(grid_for(n)), dim3(kT), 0, stream, xyz, n, stride, inv_leaf, world, rank, log2, flags);
  size_t tb = temp_bytes;
  const hipError_t e = rocprim::exclusive_scan(temp, tb, flags, pos, 0u, static_cast<size_t>(n), rocprim::plus<uint32_t>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(shard_compact_kernel, dim3(grid_for(n)), dim3(kT), 0, stream, xyz, n, stride, flags, pos, out, n_out);
  return hipGetLastError();
}


hipError_t launch_shard_origin(unsigned long long * origin, uint32_t n, uint32_t rank, hipStream_t stream)
{
  if (n) hipLaunchKernelGGL(shard_origin_kernel, dim3(grid_for(n)), dim3(kT), 0, stream, origin, n, rank);
  return hipGetLastError();
}

}  // namespace mh
