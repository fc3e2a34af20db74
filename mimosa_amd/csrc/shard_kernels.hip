// HIP kernels of the map-sharded scan-to-map factor (BASELINE configs[2]): the map is partitioned by
// spatial hash across the GPUs of a node, every source point is linearized on the rank that owns its centre voxel.
// The reference is single-process (no counterpart); what these kernels must preserve is that the SHARDED result equals
// the unsharded ICPFactor::linearize (include/mimosa/lidar/geometric_factor.hpp:231-562) bit for bit per point.
//
// Partition: shard blocks of 2^log2 voxels per axis, owner = XORVector3iHash(block) mod world (the reference's hash,
// include/mimosa/lidar/utils.hpp:228-238).  Every rank also stores the one-voxel halo of its blocks, so the 1/7/19/27
// neighbourhood of a query in an owned block is complete locally.
//   shard_filter   which points of an insert batch this rank keeps (owned blocks + halo), order preserved
//   shard_owner    owner rank of every local source point at the CURRENT pose (q = R p + t in fp64, like :276-277)
//   shard_pack     points that changed owner leave with their data-association state (112-byte records grouped by
//                  destination, stable), the rest is compacted in place (stable)
//   shard_unpack   arrivals are appended
// Streaming / sorting over <= 131 072 records per rank: HBM- and launch-bound, no MFMA.
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

// XORVector3iHash (include/mimosa/lidar/utils.hpp:228-238): size_t products of the coordinates with three 64-bit primes
__device__ __forceinline__ uint32_t owner_of_block(int bx, int by, int bz, uint32_t world)
{
  const unsigned long long h = (static_cast<unsigned long long>(static_cast<long long>(bx)) * 9132043225175502913ull) ^
                               (static_cast<unsigned long long>(static_cast<long long>(by)) * 7277549399757405689ull) ^
                               (static_cast<unsigned long long>(static_cast<long long>(bz)) * 6673468629021231217ull);
  return static_cast<uint32_t>(h % world);
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

// key = destination rank for a point that leaves, `world` for one that stays; counts[r] = points bound for rank r
__global__ __launch_bounds__(kT) void shard_owner_kernel(const ShardPose P, const float4 * src, uint32_t n, double inv_leaf, uint32_t world,
                                                          uint32_t rank, int log2, uint32_t * keys, uint32_t * idx, uint32_t * counts)
{
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) {
    const float4 sp = src[i];
    const double px = sp.x, py = sp.y, pz = sp.z;
    const double q0 = (P.R[0] * px + (P.R[1] * py + P.R[2] * pz)) + P.t[0];
    const double q1 = (P.R[3] * px + (P.R[4] * py + P.R[5] * pz)) + P.t[1];
    const double q2 = (P.R[6] * px + (P.R[7] * py + P.R[8] * pz)) + P.t[2];
    const uint32_t o = owner_of_block(fast_floor(q0 * inv_leaf) >> log2, fast_floor(q1 * inv_leaf) >> log2, fast_floor(q2 * inv_leaf) >> log2, world);
    keys[i] = o == rank ? world : o;
    idx[i] = i;
    if (o != rank) atomicAdd(&counts[o], 1u);
  }
}

__global__ __launch_bounds__(kT) void shard_origin_kernel(unsigned long long * origin, uint32_t n, uint32_t rank)
{
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) origin[i] = (static_cast<unsigned long long>(rank) << 32) | i;
}

// sorted position s < n_movers: record s of the send buffer; otherwise the point moves to slot s - n_movers of the
// compacted arrays (`out`).
__global__ __launch_bounds__(kT) void shard_pack_kernel(const ShardArrays in, const ShardArrays out, const uint32_t * sorted_idx, uint32_t n,
                                                         uint32_t n_movers, ShardRecord * send)
{
  for (uint32_t s = blockIdx.x * kT + threadIdx.x; s < n; s += gridDim.x * kT) {
    const uint32_t i = sorted_idx[s];
    if (s < n_movers) {
      ShardRecord r;
      r.src = in.src[i];
      for (int k = 0; k < 3; ++k) {
        r.q_da[k] = in.q_da[3 * static_cast<size_t>(i) + k];
        r.mean[k] = in.mean[3 * static_cast<size_t>(i) + k];
        r.normal[k] = in.normal[3 * static_cast<size_t>(i) + k];
      }
      r.status = in.status[i];
      r.pad = 0;
      r.origin = in.origin[i];
      send[s] = r;
    } else {
      const size_t d = s - n_movers;
      out.src[d] = in.src[i];
      for (int k = 0; k < 3; ++k) {
        out.q_da[3 * d + k] = in.q_da[3 * static_cast<size_t>(i) + k];
        out.mean[3 * d + k] = in.mean[3 * static_cast<size_t>(i) + k];
        out.normal[3 * d + k] = in.normal[3 * static_cast<size_t>(i) + k];
      }
      out.status[d] = in.status[i];
      out.origin[d] = in.origin[i];
    }
  }
}

__global__ __launch_bounds__(kT) void shard_unpack_kernel(const ShardArrays a, uint32_t base, const ShardRecord * recv, uint32_t n_recv)
{
  for (uint32_t j = blockIdx.x * kT + threadIdx.x; j < n_recv; j += gridDim.x * kT) {
    const ShardRecord r = recv[j];
    const size_t d = static_cast<size_t>(base) + j;
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

__global__ void shard_pack_sums_kernel(const DeviceResult * r, double * out32)
{
  const int i = threadIdx.x;
  if (i < 28) out32[i] = r->sums[i];
  if (i == 28) out32[28] = static_cast<double>(r->n_knn);
  if (i == 29) out32[29] = static_cast<double>(r->n_cand);
  if (i == 30) out32[30] = static_cast<double>(r->n_fallback);
  if (i == 31) out32[31] = static_cast<double>(r->n_scanned);
}
__global__ void shard_eig_kernel(const double * g, double * eig18)
{
  if (threadIdx.x != 0 && threadIdx.x != 64) return;
  const int o = threadIdx.x ? 3 : 0;  // rot block first (GTSAM Pose3 tangent order), one wave each
  double Hb[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const int rr = (r < c ? r : c) + o, cc = (r < c ? c : r) + o;
      Hb[3 * r + c] = g[rr * 7 - rr * (rr - 1) / 2 + (cc - rr)];  // upper triangle of the 7 x 7 sum v v^T
    }
  double loc[3], E[9];
  compute_localizability(Hb, loc, E);
  for (int i = 0; i < 9; ++i) eig18[(o ? 9 : 0) + i] = E[i];
}
__global__ void shard_pack_loc_kernel(const DeviceResult * r, double * out16)
{
  const int i = threadIdx.x;
  if (i < 6) out16[i] = r->loc_comp[i];
  if (i >= 6 && i < 15) out16[i] = static_cast<double>(r->status_hist[i - 6]);
  if (i == 15) out16[15] = 0.0;
}
}  // namespace

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

hipError_t launch_shard_plan(const ShardPose & P, const float4 * src, uint32_t n, double inv_leaf, uint32_t world, uint32_t rank, int log2,
                             uint32_t * keys_a, uint32_t * keys_b, uint32_t * idx_a, uint32_t * idx_b, uint32_t * counts, void * temp,
                             size_t temp_bytes, hipStream_t stream)
{
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(uint32_t) * world, stream);
  if (e != hipSuccess || !n) return e;
  hipLaunchKernelGGL(shard_owner_kernel, dim3(grid_for(n)), dim3(kT), 0, stream, P, src, n, inv_leaf, world, rank, log2, keys_a, idx_a, counts);
  size_t tb = temp_bytes;
  // stable: movers grouped by destination in their original order, then the points that stay, in their original order
  return rocprim::radix_sort_pairs(temp, tb, keys_a, keys_b, idx_a, idx_b, static_cast<size_t>(n), 0, 8, stream);
}

hipError_t launch_shard_origin(unsigned long long * origin, uint32_t n, uint32_t rank, hipStream_t stream)
{
  if (n) hipLaunchKernelGGL(shard_origin_kernel, dim3(grid_for(n)), dim3(kT), 0, stream, origin, n, rank);
  return hipGetLastError();
}
hipError_t launch_shard_pack(const ShardArrays & in, const ShardArrays & out, const uint32_t * sorted_idx, uint32_t n, uint32_t n_movers,
                             ShardRecord * send, hipStream_t stream)
{
  if (n) hipLaunchKernelGGL(shard_pack_kernel, dim3(grid_for(n)), dim3(kT), 0, stream, in, out, sorted_idx, n, n_movers, send);
  return hipGetLastError();
}
hipError_t launch_shard_unpack(const ShardArrays & a, uint32_t base, const ShardRecord * recv, uint32_t n_recv, hipStream_t stream)
{
  if (n_recv) hipLaunchKernelGGL(shard_unpack_kernel, dim3(grid_for(n_recv)), dim3(kT), 0, stream, a, base, recv, n_recv);
  return hipGetLastError();
}
hipError_t launch_shard_pack_sums(const DeviceResult * r, double * out32, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_pack_sums_kernel, dim3(1), dim3(64), 0, stream, r, out32);
  return hipGetLastError();
}
hipError_t launch_shard_eig(const double * global32, double * eig18, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_eig_kernel, dim3(1), dim3(128), 0, stream, global32, eig18);
  return hipGetLastError();
}
hipError_t launch_shard_pack_loc(const DeviceResult * r, double * out16, hipStream_t stream)
{
  hipLaunchKernelGGL(shard_pack_loc_kernel, dim3(1), dim3(64), 0, stream, r, out16);
  return hipGetLastError();
}

}  // namespace mh
