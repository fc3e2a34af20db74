// HIP kernels of the device-resident scan front end (SURVEY.md §8 rows a2, a4, a5 / "next" row f-3): the
// raw Ouster cloud is uploaded once and stays on the device through input filter -> deskew -> body-frame
// subset -> voxel down-sampler -> ICP factor source.
//
// Reference:
//   Manager::prepareInput   src/lidar/manager.cpp:244-336 (filter chain, points_full_, geometric subset),
//                           :340-368 (distinct timestamps)
//   Geometric::preprocess   src/lidar/geometric.cpp:154-161 (subset copy + f32 body transform)
//   Geometric::downsample   src/lidar/geometric.cpp:55-126 + FlatContainerMinimal::add
//                           include/mimosa/lidar/utils.hpp:260-278 (greedy per-voxel min-distance filter)
//
// All of it is order-dependent sequential code in the reference; the device forms reproduce the SAME outputs
// in the same order:
//   * filter: per-point predicate -> exclusive scan -> scatter (an order-preserving compaction);
//   * down-sampler: the greedy rule only couples points of one voxel, in input order.  A stable radix sort
//     by voxel key makes every voxel a contiguous segment still in input order; one thread walks each
//     segment exactly like FlatContainerMinimal::add; the output order "voxels in first-seen order, points
//     in acceptance order" is the ascending order of (first input index of the voxel, input index), one
//     more radix sort of the kept points.
// HBM-bound streaming / sorting work (32 B records, <= 131 072 of them): no MFMA, no LDS tiling to speak of.
// Compiled with -ffp-contract=off: the reference is a baseline x86-64 build (no FMA) and both the range
// filter and the voxel assignment are threshold tests on these f32 / f64 values.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "scan_device.hpp"
#include "voxel_map.hpp"

namespace mh
{
namespace
{
constexpr int kThreads = 256;
constexpr uint32_t kNoKey32 = 0xFFFFFFFFu;
constexpr uint64_t kNoKey64 = ~0ull;

int grid_for(uint32_t n) { return static_cast<int>(max(1u, min((n + kThreads - 1) / kThreads, 4096u))); }

// ---- prepareInput ------------------------------------------------------------------------------------
struct FilterParams
{
  float range_min_sq, range_max_sq, intensity_min, intensity_max, ns_max, z_offset;
  uint32_t stride, point_skip, ring_skip;
};

__device__ __forceinline__ float range_sq_of(const mh_ouster_point & p) { return p.x * p.x + p.y * p.y + p.z * p.z; }

__global__ __launch_bounds__(kThreads) void input_filter_kernel(const mh_ouster_point * raw, uint32_t n, FilterParams f,
                                                                 uint32_t * flag_full, uint32_t * flag_geo,
                                                                 ScanCounters * counters)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const mh_ouster_point p = raw[i];
    bool keep = (i % f.stride) == 0;                                                   // :246 loop stride
    keep = keep && !(isnan(p.x) || isnan(p.y) || isnan(p.z));                           // :253
    keep = keep && !(isnan(p.intensity) || p.intensity < f.intensity_min || p.intensity > f.intensity_max);  // :272-276
    const float r2 = range_sq_of(p);
    keep = keep && !(r2 < f.range_min_sq || r2 > f.range_max_sq);                       // :281-282
    keep = keep && !(static_cast<float>(p.t) > f.ns_max);                               // :306 (uint32 promoted to float)
    flag_full[i] = keep ? 1u : 0u;
    // :318-334 point-skip and ring filters select the geometric subset
    flag_geo[i] = (keep && (i % f.point_skip) == 0 && (p.ring % f.ring_skip) == 0) ? 1u : 0u;
    // (:310 last_point_ns = max t over the kept points falls out of the timestamp sort: unique_scatter_kernel)
  }
}

__global__ __launch_bounds__(kThreads) void input_scatter_kernel(const mh_ouster_point * raw, uint32_t n, FilterParams f,
                                                                  const uint32_t * flag_full, const uint32_t * flag_geo,
                                                                  const uint32_t * pos_full, const uint32_t * pos_geo,
                                                                  mh_point32 * points_full, uint32_t * geo_idx,
                                                                  ScanCounters * counters)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    if (flag_full[i]) {
      const mh_ouster_point p = raw[i];
      mh_point32 o;
      o.x = p.x;
      o.y = p.y;
      o.z = p.z + f.z_offset;
      o.pad = 0.f;
      o.intensity = p.intensity;
      o.t = p.t;
      o.idx = i;
      // :312-313 std::sqrt(float): correctly rounded.  sqrt in double then one rounding to float is exact for
      // that (53 >= 2 * 24 + 2 bits) and does not depend on how the compiler lowers f32 sqrt
      o.range = static_cast<float>(sqrt(static_cast<double>(range_sq_of(p))));
      points_full[pos_full[i]] = o;
      if (flag_geo[i]) geo_idx[pos_geo[i]] = pos_full[i];
    }
    if (i == n - 1) {
      counters->n_full = pos_full[i] + flag_full[i];
      counters->n_geometric = pos_geo[i] + flag_geo[i];
    }
  }
}

// ---- distinct timestamps --------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void ns_keys_kernel(const mh_point32 * pts, const ScanCounters * c, uint32_t n_cap,
                                                            uint32_t * keys)
{
  const uint32_t n = c->n_full;
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_cap; i += gridDim.x * kThreads)
    keys[i] = i < n ? pts[i].t : kNoKey32;
}
__global__ __launch_bounds__(kThreads) void head_flags32_kernel(const uint32_t * keys, uint32_t n, uint32_t * flags)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
    flags[i] = (keys[i] != kNoKey32 && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}
__global__ __launch_bounds__(kThreads) void unique_scatter_kernel(const uint32_t * keys, const uint32_t * flags,
                                                                   const uint32_t * pos, uint32_t n, uint32_t * out,
                                                                   ScanCounters * c)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    if (flags[i]) out[pos[i]] = keys[i];
    if (i == n - 1) c->n_unique_ns = pos[i] + flags[i];
    // the largest kept timestamp = the last non-sentinel key of the sorted list (manager.cpp:310)
    if (keys[i] != kNoKey32 && (i == n - 1 || keys[i + 1] == kNoKey32)) c->last_point_ns = keys[i];
  }
}

// ---- Geometric::preprocess --------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void gather_transform_kernel(const mh_point32 * pts, const uint32_t * geo_idx,
                                                                     uint32_t n, const float * Rt12, mh_point32 * body)
{
  __shared__ float P[12];
  if (threadIdx.x < 12) P[threadIdx.x] = Rt12[threadIdx.x];
  __syncthreads();
  for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < n; j += gridDim.x * kThreads) {
    mh_point32 p = pts[geo_idx[j]];
    const float px = p.x, py = p.y, pz = p.z;  // Eigen's coefficient order r0*x + (r1*y + r2*z), then + t
    p.x = (P[0] * px + (P[1] * py + P[2] * pz)) + P[9];
    p.y = (P[3] * px + (P[4] * py + P[5] * pz)) + P[10];
    p.z = (P[6] * px + (P[7] * py + P[8] * pz)) + P[11];
    body[j] = p;
  }
}

// ---- Geometric::downsample --------------------------------------------------------------------------------
constexpr int kCoordBits = 21;
constexpr int kCoordBias = 1 << (kCoordBits - 1);

__global__ __launch_bounds__(kThreads) void voxel_keys_kernel(const mh_point32 * pts, uint32_t n, double inv_leaf,
                                                               uint64_t * keys, uint32_t * idx, ScanCounters * c)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const mh_point32 p = pts[i];
    // :77-80 coord = fast_floor(double(p) * inv_leaf)
    const int cx = fast_floor(static_cast<double>(p.x) * inv_leaf), cy = fast_floor(static_cast<double>(p.y) * inv_leaf),
              cz = fast_floor(static_cast<double>(p.z) * inv_leaf);
    const int bx = cx + kCoordBias, by = cy + kCoordBias, bz = cz + kCoordBias;
    if (((bx | by | bz) >> kCoordBits) != 0) atomicOr(&c->bad_coord, 1u);
    keys[i] = (static_cast<uint64_t>(bx & ((1 << kCoordBits) - 1)) << (2 * kCoordBits)) |
              (static_cast<uint64_t>(by & ((1 << kCoordBits) - 1)) << kCoordBits) |
              static_cast<uint64_t>(bz & ((1 << kCoordBits) - 1));
    idx[i] = i;
  }
}

__global__ __launch_bounds__(kThreads) void head_flags64_kernel(const uint64_t * keys, uint32_t n, uint32_t * flags)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// seg_start[v] = first sorted position of voxel v; seg_start[n_voxels] = n
__global__ __launch_bounds__(kThreads) void segment_starts_kernel(const uint32_t * flags, const uint32_t * pos, uint32_t n,
                                                                   uint32_t * seg_start, ScanCounters * c)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    if (flags[i]) seg_start[pos[i]] = i;
    if (i == n - 1) {
      const uint32_t nv = pos[i] + flags[i];
      c->n_voxels = nv;
      seg_start[nv] = n;
    }
  }
}

// One WAVE per voxel: FlatContainerMinimal::add over the voxel's points in input order.  Lane j holds the j-th
// point kept so far (<= 20); every incoming point is tested against all of them at once (one fp64 distance per
// lane, one ballot) — a thread-per-voxel walk of the same lists was a chain of dependent scattered loads and took
// 250 us, 44 % of the whole front end.  keep[] is indexed by sorted position; first_idx[s] = input index of the
// first point of the voxel of position s.
__global__ __launch_bounds__(kThreads) void greedy_voxel_kernel(const mh_point32 * pts, const uint32_t * sorted_idx,
                                                                 const uint32_t * seg_start, const ScanCounters * c,
                                                                 uint32_t max_pts, double min_sq, uint32_t * keep,
                                                                 uint32_t * first_idx)
{
  const uint32_t nv = c->n_voxels;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * kThreads + threadIdx.x) >> 6, n_waves = (gridDim.x * kThreads) >> 6;
  const uint32_t cap = min(max_pts, static_cast<uint32_t>(kBucketStride));  // utils.hpp:262 size cap
  for (uint32_t v = wave; v < nv; v += n_waves) {
    const uint32_t s0 = seg_start[v], s1 = seg_start[v + 1];
    const uint32_t first = sorted_idx[s0];  // stable sort: the smallest input index of the voxel
    double kx = 0.0, ky = 0.0, kz = 0.0;    // this lane's kept point (lane < n_kept)
    uint32_t n_kept = 0;
    for (uint32_t base = s0; base < s1; base += 64u) {  // the segment, 64 points at a time (coalesced index loads)
      const uint32_t s = base + lane;
      float px = 0.f, py = 0.f, pz = 0.f;
      if (s < s1) {
        const mh_point32 p = pts[sorted_idx[s]];
        px = p.x;
        py = p.y;
        pz = p.z;
        first_idx[s] = first;
      }
      const uint32_t m = min(64u, s1 - base);
      uint64_t kept_mask = 0;
      for (uint32_t u = 0; u < m; ++u) {  // input order
        const double qx = static_cast<double>(__shfl(px, static_cast<int>(u))), qy = static_cast<double>(__shfl(py, static_cast<int>(u))),
                     qz = static_cast<double>(__shfl(pz, static_cast<int>(u)));
        const double dx = kx - qx, dy = ky - qy, dz = kz - qz;
        // Vector3d squaredNorm: p0 + (p1 + p2); utils.hpp:266-272
        const bool close = lane < n_kept && dx * dx + (dy * dy + dz * dz) < min_sq;
        const bool take = n_kept < cap && __ballot(close) == 0ull;
        if (take) {
          if (lane == n_kept) {
            kx = qx;
            ky = qy;
            kz = qz;
          }
          ++n_kept;
          kept_mask |= 1ull << u;
        }
      }
      if (s < s1) keep[s] = static_cast<uint32_t>((kept_mask >> lane) & 1ull);
    }
  }
}

// kept point at sorted position s -> key (first index of its voxel, own input index): ascending order of
// these keys is "voxels in first-seen order, points in acceptance order" (geometric.cpp:103-109)
__global__ __launch_bounds__(kThreads) void order_keys_kernel(const uint32_t * sorted_idx, const uint32_t * keep,
                                                               const uint32_t * pos, const uint32_t * first_idx, uint32_t n,
                                                               uint64_t * keys, ScanCounters * c)
{
  for (uint32_t s = blockIdx.x * kThreads + threadIdx.x; s < n; s += gridDim.x * kThreads) {
    if (keep[s]) keys[pos[s]] = (static_cast<uint64_t>(first_idx[s]) << 32) | sorted_idx[s];
    if (s == n - 1) c->n_downsampled = pos[s] + keep[s];
  }
}
__global__ __launch_bounds__(kThreads) void fill64_kernel(uint64_t * keys, uint32_t n, uint64_t v)
{
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) keys[i] = v;
}
__global__ __launch_bounds__(kThreads) void gather_kept_kernel(const mh_point32 * pts, const uint64_t * keys,
                                                                const ScanCounters * c, uint32_t n_cap, uint32_t * kept_idx,
                                                                mh_point32 * out)
{
  const uint32_t n = c->n_downsampled;
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_cap; i += gridDim.x * kThreads) {
    if (i >= n) continue;
    const uint32_t j = static_cast<uint32_t>(keys[i] & 0xFFFFFFFFull);
    kept_idx[i] = j;
    out[i] = pts[j];
  }
}

hipError_t exclusive_sum(const uint32_t * in, uint32_t * out, uint32_t n, void * temp, size_t temp_bytes, hipStream_t stream)
{
  size_t tb = temp_bytes;
  return rocprim::exclusive_scan(temp, tb, in, out, 0u, static_cast<size_t>(n), rocprim::plus<uint32_t>(), stream);
}
}  // namespace

size_t scan_temp_bytes(size_t n)
{
  if (n == 0) n = 1;
  size_t best = 0, tb = 0;
  uint32_t * k32 = nullptr;
  uint64_t * k64 = nullptr;
  (void)rocprim::radix_sort_keys(nullptr, tb, k32, k32, n, 0, 32, hipStream_t(nullptr));
  best = tb > best ? tb : best;
  tb = 0;
  (void)rocprim::radix_sort_keys(nullptr, tb, k64, k64, n, 0, 64, hipStream_t(nullptr));
  best = tb > best ? tb : best;
  tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, k64, k64, k32, k32, n, 0, 64, hipStream_t(nullptr));
  best = tb > best ? tb : best;
  tb = 0;
  (void)rocprim::exclusive_scan(nullptr, tb, k32, k32, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(nullptr));
  best = tb > best ? tb : best;
  return best + 256;
}

hipError_t launch_input_filter(const mh_ouster_point * raw, uint32_t n, const mh_input_config & cfg, uint32_t * flag_full,
                               uint32_t * flag_geo, uint32_t * pos_full, uint32_t * pos_geo, mh_point32 * points_full,
                               uint32_t * geo_idx, ScanCounters * counters, void * temp, size_t temp_bytes,
                               hipStream_t stream)
{
  FilterParams f;
  f.range_min_sq = cfg.range_min * cfg.range_min;  // manager.cpp:19-20 (float products)
  f.range_max_sq = cfg.range_max * cfg.range_max;
  f.intensity_min = cfg.intensity_min;
  f.intensity_max = cfg.intensity_max;
  f.ns_max = cfg.ns_max;
  f.z_offset = cfg.z_offset;
  f.point_skip = static_cast<uint32_t>(cfg.point_skip_divisor > 0 ? cfg.point_skip_divisor : 1);
  f.ring_skip = static_cast<uint32_t>(cfg.ring_skip_divisor > 0 ? cfg.ring_skip_divisor : 1);
  f.stride = cfg.create_full_res_pointcloud ? 1u : f.point_skip;
  hipError_t e = hipMemsetAsync(counters, 0, sizeof(ScanCounters), stream);
  if (e != hipSuccess || n == 0) return e;
  hipLaunchKernelGGL(input_filter_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, raw, n, f, flag_full, flag_geo,
                     counters);
  if ((e = exclusive_sum(flag_full, pos_full, n, temp, temp_bytes, stream)) != hipSuccess) return e;
  if ((e = exclusive_sum(flag_geo, pos_geo, n, temp, temp_bytes, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(input_scatter_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, raw, n, f, flag_full, flag_geo,
                     pos_full, pos_geo, points_full, geo_idx, counters);
  return hipGetLastError();
}

hipError_t launch_unique_ns(const mh_point32 * points_full, const ScanCounters * counters, uint32_t n_cap, uint32_t * keys_a,
                            uint32_t * keys_b, uint32_t * flags, uint32_t * pos, uint32_t * unique_ns,
                            ScanCounters * counters_out, void * temp, size_t temp_bytes, hipStream_t stream)
{
  if (n_cap == 0) return hipSuccess;
  hipLaunchKernelGGL(ns_keys_kernel, dim3(grid_for(n_cap)), dim3(kThreads), 0, stream, points_full, counters, n_cap, keys_a);
  size_t tb = temp_bytes;
  hipError_t e = rocprim::radix_sort_keys(temp, tb, keys_a, keys_b, static_cast<size_t>(n_cap), 0, 32, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(head_flags32_kernel, dim3(grid_for(n_cap)), dim3(kThreads), 0, stream, keys_b, n_cap, flags);
  if ((e = exclusive_sum(flags, pos, n_cap, temp, temp_bytes, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(unique_scatter_kernel, dim3(grid_for(n_cap)), dim3(kThreads), 0, stream, keys_b, flags, pos, n_cap,
                     unique_ns, counters_out);
  return hipGetLastError();
}

hipError_t launch_gather_transform(const mh_point32 * points_full, const uint32_t * geo_idx, uint32_t n_geo,
                                   const float * Rt12, mh_point32 * body, hipStream_t stream)
{
  if (n_geo == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_transform_kernel, dim3(grid_for(n_geo)), dim3(kThreads), 0, stream, points_full, geo_idx, n_geo,
                     Rt12, body);
  return hipGetLastError();
}

hipError_t launch_downsample(const mh_point32 * body, uint32_t n, double leaf, uint32_t max_pts, double min_dist,
                             uint64_t * keys_a, uint64_t * keys_b, uint32_t * idx_a, uint32_t * idx_b, uint32_t * flags,
                             uint32_t * pos, uint32_t * seg_start, uint32_t * first_idx, uint32_t * kept_idx,
                             mh_point32 * out, ScanCounters * counters, void * temp, size_t temp_bytes, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  const double inv_leaf = 1.0 / leaf;            // geometric.cpp:61
  const double min_sq = min_dist * min_dist;     // :63
  const dim3 g(grid_for(n)), b(kThreads);
  hipLaunchKernelGGL(voxel_keys_kernel, g, b, 0, stream, body, n, inv_leaf, keys_a, idx_a, counters);
  size_t tb = temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(temp, tb, keys_a, keys_b, idx_a, idx_b, static_cast<size_t>(n), 0, 3 * kCoordBits,
                                           stream);  // stable: input order survives inside a voxel
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(head_flags64_kernel, g, b, 0, stream, keys_b, n, flags);
  if ((e = exclusive_sum(flags, pos, n, temp, temp_bytes, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(segment_starts_kernel, g, b, 0, stream, flags, pos, n, seg_start, counters);
  hipLaunchKernelGGL(greedy_voxel_kernel, dim3(static_cast<int>(min((static_cast<size_t>(n) * 64 + kThreads - 1) / kThreads, static_cast<size_t>(8192)))), b, 0, stream,
                     body, idx_b, seg_start, counters, max_pts, min_sq, flags, first_idx);  // flags now = keep
  if ((e = exclusive_sum(flags, pos, n, temp, temp_bytes, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(fill64_kernel, g, b, 0, stream, keys_a, n, kNoKey64);
  hipLaunchKernelGGL(order_keys_kernel, g, b, 0, stream, idx_b, flags, pos, first_idx, n, keys_a, counters);
  tb = temp_bytes;
  if ((e = rocprim::radix_sort_keys(temp, tb, keys_a, keys_b, static_cast<size_t>(n), 0, 64, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(gather_kept_kernel, g, b, 0, stream, body, keys_b, counters, n, kept_idx, out);
  return hipGetLastError();
}

}  // namespace mh
