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
// All of it is order-dependent sequential code in the reference; the device forms reproduce the SAME outputs in
// the same order with 11 kernels + 1 memset per scan and no library sort:
//   prepare (3 kernels)
//     input_count      per 1024-point block: how many points pass the filter / belong to the geometric subset
//     input_scatter    block prefix (sum of the preceding block counts) + in-block scan -> order-preserving
//                      compaction into points_full_ / geometric_point_idxs_; the kept timestamps go through a
//                      block-level set in LDS, then a global hash set, first-inserters append to an unsorted list
//     unique_sort      rank sort of the distinct timestamps (distinct => rank = number of smaller ones), tiles in
//                      LDS; the last rank is last_point_ns
//   deskew (1 kernel, deskew_kernels.hip)
//   preprocess (memset + 7 kernels)
//     body_voxel       gather + f32 body transform + voxel key -> hash table: slot of the voxel, atomicMin of the
//                      first input index, atomicAdd of the point count (one probe + one pair of atomics per run of
//                      consecutive lanes in the same voxel)
//     voxel_count / voxel_offsets   scan over input positions of "count of the voxel whose first point I am":
//                      voxels laid out in first-seen order (geometric.cpp:103-109), compact voxel list
//     voxel_scatter    every point into its voxel's segment (atomic cursor: unordered inside the segment)
//     greedy_voxel     one wave per voxel: sort the segment by input index (<= 64: rank sort in registers; <= 1024:
//                      wave-level binary LSD radix in LDS; above: the same radix on global scratch), then
//                      FlatContainerMinimal::add in input order
//     keep_count / keep_scatter     order-preserving compaction of the kept points = "voxels in first-seen order,
//                      points in acceptance order"
// HBM/L2-bound streaming, hashing and small-sort work (32 B records, <= 131 072 of them): no MFMA.  Compiled with
// -ffp-contract=off: the reference is a baseline x86-64 build (no FMA) and both the range filter and the voxel
// assignment are threshold tests on these f32 / f64 values.
#include <hip/hip_runtime.h>

#include <cstring>

#include "scan_device.hpp"
#include "voxel_group.hpp"
#include "voxel_map.hpp"

namespace mh
{
namespace
{
using namespace vg;  // block helpers, voxel hash grouping, in-wave index sort (voxel_group.hpp)

__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
// ---- prepareInput ------------------------------------------------------------------------------------
struct FilterParams
{
  float range_min_sq, range_max_sq, intensity_min, intensity_max, ns_max, z_offset;
  uint32_t stride, point_skip, ring_skip;
  uint32_t canonical;    // the records came out of decode_points_kernel: `reflectivity` carries its reject flag
  uint32_t ring_filter;  // the ring filter applies to this sensor (lidar/manager.cpp:321-332)
};

__device__ __forceinline__ float range_sq_of(const mh_ouster_point & p) { return p.x * p.x + p.y * p.y + p.z * p.z; }

// bit 0: the point goes into points_full_; bit 1: it also belongs to the geometric subset
__device__ __forceinline__ uint32_t filter_point(const mh_ouster_point & p, uint32_t i, const FilterParams & f)
{
  bool keep = (i % f.stride) == 0;                                                   // :246 loop stride
  keep = keep && !(isnan(p.x) || isnan(p.y) || isnan(p.z));                           // :253
  keep = keep && !(isnan(p.intensity) || p.intensity < f.intensity_min || p.intensity > f.intensity_max);  // :272-276
  const float r2 = range_sq_of(p);
  keep = keep && !(r2 < f.range_min_sq || r2 > f.range_max_sq);                       // :281-282
  keep = keep && !(static_cast<float>(p.t) > f.ns_max);                               // :306 (uint32 promoted to float)
  keep = keep && !(f.canonical && (p.reflectivity & 1u));                             // :256-262 (Livox tag), set by the decoder
  // :318-334 point-skip and ring filters select the geometric subset
  const bool geo = keep && (i % f.point_skip) == 0 && (!f.ring_filter || (p.ring % f.ring_skip) == 0);
  return (keep ? 1u : 0u) | (geo ? 2u : 0u);
}

__global__ __launch_bounds__(kThreads) void input_count_kernel(const mh_ouster_point * __restrict__ raw, uint32_t n, FilterParams f,
                                                                uint32_t * __restrict__ blk_full, uint32_t * __restrict__ blk_geo,
                                                                uint32_t * __restrict__ ns_table, uint32_t ns_cap,
                                                                ScanCounters * counters)
{
  __shared__ uint32_t lds[8];
  // the timestamp hash set of input_scatter_kernel: every block clears its share
  const uint32_t per = (ns_cap + gridDim.x - 1) / gridDim.x, c0 = blockIdx.x * per;
  for (uint32_t i = c0 + threadIdx.x; i < min(c0 + per, ns_cap); i += kThreads) ns_table[i] = kEmpty32;
  if (blockIdx.x == 0 && threadIdx.x < sizeof(ScanCounters) / 4) reinterpret_cast<uint32_t *>(counters)[threadIdx.x] = 0u;
  const uint32_t base = blockIdx.x * kBlockItems + threadIdx.x * kItems;
  uint32_t c_full = 0, c_geo = 0;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k)
    if (base + k < n) {
      const uint32_t fl = filter_point(raw[base + k], base + k, f);
      c_full += fl & 1u;
      c_geo += fl >> 1;
    }
  c_full = wave_sum(c_full);
  c_geo = wave_sum(c_geo);
  if ((threadIdx.x & 63u) == 0) {
    lds[threadIdx.x >> 6] = c_full;
    lds[4 + (threadIdx.x >> 6)] = c_geo;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_full[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
    blk_geo[blockIdx.x] = lds[4] + lds[5] + lds[6] + lds[7];
  }
}

__global__ __launch_bounds__(kThreads) void input_scatter_kernel(const mh_ouster_point * __restrict__ raw, uint32_t n, FilterParams f,
                                                                  const uint32_t * __restrict__ blk_full,
                                                                  const uint32_t * __restrict__ blk_geo, uint32_t * ns_table,
                                                                  uint32_t ns_mask, uint32_t * __restrict__ ns_unsorted,
                                                                  mh_point32 * __restrict__ points_full,
                                                                  uint32_t * __restrict__ geo_idx, ScanCounters * counters)
{
  __shared__ uint32_t lds[8];
  uint32_t off_full, off_geo;
  block_offsets2(blk_full, blk_geo, blockIdx.x, off_full, off_geo, lds);
  const uint32_t base = blockIdx.x * kBlockItems + threadIdx.x * kItems;
  mh_ouster_point p[kItems];
  uint32_t fl[kItems];
  uint32_t c_full = 0, c_geo = 0;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k) {
    fl[k] = 0;
    if (base + k < n) {
      p[k] = raw[base + k];
      fl[k] = filter_point(p[k], base + k, f);
      c_full += fl[k] & 1u;
      c_geo += fl[k] >> 1;
    }
  }
  uint32_t e_full, e_geo, t_full, t_geo;
  block_exclusive_sum2(c_full, c_geo, e_full, e_geo, t_full, t_geo, lds);
  uint32_t pos_full = off_full + e_full, pos_geo = off_geo + e_geo;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k) {
    if (!(fl[k] & 1u)) continue;
    mh_point32 o;
    o.x = p[k].x;
    o.y = p[k].y;
    o.z = p[k].z + f.z_offset;
    o.pad = 0.f;
    o.intensity = p[k].intensity;
    o.t = p[k].t;
    o.idx = base + k;
    // :312-313 std::sqrt(float): correctly rounded.  sqrt in double then one rounding to float is exact for
    // that (53 >= 2 * 24 + 2 bits) and does not depend on how the compiler lowers f32 sqrt
    o.range = static_cast<float>(sqrt(static_cast<double>(range_sq_of(p[k]))));
    points_full[pos_full] = o;
    if (fl[k] & 2u) geo_idx[pos_geo++] = pos_full;
    ++pos_full;
  }
  // Distinct timestamps (:340-368), a set: any assignment of points to threads will do.  The block takes the points
  // blockIdx + k * gridDim: in an organised cloud (row-major, 2^k columns) those share a handful of columns, so the
  // block-level set in LDS leaves a few values per block for the global hash set (device-scope atomics: the expensive
  // part) instead of one per point.  The first thread to claim a global slot lists the value.
  __shared__ uint32_t block_set[2 * kThreads];
  block_set[threadIdx.x] = kEmpty32;
  block_set[kThreads + threadIdx.x] = kEmpty32;
  __syncthreads();
  bool claimed = false;
  uint32_t t = 0;
  {
    const uint64_t i2 = blockIdx.x + static_cast<uint64_t>(threadIdx.x) * gridDim.x;
    bool mine = false;
    if (i2 < n) {
      const mh_ouster_point q = raw[i2];
      t = q.t;
      mine = (filter_point(q, static_cast<uint32_t>(i2), f) & 1u) != 0u;
    }
    if (mine && t == kEmpty32) {  // the table's empty marker itself: carried by a flag, appended last by unique_sort_kernel
      atomicOr(&counters->has_max_ns, 1u);
      mine = false;
    }
    if (mine) {
      uint32_t slot = mix32(t) & (2 * kThreads - 1);
      for (;;) {
        const uint32_t cur = atomicCAS(&block_set[slot], kEmpty32, t);
        if (cur == kEmpty32) break;  // first in the block
        if (cur == t) {
          mine = false;
          break;
        }
        slot = (slot + 1) & (2 * kThreads - 1);
      }
    }
    if (mine) {
      uint32_t slot = mix32(t) & ns_mask;
      for (;;) {
        uint32_t cur = __hip_atomic_load(&ns_table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmpty32) {
          cur = atomicCAS(&ns_table[slot], kEmpty32, t);
          if (cur == kEmpty32) {
            claimed = true;
            break;
          }
        }
        if (cur == t) break;
        slot = (slot + 1) & ns_mask;
      }
    }
  }
  {  // the claimed values of the wave are appended with one atomicAdd
    const uint64_t cm = __ballot(claimed);
    if (cm != 0ull) {
      const uint32_t lane = threadIdx.x & 63u;
      const int leader = __ffsll(static_cast<long long>(cm)) - 1;
      uint32_t at = 0;
      if (lane == static_cast<uint32_t>(leader)) at = atomicAdd(&counters->n_unique_ns, static_cast<uint32_t>(__popcll(cm)));
      at = __shfl(at, leader);
      if (claimed) ns_unsorted[at + static_cast<uint32_t>(__popcll(cm & ((1ull << lane) - 1ull)))] = t;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    counters->n_full = off_full + t_full;
    counters->n_geometric = off_geo + t_geo;
  }
}

// ---- other sensors: any point record -> the canonical 32-byte record the filter kernels read -------------------
// (lidar/manager.cpp:177-203 transpose, :256-271 tag / reflectivity, :285-304 time decoding).  Output index j is the
// index in the (transposed) cloud; canonical.reflectivity bit 0 = rejected by the Livox tag test; ring as uint16.
__device__ __forceinline__ float load_f32(const uint8_t * p)
{
  float v;
  memcpy(&v, p, 4);
  return v;
}
// `uint32_t t_ns = <double>` as the reference's x86-64 build evaluates it: truncate to a 64-bit integer (cvttsd2si),
// keep the low word.  A point stamped slightly BEFORE the header therefore becomes ~2^32 ns and fails the ns_max test
// rather than saturating to 0 and passing it (v_cvt_u32_f64 saturates).
__device__ __forceinline__ uint32_t f64_to_u32(double v) { return static_cast<uint32_t>(static_cast<long long>(v)); }
__global__ __launch_bounds__(kThreads) void decode_points_kernel(const uint8_t * __restrict__ raw, uint32_t n, mh_point_layout L, uint32_t width,
                                                                  uint32_t height, uint32_t transpose, double header_ts,
                                                                  mh_ouster_point * __restrict__ out, uint32_t * bad_ring)
{
  const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= n) return;
  uint32_t i = j;
  if (transpose) {  // transposed[new_row * new_width + new_col] = cloud[new_col * width + new_row], new_width = height
    const uint32_t new_row = j / height, new_col = j % height;
    i = new_col * width + new_row;
  }
  const uint8_t * r = raw + static_cast<size_t>(i) * L.stride;
  mh_ouster_point o;
  o.x = load_f32(r + L.off_x);
  o.y = load_f32(r + L.off_y);
  o.z = load_f32(r + L.off_z);
  o.pad = 0.f;
  if (L.intensity_is_u16) {
    uint16_t v;
    memcpy(&v, r + L.off_intensity, 2);
    o.intensity = static_cast<float>(v);  // :265-271 reflectivity is the intensity
  } else {
    o.intensity = load_f32(r + L.off_intensity);
  }
  uint32_t t_ns = 0;
  if (L.time_kind == MH_TIME_U32_NS) {
    memcpy(&t_ns, r + L.off_time, 4);
  } else if (L.time_kind == MH_TIME_F32_S) {
    t_ns = f64_to_u32(static_cast<double>(load_f32(r + L.off_time)) * 1e9);       // :299 time * 1e9
  } else {
    double ts;
    memcpy(&ts, r + L.off_time, 8);
    t_ns = f64_to_u32(L.time_kind == MH_TIME_F64_S_ABS ? (ts - header_ts) * 1e9   // :291, :301
                                                        : ts - header_ts * 1e9);    // :293
  }
  o.t = t_ns;
  uint32_t ring = 0;
  if (L.ring_kind == MH_RING_U16) {
    uint16_t v;
    memcpy(&v, r + L.off_ring, 2);
    ring = v;
  } else if (L.ring_kind == MH_RING_U8) {
    ring = r[L.off_ring];
  } else if (L.ring_kind == MH_RING_F32) {
    ring = static_cast<uint32_t>(load_f32(r + L.off_ring));
  }
  if (ring > 0xFFFFu) ring = 0xFFFFu;
  o.ring = static_cast<uint16_t>(ring);
  if (ring >= 128u) *bad_ring = 1u;  // only organize_by_ring cares (its tables hold 128 rings, :215-217)
  uint32_t flags = 0;
  if (L.has_tag) {
    const uint32_t tag = r[L.off_tag];
    if (!((tag & 0x30u) == 0x10u || (tag & 0x30u) == 0x00u)) flags |= 1u;  // :256-262
  }
  o.reflectivity = static_cast<uint16_t>(flags);
  o.pad2 = 0;
  out[j] = o;
}

// organize_pointcloud_by_ring (:205-241): a stable counting sort by ring, 128 rings.  Per-block ring histograms, then
// every block places its records: offset of (ring, block) = points of lower rings + points of this ring in earlier blocks.
constexpr uint32_t kRings = 128;
__global__ __launch_bounds__(kThreads) void ring_histogram_kernel(const mh_ouster_point * __restrict__ in, uint32_t n, uint32_t * __restrict__ hist)
{
  __shared__ uint32_t h[kRings];
  if (threadIdx.x < kRings) h[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
  if (j < n) atomicAdd(&h[in[j].ring & (kRings - 1u)], 1u);
  __syncthreads();
  if (threadIdx.x < kRings) hist[static_cast<size_t>(blockIdx.x) * kRings + threadIdx.x] = h[threadIdx.x];
}
// per-(block, ring) counts -> exclusive offsets in place over the blocks of a ring, and the ring's total in
// hist[n_blk * kRings + r].  One wave per ring: every lane scans a contiguous run of blocks, a wave scan joins the runs.
__global__ __launch_bounds__(64) void ring_offsets_kernel(uint32_t * hist, uint32_t n_blk)
{
  const uint32_t r = blockIdx.x, lane = threadIdx.x;
  const uint32_t per = (n_blk + 63u) / 64u, b0 = lane * per, b1 = min(b0 + per, n_blk);
  uint32_t sum = 0;
  for (uint32_t b = b0; b < b1; ++b) sum += hist[static_cast<size_t>(b) * kRings + r];
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), d, 64));
    if (static_cast<int>(lane) >= d) incl += up;
  }
  uint32_t run = incl - sum;  // blocks of the lanes before this one
  for (uint32_t b = b0; b < b1; ++b) {
    const uint32_t c = hist[static_cast<size_t>(b) * kRings + r];
    hist[static_cast<size_t>(b) * kRings + r] = run;
    run += c;
  }
  if (lane == 63u) hist[static_cast<size_t>(n_blk) * kRings + r] = incl;
}
__global__ __launch_bounds__(kThreads) void ring_place_kernel(const mh_ouster_point * __restrict__ in, uint32_t n, const uint32_t * __restrict__ offs,
                                                               mh_ouster_point * __restrict__ out)
{
  __shared__ uint32_t s_ring[kThreads], s_tot[kRings], s_base[kRings];
  if (threadIdx.x < kRings) s_tot[threadIdx.x] = offs[static_cast<size_t>(gridDim.x) * kRings + threadIdx.x];
  const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
  const uint32_t ring = j < n ? (in[j].ring & (kRings - 1u)) : 0xFFFFFFFFu;
  s_ring[threadIdx.x] = ring;
  __syncthreads();
  if (threadIdx.x < kRings) {  // where ring r starts: the totals of the rings before it (:221-227)
    uint32_t base = 0;
    for (uint32_t q = 0; q < threadIdx.x; ++q) base += s_tot[q];
    s_base[threadIdx.x] = base;
  }
  __syncthreads();
  if (j >= n) return;
  uint32_t pos = s_base[ring] + offs[static_cast<size_t>(blockIdx.x) * kRings + ring];
  for (uint32_t t = 0; t < threadIdx.x; ++t) pos += s_ring[t] == ring ? 1u : 0u;  // earlier records of this block, same ring
  out[pos] = in[j];
}

// the distinct timestamps, ascending: rank = number of smaller values (they are distinct).  The whole list passes
// through LDS in 1024-value tiles; O(m^2) compares, m is the column count of the sensor (1024 / 2048) in practice.
// :310 last_point_ns = max t over the kept points = the value of the last rank.
// A block ranks 64 values; its 4 waves each scan a quarter of every tile.
constexpr uint32_t kTile = kThreads * 4;
__global__ __launch_bounds__(kThreads) void unique_sort_kernel(const uint32_t * __restrict__ ns_unsorted, uint32_t * __restrict__ unique_ns,
                                                                ScanCounters * counters)
{
  __shared__ uint4 tile[kThreads];
  __shared__ uint32_t part[4][64];
  const uint32_t m = counters->n_unique_ns;
  const bool has_max = counters->has_max_ns != 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0 && has_max) {
    unique_ns[m] = kEmpty32;
    counters->last_point_ns = kEmpty32;
  }
  if (blockIdx.x * 64u >= m) return;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t i = blockIdx.x * 64u + lane;
  const uint32_t v = i < m ? ns_unsorted[i] : 0u;
  uint32_t rank = 0;
  for (uint32_t t0 = 0; t0 < m; t0 += kTile) {
    uint4 q;
    const uint32_t j = t0 + threadIdx.x * 4;
    q.x = j + 0 < m ? ns_unsorted[j + 0] : kEmpty32;  // the padding value is never smaller than anything
    q.y = j + 1 < m ? ns_unsorted[j + 1] : kEmpty32;
    q.z = j + 2 < m ? ns_unsorted[j + 2] : kEmpty32;
    q.w = j + 3 < m ? ns_unsorted[j + 3] : kEmpty32;
    __syncthreads();
    tile[threadIdx.x] = q;
    __syncthreads();
    const uint32_t lim = min(static_cast<uint32_t>(kThreads), (m - t0 + 3) / 4);
#pragma unroll 4
    for (uint32_t u = wave * 64u; u < min(lim, wave * 64u + 64u); ++u) {
      const uint4 w = tile[u];  // broadcast read
      rank += (w.x < v) + (w.y < v) + (w.z < v) + (w.w < v);
    }
  }
  part[wave][lane] = rank;
  __syncthreads();
  if (wave == 0 && i < m) {
    rank = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    unique_ns[rank] = v;
    if (rank == m - 1 && !has_max) counters->last_point_ns = v;
  }
}

// ---- Geometric::preprocess + the voxel assignment of Geometric::downsample ------------------------------------
constexpr int kCoordBits = 21;
constexpr int kCoordBias = 1 << (kCoordBits - 1);

__global__ __launch_bounds__(kThreads) void body_voxel_kernel(const mh_point32 * __restrict__ pts, const uint32_t * __restrict__ geo_idx,
                                                               uint32_t n, Rt12 P, double inv_leaf, mh_point32 * __restrict__ body,
                                                               VoxelHash h, uint32_t * __restrict__ slot_of)
{
  const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
  const bool valid = j < n;
  uint64_t key = kEmpty64;
  if (valid) {
    mh_point32 p = pts[geo_idx[j]];
    const float px = p.x, py = p.y, pz = p.z;  // Eigen's coefficient order r0*x + (r1*y + r2*z), then + t (geometric.cpp:154-161)
    p.x = (P.v[0] * px + (P.v[1] * py + P.v[2] * pz)) + P.v[9];
    p.y = (P.v[3] * px + (P.v[4] * py + P.v[5] * pz)) + P.v[10];
    p.z = (P.v[6] * px + (P.v[7] * py + P.v[8] * pz)) + P.v[11];
    body[j] = p;
    // :77-80 coord = fast_floor(double(p) * inv_leaf)
    const int cx = fast_floor(static_cast<double>(p.x) * inv_leaf), cy = fast_floor(static_cast<double>(p.y) * inv_leaf),
              cz = fast_floor(static_cast<double>(p.z) * inv_leaf);
    const int bx = cx + kCoordBias, by = cy + kCoordBias, bz = cz + kCoordBias;
    if (((bx | by | bz) >> kCoordBits) != 0) *h.bad = 0u;  // (memset to all-ones: any other value = "a coordinate did not fit")
    key = (static_cast<uint64_t>(bx & ((1 << kCoordBits) - 1)) << (2 * kCoordBits)) |
          (static_cast<uint64_t>(by & ((1 << kCoordBits) - 1)) << kCoordBits) | static_cast<uint64_t>(bz & ((1 << kCoordBits) - 1));
  }
  const uint32_t slot = voxel_assign(h, valid, key, j);
  if (valid) slot_of[j] = slot;
}

// One WAVE per voxel.  (1) the voxel's point indices, unordered after the scatter, ascending: input order.  <= 64: rank
// sort in registers; <= kLdsSort: radix in LDS; above: the same radix on global scratch.  (2) FlatContainerMinimal::add
// over them: lane j holds the j-th point kept so far (<= 20); every incoming point is tested against all of them at
// once (one fp64 distance per lane, one ballot).  idx_sorted / keep are indexed by position in the first-seen layout.
__global__ __launch_bounds__(kThreads) void greedy_voxel_kernel(const mh_point32 * __restrict__ pts, const uint32_t * __restrict__ idx_unsorted,
                                                                 uint32_t * idx_sorted, uint32_t * idx_tmp, const uint2 * __restrict__ vox_seg,
                                                                 const ScanCounters * counters, uint32_t max_pts, double min_sq,
                                                                 uint32_t * __restrict__ keep)
{
  __shared__ uint32_t sort_lds[kThreads / 64][2][kLdsSort];
  const uint32_t nv = counters->n_voxels;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_in_block = threadIdx.x >> 6;
  const uint32_t wave = (blockIdx.x * kThreads + threadIdx.x) >> 6, n_waves = (gridDim.x * kThreads) >> 6;
  const uint32_t cap = min(max_pts, static_cast<uint32_t>(kBucketStride));  // utils.hpp:262 size cap
  for (uint32_t v = wave; v < nv; v += n_waves) {
    const uint2 seg = vox_seg[v];
    const uint32_t s0 = __builtin_amdgcn_readfirstlane(seg.x), len = __builtin_amdgcn_readfirstlane(seg.y);
    const uint32_t s1 = s0 + len;
    const uint32_t first_chunk = sort_segment_indices(idx_unsorted, idx_sorted, idx_tmp, sort_lds[wave_in_block][0], sort_lds[wave_in_block][1], s0, len);
    // FlatContainerMinimal::add, 64 candidates at a time: (a) every candidate against the points kept so far (<= 20
    // broadcasts), (b) the survivors in input order: the first one is kept and knocks out the later ones near it.  A
    // candidate is kept iff no EARLIER KEPT point is closer than min_dist and the voxel is not full — the sequential
    // rule, with the sequential part reduced to the points that are actually kept.
    float kx = 0.f, ky = 0.f, kz = 0.f;  // this lane's kept point (lane < n_kept)
    uint32_t n_kept = 0;
    for (uint32_t base = s0; base < s1; base += 64u) {
      const uint32_t s = base + lane;
      const bool valid = s < s1;
      float px = 0.f, py = 0.f, pz = 0.f;
      if (valid) {
        const uint32_t j = sorted_index_at(first_chunk, sort_lds[wave_in_block][0], idx_sorted, s0, len, s);
        if (len <= kLdsSort) idx_sorted[s] = j;
        const mh_point32 p = pts[j];
        px = p.x;
        py = p.y;
        pz = p.z;
      }
      const double qx = static_cast<double>(px), qy = static_cast<double>(py), qz = static_cast<double>(pz);
      bool blocked = !valid;
      for (uint32_t i = 0; i < n_kept; ++i) {
        const double dx = static_cast<double>(lane_value(kx, i)) - qx, dy = static_cast<double>(lane_value(ky, i)) - qy,
                     dz = static_cast<double>(lane_value(kz, i)) - qz;
        // Vector3d squaredNorm: p0 + (p1 + p2); utils.hpp:266-272
        blocked = blocked || dx * dx + (dy * dy + dz * dz) < min_sq;
      }
      uint64_t open = __ballot(!blocked), kept_mask = 0;
      while (open != 0ull && n_kept < cap) {  // utils.hpp:262 size cap
        const uint32_t u = static_cast<uint32_t>(__ffsll(static_cast<long long>(open))) - 1u;
        const float ux = lane_value(px, u), uy = lane_value(py, u), uz = lane_value(pz, u);
        if (lane == n_kept) {
          kx = ux;
          ky = uy;
          kz = uz;
        }
        ++n_kept;
        kept_mask |= 1ull << u;
        const double dx = static_cast<double>(ux) - qx, dy = static_cast<double>(uy) - qy, dz = static_cast<double>(uz) - qz;
        blocked = blocked || dx * dx + (dy * dy + dz * dz) < min_sq;
        open = __ballot(!blocked) & (u == 63u ? 0ull : (~0ull << (u + 1u)));
      }
      if (valid) keep[s] = static_cast<uint32_t>((kept_mask >> lane) & 1ull);
    }
  }
}

__global__ __launch_bounds__(kThreads) void keep_count_kernel(const uint32_t * __restrict__ keep, uint32_t n, uint32_t * __restrict__ blk_keep)
{
  __shared__ uint32_t lds[4];
  const uint32_t base = blockIdx.x * kBlockItems + threadIdx.x * kItems;
  uint32_t c = 0;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k)
    if (base + k < n) c += keep[base + k];
  c = wave_sum(c);
  if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_keep[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// kept points in position order = "voxels in first-seen order, points in acceptance order" (geometric.cpp:103-109)
__global__ __launch_bounds__(kThreads) void keep_scatter_kernel(const mh_point32 * __restrict__ pts, const uint32_t * __restrict__ idx_sorted,
                                                                 const uint32_t * __restrict__ keep, uint32_t n,
                                                                 const uint32_t * __restrict__ blk_keep, const uint32_t * bad,
                                                                 uint32_t * __restrict__ kept_idx, mh_point32 * __restrict__ out,
                                                                 ScanCounters * counters)
{
  __shared__ uint32_t lds[8];
  uint32_t off, unused;
  block_offsets2(blk_keep, blk_keep, blockIdx.x, off, unused, lds);
  const uint32_t base = blockIdx.x * kBlockItems + threadIdx.x * kItems;
  uint32_t kk[kItems], c = 0;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k) {
    kk[k] = base + k < n ? keep[base + k] : 0u;
    c += kk[k];
  }
  uint32_t e, e2, t, t2;
  block_exclusive_sum2(c, 0u, e, e2, t, t2, lds);
  uint32_t pos = off + e;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k)
    if (kk[k]) {
      const uint32_t j = idx_sorted[base + k];
      kept_idx[pos] = j;
      out[pos] = pts[j];
      ++pos;
    }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    counters->n_downsampled = off + t;
    counters->bad_coord = *bad != kEmpty32 ? 1u : 0u;
  }
}
}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------
PrepareLayout prepare_layout(size_t n)
{
  PrepareLayout L;
  L.n_blocks = blocks_for(static_cast<uint32_t>(n));
  L.ns_cap = pow2_at_least(2 * static_cast<uint64_t>(n));
  L.words = 2 * static_cast<size_t>(L.n_blocks) + L.ns_cap + (n ? n : 1);
  return L;
}

VoxelLayout voxel_layout(size_t n)
{
  const vg::Layout G = vg::layout(n);
  VoxelLayout L;
  L.n_blocks = G.n_blocks;
  L.cap = G.cap;
  L.clear_bytes = G.clear_bytes;
  L.bytes = G.bytes + ((n ? n : 1) + static_cast<size_t>(G.n_blocks)) * 4;  // + keep flags, per-block keep counts
  return L;
}

hipError_t launch_decode_points(const void * raw, uint32_t n, const mh_point_layout & layout, uint32_t width, uint32_t height, bool transpose,
                                bool organize_by_ring, double header_ts, mh_ouster_point * canon, mh_ouster_point * tmp, uint32_t * hist,
                                uint32_t * bad_ring, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  const dim3 g((n + kThreads - 1) / kThreads), b(kThreads);
  hipLaunchKernelGGL(decode_points_kernel, g, b, 0, stream, static_cast<const uint8_t *>(raw), n, layout, width, height, transpose ? 1u : 0u,
                     header_ts, organize_by_ring ? tmp : canon, bad_ring);
  if (organize_by_ring) {
    hipLaunchKernelGGL(ring_histogram_kernel, g, b, 0, stream, tmp, n, hist);
    hipLaunchKernelGGL(ring_offsets_kernel, dim3(kRings), dim3(64), 0, stream, hist, g.x);
    hipLaunchKernelGGL(ring_place_kernel, g, b, 0, stream, tmp, n, hist, canon);
  }
  return hipGetLastError();
}

hipError_t launch_prepare_input(const mh_ouster_point * raw, uint32_t n, const mh_input_config & cfg, uint32_t * scratch,
                                mh_point32 * points_full, uint32_t * geo_idx, uint32_t * unique_ns, ScanCounters * counters,
                                hipStream_t stream, bool canonical, bool ring_filter)
{
  FilterParams f;
  f.canonical = canonical ? 1u : 0u;
  f.ring_filter = ring_filter ? 1u : 0u;
  f.range_min_sq = cfg.range_min * cfg.range_min;  // manager.cpp:19-20 (float products)
  f.range_max_sq = cfg.range_max * cfg.range_max;
  f.intensity_min = cfg.intensity_min;
  f.intensity_max = cfg.intensity_max;
  f.ns_max = cfg.ns_max;
  f.z_offset = cfg.z_offset;
  f.point_skip = static_cast<uint32_t>(cfg.point_skip_divisor > 0 ? cfg.point_skip_divisor : 1);
  f.ring_skip = static_cast<uint32_t>(cfg.ring_skip_divisor > 0 ? cfg.ring_skip_divisor : 1);
  f.stride = cfg.create_full_res_pointcloud ? 1u : f.point_skip;
  const PrepareLayout L = prepare_layout(n);
  uint32_t * blk_full = scratch, * blk_geo = blk_full + L.n_blocks, * ns_table = blk_geo + L.n_blocks, * ns_unsorted = ns_table + L.ns_cap;
  const dim3 g(L.n_blocks), b(kThreads);
  hipLaunchKernelGGL(input_count_kernel, g, b, 0, stream, raw, n, f, blk_full, blk_geo, ns_table, L.ns_cap, counters);
  hipLaunchKernelGGL(input_scatter_kernel, g, b, 0, stream, raw, n, f, blk_full, blk_geo, ns_table, L.ns_cap - 1u, ns_unsorted,
                     points_full, geo_idx, counters);
  hipLaunchKernelGGL(unique_sort_kernel, dim3((n + 63u) / 64u ? (n + 63u) / 64u : 1u), b, 0, stream,
                     ns_unsorted, unique_ns, counters);
  return hipGetLastError();
}

hipError_t launch_preprocess(const mh_point32 * points_full, const uint32_t * geo_idx, uint32_t n, const Rt12 & body_from_lidar,
                             double leaf, uint32_t max_pts, double min_dist, void * scratch, mh_point32 * body,
                             uint32_t * kept_idx, mh_point32 * out, ScanCounters * counters, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  const VoxelLayout L = voxel_layout(n);
  hipError_t e = hipMemsetAsync(scratch, 0xFF, L.clear_bytes, stream);
  if (e != hipSuccess) return e;
  const vg::Buffers B = vg::carve(scratch, n);
  const VoxelHash & h = B.h;
  uint32_t * keep = reinterpret_cast<uint32_t *>(static_cast<char *>(scratch) + vg::layout(n).bytes), * blk_keep = keep + n;
  const double inv_leaf = 1.0 / leaf;            // geometric.cpp:61
  const double min_sq = min_dist * min_dist;     // :63
  const dim3 gp((n + kThreads - 1) / kThreads), gb(L.n_blocks), b(kThreads);
  hipLaunchKernelGGL(body_voxel_kernel, gp, b, 0, stream, points_full, geo_idx, n, body_from_lidar, inv_leaf, body, h, B.slot_of);
  if ((e = vg::launch_group(B, n, &counters->n_voxels, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(greedy_voxel_kernel, dim3(static_cast<uint32_t>(min((static_cast<size_t>(n) * 64 + kThreads - 1) / kThreads, static_cast<size_t>(8192)))),
                     b, 0, stream, body, B.idx_unsorted, B.idx_sorted, B.idx_tmp, B.seg, counters, max_pts, min_sq, keep);
  hipLaunchKernelGGL(keep_count_kernel, gb, b, 0, stream, keep, n, blk_keep);
  hipLaunchKernelGGL(keep_scatter_kernel, gb, b, 0, stream, body, B.idx_sorted, keep, n, blk_keep, h.bad, kept_idx, out, counters);
  return hipGetLastError();
}

}  // namespace mh
