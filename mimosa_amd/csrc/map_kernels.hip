// HIP kernels of the device-resident incremental voxel map: gtsam_points::iVox::insert,
// the LRU purge, voxel_data() and the block / hash bookkeeping, all on the device.
//
// Reference: IncrementalVoxelMapPCL::insert src/lidar/incremental_voxel_map.cpp:19-24 -> gtsam_points::iVox::insert
// (greedy first-come-first-kept per voxel, min-distance rule, 20-point cap, voxels numbered in
// creation order, LRU purge every lru_clear_cycle inserts); Geometric::updateMap's f32 world transform
// src/lidar/geometric.cpp:483-490; mimosa's own restatement of the add rule include/mimosa/lidar/utils.hpp:260-278.
//
// The insertion rule is sequential in the reference, but it only couples points of ONE voxel, in input order.
//   1. voxel key per point -> the sort-free grouping of voxel_group.hpp (hash assign, first-seen segment layout, atomic
//      scatter; the consumer wave sorts its segment's indices): every touched voxel is a contiguous segment, segments
//      in first-seen order, points of a segment in input order
//   2. a segment whose voxel is not in the map yet creates it; creation order = first-seen order (an exclusive scan of
//      the "new" flags over the segments gives the rank), so voxel ids — and with them getCloud's order — are
//      exactly the reference's
//   3. new voxels claim their home block and the <= 7 adjacent blocks whose halo they touch in the hash table
//      (64-bit compare-and-swap on the packed block coordinate; block ids are not observable)
//   4. one WAVE per touched voxel runs FlatContainer::add over its segment: the kept points sit one per lane, every
//      incoming point is tested against all of them with one fp64 distance per lane and one ballot
// Compiled with -ffp-contract=off: voxel assignment and the min-distance test are threshold tests on products the
// reference evaluates without FMA (baseline x86-64 build).
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "map_device.hpp"
#include "voxel_group.hpp"

namespace mh
{
namespace
{
constexpr int kT = 256;
int grid_for(uint32_t n) { return static_cast<int>(max(1u, min((n + kT - 1) / kT, 8192u))); }

__device__ __forceinline__ void unpack_coord_key(uint64_t key, int & x, int & y, int & z)
{
  const uint64_t m = (1ull << kVoxCoordBits) - 1;
  x = static_cast<int>((key >> (2 * kVoxCoordBits)) & m) - kVoxCoordBias;
  y = static_cast<int>((key >> kVoxCoordBits) & m) - kVoxCoordBias;
  z = static_cast<int>(key & m) - kVoxCoordBias;
}

__device__ __forceinline__ int find_block(const MapArrays & m, int bx, int by, int bz)
{
  const uint64_t key = pack_coord_key(bx, by, bz);
  uint32_t h = block_hash(bx, by, bz) & m.table_mask;
  for (;;) {
    const int4 e = m.table[h];
    const uint64_t k = static_cast<uint64_t>(static_cast<uint32_t>(e.x)) | (static_cast<uint64_t>(static_cast<uint32_t>(e.y)) << 32);
    if (k == kEmptyKey) return -1;
    if (k == key) return e.z;
    h = (h + 1) & m.table_mask;
  }
}

// Claims a table slot for the block (no-op when it is there already).  Only the claiming thread assigns the id; readers
// of ids run in a later kernel.
__device__ __forceinline__ void claim_block(const MapArrays & m, int bx, int by, int bz)
{
  const uint64_t key = pack_coord_key(bx, by, bz);
  uint32_t h = block_hash(bx, by, bz) & m.table_mask;
  for (;;) {
    unsigned long long * slot = reinterpret_cast<unsigned long long *>(&m.table[h]);
    const unsigned long long old = atomicCAS(slot, static_cast<unsigned long long>(kEmptyKey), static_cast<unsigned long long>(key));
    if (old == kEmptyKey) {
      m.table[h].z = static_cast<int>(atomicAdd(&m.state->n_blocks, 1u));
      return;
    }
    if (old == key) return;
    h = (h + 1) & m.table_mask;
  }
}

// The <= 8 blocks whose halo'd table holds voxel (cx,cy,cz): its home block and, per axis, the neighbour it touches.
// which in [0, 8): bit a selects the second block of axis a.  Returns false when that combination does not exist.
__device__ __forceinline__ bool voxel_block(int cx, int cy, int cz, int which, int & bx, int & by, int & bz)
{
  const int c[3] = {cx, cy, cz};
  int b[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int l = c[a] & (kBlockDim - 1);
    b[a] = c[a] >> kBlockLog2;
    if ((which >> a) & 1) {
      if (l == 0)
        b[a] -= 1;
      else if (l == kBlockDim - 1)
        b[a] += 1;
      else
        return false;
    }
  }
  bx = b[0];
  by = b[1];
  bz = b[2];
  return true;
}

// ---- phase A ---------------------------------------------------------------------------------------------
// phase A.1: the batch as float4 (after the optional transform), its voxel keys into the batch hash
__global__ __launch_bounds__(kT) void map_assign_kernel(const float * src, uint32_t n, uint32_t stride, const float * Rt12, double inv_leaf,
                                                         float4 * pts, VoxelHash h, uint32_t * slot_of, MapState * st)
{
  const uint32_t i = blockIdx.x * kT + threadIdx.x;
  const bool valid = i < n;
  uint64_t key = vg::kEmpty64;
  if (valid) {
    float x = src[static_cast<size_t>(i) * stride], y = src[static_cast<size_t>(i) * stride + 1], z = src[static_cast<size_t>(i) * stride + 2];
    if (Rt12) {  // geometric.cpp:483-490: f32 R p + t, Eigen's r0 x + (r1 y + r2 z) order, no FMA
      const float px = x, py = y, pz = z;
      x = (Rt12[0] * px + (Rt12[1] * py + Rt12[2] * pz)) + Rt12[9];
      y = (Rt12[3] * px + (Rt12[4] * py + Rt12[5] * pz)) + Rt12[10];
      z = (Rt12[6] * px + (Rt12[7] * py + Rt12[8] * pz)) + Rt12[11];
    }
    pts[i] = make_float4(x, y, z, 1.0f);
    const int cx = fast_floor(static_cast<double>(x) * inv_leaf), cy = fast_floor(static_cast<double>(y) * inv_leaf),
              cz = fast_floor(static_cast<double>(z) * inv_leaf);
    // one voxel of margin: the blocks a voxel touches must fit the key as well
    const int lim = kVoxCoordBias - 8;
    if (cx < -lim || cx >= lim || cy < -lim || cy >= lim || cz < -lim || cz >= lim || !(x == x) || !(y == y) || !(z == z)) {
      atomicOr(&st->bad_coord, 1u);
      key = pack_coord_key(0, 0, 0);  // keeps the grouping well-formed; the host rejects the batch before anything is modified
    } else {
      key = pack_coord_key(cx, cy, cz);
    }
  }
  const uint32_t slot = vg::voxel_assign(h, valid, key, i);
  if (valid) slot_of[i] = slot;
}

// voxel id of segment g's voxel, or ~0u when the map does not hold it yet
__device__ __forceinline__ uint32_t segment_voxel(const MapArrays & m, uint64_t key)
{
  int cx, cy, cz;
  unpack_coord_key(key, cx, cy, cz);
  const int blk = find_block(m, cx >> kBlockLog2, cy >> kBlockLog2, cz >> kBlockLog2);
  if (blk < 0) return 0xFFFFFFFFu;
  const uint32_t w = m.cells[static_cast<size_t>(blk) * kCellsPerBlock + halo_index(cx & (kBlockDim - 1), cy & (kBlockDim - 1), cz & (kBlockDim - 1))];
  return w != kEmptyCell ? (w >> 5) : 0xFFFFFFFFu;
}

// phase A.3: look every segment's voxel up (segments are in first-seen order); per-block count of the new ones
__global__ __launch_bounds__(kT) void map_lookup_kernel(const MapArrays m, VoxelHash h, const uint32_t * vox_slot, uint32_t * seg_vid,
                                                         uint32_t * blk_new)
{
  const uint32_t ns = m.state->n_segments;
  const uint32_t g = blockIdx.x * kT + threadIdx.x;
  uint32_t vid = 0;
  if (g < ns) {
    vid = segment_voxel(m, h.keys[vox_slot[g]]);
    seg_vid[g] = vid;
  }
  const int n_new = __syncthreads_count(g < ns && vid == 0xFFFFFFFFu);
  if (threadIdx.x == 0) blk_new[blockIdx.x] = static_cast<uint32_t>(n_new);
}

// phase A.4: creation ranks = exclusive scan of the "new" flags over the segments (first-seen order = creation order)
__global__ __launch_bounds__(kT) void map_new_ids_kernel(const MapArrays m, const uint32_t * blk_new, uint32_t * seg_vid)
{
  __shared__ uint32_t lds[8];
  const uint32_t ns = m.state->n_segments, nv = m.state->n_voxels;
  const uint32_t n_blocks = (ns + kT - 1) / kT;
  if (blockIdx.x >= n_blocks && !(blockIdx.x == 0)) return;
  uint32_t off, unused;
  vg::block_offsets2(blk_new, blk_new, blockIdx.x, off, unused, lds);
  const uint32_t g = blockIdx.x * kT + threadIdx.x;
  const uint32_t is_new = (g < ns && seg_vid[g] == 0xFFFFFFFFu) ? 1u : 0u;
  uint32_t e, e2, t, t2;
  vg::block_exclusive_sum2(is_new, 0u, e, e2, t, t2, lds);
  if (is_new) seg_vid[g] = nv + off + e;
  if (blockIdx.x == (n_blocks ? n_blocks - 1 : 0) && threadIdx.x == 0) m.state->n_new_voxels = ns ? off + t : 0u;
}

// ---- phase B ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void map_create_voxels_kernel(const MapArrays m, VoxelHash h, const uint32_t * vox_slot, const uint32_t * seg_vid,
                                                                uint32_t n_voxels_before, unsigned long long lru_counter)
{
  const uint32_t ns = m.state->n_segments;
  for (uint32_t s = blockIdx.x * kT + threadIdx.x; s < ns; s += gridDim.x * kT) {
    const uint32_t vid = seg_vid[s];
    if (vid < n_voxels_before) continue;
    int cx, cy, cz;
    unpack_coord_key(h.keys[vox_slot[s]], cx, cy, cz);
    m.vox[vid] = make_int4(cx, cy, cz, 0);
    m.lru[vid] = lru_counter;
    for (int w = 0; w < 8; ++w) {
      int bx, by, bz;
      if (voxel_block(cx, cy, cz, w, bx, by, bz)) claim_block(m, bx, by, bz);
    }
  }
}

// ---- phase C ---------------------------------------------------------------------------------------------
// One wave per touched voxel: FlatContainer::add over its segment (input order).  Lane j holds the j-th point of the
// bucket.  Distance in Eigen's SSE2 Vector4d squaredNorm order (dx2 + dz2) + (dy2 + dw2), dw = 0, like the k-NN.
// 64 candidates at a time: (a) every candidate against the points already in the bucket (<= 20 broadcasts), (b) the
// survivors in input order — the first one is added and knocks out the later ones within min_dist.  A candidate is
// added iff no EARLIER ADDED point (or resident point) is closer than min_dist and the bucket is not full: the
// sequential rule, with the sequential part run once per added point instead of once per point.
__device__ __forceinline__ float lane_value(float v, uint32_t lane)  // lane: wave-uniform
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), static_cast<int>(lane)));
}
__global__ __launch_bounds__(kT) void map_insert_points_kernel(const MapArrays m, const float4 * pts, const uint32_t * __restrict__ idx_unsorted,
                                                                uint32_t * idx_sorted, uint32_t * idx_tmp, const uint2 * __restrict__ seg,
                                                                const uint32_t * seg_vid, uint32_t * seg_added, uint32_t max_pts, double min_sq,
                                                                double inv_leaf, unsigned long long lru_counter)
{
  __shared__ uint32_t sort_lds[kT / 64][2][vg::kLdsSort];
  const uint32_t ns = m.state->n_segments;
  const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
  const uint32_t wave = (blockIdx.x * kT + threadIdx.x) >> 6, n_waves = (gridDim.x * kT) >> 6;
  for (uint32_t s = wave; s < ns; s += n_waves) {
    const uint2 sg = seg[s];
    const uint32_t vid = seg_vid[s], s0 = __builtin_amdgcn_readfirstlane(sg.x), len = __builtin_amdgcn_readfirstlane(sg.y), s1 = s0 + len;
    // the segment's point indices ascending = input order (the scatter left them unordered)
    const uint32_t first_chunk = vg::sort_segment_indices(idx_unsorted, idx_sorted, idx_tmp, sort_lds[wave_in_block][0], sort_lds[wave_in_block][1], s0, len);
    const int4 vx = m.vox[vid];  // created in phase B when new (count 0)
    const uint32_t old_count = static_cast<uint32_t>(vx.w);
    uint32_t count = old_count;
    float fx = 0.f, fy = 0.f, fz = 0.f;  // this lane's bucket point
    if (lane < old_count) {
      const float4 b = m.buckets[static_cast<size_t>(vid) * kBucketStride + lane];
      fx = b.x;
      fy = b.y;
      fz = b.z;
    }
    for (uint32_t base = s0; base < s1 && count < max_pts; base += 64u) {
      const uint32_t sp = base + lane;
      const bool valid = sp < s1;
      float px = 0.f, py = 0.f, pz = 0.f;
      if (valid) {
        const float4 p = pts[vg::sorted_index_at(first_chunk, sort_lds[wave_in_block][0], idx_sorted, s0, len, sp)];
        px = p.x;
        py = p.y;
        pz = p.z;
      }
      const double qx = static_cast<double>(px), qy = static_cast<double>(py), qz = static_cast<double>(pz);
      bool blocked = !valid;
      for (uint32_t i = 0; i < count; ++i) {
        const double dx = static_cast<double>(lane_value(fx, i)) - qx, dy = static_cast<double>(lane_value(fy, i)) - qy,
                     dz = static_cast<double>(lane_value(fz, i)) - qz;
        blocked = blocked || ((dx * dx + dz * dz) + (dy * dy + 0.0)) < min_sq;
      }
      uint64_t open = __ballot(!blocked);
      while (open != 0ull && count < max_pts) {
        const uint32_t u = static_cast<uint32_t>(__ffsll(static_cast<long long>(open))) - 1u;
        const float ux = lane_value(px, u), uy = lane_value(py, u), uz = lane_value(pz, u);
        if (lane == count) {
          fx = ux;
          fy = uy;
          fz = uz;
        }
        ++count;
        const double dx = static_cast<double>(ux) - qx, dy = static_cast<double>(uy) - qy, dz = static_cast<double>(uz) - qz;
        blocked = blocked || ((dx * dx + dz * dz) + (dy * dy + 0.0)) < min_sq;
        open = __ballot(!blocked) & (u == 63u ? 0ull : (~0ull << (u + 1u)));
      }
    }
    if (lane >= old_count && lane < count) {
      m.buckets[static_cast<size_t>(vid) * kBucketStride + lane] = make_float4(fx, fy, fz, 1.0f);
      // coarse copy: floor(frac(p * inv_leaf) * 1024) per axis, clamped (voxel_map.hpp)
      const double vxd = static_cast<double>(fx) * inv_leaf, vyd = static_cast<double>(fy) * inv_leaf, vzd = static_cast<double>(fz) * inv_leaf;
      const int q = 1 << kQuantBits;
      int ux = static_cast<int>((vxd - static_cast<double>(vx.x)) * static_cast<double>(q));
      int uy = static_cast<int>((vyd - static_cast<double>(vx.y)) * static_cast<double>(q));
      int uz = static_cast<int>((vzd - static_cast<double>(vx.z)) * static_cast<double>(q));
      ux = ux < 0 ? 0 : (ux > q - 1 ? q - 1 : ux);
      uy = uy < 0 ? 0 : (uy > q - 1 ? q - 1 : uy);
      uz = uz < 0 ? 0 : (uz > q - 1 ? q - 1 : uz);
      m.qbuckets[static_cast<size_t>(vid) * kBucketStride + lane] =
        static_cast<uint32_t>(ux) | (static_cast<uint32_t>(uy) << kQuantBits) | (static_cast<uint32_t>(uz) << (2 * kQuantBits));
    }
    if (lane == 0) {
      m.vox[vid] = make_int4(vx.x, vx.y, vx.z, static_cast<int>(count));
      m.lru[vid] = lru_counter;  // info.lru = lru_counter for every voxel a point of the batch fell into
      seg_added[s] = count - old_count;  // summed by map_totals_kernel: tens of thousands of atomics on one counter cost more than the insert
    }
    if (count != old_count && lane < 8) {  // the voxel's word in its home table and every halo it sits in
      int bx, by, bz;
      if (voxel_block(vx.x, vx.y, vx.z, static_cast<int>(lane), bx, by, bz)) {
        const int blk = find_block(m, bx, by, bz);
        if (blk >= 0)
          m.cells[static_cast<size_t>(blk) * kCellsPerBlock + halo_index(vx.x - bx * kBlockDim, vx.y - by * kBlockDim, vx.z - bz * kBlockDim)] =
            (vid << 5) | count;
      }
    }
  }
}

// totals after phase C: n_points += sum of the per-segment additions, n_voxels = the new count.  One workgroup.
__global__ __launch_bounds__(1024) void map_totals_kernel(const MapArrays m, const uint32_t * seg_added, uint32_t n_voxels_after)
{
  __shared__ unsigned long long s_w[16];
  const uint32_t ns = m.state->n_segments;
  unsigned long long t = 0;
  for (uint32_t s = threadIdx.x; s < ns; s += 1024u) t += seg_added[s];
#pragma unroll
  for (int mk = 32; mk >= 1; mk >>= 1) t += __shfl_xor(t, mk, 64);
  if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tot = 0;
    for (int w = 0; w < 16; ++w) tot += s_w[w];
    m.state->n_points += tot;
    m.state->n_voxels = n_voxels_after;
  }
}

__global__ __launch_bounds__(kT) void map_rehash_kernel(const int4 * old_table, uint32_t old_cap, int4 * new_table, uint32_t new_mask)
{
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < old_cap; i += gridDim.x * kT) {
    const int4 e = old_table[i];
    const uint64_t key = static_cast<uint64_t>(static_cast<uint32_t>(e.x)) | (static_cast<uint64_t>(static_cast<uint32_t>(e.y)) << 32);
    if (key == kEmptyKey) continue;
    int bx, by, bz;
    unpack_coord_key(key, bx, by, bz);
    uint32_t h = block_hash(bx, by, bz) & new_mask;
    for (;;) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&new_table[h]), static_cast<unsigned long long>(kEmptyKey),
                                               static_cast<unsigned long long>(key));
      if (old == kEmptyKey) {
        new_table[h].z = e.z;
        break;
      }
      h = (h + 1) & new_mask;
    }
  }
}

// ---- LRU purge -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void map_purge_flags_kernel(const unsigned long long * lru, uint32_t n, unsigned long long horizon,
                                                              unsigned long long counter, uint32_t * keep)
{
  for (uint32_t v = blockIdx.x * kT + threadIdx.x; v < n; v += gridDim.x * kT) keep[v] = (lru[v] + horizon < counter) ? 0u : 1u;
}
__global__ __launch_bounds__(kT) void map_purge_count_kernel(const uint32_t * keep, const uint32_t * pos, uint32_t n, MapState * st)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) st->n_keep = n ? pos[n - 1] + keep[n - 1] : 0u;
}
__global__ __launch_bounds__(kT) void map_purge_compact_kernel(const MapArrays src, const MapArrays dst, uint32_t n, const uint32_t * keep,
                                                                const uint32_t * pos)
{
  const uint32_t total = n * static_cast<uint32_t>(kBucketStride);
  unsigned long long pts = 0;
  for (uint32_t e = blockIdx.x * kT + threadIdx.x; e < total; e += gridDim.x * kT) {
    const uint32_t v = e / kBucketStride, slot = e % kBucketStride;
    if (!keep[v]) continue;
    const uint32_t d = pos[v];
    dst.buckets[static_cast<size_t>(d) * kBucketStride + slot] = src.buckets[e];
    dst.qbuckets[static_cast<size_t>(d) * kBucketStride + slot] = src.qbuckets[e];
    if (slot == 0) {
      const int4 vx = src.vox[v];
      dst.vox[d] = vx;
      dst.lru[d] = src.lru[v];
      pts += static_cast<unsigned long long>(vx.w);
    }
  }
  // one atomic per wave, not per thread
#pragma unroll
  for (int mk = 32; mk >= 1; mk >>= 1) pts += __shfl_xor(pts, mk, 64);
  if ((threadIdx.x & 63u) == 0u && pts) atomicAdd(&dst.state->n_points, pts);
}

__global__ __launch_bounds__(kT) void map_claim_blocks_kernel(const MapArrays m, uint32_t v0, uint32_t v1)
{
  for (uint32_t v = v0 + blockIdx.x * kT + threadIdx.x; v < v1; v += gridDim.x * kT) {
    const int4 vx = m.vox[v];
    for (int w = 0; w < 8; ++w) {
      int bx, by, bz;
      if (voxel_block(vx.x, vx.y, vx.z, w, bx, by, bz)) claim_block(m, bx, by, bz);
    }
  }
}
__global__ __launch_bounds__(kT) void map_write_words_kernel(const MapArrays m, uint32_t v0, uint32_t v1)
{
  const uint32_t total = (v1 - v0) * 8u;
  for (uint32_t e = blockIdx.x * kT + threadIdx.x; e < total; e += gridDim.x * kT) {
    const uint32_t v = v0 + e / 8u;
    const int4 vx = m.vox[v];
    int bx, by, bz;
    if (!voxel_block(vx.x, vx.y, vx.z, static_cast<int>(e & 7u), bx, by, bz)) continue;
    const int blk = find_block(m, bx, by, bz);
    if (blk >= 0)
      m.cells[static_cast<size_t>(blk) * kCellsPerBlock + halo_index(vx.x - bx * kBlockDim, vx.y - by * kBlockDim, vx.z - bz * kBlockDim)] =
        (v << 5) | static_cast<uint32_t>(vx.w);
  }
}

// ---- voxel_data() ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void map_counts_kernel(const int4 * vox, uint32_t n, uint32_t * counts)
{
  for (uint32_t v = blockIdx.x * kT + threadIdx.x; v < n; v += gridDim.x * kT) counts[v] = static_cast<uint32_t>(vox[v].w);
}
__global__ __launch_bounds__(kT) void map_cloud_kernel(const MapArrays m, uint32_t n, const uint32_t * counts, const uint32_t * offsets, float * out,
                                                        size_t cap_points)
{
  const uint32_t total = n * static_cast<uint32_t>(kBucketStride);
  for (uint32_t e = blockIdx.x * kT + threadIdx.x; e < total; e += gridDim.x * kT) {
    const uint32_t v = e / kBucketStride, slot = e % kBucketStride;
    if (slot == 0 && v == n - 1) m.state->cloud_points = offsets[v] + counts[v];
    if (slot >= counts[v]) continue;
    const size_t o = static_cast<size_t>(offsets[v]) + slot;
    if (o >= cap_points) continue;
    const float4 p = m.buckets[e];
    out[3 * o] = p.x;
    out[3 * o + 1] = p.y;
    out[3 * o + 2] = p.z;
  }
}

hipError_t exclusive_sum(const uint32_t * in, uint32_t * out, uint32_t n, void * temp, size_t temp_bytes, hipStream_t stream)
{
  size_t tb = temp_bytes;
  return rocprim::exclusive_scan(temp, tb, in, out, 0u, static_cast<size_t>(n), rocprim::plus<uint32_t>(), stream);
}
}  // namespace

size_t map_temp_bytes(size_t n)
{
  if (n == 0) n = 1;
  size_t best = 0, tb = 0;
  uint32_t * k32 = nullptr;
  uint64_t * k64 = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, tb, k64, k64, k32, k32, n, 0, 64, hipStream_t(nullptr));
  best = tb;
  tb = 0;
  (void)rocprim::exclusive_scan(nullptr, tb, k32, k32, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(nullptr));
  best = tb > best ? tb : best;
  return best + 256;
}

size_t map_group_bytes(size_t n) { return vg::layout(n).bytes; }

hipError_t launch_map_insert_prepare(const MapArrays & m, const float * src, uint32_t n, uint32_t stride_floats, const float * Rt12,
                                     double inv_leaf, const InsertScratch & s, hipStream_t stream)
{
  const vg::Layout L = vg::layout(n);
  const vg::Buffers B = vg::carve(s.group, n);
  hipError_t e = hipMemsetAsync(s.group, 0xFF, L.clear_bytes, stream);
  if (e != hipSuccess) return e;
  const dim3 g((n + kT - 1) / kT), b(kT);
  hipLaunchKernelGGL(map_assign_kernel, g, b, 0, stream, src, n, stride_floats, Rt12, inv_leaf, s.pts, B.h, B.slot_of, m.state);
  if ((e = vg::launch_group(B, n, &m.state->n_segments, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(map_lookup_kernel, g, b, 0, stream, m, B.h, B.vox_slot, s.seg_vid, s.blk_new);  // <= n segments
  hipLaunchKernelGGL(map_new_ids_kernel, g, b, 0, stream, m, s.blk_new, s.seg_vid);
  return hipGetLastError();
}

hipError_t launch_map_create_voxels(const MapArrays & m, uint32_t n, uint32_t n_voxels_before, unsigned long long lru_counter,
                                    const InsertScratch & s, hipStream_t stream)
{
  const vg::Buffers B = vg::carve(s.group, n);
  hipLaunchKernelGGL(map_create_voxels_kernel, dim3(grid_for(n)), dim3(kT), 0, stream, m, B.h, B.vox_slot, s.seg_vid, n_voxels_before, lru_counter);
  return hipGetLastError();
}

hipError_t launch_map_insert_points(const MapArrays & m, uint32_t n, uint32_t n_voxels_after, uint32_t max_pts, double min_sq,
                                    double inv_leaf, unsigned long long lru_counter, const InsertScratch & s, hipStream_t stream)
{
  const vg::Buffers B = vg::carve(s.group, n);
  const size_t waves = static_cast<size_t>(n);  // <= one wave per point (segments <= points)
  const int grid = static_cast<int>(std::min<size_t>((waves * 64 + kT - 1) / kT, 16384));
  hipLaunchKernelGGL(map_insert_points_kernel, dim3(grid > 0 ? grid : 1), dim3(kT), 0, stream, m, s.pts, B.idx_unsorted, B.idx_sorted, B.idx_tmp,
                     B.seg, s.seg_vid, s.seg_added, max_pts, min_sq, inv_leaf, lru_counter);
  hipLaunchKernelGGL(map_totals_kernel, dim3(1), dim3(1024), 0, stream, m, s.seg_added, n_voxels_after);
  return hipGetLastError();
}

hipError_t launch_map_rehash(const int4 * old_table, uint32_t old_cap, int4 * new_table, uint32_t new_cap, hipStream_t stream)
{
  if (old_cap) hipLaunchKernelGGL(map_rehash_kernel, dim3(grid_for(old_cap)), dim3(kT), 0, stream, old_table, old_cap, new_table, new_cap - 1);
  return hipGetLastError();
}

hipError_t launch_map_purge_flags(const MapArrays & m, uint32_t n_voxels, unsigned long long horizon, unsigned long long lru_counter,
                                  uint32_t * keep, uint32_t * pos, void * temp, size_t temp_bytes, hipStream_t stream)
{
  if (!n_voxels) return hipSuccess;
  hipLaunchKernelGGL(map_purge_flags_kernel, dim3(grid_for(n_voxels)), dim3(kT), 0, stream, m.lru, n_voxels, horizon, lru_counter, keep);
  const hipError_t e = exclusive_sum(keep, pos, n_voxels, temp, temp_bytes, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(map_purge_count_kernel, dim3(1), dim3(kT), 0, stream, keep, pos, n_voxels, m.state);
  return hipGetLastError();
}

hipError_t launch_map_purge_compact(const MapArrays & src, const MapArrays & dst, uint32_t n_voxels, const uint32_t * keep,
                                    const uint32_t * pos, hipStream_t stream)
{
  if (!n_voxels) return hipSuccess;
  hipLaunchKernelGGL(map_purge_compact_kernel, dim3(grid_for(n_voxels * kBucketStride)), dim3(kT), 0, stream, src, dst, n_voxels, keep, pos);
  return hipGetLastError();
}

hipError_t launch_map_claim_blocks(const MapArrays & m, uint32_t v0, uint32_t v1, hipStream_t stream)
{
  if (v1 > v0) hipLaunchKernelGGL(map_claim_blocks_kernel, dim3(grid_for(v1 - v0)), dim3(kT), 0, stream, m, v0, v1);
  return hipGetLastError();
}
hipError_t launch_map_write_words(const MapArrays & m, uint32_t v0, uint32_t v1, hipStream_t stream)
{
  if (v1 > v0) hipLaunchKernelGGL(map_write_words_kernel, dim3(grid_for((v1 - v0) * 8u)), dim3(kT), 0, stream, m, v0, v1);
  return hipGetLastError();
}

hipError_t launch_map_cloud(const MapArrays & m, uint32_t n_voxels, uint32_t * counts, uint32_t * offsets, float * out, size_t out_capacity_points,
                            void * temp, size_t temp_bytes, hipStream_t stream)
{
  if (!n_voxels) return hipSuccess;
  hipLaunchKernelGGL(map_counts_kernel, dim3(grid_for(n_voxels)), dim3(kT), 0, stream, m.vox, n_voxels, counts);
  const hipError_t e = exclusive_sum(counts, offsets, n_voxels, temp, temp_bytes, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(map_cloud_kernel, dim3(grid_for(n_voxels * kBucketStride)), dim3(kT), 0, stream, m, n_voxels, counts, offsets, out,
                     out_capacity_points);
  return hipGetLastError();
}

}  // namespace mh
