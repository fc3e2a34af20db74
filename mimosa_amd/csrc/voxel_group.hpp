// Grouping a batch of points by voxel WITHOUT a library sort — shared by the scan front end's down-sampler
// (scan_kernels.hip: Geometric::downsample, geometric.cpp:55-126) and the map insert (map_kernels.hip: iVox::insert).
// Both rules are sequential in the reference but only couple the points of one voxel, in input order, and both number
// their voxels in first-seen order.  The device form:
//   assign     (caller's kernel, voxel_assign below)  voxel key -> open-addressing hash: slot of the voxel, atomicMin of the
//              first input index, atomicAdd of the point count; one probe + one pair of atomics per run of consecutive
//              lanes in the same voxel
//   count / offsets   two-kernel scan over input positions of "points of the voxel whose first point I am": every voxel
//              gets a segment, voxels in first-seen order; group list (start, length) [+ hash slot]
//   scatter    every point into its voxel's segment (atomic cursor: unordered inside the segment)
//   sort_segment_indices (device function, called by the per-voxel wave of the consumer kernel): the segment's indices
//              ascending = input order.  <= 64: rank sort in registers; <= kLdsSort: wave-level binary LSD radix in LDS;
//              above: the same radix on global scratch.
// Everything lives in an anonymous namespace: each including translation unit gets its own copy of the kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mh
{
// the voxel hash (open addressing, 64-bit packed coordinates).  keys / first / cnt / cur / bad are one allocation
// cleared to all-ones per call: empty key, first = UINT_MAX for atomicMin, cnt and cur count from -1.
struct VoxelHash
{
  uint64_t * keys;
  uint32_t * first;  // smallest input index of the voxel's points
  uint32_t * cnt;    // number of points - 1
  uint32_t * cur;    // scatter cursor - 1
  uint32_t * bad;    // != all-ones: a coordinate was out of the key range
  uint32_t * off;    // start of the voxel's segment in the index list
  uint32_t mask;
};

namespace vg
{
namespace
{
constexpr int kThreads = 256;
constexpr int kItems = 1;                         // consecutive elements per thread in the blocked kernels: these kernels are
                                                  // latency-bound (<= 131 072 elements), more blocks beat wider threads
constexpr uint32_t kBlockItems = kThreads * kItems;
constexpr uint32_t kEmpty32 = 0xFFFFFFFFu;
constexpr uint64_t kEmpty64 = ~0ull;
constexpr uint32_t kLdsSort = 1024;               // segments up to this long are sorted in LDS (2 x 4 KiB per wave)

inline uint32_t blocks_for(uint32_t n) { return n ? (n + kBlockItems - 1) / kBlockItems : 1u; }
inline uint32_t pow2_at_least(uint64_t v)
{
  uint64_t c = 1024;
  while (c < v) c <<= 1;
  return static_cast<uint32_t>(c);
}

// Scratch of one grouping of n points: [hash: keys, first, cnt, cur, bad | off | seg (uint2 n) | slot_of, idx_unsorted,
// idx_sorted, idx_tmp, vox_slot (n each) | blk_pts, blk_vox (n_blocks each)]; the first clear_bytes are memset to 0xFF.
struct Layout
{
  uint32_t n_blocks, cap;
  size_t clear_bytes, bytes;
};
inline Layout layout(size_t n)
{
  Layout L;
  L.n_blocks = blocks_for(static_cast<uint32_t>(n));
  L.cap = pow2_at_least(2 * static_cast<uint64_t>(n));
  L.clear_bytes = static_cast<size_t>(L.cap) * (8 + 4 + 4 + 4) + 16;
  const size_t m = n ? n : 1;
  L.bytes = L.clear_bytes + static_cast<size_t>(L.cap) * 4 + (7 * m + 2 * static_cast<size_t>(L.n_blocks)) * 4;
  return L;
}
struct Buffers
{
  VoxelHash h;
  uint2 * seg;  // group g (first-seen order): (start, length) of its segment
  uint32_t * slot_of, * idx_unsorted, * idx_sorted, * idx_tmp, * vox_slot, * blk_pts, * blk_vox;
};
inline Buffers carve(void * scratch, size_t n)
{
  const Layout L = layout(n);
  Buffers B;
  char * p = static_cast<char *>(scratch);
  B.h.keys = reinterpret_cast<uint64_t *>(p);
  p += static_cast<size_t>(L.cap) * 8;
  B.h.first = reinterpret_cast<uint32_t *>(p);
  p += static_cast<size_t>(L.cap) * 4;
  B.h.cnt = reinterpret_cast<uint32_t *>(p);
  p += static_cast<size_t>(L.cap) * 4;
  B.h.cur = reinterpret_cast<uint32_t *>(p);
  p += static_cast<size_t>(L.cap) * 4;
  B.h.bad = reinterpret_cast<uint32_t *>(p);
  p += 16;
  B.h.off = reinterpret_cast<uint32_t *>(p);
  p += static_cast<size_t>(L.cap) * 4;
  B.h.mask = L.cap - 1u;
  uint32_t * w = reinterpret_cast<uint32_t *>(p);
  const size_t m = n ? n : 1;
  B.seg = reinterpret_cast<uint2 *>(w);  // 8-byte aligned: everything before it is a multiple of 8 bytes
  B.slot_of = w + 2 * m;
  B.idx_unsorted = w + 3 * m;
  B.idx_sorted = w + 4 * m;
  B.idx_tmp = w + 5 * m;
  B.vox_slot = w + 6 * m;
  B.blk_pts = w + 7 * m;
  B.blk_vox = B.blk_pts + L.n_blocks;
  return B;
}

// ---- block-level helpers (256 threads = 4 waves) ------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d);
    if (lane >= static_cast<uint32_t>(d)) v += o;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// exclusive prefix of (a, b) over the block's threads + block totals.  lds: 8 words.
__device__ __forceinline__ void block_exclusive_sum2(uint32_t a, uint32_t b, uint32_t & ea, uint32_t & eb, uint32_t & ta,
                                                     uint32_t & tb, uint32_t * lds)
{
  const uint32_t ia = wave_inclusive_sum(a), ib = wave_inclusive_sum(b);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (lane == 63u) {
    lds[wave] = ia;
    lds[4 + wave] = ib;
  }
  __syncthreads();
  uint32_t oa = 0, ob = 0;
  ta = tb = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4; ++w) {
    const uint32_t xa = lds[w], xb = lds[4 + w];
    if (w < wave) {
      oa += xa;
      ob += xb;
    }
    ta += xa;
    tb += xb;
  }
  ea = oa + ia - a;
  eb = ob + ib - b;
  __syncthreads();
}

// sum of blk_a[0..b) and blk_b[0..b): the block's offset in a two-kernel (count, then place) compaction
__device__ __forceinline__ void block_offsets2(const uint32_t * blk_a, const uint32_t * blk_b, uint32_t b, uint32_t & off_a,
                                               uint32_t & off_b, uint32_t * lds)
{
  uint32_t sa = 0, sb = 0;
  for (uint32_t i = threadIdx.x; i < b; i += kThreads) {
    sa += blk_a[i];
    sb += blk_b[i];
  }
  sa = wave_sum(sa);
  sb = wave_sum(sb);
  if ((threadIdx.x & 63u) == 0) {
    lds[threadIdx.x >> 6] = sa;
    lds[4 + (threadIdx.x >> 6)] = sb;
  }
  __syncthreads();
  off_a = lds[0] + lds[1] + lds[2] + lds[3];
  off_b = lds[4] + lds[5] + lds[6] + lds[7];
  __syncthreads();
}

__device__ __forceinline__ float lane_value(float v, uint32_t lane)  // lane: wave-uniform
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), static_cast<int>(lane)));
}

__device__ __forceinline__ uint32_t mix64(uint64_t k)
{
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return static_cast<uint32_t>(k);
}

// Runs of consecutive lanes with the same voxel (neighbouring columns of one ring mostly are): one hash probe and one
// set of atomics per run instead of per point.  head_lane = the run's first lane, run_len on the head lane.
struct LaneRun
{
  bool head;
  uint32_t head_lane, run_len;
};
__device__ __forceinline__ LaneRun lane_runs(bool valid, bool differs_from_previous_lane)
{
  const uint32_t lane = threadIdx.x & 63u;
  LaneRun r;
  r.head = valid && (lane == 0u || differs_from_previous_lane);
  const uint64_t heads = __ballot(r.head);
  const uint32_t n_valid = static_cast<uint32_t>(__popcll(__ballot(valid)));  // valid lanes are a prefix of the wave
  const uint64_t upto = heads & ((lane == 63u) ? ~0ull : ((2ull << lane) - 1ull));
  r.head_lane = upto ? 63u - static_cast<uint32_t>(__clzll(upto)) : 0u;
  const uint64_t later = lane == 63u ? 0ull : (heads >> (lane + 1u));
  const uint32_t end = later ? lane + static_cast<uint32_t>(__ffsll(static_cast<long long>(later))) : n_valid;
  r.run_len = end - lane;
  return r;
}

// The hash part of an assign kernel: `key` of this lane's point (kEmpty64 for lanes past the end), j its input index.
// Returns the slot; the run's head lane did the probe and the atomics.
__device__ __forceinline__ uint32_t voxel_assign(const VoxelHash & h, bool valid, uint64_t key, uint32_t j)
{
  const uint32_t klo = static_cast<uint32_t>(key), khi = static_cast<uint32_t>(key >> 32);
  const LaneRun run = lane_runs(valid, __shfl_up(klo, 1) != klo || __shfl_up(khi, 1) != khi);
  uint32_t slot = 0;
  if (run.head) {
    slot = mix64(key) & h.mask;
    for (;;) {
      unsigned long long cur = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&h.keys[slot]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == kEmpty64) {
        cur = atomicCAS(reinterpret_cast<unsigned long long *>(&h.keys[slot]), static_cast<unsigned long long>(kEmpty64),
                        static_cast<unsigned long long>(key));
        if (cur == kEmpty64) cur = key;
      }
      if (cur == key) break;
      slot = (slot + 1) & h.mask;
    }
    atomicMin(&h.first[slot], j);            // all-ones before: the voxel's first point in input order (the head has the run's smallest j)
    atomicAdd(&h.cnt[slot], run.run_len);    // all-ones before: stored value = count - 1
  }
  return __shfl(slot, static_cast<int>(run.head_lane));
}

// value of input position j in the two scans: (points of the voxel, 1) if j is the first point of its voxel, else (0, 0)
__device__ __forceinline__ void voxel_head(const VoxelHash & h, uint32_t slot, uint32_t j, uint32_t & c, uint32_t & v)
{
  const bool head = h.first[slot] == j;
  c = head ? h.cnt[slot] + 1u : 0u;
  v = head ? 1u : 0u;
}

__global__ __launch_bounds__(kThreads) void voxel_count_kernel(const uint32_t * __restrict__ slot_of, uint32_t n, VoxelHash h,
                                                                uint32_t * __restrict__ blk_pts, uint32_t * __restrict__ blk_vox)
{
  __shared__ uint32_t lds[8];
  const uint32_t base = blockIdx.x * kBlockItems + threadIdx.x * kItems;
  uint32_t c = 0, v = 0;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k)
    if (base + k < n) {
      uint32_t ck, vk;
      voxel_head(h, slot_of[base + k], base + k, ck, vk);
      c += ck;
      v += vk;
    }
  c = wave_sum(c);
  v = wave_sum(v);
  if ((threadIdx.x & 63u) == 0) {
    lds[threadIdx.x >> 6] = c;
    lds[4 + (threadIdx.x >> 6)] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_pts[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
    blk_vox[blockIdx.x] = lds[4] + lds[5] + lds[6] + lds[7];
  }
}

// off[slot] = start of the voxel's segment (voxels in first-seen order); vox_seg[v] = (start, length) of the v-th voxel
__global__ __launch_bounds__(kThreads) void voxel_offsets_kernel(const uint32_t * __restrict__ slot_of, uint32_t n, VoxelHash h,
                                                                  const uint32_t * __restrict__ blk_pts,
                                                                  const uint32_t * __restrict__ blk_vox, uint2 * __restrict__ vox_seg,
                                                                  uint32_t * __restrict__ vox_slot, uint32_t * n_groups)
{
  __shared__ uint32_t lds[8];
  uint32_t off_pts, off_vox;
  block_offsets2(blk_pts, blk_vox, blockIdx.x, off_pts, off_vox, lds);
  const uint32_t base = blockIdx.x * kBlockItems + threadIdx.x * kItems;
  uint32_t ck[kItems], vk[kItems], sl[kItems], c = 0, v = 0;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k) {
    ck[k] = vk[k] = sl[k] = 0;
    if (base + k < n) {
      sl[k] = slot_of[base + k];
      voxel_head(h, sl[k], base + k, ck[k], vk[k]);
      c += ck[k];
      v += vk[k];
    }
  }
  uint32_t ec, ev, tc, tv;
  block_exclusive_sum2(c, v, ec, ev, tc, tv, lds);
  uint32_t pc = off_pts + ec, pv = off_vox + ev;
#pragma unroll
  for (uint32_t k = 0; k < kItems; ++k)
    if (vk[k]) {
      h.off[sl[k]] = pc;
      vox_seg[pv] = make_uint2(pc, ck[k]);
      if (vox_slot) vox_slot[pv] = sl[k];
      pc += ck[k];
      ++pv;
    }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_groups = off_vox + tv;
}

__global__ __launch_bounds__(kThreads) void voxel_scatter_kernel(const uint32_t * __restrict__ slot_of, uint32_t n, VoxelHash h,
                                                                  uint32_t * __restrict__ idx_unsorted)
{
  const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
  const bool valid = j < n;
  const uint32_t slot = valid ? slot_of[j] : kEmpty32;
  const LaneRun run = lane_runs(valid, __shfl_up(slot, 1) != slot);
  uint32_t at = 0;
  if (run.head) at = h.off[slot] + (atomicAdd(&h.cur[slot], run.run_len) + 1u);  // cursor starts at all-ones
  at = __shfl(at, static_cast<int>(run.head_lane));
  if (valid) idx_unsorted[at + ((threadIdx.x & 63u) - run.head_lane)] = j;
}

// One wave sorts len distinct values ascending: binary LSD radix, one stable split per bit that varies, ping-pong between
// a and b so that the last pass lands in a.  src / a / b: LDS or global (distinct arrays).
__device__ __forceinline__ void wave_radix_sort(const uint32_t * src, uint32_t * a, uint32_t * b, uint32_t len)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t lanes_below = (1ull << lane) - 1ull;
  uint32_t zeros = 0, diff = 0;
  const uint32_t e0 = src[0];
  for (uint32_t c0 = 0; c0 < len; c0 += 64u) {
    const bool valid = c0 + lane < len;
    const uint32_t e = valid ? src[c0 + lane] : e0;
    diff |= e ^ e0;
    zeros += static_cast<uint32_t>(__popcll(__ballot(valid && !(e & 1u))));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) diff |= __shfl_xor(diff, d);
  const uint32_t n_bits = diff ? 32u - static_cast<uint32_t>(__clz(diff)) : 1u;
  for (uint32_t bit = 0; bit < n_bits; ++bit) {
    uint32_t * dst = ((n_bits - 1u - bit) & 1u) ? b : a;
    uint32_t z_run = 0, o_run = zeros, next_zeros = 0;
    for (uint32_t c0 = 0; c0 < len; c0 += 64u) {
      const bool valid = c0 + lane < len;
      const uint32_t e = valid ? src[c0 + lane] : 0u;
      const bool one = (e >> bit) & 1u;
      const uint64_t m0 = __ballot(valid && !one), m1 = __ballot(valid && one);
      if (valid) dst[one ? o_run + static_cast<uint32_t>(__popcll(m1 & lanes_below)) : z_run + static_cast<uint32_t>(__popcll(m0 & lanes_below))] = e;
      next_zeros += static_cast<uint32_t>(__popcll(__ballot(valid && !((e >> (bit + 1u)) & 1u))));
      z_run += static_cast<uint32_t>(__popcll(m0));
      o_run += static_cast<uint32_t>(__popcll(m1));
    }
    __threadfence_block();
    src = dst;
    zeros = next_zeros;
  }
}

// The indices of one group (segment [s0, s0 + len) of idx_unsorted), ascending.  Called by one whole wave; lds_a / lds_b:
// kLdsSort words each, private to the wave.  Where the sorted list ends up: len <= 64 -> the return value of lane i is
// element i; len <= kLdsSort -> lds_a; above -> idx_sorted[s0 ...] (global, visible to this wave).
__device__ __forceinline__ uint32_t sort_segment_indices(const uint32_t * __restrict__ idx_unsorted, uint32_t * idx_sorted, uint32_t * idx_tmp,
                                                         uint32_t * lds_a, uint32_t * lds_b, uint32_t s0, uint32_t len)
{
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t first_chunk = 0;
  if (len <= 64u) {
    const uint32_t e = lane < len ? idx_unsorted[s0 + lane] : kEmpty32;
    uint32_t rank = 0;
    for (uint32_t u = 0; u < len; ++u)
      rank += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e), static_cast<int>(u))) < e ? 1u : 0u;
    if (lane >= len) rank = lane;
    first_chunk = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>(rank << 2), static_cast<int>(e)));  // lane r <- the value of rank r
  } else if (len <= kLdsSort) {
    wave_radix_sort(idx_unsorted + s0, lds_a, lds_b, len);
  } else {
    wave_radix_sort(idx_unsorted + s0, idx_sorted + s0, idx_tmp + s0, len);
  }
  return first_chunk;
}
// element at position s (>= s0) of a sorted group, after sort_segment_indices
__device__ __forceinline__ uint32_t sorted_index_at(uint32_t first_chunk, const uint32_t * lds_a, const uint32_t * idx_sorted, uint32_t s0,
                                                    uint32_t len, uint32_t s)
{
  if (len <= 64u) return first_chunk;
  if (len <= kLdsSort) return lds_a[s - s0];
  return __hip_atomic_load(&idx_sorted[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// memset + count / offsets / scatter after the caller's assign kernel filled B.slot_of and the hash.  *n_groups receives the
// number of voxels touched.
inline hipError_t launch_group(const Buffers & B, uint32_t n, uint32_t * n_groups, hipStream_t stream)
{
  const Layout L = layout(n);
  const dim3 gp((n + kThreads - 1) / kThreads), gb(L.n_blocks), b(kThreads);
  hipLaunchKernelGGL(voxel_count_kernel, gb, b, 0, stream, B.slot_of, n, B.h, B.blk_pts, B.blk_vox);
  hipLaunchKernelGGL(voxel_offsets_kernel, gb, b, 0, stream, B.slot_of, n, B.h, B.blk_pts, B.blk_vox, B.seg, B.vox_slot, n_groups);
  hipLaunchKernelGGL(voxel_scatter_kernel, gp, b, 0, stream, B.slot_of, n, B.h, B.idx_unsorted);
  return hipGetLastError();
}

}  // namespace
}  // namespace vg
}  // namespace mh
