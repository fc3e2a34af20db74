// Host side of the device-resident incremental voxel map (the iVox counterpart).
//
// Replaces gtsam_points::iVox as used through IncrementalVoxelMapPCL
// (reference: include/mimosa/lidar/incremental_voxel_map.hpp:22-54,
// src/lidar/incremental_voxel_map.cpp:14-62; configuration src/lidar/geometric.cpp:23-28).
//
// Same observable semantics — greedy first-come-first-kept insertion in input order with a
// min-distance rule and a per-voxel cap, voxels numbered in creation order, LRU purge every
// lru_clear_cycle inserts, neighbour traversal in offset-generation order — but laid out for HBM,
// not for pointer chasing:
//
//   buckets : float4[n_voxels * 20]        one 320-byte bucket per voxel, points in insertion order
//   qbuckets: uint32[n_voxels * 20]        the same points, 3 x 10-bit voxel-relative fixed point
//                                          (x | y << 10 | z << 20, cell = leaf / 1024): the coarse k-NN
//                                          tier reads 4 candidates per 16-byte load; selection is
//                                          always re-decided on the exact float4 copy
//   cells   : uint32[n_blocks * 216]       4x4x4-voxel blocks stored WITH A ONE-VOXEL HALO (6x6x6 words,
//                                          z fastest); entry = voxel_id << 5 | count, ~0u = empty
//   table   : int4[capacity]               open-addressing hash of BLOCK coords -> block id (w), -1 empty
//
// Every voxel is written into the table of its home block and into the halo of each adjacent block it
// touches (<= 8 tables; a table is created as soon as any voxel falls in its halo).  All 27 neighbours
// of any voxel of a block therefore sit in THAT block's table: a query costs ONE hash probe plus nine
// 12-byte loads (one z-triple per (dx, dy) column) instead of 19 independent hash probes — the lookup
// is bound by the number of per-lane L1 transactions, not by bytes.
// Map points are float32-valued in the reference too (PointCloudCPU is built from Vector3f,
// incremental_voxel_map.cpp:40-48), so float4 storage is lossless; all distance arithmetic is fp64.
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mimosa_hip.h"

namespace mh
{
// std::vector whose resize(n) leaves trivially-constructible elements uninitialised: the big arrays of a map
// copy are sized first and then filled by several threads (first touch and memcpy in parallel) instead of one
// thread faulting in and copying ~270 MB.
// Freed blocks of >= 1 MiB are kept (at most kMaxBlocks of them) and handed out again: a keyframe update
// copies the map, and the previous copy's arrays were released moments earlier — reusing their pages avoids
// faulting in ~300 MB of fresh memory per keyframe (page faults, not bandwidth, dominated the copy).
class BigBlockCache
{
public:
  static void * take(size_t bytes)
  {
    std::lock_guard<std::mutex> g(mu());
    auto & v = blocks();
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i].cap >= bytes && v[i].cap <= bytes + bytes / 2 + (size_t(1) << 20) && (best == v.size() || v[i].cap < v[best].cap)) best = i;
    if (best == v.size()) return nullptr;
    void * p = v[best].p;
    live()[p] = v[best].cap;
    v.erase(v.begin() + static_cast<long>(best));
    return p;
  }
  static void * fresh(size_t bytes)
  {
    void * p = ::operator new(bytes);
    if (bytes >= kMinBytes) {
      std::lock_guard<std::mutex> g(mu());
      live()[p] = bytes;
    }
    return p;
  }
  static void give(void * p, size_t bytes_hint)
  {
    size_t cap = 0;
    {
      std::lock_guard<std::mutex> g(mu());
      auto it = live().find(p);
      if (it != live().end()) {
        cap = it->second;
        live().erase(it);
      }
      if (cap >= kMinBytes && blocks().size() < kMaxBlocks) {
        blocks().push_back({p, cap});
        return;
      }
    }
    (void)bytes_hint;
    ::operator delete(p);
  }
  static constexpr size_t kMinBytes = size_t(1) << 20;
  static constexpr size_t kMaxBlocks = 12;

private:
  struct Block
  {
    void * p;
    size_t cap;
  };
  static std::mutex & mu()
  {
    static std::mutex m;
    return m;
  }
  static std::vector<Block> & blocks()
  {
    static std::vector<Block> v;
    return v;
  }
  static std::unordered_map<void *, size_t> & live()
  {
    static std::unordered_map<void *, size_t> m;
    return m;
  }
};

template <typename T>
struct NoInitAlloc
{
  using value_type = T;
  NoInitAlloc() = default;
  template <typename U>
  NoInitAlloc(const NoInitAlloc<U> &) {}
  T * allocate(size_t n)
  {
    const size_t bytes = n * sizeof(T);
    if (bytes >= BigBlockCache::kMinBytes)
      if (void * p = BigBlockCache::take(bytes)) return static_cast<T *>(p);
    return static_cast<T *>(BigBlockCache::fresh(bytes));
  }
  void deallocate(T * p, size_t n) { BigBlockCache::give(p, n * sizeof(T)); }
  template <typename U, typename... A>
  void construct(U * p, A &&... a)
  {
    if constexpr (sizeof...(A) == 0)
      ::new (static_cast<void *>(p)) U;
    else
      ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
  }
  template <typename U>
  bool operator==(const NoInitAlloc<U> &) const { return true; }
  template <typename U>
  bool operator!=(const NoInitAlloc<U> &) const { return false; }
};
template <typename T>
using BigVec = std::vector<T, NoInitAlloc<T>>;

constexpr int kBlockLog2 = 2;                       // 4x4x4 voxels per block
constexpr int kBlockDim = 1 << kBlockLog2;
constexpr int kHaloDim = kBlockDim + 2;                // block + one-voxel halo
constexpr int kCellsPerBlock = kHaloDim * kHaloDim * kHaloDim;  // 216 words per block table
constexpr int kBucketStride = 20;                   // FlatContainer max_num_points_in_cell
constexpr uint32_t kEmptyCell = 0xFFFFFFFFu;
constexpr int kQuantBits = 10;                     // coarse copy: leaf / 1024 resolution

struct Int4
{
  int32_t x, y, z, w;
};
struct Float4
{
  float x, y, z, w;
};

// include/mimosa/lidar/utils.hpp:218-222 (same helper in gtsam_points/util/fast_floor.hpp)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int fast_floor(double v)
{
  const int n = static_cast<int>(v);
  return n - (v < static_cast<double>(n) ? 1 : 0);
}

// Index of the voxel with block-local coordinates l = c - 4 b, each in [-1, 4], inside a halo'd table.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int halo_index(int lx, int ly, int lz)
{
  return ((lx + 1) * kHaloDim + (ly + 1)) * kHaloDim + (lz + 1);
}

// Hash of a block coordinate.  Internal to the table (not observable), so the 32-bit Teschner
// primes + a finaliser are used instead of the 64-bit XORVector3iHash multiplies
// (include/mimosa/lidar/utils.hpp:228-238): 3 v_mul_lo_u32 on the device instead of 3 64-bit muls.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t block_hash(int bx, int by, int bz)
{
  uint32_t h = (static_cast<uint32_t>(bx) * 73856093u) ^ (static_cast<uint32_t>(by) * 19349663u) ^
               (static_cast<uint32_t>(bz) * 83492791u);
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return h;
}

inline int neighbor_offsets(int mode, int8_t out[27][3])
{
  // gtsam_points neighbor_offsets(): generation order is observable through tie-breaking
  int n = 0;
  auto push = [&](int i, int j, int k) {
    out[n][0] = static_cast<int8_t>(i);
    out[n][1] = static_cast<int8_t>(j);
    out[n][2] = static_cast<int8_t>(k);
    ++n;
  };
  if (mode == 1) {
    push(0, 0, 0);
  } else if (mode == 7) {
    push(0, 0, 0);
    push(1, 0, 0);
    push(-1, 0, 0);
    push(0, 1, 0);
    push(0, -1, 0);
    push(0, 0, 1);
    push(0, 0, -1);
  } else if (mode == 19 || mode == 27) {
    for (int i = -1; i <= 1; ++i)
      for (int j = -1; j <= 1; ++j)
        for (int k = -1; k <= 1; ++k) {
          if (mode == 19 && (i != 0) && (j != 0) && (k != 0)) continue;
          push(i, j, k);
        }
  }
  return n;
}

class HostVoxelMap
{
public:
  explicit HostVoxelMap(const mh_map_config & cfg)
  : cfg_(cfg), inv_leaf_(1.0 / cfg.leaf_size), min_sq_(cfg.min_dist_in_cell * cfg.min_dist_in_cell)
  {
    rehash_blocks(1024);
  }

  // Copy with growth headroom: a plain vector copy has capacity == size, so the first voxel created in the
  // copy (Geometric::updateMap inserts right after copying, geometric.cpp:494-495) would reallocate and
  // move the whole bucket array again.
  HostVoxelMap(const HostVoxelMap & o) { copy_from(o); }
  HostVoxelMap & operator=(const HostVoxelMap & o)
  {
    if (this != &o) copy_from(o);
    return *this;
  }
  // mh_map_fork: the whole structure changes hands in O(1); `o` is left an empty map of the same configuration
  void take_from(HostVoxelMap & o)
  {
    HostVoxelMap fresh(o.cfg_);
    swap_all(o);
    o.swap_all(fresh);
  }

  const mh_map_config & config() const { return cfg_; }
  double inv_leaf() const { return inv_leaf_; }
  size_t n_voxels() const { return vox_count_.size(); }
  size_t n_blocks() const { return n_blocks_; }
  size_t n_points() const { return n_points_; }
  uint32_t table_mask() const { return table_mask_; }
  const std::vector<Int4> & table() const { return table_; }
  const BigVec<uint32_t> & cells() const { return cells_; }
  const BigVec<Float4> & buckets() const { return buckets_; }
  const BigVec<uint32_t> & qbuckets() const { return qbuckets_; }
  const std::vector<uint8_t> & counts() const { return vox_count_; }
  // positions (indices into cells()) that hold this voxel's word: home table + adjacent halos
  int voxel_cell_positions(uint32_t vid, const uint32_t ** pos) const
  {
    *pos = &vox_cells_[static_cast<size_t>(vid) * 8];
    return vox_ncells_[vid];
  }

  // Dirty tracking for the device mirror.
  bool structure_changed() const { return structure_changed_; }
  const std::vector<uint32_t> & dirty_voxels() const { return dirty_; }
  void clear_dirty()
  {
    for (uint32_t v : dirty_) dirty_flag_[v] = 0;
    dirty_.clear();
    structure_changed_ = false;
  }

  // iVox::insert: in input order; LRU bookkeeping after the batch.
  void insert(const float * xyz, size_t n, size_t stride)
  {
    const size_t max_pts = static_cast<size_t>(cfg_.max_points_in_cell);
    // The loop below is bound by cache misses (cell word, per-voxel counters, bucket lines of a 300 MB structure,
    // ~8 per point).  A look-ahead of kAhead points issues prefetches for what the loop will touch — hints only:
    // every decision is still taken by the sequential code, in input order.
    constexpr size_t kAhead = 12;
    for (size_t i = 0; i < n; ++i) {
      if (i + kAhead < n) prefetch_for(xyz + (i + kAhead) * stride, 0);      // block table -> cell word
      if (i + kAhead / 2 < n) prefetch_for(xyz + (i + kAhead / 2) * stride, 1);  // cell word -> per-voxel data, bucket
      const float fx = xyz[i * stride + 0], fy = xyz[i * stride + 1], fz = xyz[i * stride + 2];
      const double px = fx, py = fy, pz = fz;
      const int cx = fast_floor(px * inv_leaf_), cy = fast_floor(py * inv_leaf_), cz = fast_floor(pz * inv_leaf_);
      const uint32_t vid = find_or_create_voxel(cx, cy, cz);
      vox_lru_[vid] = lru_counter_;
      const size_t cnt = vox_count_[vid];
      if (cnt >= max_pts) continue;
      Float4 * b = &buckets_[static_cast<size_t>(vid) * kBucketStride];
      bool close = false;
      for (size_t j = 0; j < cnt; ++j) {
        const double dx = static_cast<double>(b[j].x) - px, dy = static_cast<double>(b[j].y) - py,
                     dz = static_cast<double>(b[j].z) - pz;
        // Eigen SSE2 Vector4d squaredNorm order: (dx2 + dz2) + (dy2 + dw2), dw = 0
        if ((dx * dx + dz * dz) + (dy * dy + 0.0) < min_sq_) {
          close = true;
          break;
        }
      }
      if (close) continue;
      b[cnt] = Float4{fx, fy, fz, 1.0f};
      {  // coarse copy: floor(frac(p * inv_leaf) * 1024) per axis (frac is in [0,1) by construction of c)
        auto qz = [](double v, int c) {
          int u = static_cast<int>((v - static_cast<double>(c)) * static_cast<double>(1 << kQuantBits));
          return static_cast<uint32_t>(u < 0 ? 0 : (u > (1 << kQuantBits) - 1 ? (1 << kQuantBits) - 1 : u));
        };
        qbuckets_[static_cast<size_t>(vid) * kBucketStride + cnt] =
          qz(px * inv_leaf_, cx) | (qz(py * inv_leaf_, cy) << kQuantBits) | (qz(pz * inv_leaf_, cz) << (2 * kQuantBits));
      }
      vox_count_[vid] = static_cast<uint8_t>(cnt + 1);
      write_voxel_word(vid, (vid << 5) | static_cast<uint32_t>(cnt + 1));
      ++n_points_;
      mark_dirty(vid);
    }
    if ((++lru_counter_) % static_cast<uint64_t>(cfg_.lru_clear_cycle) == 0) purge_lru();
  }

  // voxel_data(): all points in voxel (creation) order
  size_t get_cloud(float * xyz, size_t capacity) const
  {
    size_t n = 0;
    for (size_t v = 0; v < vox_count_.size(); ++v)
      for (size_t j = 0; j < vox_count_[v]; ++j) {
        if (xyz && n < capacity) {
          const Float4 & p = buckets_[v * kBucketStride + j];
          xyz[3 * n + 0] = p.x;
          xyz[3 * n + 1] = p.y;
          xyz[3 * n + 2] = p.z;
        }
        ++n;
      }
    return n;
  }

private:
  int find_block(int bx, int by, int bz) const
  {
    uint32_t h = block_hash(bx, by, bz) & table_mask_;
    for (;;) {
      const Int4 & s = table_[h];
      if (s.w < 0) return -1;
      if (s.x == bx && s.y == by && s.z == bz) return s.w;
      h = (h + 1) & table_mask_;
    }
  }
  void table_put(int bx, int by, int bz, int id)
  {
    uint32_t h = block_hash(bx, by, bz) & table_mask_;
    while (table_[h].w >= 0) h = (h + 1) & table_mask_;
    table_[h] = Int4{bx, by, bz, id};
  }
  void rehash_blocks(size_t capacity)
  {
    table_.assign(capacity, Int4{0, 0, 0, -1});
    table_mask_ = static_cast<uint32_t>(capacity - 1);
    for (size_t b = 0; b < n_blocks_; ++b)
      table_put(block_coord_[3 * b], block_coord_[3 * b + 1], block_coord_[3 * b + 2], static_cast<int>(b));
    structure_changed_ = true;
  }
  int find_or_create_block(int bx, int by, int bz)
  {
    int blk = find_block(bx, by, bz);
    if (blk < 0) {
      if ((n_blocks_ + 1) * 2 > table_.size()) rehash_blocks(table_.size() * 2);
      blk = static_cast<int>(n_blocks_++);
      block_coord_.insert(block_coord_.end(), {bx, by, bz});
      cells_.resize(n_blocks_ * kCellsPerBlock, kEmptyCell);
      table_put(bx, by, bz, blk);
      structure_changed_ = true;
    }
    return blk;
  }
  void write_voxel_word(uint32_t vid, uint32_t word)
  {
    const uint32_t * pos = &vox_cells_[static_cast<size_t>(vid) * 8];
    for (int i = 0; i < vox_ncells_[vid]; ++i) cells_[pos[i]] = word;
  }
  // Registers voxel `vid` at (cx,cy,cz) in its home table and in the halo of every adjacent block it touches
  // (creating those tables), remembers the positions, writes `word` there.
  void place_voxel(uint32_t vid, int cx, int cy, int cz, uint32_t word)
  {
    const int m = kBlockDim - 1;
    int bs[3][2], nb[3];
    const int c[3] = {cx, cy, cz};
    for (int a = 0; a < 3; ++a) {
      bs[a][0] = c[a] >> kBlockLog2;
      nb[a] = 1;
      if ((c[a] & m) == 0) bs[a][nb[a]++] = (c[a] >> kBlockLog2) - 1;       // local coordinate 4 in the block below
      else if ((c[a] & m) == m) bs[a][nb[a]++] = (c[a] >> kBlockLog2) + 1;  // local coordinate -1 in the block above
    }
    int n = 0;
    for (int ix = 0; ix < nb[0]; ++ix)
      for (int iy = 0; iy < nb[1]; ++iy)
        for (int iz = 0; iz < nb[2]; ++iz) {
          const int bx = bs[0][ix], by = bs[1][iy], bz = bs[2][iz];
          const int blk = find_or_create_block(bx, by, bz);
          const uint32_t pos = static_cast<uint32_t>(static_cast<size_t>(blk) * kCellsPerBlock +
                                                     halo_index(cx - bx * kBlockDim, cy - by * kBlockDim, cz - bz * kBlockDim));
          vox_cells_[static_cast<size_t>(vid) * 8 + n++] = pos;  // the home block comes first (ix = iy = iz = 0)
          cells_[pos] = word;
        }
    vox_ncells_[vid] = static_cast<uint8_t>(n);
  }
  // stage 0: locate the cell word of the point's voxel and prefetch it; stage 1: read it (it should have arrived)
  // and prefetch the voxel's counters and bucket.  No state is changed.
  void prefetch_for(const float * p, int stage) const
  {
    const int cx = fast_floor(static_cast<double>(p[0]) * inv_leaf_), cy = fast_floor(static_cast<double>(p[1]) * inv_leaf_),
              cz = fast_floor(static_cast<double>(p[2]) * inv_leaf_);
    const int blk = find_block(cx >> kBlockLog2, cy >> kBlockLog2, cz >> kBlockLog2);
    if (blk < 0) return;
    const int m = kBlockDim - 1;
    const uint32_t * w = &cells_[static_cast<size_t>(blk) * kCellsPerBlock + halo_index(cx & m, cy & m, cz & m)];
    if (stage == 0) {
      __builtin_prefetch(w, 0, 1);
      return;
    }
    const uint32_t e = *w;
    if (e == kEmptyCell) return;
    const size_t vid = e >> 5;
    __builtin_prefetch(&vox_count_[vid], 1, 1);
    __builtin_prefetch(&vox_lru_[vid], 1, 1);
    __builtin_prefetch(&dirty_flag_[vid], 1, 1);
    const Float4 * b = &buckets_[vid * kBucketStride];
    __builtin_prefetch(b, 1, 1);
    __builtin_prefetch(b + 4, 1, 1);
    __builtin_prefetch(b + 8, 1, 1);
    __builtin_prefetch(&qbuckets_[vid * kBucketStride], 1, 1);
    __builtin_prefetch(&vox_cells_[vid * 8], 0, 1);
  }
  uint32_t find_or_create_voxel(int cx, int cy, int cz)
  {
    const int bx = cx >> kBlockLog2, by = cy >> kBlockLog2, bz = cz >> kBlockLog2;
    int blk;
    if (last_block_ >= 0 && bx == last_b_[0] && by == last_b_[1] && bz == last_b_[2]) {
      blk = last_block_;
    } else {
      blk = find_block(bx, by, bz);
      last_block_ = blk;
      last_b_[0] = bx;
      last_b_[1] = by;
      last_b_[2] = bz;
    }
    const int m = kBlockDim - 1;
    if (blk >= 0) {
      const uint32_t e = cells_[static_cast<size_t>(blk) * kCellsPerBlock + halo_index(cx & m, cy & m, cz & m)];
      if (e != kEmptyCell) return e >> 5;
    }
    const uint32_t vid = static_cast<uint32_t>(vox_count_.size());
    vox_count_.push_back(0);
    vox_lru_.push_back(lru_counter_);
    vox_cells_.resize(static_cast<size_t>(vid + 1) * 8, 0u);
    vox_ncells_.push_back(0);
    vox_coord_.insert(vox_coord_.end(), {cx, cy, cz});
    buckets_.resize(static_cast<size_t>(vid + 1) * kBucketStride, Float4{0, 0, 0, 0});
    qbuckets_.resize(static_cast<size_t>(vid + 1) * kBucketStride, 0u);
    dirty_flag_.push_back(0);
    place_voxel(vid, cx, cy, cz, vid << 5);
    last_block_ = -1;  // place_voxel may have created / rehashed blocks
    mark_dirty(vid);
    return vid;
  }
  void mark_dirty(uint32_t vid)
  {
    if (!dirty_flag_[vid]) {
      dirty_flag_[vid] = 1;
      dirty_.push_back(vid);
    }
  }
  // Remove voxels with lru + horizon < counter, keep creation order, renumber, rebuild the blocks.
  void purge_lru()
  {
    const uint64_t horizon = static_cast<uint64_t>(cfg_.lru_horizon);
    size_t keep = 0;
    bool any = false;
    for (size_t v = 0; v < vox_count_.size(); ++v)
      if (vox_lru_[v] + horizon < lru_counter_) {
        any = true;
        break;
      }
    if (!any) return;
    std::vector<int32_t> coord;
    std::vector<uint8_t> count;
    std::vector<uint64_t> lru;
    BigVec<Float4> buckets;
    BigVec<uint32_t> qb;
    for (size_t v = 0; v < vox_count_.size(); ++v) {
      if (vox_lru_[v] + horizon < lru_counter_) continue;
      coord.insert(coord.end(), {vox_coord_[3 * v], vox_coord_[3 * v + 1], vox_coord_[3 * v + 2]});
      count.push_back(vox_count_[v]);
      lru.push_back(vox_lru_[v]);
      buckets.insert(
        buckets.end(), buckets_.begin() + v * kBucketStride, buckets_.begin() + (v + 1) * kBucketStride);
      qb.insert(qb.end(), qbuckets_.begin() + v * kBucketStride, qbuckets_.begin() + (v + 1) * kBucketStride);
      ++keep;
    }
    vox_coord_.swap(coord);
    vox_count_.swap(count);
    vox_lru_.swap(lru);
    buckets_.swap(buckets);
    qbuckets_.swap(qb);
    vox_cells_.assign(keep * 8, 0u);
    vox_ncells_.assign(keep, 0);
    dirty_flag_.assign(keep, 0);
    dirty_.clear();
    n_blocks_ = 0;
    block_coord_.clear();
    cells_.clear();
    last_block_ = -1;
    table_.assign(1024, Int4{0, 0, 0, -1});
    table_mask_ = 1023u;
    n_points_ = 0;
    for (size_t v = 0; v < keep; ++v) {
      place_voxel(static_cast<uint32_t>(v), vox_coord_[3 * v], vox_coord_[3 * v + 1], vox_coord_[3 * v + 2],
                  (static_cast<uint32_t>(v) << 5) | vox_count_[v]);
      n_points_ += vox_count_[v];
    }
    structure_changed_ = true;
    full_rebuild_ = true;
  }

  template <typename V>
  static void copy_with_headroom(V & dst, const V & src)
  {
    V v;
    v.reserve(src.size() + src.size() / 8 + 4096);
    v.assign(src.begin(), src.end());
    dst.swap(v);
  }
  // the big arrays: size without touching, then copy (and first-touch) with several threads
  template <typename T>
  static void copy_with_headroom(BigVec<T> & dst, const BigVec<T> & src)
  {
    BigVec<T> v;
    v.reserve(src.size() + src.size() / 8 + 4096);
    v.resize(src.size());
    const size_t bytes = src.size() * sizeof(T);
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = bytes < (size_t(8) << 20) ? 1 : (hw >= 64 ? 16 : (hw >= 16 ? 8 : (hw >= 4 ? 4 : 1)));
    if (nt == 1) {
      if (bytes) std::memcpy(static_cast<void *>(v.data()), src.data(), bytes);
    } else {
      std::vector<std::thread> th;
      const size_t chunk = ((bytes / nt) + 4095) & ~size_t(4095);
      for (size_t k = 0; k < nt; ++k) {
        const size_t b0 = k * chunk, b1 = b0 + chunk < bytes ? b0 + chunk : bytes;
        if (b0 >= b1) break;
        th.emplace_back([&v, &src, b0, b1] {
          std::memcpy(reinterpret_cast<char *>(v.data()) + b0, reinterpret_cast<const char *>(src.data()) + b0, b1 - b0);
        });
      }
      for (auto & t : th) t.join();
    }
    dst.swap(v);
  }
  void swap_all(HostVoxelMap & o)
  {
    using std::swap;
    swap(cfg_, o.cfg_);
    swap(inv_leaf_, o.inv_leaf_);
    swap(min_sq_, o.min_sq_);
    swap(lru_counter_, o.lru_counter_);
    swap(n_points_, o.n_points_);
    vox_coord_.swap(o.vox_coord_);
    vox_count_.swap(o.vox_count_);
    vox_lru_.swap(o.vox_lru_);
    vox_cells_.swap(o.vox_cells_);
    vox_ncells_.swap(o.vox_ncells_);
    buckets_.swap(o.buckets_);
    qbuckets_.swap(o.qbuckets_);
    swap(n_blocks_, o.n_blocks_);
    block_coord_.swap(o.block_coord_);
    cells_.swap(o.cells_);
    table_.swap(o.table_);
    swap(table_mask_, o.table_mask_);
    last_block_ = o.last_block_ = -1;
    dirty_flag_.swap(o.dirty_flag_);
    dirty_.swap(o.dirty_);
    swap(structure_changed_, o.structure_changed_);
    swap(full_rebuild_, o.full_rebuild_);
  }
  void copy_from(const HostVoxelMap & o)
  {
    cfg_ = o.cfg_;
    inv_leaf_ = o.inv_leaf_;
    min_sq_ = o.min_sq_;
    lru_counter_ = o.lru_counter_;
    n_points_ = o.n_points_;
    copy_with_headroom(vox_coord_, o.vox_coord_);
    copy_with_headroom(vox_count_, o.vox_count_);
    copy_with_headroom(vox_lru_, o.vox_lru_);
    copy_with_headroom(vox_cells_, o.vox_cells_);
    copy_with_headroom(vox_ncells_, o.vox_ncells_);
    copy_with_headroom(buckets_, o.buckets_);
    copy_with_headroom(qbuckets_, o.qbuckets_);
    n_blocks_ = o.n_blocks_;
    copy_with_headroom(block_coord_, o.block_coord_);
    copy_with_headroom(cells_, o.cells_);
    table_ = o.table_;
    table_mask_ = o.table_mask_;
    last_block_ = -1;
    copy_with_headroom(dirty_flag_, o.dirty_flag_);
    dirty_ = o.dirty_;
    structure_changed_ = o.structure_changed_;
    full_rebuild_ = o.full_rebuild_;
  }

public:
  bool take_full_rebuild()
  {
    const bool r = full_rebuild_;
    full_rebuild_ = false;
    return r;
  }

private:
  mh_map_config cfg_;
  double inv_leaf_, min_sq_;
  uint64_t lru_counter_ = 0;
  size_t n_points_ = 0;
  // per voxel (creation order == iVox flat_voxels order)
  std::vector<int32_t> vox_coord_;
  std::vector<uint8_t> vox_count_;
  std::vector<uint64_t> vox_lru_;
  BigVec<uint32_t> vox_cells_;   // 8 slots per voxel: indices into cells_ that hold its word (home first)
  std::vector<uint8_t> vox_ncells_;   // how many of the 8 are used
  BigVec<Float4> buckets_;
  BigVec<uint32_t> qbuckets_;
  // blocks
  size_t n_blocks_ = 0;
  std::vector<int32_t> block_coord_;
  BigVec<uint32_t> cells_;
  std::vector<Int4> table_;
  uint32_t table_mask_ = 0;
  int last_block_ = -1;
  int last_b_[3] = {0, 0, 0};
  // device mirror bookkeeping
  std::vector<uint8_t> dirty_flag_;
  std::vector<uint32_t> dirty_;
  bool structure_changed_ = true;
  bool full_rebuild_ = false;
};

}  // namespace mh
