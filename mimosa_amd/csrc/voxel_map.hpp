// Layout of the device-resident incremental voxel map (the iVox counterpart): constants and the helpers host and
// device share.  The map itself is built and maintained by map_kernels.hip (state: map_device.hpp).
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
//                                          z fastest); entry = voxel_id << 5 | count (count >= 1), 0 = empty
//   table   : int4[capacity]               open-addressing hash of BLOCK coords -> block id: {key lo, key hi, id, -},
//                                          key = 3 x 21-bit packed block coordinate, all-ones = empty
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

#include "../../include/mimosa_hip.h"

namespace mh
{
constexpr int kBlockLog2 = 2;                       // 4x4x4 voxels per block
constexpr int kBlockDim = 1 << kBlockLog2;
constexpr int kHaloDim = kBlockDim + 2;                // block + one-voxel halo
constexpr int kCellsPerBlock = kHaloDim * kHaloDim * kHaloDim;  // 216 words per block table
constexpr int kBucketStride = 20;                   // FlatContainer max_num_points_in_cell
constexpr uint32_t kEmptyCell = 0u;  // a stored word is voxel_id << 5 | count with count >= 1: never 0.  Read as (voxel 0, count 0) by K3: no select
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

}  // namespace mh
