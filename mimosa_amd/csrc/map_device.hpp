// Device-resident incremental voxel map: state and launcher declarations shared by map_kernels.hip and the C ABI.
//
// Replaces gtsam_points::iVox as used through IncrementalVoxelMapPCL (reference: include/mimosa/lidar/
// incremental_voxel_map.hpp:22-54, src/lidar/incremental_voxel_map.cpp:14-62; configuration src/lidar/geometric.cpp:23-28).
// The layout of voxel_map.hpp (buckets, packed coarse buckets, halo'd 4x4x4 block tables, block hash) is unchanged;
// what changed in round 2 is WHO maintains it: insertion, voxel / block creation, the LRU purge, get_cloud and the
// copy all run on the device, so a keyframe update moves the keyframe cloud to the device and nothing else.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voxel_map.hpp"

namespace mh
{
constexpr int kVoxCoordBits = 21;                      // voxel / block coordinates must fit 21 bits (+-2^20 voxels)
constexpr int kVoxCoordBias = 1 << (kVoxCoordBits - 1);
constexpr uint64_t kEmptyKey = ~0ull;

#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t pack_coord_key(int x, int y, int z)
{
  const uint64_t m = (1ull << kVoxCoordBits) - 1;
  return ((static_cast<uint64_t>(x + kVoxCoordBias) & m) << (2 * kVoxCoordBits)) | ((static_cast<uint64_t>(y + kVoxCoordBias) & m) << kVoxCoordBits) |
         (static_cast<uint64_t>(z + kVoxCoordBias) & m);
}

// Counters the map kernels publish into mapped pinned host memory (read by the host after a stream synchronisation).
struct MapState
{
  uint32_t n_voxels;      // voxels in the map (creation order = iVox's flat_voxels order)
  uint32_t n_blocks;      // 4x4x4-voxel blocks with a table
  unsigned long long n_points;
  uint32_t n_segments;    // this insert: distinct voxels touched
  uint32_t n_new_voxels;  // this insert: voxels to create
  uint32_t bad_coord;     // a coordinate did not fit kVoxCoordBits
  uint32_t n_keep;        // LRU purge: voxels that survive
  uint32_t cloud_points;  // get_cloud: total points
  uint32_t pad[3];
};

// Everything the insert / purge kernels need to reach the map arrays.
struct MapArrays
{
  int4 * table;           // {key lo, key hi, block id, -}: 63-bit packed block coordinate, all-ones = empty
  uint32_t table_mask;
  uint32_t * cells;       // n_blocks x 216: voxel_id << 5 | count, ~0u empty
  float4 * buckets;       // n_voxels x 20
  uint32_t * qbuckets;    // n_voxels x 20, 3 x 10-bit voxel-relative
  int4 * vox;             // n_voxels: {cx, cy, cz, count}
  unsigned long long * lru;  // n_voxels: lru_counter at the last insert that touched the voxel
  MapState * state;       // device-visible (mapped pinned) counters
};

struct InsertScratch
{
  float4 * pts;           // n: the batch as float4 (after the optional f32 transform)
  void * group;           // voxel_group.hpp scratch of the batch (map_group_bytes(n)): hash, first-seen segments, index lists
  uint32_t * seg_vid;     // n: voxel id of segment s (existing, or n_voxels + creation rank)
  uint32_t * seg_added;   // n: points the insert added to segment s
  uint32_t * blk_new;     // (n + 255) / 256: per-block counts of segments that open a new voxel
};

size_t map_group_bytes(size_t n);
size_t map_temp_bytes(size_t n);
// phase A: voxel key per point -> first-seen grouping (voxel_group.hpp, no sort) -> which voxels exist / creation ranks
// of the new ones (first-seen order = the reference's creation order).  Publishes n_segments, n_new_voxels, bad_coord.  src: n points `stride_floats` apart (device memory); Rt12 != null applies the f32 rigid
// transform p <- R p + t first (Geometric::updateMap's world transform, geometric.cpp:483-490).
hipError_t launch_map_insert_prepare(const MapArrays & m, const float * src, uint32_t n, uint32_t stride_floats, const float * Rt12,
                                     double inv_leaf, const InsertScratch & s, hipStream_t stream);
// phase B: create the new voxels (coordinates, lru) and claim their blocks in the hash table; publishes n_blocks
hipError_t launch_map_create_voxels(const MapArrays & m, uint32_t n, uint32_t n_voxels_before, unsigned long long lru_counter,
                                    const InsertScratch & s, hipStream_t stream);
// phase C: FlatContainer::add per touched voxel, one wave each; cell words, counts, lru; publishes n_voxels, n_points
hipError_t launch_map_insert_points(const MapArrays & m, uint32_t n, uint32_t n_voxels_after, uint32_t max_pts, double min_sq,
                                    double inv_leaf, unsigned long long lru_counter, const InsertScratch & s, hipStream_t stream);
// hash table of a new capacity from the block coordinates stored in the old one
hipError_t launch_map_rehash(const int4 * old_table, uint32_t old_cap, int4 * new_table, uint32_t new_cap, hipStream_t stream);
// LRU purge (iVox: voxels with lru + horizon < counter are erased, the rest keep their order): flags + count,
// then compaction into fresh arrays and a rebuild of the block structure.
hipError_t launch_map_purge_flags(const MapArrays & m, uint32_t n_voxels, unsigned long long horizon, unsigned long long lru_counter,
                                  uint32_t * keep, uint32_t * pos, void * temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_map_purge_compact(const MapArrays & src, const MapArrays & dst, uint32_t n_voxels, const uint32_t * keep,
                                    const uint32_t * pos, hipStream_t stream);
// (re)build table + cells from the voxel coordinates: claim blocks, then write every voxel's word
hipError_t launch_map_claim_blocks(const MapArrays & m, uint32_t v0, uint32_t v1, hipStream_t stream);
hipError_t launch_map_write_words(const MapArrays & m, uint32_t v0, uint32_t v1, hipStream_t stream);
// voxel_data(): all points in voxel order.  offsets: n_voxels + 1 scratch; out: packed xyz (3 floats per point)
hipError_t launch_map_cloud(const MapArrays & m, uint32_t n_voxels, uint32_t * counts, uint32_t * offsets, float * out, size_t out_capacity_points,
                            void * temp, size_t temp_bytes, hipStream_t stream);

}  // namespace mh
