// Device-resident scan front end: launcher declarations shared by scan_kernels.hip and the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mimosa_hip.h"

namespace mh
{
// Small device-side header the front-end kernels fill; copied to the host once per stage.
struct ScanCounters
{
  uint32_t n_full;        // points that passed the input filter
  uint32_t n_geometric;   // of those, the geometric subset
  uint32_t n_unique_ns;   // distinct timestamps among the kept points
  uint32_t last_point_ns; // max t_ns among the kept points (manager.cpp:310)
  uint32_t n_voxels;      // source voxel grid filter: occupied voxels
  uint32_t n_downsampled; // points the filter kept
  uint32_t bad_coord;     // a voxel coordinate did not fit the 21-bit key field
  uint32_t pad;
};

// temp-storage sizes of the rocPRIM primitives used for n elements (max over all of them)
size_t scan_temp_bytes(size_t n);

// prepareInput (lidar/manager.cpp:244-336): filter + order-preserving compaction.
//   flags / pos: 2 x n uint32 scratch each (full, geometric)
hipError_t launch_input_filter(const mh_ouster_point * raw, uint32_t n, const mh_input_config & cfg, uint32_t * flag_full,
                               uint32_t * flag_geo, uint32_t * pos_full, uint32_t * pos_geo, mh_point32 * points_full,
                               uint32_t * geo_idx, ScanCounters * counters, void * temp, size_t temp_bytes,
                               hipStream_t stream);
// sorted distinct timestamps of points_full[0..n_full) (manager.cpp:340-368).  keys_a / keys_b: n uint32 scratch each.
hipError_t launch_unique_ns(const mh_point32 * points_full, const ScanCounters * counters, uint32_t n_cap, uint32_t * keys_a,
                            uint32_t * keys_b, uint32_t * flags, uint32_t * pos, uint32_t * unique_ns,
                            ScanCounters * counters_out, void * temp, size_t temp_bytes, hipStream_t stream);
// Geometric::preprocess (geometric.cpp:154-161): body[j] = R * points_full[geo_idx[j]] + t in f32
hipError_t launch_gather_transform(const mh_point32 * points_full, const uint32_t * geo_idx, uint32_t n_geo,
                                   const float * Rt12, mh_point32 * body, hipStream_t stream);
// Geometric::downsample (geometric.cpp:55-126) + FlatContainerMinimal::add (lidar/utils.hpp:260-278).
// Scratch: keys_a/keys_b n uint64, idx_a/idx_b n uint32, flags/pos n uint32, seg_start n+1 uint32, first_idx n uint32.
hipError_t launch_downsample(const mh_point32 * body, uint32_t n, double leaf, uint32_t max_pts, double min_dist,
                             uint64_t * keys_a, uint64_t * keys_b, uint32_t * idx_a, uint32_t * idx_b, uint32_t * flags,
                             uint32_t * pos, uint32_t * seg_start, uint32_t * first_idx, uint32_t * kept_idx,
                             mh_point32 * out, ScanCounters * counters, void * temp, size_t temp_bytes, hipStream_t stream);

}  // namespace mh
