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
  uint32_t n_unique_ns;   // distinct timestamps among the kept points (without the one has_max_ns stands for)
  uint32_t last_point_ns; // max t_ns among the kept points (manager.cpp:310)
  uint32_t n_voxels;      // source voxel grid filter: occupied voxels
  uint32_t n_downsampled; // points the filter kept
  uint32_t bad_coord;     // a voxel coordinate did not fit the 21-bit key field
  uint32_t has_max_ns;    // the timestamp 0xFFFFFFFF (the hash set's empty marker) occurred: listed last in unique_ns
};

struct Rt12  // row-major R, then t; passed to kernels by value
{
  float v[12];
};

struct PrepareLayout
{
  uint32_t n_blocks, ns_cap;
  size_t words;  // uint32 words of scratch launch_prepare_input needs
};
PrepareLayout prepare_layout(size_t n);
struct VoxelLayout
{
  uint32_t n_blocks, cap;
  size_t clear_bytes, bytes;  // bytes of scratch launch_preprocess needs (the first clear_bytes are memset per call)
};
VoxelLayout voxel_layout(size_t n);

// prepareInput (lidar/manager.cpp:244-368): filter + order-preserving compaction into points_full / geo_idx, the
// ascending distinct timestamps into unique_ns (capacity n + 1).  3 kernels; zeroes the counters first.
hipError_t launch_prepare_input(const mh_ouster_point * raw, uint32_t n, const mh_input_config & cfg, uint32_t * scratch,
                                mh_point32 * points_full, uint32_t * geo_idx, uint32_t * unique_ns, ScanCounters * counters,
                                hipStream_t stream, bool canonical = false, bool ring_filter = true);
// Any sensor's records (device memory, n * layout.stride bytes) -> canonical PointOuster-shaped records in the order the
// reference's loop sees them: transposed (lidar/manager.cpp:177-203), then organised by ring (:205-241).  canonical
// `reflectivity` bit 0 = rejected by the Livox tag test.  tmp: n records (organize only); hist: 128 x ceil(n / 256) words;
// *bad_ring != 0 afterwards: a ring number >= 128 occurred.
hipError_t launch_decode_points(const void * raw, uint32_t n, const mh_point_layout & layout, uint32_t width, uint32_t height, bool transpose,
                                bool organize_by_ring, double header_ts, mh_ouster_point * canon, mh_ouster_point * tmp, uint32_t * hist,
                                uint32_t * bad_ring, hipStream_t stream);
// Geometric::preprocess (geometric.cpp:154-161): body[j] = R * points_full[geo_idx[j]] + t in f32, then
// Geometric::downsample (geometric.cpp:55-126) + FlatContainerMinimal::add (lidar/utils.hpp:260-278): kept_idx
// (indices into body) and out = body[kept_idx], in the reference's output order.  1 memset + 7 kernels.
hipError_t launch_preprocess(const mh_point32 * points_full, const uint32_t * geo_idx, uint32_t n, const Rt12 & body_from_lidar,
                             double leaf, uint32_t max_pts, double min_dist, void * scratch, mh_point32 * body,
                             uint32_t * kept_idx, mh_point32 * out, ScanCounters * counters, hipStream_t stream);

}  // namespace mh
