// Sequence replay with the map SHARDED over the ranks of a communicator (BASELINE configs[4] "at 8 GPU", SURVEY.md §8(e) + f-4):
// the loop of replay.hpp (the reference's per-scan order, src/mimosa_rosbag.cpp:200-223 -> lidar::Manager::callback,
// src/lidar/manager.cpp:45-147) with Geometric's three jobs done by the sharded classes of sharded.hpp:
//   map          ShardedVoxelMap: this rank's blocks + their one-voxel halo (mh_map_insert_shard)
//   getFactors   ShardedICPFactor on this rank's share of the scan's down-sampled cloud; the smoother's re-linearization of
//                every live factor is ONE protocol round per iteration (ShardedICPFactor::linearizeBatch ->
//                mh_shard_icp_linearize_batch: one ncclAllToAll + ncclAllReduce(s) over xGMI)
//   updateMap    at a keyframe every rank forks its shard (copy-then-insert, geometric.cpp:494-495) and inserts its part of
//                the scan's world cloud
// One process (or, in the tests, one host thread) per rank; every rank is handed the same scans, runs the same front end and
// the same smoother on Hessians that the all-reduce has made identical, so every rank holds the same trajectory — the
// unsharded replay's, up to the order in which the shards' rows are summed.
#pragma once

#include "replay.hpp"
#include "sharded.hpp"

namespace mimosa_hip
{
namespace replay
{
class ShardedGeometric
{
public:
  using Factor = lidar::ShardedICPFactor;
  ShardedGeometric(const lidar::ShardCommunicator::Ptr & comm, const Config & cfg, size_t lru_horizon, int block_log2 = 3, bool force_collectives = false)
  : comm_(comm), force_(force_collectives)
  {
    lidar::GeometricConfig g;
    g.lru_horizon = lru_horizon;
    g.neighbor_voxel_mode = cfg.neighbor_voxel_mode;
    g.scan_to_map = cfg.reg;
    shard_ = std::make_shared<lidar::ShardedVoxelMap>(comm, g, block_log2);
  }
  void seed(const float * xyz, size_t n) { shard_->insert(xyz, n); }
  // this rank's share of sm_Be_cloud_ds_ (any split does: the first linearize routes every point to the owner of its voxel),
  // taken where the front end left it — on the device
  Factor::Ptr makeFactor(const Key Xk, lidar::ScanFrontEnd & scan, const lidar::RegistrationConfig & reg)
  {
    const mh_point32 * d_ds = nullptr;
    size_t n = 0;
    comm_->context()->check(mh_scan_device_points(scan.underlying(), 2, &d_ds, &n), "mh_scan_device_points");
    const size_t w = static_cast<size_t>(comm_->world()), r = static_cast<size_t>(comm_->rank());
    const size_t lo = n * r / w, hi = n * (r + 1) / w;
    return std::make_shared<Factor>(Xk, shard_, d_ds + lo, hi - lo, reg, force_);
  }
  // Geometric::updateMap's insert (geometric.cpp:483-495): copy-then-insert of this rank's shard, the scan's Be_cloud_ transformed,
  // filtered and inserted on the device
  void keyframe(lidar::ScanFrontEnd & scan, const Pose3 & T_W_Be)
  {
    shard_ = shard_->fork();
    shard_->insertBodyCloud(scan.underlying(), T_W_Be);
  }

private:
  lidar::ShardCommunicator::Ptr comm_;
  lidar::ShardedVoxelMap::Ptr shard_;
  bool force_;
};

class ShardedFixedLagReplay : public FixedLagReplayT<ShardedGeometric>
{
public:
  ShardedFixedLagReplay(const lidar::ShardCommunicator::Ptr & comm, const Config & cfg, size_t lru_horizon = 1000, int block_log2 = 3, bool force_collectives = false)
  : FixedLagReplayT<ShardedGeometric>(comm->context(), cfg, std::unique_ptr<ShardedGeometric>(new ShardedGeometric(comm, cfg, lru_horizon, block_log2, force_collectives)))
  {
  }
};

}  // namespace replay
}  // namespace mimosa_hip
