// C++ host mirror of the reference's LiDAR geometric classes over the C ABI (include/mimosa_hip.h):
//
//   IncrementalVoxelMapPCL   include/mimosa/lidar/incremental_voxel_map.hpp:22-54
//   ICPFactor                include/mimosa/lidar/geometric_factor.hpp:25-563
//   Geometric                include/mimosa/lidar/geometric.hpp:36-89, src/lidar/geometric.cpp
//   Manager::deskewPoints    src/lidar/manager.cpp:385-512 (the per-point part, :496-509)
//
// Same constructor / method / getter names and argument meaning, same call order, same error
// behaviour (std::runtime_error for fatal conditions, per-point failures are statuses).  ROS
// publishers, loggers and config_utilities are replaced by plain structs.  Header-only; link with
// libmimosa_hip.so.  All arithmetic happens behind the C ABI on the GPU — nothing here computes a
// residual, and there is no CPU fallback.
#pragma once

#include <algorithm>
#include <cstring>
#include <limits>
#include <unordered_map>

#include "../../../include/mimosa_hip.h"
#include "types.hpp"

namespace mimosa_hip
{
namespace lidar
{
using Point = mh_point32;                 // include/mimosa/lidar/point.hpp:18-39
using PointCloud = std::vector<Point>;    // pcl::PointCloud<Point>
using RegistrationConfig = mh_reg_config; // include/mimosa/lidar/geometric_config.hpp:17-33

inline RegistrationConfig defaultRegistrationConfig()
{
  // struct defaults of geometric_config.hpp:17-33
  RegistrationConfig c{};
  c.source_voxel_grid_filter_leaf_size = 0.5f;
  c.source_voxel_grid_min_dist_in_voxel = 0.1f;
  c.target_ivox_map_leaf_size = 0.5f;
  c.target_ivox_map_min_dist_in_voxel = 0.1f;
  c.num_corres_points = 5;
  c.max_corres_distance = 2.24f;
  c.plane_validity_distance = 0.04f;
  c.lidar_point_noise_std_dev = 0.02f;
  c.use_huber = 1;
  c.huber_threshold = 1.345f;
  c.reg_4_dof = 0;
  c.project_on_degneneracy = 1;
  c.degen_thresh_rot = 10;
  c.degen_thresh_trans = 15;
  return c;
}

// GeometricConfig, geometric_config.hpp:37-55 (ROS frame names / log settings dropped)
struct GeometricConfig
{
  bool enabled = true;
  Pose3 T_B_L = Pose3();
  int point_skip_divisor = 1;
  int ring_skip_divisor = 1;
  float map_keyframe_trans_thresh = 0.1f;
  float map_keyframe_rot_thresh_deg = 10;
  size_t initial_clouds_to_force_map_update = 10;
  size_t lru_horizon = 100;
  size_t neighbor_voxel_mode = 7;
  RegistrationConfig scan_to_map = defaultRegistrationConfig();
};

// One context per process/device; thrown errors carry mh_last_error().
class Context
{
public:
  explicit Context(int device = 0) : device_(device)
  {
    if (mh_init(device, &ctx_) != MH_OK) throw std::runtime_error(std::string("mh_init: ") + mh_last_error(nullptr));
  }
  int device() const { return device_; }
  ~Context() { mh_shutdown(ctx_); }
  Context(const Context &) = delete;
  Context & operator=(const Context &) = delete;
  mh_ctx * get() const { return ctx_; }
  void check(int rc, const char * what) const
  {
    if (rc != MH_OK) throw std::runtime_error(std::string(what) + ": " + mh_last_error(ctx_));
  }

private:
  int device_ = 0;
  mh_ctx * ctx_ = nullptr;
};

// ---------------------------------------------------------------------------------------------
class IncrementalVoxelMapPCL
{
public:
  using Ptr = std::shared_ptr<IncrementalVoxelMapPCL>;

  IncrementalVoxelMapPCL(const std::shared_ptr<Context> & ctx, const float leaf_size) : ctx_(ctx)
  {
    cfg_.leaf_size = leaf_size;  // iVox defaults, overridden by Geometric's ctor (geometric.cpp:23-28)
    cfg_.min_dist_in_cell = 0.1;
    cfg_.max_points_in_cell = 20;
    cfg_.neighbor_voxel_mode = 7;
    cfg_.lru_horizon = 100;
    cfg_.lru_clear_cycle = 10;
  }
  // Deep copy constructor (incremental_voxel_map.hpp:33-42)
  IncrementalVoxelMapPCL(const IncrementalVoxelMapPCL & other) : ctx_(other.ctx_), cfg_(other.cfg_)
  {
    if (other.map_) ctx_->check(mh_map_copy(other.map_, &map_), "mh_map_copy");
  }
  IncrementalVoxelMapPCL & operator=(const IncrementalVoxelMapPCL &) = delete;
  ~IncrementalVoxelMapPCL() { mh_map_release(map_); }

  // Successor for copy-then-insert (geometric.cpp:494-495): a deep copy, device to device (the map lives on the GPU;
  // 0.2 ms for a 5 M-point map).  THIS object stays valid and unchanged for the factors that hold it.
  std::shared_ptr<IncrementalVoxelMapPCL> fork()
  {
    ensure();
    std::shared_ptr<IncrementalVoxelMapPCL> next(new IncrementalVoxelMapPCL(ctx_, cfg_, nullptr));
    ctx_->check(mh_map_copy(map_, &next->map_), "mh_map_copy");
    return next;
  }

  // iVox setters used at geometric.cpp:25-28 — must precede the first insert
  void set_lru_horizon(size_t h) { require_empty(); cfg_.lru_horizon = static_cast<int64_t>(h); }
  void set_neighbor_voxel_mode(size_t m) { require_empty(); cfg_.neighbor_voxel_mode = static_cast<int32_t>(m); }
  void set_min_dist_in_cell(double d) { require_empty(); cfg_.min_dist_in_cell = d; }

  void insert(const PointCloud & cloud)  // incremental_voxel_map.cpp:19-24
  {
    ensure();
    if (!cloud.empty())
      ctx_->check(mh_map_insert(map_, &cloud[0].x, cloud.size(), sizeof(Point) / sizeof(float)), "mh_map_insert");
  }
  void insert(const float * xyz, size_t n)
  {
    ensure();
    ctx_->check(mh_map_insert(map_, xyz, n, 3), "mh_map_insert");
  }
  // Geometric::updateMap's insert (geometric.cpp:483-495) of a device-resident scan's Be_cloud_: f32 world transform
  // + greedy insert on the GPU, nothing crosses PCIe.  (ScanFrontEnd is declared below.)
  void insertBodyCloud(mh_scan * scan, const Pose3 & T_W_Be)
  {
    ensure();
    float Rt[12];
    toFloat12(T_W_Be, Rt);
    ctx_->check(mh_map_insert_from_scan(map_, scan, Rt, Rt + 9), "mh_map_insert_from_scan");
  }
  // incremental_voxel_map.cpp:26-32: true iff k neighbours were found; coordinates instead of ids
  bool knn_search(const V3D & point, const size_t k, std::vector<V3D> & neighbours, std::vector<double> & sq_dists)
  {
    ensure();
    neighbours.assign(k, V3D());
    sq_dists.assign(k, 0.0);
    int32_t found = 0;
    const A3 q = toArray(point);
    std::vector<double> nb(3 * k);
    ctx_->check(mh_map_knn(map_, q.data(), 1, static_cast<int>(k), nb.data(), sq_dists.data(), &found), "mh_map_knn");
    for (size_t i = 0; i < k; ++i) neighbours[i] = vector3(&nb[3 * i]);
    return static_cast<size_t>(found) == k;
  }
  PointCloud getCloud()  // incremental_voxel_map.cpp:34-38
  {
    ensure();
    size_t n = 0;
    ctx_->check(mh_map_get_cloud(map_, nullptr, 0, &n), "mh_map_get_cloud");
    std::vector<float> xyz(3 * n);
    ctx_->check(mh_map_get_cloud(map_, xyz.data(), n, &n), "mh_map_get_cloud");
    PointCloud out(n, Point{});
    for (size_t i = 0; i < n; ++i) {
      out[i].x = xyz[3 * i];
      out[i].y = xyz[3 * i + 1];
      out[i].z = xyz[3 * i + 2];
    }
    return out;
  }
  mh_map * underlying()
  {
    ensure();
    return map_;
  }
  const std::shared_ptr<Context> & context() const { return ctx_; }

private:
  IncrementalVoxelMapPCL(const std::shared_ptr<Context> & ctx, const mh_map_config & cfg, std::nullptr_t) : ctx_(ctx), cfg_(cfg) {}
  void ensure()
  {
    if (!map_) ctx_->check(mh_map_create(ctx_->get(), &cfg_, &map_), "mh_map_create");
  }
  void require_empty() const
  {
    if (map_) throw std::runtime_error("IncrementalVoxelMapPCL: settings must be applied before the first insert");
  }
  std::shared_ptr<Context> ctx_;
  mh_map_config cfg_{};
  mh_map * map_ = nullptr;
};

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Device-resident scan front end: the part of lidar::Manager between the PointCloud2 callback and the
// geometric factor (prepareInput lidar/manager.cpp:149-383, deskewPoints' per-point loop :496-509),
// with the cloud uploaded once and kept on the GPU.  Geometric::preprocess / getFactors below consume it.
using PointOuster = mh_ouster_point;  // include/mimosa/lidar/point.hpp:42-50
using ManagerInputConfig = mh_input_config;

// The reference's other sensor point types (include/mimosa/lidar/point.hpp:52-131): the records a driver publishes, with
// the members where PCL's EIGEN_ALIGN16 structs put them, and the layout that selects each type's branches of
// Manager::prepareInput<PointT> (tag filter, reflectivity-as-intensity, time decoding, whether the ring filter applies).
struct alignas(16) PointOusterOdyssey { float x, y, z, data_c; uint32_t t; uint16_t reflectivity, near_ir; };
struct alignas(16) PointOusterR8 { float x, y, z, data_c; float intensity; uint32_t t; uint16_t reflectivity; uint8_t ring; };
struct alignas(16) PointHesai { float x, y, z, data_c; float intensity; double timestamp; uint16_t ring; };
struct alignas(16) PointLivox { float x, y, z, data_c; float intensity; uint8_t tag, line; double timestamp; };
struct alignas(16) PointLivoxFromCustom2 { float x, y, z; uint32_t t; float intensity; uint8_t tag, line; };
struct alignas(16) PointVelodyne { float x, y, z, data_c; float intensity; uint16_t ring; float time; };
struct alignas(16) PointVelodyneAnybotics { float x, y, z, data_c; float intensity; float ring; float time; };
struct alignas(16) PointRslidar { float x, y, z, data_c; float intensity; uint16_t ring; double timestamp; };

namespace detail
{
inline mh_point_layout xyz(uint32_t stride, uint32_t off_intensity, uint32_t off_time, mh_time_kind tk)
{
  mh_point_layout L{};
  L.stride = stride;
  L.off_x = 0;
  L.off_y = 4;
  L.off_z = 8;
  L.off_intensity = off_intensity;
  L.off_time = off_time;
  L.time_kind = tk;
  return L;
}
inline mh_point_layout with_ring(mh_point_layout L, uint32_t off, mh_ring_kind kind, bool filter)
{
  L.off_ring = off;
  L.ring_kind = kind;
  L.ring_filter = filter ? 1 : 0;
  return L;
}
inline mh_point_layout with_tag(mh_point_layout L, uint32_t off)
{
  L.off_tag = off;
  L.has_tag = 1;
  return L;
}
}  // namespace detail

template <typename PointT> mh_point_layout layoutOf();
template <> inline mh_point_layout layoutOf<PointOuster>()
{
  return detail::with_ring(detail::xyz(sizeof(PointOuster), offsetof(PointOuster, intensity), offsetof(PointOuster, t), MH_TIME_U32_NS),
                           offsetof(PointOuster, ring), MH_RING_U16, true);
}
template <> inline mh_point_layout layoutOf<PointOusterOdyssey>()
{
  mh_point_layout L = detail::xyz(sizeof(PointOusterOdyssey), offsetof(PointOusterOdyssey, reflectivity), offsetof(PointOusterOdyssey, t), MH_TIME_U32_NS);
  L.intensity_is_u16 = 1;
  return L;
}
template <> inline mh_point_layout layoutOf<PointOusterR8>()
{
  return detail::with_ring(detail::xyz(sizeof(PointOusterR8), offsetof(PointOusterR8, intensity), offsetof(PointOusterR8, t), MH_TIME_U32_NS),
                           offsetof(PointOusterR8, ring), MH_RING_U8, true);
}
template <> inline mh_point_layout layoutOf<PointHesai>()
{
  return detail::with_ring(detail::xyz(sizeof(PointHesai), offsetof(PointHesai, intensity), offsetof(PointHesai, timestamp), MH_TIME_F64_S_ABS),
                           offsetof(PointHesai, ring), MH_RING_U16, true);
}
template <> inline mh_point_layout layoutOf<PointLivox>()
{
  return detail::with_tag(detail::xyz(sizeof(PointLivox), offsetof(PointLivox, intensity), offsetof(PointLivox, timestamp), MH_TIME_F64_NS_ABS),
                          offsetof(PointLivox, tag));
}
template <> inline mh_point_layout layoutOf<PointLivoxFromCustom2>()
{
  return detail::with_tag(detail::xyz(sizeof(PointLivoxFromCustom2), offsetof(PointLivoxFromCustom2, intensity), offsetof(PointLivoxFromCustom2, t),
                                      MH_TIME_U32_NS),
                          offsetof(PointLivoxFromCustom2, tag));
}
template <> inline mh_point_layout layoutOf<PointVelodyne>()
{
  return detail::with_ring(detail::xyz(sizeof(PointVelodyne), offsetof(PointVelodyne, intensity), offsetof(PointVelodyne, time), MH_TIME_F32_S),
                           offsetof(PointVelodyne, ring), MH_RING_U16, true);
}
template <> inline mh_point_layout layoutOf<PointVelodyneAnybotics>()
{
  return detail::with_ring(detail::xyz(sizeof(PointVelodyneAnybotics), offsetof(PointVelodyneAnybotics, intensity),
                                       offsetof(PointVelodyneAnybotics, time), MH_TIME_F32_S),
                           offsetof(PointVelodyneAnybotics, ring), MH_RING_F32, false);
}
template <> inline mh_point_layout layoutOf<PointRslidar>()
{
  return detail::with_ring(detail::xyz(sizeof(PointRslidar), offsetof(PointRslidar, intensity), offsetof(PointRslidar, timestamp), MH_TIME_F64_S_ABS),
                           offsetof(PointRslidar, ring), MH_RING_U16, true);
}

// the cloud's shape and the two Manager flags that re-order it first (lidar/manager.hpp:28-29)
struct CloudOrder
{
  uint32_t width = 0, height = 1;  // width 0: n x 1
  bool transpose_pointcloud = false, organize_pointcloud_by_ring = false;
};

inline ManagerInputConfig defaultManagerInputConfig()
{
  // struct defaults of lidar/manager.hpp:24-41 (+ GeometricConfig skip divisors)
  ManagerInputConfig c{};
  c.range_min = 0.0f;
  c.range_max = 100.0f;
  c.intensity_min = 0.0f;
  c.intensity_max = 1e10f;
  c.ns_max = 1e9f;
  c.z_offset = 0.0f;
  c.create_full_res_pointcloud = 0;
  c.point_skip_divisor = 1;
  c.ring_skip_divisor = 1;
  return c;
}

class ScanFrontEnd
{
public:
  explicit ScanFrontEnd(const std::shared_ptr<Context> & ctx) : ctx_(ctx)
  {
    ctx_->check(mh_scan_create(ctx_->get(), &scan_), "mh_scan_create");
  }
  ~ScanFrontEnd() { mh_scan_destroy(scan_); }
  ScanFrontEnd(const ScanFrontEnd &) = delete;
  ScanFrontEnd & operator=(const ScanFrontEnd &) = delete;

  // Manager::prepareInput<PointOuster>: filters, points_full_, geometric subset, unique_ns_.
  // corrected_ts_ = header_ts + last_point_ns * 1e-9 (manager.cpp:336).
  void prepareInput(const PointOuster * raw, size_t n, const ManagerInputConfig & cfg, const double header_ts)
  {
    ctx_->check(mh_scan_prepare_input(scan_, raw, n, &cfg, &info_), "mh_scan_prepare_input");
    corrected_ts_ = header_ts + info_.last_point_ns * 1.0e-9;
    unique_ns_.resize(info_.n_unique_ns);
    size_t m = 0;
    ctx_->check(mh_scan_get_unique_ns(scan_, unique_ns_.data(), unique_ns_.size(), &m), "mh_scan_get_unique_ns");
  }
  // Pipelined input: prefetch() stages the NEXT cloud (pinned copy + host-to-device copy on this front end's own copy stream)
  // while another ScanFrontEnd is being processed — it may run on another host thread; prepareInputPrefetched() is
  // prepareInput on the staged cloud (the device waits for the copy, the host does not).
  void prefetch(const PointOuster * raw, size_t n) { ctx_->check(mh_scan_prefetch(scan_, raw, n), "mh_scan_prefetch"); }
  void prepareInputPrefetched(const ManagerInputConfig & cfg, const double header_ts)
  {
    ctx_->check(mh_scan_prepare_input_prefetched(scan_, &cfg, &info_), "mh_scan_prepare_input_prefetched");
    corrected_ts_ = header_ts + info_.last_point_ns * 1.0e-9;
    unique_ns_.resize(info_.n_unique_ns);
    size_t m = 0;
    ctx_->check(mh_scan_get_unique_ns(scan_, unique_ns_.data(), unique_ns_.size(), &m), "mh_scan_get_unique_ns");
  }
  // Manager::prepareInput<PointT> for any of the reference's point types (the sensor's own records go to the device).
  // transpose_pointcloud is honoured for PointRslidar / PointVelodyneAnybotics only, as in the reference (:177-203).
  template <typename PointT>
  void prepareInput(const PointT * raw, size_t n, const ManagerInputConfig & cfg, const double header_ts, const CloudOrder & order)
  {
    const mh_point_layout L = layoutOf<PointT>();
    const bool may_transpose = std::is_same<PointT, PointRslidar>::value || std::is_same<PointT, PointVelodyneAnybotics>::value;
    const uint32_t width = order.width ? order.width : static_cast<uint32_t>(n);
    ctx_->check(mh_scan_prepare_input_layout(scan_, raw, n, &L, width, order.height, may_transpose && order.transpose_pointcloud ? 1 : 0,
                                             order.organize_pointcloud_by_ring ? 1 : 0, header_ts, &cfg, &info_),
                "mh_scan_prepare_input_layout");
    corrected_ts_ = header_ts + info_.last_point_ns * 1.0e-9;
    unique_ns_.resize(info_.n_unique_ns);
    size_t m = 0;
    ctx_->check(mh_scan_get_unique_ns(scan_, unique_ns_.data(), unique_ns_.size(), &m), "mh_scan_get_unique_ns");
  }
  // points_raw_ (manager.cpp:376-380): keep a copy of points_full_ as it was before deskewPoints — Photometric::preprocess
  // reads it (call before deskewPoints)
  void keepRaw(bool keep) { ctx_->check(mh_scan_keep_raw(scan_, keep ? 1 : 0), "mh_scan_keep_raw"); }
  const std::vector<uint32_t> & uniqueNs() const { return unique_ns_; }  // the IMU propagation runs over these
  double correctedTs() const { return corrected_ts_; }
  const mh_scan_info & info() const { return info_; }

  // Manager::deskewPoints, per-point part: T_Le_Lt[g] belongs to uniqueNs()[g] (manager.cpp:496-509)
  void deskewPoints(const std::vector<Pose3> & T_Le_Lt)
  {
    if (T_Le_Lt.size() != unique_ns_.size()) throw std::runtime_error("deskewPoints: one pose per unique timestamp");
    std::vector<float> Rt12(12 * T_Le_Lt.size());
    for (size_t g = 0; g < T_Le_Lt.size(); ++g) toFloat12(T_Le_Lt[g], &Rt12[12 * g]);
    ctx_->check(mh_scan_deskew(scan_, Rt12.data(), T_Le_Lt.size()), "mh_scan_deskew");
  }
  // the same from n poses laid out as the C ABI takes them, in double: R row-major (9), then t (3) — a caller that forms the
  // poses as plain arrays anyway (replay.hpp) skips the round trip through Pose3
  void deskewPoints(const double * Rt12, const size_t n)
  {
    if (n != unique_ns_.size()) throw std::runtime_error("deskewPoints: one pose per unique timestamp");
    std::vector<float> f(12 * n);
    for (size_t i = 0; i < 12 * n; ++i) f[i] = static_cast<float>(Rt12[i]);
    ctx_->check(mh_scan_deskew(scan_, f.data(), n), "mh_scan_deskew");
  }
  // which: 0 points_full_, 1 Be_cloud_, 2 sm_Be_cloud_ds_
  PointCloud download(int which) const
  {
    size_t n = 0;
    ctx_->check(mh_scan_get_points(scan_, which, nullptr, 0, &n), "mh_scan_get_points");
    PointCloud out(n);
    ctx_->check(mh_scan_get_points(scan_, which, out.data(), out.size(), &n), "mh_scan_get_points");
    return out;
  }
  mh_scan * underlying() { return scan_; }
  const std::shared_ptr<Context> & context() const { return ctx_; }
  mh_scan_info & mutableInfo() { return info_; }

private:
  std::shared_ptr<Context> ctx_;
  mh_scan * scan_ = nullptr;
  mh_scan_info info_{};
  std::vector<uint32_t> unique_ns_;
  double corrected_ts_ = 0;
};

// The GaussianFactor ICPFactor::linearize returns: gtsam::HessianFactor(keys()[0], H, -b, f) (geometric_factor.hpp:559-560),
// binary HessianFactor(k0, k1, H_ss, H_st, -b_s, H_tt, -b_t, f) (:459-462), from the C ABI's row-major result.
inline std::shared_ptr<HessianFactor> hessianFrom(const KeyVector & keys, bool binary, const mh_icp_result & r)
{
  gtsam::Vector g1(6);
  for (int i = 0; i < 6; ++i) g1(i) = -r.b_s[i];
  if (!binary) return std::make_shared<HessianFactor>(keys[0], matrix6(r.H_ss), g1, r.f);
  gtsam::Vector g2(6);
  for (int i = 0; i < 6; ++i) g2(i) = -r.b_t[i];
  return std::make_shared<HessianFactor>(keys[0], keys[1], matrix6(r.H_ss), matrix6(r.H_st), g1, matrix6(r.H_tt), g2, r.f);
}

class ICPFactor : public NonlinearFactor
{
public:
  using Ptr = std::shared_ptr<ICPFactor>;
  enum class RejectStatus {  // geometric_factor.hpp:35-46
    Unprocessed = 0,
    InsufficientCorresPoints,
    CorresMaxDist,
    EigenSolverFail,
    MinEigenValueLow,
    Line,
    CorresPlaneInvalid,
    MaxError,
    Valid
  };

  // unary: key_source is T_W_B, the cloud is registered to the map frame (:119-129)
  ICPFactor(const Key key_source, IncrementalVoxelMapPCL::Ptr ivox_target, const PointCloud & cloud_source,
            const RegistrationConfig & config)
  : NonlinearFactor(KeyVector{key_source}), is_binary_(false), ivox_target_(std::move(ivox_target)), n_(cloud_source.size())
  {
    create(cloud_source, config);
  }
  // binary (:131-142)
  ICPFactor(const Key key_source, const Key key_target, IncrementalVoxelMapPCL::Ptr ivox_target,
            const PointCloud & cloud_source, const RegistrationConfig & config)
  : NonlinearFactor(KeyVector{key_source, key_target}), is_binary_(true), ivox_target_(std::move(ivox_target)), n_(cloud_source.size())
  {
    create(cloud_source, config);
  }
  // unary, source cloud = the scan front end's sm_Be_cloud_ds_, taken from device memory
  ICPFactor(const Key key_source, IncrementalVoxelMapPCL::Ptr ivox_target, ScanFrontEnd & scan, const RegistrationConfig & config)
  : NonlinearFactor(KeyVector{key_source}), is_binary_(false), ivox_target_(std::move(ivox_target)), n_(scan.info().n_downsampled)
  {
    std::memset(&last_, 0, sizeof(last_));
    ctx().check(mh_icp_create_from_scan(ctx().get(), ivox_target_->underlying(), scan.underlying(), &config, 0, &icp_),
                "mh_icp_create_from_scan");
  }
  ~ICPFactor() override { mh_icp_destroy(icp_); }

  NonlinearFactor::shared_ptr clone() const override  // :160-164 deep-copies the per-point state
  {
    std::shared_ptr<ICPFactor> c(new ICPFactor(*this, CloneTag{}));
    return c;
  }
  // Whether linearize() also runs the component-localizability / status-histogram pass (getLocalizabilities' *_comp
  // outputs, lastResult().status_hist).  Geometric::getFactors reads them right after its own linearize and switches the
  // pass off before the factor goes to the smoother, whose re-linearizations never have them read.
  void computeComponents(bool on) { ctx().check(mh_icp_set_components(icp_, on ? 1 : 0), "mh_icp_set_components"); }
  size_t dim() const override { return 6; }                  // :166
  double error(const Values &) const override { return 0.0; }  // :168-174 (the reference prints and returns 0)

  std::shared_ptr<GaussianFactor> linearize(const Values & c) const override  // :231-562
  {
    // :247-257 — T_W_S = c.at<Pose3>(keys()[0]), T_W_T for the binary factor, global_z from c.at<Unit3>(G(0)) (read unconditionally)
    const PoseRM Ts = rowMajor(c.at<Pose3>(keys()[0]));
    PoseRM Tt{};
    if (is_binary_) Tt = rowMajor(c.at<Pose3>(keys()[1]));
    const A3 g = toArray(c.at<Unit3>(G(0)).unitVector());
    mh_icp_result r;
    ctx().check(mh_icp_linearize(icp_, Ts.R.data(), Ts.t.data(), is_binary_ ? Tt.R.data() : nullptr, is_binary_ ? Tt.t.data() : nullptr, g.data(), &r),
                "mh_icp_linearize");
    last_ = r;
    return toHessian(r);
  }

  // Every live factor of the smoother window re-linearized in ONE device pass (what ISAM2's update + additional
  // iterations, src/graph/manager.cpp:585-588, do one factor at a time): same results as factors[i]->linearize(c), bit
  // for bit.  All factors unary or all binary, one context.
  static std::vector<std::shared_ptr<GaussianFactor>> linearizeBatch(const std::vector<Ptr> & factors, const Values & c)
  {
    std::vector<std::shared_ptr<GaussianFactor>> out;
    if (factors.empty()) return out;
    const size_t n = factors.size();
    const bool binary = factors[0]->is_binary_;
    std::vector<mh_icp *> h(n);
    std::vector<double> Rs(9 * n), ts(3 * n), Rt(binary ? 9 * n : 0), tt(binary ? 3 * n : 0), g(3 * n);
    for (size_t i = 0; i < n; ++i) {
      const ICPFactor & f = *factors[i];
      if (f.is_binary_ != binary) throw std::runtime_error("ICPFactor::linearizeBatch: unary and binary factors mixed");
      h[i] = f.icp_;
      const PoseRM Ts = rowMajor(c.at<Pose3>(f.keys()[0]));
      std::memcpy(&Rs[9 * i], Ts.R.data(), 72);
      std::memcpy(&ts[3 * i], Ts.t.data(), 24);
      if (binary) {
        const PoseRM Tt = rowMajor(c.at<Pose3>(f.keys()[1]));
        std::memcpy(&Rt[9 * i], Tt.R.data(), 72);
        std::memcpy(&tt[3 * i], Tt.t.data(), 24);
      }
      const A3 gu = toArray(c.at<Unit3>(G(0)).unitVector());
      std::memcpy(&g[3 * i], gu.data(), 24);
    }
    std::vector<mh_icp_result> r(n);
    factors[0]->ctx().check(mh_icp_linearize_batch(h.data(), n, Rs.data(), ts.data(), binary ? Rt.data() : nullptr,
                                                   binary ? tt.data() : nullptr, g.data(), r.data()),
                            "mh_icp_linearize_batch");
    for (size_t i = 0; i < n; ++i) {
      factors[i]->last_ = r[i];
      out.push_back(factors[i]->toHessian(r[i]));
    }
    return out;
  }

  // getters, :48-72
  std::vector<RejectStatus> getStatuses() const
  {
    std::vector<int32_t> s(n_);
    if (n_) ctx().check(mh_icp_get_state(icp_, s.data(), nullptr, nullptr), "mh_icp_get_state");
    std::vector<RejectStatus> out(n_);
    for (size_t i = 0; i < n_; ++i) out[i] = static_cast<RejectStatus>(s[i]);
    return out;
  }
  std::vector<V3D> getCorresMeansTarget() const
  {
    std::vector<double> v(3 * n_);
    if (n_) ctx().check(mh_icp_get_state(icp_, nullptr, v.data(), nullptr), "mh_icp_get_state");
    std::vector<V3D> m(n_);
    for (size_t i = 0; i < n_; ++i) m[i] = vector3(&v[3 * i]);
    return m;
  }
  std::vector<V3D> getCorresNormalsTarget() const
  {
    std::vector<double> v(3 * n_);
    if (n_) ctx().check(mh_icp_get_state(icp_, nullptr, nullptr, v.data()), "mh_icp_get_state");
    std::vector<V3D> m(n_);
    for (size_t i = 0; i < n_; ++i) m[i] = vector3(&v[3 * i]);
    return m;
  }
  void getLocalizabilities(V3D & trans_comp, V3D & rot_comp, V3D & trans_final, V3D & rot_final, M33 & eigenvectors_trans,
                           M33 & eigenvectors_rot)
  {
    trans_comp = vector3(last_.loc_trans_comp);
    rot_comp = vector3(last_.loc_rot_comp);
    trans_final = vector3(last_.loc_trans_final);
    rot_final = vector3(last_.loc_rot_final);
    eigenvectors_trans = matrix3(last_.eigvec_trans);
    eigenvectors_rot = matrix3(last_.eigvec_rot);
  }
  void getDegenInfo(V3D & rot, M33 & eigenvectors_rot, V3D & trans, M33 & eigenvectors_trans)
  {
    rot = vector3(last_.degen_rot);
    eigenvectors_rot = matrix3(last_.degen_eigvec_rot);
    trans = vector3(last_.degen_trans);
    eigenvectors_trans = matrix3(last_.degen_eigvec_trans);
  }
  int getLinearizeCount() const { return last_.linearize_count; }
  const mh_icp_result & lastResult() const { return last_; }  // incl. the status histogram of geometric.cpp:280-323

private:
  std::shared_ptr<HessianFactor> toHessian(const mh_icp_result & r) const
  {
    return hessianFrom(keys(), is_binary_, r);
  }
  struct CloneTag
  {
  };
  ICPFactor(const ICPFactor & o, CloneTag)
  : NonlinearFactor(KeyVector(o.keys())), is_binary_(o.is_binary_), ivox_target_(o.ivox_target_), n_(o.n_), last_(o.last_)
  {
    ctx().check(mh_icp_clone(o.icp_, &icp_), "mh_icp_clone");
  }
  void create(const PointCloud & cloud, const RegistrationConfig & config)
  {
    std::memset(&last_, 0, sizeof(last_));
    ctx().check(mh_icp_create(ctx().get(), ivox_target_->underlying(), cloud.data(), cloud.size(), &config, is_binary_ ? 1 : 0,
                              &icp_),
                "mh_icp_create");
  }
  const Context & ctx() const { return *ivox_target_->context(); }

  const bool is_binary_;
  IncrementalVoxelMapPCL::Ptr ivox_target_;
  size_t n_;
  mh_icp * icp_ = nullptr;
  mutable mh_icp_result last_;
};

// ---------------------------------------------------------------------------------------------
// mimosa_msgs/msg/LidarGeometricDebug.msg counterpart (plain struct)
struct GeometricDebug
{
  size_t n_points_in = 0, n_points_in_sm_ds = 0;
  int n_status[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  V3D localizability_trans_comp{}, localizability_rot_comp{}, localizability_trans_final{}, localizability_rot_final{};
  bool degen_rot_bool[3] = {false, false, false}, degen_trans_bool[3] = {false, false, false};
  int n_linearize_calls = 0;
  bool map_updated = false;
};

class Geometric
{
public:
  const GeometricConfig config;

  Geometric(const std::shared_ptr<Context> & ctx, const GeometricConfig & cfg) : config(cfg), ctx_(ctx)
  {
    // geometric.cpp:23-28
    ivox_map_ = std::make_shared<IncrementalVoxelMapPCL>(ctx_, config.scan_to_map.target_ivox_map_leaf_size);
    ivox_map_->set_lru_horizon(config.lru_horizon);
    ivox_map_->set_neighbor_voxel_mode(config.neighbor_voxel_mode);
    ivox_map_->set_min_dist_in_cell(config.scan_to_map.target_ivox_map_min_dist_in_voxel);
    initial_clouds_to_force_map_update_ = config.initial_clouds_to_force_map_update;
  }

  // geometric.cpp:128-183: subset -> body frame (f32, on the GPU) -> voxel down-sample (host, :55-126)
  void preprocess(const PointCloud & points_deskewed, const std::vector<size_t> & idxs, const double ts)
  {
    if (!config.enabled) return;
    ts_ = ts;
    Be_cloud_.clear();
    Be_cloud_.reserve(idxs.size());
    for (const size_t idx : idxs) Be_cloud_.push_back(points_deskewed[idx]);
    float Rt[12];
    toFloat12(config.T_B_L, Rt);
    if (!Be_cloud_.empty())
      ctx_->check(mh_transform_f32(ctx_->get(), Be_cloud_.data(), Be_cloud_.size(), Rt, Rt + 9), "mh_transform_f32");
    downsample(Be_cloud_, sm_Be_cloud_ds_, config.scan_to_map.source_voxel_grid_filter_leaf_size, 20,
               config.scan_to_map.source_voxel_grid_min_dist_in_voxel);
    device_scan_ = nullptr;
    debug_.n_points_in = points_deskewed.size();
    debug_.n_points_in_sm_ds = sm_Be_cloud_ds_.size();
  }

  // The same on a device-resident scan: subset + body transform + down-sampler run on the GPU; Be_cloud_ stays there
  // (updateMap inserts it into the map on the device, geometric.cpp:483-495).
  void preprocess(ScanFrontEnd & scan, const double ts)
  {
    if (!config.enabled) return;
    ts_ = ts;
    float Rt[12];
    toFloat12(config.T_B_L, Rt);
    ctx_->check(mh_scan_preprocess_geometric(scan.underlying(), Rt, Rt + 9, config.scan_to_map.source_voxel_grid_filter_leaf_size,
                                             20, config.scan_to_map.source_voxel_grid_min_dist_in_voxel, &scan.mutableInfo()),
                "mh_scan_preprocess_geometric");
    Be_cloud_.clear();
    sm_Be_cloud_ds_.clear();
    device_scan_ = &scan;
    debug_.n_points_in = scan.info().n_full;
    debug_.n_points_in_sm_ds = scan.info().n_downsampled;
  }

  // geometric.cpp:185-328
  void getFactors(const Key & key, const Values & values, NonlinearFactorGraph & graph, M66 & eigenvectors_block_matrix,
                  V6D & degen_directions)
  {
    if (!config.enabled) return;
    factor_ = device_scan_ ? std::make_shared<ICPFactor>(key, ivox_map_, *device_scan_, config.scan_to_map)
                           : std::make_shared<ICPFactor>(key, ivox_map_, sm_Be_cloud_ds_, config.scan_to_map);
    auto tmp = factor_->linearize(values);  // linearized right away so localizability is available (:196)
    (void)tmp;
    V3D tc, rc, tf, rf;
    M33 et, er;
    factor_->getLocalizabilities(tc, rc, tf, rf, et, er);
    eigenvectors_block_matrix.setZero();
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        eigenvectors_block_matrix(r, c) = er(r, c);
        eigenvectors_block_matrix(3 + r, 3 + c) = et(r, c);
      }
    for (int i = 0; i < 3; ++i) {  // :218-228
      degen_directions(i) = rc(i) < config.scan_to_map.degen_thresh_rot;
      degen_directions(3 + i) = tc(i) < config.scan_to_map.degen_thresh_trans;
      debug_.degen_rot_bool[i] = degen_directions(i) != 0;
      debug_.degen_trans_bool[i] = degen_directions(3 + i) != 0;
    }
    debug_.localizability_trans_comp = tc;
    debug_.localizability_rot_comp = rc;
    debug_.localizability_trans_final = tf;
    debug_.localizability_rot_final = rf;
    for (int i = 0; i < 9; ++i) debug_.n_status[i] = factor_->lastResult().status_hist[i];  // :280-323
    // nothing reads the components or the histogram of this factor again (the smoother only takes the HessianFactor):
    // its re-linearizations skip that pass
    factor_->computeComponents(false);
    graph.add(factor_);
  }

  // geometric.cpp:427-513
  void updateMap(const Key key, const Values & values)
  {
    if (!config.enabled) return;
    if (factor_) debug_.n_linearize_calls = factor_->getLinearizeCount();
    const Pose3 T_W_Be = values.at<Pose3>(key);
    bool update_map = true;
    if (!map_poses_.empty()) {
      float min_diff_trans = std::numeric_limits<float>::max();
      size_t min_diff_index = 0;
      for (size_t i = 0; i < map_poses_.size(); ++i) {
        const Point3 & a = map_poses_[i].translation();
        const Point3 & b = T_W_Be.translation();
        const float d = static_cast<float>(std::sqrt((a(0) - b(0)) * (a(0) - b(0)) + (a(1) - b(1)) * (a(1) - b(1)) + (a(2) - b(2)) * (a(2) - b(2))));
        if (d < min_diff_trans) {
          min_diff_trans = d;
          min_diff_index = i;
        }
      }
      // rot_diff = R_B_L^-1 * between(R_kf, R_now) * R_B_L, yaw-pitch-roll magnitudes (:454-458)
      const Rot3 & RBL = config.T_B_L.rotation();
      const Rot3 dR = RBL.inverse() * map_poses_[min_diff_index].rotation().between(T_W_Be.rotation()) * RBL;
      const M33 d = dR.matrix();  // Rot3::ypr() spelled out
      const double yaw = std::atan2(d(1, 0), d(0, 0));
      const double pitch = std::atan2(-d(2, 0), std::sqrt(d(2, 1) * d(2, 1) + d(2, 2) * d(2, 2)));
      const double roll = std::atan2(d(2, 1), d(2, 2));
      const double ypr_max = std::max(std::fabs(yaw), std::max(std::fabs(pitch), std::fabs(roll)));
      if (min_diff_trans > config.map_keyframe_trans_thresh)
        update_map = true;
      else if (ypr_max > config.map_keyframe_rot_thresh_deg * 0.017453293)  // DEG2RAD = PCL's macro (x)*0.017453293
        update_map = true;
      else
        update_map = false;
    }
    if (initial_clouds_to_force_map_update_ > 0) {
      update_map = true;
      initial_clouds_to_force_map_update_--;
    }
    debug_.map_updated = update_map;
    if (!update_map) return;
    // world transform in f32 (:483-490), then copy-then-insert so live factors keep their snapshot (:494-495)
    ivox_map_ = ivox_map_->fork();  // device-to-device copy; the previous map lives on in the factors that hold it
    if (device_scan_) {
      ivox_map_->insertBodyCloud(device_scan_->underlying(), T_W_Be);  // transform + insert on the device
    } else {
      PointCloud W = Be_cloud_;
      float Rt[12];
      toFloat12(T_W_Be, Rt);
      if (!W.empty()) ctx_->check(mh_transform_f32(ctx_->get(), W.data(), W.size(), Rt, Rt + 9), "mh_transform_f32");
      ivox_map_->insert(W);
    }
    map_poses_.push_back(T_W_Be);
  }

  const GeometricDebug & debug() const { return debug_; }
  const PointCloud & sourceCloud() const { return sm_Be_cloud_ds_; }
  const IncrementalVoxelMapPCL::Ptr & map() const { return ivox_map_; }
  const ICPFactor::Ptr & factor() const { return factor_; }

  // Geometric::downsample, geometric.cpp:55-126 with FlatContainerMinimal::add (lidar/utils.hpp:260-278):
  // greedy, input order; output = voxels in first-seen order, points in acceptance order.
  static void downsample(const PointCloud & in, PointCloud & out, const double leaf_size, const size_t max_points_per_voxel,
                         const double min_dist_in_voxel)
  {
    struct Key3
    {
      int x, y, z;
      bool operator==(const Key3 & o) const { return x == o.x && y == o.y && z == o.z; }
    };
    struct Hash
    {
      size_t operator()(const Key3 & k) const
      {  // XORVector3iHash, lidar/utils.hpp:228-238
        return static_cast<size_t>((k.x * 9132043225175502913ull) ^ (k.y * 7277549399757405689ull) ^
                                   (k.z * 6673468629021231217ull));
      }
    };
    auto ffloor = [](double v) {
      const int n = static_cast<int>(v);
      return n - (v < static_cast<double>(n) ? 1 : 0);
    };
    const double inv = 1.0 / leaf_size, min_sq = min_dist_in_voxel * min_dist_in_voxel;
    std::unordered_map<Key3, size_t, Hash> voxels;
    voxels.reserve(in.size() / 2 + 1);
    std::vector<std::vector<uint32_t>> kept;
    for (size_t i = 0; i < in.size(); ++i) {
      const double px = in[i].x, py = in[i].y, pz = in[i].z;
      const Key3 c{ffloor(px * inv), ffloor(py * inv), ffloor(pz * inv)};
      auto f = voxels.find(c);
      if (f == voxels.end()) {
        f = voxels.emplace(c, kept.size()).first;
        kept.emplace_back();
      }
      auto & cell = kept[f->second];
      if (cell.size() >= max_points_per_voxel) continue;
      bool close = false;
      for (const uint32_t j : cell) {
        const double dx = in[j].x - px, dy = in[j].y - py, dz = in[j].z - pz;
        if (dx * dx + (dy * dy + dz * dz) < min_sq) {
          close = true;
          break;
        }
      }
      if (!close) cell.push_back(static_cast<uint32_t>(i));
    }
    out.clear();
    for (const auto & cell : kept)
      for (const uint32_t j : cell) out.push_back(in[j]);
  }

private:
  std::shared_ptr<Context> ctx_;
  PointCloud Be_cloud_, sm_Be_cloud_ds_;
  ScanFrontEnd * device_scan_ = nullptr;  // set by the device preprocess: the factor takes its cloud from there
  ICPFactor::Ptr factor_;
  IncrementalVoxelMapPCL::Ptr ivox_map_;
  std::vector<Pose3> map_poses_;
  size_t initial_clouds_to_force_map_update_ = 0;
  double ts_ = 0;
  GeometricDebug debug_;
};

// Manager::deskewPoints, pose part (manager.cpp:455-499), for callers that run their own IMU preintegration
// (gtsam::PreintegratedImuMeasurements stays the caller's: SURVEY.md §7).  Inputs are what that loop has in
// hand: the interpolated IMU samples (time, accelerometer, gyroscope — `imu_measurements`), the NavState the
// preintegrator predicted AT each sample time (nav[j] = state at imu_t[j]; nav[0] = the previous state), the
// bias, the gravity direction (unit) and magnitude, the scan's distinct timestamps and header time, and T_B_S.
// Output: T_Le_Lt per distinct timestamp, exactly the constant-acceleration / constant-rate extrapolation of
// :478-489 followed by T_Le_W * T_W_Bt * T_B_S (:493-497).  Timestamps after the last IMU sample get no pose
// (the reference's vector simply ends there).
using gtsam::NavState;

inline std::vector<Pose3> computeDeskewPoses(const std::vector<double> & imu_t, const std::vector<V3D> & imu_acc,
                                             const std::vector<V3D> & imu_gyro, const std::vector<NavState> & nav,
                                             const V3D & bias_acc, const V3D & bias_gyro, const V3D & gravity_unit,
                                             const double gravity_norm, const std::vector<uint32_t> & unique_ns,
                                             const double header_ts, const Pose3 & T_B_S)
{
  if (imu_t.size() < 2) throw std::runtime_error("Preintegration not possible as there are less than 2 measurements P1");  // :442-446
  if (imu_acc.size() != imu_t.size() || imu_gyro.size() != imu_t.size() || nav.size() != imu_t.size())
    throw std::runtime_error("computeDeskewPoses: one measurement and one state per IMU sample");
  std::vector<Pose3> T_W_Bts;
  T_W_Bts.reserve(unique_ns.size());
  size_t u = 0;
  for (size_t c = 0; c + 1 < imu_t.size(); ++c) {  // curr = c, next = c + 1 (:459-466)
    const NavState & curr = nav[c];
    const A9 Rc = rowMajor(curr.pose().rotation().matrix());
    const A3 pc = toArray(curr.pose().translation()), vc = toArray(curr.velocity());
    while (u < unique_ns.size()) {
      const double ts = header_ts + unique_ns[u] * 1.0e-9;  // globalTs, manager.hpp:94
      if (ts > imu_t[c + 1]) break;
      const double dt = ts - imu_t[c];
      double acc[3], omega[3];
      for (int i = 0; i < 3; ++i) {
        acc[i] = imu_acc[c](i) - bias_acc(i);     // ConstantBias::correctAccelerometer
        omega[i] = imu_gyro[c](i) - bias_gyro(i);  // ::correctGyroscope
      }
      // :478-489  R = R_c Exp(omega dt),  p = p_c + v dt + 1/2 R_c a dt^2 + 1/2 g_hat |g| dt^2
      const Rot3 R = curr.pose().rotation() * Rot3::Expmap(V3D(omega[0] * dt, omega[1] * dt, omega[2] * dt));
      double p[3];
      for (int i = 0; i < 3; ++i) {
        const double Ra = Rc[3 * i] * acc[0] + Rc[3 * i + 1] * acc[1] + Rc[3 * i + 2] * acc[2];
        p[i] = pc[i] + vc[i] * dt + 0.5 * Ra * dt * dt + 0.5 * gravity_unit(i) * gravity_norm * dt * dt;
      }
      T_W_Bts.push_back(Pose3(R, Point3(p[0], p[1], p[2])));
      ++u;
    }
  }
  const Pose3 T_Le_W = T_B_S.inverse() * nav.back().pose().inverse();  // propagated state = T_W_Be (:492-493)
  std::vector<Pose3> out;
  out.reserve(T_W_Bts.size());
  for (const Pose3 & T : T_W_Bts) out.push_back(T_Le_W * T * T_B_S);
  return out;
}

// Manager::deskewPoints' per-point part (manager.cpp:496-509).  T_Le_Lt[g] is the pose of the sensor at
// unique timestamp g in the scan-end frame — computed by the caller's IMU propagation (:455-499), which
// stays on the CPU.  Points are transformed in place on the GPU.
inline void deskewPoints(const Context & ctx, PointCloud & points_full, const std::vector<uint32_t> & unique_ns,
                         const std::vector<Pose3> & T_Le_Lt, const Pose3 * T_B_L = nullptr)
{
  if (unique_ns.size() != T_Le_Lt.size()) throw std::runtime_error("deskewPoints: one pose per unique timestamp");
  std::vector<float> Rt12(12 * T_Le_Lt.size());
  for (size_t g = 0; g < T_Le_Lt.size(); ++g) toFloat12(T_Le_Lt[g], &Rt12[12 * g]);
  float Rb[12];
  if (T_B_L) toFloat12(*T_B_L, Rb);
  ctx.check(mh_deskew(ctx.get(), points_full.data(), points_full.size(), unique_ns.data(), Rt12.data(), unique_ns.size(),
                      T_B_L ? Rb : nullptr, T_B_L ? Rb + 9 : nullptr),
            "mh_deskew");
}

}  // namespace lidar
}  // namespace mimosa_hip
