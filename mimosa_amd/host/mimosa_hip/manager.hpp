// C++ host mirror of lidar::Manager (include/mimosa/lidar/manager.hpp:44-93, src/lidar/manager.cpp:45-147, :385-584): the
// per-scan LiDAR callback in the reference's own method names and call order
//
//   callback -> prepareInput<PointT> -> graph.getStateUpto / declare -> deskewPoints -> preprocess -> getFactors -> define
//            -> postDefineUpdate
//
// over the pieces of lidar.hpp / photometric.hpp (ScanFrontEnd, Geometric, Photometric), which run on the GPU behind the C ABI.
// What the reference's Manager talks to and this build does not re-implement stays behind two interfaces with the
// reference's method names: graph::Manager (GTSAM / ISAM2: src/graph/manager.cpp) and imu::Manager + GTSAM's
// PreintegratedImuMeasurements (src/imu/manager.cpp).  A deployment hands in thin adapters over the real objects; the
// replay harness (replay.hpp: ManagerReplay) hands in stand-ins.  ROS subscription, publishers, loggers, tf broadcasting and
// the debug message are dropped; the timings of the debug message are kept as plain numbers.
#pragma once

#include <chrono>
#include <utility>

#include "photometric.hpp"

namespace mimosa_hip
{
// include/mimosa/state.hpp:22-50
class State
{
public:
  State() : key_(0), ts_(0) {}
  Key key() const { return key_; }
  double ts() const { return ts_; }
  const gtsam::NavState & navState() const { return nav_state_; }
  const V3D & biasAcc() const { return bias_acc_; }    // imuBias().accelerometer()
  const V3D & biasGyro() const { return bias_gyro_; }  // imuBias().gyroscope()
  const Unit3 & gravity() const { return gravity_; }
  void update(const Key key, const double ts, const gtsam::NavState & nav_state, const V3D & bias_acc, const V3D & bias_gyro, const Unit3 & gravity)
  {
    key_ = key;
    ts_ = ts;
    nav_state_ = nav_state;
    bias_acc_ = bias_acc;
    bias_gyro_ = bias_gyro;
    gravity_ = gravity;
  }

private:
  Key key_;
  double ts_;
  gtsam::NavState nav_state_;
  V3D bias_acc_ = V3D(0, 0, 0), bias_gyro_ = V3D(0, 0, 0);
  Unit3 gravity_ = Unit3(0.0, 0.0, -1.0);
};

// (timestamp, [acc (3), gyro (3)]): imu::Manager's ImuBuffer (head<3> = accelerometer, tail<3> = gyroscope, manager.cpp:461-462)
using ImuBuffer = std::vector<std::pair<double, V6D>>;

namespace graph
{
// include/mimosa/graph/manager.hpp:74-86
enum class DeclarationResult {
  FAILURE_CANNOT_INIT_ON_MODALITY = 0,
  FAILURE_ATTITUDE_ESTIMATION,
  FAILURE_OLDER_THAN_INITIALIZATION,
  FAILURE_OLDER_THAN_LAG,
  FAILURE_OLDER_THAN_MAX_LATENCY,
  FAILURE_CANNOT_HANDLE_OUT_OF_ORDER,
  SUCCESS_INITIALIZED,
  SUCCESS_SAME_KEY,
  SUCCESS_OUT_OF_ORDER,
  SUCCESS_NORMAL,
};
// The part of graph::Manager that lidar::Manager calls (src/lidar/manager.cpp:87-92, :114, :120, :538)
class ManagerInterface
{
public:
  virtual ~ManagerInterface() = default;
  virtual void getStateUpto(const double ts, State & state) = 0;
  virtual DeclarationResult declare(const double ts, size_t & new_key, const bool use_to_init) = 0;
  virtual Pose3 getPoseAt(const size_t key) = 0;
  virtual Values getCurrentOptimizedValues() = 0;
  virtual void define(const NonlinearFactorGraph & new_factors, Values & optimized_values, const DeclarationResult result) = 0;
};
}  // namespace graph

namespace imu
{
// imu::Manager::getInterpolatedMeasurements (src/lidar/manager.cpp:437) + the three calls made on
// gtsam::PreintegratedImuMeasurements (:448, :459-466; predict is the fork's 3-argument form with the gravity direction)
class ManagerInterface
{
public:
  virtual ~ManagerInterface() = default;
  virtual void getInterpolatedMeasurements(const double t0, const double t1, ImuBuffer & out, const bool include_end) = 0;
  virtual double gravityNorm() const = 0;  // params()->getGravity().norm()
  virtual void resetIntegrationAndSetBias(const State & state) = 0;
  virtual void integrateMeasurement(const V3D & acc, const V3D & gyro, const double dt) = 0;
  virtual gtsam::NavState predict(const State & from) = 0;
};
}  // namespace imu

namespace lidar
{
// lidar/manager.hpp:22-41 (+ SensorManagerBaseConfig's T_B_S / use_to_init; frame names, log settings, the odometry logger dropped)
struct ManagerConfig
{
  bool enabled = true;
  bool use_to_init = true;
  Pose3 T_B_S = Pose3();
  bool transpose_pointcloud = false;
  bool organize_pointcloud_by_ring = false;
  float range_min = 0.0f;
  float range_max = 100.0f;
  float intensity_min = 0.0f;
  float intensity_max = 1e10f;
  float ns_max = 1e9f;
  float z_offset = 0.0f;  // lidar_to_sensor_transform[2][3] * 1e-3 (manager.cpp:30)
  bool create_full_res_pointcloud = false;
};

struct ManagerDebug  // mimosa_msgs/LidarManagerDebug.msg, the timing fields (ms)
{
  double t_declare = 0, t_deskew = 0, t_preprocess_geo_photo = 0, t_factor_prep = 0, t_define = 0, t_post_define_update = 0, t_full = 0;
  bool initialized = false;
};

class Manager
{
public:
  Manager(const std::shared_ptr<Context> & ctx, const ManagerConfig & config, const GeometricConfig & geometric_config, const PhotometricConfig & photometric_config,
          graph::ManagerInterface & graph_manager, imu::ManagerInterface & imu_manager, const std::shared_ptr<Context> & photometric_ctx = nullptr)
  : config_(config), ctx_(ctx), scan_(ctx), graph_manager_(graph_manager), imu_manager_(imu_manager)
  {
    geometric_.reset(new Geometric(ctx_, geometric_config));                                               // manager.cpp:27-28
    photometric_.reset(new Photometric(photometric_ctx ? photometric_ctx : ctx_, photometric_config));
    if (photometric_->config.enabled) scan_.keepRaw(true);  // points_raw_ (manager.cpp:376-380)
    geometric_eigenvectors_block_matrix_ = M66::Identity();
    for (int i = 0; i < 6; ++i) geometric_degen_directions_(i) = 1.0;
  }

  // manager.cpp:45-147.  `cloud` = the PointCloud2's records as PointT (decodePointType picks the instantiation in the
  // reference; here the caller does by the type it passes), header_stamp = msg->header.stamp.
  template <typename PointT>
  void callback(const PointT * cloud, const size_t n, const double header_stamp, const CloudOrder & order = CloudOrder())
  {
    if (!config_.enabled) return;
    const auto t_begin = now();
    prepareInput<PointT>(cloud, n, header_stamp, order);

    graph_manager_.getStateUpto(header_ts_, prev_state_);

    const auto t_decl = now();
    const graph::DeclarationResult dr = graph_manager_.declare(corrected_ts_, new_key_, config_.use_to_init);
    if (!handleDeclarationResult(dr)) return;
    debug_.t_declare = ms(t_decl);

    deskewPoints();

    preprocess(X(new_key_));

    Pose3 T_W_Bk_opt;
    if (first_) {
      first_ = false;
      opt_values_ = graph_manager_.getCurrentOptimizedValues();
      initialized_ = true;
      debug_.initialized = true;
      T_W_Bk_opt = graph_manager_.getPoseAt(new_key_);
      opt_values_.insert_or_assign(X(new_key_), T_W_Bk_opt);
    } else {
      Values initial_values = opt_values_;
      initial_values.insert_or_assign(X(new_key_), propagated_state_.pose());
      NonlinearFactorGraph new_factors;
      getFactors(initial_values, new_factors);

      define(new_factors, opt_values_, dr);

      T_W_Bk_opt = opt_values_.at<Pose3>(X(new_key_));
    }

    postDefineUpdate(X(new_key_), opt_values_);
    last_pose_ = T_W_Bk_opt;
    debug_.t_full = ms(t_begin);
  }

  const Pose3 & lastPose() const { return last_pose_; }  // T_W_Bk_opt of the last callback (what publishResults logs)
  size_t lastKey() const { return new_key_; }
  const ManagerDebug & debug() const { return debug_; }
  Geometric & geometric() { return *geometric_; }
  Photometric & photometric() { return *photometric_; }
  ScanFrontEnd & scan() { return scan_; }
  const gtsam::NavState & propagatedState() const { return propagated_state_; }

private:
  using clk = std::chrono::steady_clock;
  static clk::time_point now() { return clk::now(); }
  static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
  double globalTs(const uint32_t value) const { return header_ts_ + value * 1.0e-9; }  // manager.hpp:94

  // sensor_manager_base.hpp:208-260: every FAILURE_* skips the message
  static bool handleDeclarationResult(const graph::DeclarationResult r) { return r >= graph::DeclarationResult::SUCCESS_INITIALIZED; }

  // manager.cpp:149-383: filters, points_full_, the geometric subset, the distinct timestamps — on the device
  template <typename PointT>
  void prepareInput(const PointT * cloud, const size_t n, const double header_stamp, const CloudOrder & order)
  {
    header_ts_ = header_stamp;
    ManagerInputConfig in = defaultManagerInputConfig();
    in.range_min = config_.range_min;
    in.range_max = config_.range_max;
    in.intensity_min = config_.intensity_min;
    in.intensity_max = config_.intensity_max;
    in.ns_max = config_.ns_max;
    in.z_offset = config_.z_offset;
    in.create_full_res_pointcloud = (config_.create_full_res_pointcloud || photometric_->config.enabled) ? 1 : 0;
    in.point_skip_divisor = geometric_->config.point_skip_divisor;
    in.ring_skip_divisor = geometric_->config.ring_skip_divisor;
    CloudOrder o = order;
    o.transpose_pointcloud = config_.transpose_pointcloud;
    o.organize_pointcloud_by_ring = config_.organize_pointcloud_by_ring;
    prepareInputTyped(cloud, n, in, o);
    corrected_ts_ = scan_.correctedTs();  // header + last point's offset (manager.cpp:336)
  }
  void prepareInputTyped(const PointOuster * cloud, const size_t n, const ManagerInputConfig & in, const CloudOrder &) { scan_.prepareInput(cloud, n, in, header_ts_); }
  template <typename PointT>
  void prepareInputTyped(const PointT * cloud, const size_t n, const ManagerInputConfig & in, const CloudOrder & o)
  {
    scan_.prepareInput<PointT>(cloud, n, in, header_ts_, o);
  }

  // manager.cpp:385-512
  void deskewPoints()
  {
    const auto t0 = now();
    const std::vector<uint32_t> & unique_ns = scan_.uniqueNs();
    T_Le_Lt_.clear();
    if (!initialized_) {  // :399-408 — the first cloud is not deskewed
      T_Le_Lt_.assign(unique_ns.size(), Pose3());
      scan_.deskewPoints(T_Le_Lt_);  // identity per timestamp: points_full_ = points_raw_
      propagated_state_ = prev_state_.navState();
      debug_.t_deskew = ms(t0);
      return;
    }
    State state = prev_state_;
    if (!unique_ns.empty() && state.ts() > globalTs(unique_ns.front()))  // :421-433 — the state is moved back to the first point
      state.update(state.key(), globalTs(unique_ns.front()), state.navState(), state.biasAcc(), state.biasGyro(), state.gravity());

    ImuBuffer imu_measurements;
    imu_manager_.getInterpolatedMeasurements(state.ts(), corrected_ts_, imu_measurements, true);  // :437
    if (imu_measurements.size() < 2) throw std::runtime_error("Preintegration not possible as there are less than 2 measurements P1");  // :439-446

    imu_manager_.resetIntegrationAndSetBias(state);  // :448
    std::vector<double> imu_t;
    std::vector<V3D> imu_acc, imu_gyro;
    std::vector<gtsam::NavState> nav;
    propagated_state_ = state.navState();
    nav.push_back(propagated_state_);
    for (size_t c = 0; c < imu_measurements.size(); ++c) {
      const V6D & m = imu_measurements[c].second;
      imu_t.push_back(imu_measurements[c].first);
      imu_acc.push_back(V3D(m(0), m(1), m(2)));
      imu_gyro.push_back(V3D(m(3), m(4), m(5)));
      if (c + 1 < imu_measurements.size()) {  // :455-466
        imu_manager_.integrateMeasurement(imu_acc.back(), imu_gyro.back(), imu_measurements[c + 1].first - imu_measurements[c].first);
        propagated_state_ = imu_manager_.predict(state);  // the state at the time of the next sample
        nav.push_back(propagated_state_);
      }
    }
    // :468-499 — constant-acceleration / constant-rate extrapolation to every distinct timestamp, then T_Le_W T_W_Bt T_B_S
    T_Le_Lt_ = computeDeskewPoses(imu_t, imu_acc, imu_gyro, nav, state.biasAcc(), state.biasGyro(), state.gravity().unitVector(), imu_manager_.gravityNorm(),
                                  unique_ns, header_ts_, config_.T_B_S);
    if (T_Le_Lt_.size() != unique_ns.size()) throw std::runtime_error("deskewPoints: IMU samples end before the last point of the cloud");
    scan_.deskewPoints(T_Le_Lt_);  // :501-509, per point, on the device
    debug_.t_deskew = ms(t0);
  }

  // manager.cpp:514-524
  void preprocess(const Key key)
  {
    const auto t0 = now();
    photometric_->preprocess(scan_, T_Le_Lt_, corrected_ts_, key);
    geometric_->preprocess(scan_, corrected_ts_);
    debug_.t_preprocess_geo_photo = ms(t0);
  }

  // manager.cpp:526-549
  void getFactors(const Values & initial_values, NonlinearFactorGraph & new_factors)
  {
    const auto t0 = now();
    geometric_->getFactors(X(new_key_), initial_values, new_factors, geometric_eigenvectors_block_matrix_, geometric_degen_directions_);
    // the reference hands the photometric factor the identity / all-ones selection (:541-545)
    photometric_->getFactors(initial_values, new_factors);
    debug_.t_factor_prep = ms(t0);
  }

  // manager.cpp:551-561
  void define(const NonlinearFactorGraph & new_factors, Values & optimized_values, const graph::DeclarationResult dr)
  {
    const auto t0 = now();
    graph_manager_.define(new_factors, optimized_values, dr);
    debug_.t_define = ms(t0);
  }

  // manager.cpp:563-584
  void postDefineUpdate(const Key key, const Values & values)
  {
    const auto t0 = now();
    geometric_->updateMap(key, values);
    std::vector<V3D> bias_directions;
    for (int i = 0; i < 3; ++i)
      if (geometric_degen_directions_(i + 3) != 0.0)
        bias_directions.push_back(V3D(geometric_eigenvectors_block_matrix_(3, 3 + i), geometric_eigenvectors_block_matrix_(4, 3 + i),
                                      geometric_eigenvectors_block_matrix_(5, 3 + i)));  // bottomRightCorner<3, 3>().col(i)
    if (bias_directions.empty()) {
      bias_directions.push_back(V3D(1, 0, 0));
      bias_directions.push_back(V3D(0, 1, 0));
      bias_directions.push_back(V3D(0, 0, 1));
    }
    photometric_->updateMap(values, bias_directions);
    debug_.t_post_define_update = ms(t0);
  }

  const ManagerConfig config_;
  std::shared_ptr<Context> ctx_;
  ScanFrontEnd scan_;  // points_full_, points_raw_, geometric_point_idxs_, unique_ns_: device-resident
  graph::ManagerInterface & graph_manager_;
  imu::ManagerInterface & imu_manager_;
  std::unique_ptr<Geometric> geometric_;
  std::unique_ptr<Photometric> photometric_;
  std::vector<Pose3> T_Le_Lt_;  // interpolated_map_T_Le_Lt_, in unique_ns_ order
  gtsam::NavState propagated_state_;
  State prev_state_;
  M66 geometric_eigenvectors_block_matrix_;
  V6D geometric_degen_directions_;
  Values opt_values_;
  Pose3 last_pose_;
  double header_ts_ = 0, corrected_ts_ = 0;
  size_t new_key_ = 0;
  bool first_ = true, initialized_ = false;
  ManagerDebug debug_;
};

}  // namespace lidar
}  // namespace mimosa_hip
