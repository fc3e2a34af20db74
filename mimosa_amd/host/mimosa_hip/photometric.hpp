// C++ host mirror of the reference's LiDAR photometric classes over the C ABI (include/mimosa_hip.h, mh_photo_*):
//
//   PhotometricConfig   include/mimosa/lidar/photometric_config.hpp:16-83
//   PhotometricFactor   include/mimosa/lidar/photometric_factor.hpp:22-357
//   Photometric         include/mimosa/lidar/photometric.hpp:30-89, src/lidar/photometric.cpp
//
// Same constructor / method / getter names and argument meaning, same call order (preprocess -> getFactors ->
// linearize ... -> updateMap), same error behaviour: std::runtime_error where the reference throws ("No features in
// a_features", a feature whose projection leaves the image — photometric_utils.cpp:90-97 — is reported by the library
// as a count and re-thrown here).  ROS publishers, loggers, cv::Mat visualisation and config_utilities are dropped.
// Header-only; link with libmimosa_hip.so.  All arithmetic happens behind the C ABI on the GPU: nothing here touches a
// pixel, and there is no CPU fallback.
#pragma once

#include <utility>

#include "lidar.hpp"

namespace mimosa_hip
{
namespace lidar
{
// photometric_config.hpp:16-83 (frame names, log settings, visualize and the derived fx / fy / cx dropped: the library
// derives them from the beam table exactly like declare_config does)
struct PhotometricConfig
{
  bool enabled = true;
  bool destagger = true;
  std::vector<int> pixel_shift_by_row = {};
  Pose3 T_B_L = Pose3();
  size_t rows = 128;
  size_t cols = 512;
  float range_min = 0.1f;
  float range_max = 100;
  int erosion_buffer = 2;
  int patch_size = 5;
  int margin_size = 2;
  std::vector<uint8_t> static_mask = {};  // static_mask_path's image, rows * cols, 0 = invalid; empty = none
  float intensity_scale = 0.25f;
  float intensity_gamma = 0.8f;
  bool remove_lines = true;
  bool filter_brightness = true;
  bool gaussian_blur = true;
  int gaussian_blur_size = 3;
  float gradient_threshold = 20;
  float max_dist_from_mean = 0.2f;
  float max_dist_from_plane = 0.1f;
  int nma_radius = 10;
  size_t num_features_detect = 60;
  float occlusion_range_diff_threshold = 0.1f;
  int max_feature_life_time = 30;
  std::vector<float> beam_altitude_angles = {};
  std::vector<double> high_pass_fir = {};
  std::vector<double> low_pass_fir = {};
  std::vector<int> brightness_window_size = {};  // {width, height}
  float lidar_origin_to_beam_origin_mm = 0.0f;
  bool rotate_patch_to_align_with_gradient = false;
  std::vector<std::pair<int, int>> edgelet_patch_offsets = {
    {-2, -2}, {-1, -2}, {0, -2}, {1, -2}, {2, -2}, {-2, -1}, {-1, -1}, {0, -1}, {1, -1}, {2, -1}, {-2, 0}, {-1, 0}, {0, 0},
    {1, 0},   {2, 0},   {-2, 1}, {-1, 1}, {0, 1},  {1, 1},   {2, 1},   {-2, 2}, {-1, 2}, {0, 2},  {1, 2},  {2, 2},
  };
  bool use_robust_cost_function = true;
  std::string robust_cost_function = "huber";
  double robust_cost_function_parameter = 1.345;
  double error_scale = 1.0;
  double max_error = 255.0;
  double sigma = 0.1;
};

// Feature, photometric_utils.hpp:30-52 (what Photometric keeps in map_Le_features_)
struct Feature
{
  uint32_t id = 0;
  int life_time = 0;
  std::array<double, 2> center{0, 0};
  V3D normal = V3D(0, 0, 0);
  double mean_intensity = 0, sigma_intensity = 0;
  std::vector<V3D> Le_ps;
  std::vector<double> intensities, psi_intensities;
};

class Photometric;

class PhotometricFactor : public NonlinearFactor
{
public:
  using Ptr = std::shared_ptr<PhotometricFactor>;
  enum class RejectStatus {  // photometric_factor.hpp:36-47
    Unprocessed = 0,
    PointProjectUndistorted,
    PointRange,
    PointProject,
    PointMask,
    PointMaskMargin,
    PointRangeDiff,
    MaxError,
    Valid
  };

  ~PhotometricFactor() override { mh_photo_factor_destroy(f_); }

  NonlinearFactor::shared_ptr clone() const override  // :120-124
  {
    std::shared_ptr<PhotometricFactor> c(new PhotometricFactor(ctx_, keys(), is_binary_));
    ctx_->check(mh_photo_factor_clone(f_, &c->f_), "mh_photo_factor_clone");
    c->last_ = last_;
    return c;
  }
  size_t dim() const override { return 6; }                  // :126
  double error(const Values &) const override { return 0.0; }  // :128-134 (ignored by the reference too)

  std::shared_ptr<GaussianFactor> linearize(const Values & c) const override  // :136-355
  {
    // :145-149 — T_W_Bb = c.at<Pose3>(keys()[0]) (and T_W_Ba for the binary factor)
    const PoseRM Tb = rowMajor(c.at<Pose3>(keys()[0]));
    PoseRM Ta{};
    if (is_binary_) Ta = rowMajor(c.at<Pose3>(keys()[1]));
    mh_photo_result r;
    ctx_->check(mh_photo_factor_linearize(f_, Tb.R.data(), Tb.t.data(), is_binary_ ? Ta.R.data() : nullptr, is_binary_ ? Ta.t.data() : nullptr, &r),
                "mh_photo_factor_linearize");
    return toHessian(r);
  }

  // linearize() split in two so that the smoother can queue it next to ICPFactor::linearizeBatch on the same stream
  void linearizeAsync(const Values & c) const
  {
    const PoseRM Tb = rowMajor(c.at<Pose3>(keys()[0]));
    PoseRM Ta{};
    if (is_binary_) Ta = rowMajor(c.at<Pose3>(keys()[1]));
    ctx_->check(mh_photo_factor_linearize_async(f_, Tb.R.data(), Tb.t.data(), is_binary_ ? Ta.R.data() : nullptr, is_binary_ ? Ta.t.data() : nullptr),
                "mh_photo_factor_linearize_async");
  }
  std::shared_ptr<GaussianFactor> collect() const
  {
    mh_photo_result r;
    ctx_->check(mh_photo_factor_wait(f_, &r), "mh_photo_factor_wait");
    return toHessian(r);
  }

  std::vector<RejectStatus> getStatuses() const  // :49
  {
    const size_t n = mh_photo_factor_size(f_);
    std::vector<int32_t> s(n);
    if (n) ctx_->check(mh_photo_factor_get_state(f_, s.data(), nullptr, nullptr), "mh_photo_factor_get_state");
    std::vector<RejectStatus> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<RejectStatus>(s[i]);
    return out;
  }
  std::vector<std::array<double, 2>> getCenters() const  // getFeatures()[i].center after the last linearize (:50)
  {
    const size_t n = mh_photo_factor_size(f_);
    std::vector<std::array<double, 2>> c(n);
    if (n) ctx_->check(mh_photo_factor_get_state(f_, nullptr, c[0].data(), nullptr), "mh_photo_factor_get_state");
    return c;
  }
  void getLocalizabilities(V3D & trans_final, V3D & rot_final, M33 & eigenvectors_trans, M33 & eigenvectors_rot) const  // :51-59
  {
    trans_final = vector3(last_.loc_trans_final);
    rot_final = vector3(last_.loc_rot_final);
    eigenvectors_trans = matrix3(last_.eigvec_trans);
    eigenvectors_rot = matrix3(last_.eigvec_rot);
  }
  const mh_photo_result & lastResult() const { return last_; }
  mh_photo_factor * underlying() const { return f_; }

private:
  friend class Photometric;
  std::shared_ptr<HessianFactor> toHessian(const mh_photo_result & r) const
  {
    if (r.n_exceptions > 0)  // project(): "invalid x coordinate" (photometric_utils.cpp:90-97) / interpolated_map_T_Le_Lt.at()
      throw std::runtime_error("PhotometricFactor::linearize: " + std::to_string(r.n_exceptions) +
                               " feature(s) hit a condition the reference throws on");
    last_ = r;
    gtsam::Vector gb(6);
    for (int i = 0; i < 6; ++i) gb(i) = -r.b_b[i];
    if (!is_binary_) return std::make_shared<HessianFactor>(keys()[0], matrix6(r.H_bb), gb, r.f);  // HessianFactor(key, H_bb, -b_b, f), :332-353
    gtsam::Vector ga(6);
    for (int i = 0; i < 6; ++i) ga(i) = -r.b_a[i];
    return std::make_shared<HessianFactor>(keys()[0], keys()[1], matrix6(r.H_bb), matrix6(r.H_ba), gb, matrix6(r.H_aa), ga, r.f);
  }
  PhotometricFactor(std::shared_ptr<Context> ctx, const KeyVector & keys, bool is_binary)
  : NonlinearFactor(keys), ctx_(std::move(ctx)), is_binary_(is_binary)
  {
    std::memset(&last_, 0, sizeof(last_));
  }
  std::shared_ptr<Context> ctx_;
  const bool is_binary_;
  mh_photo_factor * f_ = nullptr;
  mutable mh_photo_result last_;
};

// mimosa_msgs/msg/LidarPhotometricDebug.msg counterpart (plain struct)
struct PhotometricDebug
{
  size_t n_features_tracked = 0, n_features_in_factor = 0;
  int n_status[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};

class Photometric
{
public:
  const PhotometricConfig config;

  Photometric(const std::shared_ptr<Context> & ctx, const PhotometricConfig & cfg) : config(cfg), ctx_(ctx)  // photometric.cpp:13-70
  {
    if (!config.enabled) return;
    if (config.brightness_window_size.size() != 2) throw std::runtime_error("Photometric: brightness_window_size needs {width, height}");
    if (config.beam_altitude_angles.size() != config.rows || (config.destagger && config.pixel_shift_by_row.size() != config.rows))
      throw std::runtime_error("Photometric: beam_altitude_angles / pixel_shift_by_row need one entry per row");
    std::vector<int32_t> shift(config.pixel_shift_by_row.begin(), config.pixel_shift_by_row.end()), offsets;
    if (shift.empty()) shift.assign(config.rows, 0);
    for (const auto & o : config.edgelet_patch_offsets) {
      offsets.push_back(o.first);
      offsets.push_back(o.second);
    }
    mh_photo_config c{};
    c.rows = static_cast<int32_t>(config.rows);
    c.cols = static_cast<int32_t>(config.cols);
    c.destagger = config.destagger ? 1 : 0;
    c.pixel_shift_by_row = shift.data();
    c.beam_altitude_angles = config.beam_altitude_angles.data();
    c.range_min = config.range_min;
    c.range_max = config.range_max;
    c.erosion_buffer = config.erosion_buffer;
    c.patch_size = config.patch_size;
    c.margin_size = config.margin_size;
    c.intensity_scale = config.intensity_scale;
    c.intensity_gamma = config.intensity_gamma;
    c.remove_lines = config.remove_lines ? 1 : 0;
    c.filter_brightness = config.filter_brightness ? 1 : 0;
    c.gaussian_blur = config.gaussian_blur ? 1 : 0;
    c.gaussian_blur_size = config.gaussian_blur_size;
    c.gradient_threshold = config.gradient_threshold;
    c.max_dist_from_mean = config.max_dist_from_mean;
    c.max_dist_from_plane = config.max_dist_from_plane;
    c.nma_radius = config.nma_radius;
    c.num_features_detect = static_cast<int32_t>(config.num_features_detect);
    c.occlusion_range_diff_threshold = config.occlusion_range_diff_threshold;
    c.max_feature_life_time = config.max_feature_life_time;
    c.high_pass_fir = config.high_pass_fir.data();
    c.n_high_pass = static_cast<int32_t>(config.high_pass_fir.size());
    c.low_pass_fir = config.low_pass_fir.data();
    c.n_low_pass = static_cast<int32_t>(config.low_pass_fir.size());
    c.brightness_window_size[0] = config.brightness_window_size[0];
    c.brightness_window_size[1] = config.brightness_window_size[1];
    c.lidar_origin_to_beam_origin_mm = config.lidar_origin_to_beam_origin_mm;
    c.rotate_patch_to_align_with_gradient = config.rotate_patch_to_align_with_gradient ? 1 : 0;
    c.patch_offsets = offsets.data();
    c.n_patch_offsets = static_cast<int32_t>(config.edgelet_patch_offsets.size());
    c.use_robust_cost_function = config.use_robust_cost_function ? 1 : 0;
    if (config.robust_cost_function == "huber")
      c.robust_cost_function = 0;
    else if (config.robust_cost_function == "gemanmcclure")
      c.robust_cost_function = 1;
    else
      throw std::runtime_error("Photometric: unknown robust_cost_function " + config.robust_cost_function);
    c.robust_cost_function_parameter = config.robust_cost_function_parameter;
    c.error_scale = config.error_scale;
    c.max_error = config.max_error;
    c.sigma = config.sigma;
    const PoseRM TBL = rowMajor(config.T_B_L);
    std::memcpy(c.T_B_L_R, TBL.R.data(), sizeof(c.T_B_L_R));
    std::memcpy(c.T_B_L_t, TBL.t.data(), sizeof(c.T_B_L_t));
    c.static_mask = config.static_mask.empty() ? nullptr : config.static_mask.data();
    ctx_->check(mh_photo_create(ctx_->get(), &c, &photo_), "mh_photo_create");  // copies every table
  }
  ~Photometric() { mh_photo_destroy(photo_); }
  Photometric(const Photometric &) = delete;
  Photometric & operator=(const Photometric &) = delete;

  // photometric.cpp:92-320.  interpolated_map_T_Le_Lt: (timestamp ns, T_Le_Lt) ascending — the flat_map Manager::deskewPoints
  // fills (lidar/manager.cpp:390-405, :501-503).  Corrected intensities are written back into points_deskewed (:307-314).
  void preprocess(const PointCloud & points_raw, PointCloud & points_deskewed,
                  const std::vector<std::pair<uint32_t, Pose3>> & interpolated_map_T_Le_Lt, const double ts, const Key key)
  {
    if (!config.enabled) return;
    if (points_raw.size() != points_deskewed.size()) throw std::runtime_error("Photometric::preprocess: raw and deskewed clouds differ in size");
    std::vector<uint32_t> ns;
    std::vector<double> T;
    flatten(interpolated_map_T_Le_Lt, ns, T);
    ctx_->check(mh_photo_preprocess(photo_, points_raw.data(), points_deskewed.data(), points_raw.size(), ns.data(), T.data(), ns.size()),
                "mh_photo_preprocess");
    ts_ = ts;
    key_ = key;
  }
  // The same on a device-resident scan (ScanFrontEnd::keepRaw(true) before deskewPoints): no upload, the corrected
  // intensities stay in the scan's points_full_.
  void preprocess(ScanFrontEnd & scan, const std::vector<Pose3> & T_Le_Lt, const double ts, const Key key)
  {
    if (!config.enabled) return;
    std::vector<double> T(12 * T_Le_Lt.size());
    for (size_t g = 0; g < T_Le_Lt.size(); ++g) {
      const PoseRM p = rowMajor(T_Le_Lt[g]);
      std::memcpy(&T[12 * g], p.R.data(), 72);
      std::memcpy(&T[12 * g + 9], p.t.data(), 24);
    }
    ctx_->check(mh_photo_preprocess_scan(photo_, scan.underlying(), T.data(), T_Le_Lt.size()), "mh_photo_preprocess_scan");
    ts_ = ts;
    key_ = key;
  }
  // The same in two steps for a pipelined caller (replay.hpp): preprocessBegin builds the frame while another thread may
  // still run updateMap of the previous scan on this object; preprocessCommit (after that update has returned) makes it current.
  void preprocessBegin(ScanFrontEnd & scan, const std::vector<Pose3> & T_Le_Lt)
  {
    if (!config.enabled) return;
    std::vector<double> T(12 * T_Le_Lt.size());
    for (size_t g = 0; g < T_Le_Lt.size(); ++g) {
      const PoseRM p = rowMajor(T_Le_Lt[g]);
      std::memcpy(&T[12 * g], p.R.data(), 72);
      std::memcpy(&T[12 * g + 9], p.t.data(), 24);
    }
    ctx_->check(mh_photo_preprocess_scan_begin(photo_, scan.underlying(), T.data(), T_Le_Lt.size()), "mh_photo_preprocess_scan_begin");
  }
  void preprocessBegin(ScanFrontEnd & scan, const double * Rt12, const size_t n)  // n x {R row-major, t}, as the C ABI takes them
  {
    if (!config.enabled) return;
    ctx_->check(mh_photo_preprocess_scan_begin(photo_, scan.underlying(), Rt12, n), "mh_photo_preprocess_scan_begin");
  }
  void preprocessCommit(const double ts, const Key key)
  {
    if (!config.enabled) return;
    ctx_->check(mh_photo_preprocess_commit(photo_), "mh_photo_preprocess_commit");
    ts_ = ts;
    key_ = key;
  }
  // the frame-only part of the next updateMap (candidate pixels), enqueued now so that it runs beside the optimisation
  void detectPrefetch()
  {
    if (!config.enabled) return;
    ctx_->check(mh_photo_detect_prefetch(photo_), "mh_photo_detect_prefetch");
  }

  // photometric.cpp:373-394: unary factor on the current frame from the tracked features; V S V^T restricts it to the
  // directions `selection` keeps (the geometric factor's degenerate ones, lidar/manager.cpp:568-581)
  void getFactors(const Values &, NonlinearFactorGraph & graph, const M66 & eigenvectors_block_matrix = identity66(),
                  const V6D & selection = ones6())
  {
    if (!config.enabled) return;
    size_t nf = 0, np = 0;
    ctx_->check(mh_photo_num_features(photo_, &nf, &np), "mh_photo_num_features");
    debug_.n_features_tracked = nf;
    if (!nf) return;
    double VSVt[36];  // row-major for the C ABI
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += eigenvectors_block_matrix(i, k) * selection(k) * eigenvectors_block_matrix(j, k);
        VSVt[6 * i + j] = s;
      }
    photometric_factor_.reset(new PhotometricFactor(ctx_, KeyVector{key_}, false));
    ctx_->check(mh_photo_factor_create(photo_, VSVt, 0, &photometric_factor_->f_), "mh_photo_factor_create");
    debug_.n_features_in_factor = nf;
    graph.add(photometric_factor_);
  }

  // photometric.cpp:396-514: bookkeeping from the factor's statuses, then detectFeatures for the missing ones
  void updateMap(const Values & values, const std::vector<V3D> & bias_directions = {})
  {
    if (!config.enabled) return;
    const PoseRM T_W_Be = rowMajor(values.at<Pose3>(key_));
    std::vector<double> bias(3 * bias_directions.size());
    for (size_t i = 0; i < bias_directions.size(); ++i)
      for (int k = 0; k < 3; ++k) bias[3 * i + k] = bias_directions[i](k);
    if (photometric_factor_)
      for (int i = 0; i < 9; ++i) debug_.n_status[i] = photometric_factor_->lastResult().status_hist[i];
    ctx_->check(mh_photo_update_map(photo_, photometric_factor_ ? photometric_factor_->underlying() : nullptr, T_W_Be.R.data(),
                                    T_W_Be.t.data(), bias.empty() ? nullptr : bias.data(), bias_directions.size()),
                "mh_photo_update_map");
    photometric_factor_.reset();  // the factor stays alive in the graph that holds it
  }

  std::vector<Feature> features() const  // map_Le_features_
  {
    size_t nf = 0, np = 0;
    ctx_->check(mh_photo_num_features(photo_, &nf, &np), "mh_photo_num_features");
    std::vector<mh_photo_feature> h(nf);
    std::vector<double> Le(3 * np), in(np), psi(np);
    if (nf) ctx_->check(mh_photo_get_features(photo_, h.data(), Le.data(), in.data(), psi.data()), "mh_photo_get_features");
    std::vector<Feature> out(nf);
    size_t o = 0;
    for (size_t i = 0; i < nf; ++i) {
      Feature & f = out[i];
      f.id = h[i].id;
      f.life_time = h[i].life_time;
      f.center = {h[i].center[0], h[i].center[1]};
      f.normal = V3D(h[i].normal[0], h[i].normal[1], h[i].normal[2]);
      f.mean_intensity = h[i].mean_intensity;
      f.sigma_intensity = h[i].sigma_intensity;
      for (int k = 0; k < h[i].n_points; ++k, ++o) {
        f.Le_ps.push_back(V3D(Le[3 * o], Le[3 * o + 1], Le[3 * o + 2]));
        f.intensities.push_back(in[o]);
        f.psi_intensities.push_back(psi[o]);
      }
    }
    return out;
  }
  // which: see mh_photo_get_image (0 img_intensity ... 4 img_mask ...); T must match the plane's element type
  template <typename T>
  std::vector<T> image(int which) const
  {
    std::vector<T> out(config.rows * config.cols * (which == 7 ? 10 : 1));
    ctx_->check(mh_photo_get_image(photo_, which, out.data(), out.size() * sizeof(T)), "mh_photo_get_image");
    return out;
  }
  const PhotometricDebug & debug() const { return debug_; }
  const PhotometricFactor::Ptr & factor() const { return photometric_factor_; }
  mh_photo * underlying() { return photo_; }

  static M66 identity66() { return M66::Identity(); }
  static V6D ones6()
  {
    V6D v;
    for (int i = 0; i < 6; ++i) v(i) = 1.0;
    return v;
  }

private:
  static void flatten(const std::vector<std::pair<uint32_t, Pose3>> & m, std::vector<uint32_t> & ns, std::vector<double> & T)
  {
    ns.resize(m.size());
    T.resize(12 * m.size());
    for (size_t g = 0; g < m.size(); ++g) {
      ns[g] = m[g].first;
      const PoseRM p = rowMajor(m[g].second);
      std::memcpy(&T[12 * g], p.R.data(), 72);
      std::memcpy(&T[12 * g + 9], p.t.data(), 24);
    }
  }
  std::shared_ptr<Context> ctx_;
  mh_photo * photo_ = nullptr;
  PhotometricFactor::Ptr photometric_factor_;
  double ts_ = 0;
  Key key_ = 0;
  PhotometricDebug debug_;
};

}  // namespace lidar
}  // namespace mimosa_hip
