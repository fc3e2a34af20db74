// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement (C++17, zero dependencies) of the LiDAR PHOTOMETRIC path of ntnu-arl/mimosa
// (BASELINE configs[3]): Ouster intensity-image formation, the image filter
// chain, feature detection and the NCC patch factor with its projection Jacobians.
//
// PARITY UNPINNED.  The reference has no tests and cannot be compiled here (OpenCV, Eigen, GTSAM, PCL,
// ROS absent).  Three third-party dependencies hold arithmetic of this path and are NOT under
// /root/reference; they are restated from their published behaviour, and every assumption is listed
// here (the photometric counterpart of SURVEY.md Appendix B):
//
//  OpenCV (system, Ubuntu 20.04 -> 4.2) — assumptions
//   O1  Every filter below uses BORDER_DEFAULT = BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), anchor at the
//       kernel centre: filter2D, blur, GaussianBlur, Sobel are called without border / anchor arguments
//       (photometric.cpp:283-288, 322-345).
//   O2  filter2D on CV_32F data with a 33-tap kernel (area < 50) takes the direct path: float accumulation
//       over the kernel taps in row-major tap order starting from 0, as CORRELATION (no kernel flip), no FMA.
//   O3  blur(): normalised box filter; for CV_32F sources the window sums are accumulated in double and the
//       product with 1 / (w * h) (double) is rounded to float once.  (OpenCV keeps sliding running sums; a
//       direct window sum differs from that by < 1e-15 relative before the final float rounding.)
//   O4  GaussianBlur(ksize 3, sigma 0) uses the fixed kernel {0.25, 0.5, 0.25} separably, rows then columns,
//       float arithmetic  s = k0 * c + k1 * (l + r).
//   O5  Sobel(dx or dy = 1, ksize = 1, scale 0.5) is the 3 x 1 central difference (I[+1] - I[-1]) * 0.5; with
//       REFLECT_101 the derivative at the image border is exactly 0.
//   O6  erode() with a ones kernel, default border = constant +inf: out-of-image pixels never lower the minimum.
//   O7  Mat *= s, Mat += s, s * A / B on CV_32F are evaluated in float per element: a * float(s);
//       a + float(s); (a * float(s)) / b  (cv::divide, 0 where b == 0).
//   O8  threshold(THRESH_TRUNC, 255) = min(x, 255).
//   O9  convertScaleAbs = saturate_cast<uchar>(|x|) and addWeighted(a, .5, b, .5) on uchar =
//       saturate_cast<uchar>(float(a) * .5f + float(b) * .5f), both with round-half-to-even (cvRound).
//   O10 circle(..., thickness -1) fills the midpoint-circle spans of modules/imgproc/src/drawing.cpp Circle().
//   O11 cornerEigenValsAndVecs(roi 7 x 7, blockSize 5, ksize 3): Sobel 3 x 3 with scale 1 / (4 * 5), products
//       dx*dx, dx*dy, dy*dy in float, un-normalised 5 x 5 box sum (float accumulation), then calcEigenValsVecs'
//       closed form in double; only the centre pixel is read, so no border is involved.
//   O12 cv::pow (intensity_gamma != 1) is OpenCV's own log/exp approximation and is NOT restated: every shipped
//       configuration sets intensity_gamma: 1 (e.g. config/enwide/params.yaml:112) which skips the call; a gamma
//       != 1 is evaluated with std::pow(float) here (documented deviation).
//  Eigen 3.3.7 — VectorXd::mean / norm / squaredNorm and the small dynamic products are evaluated as plain
//       left-to-right double loops (Eigen vectorises them with a different association; differences ~1e-16
//       relative, far inside the 1e-5 bar); SelfAdjointEigenSolver<M3D> as in ref_cpu.hpp; M66::inverse() by
//       Gauss-Jordan with partial pivoting (Eigen: PartialPivLU).
//  libstdc++ std::sort — detectFeatures sorts (gradient, pixel) pairs with a comparator on the gradient only
//       (photometric.cpp:556-560); the order of equal gradients is whatever std::sort does.  This oracle and the
//       product's host-side detection both call the same std::sort of the same libstdc++ on the same sequence.
//  atan2 in the yaw table (photometric.cpp:127) is evaluated in double and rounded to float.
//  PCL — DEG2RAD(x) = x * 0.017453293 and RAD2DEG(x) = x * 57.29578 (pcl/pcl_macros.h; the reference defines neither).
//
// Every function cites the reference file:line it follows (paths relative to /root/reference/mimosa/).
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <utility>
#include <set>
#include <vector>

#include "ref_cpu.hpp"

namespace refphoto
{
using refcpu::M3;
using refcpu::Point32;
using refcpu::Pose;
using refcpu::V3;

constexpr int kDuplicatePoints = 10;  // include/mimosa/lidar/photometric_utils.hpp:17

// include/mimosa/lidar/photometric_config.hpp:15-87 (the fields the arithmetic reads) + the derived
// parameters of src/lidar/photometric_config.cpp:98-110.
struct PhotoConfig
{
  int rows = 128, cols = 1024;
  int destagger = 1;
  std::vector<int> pixel_shift_by_row;
  std::vector<float> beam_altitude_angles;  // degrees, descending
  float range_min = 0.1f, range_max = 100.f;
  int erosion_buffer = 2, patch_size = 5, margin_size = 2;
  float intensity_scale = 0.25f, intensity_gamma = 0.8f;
  int remove_lines = 1, filter_brightness = 1, gaussian_blur = 1, gaussian_blur_size = 3;
  float gradient_threshold = 20.f, max_dist_from_mean = 0.2f, max_dist_from_plane = 0.1f;
  int nma_radius = 10;
  bool rotate_patch_to_align_with_gradient = false;  // photometric_config.hpp:61
  int num_features_detect = 60;
  float occlusion_range_diff_threshold = 0.1f;
  int max_feature_life_time = 30;
  std::vector<double> high_pass_fir, low_pass_fir;
  int brightness_window_w = 1, brightness_window_h = 1;  // cv::Size(width, height) = brightness_window_size[0], [1]
  float lidar_origin_to_beam_origin_mm = 0.f;
  std::vector<std::pair<int, int>> patch_offsets;  // edgelet_patch_offsets
  int use_robust_cost_function = 1;
  int robust_is_huber = 1;  // else gemanmcclure
  double robust_cost_function_parameter = 1.345, error_scale = 1.0, max_error = 255.0, sigma = 0.1;
  Pose T_B_L;
  std::vector<uint8_t> static_mask;  // rows * cols, 0 = invalid; empty = none (static_mask_path == "")
  // derived (photometric_config.cpp:98-110)
  double fx = 1, fy = 1, cx = 0;
  float beam_offset_m = 0;
  void derive();
};
inline void PhotoConfig::derive()
{
  {
    fx = -static_cast<float>(cols) / (2 * M_PI);
    cx = static_cast<float>(cols) / 2.0;
    const double span_deg = beam_altitude_angles.front() - beam_altitude_angles.back();  // float difference promoted
    fy = -static_cast<float>(rows) / std::fabs(static_cast<double>(static_cast<float>(span_deg)) * 0.017453293);  // DEG2RAD
    beam_offset_m = static_cast<float>(lidar_origin_to_beam_origin_mm / 1000.0);
  }
}

// DEG2RAD / RAD2DEG are not defined anywhere in the reference: they are PCL's macros (pcl/pcl_macros.h,
// reached through the PCL headers of lidar/point.hpp): ((x)*0.017453293) and ((x)*57.29578) — truncated constants.
inline double deg2rad(double d) { return d * 0.017453293; }
inline double rad2deg(double r) { return r * 57.29578; }

// include/mimosa/lidar/photometric_utils.hpp:42-92
struct Frame
{
  int rows = 0, cols = 0;
  std::vector<Point32> points_deskewed;
  std::vector<float> img_intensity, img_range, img_dx, img_dy;
  std::vector<int32_t> img_idx;    // img_deskewed_cloud_idx
  std::vector<int32_t> proj_idx;   // rows * cols * kDuplicatePoints
  std::vector<uint8_t> img_mask;
  std::vector<float> yaw;          // yaw_angles[v][u]
  std::vector<uint32_t> pose_ns;   // interpolated_map_T_Le_Lt keys (ascending) ...
  std::vector<Pose> pose_T;        // ... and values
  const Pose * pose_at(uint32_t ns) const
  {
    const auto it = std::lower_bound(pose_ns.begin(), pose_ns.end(), ns);
    if (it == pose_ns.end() || *it != ns) return nullptr;
    return &pose_T[static_cast<size_t>(it - pose_ns.begin())];
  }
};

// src/lidar/photometric.cpp:72-90 — raw point index -> destaggered pixel
inline void idx_to_pixel_maps(const PhotoConfig & c, std::vector<int> & idx_to_u, std::vector<int> & idx_to_v)
{
  idx_to_u.assign(static_cast<size_t>(c.cols) * c.rows, -1);
  idx_to_v.assign(static_cast<size_t>(c.cols) * c.rows, -1);
  for (int v = 0; v < c.rows; ++v)
    for (int u = 0; u < c.cols; ++u) {
      const int uu = (u + c.cols - c.pixel_shift_by_row[v]) % c.cols;
      const int idx = v * c.cols + (c.destagger ? uu : u);
      idx_to_u[idx] = u;
      idx_to_v[idx] = v;
    }
}

inline int reflect101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

inline bool in_fov(double x, double y, int rows, int cols) { return x >= 0 && x <= cols - 1 && y >= 0 && y <= rows - 1; }

struct ProjectThrow : std::runtime_error
{
  using std::runtime_error::runtime_error;
};

// src/lidar/photometric_utils.cpp:80-198 — project() with the per-row yaw table
inline bool project(const V3 & p, double uv[2], const std::vector<float> & yaw, const PhotoConfig & c)
{
  const double L = std::sqrt(p.x * p.x + p.y * p.y) - c.beam_offset_m;
  const double R = std::sqrt(L * L + p.z * p.z);
  const double phi = std::atan2(p.y, p.x);
  const double theta = std::asin(p.z / R);
  uv[0] = c.fx * phi + c.cx;
  if (uv[0] < 0 || uv[0] >= c.cols) throw ProjectThrow("Invalid x coordinate");  // :90-97
  if (uv[0] < 5 || uv[0] > c.cols - 5) return false;                           // :100
  if (theta > deg2rad(c.beam_altitude_angles.front()) || theta < deg2rad(c.beam_altitude_angles.back())) return false;  // :104-107
  // :109-121 — upper_bound over the reversed (ascending) altitude table, then `+ 1` and .base()
  const auto & alt = c.beam_altitude_angles;
  const double th_deg = rad2deg(theta);
  const auto rit = std::upper_bound(alt.rbegin(), alt.rend(), th_deg, [](double v, float a) { return v < a; });
  // DEG2RAD * RAD2DEG = 1 + 1e-8, so an angle that passed the radian bounds can lie (by < 1e-8 relative) outside the
  // table in degrees: the reference then dereferences one element before / after the table (undefined behaviour).
  // Defined here, and identically in the product: the table index is clamped to [0, rows - 2].
  auto greater = rit == alt.rend() ? alt.begin() : (rit + 1).base();
  if (greater + 1 == alt.end()) greater = alt.end() - 2;
  const auto smaller = greater + 1;
  uv[1] = static_cast<double>(greater - alt.begin());
  uv[1] += (*greater - th_deg) / (*greater - *smaller);  // float difference in the denominator, double quotient
  // :124-147 — binary search in the yaw row (descending) around the analytic column
  const int approx_y = static_cast<int>(std::round(uv[1]));
  const float * row = &yaw[static_cast<size_t>(approx_y) * c.cols];
  int il = static_cast<int>(uv[0]) - 5, ir = static_cast<int>(uv[0]) + 5;
  while (ir - il > 1) {
    const int mid = il + (ir - il) / 2;
    if (row[mid] == phi) {
      uv[0] = mid;
      return in_fov(uv[0], uv[1], c.rows, c.cols);
    } else if (row[mid] < phi) {
      ir = mid;
    } else {
      il = mid;
    }
  }
  double new_x = il;
  new_x += (row[il] - phi) / (row[il] - row[ir]);  // (float - double) / (float - float)
  uv[0] = new_x;
  return in_fov(uv[0], uv[1], c.rows, c.cols);
}

// src/lidar/photometric_utils.cpp:186-198
inline void projection_jacobian(const V3 & p, const PhotoConfig & c, double H[6])
{
  const double rxy = std::sqrt(p.x * p.x + p.y * p.y);
  const double L = rxy - c.beam_offset_m;
  const double R2 = L * L + p.z * p.z;
  const double irxy = 1.0 / rxy;
  const double irxy2 = irxy * irxy;
  const double fx_irxy2 = c.fx * irxy2;
  H[0] = -fx_irxy2 * p.y;
  H[1] = fx_irxy2 * p.x;
  H[2] = 0;
  H[3] = -c.fy * p.x * p.z / ((L + c.beam_offset_m) * R2);
  H[4] = -c.fy * p.y * p.z / ((L + c.beam_offset_m) * R2);
  H[5] = c.fy * L / R2;
}

inline int vector_index(int row, int col, const PhotoConfig & c) { return (row * c.cols + col) * kDuplicatePoints; }  // :200-203

// src/lidar/photometric_utils.cpp:287-366 — projectUndistorted (yaw-table overload).  Returns 0 ok, 1 = false,
// throws ProjectThrow where the reference throws.
inline bool project_undistorted(const V3 & Le_p, V3 & Li_p, double Li_uv[2], Pose & T_Le_Lt, const Frame & f, const PhotoConfig & c)
{
  double Lk_uv[2];
  if (!project(Le_p, Lk_uv, f.yaw, c)) return false;
  Lk_uv[0] = std::round(Lk_uv[0]);
  Lk_uv[1] = std::round(Lk_uv[1]);
  int distortion_idx = -1;
  size_t row = static_cast<size_t>(Lk_uv[1]);
  const size_t col = static_cast<size_t>(Lk_uv[0]);
  int idx = vector_index(static_cast<int>(row), static_cast<int>(col), c);
  if (f.proj_idx[idx] == 0) {
    row = 0;
    for (; row < static_cast<size_t>(c.rows); row++) {
      idx = vector_index(static_cast<int>(row), static_cast<int>(col), c);
      if (f.proj_idx[idx] > 0) break;
    }
    if (row >= static_cast<size_t>(c.rows)) return false;
  }
  if (f.proj_idx[idx] > 1) {
    float min_sq = std::numeric_limits<float>::max();
    for (int i = 1; i <= f.proj_idx[idx]; i++) {
      const int j = f.proj_idx[idx + i];
      const Point32 & q = f.points_deskewed[j];
      const double dx = Le_p.x - q.x, dy = Le_p.y - q.y, dz = Le_p.z - q.z;
      const float sq = static_cast<float>(dx * dx + (dy * dy + dz * dz));
      if (sq < min_sq) {
        min_sq = sq;
        distortion_idx = j;
      }
    }
  } else {
    distortion_idx = f.proj_idx[idx + 1];
  }
  if (distortion_idx < 0) return false;
  const Pose * T = f.pose_at(f.points_deskewed[distortion_idx].t);
  if (!T) throw ProjectThrow("interpolated_map_T_Le_Lt.at(): out_of_range");
  T_Le_Lt = *T;
  Li_p = T_Le_Lt.inverse() * Le_p;
  return project(Li_p, Li_uv, f.yaw, c);
}

// src/lidar/photometric_utils.cpp:368-388 — bilinear sample of a CV_32F image
inline double sub_pixel(const std::vector<float> & img, int rows, int cols, double x, double y)
{
  if (!in_fov(x, y, rows, cols)) throw ProjectThrow("getSubPixelValue: not in the field of view");
  const int x0 = static_cast<int>(std::floor(x)), x1 = x0 + 1, y0 = static_cast<int>(std::floor(y)), y1 = y0 + 1;
  const double dx = x - x0, dy = y - y0;
  auto at = [&](int yy, int xx) { return static_cast<double>(img[static_cast<size_t>(yy) * cols + xx]); };
  return (1 - dx) * (1 - dy) * at(y0, x0) + dx * (1 - dy) * at(y0, x1) + (1 - dx) * dy * at(y1, x0) + dx * dy * at(y1, x1);
}

// src/lidar/photometric_utils.cpp:13-19
inline void get_psi(const std::vector<double> & I, double & mean, double & sigma, std::vector<double> & psi)
{
  double s = 0;
  for (double v : I) s += v;
  mean = s / static_cast<double>(I.size());
  double ss = 0;
  for (double v : I) ss += (v - mean) * (v - mean);
  sigma = std::sqrt(ss);
  psi.resize(I.size());
  for (size_t i = 0; i < I.size(); ++i) psi[i] = (I[i] - mean) / sigma;
}

// ----------------------------------------------------------------------------------------------
// Image filter chain (src/lidar/photometric.cpp:246-345)
// ----------------------------------------------------------------------------------------------
inline void correlate_1d(const std::vector<float> & src, std::vector<float> & dst, int rows, int cols, const std::vector<double> & taps,
                         bool vertical)
{
  // O2: cv::Mat(std::vector<double>) keeps CV_64F coefficients; filter2D converts the kernel to the float
  // working type for CV_32F data; float accumulation in tap order.
  const int n = static_cast<int>(taps.size()), a = n / 2;
  std::vector<float> kf(taps.size());
  for (size_t i = 0; i < taps.size(); ++i) kf[i] = static_cast<float>(taps[i]);
  dst.assign(src.size(), 0.f);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float s = 0.f;
      for (int k = 0; k < n; ++k) {
        const int yy = vertical ? reflect101(y + k - a, rows) : y, xx = vertical ? x : reflect101(x + k - a, cols);
        s = s + kf[k] * src[static_cast<size_t>(yy) * cols + xx];
      }
      dst[static_cast<size_t>(y) * cols + x] = s;
    }
}

inline void remove_lines(std::vector<float> & img, int rows, int cols, const PhotoConfig & c)  // :322-337
{
  std::vector<float> hp, lines;
  correlate_1d(img, hp, rows, cols, c.high_pass_fir, true);    // high_pass_fir is a column kernel (N x 1)
  correlate_1d(hp, lines, rows, cols, c.low_pass_fir, false);  // low_pass_fir.t() is a row kernel (1 x N)
  for (size_t i = 0; i < img.size(); ++i) {
    img[i] = img[i] - lines[i];
    if (img[i] < 0) img[i] = 0.f;
  }
}

inline void filter_brightness(std::vector<float> & img, int rows, int cols, const PhotoConfig & c)  // :339-347
{
  const int w = c.brightness_window_w, h = c.brightness_window_h, ax = w / 2, ay = h / 2;
  const double scale = 1.0 / (static_cast<double>(w) * h);
  std::vector<float> out(img.size());
  // O3: row sums then column sums, both in double
  std::vector<double> rs(img.size());
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      double s = 0;
      for (int k = 0; k < w; ++k) s += img[static_cast<size_t>(y) * cols + reflect101(x + k - ax, cols)];
      rs[static_cast<size_t>(y) * cols + x] = s;
    }
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      double s = 0;
      for (int k = 0; k < h; ++k) s += rs[static_cast<size_t>(reflect101(y + k - ay, rows)) * cols + x];
      const float b = static_cast<float>(s * scale) + 1.0f;  // brightness += 1 (O7)
      const float v = img[static_cast<size_t>(y) * cols + x];
      out[static_cast<size_t>(y) * cols + x] = b != 0.f ? (v * 140.0f) / b : 0.f;
    }
  img.swap(out);
}

inline void gaussian3(std::vector<float> & img, int rows, int cols)  // :350-353, O4
{
  std::vector<float> tmp(img.size());
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const float l = img[static_cast<size_t>(y) * cols + reflect101(x - 1, cols)], r = img[static_cast<size_t>(y) * cols + reflect101(x + 1, cols)];
      tmp[static_cast<size_t>(y) * cols + x] = 0.5f * img[static_cast<size_t>(y) * cols + x] + 0.25f * (l + r);
    }
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const float u = tmp[static_cast<size_t>(reflect101(y - 1, rows)) * cols + x], d = tmp[static_cast<size_t>(reflect101(y + 1, rows)) * cols + x];
      img[static_cast<size_t>(y) * cols + x] = 0.5f * tmp[static_cast<size_t>(y) * cols + x] + 0.25f * (u + d);
    }
}

inline void erode_ones(std::vector<uint8_t> & m, int rows, int cols, int k)  // O6; anchor k / 2
{
  std::vector<uint8_t> out(m.size());
  const int a = k / 2;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      uint8_t mn = 255;
      for (int dy = -a; dy < k - a; ++dy)
        for (int dx = -a; dx < k - a; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
          mn = std::min(mn, m[static_cast<size_t>(yy) * cols + xx]);
        }
      out[static_cast<size_t>(y) * cols + x] = mn;
    }
  m.swap(out);
}

// ----------------------------------------------------------------------------------------------
// Photometric::preprocess (src/lidar/photometric.cpp:92-320)
// ----------------------------------------------------------------------------------------------
inline void preprocess(const PhotoConfig & c, const Point32 * raw, Point32 * deskewed, size_t n, const uint32_t * pose_ns,
                       const double * pose_Rt12, size_t n_poses, Frame & f)
{
  const int rows = c.rows, cols = c.cols;
  const size_t npx = static_cast<size_t>(rows) * cols;
  if (n > npx) throw std::runtime_error("Number of points exceeds the image size");
  f.rows = rows;
  f.cols = cols;
  f.points_deskewed.assign(deskewed, deskewed + n);
  f.img_intensity.assign(npx, 0.f);
  f.img_range.assign(npx, 0.f);
  f.img_idx.assign(npx, -1);
  f.proj_idx.assign(npx * kDuplicatePoints, 0);
  f.img_mask.assign(npx, 0);
  f.img_dx.assign(npx, 0.f);
  f.img_dy.assign(npx, 0.f);
  f.yaw.assign(npx, std::numeric_limits<float>::quiet_NaN());
  f.pose_ns.assign(pose_ns, pose_ns + n_poses);
  f.pose_T.resize(n_poses);
  for (size_t g = 0; g < n_poses; ++g) {
    std::memcpy(f.pose_T[g].R.m, pose_Rt12 + 12 * g, 9 * sizeof(double));
    f.pose_T[g].t = {pose_Rt12[12 * g + 9], pose_Rt12[12 * g + 10], pose_Rt12[12 * g + 11]};
  }
  std::vector<int> idx_to_u, idx_to_v;
  idx_to_pixel_maps(c, idx_to_u, idx_to_v);
  std::vector<uint8_t> yaw_valid(npx, 0);
  // :121-130
  for (size_t i = 0; i < n; ++i) {
    const Point32 & p = raw[i];
    const int u = idx_to_u[p.idx], v = idx_to_v[p.idx];
    f.yaw[static_cast<size_t>(v) * cols + u] = static_cast<float>(std::atan2(static_cast<double>(p.y), static_cast<double>(p.x)));
    yaw_valid[static_cast<size_t>(v) * cols + u] = 1;
  }
  // :135-199 — per-row interpolation of the missing yaw angles
  for (int v = 0; v < rows; ++v) {
    float * yr = &f.yaw[static_cast<size_t>(v) * cols];
    std::vector<int> valid_cols;
    for (int u = 0; u < cols; ++u)
      if (yaw_valid[static_cast<size_t>(v) * cols + u]) valid_cols.push_back(u);
    if (valid_cols.empty()) {
      for (int u = 0; u < cols; ++u) {
        const float t = static_cast<float>(u) / static_cast<float>(cols - 1);
        yr[u] = static_cast<float>((1.0f - t) * M_PI + t * (-M_PI));
      }
      continue;
    }
    const int first = valid_cols.front();
    if (first > 0) {
      const float yaw_first = yr[first];
      for (int u = 0; u < first; ++u) {
        const float t = static_cast<float>(u) / static_cast<float>(first);
        yr[u] = static_cast<float>((1.0f - t) * M_PI + t * yaw_first);
      }
    }
    for (size_t i = 0; i + 1 < valid_cols.size(); ++i) {
      const int lc = valid_cols[i], rc = valid_cols[i + 1];
      if (rc - lc <= 1) continue;
      const float yl = yr[lc], yrr = yr[rc], denom = static_cast<float>(rc - lc);
      for (int fc = lc + 1; fc < rc; ++fc) {
        const float t = static_cast<float>(fc - lc) / denom;
        yr[fc] = yl + t * (yrr - yl);
      }
    }
    const int last = valid_cols.back();
    if (last < cols - 1) {
      const float yaw_last = yr[last];
      const int rightmost = cols - 1, gap = rightmost - last;
      for (int u = last + 1; u <= rightmost; ++u) {
        const float t = static_cast<float>(u - last) / static_cast<float>(gap);
        yr[u] = static_cast<float>((1.0f - t) * yaw_last + t * (-M_PI));
      }
    }
  }
  // :204-230 — fill the images, project every deskewed point
  std::vector<int> uk(n, -1), vk(n, -1);
  for (size_t i = 0; i < n; ++i) {
    const Point32 & p = deskewed[i];
    if (p.range < c.range_min || p.range > c.range_max) continue;
    const int u = idx_to_u[p.idx], v = idx_to_v[p.idx];
    const size_t px = static_cast<size_t>(v) * cols + u;
    f.img_intensity[px] = p.intensity;
    f.img_range[px] = p.range;
    f.img_mask[px] = 1;
    f.img_idx[px] = static_cast<int32_t>(i);
    double uv[2];
    if (!project(V3{p.x, p.y, p.z}, uv, f.yaw, c)) continue;
    uk[i] = static_cast<int>(std::round(uv[0]));
    vk[i] = static_cast<int>(std::round(uv[1]));
  }
  // :232-244 — proj_idx: up to 9 point indices per pixel, in index order
  for (size_t i = 0; i < n; ++i) {
    const int u = uk[i], v = vk[i];
    if (u < 0 || v < 0) continue;
    const int start = vector_index(v, u, c);
    const int offset = f.proj_idx[start] + 1;
    if (offset >= kDuplicatePoints) continue;
    f.proj_idx[start + offset] = static_cast<int32_t>(i);
    f.proj_idx[start] = offset;
  }
  std::vector<float> & I = f.img_intensity;
  if (c.intensity_scale != 1.0) {  // :251-257 (float field promoted to double for the comparison)
    const float s = static_cast<float>(static_cast<double>(c.intensity_scale));
    for (float & v : I) v = v * s;
  }
  if (c.intensity_gamma != 1.0)  // :260-266, O12
    for (float & v : I) v = std::pow(v, c.intensity_gamma);
  if (c.remove_lines) remove_lines(I, rows, cols, c);
  if (c.filter_brightness) filter_brightness(I, rows, cols, c);
  if (c.gaussian_blur) gaussian3(I, rows, cols);  // gaussian_blur_size 3 in every shipped config
  for (float & v : I) v = v > 255.0f ? 255.0f : v;  // :298-300, O8
  // :307-314 — corrected intensities back into the cloud
  for (size_t px = 0; px < npx; ++px) {
    const int idx = f.img_idx[px];
    if (idx == -1) continue;
    deskewed[idx].intensity = I[px];  // the caller's cloud; the Frame keeps the copy taken at construction (:115-116)
  }
  // :316-317, O5
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t px = static_cast<size_t>(y) * cols + x;
      f.img_dx[px] = (I[static_cast<size_t>(y) * cols + reflect101(x + 1, cols)] - I[static_cast<size_t>(y) * cols + reflect101(x - 1, cols)]) * 0.5f;
      f.img_dy[px] = (I[static_cast<size_t>(reflect101(y + 1, rows)) * cols + x] - I[static_cast<size_t>(reflect101(y - 1, rows)) * cols + x]) * 0.5f;
    }
  // createMask :349-371
  if (!c.static_mask.empty())
    for (size_t px = 0; px < npx; ++px)
      if (c.static_mask[px] == 0) f.img_mask[px] = 0;
  erode_ones(f.img_mask, rows, cols, c.patch_size + c.erosion_buffer);
}

// ----------------------------------------------------------------------------------------------
// Features and the factor (include/mimosa/lidar/photometric_utils.hpp:25-40, photometric_factor.hpp)
// ----------------------------------------------------------------------------------------------
struct Feature
{
  uint32_t id = 0;
  int life_time = 0;
  double center[2] = {0, 0};
  std::vector<double> intensities;
  std::vector<V3> Le_ps;
  std::vector<double> psi;
  V3 normal;
  double mean_intensity = 0, sigma_intensity = 0;
};

enum PhotoStatus
{
  kUnprocessed = 0,
  kPointProjectUndistorted,
  kPointRange,
  kPointProject,
  kPointMask,
  kPointMaskMargin,
  kPointRangeDiff,
  kMaxError,
  kValid
};

struct PhotoResult
{
  double H_bb[36], H_ba[36], H_aa[36], b_b[6], b_a[6], f;
  double loc_trans_final[3], loc_rot_final[3], eigvec_trans[9], eigvec_rot[9];
  int32_t status_hist[9];
  int32_t n_exceptions;  // features where the reference would have thrown (project / .at()); status = PointProjectUndistorted
};

inline void mat6_mul(const double * A, const double * B, double * C)
{
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += A[6 * i + k] * B[6 * k + j];
      C[6 * i + j] = s;
    }
}
inline bool mat6_inv(const double * A, double * inv)
{
  double a[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      a[i][j] = A[6 * i + j];
      a[i][6 + j] = i == j ? 1.0 : 0.0;
    }
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
    if (p != c)
      for (int j = 0; j < 12; ++j) std::swap(a[p][j], a[c][j]);
    const double d = a[c][c];
    for (int j = 0; j < 12; ++j) a[c][j] /= d;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double m = a[r][c];
      for (int j = 0; j < 12; ++j) a[r][j] -= m * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) inv[6 * i + j] = a[i][6 + j];
  return true;
}

// PhotometricFactor::linearize (include/mimosa/lidar/photometric_factor.hpp:136-355).  T_b = Values[keys[0]],
// T_a = Values[keys[1]] for the binary form (NULL: identity).  VSVt (6x6, row-major) multiplies the unary Hessian.
// Per-feature outputs: status and (for Valid features) the new centre; rows (optional) = per-feature whitened
// residual vector and Jacobian J_final_b (m x 6), for the per-patch parity checks.
inline void linearize(const PhotoConfig & c, const Frame & fr, std::vector<Feature> & feats, const Pose & T_b, const Pose * T_a,
                      const double * VSVt, PhotoResult & out, std::vector<int32_t> & statuses,
                      std::vector<std::vector<double>> * e_rows = nullptr, std::vector<std::vector<double>> * J_rows = nullptr)
{
  const bool binary = T_a != nullptr;
  const Pose Ta = binary ? *T_a : Pose();
  const Pose d_Be = T_b.inverse() * Ta;                          // :147
  const Pose d_Le = c.T_B_L.inverse() * d_Be * c.T_B_L;          // :148-149
  statuses.assign(feats.size(), kUnprocessed);
  std::memset(&out, 0, sizeof(out));
  double Hbb[36] = {0}, Hba[36] = {0}, Haa[36] = {0}, bb[6] = {0}, ba[6] = {0}, fsum = 0;
  if (e_rows) e_rows->assign(feats.size(), {});
  if (J_rows) J_rows->assign(feats.size(), {});
  for (size_t fid = 0; fid < feats.size(); ++fid) {
    Feature & ft = feats[fid];
    const size_t m = ft.Le_ps.size();
    std::vector<double> uvs(2 * m), I_b(m);
    std::vector<V3> p_Lk(m);
    std::vector<Pose> T_Le_Lt(m);
    int st = kUnprocessed;
    for (size_t i = 0; i < m; ++i) {
      const V3 p_Le_b = d_Le * ft.Le_ps[i];
      bool ok;
      try {
        ok = project_undistorted(p_Le_b, p_Lk[i], &uvs[2 * i], T_Le_Lt[i], fr, c);
      } catch (const ProjectThrow &) {
        ok = false;
        out.n_exceptions++;
      }
      if (!ok) {
        st = kPointProjectUndistorted;
        break;
      }
      const double rng = refcpu::norm(p_Lk[i]);
      if (rng < c.range_min || rng > c.range_max) {
        st = kPointRange;
        break;
      }
      const int ux = static_cast<int>(std::round(uvs[2 * i])), uy = static_cast<int>(std::round(uvs[2 * i + 1]));
      if (!fr.img_mask[static_cast<size_t>(uy) * c.cols + ux]) {
        st = kPointMask;
        break;
      }
      if (ux < c.margin_size || ux >= c.cols - c.margin_size || uy < c.margin_size || uy >= c.rows - c.margin_size) {
        st = kPointMaskMargin;
        break;
      }
      if (std::abs(fr.img_range[static_cast<size_t>(uy) * c.cols + ux] - rng) > c.occlusion_range_diff_threshold) {
        st = kPointRangeDiff;
        break;
      }
      I_b[i] = sub_pixel(fr.img_intensity, c.rows, c.cols, uvs[2 * i], uvs[2 * i + 1]);
    }
    if (st != kUnprocessed) {
      statuses[fid] = st;
      continue;
    }
    std::vector<double> psi_b;
    double mean_b, sigma_b;
    get_psi(I_b, mean_b, sigma_b, psi_b);
    std::vector<double> e(m);
    double e2 = 0;
    for (size_t i = 0; i < m; ++i) {
      e[i] = psi_b[i] - ft.psi[i];
      e2 += e[i] * e[i];
    }
    const double e_ncc = (2 - e2) / 2;
    if (e_ncc < c.max_error) {
      statuses[fid] = kMaxError;
      continue;
    }
    statuses[fid] = kValid;
    ft.center[0] = uvs[2 * (m / 2)];
    ft.center[1] = uvs[2 * (m / 2) + 1];
    std::vector<double> D_b(6 * m), D_a(6 * m, 0.0);
    for (size_t i = 0; i < m; ++i) {
      const double gx = sub_pixel(fr.img_dx, c.rows, c.cols, uvs[2 * i], uvs[2 * i + 1]);
      const double gy = sub_pixel(fr.img_dy, c.rows, c.cols, uvs[2 * i], uvs[2 * i + 1]);
      double P[6];
      projection_jacobian(p_Lk[i], c, P);
      double g3[3];  // dI_duv * duv_dp (1 x 3)
      for (int k = 0; k < 3; ++k) g3[k] = gx * P[k] + gy * P[3 + k];
      const V3 p_Be_a = c.T_B_L * ft.Le_ps[i];
      const V3 p_Be_b = d_Be * p_Be_a;
      const M3 Rk = refcpu::transpose(T_Le_Lt[i].R) * refcpu::transpose(c.T_B_L.R);  // R_Lk_b_Be_b
      auto hat = [](const V3 & v) {
        M3 h;
        h(0, 1) = -v.z; h(0, 2) = v.y; h(1, 0) = v.z; h(1, 2) = -v.x; h(2, 0) = -v.y; h(2, 1) = v.x;
        return h;
      };
      const M3 A = Rk * hat(p_Be_b);
      for (int k = 0; k < 3; ++k) {
        D_b[6 * i + k] = g3[0] * A(0, k) + g3[1] * A(1, k) + g3[2] * A(2, k);
        D_b[6 * i + 3 + k] = -(g3[0] * Rk(0, k) + g3[1] * Rk(1, k) + g3[2] * Rk(2, k));
      }
      if (binary) {
        const M3 Rka = Rk * d_Be.R;
        const M3 Aa = Rka * hat(p_Be_a);
        for (int k = 0; k < 3; ++k) {
          D_a[6 * i + k] = -(g3[0] * Aa(0, k) + g3[1] * Aa(1, k) + g3[2] * Aa(2, k));
          D_a[6 * i + 3 + k] = g3[0] * Rka(0, k) + g3[1] * Rka(1, k) + g3[2] * Rka(2, k);
        }
      }
    }
    // J_psi = ((I - psi psi^T) / sigma) (I - 1 1^T / m)   (photometric_utils.cpp:21-27), applied to D
    auto apply_jpsi = [&](const std::vector<double> & D, std::vector<double> & J) {
      J.assign(6 * m, 0.0);
      for (int k = 0; k < 6; ++k) {
        double cm = 0;
        for (size_t i = 0; i < m; ++i) cm += D[6 * i + k];
        cm /= static_cast<double>(m);
        double pd = 0;
        for (size_t i = 0; i < m; ++i) pd += psi_b[i] * (D[6 * i + k] - cm);
        for (size_t i = 0; i < m; ++i) J[6 * i + k] = ((D[6 * i + k] - cm) - psi_b[i] * pd) / sigma_b;
      }
    };
    std::vector<double> J_b, J_a;
    apply_jpsi(D_b, J_b);
    if (binary) apply_jpsi(D_a, J_a);
    const double whitened = std::sqrt(e2) / c.sigma;
    double sw = 1.0;
    if (c.use_robust_cost_function) {
      const double p = c.robust_cost_function_parameter;
      if (c.robust_is_huber)
        sw = std::fabs(whitened) <= p ? 1.0 : std::sqrt(p / std::fabs(whitened));
      else
        sw = p * p / (p * p + whitened * whitened);
    }
    const double w = sw / c.sigma;
    for (double & v : J_b) v *= w;
    for (double & v : e) v *= w;
    for (int r = 0; r < 6; ++r) {
      for (int cc = 0; cc < 6; ++cc) {
        double s = 0;
        for (size_t i = 0; i < m; ++i) s += J_b[6 * i + r] * J_b[6 * i + cc];
        Hbb[6 * r + cc] += s;
      }
      double s = 0;
      for (size_t i = 0; i < m; ++i) s += J_b[6 * i + r] * e[i];
      bb[r] += s;
    }
    {
      double s = 0;
      for (size_t i = 0; i < m; ++i) s += e[i] * e[i];
      fsum += s;
    }
    if (binary) {
      for (double & v : J_a) v *= w;
      for (int r = 0; r < 6; ++r) {
        for (int cc = 0; cc < 6; ++cc) {
          double s1 = 0, s2 = 0;
          for (size_t i = 0; i < m; ++i) {
            s1 += J_a[6 * i + r] * J_a[6 * i + cc];
            s2 += J_b[6 * i + r] * J_a[6 * i + cc];
          }
          Haa[6 * r + cc] += s1;
          Hba[6 * r + cc] += s2;
        }
        double s = 0;
        for (size_t i = 0; i < m; ++i) s += J_a[6 * i + r] * e[i];
        ba[r] += s;
      }
    }
    if (e_rows) (*e_rows)[fid] = e;
    if (J_rows) (*J_rows)[fid] = J_b;
  }
  for (int32_t s : statuses) out.status_hist[s]++;
  out.f = fsum;
  if (binary) {
    std::memcpy(out.H_bb, Hbb, sizeof(Hbb));
    std::memcpy(out.H_ba, Hba, sizeof(Hba));
    std::memcpy(out.H_aa, Haa, sizeof(Haa));
    std::memcpy(out.b_b, bb, sizeof(bb));
    std::memcpy(out.b_a, ba, sizeof(ba));
    return;
  }
  // :336-351 — J_b_T_J_b = VSVt J_I VSVt;  J_b_T_b = VSVt J_I VSVt J_I^-1 b_I
  double I6[36];
  for (int i = 0; i < 36; ++i) I6[i] = (i % 7 == 0) ? 1.0 : 0.0;
  const double * V = VSVt ? VSVt : I6;
  double t1[36], t2[36], inv[36], t3[36];
  mat6_mul(V, Hbb, t1);
  mat6_mul(t1, V, t2);   // VSVt J_I VSVt
  mat6_inv(Hbb, inv);
  mat6_mul(t2, inv, t3);
  std::memcpy(out.H_bb, t2, sizeof(t2));
  for (int r = 0; r < 6; ++r) {
    double s = 0;
    for (int k = 0; k < 6; ++k) s += t3[6 * r + k] * bb[k];
    out.b_b[r] = s;
  }
  M3 Hr, Ht;
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) {
      Hr(r, cc) = t2[6 * r + cc];
      Ht(r, cc) = t2[6 * (3 + r) + 3 + cc];
    }
  V3 lr, lt;
  M3 Er, Et;
  refcpu::compute_localizability(Hr, lr, Er);
  refcpu::compute_localizability(Ht, lt, Et);
  for (int i = 0; i < 3; ++i) {
    out.loc_rot_final[i] = lr[i];
    out.loc_trans_final[i] = lt[i];
  }
  std::memcpy(out.eigvec_rot, Er.m, sizeof(Er.m));
  std::memcpy(out.eigvec_trans, Et.m, sizeof(Et.m));
}

// ----------------------------------------------------------------------------------------------
// Photometric::detectFeatures (src/lidar/photometric.cpp:516-745)
// ----------------------------------------------------------------------------------------------
inline uint8_t sat_u8_round(double v)  // saturate_cast<uchar>(cvRound(v)): round half to even
{
  const double r = std::nearbyint(v);
  return static_cast<uint8_t>(r < 0 ? 0 : (r > 255 ? 255 : r));
}

// O10: modules/imgproc/src/drawing.cpp Circle(), fill = 1, colour 0, clipped to the image
inline void fill_circle_zero(std::vector<uint8_t> & img, int rows, int cols, int cx, int cy, int radius)
{
  auto hline = [&](int y, int x0, int x1) {
    if (y < 0 || y >= rows) return;
    x0 = std::max(x0, 0);
    x1 = std::min(x1, cols - 1);
    for (int x = x0; x <= x1; ++x) img[static_cast<size_t>(y) * cols + x] = 0;
  };
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    hline(cy - dy, cx - dx, cx + dx);
    hline(cy + dy, cx - dx, cx + dx);
    hline(cy - dx, cx - dy, cx + dy);
    hline(cy + dx, cx - dy, cx + dy);
    dy++;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

// src/lidar/photometric_utils.cpp:453-483 — nearest free grid point (the 8 neighbours when the rounded one is taken)
inline std::pair<int, int> snap_point(const std::pair<double, double> & p, std::set<std::pair<int, int>> & used)
{
  const int grid_x = static_cast<int>(std::round(p.first)), grid_y = static_cast<int>(std::round(p.second));
  const std::pair<int, int> candidate{grid_x, grid_y};
  if (used.find(candidate) == used.end()) {
    used.insert(candidate);
    return candidate;
  }
  double best_distance = std::numeric_limits<double>::max();
  std::pair<int, int> best = candidate;
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy) {
      if (dx == 0 && dy == 0) continue;
      const std::pair<int, int> alt{grid_x + dx, grid_y + dy};
      if (used.find(alt) == used.end()) {
        const double ddx = alt.first - p.first, ddy = alt.second - p.second;
        const double dist = std::sqrt(ddx * ddx + ddy * ddy);
        if (dist < best_distance) {
          best_distance = dist;
          best = alt;
        }
      }
    }
  used.insert(best);
  return best;
}

// src/lidar/photometric_utils.cpp:485-518 — the patch pattern rotated into the (edge normal, edge tangent) frame and
// snapped to distinct pixels
inline std::vector<std::pair<int, int>> gradient_based_locations(float grad_x, float grad_y, const std::vector<std::pair<int, int>> & pattern)
{
  const double norm = std::sqrt(grad_x * grad_x + grad_y * grad_y) + 1e-6;  // float products, promoted by the sqrt
  const double normal_x = -grad_y / norm, normal_y = grad_x / norm, tangent_x = grad_x / norm, tangent_y = grad_y / norm;
  std::vector<std::pair<int, int>> snapped;
  std::set<std::pair<int, int>> used;
  for (const auto & p : pattern) {
    const double x = static_cast<double>(p.first), y = static_cast<double>(p.second);
    snapped.push_back(snap_point({normal_x * x + tangent_x * y, normal_y * x + tangent_y * y}, used));
  }
  return snapped;
}

// O11: centre value of cornerEigenValsAndVecs(roi(7 x 7), blockSize 5, ksize 3) -> gradient direction (i_x, i_y)
inline void patch_gradient_direction(const std::vector<float> & I, int cols, int x, int y, int patch_size, float & ix, float & iy)
{
  const int off = patch_size / 2 + 1;  // :583, ROI is (patch_size + 2)^2 and the value is read at (off, off)
  const int blk = 5, half = blk / 2;
  const float scale = static_cast<float>(1.0 / (4.0 * blk));
  float a = 0, b = 0, cc = 0;
  (void)off;
  for (int dy = -half; dy <= half; ++dy)
    for (int dx = -half; dx <= half; ++dx) {
      auto at = [&](int yy, int xx) { return I[static_cast<size_t>(y + dy + yy) * cols + (x + dx + xx)]; };
      // Sobel 3x3 (separable [1 2 1] x [-1 0 1]), scaled
      const float gx = ((at(-1, 1) - at(-1, -1)) + 2.f * (at(0, 1) - at(0, -1)) + (at(1, 1) - at(1, -1))) * scale;
      const float gy = ((at(1, -1) - at(-1, -1)) + 2.f * (at(1, 0) - at(-1, 0)) + (at(1, 1) - at(-1, 1))) * scale;
      a += gx * gx;
      b += gx * gy;
      cc += gy * gy;
    }
  auto eigvec = [&](double l, float & ex, float & ey) {
    double xx = b, yy = l - a, e = std::fabs(xx);
    if (e + std::fabs(yy) < 1e-4) {
      yy = b;
      xx = l - cc;
      e = std::fabs(xx);
      if (e + std::fabs(yy) < 1e-4) {
        e = 1. / (e + std::fabs(yy) + FLT_EPSILON);
        xx *= e;
        yy *= e;
      }
    }
    const double d = 1. / std::sqrt(xx * xx + yy * yy + DBL_EPSILON);
    ex = static_cast<float>(xx * d);
    ey = static_cast<float>(yy * d);
  };
  const double u = (a + cc) * 0.5, v = std::sqrt((a - cc) * (a - cc) * 0.25 + static_cast<double>(b) * b);
  const float e1 = static_cast<float>(u + v), e2 = static_cast<float>(u - v);
  float x1, y1, x2, y2;
  eigvec(u + v, x1, y1);
  eigvec(u - v, x2, y2);
  if (e1 > e2) {  // :590-596
    ix = x1;
    iy = y1;
  } else {
    ix = x2;
    iy = y2;
  }
}

// gradient magnitude image (photometric.cpp:536-540) and detection mask before the per-feature circles (:524-525)
inline void detection_images(const PhotoConfig & c, const Frame & f, std::vector<uint8_t> & grad, std::vector<uint8_t> & mask)
{
  const size_t npx = static_cast<size_t>(c.rows) * c.cols;
  grad.resize(npx);
  mask.assign(npx, 0);
  for (int v = 0; v < c.rows; ++v)
    for (int u = 0; u < c.cols; ++u) {
      const size_t px = static_cast<size_t>(v) * c.cols + u;
      const bool margin = v >= c.margin_size && v < c.rows - c.margin_size && u >= c.margin_size && u < c.cols - c.margin_size;
      mask[px] = (f.img_mask[px] & (margin ? 1 : 0));
      const uint8_t ax = sat_u8_round(std::fabs(f.img_dx[px])), ay = sat_u8_round(std::fabs(f.img_dy[px]));
      grad[px] = sat_u8_round(static_cast<double>(static_cast<float>(ax) * 0.5f + static_cast<float>(ay) * 0.5f));
    }
  erode_ones(mask, c.rows, c.cols, c.patch_size + c.erosion_buffer);
}

// detectFeatures proper.  T_W_Be: pose of the frame's key; bias_directions: 3-vectors.  next_id: monotonic feature id.
inline void detect_features(const PhotoConfig & c, const Frame & f, int num_to_detect, std::vector<Feature> & features,
                            const Pose & T_W_Be, const std::vector<V3> & bias_directions, uint32_t & next_id)
{
  if (num_to_detect <= 0) return;
  std::vector<uint8_t> grad, mask;
  detection_images(c, f, grad, mask);
  for (const auto & ft : features) fill_circle_zero(mask, c.rows, c.cols, static_cast<int>(ft.center[0]), static_cast<int>(ft.center[1]), c.nma_radius);
  std::vector<std::pair<double, std::pair<int, int>>> gradients;  // (gradient, (u, v))
  for (int v = 0; v < c.rows; ++v)
    for (int u = 0; u < c.cols; ++u) {
      const size_t px = static_cast<size_t>(v) * c.cols + u;
      if (!mask[px]) continue;
      if (grad[px] > c.gradient_threshold) gradients.emplace_back(grad[px], std::make_pair(u, v));
    }
  std::sort(gradients.begin(), gradients.end(),
            [](const std::pair<double, std::pair<int, int>> & a, const std::pair<double, std::pair<int, int>> & b) { return a.first > b.first; });
  std::vector<std::pair<int, int>> candidates;
  for (const auto & g : gradients) {
    const int u = g.second.first, v = g.second.second;
    if (!mask[static_cast<size_t>(v) * c.cols + u]) continue;
    candidates.push_back(g.second);
    fill_circle_zero(mask, c.rows, c.cols, u, v, c.nma_radius);
  }
  const size_t nb = bias_directions.size();
  std::vector<std::vector<std::pair<double, int>>> scores(nb, std::vector<std::pair<double, int>>(candidates.size(), {0.0, 0}));
  for (size_t i = 0; i < candidates.size(); ++i) {
    float ix, iy;
    patch_gradient_direction(f.img_intensity, c.cols, candidates[i].first, candidates[i].second, c.patch_size, ix, iy);
    const int pidx = f.img_idx[static_cast<size_t>(candidates[i].second) * c.cols + candidates[i].first];
    if (pidx < 0) continue;
    const Point32 & q = f.points_deskewed[pidx];
    double P[6];
    projection_jacobian(V3{q.x, q.y, q.z}, c, P);
    for (size_t b = 0; b < nb; ++b) {
      const V3 & d = bias_directions[b];
      double w0 = P[0] * d.x + P[1] * d.y + P[2] * d.z, w1 = P[3] * d.x + P[4] * d.y + P[5] * d.z;
      const double nn = std::sqrt(w0 * w0 + w1 * w1);
      if (nn > 0) {  // Eigen normalized(): unchanged when the norm is zero
        w0 /= nn;
        w1 /= nn;
      }
      scores[b][i] = {std::fabs(ix * w0 + iy * w1), static_cast<int>(i)};
    }
  }
  for (size_t b = 0; b < nb; ++b)
    std::sort(scores[b].begin(), scores[b].end(), [](const std::pair<double, int> & a, const std::pair<double, int> & bb) { return a.first > bb.first; });
  std::vector<int> selected;
  if (nb)
    for (size_t i = 0; i < scores[0].size(); ++i)
      for (size_t b = 0; b < nb; ++b) {
        const int idx = scores[b][i].second;
        if (std::find(selected.begin(), selected.end(), idx) == selected.end()) selected.push_back(idx);
      }
  int num_added = 0;
  const Pose T_map = c.T_B_L.inverse() * T_W_Be * c.T_B_L;  // :666-667
  for (const int idx : selected) {
    const int lx = candidates[idx].first, ly = candidates[idx].second;
    Feature ft;
    ft.id = next_id++;
    ft.life_time = 1;
    ft.center[0] = lx;
    ft.center[1] = ly;
    std::vector<std::pair<int, int>> sampling_locations = c.patch_offsets;
    if (c.rotate_patch_to_align_with_gradient) {  // :659-684 (false in every shipped configuration)
      float ix, iy;
      patch_gradient_direction(f.img_intensity, c.cols, lx, ly, c.patch_size, ix, iy);
      sampling_locations = gradient_based_locations(ix, iy, c.patch_offsets);
    }
    bool missing = false;
    for (const auto & o : sampling_locations) {
      const int u = lx + o.first, v = ly + o.second;
      // A rotated pattern reaches up to 2 * sqrt(2) + 1 pixels further than the eroded mask vouches for when
      // erosion_buffer is small: the reference then indexes the cloud with -1 (or reads outside the image) — undefined
      // behaviour there, the candidate is skipped here and in the product.
      const int pi = (u >= 0 && u < c.cols && v >= 0 && v < c.rows) ? f.img_idx[static_cast<size_t>(v) * c.cols + u] : -1;
      if (pi < 0) {
        missing = true;
        break;
      }
      const Point32 & q = f.points_deskewed[pi];
      ft.Le_ps.push_back(T_map * V3{q.x, q.y, q.z});
      ft.intensities.push_back(f.img_intensity[static_cast<size_t>(v) * c.cols + u]);
    }
    if (missing) continue;
    const size_t m = ft.Le_ps.size();
    V3 mean;
    for (const V3 & p : ft.Le_ps) mean = mean + p;
    mean = (1.0 / static_cast<double>(m)) * mean;
    bool far = false;
    double cov[9] = {0};
    for (const V3 & p : ft.Le_ps) {
      const V3 d = p - mean;
      if (refcpu::norm(d) > c.max_dist_from_mean) far = true;
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) cov[3 * r + cc] += d[r] * d[cc];
    }
    if (far) continue;  // :685-687 (the id stays consumed)
    M3 C;
    for (int i = 0; i < 9; ++i) C.m[i] = cov[i] / static_cast<double>(m - 1);
    V3 ev;
    M3 E;
    refcpu::self_adjoint_eigen3(C, ev, E);
    V3 normal{E(0, 0), E(1, 0), E(2, 0)};
    bool off_plane = false;
    for (const V3 & p : ft.Le_ps)
      if (std::fabs(refcpu::dot(p - mean, normal)) > c.max_dist_from_plane) off_plane = true;
    if (off_plane) continue;  // :694-696
    if (refcpu::dot(normal, mean) / refcpu::norm(mean) > 0) normal = -normal;
    ft.normal = normal;
    get_psi(ft.intensities, ft.mean_intensity, ft.sigma_intensity, ft.psi);
    features.push_back(ft);
    fill_circle_zero(mask, c.rows, c.cols, lx, ly, c.nma_radius);  // (the mask is not read again)
    num_added++;
    if (num_added >= num_to_detect) break;
  }
}

}  // namespace refphoto
