// LiDAR photometric path: device-side structs and launcher declarations shared by
// photo_kernels.hip and the C ABI (photo_api.hip).
// Reference: src/lidar/photometric.cpp, include/mimosa/lidar/photometric_factor.hpp, src/lidar/photometric_utils.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mimosa_hip.h"

namespace mh
{
constexpr int kPhotoDup = 10;        // DUPLICATE_POINTS, include/mimosa/lidar/photometric_utils.hpp:17
constexpr int kPhotoMaxPatch = 64;   // points per feature: one wave lane each (5 x 5 = 25 by default, 8 x 8 = 64)
constexpr int kPhotoMaxTaps = 129;   // FIR length limit (the shipped filters have 33 taps)
constexpr int kPhotoPartial = 96;    // per-feature sums: 28 (unary) or 91 (binary) used

// Projection model + thresholds: the derived parameters of src/lidar/photometric_config.cpp:98-110 and the
// PhotometricConfig fields the kernels read.  Plain scalars only (kernel argument).
struct PhotoModel
{
  int rows, cols;
  int destagger;
  double fx, fy, cx;
  float beam_offset_m;
  float range_min, range_max;
  float alt_first, alt_last;  // beam_altitude_angles.front() / .back()
  int margin_size;
  float occlusion_range_diff_threshold;
  const float * alt;          // beam_altitude_angles (rows, degrees, descending)
  const int * pixel_shift;    // pixel_shift_by_row (rows)
};

// Device view of a Frame (include/mimosa/lidar/photometric_utils.hpp:42-92)
struct PhotoFrameView
{
  const mh_point32 * points;  // points_deskewed
  int n_points;
  const float * intensity;
  const float * range;
  const float * dx;
  const float * dy;
  const uint8_t * mask;
  const int32_t * idx;       // img_deskewed_cloud_idx
  const int32_t * proj;      // proj_idx: rows * cols * kPhotoDup
  const float * yaw;
  const uint32_t * pose_ns;  // interpolated_map_T_Le_Lt keys, ascending
  const double * pose_Rt;    // 12 doubles per key: R row-major, t
  int n_poses;
};

struct PhotoCounters
{
  uint32_t project_throw;  // project(): "Invalid x coordinate" (photometric_utils.cpp:90-97) — the reference throws
  uint32_t pose_missing;   // interpolated_map_T_Le_Lt.at(): out_of_range
  uint32_t pad[2];
};

// frame reset: intensity / range / masks 0, idx -1, proj_idx "empty", corrected-intensity staging NaN
// (+ a second job in the same launch: copy_bytes (a multiple of 16) from a mapped pinned block to the device — the frame's pose table)
hipError_t launch_photo_clear(int npx, int n_pts, float * img_raw, float * range, uint8_t * mask_raw, uint8_t * yaw_valid,
                              int32_t * idx, int32_t * proj, float * int_out, const void * copy_src, void * copy_dst, size_t copy_bytes, hipStream_t stream);
// preprocess stage 1 (photometric.cpp:121-130, 204-217): yaw of the raw points, image fill from the deskewed ones
hipError_t launch_photo_scatter(const PhotoModel & m, const mh_point32 * raw, const mh_point32 * desk, int n, float * yaw,
                                uint8_t * yaw_valid, float * intensity, float * range, uint8_t * mask, int32_t * idx,
                                hipStream_t stream);
// stage 2 (:135-199): per-row interpolation of the missing yaw angles
hipError_t launch_photo_yaw_fill(const PhotoModel & m, float * yaw, const uint8_t * yaw_valid, hipStream_t stream);
// stage 3 (:218-244): project every deskewed point with the yaw table, build proj_idx (first 9 indices per pixel)
hipError_t launch_photo_project(const PhotoModel & m, const mh_point32 * desk, int n, const float * yaw, int32_t * proj,
                                PhotoCounters * counters, hipStream_t stream);
hipError_t launch_photo_proj_finalize(int n_pixels, int32_t * proj, hipStream_t stream);
// filter chain (:246-317): every stage reads `in` and writes `out` (rows x cols f32)
hipError_t launch_photo_vfir(const float * in, float * out, int rows, int cols, const float * taps, int n_taps, float scale,
                             float gamma, hipStream_t stream);
hipError_t launch_photo_hfir_sub(const float * hp, const float * raw_in, float * out, int rows, int cols, const float * taps,
                                 int n_taps, float scale, float gamma, hipStream_t stream);
hipError_t launch_photo_scale(const float * in, float * out, int n, float scale, float gamma, hipStream_t stream);
hipError_t launch_photo_brightness(const float * in, float * out, int rows, int cols, int win_w, int win_h, hipStream_t stream);
hipError_t launch_photo_gauss_trunc(const float * in, float * out, int rows, int cols, int do_gauss, hipStream_t stream);
// Sobel (:316-317) + corrected intensities back into the cloud (:307-314)
hipError_t launch_photo_sobel_writeback(const float * img, float * dx, float * dy, const int32_t * idx, mh_point32 * desk,
                                        float * intensity_out, int rows, int cols, hipStream_t stream);
// createMask (:349-371): static mask, erosion with a k x k ones kernel (O6: out-of-image pixels are ignored)
hipError_t launch_photo_erode(const uint8_t * in, const uint8_t * static_mask, int margin, uint8_t * out, int rows, int cols,
                              int k, hipStream_t stream);
// detectFeatures' per-pixel part (:524-540): gradient magnitude image
// Photometric::preprocess behind the scatter — the image chain (photometric.cpp:246-320), the mask erosion (:349-371), the
// yaw-table fill (:135-199) and the projection index (:218-244) — the scatter and three multi-job launches, four in all
// (photo_kernels.hip, "Round 4" and photo_scatter_stamp_kernel)
constexpr size_t kPhotoProjPreBytes = 32;
struct PhotoChain
{
  const float * raw;
  float * ta;
  float * tb;
  float * fin;
  float * dx;
  float * dy;
  const int32_t * idx;
  float * intensity_out;  // n_pts floats: NaN = the point owns no pixel
  const float * hp;
  const float * lp;
  const uint8_t * static_mask;
  uint8_t * mask_out;
  float * yaw;
  const mh_point32 * raw_points;   // the scan as it came (yaw angles)
  const mh_point32 * desk_points;  // deskewed (image fill, projection): the frame's cloud
  const mh_point32 * desk_src;     // where the deskewed points are read from when the frame's cloud is still to be filled (the
                                   // scatter copies them over), or nullptr / == desk_points
  mh_point32 * desk_writeback;     // a cloud that also receives the corrected intensities (:307-314), or nullptr
  // no frame-reset launch: per-pixel frame stamps (2 x rows x cols words: [raw point landed | deskewed point in range landed],
  // zero when allocated; a pixel is marked when its word equals seq != 0), and the arrays the scatter fills
  uint32_t * stamps;
  uint32_t seq;
  float * raw_w;
  float * range;
  int32_t * idx_w;
  void * proj_pre;  // n_pts x kPhotoProjPreBytes of scratch: the projection's front (stage A) hands its findings to the back (stage B)
  // the frame's pose table, copied by a job of stage A: copy_bytes (a multiple of 16) from a mapped pinned block
  const void * copy_src;
  void * copy_dst;
  size_t copy_bytes;
  int32_t * proj;
  PhotoCounters * counters;
  int rows, cols, n_pts, n_hp, n_lp, remove_lines, filter_brightness, bw, bh, do_gauss, erode_k;
  float scale, gamma;
};
bool photo_stages_fit(const PhotoChain & c);
hipError_t launch_photo_stages(const PhotoChain & c, const PhotoModel & m, hipStream_t stream);
hipError_t launch_photo_grad(const float * dx, const float * dy, uint8_t * grad, int n, hipStream_t stream);
// (:541-555) pixels with mask != 0 and gradient > thr in row-major order: out[i] = px | grad << 24, *n_out = how many.
// blk: (npx + 255) / 256 words of scratch.
hipError_t launch_photo_candidates(const uint8_t * grad, const uint8_t * mask, int npx, float thr, uint32_t * blk, uint32_t * out,
                                   uint32_t * n_out, hipStream_t stream);
// per candidate c (centre uv[n_off + c]): win49[c] = the 7 x 7 intensity window, rec[c][o] = (x, y, z, intensity) and
// rec_idx[c][o] = point index of the pixel centre + uv[o] (o < n_off) / the centre itself (o == n_off); -1 = no point
hipError_t launch_photo_gather(const int2 * uv, int n_off, int n_cand, bool per_candidate, const float * I, const int32_t * idx,
                               const mh_point32 * pts, int rows, int cols, float * win49, float4 * rec, int32_t * rec_idx, hipStream_t stream);

// PhotometricFactor::linearize (photometric_factor.hpp:136-355): one wave per feature
struct PhotoLinArgs
{
  PhotoModel model;
  PhotoFrameView frame;
  const double * Le_ps;   // n_features x kPhotoMaxPatch x 3
  const double * psi_a;   // n_features x kPhotoMaxPatch
  const int32_t * n_pts;  // n_features
  int n_features;
  int binary;
  double dLe_R[9], dLe_t[3];  // delta_pose_b_a_Le
  double dBe_R[9], dBe_t[3];  // delta_pose_b_a_Be
  double TBL_R[9], TBL_t[3];  // T_B_L
  double sigma, max_error, robust_param;
  int use_robust, robust_is_huber;
  // outputs
  int32_t * status;      // n_features
  double * centers;      // n_features x 2 (written for Valid features)
  double * partials;     // n_features x kPhotoPartial: upper triangle of sum v v^T, v = [J_b(6) (, J_a(6)), e]
  double * rows_out;     // optional (parity tooling): n_features x kPhotoMaxPatch x 8 = {e, J_b[6], valid}
  PhotoCounters * counters;
  unsigned int seq;        // != 0: the last block publishes it to *host_seq (mapped pinned) after everything else
  unsigned int * ticket;   // device counter of finished blocks
  unsigned int * host_seq;
};
hipError_t launch_photo_linearize(const PhotoLinArgs & a, hipStream_t stream);

}  // namespace mh
