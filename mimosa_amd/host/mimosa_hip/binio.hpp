// Little-endian length-prefixed vectors: the exchange format between the Python tooling (tests/, bench.py,
// mimosa_amd/replay.py: write_native_input) and the C++ drivers of the host mirror.
#pragma once

#include <fstream>
#include <stdexcept>
#include <vector>

#include "photometric.hpp"

namespace mimosa_hip
{
namespace binio
{
template <typename T>
inline std::vector<T> read_vec(std::ifstream & f)
{
  uint64_t n = 0;
  f.read(reinterpret_cast<char *>(&n), 8);
  if (!f) throw std::runtime_error("binio: truncated input");
  {
    // the length is file-supplied: it may not promise more than the file still holds
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(here);
    if (here < 0 || end < here || n > static_cast<uint64_t>(end - here) / sizeof(T)) throw std::runtime_error("binio: vector length exceeds the file");
  }
  std::vector<T> v(n);
  f.read(reinterpret_cast<char *>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
  if (!f) throw std::runtime_error("binio: truncated input");
  return v;
}
inline Pose3 pose_from(const std::vector<double> & p, size_t offset = 0)  // 12 doubles: R row-major, t
{
  if (p.size() < offset + 12) throw std::runtime_error("binio: a pose needs 12 doubles");
  return pose3(p.data() + offset, p.data() + offset + 9);
}
// the photometric block: 18 ints, 13 doubles, pixel shifts, beam angles, two FIR kernels, patch offsets, T_B_L
inline lidar::PhotometricConfig read_photo_config(std::ifstream & f)
{
  const auto I = read_vec<int32_t>(f);
  const auto D = read_vec<double>(f);
  if (I.size() != 18 || D.size() != 13) throw std::runtime_error("binio: photometric block has the wrong shape");
  lidar::PhotometricConfig cfg;
  cfg.rows = static_cast<size_t>(I[0]);
  cfg.cols = static_cast<size_t>(I[1]);
  cfg.destagger = I[2] != 0;
  cfg.erosion_buffer = I[3];
  cfg.patch_size = I[4];
  cfg.margin_size = I[5];
  cfg.remove_lines = I[6] != 0;
  cfg.filter_brightness = I[7] != 0;
  cfg.gaussian_blur = I[8] != 0;
  cfg.gaussian_blur_size = I[9];
  cfg.nma_radius = I[10];
  cfg.num_features_detect = static_cast<size_t>(I[11]);
  cfg.max_feature_life_time = I[12];
  cfg.rotate_patch_to_align_with_gradient = I[13] != 0;
  cfg.use_robust_cost_function = I[14] != 0;
  cfg.robust_cost_function = I[15] == 0 ? "huber" : "gemanmcclure";
  cfg.brightness_window_size = {I[16], I[17]};
  cfg.range_min = static_cast<float>(D[0]);
  cfg.range_max = static_cast<float>(D[1]);
  cfg.intensity_scale = static_cast<float>(D[2]);
  cfg.intensity_gamma = static_cast<float>(D[3]);
  cfg.gradient_threshold = static_cast<float>(D[4]);
  cfg.max_dist_from_mean = static_cast<float>(D[5]);
  cfg.max_dist_from_plane = static_cast<float>(D[6]);
  cfg.occlusion_range_diff_threshold = static_cast<float>(D[7]);
  cfg.lidar_origin_to_beam_origin_mm = static_cast<float>(D[8]);
  cfg.robust_cost_function_parameter = D[9];
  cfg.error_scale = D[10];
  cfg.max_error = D[11];
  cfg.sigma = D[12];
  cfg.pixel_shift_by_row = read_vec<int32_t>(f);
  cfg.beam_altitude_angles = read_vec<float>(f);
  cfg.high_pass_fir = read_vec<double>(f);
  cfg.low_pass_fir = read_vec<double>(f);
  const auto offs = read_vec<int32_t>(f);
  cfg.edgelet_patch_offsets.clear();
  for (size_t i = 0; i + 1 < offs.size(); i += 2) cfg.edgelet_patch_offsets.emplace_back(offs[i], offs[i + 1]);
  const auto TBL = read_vec<double>(f);
  cfg.T_B_L = pose_from(TBL);
  return cfg;
}
}  // namespace binio
}  // namespace mimosa_hip
