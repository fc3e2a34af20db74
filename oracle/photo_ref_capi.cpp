// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over oracle/photo_ref.hpp (the CPU restatement of the
// photometric path) for the ctypes binding oracle/photo_ref.py.  PARITY UNPINNED, see photo_ref.hpp.
// The configuration / feature / result structs have the layout of include/mimosa_hip.h's mh_photo_config,
// mh_photo_feature and mh_photo_result so one ctypes definition serves both sides of a parity test.
#include <cstring>
#include <memory>

#include "../include/mimosa_hip.h"
#include "photo_ref.hpp"

using namespace refphoto;

namespace
{
struct Handle
{
  PhotoConfig cfg;
  std::shared_ptr<Frame> frame;
  std::vector<Feature> features;  // map_Le_features_
  uint32_t next_id = 0;
};
struct FactorHandle
{
  Handle * h;
  std::shared_ptr<Frame> frame;
  std::vector<Feature> features;
  bool binary;
  double VSVt[36];
  std::vector<int32_t> statuses;
  std::vector<std::vector<double>> e_rows, J_rows;
};
Pose pose_from(const double * R, const double * t)
{
  Pose p;
  std::memcpy(p.R.m, R, 9 * sizeof(double));
  p.t = {t[0], t[1], t[2]};
  return p;
}
}  // namespace

extern "C" {

void * refphoto_create(const mh_photo_config * c)
{
  Handle * h = new Handle;
  PhotoConfig & p = h->cfg;
  p.rows = c->rows;
  p.cols = c->cols;
  p.destagger = c->destagger;
  p.pixel_shift_by_row.assign(c->pixel_shift_by_row, c->pixel_shift_by_row + c->rows);
  p.beam_altitude_angles.assign(c->beam_altitude_angles, c->beam_altitude_angles + c->rows);
  p.range_min = c->range_min;
  p.range_max = c->range_max;
  p.erosion_buffer = c->erosion_buffer;
  p.patch_size = c->patch_size;
  p.margin_size = c->margin_size;
  p.intensity_scale = c->intensity_scale;
  p.intensity_gamma = c->intensity_gamma;
  p.remove_lines = c->remove_lines;
  p.filter_brightness = c->filter_brightness;
  p.gaussian_blur = c->gaussian_blur;
  p.gaussian_blur_size = c->gaussian_blur_size;
  p.gradient_threshold = c->gradient_threshold;
  p.max_dist_from_mean = c->max_dist_from_mean;
  p.max_dist_from_plane = c->max_dist_from_plane;
  p.nma_radius = c->nma_radius;
  p.rotate_patch_to_align_with_gradient = c->rotate_patch_to_align_with_gradient != 0;
  p.num_features_detect = c->num_features_detect;
  p.occlusion_range_diff_threshold = c->occlusion_range_diff_threshold;
  p.max_feature_life_time = c->max_feature_life_time;
  if (c->high_pass_fir) p.high_pass_fir.assign(c->high_pass_fir, c->high_pass_fir + c->n_high_pass);
  if (c->low_pass_fir) p.low_pass_fir.assign(c->low_pass_fir, c->low_pass_fir + c->n_low_pass);
  p.brightness_window_w = c->brightness_window_size[0];
  p.brightness_window_h = c->brightness_window_size[1];
  p.lidar_origin_to_beam_origin_mm = c->lidar_origin_to_beam_origin_mm;
  for (int i = 0; i < c->n_patch_offsets; ++i) p.patch_offsets.emplace_back(c->patch_offsets[2 * i], c->patch_offsets[2 * i + 1]);
  p.use_robust_cost_function = c->use_robust_cost_function;
  p.robust_is_huber = c->robust_cost_function == 0;
  p.robust_cost_function_parameter = c->robust_cost_function_parameter;
  p.error_scale = c->error_scale;
  p.max_error = c->max_error;
  p.sigma = c->sigma;
  p.T_B_L = pose_from(c->T_B_L_R, c->T_B_L_t);
  if (c->static_mask) p.static_mask.assign(c->static_mask, c->static_mask + static_cast<size_t>(c->rows) * c->cols);
  p.derive();
  return h;
}
void refphoto_destroy(void * h) { delete static_cast<Handle *>(h); }

// getGradientBasedLocations (photometric_utils.cpp:485-518) on its own, for the check against the numpy twin
void refphoto_gradient_locations(float grad_x, float grad_y, const int32_t * pattern, int n, int32_t * out)
{
  std::vector<std::pair<int, int>> pat;
  for (int i = 0; i < n; ++i) pat.emplace_back(pattern[2 * i], pattern[2 * i + 1]);
  const auto loc = gradient_based_locations(grad_x, grad_y, pat);
  for (int i = 0; i < n; ++i) {
    out[2 * i] = loc[i].first;
    out[2 * i + 1] = loc[i].second;
  }
}

// returns 0 ok, 1 = the reference would have thrown
int refphoto_preprocess(void * hv, const Point32 * raw, Point32 * desk, size_t n, const uint32_t * ns, const double * T, size_t ng)
{
  Handle * h = static_cast<Handle *>(hv);
  auto f = std::make_shared<Frame>();
  try {
    preprocess(h->cfg, raw, desk, n, ns, T, ng, *f);
  } catch (const std::exception &) {
    return 1;
  }
  h->frame = f;
  return 0;
}

void refphoto_get_image(void * hv, int which, void * out)
{
  Handle * h = static_cast<Handle *>(hv);
  const Frame & f = *h->frame;
  const size_t npx = static_cast<size_t>(f.rows) * f.cols;
  std::vector<uint8_t> grad, mask;
  switch (which) {
    case 0: std::memcpy(out, f.img_intensity.data(), npx * 4); break;
    case 1: std::memcpy(out, f.img_range.data(), npx * 4); break;
    case 2: std::memcpy(out, f.img_dx.data(), npx * 4); break;
    case 3: std::memcpy(out, f.img_dy.data(), npx * 4); break;
    case 4: std::memcpy(out, f.img_mask.data(), npx); break;
    case 5: std::memcpy(out, f.img_idx.data(), npx * 4); break;
    case 6: std::memcpy(out, f.yaw.data(), npx * 4); break;
    case 7: std::memcpy(out, f.proj_idx.data(), npx * 4 * kDuplicatePoints); break;
    case 8:
    case 9:
      detection_images(h->cfg, f, grad, mask);
      std::memcpy(out, which == 8 ? grad.data() : mask.data(), npx);
      break;
    default: break;
  }
}

void refphoto_num_features(void * hv, size_t * nf, size_t * np)
{
  Handle * h = static_cast<Handle *>(hv);
  size_t s = 0;
  for (const auto & f : h->features) s += f.Le_ps.size();
  *nf = h->features.size();
  *np = s;
}
void refphoto_get_features(void * hv, mh_photo_feature * feats, double * Le, double * I, double * psi)
{
  Handle * h = static_cast<Handle *>(hv);
  size_t o = 0;
  for (size_t i = 0; i < h->features.size(); ++i) {
    const Feature & f = h->features[i];
    feats[i].id = f.id;
    feats[i].life_time = f.life_time;
    feats[i].n_points = static_cast<int32_t>(f.Le_ps.size());
    feats[i].pad = 0;
    feats[i].center[0] = f.center[0];
    feats[i].center[1] = f.center[1];
    feats[i].normal[0] = f.normal.x;
    feats[i].normal[1] = f.normal.y;
    feats[i].normal[2] = f.normal.z;
    feats[i].mean_intensity = f.mean_intensity;
    feats[i].sigma_intensity = f.sigma_intensity;
    for (size_t k = 0; k < f.Le_ps.size(); ++k, ++o) {
      Le[3 * o] = f.Le_ps[k].x;
      Le[3 * o + 1] = f.Le_ps[k].y;
      Le[3 * o + 2] = f.Le_ps[k].z;
      I[o] = f.intensities[k];
      psi[o] = f.psi[k];
    }
  }
}
void refphoto_set_features(void * hv, const mh_photo_feature * feats, size_t nf, const double * Le, const double * I, const double * psi)
{
  Handle * h = static_cast<Handle *>(hv);
  h->features.clear();
  size_t o = 0;
  for (size_t i = 0; i < nf; ++i) {
    Feature f;
    f.id = feats[i].id;
    f.life_time = feats[i].life_time;
    f.center[0] = feats[i].center[0];
    f.center[1] = feats[i].center[1];
    f.normal = {feats[i].normal[0], feats[i].normal[1], feats[i].normal[2]};
    f.mean_intensity = feats[i].mean_intensity;
    f.sigma_intensity = feats[i].sigma_intensity;
    for (int k = 0; k < feats[i].n_points; ++k, ++o) {
      f.Le_ps.push_back({Le[3 * o], Le[3 * o + 1], Le[3 * o + 2]});
      f.intensities.push_back(I[o]);
      f.psi.push_back(psi[o]);
    }
    if (f.id >= h->next_id) h->next_id = f.id + 1;
    h->features.push_back(f);
  }
}

void refphoto_detect(void * hv, int num, const double * R, const double * t, const double * bias, size_t nd)
{
  Handle * h = static_cast<Handle *>(hv);
  std::vector<V3> dirs;
  for (size_t i = 0; i < nd; ++i) dirs.push_back({bias[3 * i], bias[3 * i + 1], bias[3 * i + 2]});
  detect_features(h->cfg, *h->frame, num, h->features, pose_from(R, t), dirs, h->next_id);
}

void * refphoto_factor_create(void * hv, const double * VSVt, int binary)
{
  Handle * h = static_cast<Handle *>(hv);
  FactorHandle * f = new FactorHandle;
  f->h = h;
  f->frame = h->frame;
  f->features = h->features;
  f->binary = binary != 0;
  for (int i = 0; i < 36; ++i) f->VSVt[i] = VSVt ? VSVt[i] : (i % 7 == 0 ? 1.0 : 0.0);
  return f;
}
void refphoto_factor_destroy(void * fv) { delete static_cast<FactorHandle *>(fv); }

void refphoto_factor_linearize(void * fv, const double * Rb, const double * tb, const double * Ra, const double * ta, mh_photo_result * out)
{
  FactorHandle * f = static_cast<FactorHandle *>(fv);
  const Pose Tb = pose_from(Rb, tb);
  Pose Ta;
  if (f->binary) Ta = pose_from(Ra, ta);
  PhotoResult r;
  linearize(f->h->cfg, *f->frame, f->features, Tb, f->binary ? &Ta : nullptr, f->VSVt, r, f->statuses, &f->e_rows, &f->J_rows);
  std::memset(out, 0, sizeof(*out));
  std::memcpy(out->H_bb, r.H_bb, sizeof(r.H_bb));
  std::memcpy(out->H_ba, r.H_ba, sizeof(r.H_ba));
  std::memcpy(out->H_aa, r.H_aa, sizeof(r.H_aa));
  std::memcpy(out->b_b, r.b_b, sizeof(r.b_b));
  std::memcpy(out->b_a, r.b_a, sizeof(r.b_a));
  out->f = r.f;
  std::memcpy(out->loc_trans_final, r.loc_trans_final, sizeof(r.loc_trans_final));
  std::memcpy(out->loc_rot_final, r.loc_rot_final, sizeof(r.loc_rot_final));
  std::memcpy(out->eigvec_trans, r.eigvec_trans, sizeof(r.eigvec_trans));
  std::memcpy(out->eigvec_rot, r.eigvec_rot, sizeof(r.eigvec_rot));
  std::memcpy(out->status_hist, r.status_hist, sizeof(r.status_hist));
  out->n_exceptions = r.n_exceptions;
  out->gpu_ms = -1.f;
}

// statuses[nf], centers[2 nf], rows[nf * 64 * 8] = {e, J_b[6], valid}
void refphoto_factor_get_state(void * fv, int32_t * statuses, double * centers, double * rows)
{
  FactorHandle * f = static_cast<FactorHandle *>(fv);
  const size_t nf = f->features.size();
  for (size_t i = 0; i < nf; ++i) {
    statuses[i] = f->statuses.empty() ? 0 : f->statuses[i];
    centers[2 * i] = f->features[i].center[0];
    centers[2 * i + 1] = f->features[i].center[1];
    if (!rows) continue;
    double * r = rows + i * 64 * 8;
    std::memset(r, 0, 64 * 8 * sizeof(double));
    if (i < f->e_rows.size())
      for (size_t k = 0; k < f->e_rows[i].size(); ++k) {
        r[8 * k] = f->e_rows[i][k];
        for (int j = 0; j < 6; ++j) r[8 * k + 1 + j] = f->J_rows[i][6 * k + j];
        r[8 * k + 7] = 1.0;
      }
  }
}

// Photometric::updateMap (src/lidar/photometric.cpp:396-514): bookkeeping from the factor's statuses, then detection
void refphoto_update_map(void * hv, void * fv, const double * R, const double * t, const double * bias, size_t nd)
{
  Handle * h = static_cast<Handle *>(hv);
  if (fv) {
    FactorHandle * f = static_cast<FactorHandle *>(fv);
    std::vector<size_t> invalid;
    for (size_t i = 0; i < f->statuses.size(); ++i) {
      if (f->statuses[i] != kValid) {
        invalid.push_back(i);
      } else {
        h->features[i].center[0] = f->features[i].center[0];
        h->features[i].center[1] = f->features[i].center[1];
        h->features[i].life_time++;
        if (h->features[i].life_time >= h->cfg.max_feature_life_time) invalid.push_back(i);
      }
    }
    for (auto it = invalid.rbegin(); it != invalid.rend(); ++it) h->features.erase(h->features.begin() + static_cast<long>(*it));
  }
  refphoto_detect(hv, h->cfg.num_features_detect - static_cast<int>(h->features.size()), R, t, bias, nd);
}

}  // extern "C"
