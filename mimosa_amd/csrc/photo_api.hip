// C ABI of the photometric path (include/mimosa_hip.h: mh_photo_*): lidar::Photometric and PhotometricFactor.
//
// Reference: src/lidar/photometric.cpp (preprocess :92-320, createMask :349-371, updateMap :396-514, detectFeatures
// :516-745), include/mimosa/lidar/photometric_factor.hpp:136-355, src/lidar/photometric_utils.cpp.
// Device work is in photo_kernels.hip; what stays on the host here is what the reference does sequentially on a few
// hundred elements: the std::sort + greedy non-maximum suppression of detectFeatures, the per-candidate 2 x 2
// structure-tensor eigenvector, the plane check of a new feature, and the 6 x 6 epilogue of the unary factor.
// There is no CPU fallback: every entry point needs the context's HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <new>
#include <set>
#include <thread>
#include <vector>

#include <unistd.h>

#include "exact_sort.hpp"
#include "math3.hpp"
#include "mh_internal.hpp"
#include "photo_device.hpp"

namespace
{
struct PhotoFrame  // include/mimosa/lidar/photometric_utils.hpp:42-92, shared_ptr semantics by `refs`
{
  std::atomic<int> refs{1};
  mh_ctx * ctx = nullptr;
  int rows = 0, cols = 0, n_poses = 0;
  size_t n_points = 0;
  DevBuf d_points, d_intensity, d_range, d_dx, d_dy, d_mask, d_idx, d_proj, d_yaw, d_pose_ns, d_pose_Rt;  // (d_pose_Rt unused: the poses sit behind the timestamps in d_pose_ns)
  size_t pose_rt_offset = 0;
  void * h_pose = nullptr;  // pinned: the pose table and its timestamps on their way to the device (no pageable copy, no wait)
  size_t h_pose_cap = 0;
  void release_buffers()
  {
    for (DevBuf * b : {&d_points, &d_intensity, &d_range, &d_dx, &d_dy, &d_mask, &d_idx, &d_proj, &d_yaw, &d_pose_ns, &d_pose_Rt})
      b->release();
    if (h_pose) AllocCache::free_pinned(h_pose, h_pose_cap);
    h_pose = nullptr;
  }
};

void frame_release(PhotoFrame * f)
{
  if (!f) return;
  if (f->refs.fetch_sub(1) == 1) {
    (void)mh_enter(f->ctx);
    (void)hipStreamSynchronize(f->ctx->stream);
    f->release_buffers();
    delete f;
  }
}

struct HostFeature
{
  mh_photo_feature hdr{};
  std::vector<double> Le_ps, intensities, psi;
};

}  // namespace

struct mh_photo
{
  mh_ctx * ctx = nullptr;
  std::atomic<int> refs{1};
  mh_photo_config cfg{};  // scalar fields; the pointer members are re-pointed at the copies below
  std::vector<int32_t> shift, offsets;
  std::vector<float> alt, hp_f, lp_f;
  std::vector<uint8_t> static_mask;
  mh::PhotoModel model{};
  DevBuf d_alt, d_shift, d_hp, d_lp, d_static;
  // per-frame scratch
  DevBuf d_raw_pts, d_img_raw, d_tmp_a, d_tmp_b, d_mask_raw, d_yaw_valid, d_int_out, d_grad, d_detmask, d_xyz, d_cand, d_gather, d_stamps, d_proj_pre;
  // frame stamps of the four-launch preprocess (photo_scatter_stamp_kernel): zeroed when (re)allocated, seq counts the frames
  void * stamps_zeroed_at = nullptr;
  size_t stamps_zeroed_bytes = 0;
  uint32_t frame_seq = 0;
  // pinned, mapped: [0] the preprocess kernels' counters (project() errors of a frame), [1] the factor kernels' — a factor
  // linearize between mh_photo_preprocess_scan_begin and _commit must not erase or mix into the frame's count
  mh::PhotoCounters * h_counters = nullptr;
  mh::PhotoCounters * d_counters = nullptr;
  float * h_int_out = nullptr;  // pinned staging of the corrected intensities
  size_t h_int_cap = 0;
  char * h_stage = nullptr;     // pinned staging of detectFeatures' small transfers (candidate list, gather records)
  size_t h_stage_cap = 0;
  PhotoFrame * frame = nullptr;
  PhotoFrame * next_frame = nullptr;  // built by mh_photo_preprocess_scan_begin, current after mh_photo_preprocess_commit
  mh_scan * next_scan = nullptr;      // ... the scan it was built from (its cloud receives the corrected intensities at the commit)
  hipEvent_t next_ev = nullptr;       // ... recorded behind its last kernel
  hipEvent_t scan_ev = nullptr;       // cross-stream order with the scan's context (device-side waits, no host wait)
  PhotoFrame * cand_frame = nullptr;  // mh_photo_detect_prefetch: the frame whose candidate list is on its way to h_stage (a reference)
  hipEvent_t cand_ev = nullptr;       // ... recorded behind the copies
  std::vector<HostFeature> features;  // map_Le_features_
  uint32_t next_id = 0;               // monotonic_feature_id_
};

struct mh_photo_factor
{
  mh_photo * photo = nullptr;
  PhotoFrame * frame = nullptr;
  bool binary = false;
  double VSVt[36];
  std::vector<HostFeature> features;  // a_features_
  DevBuf d_Le, d_psi, d_npts, d_rows;
  // per-feature outputs of the kernel (statuses, new centres, Hessian sums): mapped pinned host memory the kernel writes
  // straight into — ~21 KB per linearize, no copy nodes behind the kernel
  void * h_out = nullptr;
  void * d_out = nullptr;  // device address of h_out
  size_t out_bytes = 0;
  bool pending = false, pending_timed = false;  // a linearize is enqueued and not yet collected
  DevBuf d_ticket;              // one counter: blocks of the linearize kernel that have finished
  unsigned int seq_counter = 0, pending_seq = 0;  // completion number the pending call publishes (0: wait on the stream)
  std::vector<int32_t> statuses;
  std::vector<double> centers, partials, rows;
  hipEvent_t ev[2] = {nullptr, nullptr};
};

namespace
{
void photo_release(mh_photo * p)
{
  if (!p) return;
  if (p->refs.fetch_sub(1) != 1) return;
  (void)mh_enter(p->ctx);
  (void)hipStreamSynchronize(p->ctx->stream);
  frame_release(p->frame);
  frame_release(p->next_frame);
  frame_release(p->cand_frame);
  if (p->cand_ev) (void)hipEventDestroy(p->cand_ev);
  if (p->next_ev) (void)hipEventDestroy(p->next_ev);
  if (p->scan_ev) (void)hipEventDestroy(p->scan_ev);
  for (DevBuf * b : {&p->d_alt, &p->d_shift, &p->d_hp, &p->d_lp, &p->d_static, &p->d_raw_pts, &p->d_img_raw, &p->d_tmp_a, &p->d_tmp_b,
                     &p->d_mask_raw, &p->d_yaw_valid, &p->d_int_out, &p->d_grad, &p->d_detmask, &p->d_xyz, &p->d_cand, &p->d_gather, &p->d_stamps, &p->d_proj_pre})
    b->release();
  if (p->h_counters) (void)hipHostFree(p->h_counters);
  if (p->h_int_out) (void)hipHostFree(p->h_int_out);
  if (p->h_stage) (void)hipHostFree(p->h_stage);
  delete p;
}

// cv::circle(mask, centre, radius, 0, -1): the midpoint-circle spans of OpenCV's drawing.cpp Circle() (assumption O10)
void fill_circle_zero(std::vector<uint8_t> & img, int rows, int cols, int cx, int cy, int radius)
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

// centre value of cv::cornerEigenValsAndVecs(roi, 5, 3) (assumption O11) -> eigenvector of the larger eigenvalue
void patch_gradient_direction(const float * I, int cols, int x, int y, float & ix, float & iy)
{
  const float scale = static_cast<float>(1.0 / (4.0 * 5.0));
  float a = 0, b = 0, c = 0;
  for (int dy = -2; dy <= 2; ++dy)
    for (int dx = -2; dx <= 2; ++dx) {
      const float * p = I + static_cast<size_t>(y + dy) * cols + (x + dx);
      const float tl = p[-cols - 1], tc = p[-cols], tr = p[-cols + 1], ml = p[-1], mr = p[1], bl = p[cols - 1], bc = p[cols], br = p[cols + 1];
      const float gx = ((tr - tl) + 2.f * (mr - ml) + (br - bl)) * scale;
      const float gy = ((bl - tl) + 2.f * (bc - tc) + (br - tr)) * scale;
      a += gx * gx;
      b += gx * gy;
      c += gy * gy;
    }
  auto eigvec = [&](double l, float & ex, float & ey) {
    double xx = b, yy = l - a, e = std::fabs(xx);
    if (e + std::fabs(yy) < 1e-4) {
      yy = b;
      xx = l - c;
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
  const double u = (a + c) * 0.5, v = std::sqrt((a - c) * (a - c) * 0.25 + static_cast<double>(b) * b);
  const float e1 = static_cast<float>(u + v), e2 = static_cast<float>(u - v);
  float x1, y1, x2, y2;
  eigvec(u + v, x1, y1);
  eigvec(u - v, x2, y2);
  ix = e1 > e2 ? x1 : x2;
  iy = e1 > e2 ? y1 : y2;
}

void projection_jacobian(const mh::PhotoModel & m, double x, double y, double z, double H[6])  // photometric_utils.cpp:186-198
{
  const double rxy = std::sqrt(x * x + y * y);
  const double L = rxy - m.beam_offset_m;
  const double R2 = L * L + z * z;
  const double irxy = 1.0 / rxy;
  const double fx_irxy2 = m.fx * (irxy * irxy);
  H[0] = -fx_irxy2 * y;
  H[1] = fx_irxy2 * x;
  H[2] = 0;
  H[3] = -m.fy * x * z / ((L + m.beam_offset_m) * R2);
  H[4] = -m.fy * y * z / ((L + m.beam_offset_m) * R2);
  H[5] = m.fy * L / R2;
}

void get_psi(const std::vector<double> & I, double & mean, double & sigma, std::vector<double> & psi)  // photometric_utils.cpp:13-19
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

struct Pose
{
  double R[9], t[3];
};
Pose pose_mul(const Pose & a, const Pose & b)
{
  Pose r;
  mh::mat3_mul(a.R, b.R, r.R);
  for (int i = 0; i < 3; ++i) r.t[i] = a.t[i] + (a.R[3 * i] * b.t[0] + (a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2]));
  return r;
}
Pose pose_inv(const Pose & a)
{
  Pose r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.R[3 * i + j] = a.R[3 * j + i];
  for (int i = 0; i < 3; ++i) r.t[i] = -(r.R[3 * i] * a.t[0] + (r.R[3 * i + 1] * a.t[1] + r.R[3 * i + 2] * a.t[2]));
  return r;
}
void pose_act(const Pose & a, const double p[3], double o[3])
{
  for (int i = 0; i < 3; ++i) o[i] = (a.R[3 * i] * p[0] + (a.R[3 * i + 1] * p[1] + a.R[3 * i + 2] * p[2])) + a.t[i];
}

bool mat6_inv(const double * A, double * inv)  // M66::inverse() (Eigen: partial-pivoting LU)
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
      const double mlt = a[r][c];
      for (int j = 0; j < 12; ++j) a[r][j] -= mlt * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) inv[6 * i + j] = a[i][6 + j];
  return true;
}
void mat6_mul(const double * A, const double * B, double * C)
{
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += A[6 * i + k] * B[6 * k + j];
      C[6 * i + j] = s;
    }
}

int upload(mh_ctx * ctx, DevBuf & b, const void * src, size_t bytes)
{
  MH_HIP(ctx, b.reserve(bytes ? bytes : 16, ctx->stream, false));
  if (bytes) MH_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return MH_OK;
}
// The same out of a pinned block of this library (readable up to the next multiple of 16 bytes), as a copy KERNEL: a small
// hipMemcpyAsync can block its caller behind another thread's big upload (mh_scan_deskew has the measurement).
int upload_pinned(mh_ctx * ctx, DevBuf & b, const void * pinned_src, size_t bytes)
{
  MH_HIP(ctx, b.reserve(bytes + 16, ctx->stream, false));
  if (!bytes) return MH_OK;
  void * d_src = nullptr;
  MH_HIP(ctx, hipHostGetDevicePointer(&d_src, const_cast<void *>(pinned_src), 0));
  MH_HIP(ctx, mh::launch_copy16(d_src, b.p, (bytes + 15) & ~size_t(15), ctx->stream));
  return MH_OK;
}

// The device part of preprocess on points already resident (d_raw / frame->d_points), then the pose table.
// scan_cloud: the points are a scan's resident deskewed cloud — the frame's copy (frame->d_points) is still to be made;
// scan_writeback: the corrected intensities also go straight into scan_cloud (a committing mh_photo_preprocess_scan).
// *folded says whether both were done inside the chain's launches (else the caller copies first / writes back afterwards).
int preprocess_device(mh_photo * ph, PhotoFrame * fr, const mh_point32 * d_raw, size_t n, const uint32_t * unique_ns,
                      const double * T_Le_Lt, size_t n_groups, mh_point32 * scan_cloud = nullptr, bool scan_writeback = false, bool * folded = nullptr)
{
  if (folded) *folded = false;
  mh_ctx * ctx = ph->ctx;
  const mh::PhotoModel & m = ph->model;
  const int rows = m.rows, cols = m.cols, npx = rows * cols;
  const size_t fb = static_cast<size_t>(npx) * sizeof(float);
  for (DevBuf * b : {&fr->d_intensity, &fr->d_range, &fr->d_dx, &fr->d_dy, &fr->d_yaw, &ph->d_img_raw, &ph->d_tmp_a, &ph->d_tmp_b})
    MH_HIP(ctx, b->reserve(fb, ctx->stream, false));
  MH_HIP(ctx, fr->d_idx.reserve(static_cast<size_t>(npx) * sizeof(int32_t), ctx->stream, false));
  MH_HIP(ctx, fr->d_proj.reserve(static_cast<size_t>(npx) * mh::kPhotoDup * sizeof(int32_t), ctx->stream, false));
  MH_HIP(ctx, fr->d_mask.reserve(npx, ctx->stream, false));
  MH_HIP(ctx, ph->d_mask_raw.reserve(npx, ctx->stream, false));
  MH_HIP(ctx, ph->d_yaw_valid.reserve(npx, ctx->stream, false));
  MH_HIP(ctx, ph->d_int_out.reserve((n ? n : 1) * sizeof(float), ctx->stream, false));
  const void * pose_src = nullptr;
  size_t pose_bytes = 0;
  {
    const size_t b_ns = (n_groups * sizeof(uint32_t) + 255) & ~size_t(255), b_rt = n_groups * 12 * sizeof(double);
    size_t cap = size_t(64) << 10;
    while (cap < b_ns + b_rt) cap <<= 1;
    MH_HIP(ctx, AllocCache::alloc_pinned(&fr->h_pose, cap));
    fr->h_pose_cap = cap;
    char * h = static_cast<char *>(fr->h_pose);
    if (n_groups) {
      std::memcpy(h, unique_ns, n_groups * sizeof(uint32_t));
      std::memcpy(h + b_ns, T_Le_Lt, b_rt);
    }
    // copy kernels reading the mapped block (see mh_scan_deskew: a small hipMemcpyAsync can block behind another thread's upload)
    void * d_src = nullptr;
    MH_HIP(ctx, hipHostGetDevicePointer(&d_src, h, 0));
    // one block [timestamps | poses] on the device as in the pinned buffer: ONE copy launch
    MH_HIP(ctx, fr->d_pose_ns.reserve(b_ns + b_rt + 32, ctx->stream, false));
    fr->pose_rt_offset = b_ns;
    pose_src = d_src;
    pose_bytes = b_ns + ((b_rt + 15) & ~size_t(15));  // copied by the frame-reset launch below
  }
  fr->n_poses = static_cast<int>(n_groups);
  fr->rows = rows;
  fr->cols = cols;
  fr->n_points = n;
  auto * desk = static_cast<mh_point32 *>(fr->d_points.p);
  float * img_raw = static_cast<float *>(ph->d_img_raw.p);
  ph->h_counters->project_throw = ph->h_counters->pose_missing = 0;
  const int ni = static_cast<int>(n);
  // behind the scatter: the image chain (photometric.cpp:246-320), the mask erosion, the yaw table and the projection index in
  // the scatter + THREE multi-job launches (photo_kernels.hip, "Round 4": 13 launches -> 5; round 5: no reset launch, 4), or the single-stage kernels when a filter is
  // larger than the tiles were sized for (and with MH_PHOTO_UNFUSED=1, diagnostic)
  const mh_photo_config & c = ph->cfg;
  const float scale = static_cast<float>(static_cast<double>(c.intensity_scale)), gamma = c.intensity_gamma;
  float * ta = static_cast<float *>(ph->d_tmp_a.p);
  float * tb = static_cast<float *>(ph->d_tmp_b.p);
  float * fin = static_cast<float *>(fr->d_intensity.p);
  mh::PhotoChain pc;
  pc.raw = img_raw;
  pc.ta = ta;
  pc.tb = tb;
  pc.fin = fin;
  pc.dx = static_cast<float *>(fr->d_dx.p);
  pc.dy = static_cast<float *>(fr->d_dy.p);
  pc.idx = static_cast<const int32_t *>(fr->d_idx.p);
  pc.intensity_out = static_cast<float *>(ph->d_int_out.p);
  pc.hp = static_cast<const float *>(ph->d_hp.p);
  pc.lp = static_cast<const float *>(ph->d_lp.p);
  pc.static_mask = ph->static_mask.empty() ? nullptr : static_cast<const uint8_t *>(ph->d_static.p);
  pc.mask_out = static_cast<uint8_t *>(fr->d_mask.p);
  pc.yaw = static_cast<float *>(fr->d_yaw.p);
  pc.desk_points = desk;
  pc.desk_src = nullptr;
  pc.desk_writeback = nullptr;
  pc.proj_pre = nullptr;
  pc.proj = static_cast<int32_t *>(fr->d_proj.p);
  pc.counters = ph->d_counters;
  pc.rows = rows;
  pc.cols = cols;
  pc.n_pts = ni;
  pc.n_hp = c.n_high_pass;
  pc.n_lp = c.n_low_pass;
  pc.remove_lines = c.remove_lines ? 1 : 0;
  pc.filter_brightness = c.filter_brightness ? 1 : 0;
  pc.bw = c.brightness_window_size[0];
  pc.bh = c.brightness_window_size[1];
  pc.do_gauss = c.gaussian_blur ? 1 : 0;
  pc.erode_k = c.patch_size + c.erosion_buffer;
  pc.scale = scale;
  pc.gamma = gamma;
  static const bool unfused = std::getenv("MH_PHOTO_UNFUSED") != nullptr;
  if (!unfused && mh::photo_stages_fit(pc)) {
    // four launches, no reset: the marks are frame stamps (zero when allocated, so no frame ever carries the number 0)
    const size_t sb = static_cast<size_t>(npx) * 2 * sizeof(uint32_t);
    MH_HIP(ctx, ph->d_stamps.reserve(sb, ctx->stream, false));
    if (ph->d_stamps.p != ph->stamps_zeroed_at || ph->stamps_zeroed_bytes < sb || ph->frame_seq == 0xFFFFFFFFu) {
      MH_HIP(ctx, hipMemsetAsync(ph->d_stamps.p, 0, sb, ctx->stream));
      ph->stamps_zeroed_at = ph->d_stamps.p;
      ph->stamps_zeroed_bytes = sb;
      ph->frame_seq = 0;
    }
    MH_HIP(ctx, ph->d_proj_pre.reserve((n ? n : 1) * mh::kPhotoProjPreBytes, ctx->stream, false));
    pc.proj_pre = ph->d_proj_pre.p;
    pc.stamps = static_cast<uint32_t *>(ph->d_stamps.p);
    pc.seq = ++ph->frame_seq;
    pc.raw_points = d_raw;
    pc.raw_w = img_raw;
    pc.range = static_cast<float *>(fr->d_range.p);
    pc.idx_w = static_cast<int32_t *>(fr->d_idx.p);
    pc.copy_src = pose_src;
    pc.copy_dst = fr->d_pose_ns.p;
    pc.copy_bytes = pose_bytes;
    pc.desk_src = scan_cloud;
    pc.desk_writeback = scan_writeback ? scan_cloud : nullptr;
    MH_HIP(ctx, mh::launch_photo_stages(pc, m, ctx->stream));
    if (folded) *folded = true;
    return MH_OK;
  }
  if (scan_cloud && n) MH_HIP(ctx, mh::launch_copy16(scan_cloud, fr->d_points.p, n * sizeof(mh_point32), ctx->stream));
  // the single-stage kernels (a filter larger than the stage tiles, or MH_PHOTO_UNFUSED=1): reset, scatter, one launch per stage
  MH_HIP(ctx, mh::launch_photo_clear(npx, static_cast<int>(n), img_raw, static_cast<float *>(fr->d_range.p),
                                     static_cast<uint8_t *>(ph->d_mask_raw.p), static_cast<uint8_t *>(ph->d_yaw_valid.p),
                                     static_cast<int32_t *>(fr->d_idx.p), static_cast<int32_t *>(fr->d_proj.p),
                                     static_cast<float *>(ph->d_int_out.p), pose_src, fr->d_pose_ns.p, pose_bytes, ctx->stream));
  MH_HIP(ctx, mh::launch_photo_scatter(m, d_raw, desk, ni, static_cast<float *>(fr->d_yaw.p), static_cast<uint8_t *>(ph->d_yaw_valid.p),
                                       img_raw, static_cast<float *>(fr->d_range.p), static_cast<uint8_t *>(ph->d_mask_raw.p),
                                       static_cast<int32_t *>(fr->d_idx.p), ctx->stream));
  MH_HIP(ctx, mh::launch_photo_yaw_fill(m, static_cast<float *>(fr->d_yaw.p), static_cast<const uint8_t *>(ph->d_yaw_valid.p), ctx->stream));
  MH_HIP(ctx, mh::launch_photo_project(m, desk, ni, static_cast<const float *>(fr->d_yaw.p), static_cast<int32_t *>(fr->d_proj.p),
                                       ph->d_counters, ctx->stream));
  MH_HIP(ctx, mh::launch_photo_proj_finalize(npx, static_cast<int32_t *>(fr->d_proj.p), ctx->stream));
  const float * cur;
  if (c.remove_lines) {
    MH_HIP(ctx, mh::launch_photo_vfir(img_raw, ta, rows, cols, static_cast<const float *>(ph->d_hp.p), c.n_high_pass, scale, gamma, ctx->stream));
    MH_HIP(ctx, mh::launch_photo_hfir_sub(ta, img_raw, tb, rows, cols, static_cast<const float *>(ph->d_lp.p), c.n_low_pass, scale, gamma,
                                          ctx->stream));
    cur = tb;
  } else {
    MH_HIP(ctx, mh::launch_photo_scale(img_raw, tb, npx, scale, gamma, ctx->stream));
    cur = tb;
  }
  if (c.filter_brightness) {
    MH_HIP(ctx, mh::launch_photo_brightness(cur, ta, rows, cols, c.brightness_window_size[0], c.brightness_window_size[1], ctx->stream));
    cur = ta;
  }
  MH_HIP(ctx, mh::launch_photo_gauss_trunc(cur, fin, rows, cols, c.gaussian_blur ? 1 : 0, ctx->stream));
  MH_HIP(ctx, mh::launch_photo_sobel_writeback(fin, static_cast<float *>(fr->d_dx.p), static_cast<float *>(fr->d_dy.p),
                                               static_cast<const int32_t *>(fr->d_idx.p), nullptr, static_cast<float *>(ph->d_int_out.p),
                                               rows, cols, ctx->stream));
  MH_HIP(ctx, mh::launch_photo_erode(static_cast<const uint8_t *>(ph->d_mask_raw.p),
                                     ph->static_mask.empty() ? nullptr : static_cast<const uint8_t *>(ph->d_static.p), -1,
                                     static_cast<uint8_t *>(fr->d_mask.p), rows, cols, c.patch_size + c.erosion_buffer, ctx->stream));
  return MH_OK;
}

// grow-only pinned staging block of a photo object (pageable transfers go through the runtime's own staging, one extra
// synchronisation each)
int photo_stage(mh_photo * ph, size_t bytes)
{
  if (ph->h_stage_cap >= bytes) return MH_OK;
  if (ph->h_stage) (void)hipHostFree(ph->h_stage);
  ph->h_stage = nullptr;
  ph->h_stage_cap = 0;
  const size_t cap = (bytes + bytes / 2 + 4095) & ~size_t(4095);
  MH_HIP(ph->ctx, hipHostMalloc(reinterpret_cast<void **>(&ph->h_stage), cap, hipHostMallocDefault));
  ph->h_stage_cap = cap;
  return MH_OK;
}

// photometric_utils.cpp:453-483 — the nearest free pixel (one of the 8 neighbours when the rounded one is taken)
std::pair<int, int> snap_point(const std::pair<double, double> & p, std::set<std::pair<int, int>> & used)
{
  const int gx = static_cast<int>(std::round(p.first)), gy = static_cast<int>(std::round(p.second));
  const std::pair<int, int> first{gx, gy};
  if (used.insert(first).second) return first;
  double best_d = std::numeric_limits<double>::max();
  std::pair<int, int> best = first;
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy) {
      if (dx == 0 && dy == 0) continue;
      const std::pair<int, int> alt{gx + dx, gy + dy};
      if (used.count(alt)) continue;
      const double ex = alt.first - p.first, ey = alt.second - p.second;
      const double d = std::sqrt(ex * ex + ey * ey);
      if (d < best_d) {
        best_d = d;
        best = alt;
      }
    }
  used.insert(best);
  return best;
}

// photometric_utils.cpp:485-518 — getGradientBasedLocations: the pattern in the (edge normal, edge tangent) frame of the
// candidate, snapped to distinct pixels (rotate_patch_to_align_with_gradient, photometric.cpp:659-684)
void gradient_based_locations(float gx, float gy, const std::vector<int32_t> & pattern, int32_t * out)
{
  const double norm = std::sqrt(gx * gx + gy * gy) + 1e-6;  // float products and float sqrt, as there
  const double nx = -gy / norm, ny = gx / norm, tx = gx / norm, ty = gy / norm;
  std::set<std::pair<int, int>> used;
  for (size_t o = 0; o < pattern.size() / 2; ++o) {
    const double x = pattern[2 * o], y = pattern[2 * o + 1];
    const std::pair<int, int> g = snap_point({nx * x + tx * y, ny * x + ty * y}, used);
    out[2 * o] = g.first;
    out[2 * o + 1] = g.second;
  }
}

// The part of detectFeatures that depends on the frame alone (photometric.cpp:524-555 before the per-feature circles):
// gradient magnitude, detection mask = erode(img_mask & mask_margin_), candidate pixels (mask != 0, gradient > threshold)
// compacted in row-major order; the count and a 64 K-entry prefix of the list are on their way to h_stage when this returns
// (~100 KB come back instead of the gradient / mask / intensity / index planes and the cloud, 5.3 MB).
int detect_enqueue_candidates(mh_photo * ph, PhotoFrame * fr)
{
  mh_ctx * ctx = ph->ctx;
  const mh_photo_config & c = ph->cfg;
  const int rows = c.rows, cols = c.cols, npx = rows * cols;
  MH_HIP(ctx, ph->d_grad.reserve(npx, ctx->stream, false));
  MH_HIP(ctx, ph->d_detmask.reserve(npx, ctx->stream, false));
  MH_HIP(ctx, mh::launch_photo_grad(static_cast<const float *>(fr->d_dx.p), static_cast<const float *>(fr->d_dy.p),
                                    static_cast<uint8_t *>(ph->d_grad.p), npx, ctx->stream));
  MH_HIP(ctx, mh::launch_photo_erode(static_cast<const uint8_t *>(fr->d_mask.p), nullptr, c.margin_size, static_cast<uint8_t *>(ph->d_detmask.p),
                                     rows, cols, c.patch_size + c.erosion_buffer, ctx->stream));
  const int n_blk = (npx + 255) / 256;
  MH_HIP(ctx, ph->d_cand.reserve((static_cast<size_t>(npx) + n_blk + 4) * sizeof(uint32_t), ctx->stream, false));
  uint32_t * d_list = static_cast<uint32_t *>(ph->d_cand.p), * d_blk = d_list + npx, * d_n = d_blk + n_blk;
  MH_HIP(ctx, mh::launch_photo_candidates(static_cast<const uint8_t *>(ph->d_grad.p), static_cast<const uint8_t *>(ph->d_detmask.p), npx,
                                          c.gradient_threshold, d_blk, d_list, d_n, ctx->stream));
  // the count and a prefix of the list come back together (one wait; a longer list needs a second copy)
  const size_t prefix = std::min<size_t>(static_cast<size_t>(npx), 65536);
  {
    const int rcs = photo_stage(ph, 256 + static_cast<size_t>(npx) * sizeof(uint32_t));
    if (rcs != MH_OK) return rcs;
  }
  // copy KERNELS writing the pinned block (small hipMemcpyAsync calls can block their caller behind another thread's upload,
  // see mh_scan_deskew)
  void * hd = nullptr;
  MH_HIP(ctx, hipHostGetDevicePointer(&hd, ph->h_stage, 0));
  {
    // d_n's 16-byte neighbourhood lands at the start of the block; the reader picks the word at the same offset
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_n) & ~uintptr_t(15);
    MH_HIP(ctx, mh::launch_copy16(reinterpret_cast<const void *>(a), hd, 16, ctx->stream));
    MH_HIP(ctx, mh::launch_copy16(d_list, static_cast<char *>(hd) + 256, (prefix * sizeof(uint32_t) + 15) & ~size_t(15), ctx->stream));
  }
  return MH_OK;
}

int detect_features_impl(mh_photo * ph, int num_to_detect, const double R_W_Be[9], const double t_W_Be[3], const double * bias,
                         size_t n_dirs)
{
  if (num_to_detect <= 0) return MH_OK;  // photometric.cpp:522
  mh_ctx * ctx = ph->ctx;
  PhotoFrame * fr = ph->frame;
  // MH_DETECT_TRACE=1: wall time of the stages of this call on stderr (tools/detect_time.py)
  static const bool trace = std::getenv("MH_DETECT_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t_prev = std::chrono::steady_clock::now();
  double t_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto lap = [&](int i) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    t_stage[i] += std::chrono::duration<double, std::micro>(now - t_prev).count();
    t_prev = now;
  };
  if (!fr) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_detect_features: no frame (call mh_photo_preprocess first)");
  const mh_photo_config & c = ph->cfg;
  const int rows = c.rows, cols = c.cols, npx = rows * cols;
  const size_t prefix = std::min<size_t>(static_cast<size_t>(npx), 65536);
  if (ph->cand_frame == fr) {
    // mh_photo_detect_prefetch enqueued this frame's candidate pass earlier: only its copies are waited for
    MH_HIP(ctx, hipEventSynchronize(ph->cand_ev));
  } else {
    const int rcq = detect_enqueue_candidates(ph, fr);
    if (rcq != MH_OK) return rcq;
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  frame_release(ph->cand_frame);
  ph->cand_frame = nullptr;
  uint32_t * h_list = reinterpret_cast<uint32_t *>(ph->h_stage + 256);
  const uint32_t n_list = [&] {  // the count's 16-byte neighbourhood was copied: same offset inside it
    const int n_blk_ = (npx + 255) / 256;
    const uintptr_t dn = reinterpret_cast<uintptr_t>(static_cast<uint32_t *>(ph->d_cand.p) + npx + n_blk_);
    return reinterpret_cast<const uint32_t *>(ph->h_stage)[(dn & 15u) / 4];
  }();
  if (n_list > prefix) {
    const uint32_t * d_all = static_cast<const uint32_t *>(ph->d_cand.p);
    MH_HIP(ctx, hipMemcpyAsync(h_list + prefix, d_all + prefix, (n_list - prefix) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  lap(0);  // device: gradient, erosion, compaction; list read-back
  std::vector<uint32_t> gradients(h_list, h_list + n_list);
  // the detection mask is only ever asked about candidate pixels (set by construction): `alive` carries the circles
  std::vector<uint8_t> alive(static_cast<size_t>(npx), 1);
  for (const HostFeature & ft : ph->features)  // :526-530
    fill_circle_zero(alive, rows, cols, static_cast<int>(ft.hdr.center[0]), static_cast<int>(ft.hdr.center[1]), c.nma_radius);
  // the reference zeroes the circles in the mask BEFORE it collects the candidates (:526-555): the sequence handed to
  // std::sort must be exactly that one
  gradients.erase(std::remove_if(gradients.begin(), gradients.end(), [&](uint32_t g) { return !alive[static_cast<size_t>(g & 0xFFFFFFu)]; }),
                  gradients.end());
  // :556-560 — std::sort with a comparator on the gradient only: the order of equal gradients is libstdc++'s (it depends
  // on the sequence of comparison outcomes alone, so sorting the packed words moves them exactly like the reference's pairs)
  // For a long list the same introsort runs on four host threads (exact_sort.hpp: same pivots, same partitions, same final
  // insertion sort — element for element std::sort's result, tests/cpp/exact_sort_check.cpp); MH_SORT_THREADS=1 forces
  // the plain library call.
  lap(1);  // circles of the tracked features, filter
  {
    auto by_gradient = [](uint32_t a, uint32_t b) { return (a >> 24) > (b >> 24); };
    static const int sort_threads = [] {
      const char * e = std::getenv("MH_SORT_THREADS");
      const int hw = static_cast<int>(std::thread::hardware_concurrency());
      const int want = e ? std::atoi(e) : 4;
      return std::max(1, std::min(want, hw > 0 ? hw : 1));
    }();
    if (sort_threads > 1 && gradients.size() >= 16384) {
      static mh::exact_sort::Pool pool(sort_threads - 1);  // helpers that stay around between frames (joined at exit)
      static const pid_t pool_owner = getpid();            // threads do not survive fork(): a child sorts with threads of its own
      mh::exact_sort::sort_parallel(gradients.data(), gradients.data() + gradients.size(), by_gradient, sort_threads, 4096,
                                    getpid() == pool_owner ? &pool : nullptr);
    } else
      std::sort(gradients.begin(), gradients.end(), by_gradient);
  }
  lap(2);  // sort
  std::vector<std::pair<int, int>> cand;
  for (const uint32_t g : gradients) {  // :565-571 non-maximum suppression
    const int px = static_cast<int>(g & 0xFFFFFFu);
    if (!alive[static_cast<size_t>(px)]) continue;
    cand.emplace_back(px % cols, px / cols);
    fill_circle_zero(alive, rows, cols, px % cols, px / cols, c.nma_radius);
  }
  lap(3);  // non-maximum suppression
  // what the selection below reads, gathered on the device for the surviving candidates only
  const int m_off = c.n_patch_offsets, n_cand = static_cast<int>(cand.size()), per = m_off + 1;
  std::vector<float> win, rec;
  std::vector<int32_t> rec_idx;
  // uv (offsets + centres, see photo_gather_kernel) goes up, the windows / patch records / point indices of `count` candidates
  // come back: one upload, one kernel, ONE download, through the pinned staging block
  auto gather = [&](const std::vector<int32_t> & uv, int count, bool per_candidate) -> int {
    const size_t n_win = static_cast<size_t>(count) * 49, n_rec = static_cast<size_t>(count) * per * 4, n_idx = static_cast<size_t>(count) * per;
    const size_t b_uv = (uv.size() * 4 + 255) & ~size_t(255), b_win = (n_win * 4 + 255) & ~size_t(255), b_rec = (n_rec * 4 + 255) & ~size_t(255);
    const size_t b_out = b_win + b_rec + n_idx * 4;
    MH_HIP(ctx, ph->d_gather.reserve(b_uv + b_out + 256, ctx->stream, false));
    const int rcs = photo_stage(ph, b_uv + b_out + 16);
    if (rcs != MH_OK) return rcs;
    char * d = static_cast<char *>(ph->d_gather.p);
    std::memcpy(ph->h_stage, uv.data(), uv.size() * 4);
    {
      void * d_src = nullptr;  // (a copy kernel, see upload_pinned; b_uv is a multiple of 256 on both sides)
      MH_HIP(ctx, hipHostGetDevicePointer(&d_src, ph->h_stage, 0));
      MH_HIP(ctx, mh::launch_copy16(d_src, d, (uv.size() * 4 + 15) & ~size_t(15), ctx->stream));
    }
    MH_HIP(ctx, mh::launch_photo_gather(reinterpret_cast<const int2 *>(d), m_off, count, per_candidate, static_cast<const float *>(fr->d_intensity.p),
                                        static_cast<const int32_t *>(fr->d_idx.p), static_cast<const mh_point32 *>(fr->d_points.p), rows, cols,
                                        reinterpret_cast<float *>(d + b_uv), reinterpret_cast<float4 *>(d + b_uv + b_win),
                                        reinterpret_cast<int32_t *>(d + b_uv + b_win + b_rec), ctx->stream));
    {
      void * hd = nullptr;  // (a copy kernel writing the pinned block, see detect_enqueue_candidates)
      MH_HIP(ctx, hipHostGetDevicePointer(&hd, ph->h_stage, 0));
      MH_HIP(ctx, mh::launch_copy16(d + b_uv, static_cast<char *>(hd) + b_uv, (b_out + 15) & ~size_t(15), ctx->stream));
    }
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const char * h = ph->h_stage + b_uv;
    win.assign(reinterpret_cast<const float *>(h), reinterpret_cast<const float *>(h) + n_win);
    rec.assign(reinterpret_cast<const float *>(h + b_win), reinterpret_cast<const float *>(h + b_win) + n_rec);
    rec_idx.assign(reinterpret_cast<const int32_t *>(h + b_win + b_rec), reinterpret_cast<const int32_t *>(h + b_win + b_rec) + n_idx);
    return MH_OK;
  };
  if (n_cand) {
    std::vector<int32_t> uv(2 * static_cast<size_t>(m_off + n_cand));
    std::memcpy(uv.data(), ph->offsets.data(), 2 * static_cast<size_t>(m_off) * sizeof(int32_t));
    for (int i = 0; i < n_cand; ++i) {
      uv[2 * (m_off + i)] = cand[i].first;
      uv[2 * (m_off + i) + 1] = cand[i].second;
    }
    const int rcg = gather(uv, n_cand, false);
    if (rcg != MH_OK) return rcg;
  }
  lap(4);  // gather round trip
  // :575-625 scores of every candidate along every bias direction
  std::vector<std::vector<std::pair<double, int>>> scores(n_dirs, std::vector<std::pair<double, int>>(cand.size(), {0.0, 0}));
  std::vector<float> grad_dir(2 * cand.size());
  for (size_t i = 0; i < cand.size(); ++i) {
    float ix, iy;
    patch_gradient_direction(&win[i * 49], 7, 3, 3, ix, iy);
    grad_dir[2 * i] = ix;
    grad_dir[2 * i + 1] = iy;
    const size_t ctr = i * per + m_off;
    if (rec_idx[ctr] < 0) continue;
    double P[6];
    projection_jacobian(ph->model, rec[4 * ctr], rec[4 * ctr + 1], rec[4 * ctr + 2], P);
    for (size_t b = 0; b < n_dirs; ++b) {
      const double * d = bias + 3 * b;
      double w0 = P[0] * d[0] + P[1] * d[1] + P[2] * d[2], w1 = P[3] * d[0] + P[4] * d[1] + P[5] * d[2];
      const double nn = std::sqrt(w0 * w0 + w1 * w1);
      if (nn > 0) {
        w0 /= nn;
        w1 /= nn;
      }
      scores[b][i] = {std::fabs(ix * w0 + iy * w1), static_cast<int>(i)};
    }
  }
  for (size_t b = 0; b < n_dirs; ++b)
    std::sort(scores[b].begin(), scores[b].end(), [](const std::pair<double, int> & a, const std::pair<double, int> & bb) { return a.first > bb.first; });
  std::vector<int> selected;  // :639-651 round-robin over the directions; "not selected yet" (a std::find there) is a flag here
  std::vector<uint8_t> taken(cand.size(), 0);
  if (n_dirs)
    for (size_t i = 0; i < scores[0].size(); ++i)
      for (size_t b = 0; b < n_dirs; ++b) {
        const int k = scores[b][i].second;
        if (!taken[static_cast<size_t>(k)]) {
          taken[static_cast<size_t>(k)] = 1;
          selected.push_back(k);
        }
      }
  // rotate_patch_to_align_with_gradient (:659-684): every selected candidate samples its own rotated pattern — a second,
  // per-candidate gather of just those (a handful of) candidates replaces the records of the fixed pattern
  std::vector<size_t> rec_at(cand.size());  // where candidate k's patch records start
  for (size_t k = 0; k < cand.size(); ++k) rec_at[k] = k * per;
  if (c.rotate_patch_to_align_with_gradient && !selected.empty()) {
    const int n_sel = static_cast<int>(selected.size());
    std::vector<int32_t> uv(2 * static_cast<size_t>(n_sel) * (m_off + 1));
    for (int j = 0; j < n_sel; ++j) {
      const int k = selected[j];
      gradient_based_locations(grad_dir[2 * k], grad_dir[2 * k + 1], ph->offsets, &uv[2 * static_cast<size_t>(j) * m_off]);
      uv[2 * (static_cast<size_t>(n_sel) * m_off + j)] = cand[k].first;
      uv[2 * (static_cast<size_t>(n_sel) * m_off + j) + 1] = cand[k].second;
      rec_at[k] = static_cast<size_t>(j) * per;
    }
    const int rcg = gather(uv, n_sel, true);
    if (rcg != MH_OK) return rcg;
  }
  lap(5);  // scores, per-direction sorts, round-robin selection
  Pose TBL, TWB;
  std::memcpy(TBL.R, c.T_B_L_R, sizeof(TBL.R));
  std::memcpy(TBL.t, c.T_B_L_t, sizeof(TBL.t));
  std::memcpy(TWB.R, R_W_Be, sizeof(TWB.R));
  std::memcpy(TWB.t, t_W_Be, sizeof(TWB.t));
  const Pose T_map = pose_mul(pose_mul(pose_inv(TBL), TWB), TBL);  // :666-667
  int num_added = 0;
  for (const int k : selected) {
    const int lx = cand[k].first, ly = cand[k].second;
    HostFeature ft;
    ft.hdr.id = ph->next_id++;
    ft.hdr.life_time = 1;
    ft.hdr.center[0] = lx;
    ft.hdr.center[1] = ly;
    const int m = c.n_patch_offsets;
    ft.hdr.n_points = m;
    bool missing = false;
    for (int o = 0; o < m; ++o) {
      const size_t at = rec_at[k] + o;  // pixel (lx, ly) + patch offset o
      if (rec_idx[at] < 0) {  // the reference indexes the cloud with -1 here (undefined behaviour); the eroded mask makes it
                              // unreachable for the plain pattern, a rotated one can get there when erosion_buffer is small
        missing = true;
        break;
      }
      const double q[3] = {rec[4 * at], rec[4 * at + 1], rec[4 * at + 2]};
      double w[3];
      pose_act(T_map, q, w);
      ft.Le_ps.insert(ft.Le_ps.end(), {w[0], w[1], w[2]});
      ft.intensities.push_back(rec[4 * at + 3]);
    }
    if (missing) continue;
    double mean[3] = {0, 0, 0};
    for (int i = 0; i < m; ++i)
      for (int a = 0; a < 3; ++a) mean[a] += ft.Le_ps[3 * i + a];
    for (double & v : mean) v /= static_cast<double>(m);
    bool far = false;
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < m; ++i) {
      const double d[3] = {ft.Le_ps[3 * i] - mean[0], ft.Le_ps[3 * i + 1] - mean[1], ft.Le_ps[3 * i + 2] - mean[2]};
      if (std::sqrt(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2])) > c.max_dist_from_mean) far = true;
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) cov[3 * r + cc] += d[r] * d[cc];
    }
    if (far) continue;  // :685-687
    for (double & v : cov) v /= static_cast<double>(m - 1);
    double ev[3], E[9];
    mh::sym_eigen3(cov, ev, E);
    double nrm[3] = {E[0], E[3], E[6]};  // eigenvectors().col(0)
    bool off_plane = false;
    for (int i = 0; i < m; ++i) {
      const double d = (ft.Le_ps[3 * i] - mean[0]) * nrm[0] + ((ft.Le_ps[3 * i + 1] - mean[1]) * nrm[1] + (ft.Le_ps[3 * i + 2] - mean[2]) * nrm[2]);
      if (std::fabs(d) > c.max_dist_from_plane) off_plane = true;
    }
    if (off_plane) continue;  // :694-696
    const double mn = std::sqrt(mean[0] * mean[0] + (mean[1] * mean[1] + mean[2] * mean[2]));
    if ((nrm[0] * mean[0] + (nrm[1] * mean[1] + nrm[2] * mean[2])) / mn > 0)
      for (double & v : nrm) v = -v;  // :699-701 the normal points towards the sensor
    std::memcpy(ft.hdr.normal, nrm, sizeof(nrm));
    get_psi(ft.intensities, ft.hdr.mean_intensity, ft.hdr.sigma_intensity, ft.psi);
    ph->features.push_back(std::move(ft));
    num_added++;
    if (num_added >= num_to_detect) break;
  }
  lap(6);  // feature construction
  if (trace)
    std::fprintf(stderr, "detect_features: device+readback %.0f  filter %.0f  sort %.0f  nms %.0f  gather %.0f  select %.0f  build %.0f us  (%zu candidates, %zu after nms, %d added)\n",
                 t_stage[0], t_stage[1], t_stage[2], t_stage[3], t_stage[4], t_stage[5], t_stage[6], gradients.size(), cand.size(), num_added);
  return MH_OK;
}
}  // namespace

extern "C" {

int mh_photo_create(mh_ctx * ctx, const mh_photo_config * cfg, mh_photo ** out)
{
  if (!ctx || !cfg || !out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_create: NULL argument");
  *out = nullptr;
  return guarded(ctx, "mh_photo_create", [&]() -> int {
    if (cfg->rows < 1 || cfg->cols < 16 || cfg->cols > 4096) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_photo_create: rows >= 1, 16 <= cols <= 4096");
    // feature candidates are packed as pixel | gradient << 24 (photo_kernels.hip: photo_cand_scatter): 2^24 pixels at most
    if (cfg->rows > 4096 || static_cast<int64_t>(cfg->rows) * cfg->cols > (int64_t(1) << 24))
      return fail(ctx, MH_ERR_UNSUPPORTED, "mh_photo_create: rows <= 4096 and rows * cols <= 2^24");
    if (!cfg->pixel_shift_by_row || !cfg->beam_altitude_angles) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_create: NULL table");
    for (int r = 0; r < cfg->rows; ++r)
      if (cfg->pixel_shift_by_row[r] <= -cfg->cols || cfg->pixel_shift_by_row[r] >= cfg->cols)
        return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_create: |pixel_shift_by_row| must be below cols");
    if (cfg->rows < 2) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_create: at least two beams");
    if (cfg->n_patch_offsets < 2 || cfg->n_patch_offsets > mh::kPhotoMaxPatch || !cfg->patch_offsets)
      return fail(ctx, MH_ERR_UNSUPPORTED, "mh_photo_create: 2..64 patch offsets");
    if (cfg->remove_lines && (cfg->n_high_pass < 1 || cfg->n_low_pass < 1 || cfg->n_high_pass > mh::kPhotoMaxTaps || cfg->n_low_pass > mh::kPhotoMaxTaps ||
                              !(cfg->n_high_pass & 1) || !(cfg->n_low_pass & 1) || !cfg->high_pass_fir || !cfg->low_pass_fir))
      return fail(ctx, MH_ERR_UNSUPPORTED, "mh_photo_create: FIR kernels must have an odd length <= 129");
    if (cfg->gaussian_blur && cfg->gaussian_blur_size != 3) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_photo_create: gaussian_blur_size must be 3");
    if (cfg->filter_brightness && (cfg->brightness_window_size[0] < 1 || cfg->brightness_window_size[0] > 63 || cfg->brightness_window_size[1] < 1 ||
                                   cfg->brightness_window_size[1] > 31))
      return fail(ctx, MH_ERR_UNSUPPORTED, "mh_photo_create: brightness window up to 63 x 31");
    const int ek = cfg->patch_size + cfg->erosion_buffer;
    if (ek < 1 || ek > 33) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_photo_create: patch_size + erosion_buffer must be in 1..33");
    if (!(cfg->range_min < cfg->range_max)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_create: range_min < range_max");
    MH_HIP(ctx, mh_enter(ctx));
    mh_photo * p = new mh_photo;
    p->ctx = ctx;
    p->cfg = *cfg;
    p->shift.assign(cfg->pixel_shift_by_row, cfg->pixel_shift_by_row + cfg->rows);
    p->alt.assign(cfg->beam_altitude_angles, cfg->beam_altitude_angles + cfg->rows);
    p->offsets.assign(cfg->patch_offsets, cfg->patch_offsets + 2 * cfg->n_patch_offsets);
    if (cfg->remove_lines) {
      for (int i = 0; i < cfg->n_high_pass; ++i) p->hp_f.push_back(static_cast<float>(cfg->high_pass_fir[i]));  // O2: kernel in the float working type
      for (int i = 0; i < cfg->n_low_pass; ++i) p->lp_f.push_back(static_cast<float>(cfg->low_pass_fir[i]));
    }
    if (cfg->static_mask) p->static_mask.assign(cfg->static_mask, cfg->static_mask + static_cast<size_t>(cfg->rows) * cfg->cols);
    p->cfg.pixel_shift_by_row = p->shift.data();
    p->cfg.beam_altitude_angles = p->alt.data();
    p->cfg.patch_offsets = p->offsets.data();
    p->cfg.high_pass_fir = p->cfg.low_pass_fir = nullptr;
    p->cfg.static_mask = nullptr;
    int rc = upload(ctx, p->d_alt, p->alt.data(), p->alt.size() * sizeof(float));
    if (rc == MH_OK) rc = upload(ctx, p->d_shift, p->shift.data(), p->shift.size() * sizeof(int32_t));
    if (rc == MH_OK) rc = upload(ctx, p->d_hp, p->hp_f.data(), p->hp_f.size() * sizeof(float));
    if (rc == MH_OK) rc = upload(ctx, p->d_lp, p->lp_f.data(), p->lp_f.size() * sizeof(float));
    if (rc == MH_OK) rc = upload(ctx, p->d_static, p->static_mask.data(), p->static_mask.size());
    hipError_t e = hipSuccess;
    if (rc == MH_OK) e = hipHostMalloc(reinterpret_cast<void **>(&p->h_counters), 2 * sizeof(mh::PhotoCounters), hipHostMallocMapped);
    if (rc == MH_OK && e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&p->d_counters), p->h_counters, 0);
    if (rc == MH_OK && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (rc != MH_OK || e != hipSuccess) {
      photo_release(p);
      return rc != MH_OK ? rc : hip_fail(ctx, e, "mh_photo_create");
    }
    std::memset(p->h_counters, 0, 2 * sizeof(*p->h_counters));
    // derived parameters (photometric_config.cpp:98-110); DEG2RAD is PCL's macro (x * 0.017453293)
    mh::PhotoModel & m = p->model;
    m.rows = cfg->rows;
    m.cols = cfg->cols;
    m.destagger = cfg->destagger;
    m.fx = -static_cast<float>(cfg->cols) / (2 * M_PI);
    m.cx = static_cast<float>(cfg->cols) / 2.0;
    m.fy = -static_cast<float>(cfg->rows) / std::fabs(static_cast<double>(p->alt.front() - p->alt.back()) * 0.017453293);
    m.beam_offset_m = static_cast<float>(cfg->lidar_origin_to_beam_origin_mm / 1000.0);
    m.range_min = cfg->range_min;
    m.range_max = cfg->range_max;
    m.alt_first = p->alt.front();
    m.alt_last = p->alt.back();
    m.margin_size = cfg->margin_size;
    m.occlusion_range_diff_threshold = cfg->occlusion_range_diff_threshold;
    m.alt = static_cast<const float *>(p->d_alt.p);
    m.pixel_shift = static_cast<const int *>(p->d_shift.p);
    *out = p;
    return MH_OK;
  });
}

void mh_photo_destroy(mh_photo * photo) { photo_release(photo); }

int mh_scan_keep_raw(mh_scan * scan, int keep)
{
  if (!scan) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_scan_keep_raw: scan is NULL");
  scan->keep_raw = keep != 0;
  return MH_OK;
}

static int photo_finish_preprocess(mh_photo * photo, PhotoFrame * fr, mh_point32 * host_desk, size_t n, bool commit = true)
{
  mh_ctx * ctx = photo->ctx;
  // corrected intensities back into the caller's cloud (:307-314)
  if (host_desk && n) {
    if (photo->h_int_cap < n) {
      if (photo->h_int_out) (void)hipHostFree(photo->h_int_out);
      photo->h_int_out = nullptr;
      photo->h_int_cap = 0;
      MH_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&photo->h_int_out), (n + n / 2) * sizeof(float), hipHostMallocDefault));
      photo->h_int_cap = n + n / 2;
    }
    MH_HIP(ctx, hipMemcpyAsync(photo->h_int_out, photo->d_int_out.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  }
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (photo->h_counters->project_throw) {
    frame_release(fr);
    return fail(ctx, MH_ERR_INVALID_ARG,
                "mh_photo_preprocess: project(): invalid x coordinate for a deskewed point (the reference throws, photometric_utils.cpp:90-97)");
  }
  if (commit) {
    frame_release(photo->frame);
    photo->frame = fr;
  } else {
    frame_release(photo->next_frame);  // a frame that was begun and never committed
    photo->next_frame = fr;
  }
  return MH_OK;
}

int mh_photo_preprocess(mh_photo * photo, const mh_point32 * points_raw, mh_point32 * points_deskewed, size_t n,
                        const uint32_t * unique_ns, const double * T_Le_Lt, size_t n_groups)
{
  if (!photo || (n && (!points_raw || !points_deskewed)) || (n_groups && (!unique_ns || !T_Le_Lt)))
    return fail(photo ? photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_preprocess: NULL argument");
  mh_ctx * ctx = photo->ctx;
  return guarded(ctx, "mh_photo_preprocess", [&]() -> int {
    if (n > static_cast<size_t>(photo->cfg.rows) * photo->cfg.cols)
      return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_preprocess: number of points exceeds the image size (photometric.cpp:103-110)");
    MH_HIP(ctx, mh_enter(ctx));
    PhotoFrame * fr = new PhotoFrame;
    fr->ctx = ctx;
    const size_t pb = (n ? n : 1) * sizeof(mh_point32);
    hipError_t e = fr->d_points.reserve(pb, ctx->stream, false);
    if (e == hipSuccess) e = photo->d_raw_pts.reserve(pb, ctx->stream, false);
    if (e == hipSuccess && n) e = hipMemcpyAsync(photo->d_raw_pts.p, points_raw, n * sizeof(mh_point32), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(fr->d_points.p, points_deskewed, n * sizeof(mh_point32), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
      frame_release(fr);
      return hip_fail(ctx, e, "mh_photo_preprocess: upload");
    }
    int rc = preprocess_device(photo, fr, static_cast<const mh_point32 *>(photo->d_raw_pts.p), n, unique_ns, T_Le_Lt, n_groups);
    if (rc != MH_OK) {
      frame_release(fr);
      return rc;
    }
    rc = photo_finish_preprocess(photo, fr, points_deskewed, n);
    if (rc != MH_OK) return rc;
    // the Frame keeps the cloud as it was handed over; the corrected intensities go to the caller's copy only (:115-116, :307-314)
    for (size_t i = 0; i < n; ++i) {
      const float v = photo->h_int_out[i];
      if (v == v) points_deskewed[i].intensity = v;  // NaN = "no pixel"
    }
    return MH_OK;
  });
}

// corrected intensities into the scan's resident cloud (:307-314), ordered before whatever the scan's own stream does next
// (already_written: the chain's last launch did it — only the ordering against the scan's stream is left)
static int photo_writeback_scan(mh_photo * photo, PhotoFrame * fr, mh_scan * scan, bool already_written = false)
{
  mh_ctx * ctx = photo->ctx;
  if (!scan->c.n_full) return MH_OK;
  if (!already_written)
    MH_HIP(ctx, mh::launch_photo_sobel_writeback(static_cast<const float *>(fr->d_intensity.p), static_cast<float *>(fr->d_dx.p),
                                               static_cast<float *>(fr->d_dy.p), static_cast<const int32_t *>(fr->d_idx.p),
                                               static_cast<mh_point32 *>(scan->d_full.p), nullptr, photo->cfg.rows, photo->cfg.cols, ctx->stream));
  if (scan->ctx != ctx) {
    if (!photo->scan_ev) MH_HIP(ctx, hipEventCreateWithFlags(&photo->scan_ev, hipEventDisableTiming));
    MH_HIP(ctx, hipEventRecord(photo->scan_ev, ctx->stream));
    MH_HIP(ctx, hipStreamWaitEvent(scan->ctx->stream, photo->scan_ev, 0));
  }
  return MH_OK;
}

static int photo_preprocess_scan(mh_photo * photo, mh_scan * scan, const double * T_Le_Lt, size_t n_groups, bool commit)
{
  if (!photo || !scan || (n_groups && !T_Le_Lt))
    return fail(photo ? photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_preprocess_scan: NULL argument");
  mh_ctx * ctx = photo->ctx;
  return guarded(ctx, "mh_photo_preprocess_scan", [&]() -> int {
    if (scan->ctx->device != ctx->device) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_preprocess_scan: scan lives on another device");
    if (!scan->prepared) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_preprocess_scan: no mh_scan_prepare_input before");
    if (!scan->raw_valid) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_preprocess_scan: call mh_scan_keep_raw(scan, 1) before mh_scan_deskew");
    if (n_groups != scan->c.n_unique_ns) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_preprocess_scan: one pose per unique timestamp");
    const size_t n = scan->c.n_full;
    if (n > static_cast<size_t>(photo->cfg.rows) * photo->cfg.cols)
      return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_preprocess_scan: number of points exceeds the image size");
    MH_HIP(ctx, mh_enter(ctx));
    if (scan->ctx != ctx) {  // the scan's deskew may still be running on its own stream: this stream queues behind it
      if (!photo->scan_ev) MH_HIP(ctx, hipEventCreateWithFlags(&photo->scan_ev, hipEventDisableTiming));
      MH_HIP(ctx, hipEventRecord(photo->scan_ev, scan->ctx->stream));
      MH_HIP(ctx, hipStreamWaitEvent(ctx->stream, photo->scan_ev, 0));
    }
    PhotoFrame * fr = new PhotoFrame;
    fr->ctx = ctx;
    hipError_t e = fr->d_points.reserve((n ? n : 1) * sizeof(mh_point32), ctx->stream, false);
    if (e != hipSuccess) {
      frame_release(fr);
      return hip_fail(ctx, e, "mh_photo_preprocess_scan: frame cloud");
    }
    std::vector<uint32_t> uns(n_groups);
    if (n_groups && scan->n_unique_cached == n_groups) {  // the host copy that came back with mh_scan_prepare_input's counters
      std::memcpy(uns.data(), scan->h_c + 1, n_groups * sizeof(uint32_t));
    } else {
      if (n_groups) MH_HIP(ctx, hipMemcpyAsync(uns.data(), scan->d_unique.p, n_groups * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // the frame's copy of the cloud is made by the chain's first launch, and a committing call's write-back of the corrected
    // intensities into the scan by its last (six launches -> four)
    bool folded = false;
    int rc = preprocess_device(photo, fr, static_cast<const mh_point32 *>(scan->d_full_raw.p), n, uns.data(), T_Le_Lt, n_groups,
                               static_cast<mh_point32 *>(scan->d_full.p), commit, &folded);
    if (rc != MH_OK) {
      frame_release(fr);
      return rc;
    }
    if (!commit) {
      // begin: nothing is waited for and the scan is not written; mh_photo_preprocess_commit does both
      if (!photo->next_ev) MH_HIP(ctx, hipEventCreateWithFlags(&photo->next_ev, hipEventDisableTiming));
      MH_HIP(ctx, hipEventRecord(photo->next_ev, ctx->stream));
      frame_release(photo->next_frame);  // a frame that was begun and never committed
      photo->next_frame = fr;
      photo->next_scan = scan;
      return MH_OK;
    }
    rc = photo_writeback_scan(photo, fr, scan, folded);
    if (rc != MH_OK) {
      frame_release(fr);
      return rc;
    }
    return photo_finish_preprocess(photo, fr, nullptr, n, true);
  });
}
int mh_photo_preprocess_scan(mh_photo * photo, mh_scan * scan, const double * T_Le_Lt, size_t n_groups)
{
  return photo_preprocess_scan(photo, scan, T_Le_Lt, n_groups, true);
}
int mh_photo_preprocess_scan_begin(mh_photo * photo, mh_scan * scan, const double * T_Le_Lt, size_t n_groups)
{
  return photo_preprocess_scan(photo, scan, T_Le_Lt, n_groups, false);
}
int mh_photo_preprocess_commit(mh_photo * photo)
{
  if (!photo) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_photo_preprocess_commit: NULL argument");
  return guarded(photo->ctx, "mh_photo_preprocess_commit", [&]() -> int {
    mh_ctx * ctx = photo->ctx;
    if (!photo->next_frame) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_preprocess_commit: no frame was begun (mh_photo_preprocess_scan_begin)");
    MH_HIP(ctx, mh_enter(ctx));
    MH_HIP(ctx, hipEventSynchronize(photo->next_ev));  // the frame's kernels (normally long done: they ran beside the caller's work)
    PhotoFrame * fr = photo->next_frame;
    photo->next_frame = nullptr;
    if (photo->h_counters->project_throw) {
      frame_release(fr);
      return fail(ctx, MH_ERR_INVALID_ARG,
                  "mh_photo_preprocess: project(): invalid x coordinate for a deskewed point (the reference throws, photometric_utils.cpp:90-97)");
    }
    const int rc = photo_writeback_scan(photo, fr, photo->next_scan);
    photo->next_scan = nullptr;
    if (rc != MH_OK) {
      frame_release(fr);
      return rc;
    }
    frame_release(photo->frame);
    photo->frame = fr;
    return MH_OK;
  });
}

int mh_photo_get_image(mh_photo * photo, int which, void * out, size_t capacity_bytes)
{
  if (!photo || !out) return fail(photo ? photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_get_image: NULL argument");
  mh_ctx * ctx = photo->ctx;
  return guarded(ctx, "mh_photo_get_image", [&]() -> int {
    PhotoFrame * fr = photo->frame;
    if (!fr) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_get_image: no frame");
    MH_HIP(ctx, mh_enter(ctx));
    const size_t npx = static_cast<size_t>(fr->rows) * fr->cols;
    const void * src = nullptr;
    size_t bytes = 0;
    switch (which) {
      case 0: src = fr->d_intensity.p; bytes = npx * 4; break;
      case 1: src = fr->d_range.p; bytes = npx * 4; break;
      case 2: src = fr->d_dx.p; bytes = npx * 4; break;
      case 3: src = fr->d_dy.p; bytes = npx * 4; break;
      case 4: src = fr->d_mask.p; bytes = npx; break;
      case 5: src = fr->d_idx.p; bytes = npx * 4; break;
      case 6: src = fr->d_yaw.p; bytes = npx * 4; break;
      case 7: src = fr->d_proj.p; bytes = npx * 4 * mh::kPhotoDup; break;
      case 8:
      case 9: {
        MH_HIP(ctx, photo->d_grad.reserve(npx, ctx->stream, false));
        MH_HIP(ctx, photo->d_detmask.reserve(npx, ctx->stream, false));
        MH_HIP(ctx, mh::launch_photo_grad(static_cast<const float *>(fr->d_dx.p), static_cast<const float *>(fr->d_dy.p),
                                          static_cast<uint8_t *>(photo->d_grad.p), static_cast<int>(npx), ctx->stream));
        MH_HIP(ctx, mh::launch_photo_erode(static_cast<const uint8_t *>(fr->d_mask.p), nullptr, photo->cfg.margin_size,
                                           static_cast<uint8_t *>(photo->d_detmask.p), fr->rows, fr->cols,
                                           photo->cfg.patch_size + photo->cfg.erosion_buffer, ctx->stream));
        src = which == 8 ? photo->d_grad.p : photo->d_detmask.p;
        bytes = npx;
        break;
      }
      default: return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_get_image: which must be 0..9");
    }
    if (capacity_bytes < bytes) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_get_image: buffer too small");
    MH_HIP(ctx, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MH_OK;
  });
}

int mh_photo_num_features(const mh_photo * photo, size_t * n_features, size_t * n_points_total)
{
  if (!photo) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_photo_num_features: photo is NULL");
  size_t np = 0;
  for (const HostFeature & f : photo->features) np += static_cast<size_t>(f.hdr.n_points);
  if (n_features) *n_features = photo->features.size();
  if (n_points_total) *n_points_total = np;
  return MH_OK;
}

int mh_photo_get_features(const mh_photo * photo, mh_photo_feature * features, double * Le_ps, double * intensities, double * psi)
{
  if (!photo) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_photo_get_features: photo is NULL");
  size_t o = 0;
  for (size_t i = 0; i < photo->features.size(); ++i) {
    const HostFeature & f = photo->features[i];
    const size_t m = static_cast<size_t>(f.hdr.n_points);
    if (features) features[i] = f.hdr;
    if (Le_ps) std::memcpy(Le_ps + 3 * o, f.Le_ps.data(), 3 * m * sizeof(double));
    if (intensities) std::memcpy(intensities + o, f.intensities.data(), m * sizeof(double));
    if (psi) std::memcpy(psi + o, f.psi.data(), m * sizeof(double));
    o += m;
  }
  return MH_OK;
}

int mh_photo_set_features(mh_photo * photo, const mh_photo_feature * features, size_t n_features, const double * Le_ps,
                          const double * intensities, const double * psi)
{
  if (!photo || (n_features && (!features || !Le_ps || !intensities || !psi)))
    return fail(photo ? photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_set_features: NULL argument");
  return guarded(photo->ctx, "mh_photo_set_features", [&]() -> int {
    std::vector<HostFeature> nf(n_features);
    size_t o = 0;
    for (size_t i = 0; i < n_features; ++i) {
      const int m = features[i].n_points;
      if (m < 2 || m > mh::kPhotoMaxPatch) return fail(photo->ctx, MH_ERR_UNSUPPORTED, "mh_photo_set_features: 2..64 points per feature");
      nf[i].hdr = features[i];
      nf[i].Le_ps.assign(Le_ps + 3 * o, Le_ps + 3 * (o + m));
      nf[i].intensities.assign(intensities + o, intensities + o + m);
      nf[i].psi.assign(psi + o, psi + o + m);
      o += static_cast<size_t>(m);
      if (features[i].id >= photo->next_id) photo->next_id = features[i].id + 1;
    }
    photo->features.swap(nf);
    return MH_OK;
  });
}

int mh_photo_detect_features(mh_photo * photo, int num_to_detect, const double R_W_Be[9], const double t_W_Be[3],
                             const double * bias_directions, size_t n_directions)
{
  if (!photo || !R_W_Be || !t_W_Be || (n_directions && !bias_directions))
    return fail(photo ? photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_detect_features: NULL argument");
  return guarded(photo->ctx, "mh_photo_detect_features", [&]() -> int {
    MH_HIP(photo->ctx, mh_enter(photo->ctx));
    return detect_features_impl(photo, num_to_detect, R_W_Be, t_W_Be, bias_directions, n_directions);
  });
}

int mh_photo_detect_prefetch(mh_photo * photo)
{
  if (!photo) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_photo_detect_prefetch: NULL argument");
  return guarded(photo->ctx, "mh_photo_detect_prefetch", [&]() -> int {
    mh_ctx * ctx = photo->ctx;
    if (!photo->frame) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_detect_prefetch: no frame (call mh_photo_preprocess first)");
    MH_HIP(ctx, mh_enter(ctx));
    if (photo->cand_frame == photo->frame) return MH_OK;
    if (photo->cand_frame) {  // a prefetch nobody consumed: its copies must have landed before h_stage is written again
      MH_HIP(ctx, hipEventSynchronize(photo->cand_ev));
      frame_release(photo->cand_frame);
      photo->cand_frame = nullptr;
    }
    if (!photo->cand_ev) MH_HIP(ctx, hipEventCreateWithFlags(&photo->cand_ev, hipEventDisableTiming));
    const int rc = detect_enqueue_candidates(photo, photo->frame);
    if (rc != MH_OK) return rc;
    MH_HIP(ctx, hipEventRecord(photo->cand_ev, ctx->stream));
    photo->cand_frame = photo->frame;
    photo->cand_frame->refs.fetch_add(1);
    return MH_OK;
  });
}

int mh_photo_update_map(mh_photo * photo, mh_photo_factor * factor, const double R_W_Be[9], const double t_W_Be[3],
                        const double * bias_directions, size_t n_directions)
{
  if (!photo || !R_W_Be || !t_W_Be || (n_directions && !bias_directions))
    return fail(photo ? photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_update_map: NULL argument");
  return guarded(photo->ctx, "mh_photo_update_map", [&]() -> int {
    MH_HIP(photo->ctx, mh_enter(photo->ctx));
    if (factor) {  // photometric.cpp:402-494
      if (factor->photo != photo) return fail(photo->ctx, MH_ERR_INVALID_ARG, "mh_photo_update_map: the factor belongs to another mh_photo");
      if (factor->statuses.size() != photo->features.size())
        return fail(photo->ctx, MH_ERR_INVALID_ARG, "mh_photo_update_map: the tracked features changed since the factor was built");
      std::vector<size_t> invalid;
      for (size_t i = 0; i < factor->statuses.size(); ++i) {
        if (factor->statuses[i] != MH_PHOTO_VALID) {
          invalid.push_back(i);
        } else {
          HostFeature & f = photo->features[i];
          f.hdr.center[0] = factor->centers[2 * i];
          f.hdr.center[1] = factor->centers[2 * i + 1];
          f.hdr.life_time++;
          if (f.hdr.life_time >= photo->cfg.max_feature_life_time) invalid.push_back(i);
        }
      }
      for (auto it = invalid.rbegin(); it != invalid.rend(); ++it) photo->features.erase(photo->features.begin() + static_cast<long>(*it));
    }
    const int want = photo->cfg.num_features_detect - static_cast<int>(photo->features.size());  // :507-509
    return detect_features_impl(photo, want, R_W_Be, t_W_Be, bias_directions, n_directions);
  });
}

// PhotometricFactor ctor body shared by create (current frame + tracked features) and clone (the source factor's)
static int photo_factor_build(mh_photo * photo, PhotoFrame * frame, const std::vector<HostFeature> & feats, const double * VSVt,
                              bool binary, mh_photo_factor ** out)
{
  mh_ctx * ctx = photo->ctx;
  MH_HIP(ctx, mh_enter(ctx));
  mh_photo_factor * f = new mh_photo_factor;
  f->photo = photo;
  photo->refs.fetch_add(1);
  f->frame = frame;
  f->frame->refs.fetch_add(1);
  f->binary = binary;
  for (int i = 0; i < 36; ++i) f->VSVt[i] = VSVt ? VSVt[i] : ((i % 7 == 0) ? 1.0 : 0.0);
  f->features = feats;
  const size_t nf = f->features.size();
  // The factor's pinned block holds the kernel's outputs and, behind them, the staged inputs (patch points, psi, point
  // counts): the uploads run from memory that lives as long as the factor, so creating a factor does not wait for them.
  const size_t out_part = (nf * (2 + mh::kPhotoPartial) * sizeof(double) + nf * sizeof(int32_t) + 64 + 255) & ~size_t(255);
  const size_t b_Le = nf * mh::kPhotoMaxPatch * 3 * sizeof(double), b_ps = nf * mh::kPhotoMaxPatch * sizeof(double), b_np = nf * sizeof(int32_t);
  f->out_bytes = (out_part + b_Le + b_ps + b_np + 64 + 4095) & ~size_t(4095);  // few size classes for the pinned cache
  hipError_t e = AllocCache::alloc_pinned(&f->h_out, f->out_bytes);
  int rc = MH_OK;
  if (e == hipSuccess) e = hipHostGetDevicePointer(&f->d_out, f->h_out, 0);
  if (e == hipSuccess) {
    char * stage = static_cast<char *>(f->h_out) + out_part;
    double * Le = reinterpret_cast<double *>(stage);
    double * ps = reinterpret_cast<double *>(stage + b_Le);
    int32_t * np = reinterpret_cast<int32_t *>(stage + b_Le + b_ps);
    std::memset(stage, 0, b_Le + b_ps + b_np);
    for (size_t i = 0; i < nf; ++i) {
      const HostFeature & hf = f->features[i];
      np[i] = hf.hdr.n_points;
      std::memcpy(&Le[i * mh::kPhotoMaxPatch * 3], hf.Le_ps.data(), hf.Le_ps.size() * sizeof(double));
      std::memcpy(&ps[i * mh::kPhotoMaxPatch], hf.psi.data(), hf.psi.size() * sizeof(double));
    }
    rc = upload_pinned(ctx, f->d_Le, Le, b_Le);
    if (rc == MH_OK) rc = upload_pinned(ctx, f->d_psi, ps, b_ps);
    if (rc == MH_OK) rc = upload_pinned(ctx, f->d_npts, np, b_np);
  }
  if (rc == MH_OK && e == hipSuccess) e = f->d_rows.reserve(nf * mh::kPhotoMaxPatch * 8 * sizeof(double), ctx->stream, false);
  if (rc == MH_OK && e == hipSuccess) e = f->d_ticket.reserve(64, ctx->stream, false);
  if (rc == MH_OK && e == hipSuccess) e = hipMemsetAsync(f->d_ticket.p, 0, 64, ctx->stream);
  if (rc != MH_OK || e != hipSuccess) {
    mh_photo_factor_destroy(f);
    return rc != MH_OK ? rc : hip_fail(ctx, e, "mh_photo_factor_create");
  }
  f->statuses.assign(nf, MH_PHOTO_UNPROCESSED);
  f->centers.assign(2 * nf, 0.0);
  for (size_t i = 0; i < nf; ++i) {
    f->centers[2 * i] = f->features[i].hdr.center[0];
    f->centers[2 * i + 1] = f->features[i].hdr.center[1];
  }
  *out = f;
  return MH_OK;
}

int mh_photo_factor_create(mh_photo * photo, const double * VSVt, int is_binary, mh_photo_factor ** out)
{
  if (!photo || !out) return fail(photo ? photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_factor_create: NULL argument");
  *out = nullptr;
  mh_ctx * ctx = photo->ctx;
  return guarded(ctx, "mh_photo_factor_create", [&]() -> int {
    if (!photo->frame) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_factor_create: no frame (call mh_photo_preprocess first)");
    if (photo->features.empty()) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_factor_create: No features in a_features (photometric_factor.hpp:97-99)");
    return photo_factor_build(photo, photo->frame, photo->features, VSVt, is_binary != 0, out);
  });
}

int mh_photo_factor_clone(const mh_photo_factor * src, mh_photo_factor ** out)
{
  if (!src || !out) return fail(src ? src->photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_factor_clone: NULL argument");
  *out = nullptr;
  mh_ctx * ctx = src->photo->ctx;
  return guarded(ctx, "mh_photo_factor_clone", [&]() -> int {
    const int rc = photo_factor_build(src->photo, src->frame, src->features, src->VSVt, src->binary, out);
    if (rc != MH_OK) return rc;
    (*out)->statuses = src->statuses;  // the copy constructor's member-wise copy (photometric_factor.hpp:120-124)
    (*out)->centers = src->centers;
    (*out)->features = src->features;
    return MH_OK;
  });
}

void mh_photo_factor_destroy(mh_photo_factor * f)
{
  if (!f) return;
  mh_ctx * ctx = f->photo->ctx;
  (void)mh_enter(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  for (DevBuf * b : {&f->d_Le, &f->d_psi, &f->d_npts, &f->d_rows, &f->d_ticket}) b->release(true);
  if (f->h_out) AllocCache::free_pinned(f->h_out, f->out_bytes);
  for (auto & e : f->ev)
    if (e) (void)hipEventDestroy(e);
  frame_release(f->frame);
  photo_release(f->photo);
  delete f;
}

size_t mh_photo_factor_size(const mh_photo_factor * f) { return f ? f->features.size() : 0; }

static int photo_linearize_enqueue(mh_photo_factor * f, const double R_b[9], const double t_b[3], const double * R_a, const double * t_a)
{
  mh_ctx * ctx = f->photo->ctx;
  const bool timed = ctx->profiling > 0;
  {
    if (f->binary && (!R_a || !t_a)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_factor_linearize: the binary factor needs T_a");
    MH_HIP(ctx, mh_enter(ctx));
    const mh_photo_config & c = f->photo->cfg;
    const size_t nf = f->features.size();
    Pose Tb, Ta, TBL;
    std::memcpy(Tb.R, R_b, sizeof(Tb.R));
    std::memcpy(Tb.t, t_b, sizeof(Tb.t));
    for (int i = 0; i < 9; ++i) Ta.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    Ta.t[0] = Ta.t[1] = Ta.t[2] = 0.0;
    if (f->binary) {
      std::memcpy(Ta.R, R_a, sizeof(Ta.R));
      std::memcpy(Ta.t, t_a, sizeof(Ta.t));
    }
    std::memcpy(TBL.R, c.T_B_L_R, sizeof(TBL.R));
    std::memcpy(TBL.t, c.T_B_L_t, sizeof(TBL.t));
    const Pose dBe = pose_mul(pose_inv(Tb), Ta);                     // :147
    const Pose dLe = pose_mul(pose_mul(pose_inv(TBL), dBe), TBL);    // :148-149
    mh::PhotoLinArgs a;
    a.model = f->photo->model;
    PhotoFrame * fr = f->frame;
    a.frame.points = static_cast<const mh_point32 *>(fr->d_points.p);
    a.frame.n_points = static_cast<int>(fr->n_points);
    a.frame.intensity = static_cast<const float *>(fr->d_intensity.p);
    a.frame.range = static_cast<const float *>(fr->d_range.p);
    a.frame.dx = static_cast<const float *>(fr->d_dx.p);
    a.frame.dy = static_cast<const float *>(fr->d_dy.p);
    a.frame.mask = static_cast<const uint8_t *>(fr->d_mask.p);
    a.frame.idx = static_cast<const int32_t *>(fr->d_idx.p);
    a.frame.proj = static_cast<const int32_t *>(fr->d_proj.p);
    a.frame.yaw = static_cast<const float *>(fr->d_yaw.p);
    a.frame.pose_ns = static_cast<const uint32_t *>(fr->d_pose_ns.p);
    a.frame.pose_Rt = reinterpret_cast<const double *>(static_cast<const char *>(fr->d_pose_ns.p) + fr->pose_rt_offset);
    a.frame.n_poses = fr->n_poses;
    a.Le_ps = static_cast<const double *>(f->d_Le.p);
    a.psi_a = static_cast<const double *>(f->d_psi.p);
    a.n_pts = static_cast<const int32_t *>(f->d_npts.p);
    a.n_features = static_cast<int>(nf);
    a.binary = f->binary ? 1 : 0;
    std::memcpy(a.dLe_R, dLe.R, sizeof(a.dLe_R));
    std::memcpy(a.dLe_t, dLe.t, sizeof(a.dLe_t));
    std::memcpy(a.dBe_R, dBe.R, sizeof(a.dBe_R));
    std::memcpy(a.dBe_t, dBe.t, sizeof(a.dBe_t));
    std::memcpy(a.TBL_R, TBL.R, sizeof(a.TBL_R));
    std::memcpy(a.TBL_t, TBL.t, sizeof(a.TBL_t));
    a.sigma = c.sigma;
    a.max_error = c.max_error;
    a.robust_param = c.robust_cost_function_parameter;
    a.use_robust = c.use_robust_cost_function;
    a.robust_is_huber = c.robust_cost_function == 0;
    // layout of the mapped block: centres (2 nf doubles) | partial sums (nf x kPhotoPartial doubles) | statuses (nf int32)
    a.centers = static_cast<double *>(f->d_out);
    a.partials = a.centers + 2 * nf;
    a.status = reinterpret_cast<int32_t *>(a.partials + nf * mh::kPhotoPartial);
    a.rows_out = static_cast<double *>(f->d_rows.p);
    a.counters = f->photo->d_counters + 1;
    // the completion word sits behind the statuses in the mapped block (the block is sized with 64 spare bytes)
    const size_t seq_off = ((2 * nf + nf * mh::kPhotoPartial) * sizeof(double) + nf * sizeof(int32_t) + 7) & ~size_t(7);
    a.seq = 0;
    a.ticket = static_cast<unsigned int *>(f->d_ticket.p);
    a.host_seq = reinterpret_cast<unsigned int *>(static_cast<char *>(f->d_out) + seq_off);
    f->pending_seq = 0;
    if (!timed && nf > 0) {  // (no kernel is launched for an empty factor)
      if (++f->seq_counter == 0) ++f->seq_counter;
      a.seq = f->pending_seq = f->seq_counter;
      __atomic_store_n(reinterpret_cast<unsigned int *>(static_cast<char *>(f->h_out) + seq_off), 0u, __ATOMIC_RELEASE);  // re-arm
    }
    f->photo->h_counters[1].project_throw = f->photo->h_counters[1].pose_missing = 0;
    if (timed && !f->ev[0]) {
      MH_HIP(ctx, hipEventCreate(&f->ev[0]));
      MH_HIP(ctx, hipEventCreate(&f->ev[1]));
    }
    if (timed) MH_HIP(ctx, hipEventRecord(f->ev[0], ctx->stream));
    MH_HIP(ctx, mh::launch_photo_linearize(a, ctx->stream));
    if (timed) MH_HIP(ctx, hipEventRecord(f->ev[1], ctx->stream));
  }
  f->pending_timed = timed;
  f->pending = true;
  return MH_OK;
}

static int photo_linearize_finish(mh_photo_factor * f, mh_photo_result * out)
{
  mh_ctx * ctx = f->photo->ctx;
  const mh_photo_config & c = f->photo->cfg;
  (void)c;
  const size_t nf = f->features.size();
  const bool timed = f->pending_timed;
  f->pending = false;
  {
    bool need_sync = f->pending_seq == 0;
    if (!need_sync) {  // spin on the completion number the kernel's last block publishes (mh_icp_wait does the same)
      const size_t seq_off = ((2 * nf + nf * mh::kPhotoPartial) * sizeof(double) + nf * sizeof(int32_t) + 7) & ~size_t(7);
      const volatile unsigned int * flag = reinterpret_cast<const volatile unsigned int *>(static_cast<const char *>(f->h_out) + seq_off);
      timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (unsigned spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != f->pending_seq; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u) {
          timespec t1;
          clock_gettime(CLOCK_MONOTONIC, &t1);
          if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 20000000L) {  // 20 ms: fall back
            need_sync = true;
            break;
          }
        }
      }
    }
    if (need_sync) MH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the end of the kernel makes its host writes visible
    const double * new_centers = static_cast<const double *>(f->h_out);
    f->partials.assign(new_centers + 2 * nf, new_centers + 2 * nf + nf * mh::kPhotoPartial);
    std::memcpy(f->statuses.data(), new_centers + 2 * nf + nf * mh::kPhotoPartial, nf * sizeof(int32_t));
    std::memset(out, 0, sizeof(*out));
    out->gpu_ms = -1.f;
    if (timed) (void)hipEventElapsedTime(&out->gpu_ms, f->ev[0], f->ev[1]);
    out->n_exceptions = static_cast<int32_t>(f->photo->h_counters[1].project_throw + f->photo->h_counters[1].pose_missing);
    // accumulate the per-feature sums in feature order (the reference's serial loop, :157-330)
    const int NV = f->binary ? 13 : 7;
    auto ent = [NV](int r, int cc) {
      if (r > cc) std::swap(r, cc);
      return r * NV - r * (r - 1) / 2 + (cc - r);
    };
    double Hbb[36] = {0}, Hba[36] = {0}, Haa[36] = {0}, bb[6] = {0}, ba[6] = {0}, fs = 0;
    for (size_t i = 0; i < nf; ++i) {
      out->status_hist[f->statuses[i]]++;
      if (f->statuses[i] != MH_PHOTO_VALID) continue;
      f->centers[2 * i] = new_centers[2 * i];
      f->centers[2 * i + 1] = new_centers[2 * i + 1];
      f->features[i].hdr.center[0] = new_centers[2 * i];
      f->features[i].hdr.center[1] = new_centers[2 * i + 1];
      const double * p = &f->partials[i * mh::kPhotoPartial];
      for (int r = 0; r < 6; ++r) {
        for (int cc = 0; cc < 6; ++cc) Hbb[6 * r + cc] += p[ent(r, cc)];
        bb[r] += p[ent(r, NV - 1)];
      }
      fs += p[ent(NV - 1, NV - 1)];
      if (f->binary)
        for (int r = 0; r < 6; ++r) {
          for (int cc = 0; cc < 6; ++cc) {
            Hba[6 * r + cc] += p[ent(r, 6 + cc)];
            Haa[6 * r + cc] += p[ent(6 + r, 6 + cc)];
          }
          ba[r] += p[ent(6 + r, 12)];
        }
    }
    out->f = fs;
    if (f->binary) {
      std::memcpy(out->H_bb, Hbb, sizeof(Hbb));
      std::memcpy(out->H_ba, Hba, sizeof(Hba));
      std::memcpy(out->H_aa, Haa, sizeof(Haa));
      std::memcpy(out->b_b, bb, sizeof(bb));
      std::memcpy(out->b_a, ba, sizeof(ba));
      return MH_OK;
    }
    // :336-351  J_b_T_J_b = VSVt J_I VSVt;  J_b_T_b = VSVt J_I VSVt J_I^-1 b_I;  localizabilities of the result
    double t1[36], t2[36], inv[36], t3[36];
    mat6_mul(f->VSVt, Hbb, t1);
    mat6_mul(t1, f->VSVt, t2);
    mat6_inv(Hbb, inv);
    mat6_mul(t2, inv, t3);
    std::memcpy(out->H_bb, t2, sizeof(t2));
    for (int r = 0; r < 6; ++r) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += t3[6 * r + k] * bb[k];
      out->b_b[r] = s;
    }
    double Hr[9], Ht[9];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        Hr[3 * r + cc] = t2[6 * r + cc];
        Ht[3 * r + cc] = t2[6 * (3 + r) + 3 + cc];
      }
    mh::compute_localizability(Hr, out->loc_rot_final, out->eigvec_rot);
    mh::compute_localizability(Ht, out->loc_trans_final, out->eigvec_trans);
    return MH_OK;
  }
}

int mh_photo_factor_linearize(mh_photo_factor * f, const double R_b[9], const double t_b[3], const double * R_a, const double * t_a,
                              mh_photo_result * out)
{
  if (!f || !R_b || !t_b || !out) return fail(f ? f->photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_factor_linearize: NULL argument");
  mh_ctx * ctx = f->photo->ctx;
  return guarded(ctx, "mh_photo_factor_linearize", [&]() -> int {
    if (f->pending) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_factor_linearize: a call is in flight (mh_photo_factor_wait first)");
    const int rc = photo_linearize_enqueue(f, R_b, t_b, R_a, t_a);
    return rc != MH_OK ? rc : photo_linearize_finish(f, out);
  });
}

int mh_photo_factor_linearize_async(mh_photo_factor * f, const double R_b[9], const double t_b[3], const double * R_a, const double * t_a)
{
  if (!f || !R_b || !t_b) return fail(f ? f->photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_factor_linearize_async: NULL argument");
  mh_ctx * ctx = f->photo->ctx;
  return guarded(ctx, "mh_photo_factor_linearize_async", [&]() -> int {
    if (f->pending) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_factor_linearize_async: a call is already in flight");
    return photo_linearize_enqueue(f, R_b, t_b, R_a, t_a);
  });
}

int mh_photo_factor_wait(mh_photo_factor * f, mh_photo_result * out)
{
  if (!f || !out) return fail(f ? f->photo->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_photo_factor_wait: NULL argument");
  mh_ctx * ctx = f->photo->ctx;
  return guarded(ctx, "mh_photo_factor_wait", [&]() -> int {
    if (!f->pending) return fail(ctx, MH_ERR_INVALID_ARG, "mh_photo_factor_wait: no call in flight");
    return photo_linearize_finish(f, out);
  });
}

int mh_photo_factor_get_state(const mh_photo_factor * f, int32_t * statuses, double * centers, double * rows)
{
  if (!f) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_photo_factor_get_state: factor is NULL");
  mh_ctx * ctx = f->photo->ctx;
  const size_t nf = f->features.size();
  if (statuses) std::memcpy(statuses, f->statuses.data(), nf * sizeof(int32_t));
  if (centers) std::memcpy(centers, f->centers.data(), 2 * nf * sizeof(double));
  if (rows) {
    MH_HIP(ctx, mh_enter(ctx));
    MH_HIP(ctx, hipMemcpy(rows, f->d_rows.p, nf * mh::kPhotoMaxPatch * 8 * sizeof(double), hipMemcpyDeviceToHost));
  }
  return MH_OK;
}

}  // extern "C"
