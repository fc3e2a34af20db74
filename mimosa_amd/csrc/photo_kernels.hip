// HIP kernels (gfx950 / CDNA4, wave64) for the LiDAR photometric path — SURVEY.md §8 row f-2, BASELINE configs[3].
//
// Reference: Photometric::preprocess src/lidar/photometric.cpp:92-320 (image formation, yaw table, proj_idx, filter
// chain, Sobel, mask), Photometric::detectFeatures :524-540 (gradient magnitude), PhotometricFactor::linearize
// include/mimosa/lidar/photometric_factor.hpp:136-355, project / projectUndistorted / getProjectionJacobian /
// getSubPixelValue / getPsi / getPsiJacobian src/lidar/photometric_utils.cpp:13-388.  The OpenCV / Eigen / PCL
// behaviour these kernels reproduce is listed as assumptions O1-O12 in oracle/photo_ref.hpp.
//
// Compiled with -ffp-contract=off: the image chain is f32 arithmetic whose rounding the oracle reproduces bit for
// bit (no FMA in the reference's baseline x86-64 build), and projected pixel coordinates feed round() / floor().
//
// Shapes: a 128 x 1024 f32 image is 512 KiB — every stage is a streaming pass over <= 131 072 points or pixels, far
// from any bandwidth limit; what the chain costs is launches (each stage is one kernel, the whole preprocess is 12).
// Separable filters stage their tile + halo in LDS (REFLECT_101 resolved at load time); the factor runs one wave
// per feature, one lane per patch point, with wave shuffles for the NCC sums and the 28 / 91 Hessian sums.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "photo_device.hpp"
#include "wave_dpp.hpp"

namespace mh
{
namespace
{
constexpr int kT = 256;
constexpr int kProjEmpty = 0x7F7F7F7F;  // proj_idx slots are memset to 0x7F bytes before the build

__device__ __forceinline__ int reflect101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

// raw point index -> destaggered pixel (photometric.cpp:72-90, inverted)
__device__ __forceinline__ void idx_to_pixel(const PhotoModel & m, uint32_t idx, int & u, int & v)
{
  v = static_cast<int>(idx / static_cast<uint32_t>(m.cols));
  const int c = static_cast<int>(idx % static_cast<uint32_t>(m.cols));
  const int sh = (c + m.pixel_shift[v]) % m.cols;  // shifts may be negative (the real OS0-128 table is 31, 10, -10, -30 ...)
  u = m.destagger ? (sh < 0 ? sh + m.cols : sh) : c;
}

__device__ __forceinline__ float prep(float v, float scale, float gamma)
{
  if (scale != 1.0f) v = v * scale;          // img *= intensity_scale (O7)
  if (gamma != 1.0f) v = powf(v, gamma);     // cv::pow (O12: not restated; skipped by every shipped config)
  return v;
}

// ------------------------------------------------------------------------------------------------
// frame reset: one launch instead of six memsets (each a ~4 us fill kernel of its own)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void photo_clear_kernel(int npx, int n_pts, float * img_raw, float * range, uint8_t * mask_raw,
                                                          uint8_t * yaw_valid, int32_t * idx, int32_t * proj, float * int_out, int n_clear_blocks,
                                                          const uint4 * copy_src, uint4 * copy_dst, int n_copy16)
{
  if (static_cast<int>(blockIdx.x) >= n_clear_blocks) {  // second job of the launch: the frame's pose table from its mapped pinned block
    for (int k = (static_cast<int>(blockIdx.x) - n_clear_blocks) * kT + static_cast<int>(threadIdx.x); k < n_copy16; k += (static_cast<int>(gridDim.x) - n_clear_blocks) * kT)
      copy_dst[k] = copy_src[k];
    return;
  }
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < npx) {
    img_raw[i] = 0.f;
    range[i] = 0.f;
    mask_raw[i] = 0;
    yaw_valid[i] = 0;
    idx[i] = -1;
  }
  for (int k = i; k < npx * kPhotoDup; k += n_clear_blocks * kT) proj[k] = kProjEmpty;
  if (i < n_pts) int_out[i] = __uint_as_float(0xFFFFFFFFu);  // NaN = "this point owns no pixel"
}

// ------------------------------------------------------------------------------------------------
// preprocess stage 1
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void photo_scatter_kernel(const PhotoModel m, const mh_point32 * raw, const mh_point32 * desk,
                                                            int n, float * yaw, uint8_t * yaw_valid, float * intensity,
                                                            float * range, uint8_t * mask, int32_t * idx)
{
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const int npx = m.rows * m.cols;
  {
    const mh_point32 p = raw[i];
    if (p.idx < static_cast<uint32_t>(npx)) {
      int u, v;
      idx_to_pixel(m, p.idx, u, v);
      yaw[v * m.cols + u] = static_cast<float>(atan2(static_cast<double>(p.y), static_cast<double>(p.x)));  // :127
      yaw_valid[v * m.cols + u] = 1;
    }
  }
  {
    const mh_point32 p = desk[i];
    if (p.range < m.range_min || p.range > m.range_max) return;  // :209
    if (p.idx >= static_cast<uint32_t>(npx)) return;
    int u, v;
    idx_to_pixel(m, p.idx, u, v);
    const int px = v * m.cols + u;
    intensity[px] = p.intensity;
    range[px] = p.range;
    mask[px] = 1;
    idx[px] = i;
  }
}

// Round 5: scatter WITHOUT a reset launch ahead of it.  A pixel's "a raw point landed here" / "a deskewed point in range landed
// here" marks are 32-bit frame stamps (stamp[px] == seq) instead of bytes that have to be zeroed first; the consumers of the
// next launch (stage A) read through the stamps, and one of its jobs writes the reset values (intensity / range 0, idx -1)
// into the pixels no point claimed, so everything behind stage A sees the arrays the reset launch used to leave.  The
// projection index's "empty" fill and the corrected-intensity NaNs do not depend on the scatter: second and third job here.
// desk_copy: the frame's own copy of the deskewed cloud when the points are read out of a scan's resident cloud (the copy launch
// that used to precede the chain); nullptr when desk already is the frame's.
__global__ __launch_bounds__(kT) void photo_scatter_stamp_kernel(const PhotoModel m, const mh_point32 * raw, const mh_point32 * desk, mh_point32 * desk_copy,
                                                                  int n, float * yaw, uint32_t * ystamp, uint32_t * pstamp, uint32_t seq, float * intensity,
                                                                  float * range, int32_t * idx, float * int_out, int32_t * proj, int n_point_blocks)
{
  const int npx = m.rows * m.cols;
  if (static_cast<int>(blockIdx.x) >= n_point_blocks) {
    const int nb = static_cast<int>(gridDim.x) - n_point_blocks;
    for (int k = (static_cast<int>(blockIdx.x) - n_point_blocks) * kT + static_cast<int>(threadIdx.x); k < npx * kPhotoDup; k += nb * kT) proj[k] = kProjEmpty;
    return;
  }
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  int_out[i] = __uint_as_float(0xFFFFFFFFu);  // NaN = "this point owns no pixel" (stage C overwrites the owners')
  {
    const mh_point32 p = raw[i];
    if (p.idx < static_cast<uint32_t>(npx)) {
      int u, v;
      idx_to_pixel(m, p.idx, u, v);
      yaw[v * m.cols + u] = static_cast<float>(atan2(static_cast<double>(p.y), static_cast<double>(p.x)));  // :127
      ystamp[v * m.cols + u] = seq;
    }
  }
  {
    const mh_point32 p = desk[i];
    if (desk_copy) desk_copy[i] = p;
    if (p.range < m.range_min || p.range > m.range_max) return;  // :209
    if (p.idx >= static_cast<uint32_t>(npx)) return;
    int u, v;
    idx_to_pixel(m, p.idx, u, v);
    const int px = v * m.cols + u;
    intensity[px] = p.intensity;
    range[px] = p.range;
    idx[px] = i;
    pstamp[px] = seq;
  }
}

// ------------------------------------------------------------------------------------------------
// preprocess stage 2: one workgroup per image row
// ------------------------------------------------------------------------------------------------
constexpr int kMaxCols = 4096;
constexpr int kYawLdsWords = 3 * kMaxCols + 2 * 1024;  // s_yaw, s_prev, s_next, s_clast, s_cfirst (<= 1024 threads)
// row v of the yaw table; `lds`: kYawLdsWords words; any workgroup size up to 1024
__device__ __forceinline__ void yaw_fill_row(const PhotoModel & m, float * yaw, const uint8_t * yaw_valid, const int v, uint32_t * lds,
                                             const uint32_t * stamp = nullptr, const uint32_t seq = 0)
{
  float * s_yaw = reinterpret_cast<float *>(lds);
  int * s_prev = reinterpret_cast<int *>(lds + kMaxCols);
  int * s_next = reinterpret_cast<int *>(lds + 2 * kMaxCols);
  int * s_clast = reinterpret_cast<int *>(lds + 3 * kMaxCols);
  int * s_cfirst = s_clast + 1024;
  const int nthr = static_cast<int>(blockDim.x);
  const int cols = m.cols;
  float * yr = yaw + static_cast<size_t>(v) * cols;
  const uint8_t * vr = yaw_valid + static_cast<size_t>(v) * cols;
  const uint32_t * sr = stamp ? stamp + static_cast<size_t>(v) * cols : nullptr;
  auto valid = [&](int u) { return sr ? sr[u] == seq : vr[u] != 0; };
  const int per = (cols + nthr - 1) / nthr, c0 = min(cols, static_cast<int>(threadIdx.x) * per), c1 = min(cols, c0 + per);
  int last = -1, first = -1;
  for (int u = c0; u < c1; ++u) {
    const bool ok = valid(u);
    s_yaw[u] = ok ? yr[u] : 0.f;
    last = ok ? u : last;
    s_prev[u] = last;  // chunk-local
  }
  for (int u = c1 - 1; u >= c0; --u) {
    first = valid(u) ? u : first;
    s_next[u] = first;
  }
  s_clast[threadIdx.x] = last;
  s_cfirst[threadIdx.x] = first;
  __syncthreads();
  int carry_prev = -1, carry_next = -1;
  for (int t = static_cast<int>(threadIdx.x) - 1; t >= 0 && carry_prev < 0; --t) carry_prev = s_clast[t];
  for (int t = threadIdx.x + 1; t < nthr && carry_next < 0; ++t) carry_next = s_cfirst[t];
  const double kPi = 3.14159265358979323846;
  for (int u = c0; u < c1; ++u) {
    if (valid(u)) continue;
    const int pv = s_prev[u] >= 0 ? s_prev[u] : carry_prev, nx = s_next[u] >= 0 ? s_next[u] : carry_next;
    float out;
    if (pv < 0 && nx < 0) {  // :148-155 no valid column in this row
      const float t = static_cast<float>(u) / static_cast<float>(cols - 1);
      out = static_cast<float>(static_cast<double>(1.0f - t) * kPi + static_cast<double>(t) * (-kPi));
    } else if (pv < 0) {  // :158-167 before the first valid column
      const float t = static_cast<float>(u) / static_cast<float>(nx);
      out = static_cast<float>(static_cast<double>(1.0f - t) * kPi + static_cast<double>(t * s_yaw[nx]));
    } else if (nx < 0) {  // :187-198 after the last valid column
      const int gap = cols - 1 - pv;
      const float t = static_cast<float>(u - pv) / static_cast<float>(gap);
      out = static_cast<float>(static_cast<double>((1.0f - t) * s_yaw[pv]) + static_cast<double>(t) * (-kPi));
    } else {  // :170-185 between two valid columns
      const float yl = s_yaw[pv], yrr = s_yaw[nx], denom = static_cast<float>(nx - pv);
      const float t = static_cast<float>(u - pv) / denom;
      out = yl + t * (yrr - yl);
    }
    yr[u] = out;
  }
}
__global__ __launch_bounds__(kT) void photo_yaw_fill_kernel(const PhotoModel m, float * yaw, const uint8_t * yaw_valid)
{
  __shared__ uint32_t s_lds[kYawLdsWords];
  yaw_fill_row(m, yaw, yaw_valid, static_cast<int>(blockIdx.x), s_lds);
}

// ------------------------------------------------------------------------------------------------
// project() with the yaw table (photometric_utils.cpp:80-198).  Returns 1 ok, 0 = false, -1 = the reference throws.
// PCL's DEG2RAD / RAD2DEG macro constants (see oracle/photo_ref.hpp).
// ------------------------------------------------------------------------------------------------
// In two parts: the FRONT needs no yaw table (the fp64 square roots, atan2, asin and the altitude search: most of the arithmetic),
// the BACK is the column search in the table's row.  project_yaw = front, then back; the preprocess chain runs the front of every
// point in stage A's launch, beside the jobs that build the yaw table, and the back in stage B's (same operations, same order).
// alt: the beam-altitude table (m.alt), or a copy of it in LDS (the factor kernel: the seven probes of the search are dependent loads)
// front: -1 = the reference throws, 0 = false, 2 = go on with (phi, ux, uy)
__device__ __forceinline__ int project_yaw_front(const PhotoModel & m, double px, double py, double pz, double & phi, double & ux, double & uy,
                                                 const float * alt = nullptr)
{
  if (!alt) alt = m.alt;
  const double L = sqrt(px * px + py * py) - static_cast<double>(m.beam_offset_m);
  const double R = sqrt(L * L + pz * pz);
  phi = atan2(py, px);
  const double theta = asin(pz / R);
  ux = m.fx * phi + m.cx;
  if (ux < 0 || ux >= static_cast<double>(m.cols)) return -1;
  if (ux < 5 || ux > static_cast<double>(m.cols - 5)) return 0;
  if (theta > static_cast<double>(m.alt_first) * 0.017453293 || theta < static_cast<double>(m.alt_last) * 0.017453293) return 0;
  const double th_deg = theta * 57.29578;
  // greater = last altitude (descending table) that is > th_deg; clamped where the reference would read out of range
  int lo = 0, hi = m.rows;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (static_cast<double>(alt[mid]) > th_deg)
      lo = mid + 1;
    else
      hi = mid;
  }
  int g = lo - 1;
  g = g < 0 ? 0 : (g > m.rows - 2 ? m.rows - 2 : g);
  const float ag = alt[g], as = alt[g + 1];
  uy = static_cast<double>(g) + (static_cast<double>(ag) - th_deg) / static_cast<double>(ag - as);
  const int approx_y = static_cast<int>(round(uy));
  if (approx_y < 0 || approx_y >= m.rows) return 0;
  return 2;
}
__device__ __forceinline__ int project_yaw_back(const PhotoModel & m, const float * yaw, const double phi, double & ux, const double uy)
{
  const int approx_y = static_cast<int>(round(uy));
  const float * row = yaw + static_cast<size_t>(approx_y) * m.cols;
  int il = static_cast<int>(ux) - 5, ir = static_cast<int>(ux) + 5;
  il = il < 0 ? 0 : il;
  ir = ir > m.cols - 1 ? m.cols - 1 : ir;
  while (ir - il > 1) {
    const int mid = il + (ir - il) / 2;
    const float ym = row[mid];
    if (static_cast<double>(ym) == phi) {
      ux = static_cast<double>(mid);
      return (ux >= 0 && ux <= m.cols - 1 && uy >= 0 && uy <= m.rows - 1) ? 1 : 0;
    } else if (static_cast<double>(ym) < phi) {
      ir = mid;
    } else {
      il = mid;
    }
  }
  const float yl = row[il], yr = row[ir];
  ux = static_cast<double>(il) + (static_cast<double>(yl) - phi) / static_cast<double>(yl - yr);
  return (ux >= 0 && ux <= m.cols - 1 && uy >= 0 && uy <= m.rows - 1) ? 1 : 0;
}
__device__ __forceinline__ int project_yaw(const PhotoModel & m, const float * yaw, double px, double py, double pz, double & ux,
                                           double & uy, const float * alt = nullptr)
{
  double phi;
  const int r = project_yaw_front(m, px, py, pz, phi, ux, uy, alt);
  return r == 2 ? project_yaw_back(m, yaw, phi, ux, uy) : r;
}

// preprocess stage 3: project + proj_idx.  Slots 1..9 of a pixel hold its 9 smallest point indices in ascending
// order, kept by a carry chain of atomicMin (the displaced larger value moves on to the next slot); the reference
// appends in index order and stops at 9 (photometric.cpp:232-244) — the same set in the same order.
__device__ __forceinline__ void project_point(const PhotoModel & m, const mh_point32 * desk, int n, const float * yaw, int32_t * proj,
                                              PhotoCounters * counters, const int i)
{
  if (i >= n) return;
  const mh_point32 p = desk[i];
  if (p.range < m.range_min || p.range > m.range_max) return;
  double ux, uy;
  const int r = project_yaw(m, yaw, static_cast<double>(p.x), static_cast<double>(p.y), static_cast<double>(p.z), ux, uy);
  if (r < 0) atomicAdd(&counters->project_throw, 1u);
  if (r <= 0) return;
  const int u = static_cast<int>(round(ux)), v = static_cast<int>(round(uy));
  if (u < 0 || v < 0) return;
  int32_t * slot = proj + (static_cast<size_t>(v) * m.cols + u) * kPhotoDup;
  int cur = i;
#pragma unroll 1
  for (int s = 1; s < kPhotoDup && cur != kProjEmpty; ++s) {
    const int old = atomicMin(&slot[s], cur);
    cur = max(old, cur);
  }
}
// The same in the chain's two launches: what the front found out about point i travels in a 32-byte record.
struct ProjPre
{
  double phi, ux, uy;
  int32_t code;  // project_yaw_front's result; 0 also for a point outside the range gate
  int32_t pad;
};
static_assert(sizeof(ProjPre) == kPhotoProjPreBytes, "photo_api.hip sizes the scratch by kPhotoProjPreBytes");
__device__ __forceinline__ void project_point_front(const PhotoModel & m, const mh_point32 * desk, int n, ProjPre * pre, const int i)
{
  if (i >= n) return;
  const mh_point32 p = desk[i];
  ProjPre r;
  r.phi = r.ux = r.uy = 0.0;
  r.pad = 0;
  r.code = 0;
  if (!(p.range < m.range_min || p.range > m.range_max))
    r.code = project_yaw_front(m, static_cast<double>(p.x), static_cast<double>(p.y), static_cast<double>(p.z), r.phi, r.ux, r.uy);
  pre[i] = r;
}
__device__ __forceinline__ void project_point_back(const PhotoModel & m, int n, const float * yaw, int32_t * proj, PhotoCounters * counters,
                                                   const ProjPre * pre, const int i)
{
  if (i >= n) return;
  const ProjPre q = pre[i];
  if (q.code < 0) atomicAdd(&counters->project_throw, 1u);
  if (q.code != 2) return;
  double ux = q.ux;
  if (project_yaw_back(m, yaw, q.phi, ux, q.uy) <= 0) return;
  const int u = static_cast<int>(round(ux)), v = static_cast<int>(round(q.uy));
  if (u < 0 || v < 0) return;
  int32_t * slot = proj + (static_cast<size_t>(v) * m.cols + u) * kPhotoDup;
  int cur = i;
#pragma unroll 1
  for (int s = 1; s < kPhotoDup && cur != kProjEmpty; ++s) {
    const int old = atomicMin(&slot[s], cur);
    cur = max(old, cur);
  }
}
__global__ __launch_bounds__(kT) void photo_project_kernel(const PhotoModel m, const mh_point32 * desk, int n, const float * yaw,
                                                            int32_t * proj, PhotoCounters * counters)
{
  project_point(m, desk, n, yaw, proj, counters, static_cast<int>(blockIdx.x * kT + threadIdx.x));
}

__device__ __forceinline__ void proj_finalize_px(int n_pixels, int32_t * proj, const int px)
{
  if (px >= n_pixels) return;
  int32_t * slot = proj + static_cast<size_t>(px) * kPhotoDup;
  int v[kPhotoDup];
#pragma unroll
  for (int s = 1; s < kPhotoDup; ++s) v[s] = slot[s];  // all requested together: one round trip, not nine dependent ones
  int cnt = 0;
#pragma unroll
  for (int s = 1; s < kPhotoDup; ++s) {
    if (v[s] != kProjEmpty)
      ++cnt;
    else
      slot[s] = 0;
  }
  slot[0] = cnt;
}
__global__ __launch_bounds__(kT) void photo_proj_finalize_kernel(int n_pixels, int32_t * proj)
{
  proj_finalize_px(n_pixels, proj, static_cast<int>(blockIdx.x * kT + threadIdx.x));
}

// ------------------------------------------------------------------------------------------------
// filter chain
// ------------------------------------------------------------------------------------------------
// removeLines part 1 (:322-327): vertical correlation with the high-pass FIR.  Tile = 64 rows x 64 columns + halo.
constexpr int kVT_R = 16, kVT_C = 64;  // 8 x 16 = 128 workgroups for a 128 x 1024 image
constexpr int kVfirLdsWords = (kVT_R + kPhotoMaxTaps - 1) * kVT_C + kPhotoMaxTaps;
__device__ __forceinline__ void vfir_tile(const float * in, float * out, int rows, int cols, const float * taps, int n_taps, float scale, float gamma,
                                          const int bx, const int by, uint32_t * pool, const uint32_t * stamp = nullptr, const uint32_t seq = 0)
{
  const int nthr = static_cast<int>(blockDim.x);
  float * s_tile = reinterpret_cast<float *>(pool);
  float * s_taps = s_tile + (kVT_R + kPhotoMaxTaps - 1) * kVT_C;
  const int a = n_taps / 2, r0 = by * kVT_R, c0 = bx * kVT_C, th = kVT_R + n_taps - 1;
  for (int t = threadIdx.x; t < n_taps; t += nthr) s_taps[t] = taps[t];
  for (int e = threadIdx.x; e < th * kVT_C; e += nthr) {
    const int ty = e / kVT_C, tx = e % kVT_C;
    const int y = reflect101(r0 + ty - a, rows), x = c0 + tx;
    float v = 0.f;
    if (x < cols) {
      const size_t px = static_cast<size_t>(y) * cols + x;
      v = prep((!stamp || stamp[px] == seq) ? in[px] : 0.f, scale, gamma);  // an unclaimed pixel holds the reset value 0
    }
    s_tile[e] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kVT_R * kVT_C; e += nthr) {
    const int ty = e / kVT_C, tx = e % kVT_C;
    if (r0 + ty >= rows || c0 + tx >= cols) continue;
    float s = 0.f;
    for (int k = 0; k < n_taps; ++k) s = s + s_taps[k] * s_tile[(ty + k) * kVT_C + tx];  // O2: tap order, float
    out[static_cast<size_t>(r0 + ty) * cols + c0 + tx] = s;
  }
}
__global__ __launch_bounds__(kT) void photo_vfir_kernel(const float * in, float * out, int rows, int cols, const float * taps,
                                                         int n_taps, float scale, float gamma)
{
  __shared__ uint32_t s_pool[kVfirLdsWords];
  vfir_tile(in, out, rows, cols, taps, n_taps, scale, gamma, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), s_pool);
}

// removeLines part 2 (:328-336): horizontal low-pass of the high-passed image = the line artefacts; subtract, clip.
constexpr int kHT_R = 8, kHT_C = 128;
constexpr int kHfirLdsWords = kHT_R * (kHT_C + kPhotoMaxTaps - 1) + kPhotoMaxTaps;
__device__ __forceinline__ void hfir_sub_tile(const float * hp, const float * raw_in, float * out, int rows, int cols, const float * taps, int n_taps,
                                              float scale, float gamma, const int bx, const int by, uint32_t * pool)
{
  float * s_tile = reinterpret_cast<float *>(pool);
  float * s_taps = s_tile + kHT_R * (kHT_C + kPhotoMaxTaps - 1);
  const int a = n_taps / 2, r0 = by * kHT_R, c0 = bx * kHT_C, tw = kHT_C + n_taps - 1;
  const int nthr = static_cast<int>(blockDim.x);
  for (int t = threadIdx.x; t < n_taps; t += nthr) s_taps[t] = taps[t];
  for (int e = threadIdx.x; e < kHT_R * tw; e += nthr) {
    const int ty = e / tw, tx = e % tw;
    const int y = r0 + ty, x = reflect101(c0 + tx - a, cols);
    s_tile[e] = y < rows ? hp[static_cast<size_t>(y) * cols + x] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kHT_R * kHT_C; e += nthr) {
    const int ty = e / kHT_C, tx = e % kHT_C;
    if (r0 + ty >= rows || c0 + tx >= cols) continue;
    float s = 0.f;
    for (int k = 0; k < n_taps; ++k) s = s + s_taps[k] * s_tile[ty * tw + tx + k];
    const size_t px = static_cast<size_t>(r0 + ty) * cols + c0 + tx;
    const float v = prep(raw_in[px], scale, gamma) - s;
    out[px] = v < 0.f ? 0.f : v;
  }
}
__global__ __launch_bounds__(kT) void photo_hfir_sub_kernel(const float * hp, const float * raw_in, float * out, int rows, int cols,
                                                             const float * taps, int n_taps, float scale, float gamma)
{
  __shared__ uint32_t s_pool[kHfirLdsWords];
  hfir_sub_tile(hp, raw_in, out, rows, cols, taps, n_taps, scale, gamma, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), s_pool);
}

__global__ __launch_bounds__(kT) void photo_scale_kernel(const float * in, float * out, int n, float scale, float gamma)
{
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) out[i] = prep(in[i], scale, gamma);
}

// filterBrightness (:339-347): normalised box mean (double sums, O3) + 1, img = 140 * img / brightness (O7)
constexpr int kBT_R = 16, kBT_C = 64, kBMaxW = 63, kBMaxH = 31;
__global__ __launch_bounds__(kT) void photo_brightness_kernel(const float * in, float * out, int rows, int cols, int w, int h)
{
  __shared__ float s_raw[(kBT_R + kBMaxH - 1) * (kBT_C + kBMaxW - 1)];
  __shared__ double s_rs[(kBT_R + kBMaxH - 1) * kBT_C];
  const int ax = w / 2, ay = h / 2, r0 = blockIdx.y * kBT_R, c0 = blockIdx.x * kBT_C, tw = kBT_C + w - 1, th = kBT_R + h - 1;
  for (int e = threadIdx.x; e < th * tw; e += kT) {
    const int ty = e / tw, tx = e % tw;
    s_raw[e] = in[static_cast<size_t>(reflect101(r0 + ty - ay, rows)) * cols + reflect101(c0 + tx - ax, cols)];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < th * kBT_C; e += kT) {
    const int ty = e / kBT_C, tx = e % kBT_C;
    double s = 0;
    for (int k = 0; k < w; ++k) s += static_cast<double>(s_raw[ty * tw + tx + k]);
    s_rs[e] = s;
  }
  __syncthreads();
  const double scl = 1.0 / (static_cast<double>(w) * static_cast<double>(h));
  for (int e = threadIdx.x; e < kBT_R * kBT_C; e += kT) {
    const int ty = e / kBT_C, tx = e % kBT_C;
    if (r0 + ty >= rows || c0 + tx >= cols) continue;
    double s = 0;
    for (int k = 0; k < h; ++k) s += s_rs[(ty + k) * kBT_C + tx];
    const float b = static_cast<float>(s * scl) + 1.0f;
    const float v = s_raw[(ty + ay) * tw + tx + ax];
    out[static_cast<size_t>(r0 + ty) * cols + c0 + tx] = b != 0.f ? (v * 140.0f) / b : 0.f;
  }
}

// GaussianBlur 3 x 3 (:350-353, O4) + threshold TRUNC 255 (:298-300, O8)
constexpr int kGT_R = 16, kGT_C = 64;
__global__ __launch_bounds__(kT) void photo_gauss_trunc_kernel(const float * in, float * out, int rows, int cols, int do_gauss)
{
  __shared__ float s_raw[(kGT_R + 2) * (kGT_C + 2)];
  __shared__ float s_tmp[(kGT_R + 2) * kGT_C];
  const int r0 = blockIdx.y * kGT_R, c0 = blockIdx.x * kGT_C, tw = kGT_C + 2;
  for (int e = threadIdx.x; e < (kGT_R + 2) * tw; e += kT) {
    const int ty = e / tw, tx = e % tw;
    s_raw[e] = in[static_cast<size_t>(reflect101(r0 + ty - 1, rows)) * cols + reflect101(c0 + tx - 1, cols)];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < (kGT_R + 2) * kGT_C; e += kT) {
    const int ty = e / kGT_C, tx = e % kGT_C;
    const float l = s_raw[ty * tw + tx], c = s_raw[ty * tw + tx + 1], r = s_raw[ty * tw + tx + 2];
    s_tmp[e] = do_gauss ? 0.5f * c + 0.25f * (l + r) : c;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kGT_R * kGT_C; e += kT) {
    const int ty = e / kGT_C, tx = e % kGT_C;
    if (r0 + ty >= rows || c0 + tx >= cols) continue;
    const float u = s_tmp[ty * kGT_C + tx], c = s_tmp[(ty + 1) * kGT_C + tx], d = s_tmp[(ty + 2) * kGT_C + tx];
    float v = do_gauss ? 0.5f * c + 0.25f * (u + d) : c;
    v = v > 255.0f ? 255.0f : v;
    out[static_cast<size_t>(r0 + ty) * cols + c0 + tx] = v;
  }
}

// Sobel ksize 1, scale 0.5 (:316-317, O5) + corrected intensities back into the cloud (:307-314)
__global__ __launch_bounds__(kT) void photo_sobel_writeback_kernel(const float * img, float * dx, float * dy, const int32_t * idx,
                                                                    mh_point32 * desk, float * intensity_out, int rows, int cols)
{
  const int px = blockIdx.x * kT + threadIdx.x;
  if (px >= rows * cols) return;
  const int y = px / cols, x = px % cols;
  const float c = img[px];
  dx[px] = (img[y * cols + reflect101(x + 1, cols)] - img[y * cols + reflect101(x - 1, cols)]) * 0.5f;
  dy[px] = (img[reflect101(y + 1, rows) * cols + x] - img[reflect101(y - 1, rows) * cols + x]) * 0.5f;
  const int i = idx[px];
  if (i >= 0) {
    if (desk) desk[i].intensity = c;
    if (intensity_out) intensity_out[i] = c;
  }
}

// erode with a k x k ones kernel, anchor k / 2 (O6).  in &= static_mask (and the margin rectangle when margin >= 0:
// detectFeatures' `img_mask & mask_margin_`, photometric.cpp:524) before the erosion.
constexpr int kET_R = 16, kET_C = 64, kEMaxK = 33;
constexpr int kErodeLdsWords = ((kET_R + kEMaxK - 1) * (kET_C + kEMaxK - 1) + (kET_R + kEMaxK - 1) * kET_C + 3) / 4;
__device__ __forceinline__ void erode_tile(const uint8_t * in, const uint8_t * static_mask, int margin, uint8_t * out, int rows, int cols, int k,
                                           const int bx, const int by, uint32_t * pool, const uint32_t * stamp = nullptr, const uint32_t seq = 0)
{
  const int nthr = static_cast<int>(blockDim.x);
  uint8_t * s_raw = reinterpret_cast<uint8_t *>(pool);
  uint8_t * s_row = s_raw + (kET_R + kEMaxK - 1) * (kET_C + kEMaxK - 1);
  const int a = k / 2, r0 = by * kET_R, c0 = bx * kET_C, tw = kET_C + k - 1, th = kET_R + k - 1;
  for (int e = threadIdx.x; e < th * tw; e += nthr) {
    const int ty = e / tw, tx = e % tw;
    const int y = r0 + ty - a, x = c0 + tx - a;
    uint8_t v = 255;  // outside the image: never lowers the minimum
    if (y >= 0 && y < rows && x >= 0 && x < cols) {
      v = stamp ? (stamp[static_cast<size_t>(y) * cols + x] == seq ? 1 : 0) : in[static_cast<size_t>(y) * cols + x];
      if (static_mask && static_mask[static_cast<size_t>(y) * cols + x] == 0) v = 0;
      if (margin >= 0 && !(y >= margin && y < rows - margin && x >= margin && x < cols - margin)) v = 0;
    }
    s_raw[e] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < th * kET_C; e += nthr) {
    const int ty = e / kET_C, tx = e % kET_C;
    uint8_t mn = 255;
    for (int j = 0; j < k; ++j) mn = min(mn, s_raw[ty * tw + tx + j]);
    s_row[e] = mn;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kET_R * kET_C; e += nthr) {
    const int ty = e / kET_C, tx = e % kET_C;
    if (r0 + ty >= rows || c0 + tx >= cols) continue;
    uint8_t mn = 255;
    for (int j = 0; j < k; ++j) mn = min(mn, s_row[(ty + j) * kET_C + tx]);
    out[static_cast<size_t>(r0 + ty) * cols + c0 + tx] = mn;
  }
}
__global__ __launch_bounds__(kT) void photo_erode_kernel(const uint8_t * in, const uint8_t * static_mask, int margin, uint8_t * out,
                                                          int rows, int cols, int k)
{
  __shared__ uint32_t s_pool[kErodeLdsWords];
  erode_tile(in, static_mask, margin, out, rows, cols, k, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), s_pool);
}

// ------------------------------------------------------------------------------------------------
// Round 4: the preprocess chain in FIVE launches instead of thirteen.  A 128 x 1024 image is 512 KiB: every stage above is a
// 3-10 us kernel and the chain was launch-bound (2 % of HBM peak).  Two things cut the launches without adding work:
//   * stages that do not depend on each other share a launch (a workgroup finds its job by its index):
//       stage A = vertical FIR tiles | mask-erosion tiles | yaw-table rows          (all need only the scattered image)
//       stage B = horizontal FIR + subtract tiles | projection of the points        (the projection needs the yaw table)
//       stage C = brightness + Gaussian + Sobel tiles | projection-index finalise
//   * brightness, GaussianBlur 3 x 3 + TRUNC and Sobel + write-back run on ONE tile (16 x 64 outputs, a 2-pixel ring of
//     brightness values around it in LDS): the stages' reach is 1 pixel each, so the ring costs 1.3x the brightness work.
// (Tried first and rejected: the WHOLE chain on column strips with all rows in LDS — 32 workgroups with a 38-column halo per
// side for the 33-tap and 41-wide filters: 132 us, three times the six launches it replaced.)  Every stage evaluates the
// expression of its single-stage kernel in the same operation order: the images stay bit-identical to the oracle's.
// ------------------------------------------------------------------------------------------------
constexpr int kFT_R = 16, kFT_C = 64;
constexpr int kBgsLdsWords = (kFT_R + 4 + kBMaxH - 1) * (kFT_C + 4 + kBMaxW - 1) + 2 * (kFT_R + 4 + kBMaxH - 1) * (kFT_C + 4) + 2 * (kFT_R + 4) * (kFT_C + 4);
// filterBrightness -> GaussianBlur + TRUNC -> Sobel + write-back for the output tile (bx, by).  Each stage of the reference
// reflects ITS OWN input at the image border (BORDER_REFLECT_101 per filter): a neighbour outside the image is read at the
// reflected position of the same plane, which lies inside the tile's ring; ring positions outside the image are never computed.
__device__ __forceinline__ void bgs_tile(const float * in, float * fin, float * dx, float * dy, const int32_t * idx, mh_point32 * desk,
                                         float * intensity_out, int rows, int cols, int filter_brightness, int w, int h, int do_gauss, const int bx,
                                         const int by, uint32_t * pool)
{
  constexpr int DR = kFT_R + 4, DC = kFT_C + 4;
  const int ax = filter_brightness ? w / 2 : 0, ay = filter_brightness ? h / 2 : 0;
  const int r0 = by * kFT_R - 2, c0 = bx * kFT_C - 2;  // image position of ring position (0, 0)
  const int tw = DC + 2 * ax, th = DR + 2 * ay;
  float * s_raw = reinterpret_cast<float *>(pool);                                   // th x tw: input with the window halo
  double * s_rs = reinterpret_cast<double *>(s_raw + (kFT_R + 4 + kBMaxH - 1) * (kFT_C + 4 + kBMaxW - 1));  // th x DC row sums
  float * s_D = reinterpret_cast<float *>(s_rs + (kFT_R + 4 + kBMaxH - 1) * (kFT_C + 4));                   // DR x DC
  float * s_T = s_D + DR * DC;                                                                               // DR x DC
  const int nthr = static_cast<int>(blockDim.x), tid = static_cast<int>(threadIdx.x);
  // (tw is a run-time value: the row of a flat index by a float reciprocal — exact for these sizes — instead of an integer division)
  const float inv_tw = 1.0f / static_cast<float>(tw);
  for (int e = tid; e < th * tw; e += nthr) {
    int ty = static_cast<int>((static_cast<float>(e) + 0.5f) * inv_tw);
    int tx = e - ty * tw;
    if (tx < 0) {
      --ty;
      tx += tw;
    } else if (tx >= tw) {
      ++ty;
      tx -= tw;
    }
    s_raw[e] = in[static_cast<size_t>(reflect101(r0 + ty - ay, rows)) * cols + reflect101(c0 + tx - ax, cols)];
  }
  __syncthreads();
  auto inside = [&](int ty, int tx) { return r0 + ty >= 0 && r0 + ty < rows && c0 + tx >= 0 && c0 + tx < cols; };
  if (filter_brightness) {
    for (int e = tid; e < th * DC; e += nthr) {
      const int ty = e / DC, tx = e - ty * DC;
      double s = 0;
      const float * src = s_raw + ty * tw + tx;
      for (int k = 0; k < w; ++k) s += static_cast<double>(src[k]);
      s_rs[e] = s;
    }
    __syncthreads();
    const double scl = 1.0 / (static_cast<double>(w) * static_cast<double>(h));
    for (int e = tid; e < DR * DC; e += nthr) {
      const int ty = e / DC, tx = e - ty * DC;
      if (!inside(ty, tx)) continue;
      double s = 0;
      for (int k = 0; k < h; ++k) s += s_rs[(ty + k) * DC + tx];
      const float b = static_cast<float>(s * scl) + 1.0f;
      const float v = s_raw[(ty + ay) * tw + tx + ax];
      s_D[e] = b != 0.f ? (v * 140.0f) / b : 0.f;
    }
  } else {
    for (int e = tid; e < DR * DC; e += nthr) {
      const int ty = e / DC, tx = e - ty * DC;
      if (inside(ty, tx)) s_D[e] = s_raw[ty * tw + tx];
    }
  }
  __syncthreads();
  auto trow = [&](int y) { return reflect101(y, rows) - r0; };
  auto tcol = [&](int x) { return reflect101(x, cols) - c0; };
  // GaussianBlur 3 x 3: horizontal into s_T (all ring rows), vertical back into s_D (ring rows 1 .. DR - 2), TRUNC 255
  for (int e = tid; e < DR * DC; e += nthr) {
    const int ty = e / DC, tx = e - ty * DC;
    if (!inside(ty, tx) || tx < 1 || tx >= DC - 1) continue;
    const float l = s_D[ty * DC + tcol(c0 + tx - 1)], c = s_D[e], r = s_D[ty * DC + tcol(c0 + tx + 1)];
    s_T[e] = do_gauss ? 0.5f * c + 0.25f * (l + r) : c;
  }
  __syncthreads();
  for (int e = tid; e < DR * DC; e += nthr) {
    const int ty = e / DC, tx = e - ty * DC;
    if (!inside(ty, tx) || tx < 1 || tx >= DC - 1 || ty < 1 || ty >= DR - 1) continue;
    const float u = s_T[trow(r0 + ty - 1) * DC + tx], c = s_T[e], d = s_T[trow(r0 + ty + 1) * DC + tx];
    float v = do_gauss ? 0.5f * c + 0.25f * (u + d) : c;
    v = v > 255.0f ? 255.0f : v;
    s_D[e] = v;
  }
  __syncthreads();
  // Sobel ksize 1, scale 0.5 + the corrected intensities back into the cloud
  for (int e = tid; e < kFT_R * kFT_C; e += nthr) {
    const int oy = e / kFT_C, ox = e - oy * kFT_C, ty = oy + 2, tx = ox + 2;
    const int y = r0 + ty, x = c0 + tx;
    if (y >= rows || x >= cols) continue;
    const size_t px = static_cast<size_t>(y) * cols + x;
    const float c = s_D[ty * DC + tx];
    fin[px] = c;
    dx[px] = (s_D[ty * DC + tcol(x + 1)] - s_D[ty * DC + tcol(x - 1)]) * 0.5f;
    dy[px] = (s_D[trow(y + 1) * DC + tx] - s_D[trow(y - 1) * DC + tx]) * 0.5f;
    const int i = idx[px];
    if (i >= 0) {
      if (desk) desk[i].intensity = c;
      if (intensity_out) intensity_out[i] = c;
    }
  }
}

struct PhotoStageArgs  // one argument block for the three multi-job launches
{
  const float * raw;   // scattered intensity image
  float * ta;          // high-passed image (stage A -> B)
  float * tb;          // line-free image (stage B -> C), or prep(raw) when removeLines is off
  float * fin;
  float * dx;
  float * dy;
  const int32_t * idx;
  float * intensity_out;
  const float * hp;
  const float * lp;
  const uint8_t * static_mask;
  uint8_t * mask_out;
  float * yaw;
  const mh_point32 * desk;
  int32_t * proj;
  PhotoCounters * counters;
  int rows, cols, n_pts, n_hp, n_lp, remove_lines, filter_brightness, bw, bh, do_gauss, erode_k;
  float scale, gamma;
  int n_job0, n_job1;  // workgroups of the launch's first / second job (the rest run the third)
  // round 5 (no reset launch): frame stamps [yaw | pixel] of rows x cols words each, the frame's sequence number, the arrays
  // stage A's fix-up job completes, and the frame's pose table (mapped pinned block -> device, n_copy16 x 16 bytes)
  const uint32_t * stamps;
  uint32_t seq;
  float * raw_w;
  float * range;
  int32_t * idx_w;
  const uint4 * copy_src;
  uint4 * copy_dst;
  int n_copy16, n_job2, n_job3, n_job4;
  ProjPre * pre;  // per-point records of the projection's front (stage A) for its back (stage B)
  mh_point32 * desk_writeback;  // stage C: the corrected intensities also go into this cloud (a scan's resident one), or nullptr
};
// Workgroups of the three stage launches: 1024 threads.  The tiles are the ones of the 256-thread single-stage kernels; a
// tile's phases are loops over its elements in steps of the workgroup size, so four times the threads means a quarter of the
// dependent iterations per thread and sixteen waves per CU to hide the LDS and memory latency behind (the chain is
// latency-bound: 512 KiB images).
constexpr int kTS = 1024;
constexpr int kStageALds = kYawLdsWords > kVfirLdsWords ? kYawLdsWords : kVfirLdsWords;
__global__ __launch_bounds__(kTS) void photo_stage_a_kernel(const PhotoStageArgs a, const PhotoModel m)
{
  __shared__ uint32_t s_pool[kStageALds > kErodeLdsWords ? kStageALds : kErodeLdsWords];
  const int b = static_cast<int>(blockIdx.x), npx = a.rows * a.cols;
  const uint32_t * ystamp = a.stamps, * pstamp = a.stamps + npx;
  if (b < a.n_job0) {  // vertical high-pass tiles (or the plain intensity scaling when removeLines is off)
    if (a.remove_lines) {
      const int gx = (a.cols + kVT_C - 1) / kVT_C;
      vfir_tile(a.raw, a.ta, a.rows, a.cols, a.hp, a.n_hp, a.scale, a.gamma, b % gx, b / gx, s_pool, pstamp, a.seq);
    } else {
      const int i = b * kTS + static_cast<int>(threadIdx.x);
      if (i < npx) a.tb[i] = prep(pstamp[i] == a.seq ? a.raw[i] : 0.f, a.scale, a.gamma);
    }
  } else if (b < a.n_job0 + a.n_job1) {  // mask erosion tiles
    const int t = b - a.n_job0, gx = (a.cols + kET_C - 1) / kET_C;
    erode_tile(nullptr, a.static_mask, -1, a.mask_out, a.rows, a.cols, a.erode_k, t % gx, t / gx, s_pool, pstamp, a.seq);
  } else if (b < a.n_job0 + a.n_job1 + a.n_job2) {  // yaw-table rows
    yaw_fill_row(m, a.yaw, nullptr, b - a.n_job0 - a.n_job1, s_pool, ystamp, a.seq);
  } else if (b < a.n_job0 + a.n_job1 + a.n_job2 + a.n_job3) {  // the reset values of the pixels no point claimed
    const int i = (b - a.n_job0 - a.n_job1 - a.n_job2) * kTS + static_cast<int>(threadIdx.x);
    if (i < npx && pstamp[i] != a.seq) {
      a.raw_w[i] = 0.f;
      a.range[i] = 0.f;
      a.idx_w[i] = -1;
    }
  } else if (b < a.n_job0 + a.n_job1 + a.n_job2 + a.n_job3 + a.n_job4) {  // the projection's front: everything that needs no yaw table
    project_point_front(m, a.desk, a.n_pts, a.pre, (b - a.n_job0 - a.n_job1 - a.n_job2 - a.n_job3) * kTS + static_cast<int>(threadIdx.x));
  } else {  // the frame's pose table from its mapped pinned block (read by the factors, not by this chain)
    const int nb = static_cast<int>(gridDim.x) - (a.n_job0 + a.n_job1 + a.n_job2 + a.n_job3 + a.n_job4);
    for (int k = (b - (static_cast<int>(gridDim.x) - nb)) * kTS + static_cast<int>(threadIdx.x); k < a.n_copy16; k += nb * kTS) a.copy_dst[k] = a.copy_src[k];
  }
}
__global__ __launch_bounds__(kTS) void photo_stage_b_kernel(const PhotoStageArgs a, const PhotoModel m)
{
  __shared__ uint32_t s_pool[kHfirLdsWords];
  const int b = static_cast<int>(blockIdx.x);
  if (b < a.n_job0) {
    const int gx = (a.cols + kHT_C - 1) / kHT_C;
    hfir_sub_tile(a.ta, a.raw, a.tb, a.rows, a.cols, a.lp, a.n_lp, a.scale, a.gamma, b % gx, b / gx, s_pool);
  } else {
    project_point_back(m, a.n_pts, a.yaw, a.proj, a.counters, a.pre, (b - a.n_job0) * kTS + static_cast<int>(threadIdx.x));
  }
}
__global__ __launch_bounds__(kTS) void photo_stage_c_kernel(const PhotoStageArgs a)
{
  __shared__ __attribute__((aligned(8))) uint32_t s_pool[kBgsLdsWords];
  const int b = static_cast<int>(blockIdx.x);
  if (b < a.n_job0) {
    const int gx = (a.cols + kFT_C - 1) / kFT_C;
    bgs_tile(a.tb, a.fin, a.dx, a.dy, a.idx, a.desk_writeback, a.intensity_out, a.rows, a.cols, a.filter_brightness, a.bw, a.bh, a.do_gauss, b % gx, b / gx, s_pool);
  } else {
    proj_finalize_px(a.rows * a.cols, a.proj, (b - a.n_job0) * kTS + static_cast<int>(threadIdx.x));
  }
}

// convertScaleAbs + addWeighted(.5, .5) (photometric.cpp:536-540, O9): round half to even, saturate
__device__ __forceinline__ uint8_t sat_u8(float v)
{
  const float r = rintf(v);
  return static_cast<uint8_t>(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}
__global__ __launch_bounds__(kT) void photo_grad_kernel(const float * dx, const float * dy, uint8_t * grad, int n)
{
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const uint8_t ax = sat_u8(fabsf(dx[i])), ay = sat_u8(fabsf(dy[i]));
  grad[i] = sat_u8(static_cast<float>(ax) * 0.5f + static_cast<float>(ay) * 0.5f);
}

// ------------------------------------------------------------------------------------------------
// PhotometricFactor::linearize — one wave per feature, lane = patch point
// ------------------------------------------------------------------------------------------------
// wave-wide sums: DPP (wave_dpp.hpp), every lane gets the total.  (Round 5: the 43 butterflies of __shfl_xor on doubles of a
// unary feature — 516 ds_bpermute — were a third of the kernel's 30 us: 29.8 -> 19.4-20.1 us.  On top, same box (gpurun c31):
// the altitude table in LDS 19.1, the pose index by interpolation guess 18.8 (both kept); the yaw window of project() requested
// at once and searched in registers 21.6 (eleven loads and forty selects cost more than four dependent probes: not kept).)
__device__ __forceinline__ double wave_sum(double v) { return wave_allsum_f64(v); }

__device__ __forceinline__ double bilinear(const float * img, int cols, double x, double y)  // photometric_utils.cpp:368-388
{
  const int x0 = static_cast<int>(floor(x)), y0 = static_cast<int>(floor(y));
  const double dx = x - x0, dy = y - y0;
  const float * r0 = img + static_cast<size_t>(y0) * cols + x0;
  const float * r1 = r0 + cols;
  return (1 - dx) * (1 - dy) * static_cast<double>(r0[0]) + dx * (1 - dy) * static_cast<double>(r0[1]) +
         (1 - dx) * dy * static_cast<double>(r1[0]) + dx * dy * static_cast<double>(r1[1]);
}

enum
{
  PS_UNPROCESSED = 0,
  PS_PROJECT_UNDISTORTED,
  PS_RANGE,
  PS_PROJECT,
  PS_MASK,
  PS_MASK_MARGIN,
  PS_RANGE_DIFF,
  PS_MAX_ERROR,
  PS_VALID
};

__device__ __forceinline__ void mat3_vec(const double * R, double x, double y, double z, double & ox, double & oy, double & oz)
{
  ox = R[0] * x + (R[1] * y + R[2] * z);
  oy = R[3] * x + (R[4] * y + R[5] * z);
  oz = R[6] * x + (R[7] * y + R[8] * z);
}

constexpr int kFeatPerBlock = 4;
// one wave = one feature; returns as soon as the feature's status is known
__device__ __forceinline__ void photo_linearize_feature(const PhotoLinArgs & a, const float * s_alt)
{
  const int lane = threadIdx.x & 63, f = blockIdx.x * kFeatPerBlock + (threadIdx.x >> 6);
  if (f >= a.n_features) return;  // whole waves
  const PhotoModel & m = a.model;
  const PhotoFrameView & fr = a.frame;
  if (a.rows_out) {  // rows of features that do not end Valid read as zero
    double * ro = a.rows_out + (static_cast<size_t>(f) * kPhotoMaxPatch + lane) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) ro[k] = 0.0;
  }
  const int npts = a.n_pts[f];
  const bool act = lane < npts;
  const double * Lp = a.Le_ps + (static_cast<size_t>(f) * kPhotoMaxPatch + (act ? lane : 0)) * 3;
  const double ax = Lp[0], ay = Lp[1], az = Lp[2];
  int st = PS_UNPROCESSED;
  int threw = 0;  // 1: project() would throw, 2: interpolated_map_T_Le_Lt.at() would throw
  double ux = 0, uy = 0, lx = 0, ly = 0, lz = 0, Ib = 0;
  double TR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (act) {
    // p_Le_b = delta_pose_b_a_Le * Le_p (:168)
    double bx, by, bz;
    mat3_vec(a.dLe_R, ax, ay, az, bx, by, bz);
    bx += a.dLe_t[0];
    by += a.dLe_t[1];
    bz += a.dLe_t[2];
    // projectUndistorted (photometric_utils.cpp:287-366)
    double kx, ky;
    int r = project_yaw(m, fr.yaw, bx, by, bz, kx, ky, s_alt);
    bool ok = r > 0;
    if (r < 0) threw = 1;
    int dist_idx = -1;
    if (ok) {
      kx = round(kx);
      ky = round(ky);
      int row = static_cast<int>(ky);
      const int col = static_cast<int>(kx);
      const int32_t * pj = fr.proj + (static_cast<size_t>(row) * m.cols + col) * kPhotoDup;
      if (pj[0] == 0) {  // search the column for any row with projections (:305-313)
        row = 0;
        for (; row < m.rows; ++row) {
          pj = fr.proj + (static_cast<size_t>(row) * m.cols + col) * kPhotoDup;
          if (pj[0] > 0) break;
        }
        if (row >= m.rows) ok = false;
      }
      if (ok) {
        if (pj[0] > 1) {
          float min_sq = 3.402823466e+38f;
          for (int i = 1; i <= pj[0]; ++i) {
            const int j = pj[i];
            const mh_point32 q = fr.points[j];
            const double ddx = bx - static_cast<double>(q.x), ddy = by - static_cast<double>(q.y), ddz = bz - static_cast<double>(q.z);
            const float sq = static_cast<float>(ddx * ddx + (ddy * ddy + ddz * ddz));
            if (sq < min_sq) {
              min_sq = sq;
              dist_idx = j;
            }
          }
        } else {
          dist_idx = pj[1];
        }
        if (dist_idx < 0) ok = false;
      }
    }
    if (ok) {
      // T_Le_Lt = interpolated_map_T_Le_Lt.at(points_deskewed[distortion_idx].t)
      const uint32_t ns = fr.points[dist_idx].t;
      // lower bound of ns in the sorted timestamps.  The columns fire at a constant rate, so the index is guessed by linear
      // interpolation between the first and the last timestamp and verified (one round trip instead of ten dependent ones);
      // whatever the guess misses goes through the binary search.
      int lo = 0, hi = fr.n_poses;
      if (fr.n_poses > 1) {
        const uint32_t t0 = fr.pose_ns[0], t1 = fr.pose_ns[fr.n_poses - 1];
        if (ns >= t0 && ns <= t1 && t1 > t0) {
          const int gq = static_cast<int>((static_cast<double>(ns - t0) / static_cast<double>(t1 - t0)) * static_cast<double>(fr.n_poses - 1) + 0.5);
          const int gl = gq > 0 ? gq - 1 : 0, gh = gq + 1 < fr.n_poses ? gq + 1 : fr.n_poses - 1;
          const uint32_t a0 = fr.pose_ns[gl], a1 = fr.pose_ns[gq], a2 = fr.pose_ns[gh];  // three neighbours, requested together
          if (a1 == ns && (gq == 0 || a0 < ns))
            lo = hi = gq;  // the first entry equal to ns
          else if (a0 == ns && (gl == 0 || fr.pose_ns[gl - 1] < ns))
            lo = hi = gl;
          else if (a2 == ns && a1 < ns)
            lo = hi = gh;
        }
      }
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fr.pose_ns[mid] < ns)
          lo = mid + 1;
        else
          hi = mid;
      }
      if (lo >= fr.n_poses || fr.pose_ns[lo] != ns) {
        threw = 2;
        ok = false;
      } else {
        const double * T = fr.pose_Rt + static_cast<size_t>(lo) * 12;
#pragma unroll
        for (int q = 0; q < 9; ++q) TR[q] = T[q];
        // Li_p = T_Le_Lt.inverse() * Le_p:  R^T p + (-(R^T t))   (gtsam Pose3::inverse then act)
        const double itx = -(TR[0] * T[9] + (TR[3] * T[10] + TR[6] * T[11])), ity = -(TR[1] * T[9] + (TR[4] * T[10] + TR[7] * T[11])),
                     itz = -(TR[2] * T[9] + (TR[5] * T[10] + TR[8] * T[11]));
        lx = (TR[0] * bx + (TR[3] * by + TR[6] * bz)) + itx;
        ly = (TR[1] * bx + (TR[4] * by + TR[7] * bz)) + ity;
        lz = (TR[2] * bx + (TR[5] * by + TR[8] * bz)) + itz;
        r = project_yaw(m, fr.yaw, lx, ly, lz, ux, uy, s_alt);
        if (r < 0) threw = 1;
        ok = r > 0;
      }
    }
    if (!ok) {
      st = PS_PROJECT_UNDISTORTED;
    } else {
      const double rng = sqrt(lx * lx + (ly * ly + lz * lz));
      const int rx = static_cast<int>(round(ux)), ry = static_cast<int>(round(uy));
      if (rng < static_cast<double>(m.range_min) || rng > static_cast<double>(m.range_max)) {
        st = PS_RANGE;
      } else if (!fr.mask[static_cast<size_t>(ry) * m.cols + rx]) {
        st = PS_MASK;
      } else if (rx < m.margin_size || rx >= m.cols - m.margin_size || ry < m.margin_size || ry >= m.rows - m.margin_size) {
        st = PS_MASK_MARGIN;
      } else if (fabs(static_cast<double>(fr.range[static_cast<size_t>(ry) * m.cols + rx]) - rng) >
                 static_cast<double>(m.occlusion_range_diff_threshold)) {
        st = PS_RANGE_DIFF;
      } else if (ux > m.cols - 2 || uy > m.rows - 2) {
        st = PS_PROJECT_UNDISTORTED;  // bilinear footprint outside the image: the reference's getSubPixelValue reads out of bounds
      } else {
        Ib = bilinear(fr.intensity, m.cols, ux, uy);
      }
    }
  }
  // the loop over the patch breaks at the FIRST failing point (:170-216): its status is the feature's
  const unsigned long long failing = __ballot(act && st != PS_UNPROCESSED);
  if (failing) {
    const int first = __ffsll(static_cast<long long>(failing)) - 1;
    const int fst = __shfl(st, first, 64);
    if (lane == 0) a.status[f] = fst;
    // only the first failing point is ever evaluated by the reference: count ITS exception, if any
    if (lane == first && threw) atomicAdd(threw == 1 ? &a.counters->project_throw : &a.counters->pose_missing, 1u);
    return;
  }
  // getPsi (photometric_utils.cpp:13-19)
  const double md = static_cast<double>(npts);
  const double mean = wave_sum(act ? Ib : 0.0) / md;
  const double cen = act ? Ib - mean : 0.0;
  const double sigma = sqrt(wave_sum(cen * cen));
  const double psi = cen / sigma;
  const double e0 = act ? psi - a.psi_a[static_cast<size_t>(f) * kPhotoMaxPatch + lane] : 0.0;
  const double e2 = wave_sum(e0 * e0);
  const double e_ncc = (2 - e2) / 2;
  if (e_ncc < a.max_error) {
    if (lane == 0) a.status[f] = PS_MAX_ERROR;
    return;
  }
  {
    const int cl = npts / 2;  // a_feature.center = uv_bs[uv_bs.size() / 2]
    const double cxv = __shfl(ux, cl, 64), cyv = __shfl(uy, cl, 64);
    if (lane == 0) {
      a.status[f] = PS_VALID;
      a.centers[2 * f] = cxv;
      a.centers[2 * f + 1] = cyv;
    }
  }
  // Jacobian rows: dI/duv * duv/dp * dp/dT  (:243-279)
  double Db[6] = {0, 0, 0, 0, 0, 0}, Da[6] = {0, 0, 0, 0, 0, 0};
  if (act) {
    const double gx = bilinear(fr.dx, m.cols, ux, uy), gy = bilinear(fr.dy, m.cols, ux, uy);
    // getProjectionJacobian (photometric_utils.cpp:186-198)
    const double rxy = sqrt(lx * lx + ly * ly);
    const double L = rxy - static_cast<double>(m.beam_offset_m);
    const double R2 = L * L + lz * lz;
    const double irxy = 1.0 / rxy;
    const double fx_irxy2 = m.fx * (irxy * irxy);
    const double den = (L + static_cast<double>(m.beam_offset_m)) * R2;
    const double P0 = -fx_irxy2 * ly, P1 = fx_irxy2 * lx, P3 = -m.fy * lx * lz / den, P4 = -m.fy * ly * lz / den, P5 = m.fy * L / R2;
    const double g0 = gx * P0 + gy * P3, g1 = gx * P1 + gy * P4, g2 = gy * P5;
    // p_Be_a = T_B_L * Le_p;  p_Be_b = delta_pose_b_a_Be * p_Be_a
    double pax, pay, paz, pbx, pby, pbz;
    mat3_vec(a.TBL_R, ax, ay, az, pax, pay, paz);
    pax += a.TBL_t[0];
    pay += a.TBL_t[1];
    paz += a.TBL_t[2];
    mat3_vec(a.dBe_R, pax, pay, paz, pbx, pby, pbz);
    pbx += a.dBe_t[0];
    pby += a.dBe_t[1];
    pbz += a.dBe_t[2];
    // R_Lk_b_Be_b = R_Le_Lt^T * R_B_L^T ;  w = g^T R  (1 x 3)
    double Rk[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rk[3 * i + j] = TR[i] * a.TBL_R[3 * j] + (TR[3 + i] * a.TBL_R[3 * j + 1] + TR[6 + i] * a.TBL_R[3 * j + 2]);
    const double w0 = g0 * Rk[0] + g1 * Rk[3] + g2 * Rk[6], w1 = g0 * Rk[1] + g1 * Rk[4] + g2 * Rk[7],
                 w2 = g0 * Rk[2] + g1 * Rk[5] + g2 * Rk[8];
    // w * Hat(p) = (p x w)^T ... row vector times skew: (w^T [p]x)_k = (w x p)_k with sign: w^T [p]x = (p x w)^T * (-1)?
    // [p]x v = p x v, so w^T [p]x = -(p x w)^T... computed explicitly:
    Db[0] = w1 * pbz - w2 * pby;  // (w^T Hat(p))_0 = w1 * p_z - w2 * p_y
    Db[1] = w2 * pbx - w0 * pbz;
    Db[2] = w0 * pby - w1 * pbx;
    Db[3] = -w0;
    Db[4] = -w1;
    Db[5] = -w2;
    if (a.binary) {
      // R_Lk_b_Be_a = R_Lk_b_Be_b * R(delta_pose_b_a_Be);  wa = g^T R_a
      const double wa0 = w0 * a.dBe_R[0] + w1 * a.dBe_R[3] + w2 * a.dBe_R[6], wa1 = w0 * a.dBe_R[1] + w1 * a.dBe_R[4] + w2 * a.dBe_R[7],
                   wa2 = w0 * a.dBe_R[2] + w1 * a.dBe_R[5] + w2 * a.dBe_R[8];
      Da[0] = -(wa1 * paz - wa2 * pay);
      Da[1] = -(wa2 * pax - wa0 * paz);
      Da[2] = -(wa0 * pay - wa1 * pax);
      Da[3] = wa0;
      Da[4] = wa1;
      Da[5] = wa2;
    }
  }
  // J = ((I - psi psi^T) / sigma) (I - 1 1^T / m) D   (getPsiJacobian, photometric_utils.cpp:21-27)
  const double whitened = sqrt(e2) / a.sigma;
  double sw = 1.0;
  if (a.use_robust) {
    const double p = a.robust_param;
    sw = a.robust_is_huber ? (fabs(whitened) <= p ? 1.0 : sqrt(p / fabs(whitened))) : p * p / (p * p + whitened * whitened);
  }
  const double wgt = sw / a.sigma;
  double row[13];
  auto psi_rows = [&](const double (&D)[6], const int at) {  // six columns at a time: two 6-wide sums instead of twelve single ones
    double cm[6], dc[6], pd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cm[k] = D[k];
    wave_allsum_f64<6>(cm);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      dc[k] = act ? D[k] - cm[k] / md : 0.0;
      pd[k] = psi * dc[k];
    }
    wave_allsum_f64<6>(pd);
#pragma unroll
    for (int k = 0; k < 6; ++k) row[at + k] = act ? ((dc[k] - psi * pd[k]) / sigma) * wgt : 0.0;
  };
  psi_rows(Db, 0);
  if (a.binary) {
    psi_rows(Da, 6);
    row[12] = e0 * wgt;
  } else {
    row[6] = e0 * wgt;
  }
  if (a.rows_out && act) {
    double * ro = a.rows_out + (static_cast<size_t>(f) * kPhotoMaxPatch + lane) * 8;
    ro[0] = e0 * wgt;
#pragma unroll
    for (int k = 0; k < 6; ++k) ro[1 + k] = row[k];
    ro[7] = 1.0;
  }
  // per-feature sums of v v^T (upper triangle), v = [J_b (, J_a), e]: 28 / 91 products, seven sums at a time
  double * part = a.partials + static_cast<size_t>(f) * kPhotoPartial;
  auto triangle = [&](auto nv_tag) {
    constexpr int NV = decltype(nv_tag)::value, NE = NV * (NV + 1) / 2;
    static_assert(NE % 7 == 0, "28 and 91 are multiples of 7");
    double prod[NE];
    int ent = 0;
#pragma unroll
    for (int r = 0; r < NV; ++r)
#pragma unroll
      for (int c = r; c < NV; ++c) prod[ent++] = row[r] * row[c];
#pragma unroll
    for (int b = 0; b < NE / 7; ++b) {
      double v[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) v[j] = prod[7 * b + j];
      wave_allsum_f64<7>(v);
      if (lane < 7) {
        double s = v[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) s = lane == j ? v[j] : s;
        part[7 * b + lane] = s;
      }
    }
  };
  if (a.binary)
    triangle(std::integral_constant<int, 13>{});
  else
    triangle(std::integral_constant<int, 7>{});
}

constexpr int kAltLds = 512;  // beam altitudes kept in LDS by the factor kernel (taller sensors read the table in memory)
__global__ __launch_bounds__(64 * kFeatPerBlock) void photo_linearize_kernel(const PhotoLinArgs a)
{
  __shared__ float s_alt[kAltLds];
  const bool alt_lds = a.model.rows <= kAltLds;
  if (alt_lds) {
    for (int i = threadIdx.x; i < a.model.rows; i += 64 * kFeatPerBlock) s_alt[i] = a.model.alt[i];
    __syncthreads();
  }
  photo_linearize_feature(a, alt_lds ? s_alt : nullptr);
  // Completion number for a host that spins on the mapped block instead of paying a stream synchronisation (as K4 does
  // for the ICP factor): every block makes its host writes visible (system-scope fence), takes a ticket, the last one
  // re-arms the ticket and publishes.
  if (a.seq) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int prev = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gridDim.x - 1) {
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// detectFeatures' candidate list (photometric.cpp:541-555): the pixels with mask != 0 and gradient > threshold, in
// row-major order (the order the reference pushes them in, which decides how std::sort arranges equal gradients).
// Two-kernel order-preserving compaction: per-block counts, then block prefix + in-block scan.  Entry = px | grad << 24.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_candidate(const uint8_t * grad, const uint8_t * mask, int px, float thr)
{
  return mask[px] != 0 && static_cast<float>(grad[px]) > thr;
}
__global__ __launch_bounds__(kT) void photo_cand_count_kernel(const uint8_t * grad, const uint8_t * mask, int npx, float thr, uint32_t * blk)
{
  const int px = blockIdx.x * kT + threadIdx.x;
  const bool c = px < npx && is_candidate(grad, mask, px, thr);
  const int n = __syncthreads_count(c ? 1 : 0);
  if (threadIdx.x == 0) blk[blockIdx.x] = static_cast<uint32_t>(n);
}
__global__ __launch_bounds__(kT) void photo_cand_scatter_kernel(const uint8_t * grad, const uint8_t * mask, int npx, float thr,
                                                                 const uint32_t * blk, uint32_t * out, uint32_t * n_out)
{
  __shared__ uint32_t s_w[kT / 64], s_off;
  uint32_t off = 0;
  for (int i = threadIdx.x; i < static_cast<int>(blockIdx.x); i += kT) off += blk[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) off += __shfl_xor(off, d);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = off;
  __syncthreads();
  if (threadIdx.x == 0) s_off = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  const int px = blockIdx.x * kT + threadIdx.x;
  const bool c = px < npx && is_candidate(grad, mask, px, thr);
  const uint64_t m = __ballot(c);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_w[wave] = static_cast<uint32_t>(__popcll(m));
  __syncthreads();
  uint32_t before = s_off;
  for (uint32_t w = 0; w < wave; ++w) before += s_w[w];
  if (c) out[before + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)))] = static_cast<uint32_t>(px) | (static_cast<uint32_t>(grad[px]) << 24);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = s_off + s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// What the host-side selection needs of each candidate that survived the non-maximum suppression: the 7 x 7 intensity
// window around it (cornerEigenValsAndVecs, O11), and for the patch pixels + the centre: point index, point, intensity.
// uv: n_off patch offsets, then n_cand candidate centres — or, per_candidate (rotate_patch_to_align_with_gradient: every
// candidate has its own rotated pattern), n_cand x n_off offsets, then the centres.  One block per candidate.
__global__ __launch_bounds__(128) void photo_gather_kernel(const int2 * uv, int n_off, int n_cand, int per_candidate, const float * I,
                                                           const int32_t * idx, const mh_point32 * pts, int rows, int cols, float * win49,
                                                           float4 * rec, int32_t * rec_idx)
{
  const int c = blockIdx.x;
  if (c >= n_cand) return;
  const int2 ctr = uv[(per_candidate ? n_cand * n_off : n_off) + c];
  const int2 * offs = uv + (per_candidate ? c * n_off : 0);
  const int t = threadIdx.x;
  if (t < 49) {
    const int u = min(max(ctr.x + (t % 7) - 3, 0), cols - 1), v = min(max(ctr.y + (t / 7) - 3, 0), rows - 1);
    win49[static_cast<size_t>(c) * 49 + t] = I[static_cast<size_t>(v) * cols + u];
  }
  if (t <= n_off) {
    const int u = ctr.x + (t < n_off ? offs[t].x : 0), v = ctr.y + (t < n_off ? offs[t].y : 0);
    int32_t pi = -1;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u >= 0 && u < cols && v >= 0 && v < rows) {
      pi = idx[static_cast<size_t>(v) * cols + u];
      r.w = I[static_cast<size_t>(v) * cols + u];
      if (pi >= 0) {
        const mh_point32 p = pts[pi];
        r.x = p.x;
        r.y = p.y;
        r.z = p.z;
      }
    }
    rec[static_cast<size_t>(c) * (n_off + 1) + t] = r;
    rec_idx[static_cast<size_t>(c) * (n_off + 1) + t] = pi;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline dim3 g1(int n) { return dim3((n + kT - 1) / kT); }

hipError_t launch_photo_clear(int npx, int n_pts, float * img_raw, float * range, uint8_t * mask_raw, uint8_t * yaw_valid,
                              int32_t * idx, int32_t * proj, float * int_out, const void * copy_src, void * copy_dst, size_t copy_bytes, hipStream_t stream)
{
  const int n = npx > n_pts ? npx : n_pts;
  const int nb = (n + kT - 1) / kT, n16 = static_cast<int>(copy_bytes / 16), ncopy = n16 ? (n16 + kT - 1) / kT : 0;
  hipLaunchKernelGGL(photo_clear_kernel, dim3(nb + (ncopy > 16 ? 16 : ncopy)), dim3(kT), 0, stream, npx, n_pts, img_raw, range, mask_raw, yaw_valid, idx, proj, int_out, nb,
                     static_cast<const uint4 *>(copy_src), static_cast<uint4 *>(copy_dst), n16);
  return hipGetLastError();
}
hipError_t launch_photo_scatter(const PhotoModel & m, const mh_point32 * raw, const mh_point32 * desk, int n, float * yaw,
                                uint8_t * yaw_valid, float * intensity, float * range, uint8_t * mask, int32_t * idx,
                                hipStream_t stream)
{
  if (n > 0) hipLaunchKernelGGL(photo_scatter_kernel, g1(n), dim3(kT), 0, stream, m, raw, desk, n, yaw, yaw_valid, intensity, range, mask, idx);
  return hipGetLastError();
}
hipError_t launch_photo_yaw_fill(const PhotoModel & m, float * yaw, const uint8_t * yaw_valid, hipStream_t stream)
{
  hipLaunchKernelGGL(photo_yaw_fill_kernel, dim3(m.rows), dim3(kT), 0, stream, m, yaw, yaw_valid);
  return hipGetLastError();
}
hipError_t launch_photo_project(const PhotoModel & m, const mh_point32 * desk, int n, const float * yaw, int32_t * proj,
                                PhotoCounters * counters, hipStream_t stream)
{
  if (n > 0) hipLaunchKernelGGL(photo_project_kernel, g1(n), dim3(kT), 0, stream, m, desk, n, yaw, proj, counters);
  return hipGetLastError();
}
hipError_t launch_photo_proj_finalize(int n_pixels, int32_t * proj, hipStream_t stream)
{
  hipLaunchKernelGGL(photo_proj_finalize_kernel, g1(n_pixels), dim3(kT), 0, stream, n_pixels, proj);
  return hipGetLastError();
}
hipError_t launch_photo_vfir(const float * in, float * out, int rows, int cols, const float * taps, int n_taps, float scale,
                             float gamma, hipStream_t stream)
{
  const dim3 grid((cols + kVT_C - 1) / kVT_C, (rows + kVT_R - 1) / kVT_R);
  hipLaunchKernelGGL(photo_vfir_kernel, grid, dim3(kT), 0, stream, in, out, rows, cols, taps, n_taps, scale, gamma);
  return hipGetLastError();
}
hipError_t launch_photo_hfir_sub(const float * hp, const float * raw_in, float * out, int rows, int cols, const float * taps,
                                 int n_taps, float scale, float gamma, hipStream_t stream)
{
  const dim3 grid((cols + kHT_C - 1) / kHT_C, (rows + kHT_R - 1) / kHT_R);
  hipLaunchKernelGGL(photo_hfir_sub_kernel, grid, dim3(kT), 0, stream, hp, raw_in, out, rows, cols, taps, n_taps, scale, gamma);
  return hipGetLastError();
}
hipError_t launch_photo_scale(const float * in, float * out, int n, float scale, float gamma, hipStream_t stream)
{
  hipLaunchKernelGGL(photo_scale_kernel, g1(n), dim3(kT), 0, stream, in, out, n, scale, gamma);
  return hipGetLastError();
}
hipError_t launch_photo_brightness(const float * in, float * out, int rows, int cols, int win_w, int win_h, hipStream_t stream)
{
  if (win_w > kBMaxW || win_h > kBMaxH || win_w < 1 || win_h < 1) return hipErrorInvalidValue;
  const dim3 grid((cols + kBT_C - 1) / kBT_C, (rows + kBT_R - 1) / kBT_R);
  hipLaunchKernelGGL(photo_brightness_kernel, grid, dim3(kT), 0, stream, in, out, rows, cols, win_w, win_h);
  return hipGetLastError();
}
hipError_t launch_photo_gauss_trunc(const float * in, float * out, int rows, int cols, int do_gauss, hipStream_t stream)
{
  const dim3 grid((cols + kGT_C - 1) / kGT_C, (rows + kGT_R - 1) / kGT_R);
  hipLaunchKernelGGL(photo_gauss_trunc_kernel, grid, dim3(kT), 0, stream, in, out, rows, cols, do_gauss);
  return hipGetLastError();
}
hipError_t launch_photo_sobel_writeback(const float * img, float * dx, float * dy, const int32_t * idx, mh_point32 * desk,
                                        float * intensity_out, int rows, int cols, hipStream_t stream)
{
  hipLaunchKernelGGL(photo_sobel_writeback_kernel, g1(rows * cols), dim3(kT), 0, stream, img, dx, dy, idx, desk, intensity_out, rows, cols);
  return hipGetLastError();
}
hipError_t launch_photo_erode(const uint8_t * in, const uint8_t * static_mask, int margin, uint8_t * out, int rows, int cols,
                              int k, hipStream_t stream)
{
  if (k < 1 || k > kEMaxK) return hipErrorInvalidValue;
  const dim3 grid((cols + kET_C - 1) / kET_C, (rows + kET_R - 1) / kET_R);
  hipLaunchKernelGGL(photo_erode_kernel, grid, dim3(kT), 0, stream, in, static_mask, margin, out, rows, cols, k);
  return hipGetLastError();
}
// The preprocess chain behind the scatter in three multi-job launches (photo_stage_a/b/c_kernel).  false = a filter is
// larger than the tiles were sized for: the caller runs the single-stage kernels.
bool photo_stages_fit(const PhotoChain & c)
{
  if (c.n_hp > kPhotoMaxTaps || c.n_lp > kPhotoMaxTaps) return false;
  if (c.filter_brightness && (c.bw > kBMaxW || c.bh > kBMaxH || c.bw < 1 || c.bh < 1)) return false;
  if (c.erode_k < 1 || c.erode_k > kEMaxK) return false;
  return c.cols <= kMaxCols && c.rows >= 2 && c.cols >= 2;
}
hipError_t launch_photo_stages(const PhotoChain & c, const PhotoModel & m, hipStream_t stream)
{
  PhotoStageArgs a;
  a.raw = c.raw;
  a.ta = c.ta;
  a.tb = c.tb;
  a.fin = c.fin;
  a.dx = c.dx;
  a.dy = c.dy;
  a.idx = c.idx;
  a.intensity_out = c.intensity_out;
  a.hp = c.hp;
  a.lp = c.lp;
  a.static_mask = c.static_mask;
  a.mask_out = c.mask_out;
  a.yaw = c.yaw;
  a.desk = c.desk_points;
  a.proj = c.proj;
  a.counters = c.counters;
  a.rows = c.rows;
  a.cols = c.cols;
  a.n_pts = c.n_pts;
  a.n_hp = c.n_hp;
  a.n_lp = c.n_lp;
  a.remove_lines = c.remove_lines;
  a.filter_brightness = c.filter_brightness;
  a.bw = c.bw;
  a.bh = c.bh;
  a.do_gauss = c.do_gauss;
  a.erode_k = c.erode_k;
  a.scale = c.scale;
  a.gamma = c.gamma;
  a.stamps = c.stamps;
  a.seq = c.seq;
  a.raw_w = c.raw_w;
  a.range = c.range;
  a.idx_w = c.idx_w;
  a.copy_src = static_cast<const uint4 *>(c.copy_src);
  a.copy_dst = static_cast<uint4 *>(c.copy_dst);
  a.n_copy16 = static_cast<int>(c.copy_bytes / 16);
  a.desk_writeback = c.desk_writeback;
  const int npx = c.rows * c.cols;
  auto tiles = [&](int tr, int tc) { return ((c.cols + tc - 1) / tc) * ((c.rows + tr - 1) / tr); };
  // the launch ahead of the stages: scatter | projection-index reset (no frame-reset launch: see photo_scatter_stamp_kernel)
  {
    const int npb = (c.n_pts + kT - 1) / kT, nproj = (npx * kPhotoDup + 8 * kT - 1) / (8 * kT);
    const bool copy = c.desk_src && c.desk_src != c.desk_points;
    hipLaunchKernelGGL(photo_scatter_stamp_kernel, dim3(npb + nproj), dim3(kT), 0, stream, m, c.raw_points, copy ? c.desk_src : c.desk_points,
                       copy ? const_cast<mh_point32 *>(c.desk_points) : nullptr, c.n_pts, c.yaw, c.stamps, c.stamps + npx, c.seq, c.raw_w, c.range, c.idx_w,
                       c.intensity_out, c.proj, npb);
  }
  // A: vertical FIR (or scaling) | erosion | yaw rows | reset values of unclaimed pixels | the projection's front | pose-table copy
  a.n_job0 = c.remove_lines ? tiles(kVT_R, kVT_C) : (npx + kTS - 1) / kTS;
  a.n_job1 = tiles(kET_R, kET_C);
  a.n_job2 = m.rows;
  a.n_job3 = (npx + kTS - 1) / kTS;
  a.n_job4 = (c.n_pts + kTS - 1) / kTS;
  a.pre = static_cast<ProjPre *>(c.proj_pre);
  const int ncopy = a.n_copy16 ? std::min(4, (a.n_copy16 + kTS - 1) / kTS) : 0;
  hipLaunchKernelGGL(photo_stage_a_kernel, dim3(a.n_job0 + a.n_job1 + a.n_job2 + a.n_job3 + a.n_job4 + ncopy), dim3(kTS), 0, stream, a, m);
  // B: horizontal FIR + subtract | the projection's back
  a.n_job0 = c.remove_lines ? tiles(kHT_R, kHT_C) : 0;
  a.n_job1 = (c.n_pts + kTS - 1) / kTS;
  if (a.n_job0 + a.n_job1 > 0) hipLaunchKernelGGL(photo_stage_b_kernel, dim3(a.n_job0 + a.n_job1), dim3(kTS), 0, stream, a, m);
  // C: brightness + Gaussian + Sobel + write-back | projection-index finalise
  a.n_job0 = tiles(kFT_R, kFT_C);
  a.n_job1 = (npx + kTS - 1) / kTS;
  hipLaunchKernelGGL(photo_stage_c_kernel, dim3(a.n_job0 + a.n_job1), dim3(kTS), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_photo_grad(const float * dx, const float * dy, uint8_t * grad, int n, hipStream_t stream)
{
  hipLaunchKernelGGL(photo_grad_kernel, g1(n), dim3(kT), 0, stream, dx, dy, grad, n);
  return hipGetLastError();
}
hipError_t launch_photo_candidates(const uint8_t * grad, const uint8_t * mask, int npx, float thr, uint32_t * blk, uint32_t * out,
                                   uint32_t * n_out, hipStream_t stream)
{
  const dim3 g((npx + kT - 1) / kT);
  hipLaunchKernelGGL(photo_cand_count_kernel, g, dim3(kT), 0, stream, grad, mask, npx, thr, blk);
  hipLaunchKernelGGL(photo_cand_scatter_kernel, g, dim3(kT), 0, stream, grad, mask, npx, thr, blk, out, n_out);
  return hipGetLastError();
}
hipError_t launch_photo_gather(const int2 * uv, int n_off, int n_cand, bool per_candidate, const float * I, const int32_t * idx,
                               const mh_point32 * pts, int rows, int cols, float * win49, float4 * rec, int32_t * rec_idx, hipStream_t stream)
{
  if (n_cand <= 0) return hipSuccess;
  hipLaunchKernelGGL(photo_gather_kernel, dim3(n_cand), dim3(128), 0, stream, uv, n_off, n_cand, per_candidate ? 1 : 0, I, idx, pts, rows, cols,
                     win49, rec, rec_idx);
  return hipGetLastError();
}
hipError_t launch_photo_linearize(const PhotoLinArgs & a, hipStream_t stream)
{
  if (a.n_features <= 0) return hipSuccess;
  const int grid = (a.n_features + kFeatPerBlock - 1) / kFeatPerBlock;
  hipLaunchKernelGGL(photo_linearize_kernel, dim3(grid), dim3(64 * kFeatPerBlock), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mh
