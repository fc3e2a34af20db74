// C ABI of libmimosa_hip.so (include/mimosa_hip.h): contexts, the device-resident voxel map, the
// ICP factor handle and the deskew entry points.  Host-side epilogue of linearize()
// (geometric_factor.hpp:405-428, 459-561) lives here; the per-point work is in icp_kernels.hip.
//
// There is deliberately no CPU fallback: without a HIP device every entry point fails with
// MH_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/mimosa_hip.h"
#include "icp_device.hpp"
#include "math3.hpp"
#include "mh_internal.hpp"
#include "scan_device.hpp"
#include "shard_device.hpp"
#include "voxel_map.hpp"

namespace
{
void pose_inverse_compose(const double * Rs, const double * ts, const double * Rt, const double * tt, double * R, double * t)
{
  // delta = T_tgt^-1 * T_src (geometric_factor.hpp:251); unary: T_tgt = identity
  if (!Rt || !tt) {
    std::memcpy(R, Rs, sizeof(double) * 9);
    std::memcpy(t, ts, sizeof(double) * 3);
    return;
  }
  double Rinv[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rinv[3 * i + j] = Rt[3 * j + i];
  double tinv[3];
  for (int i = 0; i < 3; ++i) tinv[i] = -(Rinv[3 * i] * tt[0] + (Rinv[3 * i + 1] * tt[1] + Rinv[3 * i + 2] * tt[2]));
  mh::mat3_mul(Rinv, Rs, R);
  for (int i = 0; i < 3; ++i) t[i] = tinv[i] + (Rinv[3 * i] * ts[0] + (Rinv[3 * i + 1] * ts[1] + Rinv[3 * i + 2] * ts[2]));
}

// include/mimosa/lidar/utils.hpp:191-213
bool projection_matrix(const double loc[3], double thresh, const double E[9], double P[9])
{
  if (loc[0] > thresh && loc[1] > thresh && loc[2] > thresh) {
    for (int i = 0; i < 9; ++i) P[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return false;
  }
  for (int i = 0; i < 9; ++i) P[i] = 0.0;
  for (int i = 0; i < 3; ++i)
    if (loc[i] > thresh)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) P[3 * r + c] += E[3 * r + i] * E[3 * c + i];
  return true;
}

// Host epilogue of linearize(): unpack the device sums into the HessianFactor blocks, Schur
// degeneracy info (:413-428), 4-DoF projection (:464-475), degeneracy projection quirk (:477-557).
void finish_result(const mh_icp * icp, const mh::DeviceResult & d, const PendingCall & pc, mh_icp_result * out)
{
  std::memset(out, 0, sizeof(*out));
  const int NV = icp->binary ? 13 : 7;
  auto ent = [NV](int r, int c) {
    if (r > c) std::swap(r, c);
    return r * NV - r * (r - 1) / 2 + (c - r);
  };
  for (int r = 0; r < 6; ++r) {
    for (int c = 0; c < 6; ++c) out->H_ss[6 * r + c] = d.sums[ent(r, c)];
    out->b_s[r] = d.sums[ent(r, NV - 1)];
  }
  out->f = d.sums[ent(NV - 1, NV - 1)];
  if (icp->binary) {
    for (int r = 0; r < 6; ++r) {
      for (int c = 0; c < 6; ++c) {
        out->H_st[6 * r + c] = d.sums[ent(r, 6 + c)];
        out->H_tt[6 * r + c] = d.sums[ent(6 + r, 6 + c)];
      }
      out->b_t[r] = d.sums[ent(6 + r, 12)];
    }
  }
  // computeLocalizability of the rot / trans blocks of J_s^T J_s (:405-411), here on the host from the Hessian sums: on the
  // device it was two serial eigen-decompositions in K3's tail (2.3 us); K4 derives the bases it needs itself.
  {
    double Hr[9], Ht[9], er[9], et[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        Hr[3 * r + c] = d.sums[ent(r, c)];
        Ht[3 * r + c] = d.sums[ent(3 + r, 3 + c)];
      }
    mh::compute_localizability(Hr, out->loc_rot_final, er);
    mh::compute_localizability(Ht, out->loc_trans_final, et);
    // With the component pass on, the eigenvectors reported are the ones K4 projected on (it derives them on the device from
    // the same sums; with clustered eigenvalues the host's decomposition may return a differently rotated basis of the same
    // eigenspace, and loc_*_comp — thresholded at 0.5 per direction — belongs to K4's).  Eigenvalues do not depend on that.
    const bool from_k4 = pc.components && icp->n > 0 && pc.seq_has_basis;
    std::memcpy(out->eigvec_rot, from_k4 ? d.eig_rot : er, sizeof(er));
    std::memcpy(out->eigvec_trans, from_k4 ? d.eig_trans : et, sizeof(et));
  }
  for (int i = 0; i < 3; ++i) {
    // switched off (mh_icp_set_components): NaN, so that a caller who reads them anyway notices
    out->loc_trans_comp[i] = pc.components ? d.loc_comp[i] : std::numeric_limits<double>::quiet_NaN();
    out->loc_rot_comp[i] = pc.components ? d.loc_comp[3 + i] : std::numeric_limits<double>::quiet_NaN();
  }

  double Hrr[9], Hrt[9], Htr[9], Htt[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      Hrr[3 * r + c] = out->H_ss[6 * r + c];
      Hrt[3 * r + c] = out->H_ss[6 * r + 3 + c];
      Htr[3 * r + c] = out->H_ss[6 * (3 + r) + c];
      Htt[3 * r + c] = out->H_ss[6 * (3 + r) + 3 + c];
    }
  {
    double inv[9], tmp[9], tmp2[9], S[9], Sigma[9];
    mh::mat3_inv(Htt, inv);
    mh::mat3_mul(Hrt, inv, tmp);
    mh::mat3_mul(tmp, Htr, tmp2);
    for (int i = 0; i < 9; ++i) S[i] = Hrr[i] - tmp2[i];
    mh::mat3_inv(S, Sigma);
    mh::compute_localizability(Sigma, out->degen_rot, out->degen_eigvec_rot);
    for (int i = 0; i < 3; ++i) out->degen_rot[i] *= 57.29578;  // RAD2DEG (:428) is PCL's macro: (x)*57.29578
    mh::mat3_inv(Hrr, inv);
    mh::mat3_mul(Htr, inv, tmp);
    mh::mat3_mul(tmp, Hrt, tmp2);
    for (int i = 0; i < 9; ++i) S[i] = Htt[i] - tmp2[i];
    mh::mat3_inv(S, Sigma);
    mh::compute_localizability(Sigma, out->degen_trans, out->degen_eigvec_trans);
  }

  if (!icp->binary) {
    if (icp->cfg.reg_4_dof) {
      // local_z = R^T global_z; Pi = local_z local_z^T  (:257-259, :464-475)
      double lz[3];
      for (int i = 0; i < 3; ++i) lz[i] = pc.R[i] * pc.gz[0] + (pc.R[3 + i] * pc.gz[1] + pc.R[6 + i] * pc.gz[2]);
      double Pi[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Pi[3 * r + c] = lz[r] * lz[c];
      double a[9], b[9], c9[9], tmp[9];
      mh::mat3_mul(Pi, Hrr, tmp);
      mh::mat3_mul(tmp, Pi, a);
      mh::mat3_mul(Pi, Hrt, b);
      mh::mat3_mul(Htr, Pi, c9);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          out->H_ss[6 * r + c] = a[3 * r + c];
          out->H_ss[6 * r + 3 + c] = b[3 * r + c];
          out->H_ss[6 * (3 + r) + c] = c9[3 * r + c];
        }
      double br[3];
      for (int i = 0; i < 3; ++i) br[i] = Pi[3 * i] * out->b_s[0] + (Pi[3 * i + 1] * out->b_s[1] + Pi[3 * i + 2] * out->b_s[2]);
      for (int i = 0; i < 3; ++i) out->b_s[i] = br[i];
    }
    if (icp->cfg.project_on_degneneracy) {
      double P[9];
      const bool rot_degen = projection_matrix(out->loc_rot_final, icp->cfg.degen_thresh_rot, out->eigvec_rot, P);
      const bool trans_degen = projection_matrix(out->loc_trans_final, icp->cfg.degen_thresh_trans, out->eigvec_trans, P);
      if (rot_degen || trans_degen) {
        // Reference behaviour: H and b are rebuilt from two per-point arrays that
        // are allocated zero and never written (geometric_factor.hpp:270-271, 496-532), so the
        // rebuilt H and b are exactly zero and the localizabilities are recomputed from zero.
        std::memset(out->H_ss, 0, sizeof(out->H_ss));
        std::memset(out->b_s, 0, sizeof(out->b_s));
        const double Z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        mh::compute_localizability(Z, out->loc_rot_final, out->eigvec_rot);
        mh::compute_localizability(Z, out->loc_trans_final, out->eigvec_trans);
      }
    }
  }
  for (int i = 0; i < 9; ++i) out->status_hist[i] = pc.components ? static_cast<int32_t>(d.status_hist[i]) : -1;
  out->n_knn = static_cast<int64_t>(d.n_knn);
  out->n_exact_fallback = static_cast<int64_t>(d.n_fallback);
  out->mean_scanned = d.n_knn ? static_cast<double>(d.n_scanned) / static_cast<double>(d.n_knn) : 0.0;
  out->mean_candidates = d.n_knn ? static_cast<double>(d.n_cand) / static_cast<double>(d.n_knn) : 0.0;
  out->linearize_count = pc.linearize_count;
}
}  // namespace

// MH_WAIT_TRACE=1 (diagnostic): where the host side of a synchronous call spends its time — enqueue (argument blocks + the two
// launches), the wait for the first flagged word (kernels + dispatch + PCIe flight), the fold of K4's rows and the rest of the
// slot, the epilogue.  Averages are printed to stderr at mh_shutdown.
struct WaitTrace
{
  bool on = std::getenv("MH_WAIT_TRACE") != nullptr;
  double enq = 0, k3launch = 0, first = 0, fold = 0, fin = 0;
  long n = 0;
  double b_prep = 0, b_launch = 0, b_wait = 0;  // window batches: argument blocks, the launches, collecting every factor
  long b_n = 0;
  static double now()
  {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return static_cast<double>(t.tv_sec) * 1e9 + static_cast<double>(t.tv_nsec);
  }
};
static WaitTrace g_wt;

extern "C" {

int mh_abi_version(void) { return MH_ABI_VERSION; }

const char * mh_last_error(const mh_ctx * ctx) { return ctx ? ctx->err.c_str() : g_mh_err.c_str(); }

static int mh_init_impl(int device, mh_ctx ** out)
{
  if (!out) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_init: out is NULL");
  *out = nullptr;
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, MH_ERR_NO_DEVICE, "mh_init: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= count) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_init: device index out of range");
  {
    // the kernels are gfx950 code objects (wave64, DPP, s_memtime, agent-scope write-through hand-offs): refuse anything else
    hipDeviceProp_t prop;
    MH_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(nullptr, MH_ERR_UNSUPPORTED, std::string("mh_init: device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
  }
  MH_HIP(nullptr, hipSetDevice(device));
  mh_ctx * ctx = new (std::nothrow) mh_ctx;
  if (!ctx) return fail(nullptr, MH_ERR_OOM, "mh_init: host allocation failed");
  ctx->device = device;
  MH_HIP(nullptr, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  MH_HIP(nullptr, hipEventCreate(&ctx->timer[0]));
  MH_HIP(nullptr, hipEventCreate(&ctx->timer[1]));
  AllocCache::context_created(device);
  *out = ctx;
  return MH_OK;
}
int mh_init(int device, mh_ctx ** out)
{
  return guarded(nullptr, "mh_init", [&]() -> int { return mh_init_impl(device, out); });
}

void mh_shutdown(mh_ctx * ctx)
{
  if (!ctx) return;
  if (g_wt.on && g_wt.n) {
    const double n = static_cast<double>(g_wt.n);
    std::fprintf(stderr, "MH_WAIT_TRACE: %ld calls; per call: enqueue %.2f us (of which the K3 launch call %.2f), wait for the first word %.2f, "
                         "fold of the slot %.2f, epilogue %.2f\n", g_wt.n, g_wt.enq / n * 1e-3, g_wt.k3launch / n * 1e-3, g_wt.first / n * 1e-3,
                 (g_wt.fold - g_wt.first) / n * 1e-3, g_wt.fin / n * 1e-3);
    if (g_wt.b_n)
      std::fprintf(stderr, "MH_WAIT_TRACE: %ld window batches; per batch: argument blocks %.2f us, launches %.2f, collecting the factors %.2f\n", g_wt.b_n,
                   g_wt.b_prep / g_wt.b_n * 1e-3, g_wt.b_launch / g_wt.b_n * 1e-3, g_wt.b_wait / g_wt.b_n * 1e-3);
    g_wt = WaitTrace{};
  }
  (void)mh_enter(ctx);
  mhi::shard_ctx_gone(ctx);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  }
  if (ctx->timer[0]) (void)hipEventDestroy(ctx->timer[0]);
  if (ctx->timer[1]) (void)hipEventDestroy(ctx->timer[1]);
  if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
  if (ctx->d_batch) (void)hipFree(ctx->d_batch);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  const int dev = ctx->device;
  delete ctx;
  AllocCache::context_destroyed(dev);  // the device's last context: its cached blocks (and, with no context left anywhere, the pinned ones) go back
}

static int mh_set_profiling_impl(mh_ctx * ctx, int on)
{
  if (!ctx) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_set_profiling: ctx is NULL");
  ctx->profiling = on < 0 ? 0 : on;
  return MH_OK;
}
int mh_set_profiling(mh_ctx * ctx, int on)
{
  return guarded(ctx, "mh_set_profiling", [&]() -> int { return mh_set_profiling_impl(ctx, on); });
}

void * mh_stream(mh_ctx * ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

static int mh_synchronize_impl(mh_ctx * ctx)
{
  if (!ctx) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_synchronize: ctx is NULL");
  MH_HIP(ctx, mh_enter(ctx));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int mh_synchronize(mh_ctx * ctx)
{
  return guarded(ctx, "mh_synchronize", [&]() -> int { return mh_synchronize_impl(ctx); });
}

static int mh_timer_begin_impl(mh_ctx * ctx)
{
  if (!ctx) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_timer_begin: ctx is NULL");
  MH_HIP(ctx, mh_enter(ctx));
  MH_HIP(ctx, hipEventRecord(ctx->timer[0], ctx->stream));
  return MH_OK;
}
int mh_timer_begin(mh_ctx * ctx)
{
  return guarded(ctx, "mh_timer_begin", [&]() -> int { return mh_timer_begin_impl(ctx); });
}

static int mh_timer_end_impl(mh_ctx * ctx, float * ms)
{
  if (!ctx || !ms) return fail(ctx, MH_ERR_INVALID_ARG, "mh_timer_end: NULL argument");
  MH_HIP(ctx, mh_enter(ctx));
  MH_HIP(ctx, hipEventRecord(ctx->timer[1], ctx->stream));
  MH_HIP(ctx, hipEventSynchronize(ctx->timer[1]));
  MH_HIP(ctx, hipEventElapsedTime(ms, ctx->timer[0], ctx->timer[1]));
  return MH_OK;
}
int mh_timer_end(mh_ctx * ctx, float * ms)
{
  return guarded(ctx, "mh_timer_end", [&]() -> int { return mh_timer_end_impl(ctx, ms); });
}

}  // extern "C"

extern "C" {
// ---- factor ------------------------------------------------------------------------------------
static int icp_alloc(mh_icp * icp)
{
  mh_ctx * ctx = icp->ctx;
  const size_t nn = icp->cap_n > icp->n ? icp->cap_n : icp->n;  // map-sharded factors reserve room for arrivals
  const size_t n = nn ? nn : 1;
  MH_HIP(ctx, icp->d_src.reserve(n * sizeof(float4), ctx->stream, false));
  MH_HIP(ctx, icp->d_qda.reserve(n * 3 * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, icp->d_mean.reserve(n * 3 * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, icp->d_normal.reserve(n * 3 * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, icp->d_status.reserve(n * sizeof(int32_t), ctx->stream, false));
  const size_t max_grid = static_cast<size_t>(mh::linearize_grid_max(static_cast<int>(n)));
  MH_HIP(ctx, icp->d_partials.reserve(max_grid * mh::kPartialStride * sizeof(double), ctx->stream, false));
  MH_HIP(ctx, icp->d_ticket.reserve(4 * sizeof(unsigned int), ctx->stream, false));  // K3's ticket, K4's ticket, the point count of the two-phase forms
  MH_HIP(ctx, icp->d_result.reserve(sizeof(mh::DeviceResult), ctx->stream, false));
#ifdef MH_TIMELINE
  MH_HIP(ctx, icp->d_dbg.reserve(2 * max_grid * 8 * 16 * sizeof(unsigned long long), ctx->stream, false));  // K3's waves, then K4's
  MH_HIP(ctx, hipMemsetAsync(icp->d_dbg.p, 0, icp->d_dbg.cap, ctx->stream));
#endif
  MH_HIP(ctx, AllocCache::alloc_pinned(reinterpret_cast<void **>(&icp->h_results), sizeof(mh::DeviceResult) * kMaxPending));
  MH_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&icp->d_h_results), icp->h_results, 0));
  // flagged-word slots: a recycled pinned block may hold another factor's words — harmless, sequence numbers are drawn from
  // one process-wide counter (next_call_seq), so nothing stale ever carries the number of a call of this factor
  // (row capacity in steps of 16 workgroups: factors of similar size — a scan's down-sampled cloud from one keyframe to the
  // next — ask the pinned cache for the SAME size and get a recycled block instead of a fresh hipHostMalloc)
  {
    // rows of K4's workgroups: the factor's own launch class, or the class of a window batch it may be linearized in
    const int ni = static_cast<int>(n);
    int rows = 0;
    for (int ppw : {64, 128, 256, 512}) rows = std::max(rows, mh::class_loc_grid(ni, ppw));
    icp->ll_words = mh::ll_slot_words((rows + 15) & ~15);
  }
  MH_HIP(ctx, AllocCache::alloc_pinned(reinterpret_cast<void **>(&icp->h_ll), icp->ll_words * sizeof(uint4) * kMaxPending));
  MH_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&icp->d_h_ll), icp->h_ll, 0));
  return MH_OK;
}

// Source cloud (packed, Morton-ordered) of a freshly allocated factor.  Everything is enqueued on the context stream;
// the temporaries live in the context's stream-ordered scratch (no allocation, no free, no synchronisation per factor),
// the only wait is for a HOST source buffer, which the caller owns.
static int icp_init_source(mh_icp * icp, const mh_point32 * source, const mh_point32 * d_source)
{
  mh_ctx * ctx = icp->ctx;
  const size_t n = icp->n;
  auto * zero_a = static_cast<uint32_t *>(icp->d_ticket.p);
  auto * zero_b = static_cast<uint32_t *>(icp->d_result.p);
  if (n) {
    const char * ns = std::getenv("MH_NO_SORT");  // MH_NO_SORT=1 keeps input order (diagnostics)
    const bool order = !(ns && ns[0] == '1') && !icp->no_order;
    const int ni = static_cast<int>(n);
    const size_t up = d_source ? 0 : ((n * sizeof(mh_point32) + 255) & ~size_t(255));
    const size_t need = up + (order ? mh::source_order_scratch_bytes(ni) : 0) + 256;
    if (need > ctx->d_scratch_cap) {
      MH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // earlier users of the old block
      if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
      ctx->d_scratch = nullptr;
      ctx->d_scratch_cap = 0;
      MH_HIP(ctx, hipMalloc(&ctx->d_scratch, need + need / 4));
      ctx->d_scratch_cap = need + need / 4;
    }
    char * sc = static_cast<char *>(ctx->d_scratch);
    const mh_point32 * d_pts = d_source;
    if (!d_source) {
      MH_HIP(ctx, hipMemcpyAsync(sc, source, n * sizeof(mh_point32), hipMemcpyHostToDevice, ctx->stream));
      d_pts = reinterpret_cast<const mh_point32 *>(sc);
    }
    if (order) {  // spatial (Morton) ordering of the copy: see order_kernels.hip
      MH_HIP(ctx, icp->d_perm.reserve(n * sizeof(uint32_t), ctx->stream, false));
      float cell = 0.25f;
      if (const char * cs = std::getenv("MH_SORT_CELL")) cell = static_cast<float>(std::atof(cs));
      MH_HIP(ctx, mh::launch_source_order(d_pts, ni, cell, sc + up, static_cast<uint32_t *>(icp->d_perm.p), static_cast<float4 *>(icp->d_src.p),
                                          zero_a, 2, zero_b, static_cast<int>(sizeof(mh::DeviceResult) / 4), ctx->stream));
      icp->ordered = true;
    } else {
      MH_HIP(ctx, mh::launch_pack_xyz(d_pts, ni, static_cast<float4 *>(icp->d_src.p), ctx->stream));
      MH_HIP(ctx, hipMemsetAsync(icp->d_ticket.p, 0, 2 * sizeof(unsigned int), ctx->stream));
      MH_HIP(ctx, hipMemsetAsync(icp->d_result.p, 0, sizeof(mh::DeviceResult), ctx->stream));
    }
  } else {
    MH_HIP(ctx, hipMemsetAsync(icp->d_ticket.p, 0, 2 * sizeof(unsigned int), ctx->stream));
    MH_HIP(ctx, hipMemsetAsync(icp->d_result.p, 0, sizeof(mh::DeviceResult), ctx->stream));
  }
  // commonConstructor(): all per-point state zero (geometric_factor.hpp:144-156).  A cold factor's state is never read
  // (the first linearize treats it as zero without touching memory, the getters answer zeros while `cold`); only a
  // factor that receives foreign records before its first linearize (the sharded path, no_order) needs real zeros.
  if (icp->no_order) {
    MH_HIP(ctx, hipMemsetAsync(icp->d_qda.p, 0, icp->d_qda.cap, ctx->stream));
    MH_HIP(ctx, hipMemsetAsync(icp->d_mean.p, 0, icp->d_mean.cap, ctx->stream));
    MH_HIP(ctx, hipMemsetAsync(icp->d_normal.p, 0, icp->d_normal.cap, ctx->stream));
    MH_HIP(ctx, hipMemsetAsync(icp->d_status.p, 0, icp->d_status.cap, ctx->stream));
  }
  if (!d_source) MH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's host buffer may go away now
  return MH_OK;
}

// source: host cloud (d_source == nullptr) or a cloud already on the device (source == nullptr)
static int icp_create_common(mh_ctx * ctx, mh_map * map, const mh_point32 * source, const mh_point32 * d_source, size_t n,
                             const mh_reg_config * cfg, int is_binary, mh_icp ** out, bool no_order = false, size_t capacity = 0)
{
  *out = nullptr;
  // A map may be shared read-only by factors of several contexts (= HIP streams) of the SAME device:
  // uploads are host-synchronised on the map's own stream before any factor kernel is enqueued.
  if (map->ctx->device != ctx->device) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_create: map lives on another device");
  if (map->poisoned) return fail(ctx, MH_ERR_HIP, "mh_icp_create: the map is inconsistent after a failed mutation");
  if (cfg->num_corres_points < 2 || cfg->num_corres_points > 8)
    return fail(ctx, MH_ERR_UNSUPPORTED, "mh_icp_create: num_corres_points must be in 2..8");
  if (n > 0x3fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_icp_create: cloud too large");
  MH_HIP(ctx, mh_enter(ctx));
  mh_icp * icp = new (std::nothrow) mh_icp;
  if (!icp) return fail(ctx, MH_ERR_OOM, "mh_icp_create: host allocation failed");
  icp->ctx = ctx;
  icp->map = map;
  mh_map_retain(map);
  map_add_reader(map, ctx);
  icp->n = n;
  icp->cfg = *cfg;
  icp->binary = is_binary != 0;
  icp->no_order = no_order;
  icp->cap_n = capacity > n ? capacity : n;
  int rc = icp_alloc(icp);
  if (rc != MH_OK) {
    mh_icp_destroy(icp);
    return rc;
  }
  rc = icp_init_source(icp, source, d_source);
  if (rc != MH_OK) {
    mh_icp_destroy(icp);
    return rc;
  }
  icp->cold = true;
  *out = icp;
  return MH_OK;
}

static int mh_icp_create_impl(mh_ctx * ctx, mh_map * map, const mh_point32 * source, size_t n, const mh_reg_config * cfg,
                  int is_binary, mh_icp ** out)
{
  if (!ctx || !map || !cfg || !out || (!source && n)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_create: NULL argument");
  return icp_create_common(ctx, map, source, nullptr, n, cfg, is_binary, out);
}
int mh_icp_create(mh_ctx * ctx, mh_map * map, const mh_point32 * source, size_t n, const mh_reg_config * cfg,
                  int is_binary, mh_icp ** out)
{
  return guarded(ctx, "mh_icp_create", [&]() -> int { return mh_icp_create_impl(ctx, map, source, n, cfg, is_binary, out); });
}

static int mh_icp_clone_impl(const mh_icp * src, mh_icp ** out)
{
  if (!src || !out) return fail(src ? src->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_icp_clone: NULL argument");
  *out = nullptr;
  mh_ctx * ctx = src->ctx;
  if (src->n_pending) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_clone: source has linearize calls in flight");
  MH_HIP(ctx, mh_enter(ctx));
  mh_icp * icp = new (std::nothrow) mh_icp;
  if (!icp) return fail(ctx, MH_ERR_OOM, "mh_icp_clone: host allocation failed");
  icp->ctx = ctx;
  icp->map = src->map;
  mh_map_retain(icp->map);
  map_add_reader(icp->map, ctx);
  icp->n = src->n;
  icp->cfg = src->cfg;
  icp->binary = src->binary;
  int rc = icp_alloc(icp);
  if (rc != MH_OK) {
    mh_icp_destroy(icp);
    return rc;
  }
  const size_t n = src->n;
  auto cp = [&](const DevBuf & a, DevBuf & b, size_t bytes) {
    return bytes ? hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, ctx->stream) : hipSuccess;
  };
  MH_HIP(ctx, cp(src->d_src, icp->d_src, n * sizeof(float4)));
  MH_HIP(ctx, cp(src->d_qda, icp->d_qda, n * 3 * sizeof(double)));
  MH_HIP(ctx, cp(src->d_mean, icp->d_mean, n * 3 * sizeof(double)));
  MH_HIP(ctx, cp(src->d_normal, icp->d_normal, n * 3 * sizeof(double)));
  MH_HIP(ctx, cp(src->d_status, icp->d_status, n * sizeof(int32_t)));
  MH_HIP(ctx, cp(src->d_result, icp->d_result, sizeof(mh::DeviceResult)));
  if (src->ordered) {
    MH_HIP(ctx, icp->d_perm.reserve(n * sizeof(uint32_t), ctx->stream, false));
    MH_HIP(ctx, cp(src->d_perm, icp->d_perm, n * sizeof(uint32_t)));
    icp->ordered = true;
  }
  MH_HIP(ctx, hipMemsetAsync(icp->d_ticket.p, 0, 2 * sizeof(unsigned int), ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  icp->cold = src->cold;
  icp->components = src->components;
  icp->linearize_count = src->linearize_count;
  *out = icp;
  return MH_OK;
}
int mh_icp_clone(const mh_icp * src, mh_icp ** out)
{
  return guarded(src ? src->ctx : nullptr, "mh_icp_clone", [&]() -> int { return mh_icp_clone_impl(src, out); });
}

void mh_icp_destroy(mh_icp * icp)
{
  if (!icp) return;
  (void)mh_enter(icp->ctx);
  (void)hipStreamSynchronize(icp->ctx->stream);
  icp->d_rec.release(true);
  icp->d_src.release(true);
  icp->d_qda.release(true);
  icp->d_mean.release(true);
  icp->d_normal.release(true);
  icp->d_status.release(true);
  icp->d_partials.release(true);
  icp->d_ticket.release(true);
  icp->d_result.release(true);
  icp->d_dbg.release(true);
  icp->d_perm.release(true);
  icp->d_eig.release(true);
  for (DevBuf * b : {&icp->d_origin, &icp->x_src, &icp->x_qda, &icp->x_mean, &icp->x_normal, &icp->x_status, &icp->x_origin, &icp->s_keys_a,
                     &icp->s_keys_b, &icp->s_idx_a, &icp->s_idx_b, &icp->s_counts, &icp->s_temp, &icp->d_sums})
    b->release(true);
  if (icp->h_counts) (void)hipHostFree(icp->h_counts);
  if (icp->h_results) AllocCache::free_pinned(icp->h_results, sizeof(mh::DeviceResult) * kMaxPending);
  if (icp->h_ll) AllocCache::free_pinned(icp->h_ll, icp->ll_words * sizeof(uint4) * kMaxPending);
  if (icp->events_ready)
    for (auto & ev : icp->events)
      for (auto & e : ev) (void)hipEventDestroy(e);
  if (icp->map) {
    map_remove_reader(icp->map, icp->ctx);
    mh_map_release(icp->map);
  }
  delete icp;
}

size_t mh_icp_size(const mh_icp * icp) { return icp ? icp->n : 0; }

#ifdef MH_TIMELINE
// Diagnostic build only: per-wave s_memtime stamps of the last linearize (8 per wave).
int mh_icp_timeline(mh_icp * icp, unsigned long long * out, size_t capacity_words, size_t * n_words)
{
  if (!icp || !out || !n_words) return MH_ERR_INVALID_ARG;
  mh_ctx * ctx = icp->ctx;
  const size_t words = 2 * static_cast<size_t>(mh::linearize_grid_max(static_cast<int>(icp->n))) * 8 * 16;  // K3's half, then K4's
  *n_words = words;
  if (capacity_words < words) return MH_ERR_INVALID_ARG;
  MH_HIP(ctx, mh_enter(ctx));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MH_HIP(ctx, hipMemcpy(out, icp->d_dbg.p, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return MH_OK;
}
#endif

static int mh_icp_reset_impl(mh_icp * icp)
{
  if (!icp) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_icp_reset: icp is NULL");
  icp->cold = true;  // the next linearize treats the cached state as all-zero (no memset needed)
  return MH_OK;
}
int mh_icp_reset(mh_icp * icp)
{
  return guarded(icp ? icp->ctx : nullptr, "mh_icp_reset", [&]() -> int { return mh_icp_reset_impl(icp); });
}

static int mh_icp_set_components_impl(mh_icp * icp, int enabled)
{
  if (!icp) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_icp_set_components: icp is NULL");
  if (icp->n_pending) return fail(icp->ctx, MH_ERR_INVALID_ARG, "mh_icp_set_components: linearize calls in flight");
  icp->components = enabled != 0;
  return MH_OK;
}
int mh_icp_set_components(mh_icp * icp, int enabled)
{
  return guarded(icp ? icp->ctx : nullptr, "mh_icp_set_components", [&]() -> int { return mh_icp_set_components_impl(icp, enabled); });
}

// What linearize_prepare changes on the handle before anything is enqueued: put back unless the launches went through (a
// recoverable failure — an allocation, a launch error — must not leave a never-linearized factor looking warm, or a pending
// slot claimed for ever).
struct LinearizeTxn
{
  mh_icp * icp;
  bool cold = true, committed = false;
  int count = 0, pending = 0;
  explicit LinearizeTxn(mh_icp * i) : icp(i)
  {
    if (icp) {
      cold = icp->cold;
      count = icp->linearize_count;
      pending = icp->n_pending;
    }
  }
  LinearizeTxn(LinearizeTxn && o) noexcept : icp(o.icp), cold(o.cold), committed(o.committed), count(o.count), pending(o.pending) { o.icp = nullptr; }
  LinearizeTxn(const LinearizeTxn &) = delete;
  LinearizeTxn & operator=(const LinearizeTxn &) = delete;
  void commit() { committed = true; }
  ~LinearizeTxn()
  {
    if (icp && !committed) {
      icp->cold = cold;
      icp->linearize_count = count;
      icp->n_pending = pending;
    }
  }
};

// Sequence numbers of calls: process-wide, never 0 (what a fresh or recycled flagged-word slot cannot hold by accident).
static unsigned int next_call_seq()
{
  static std::atomic<unsigned int> counter{0};
  unsigned int v = counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
  if (v == 0u) v = counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
  return v;
}

// Argument blocks of one linearize call of `icp` in pending slot n_pending (which it claims): everything of
// linearize_enqueue except the launches.  want_flag: the last kernel publishes a completion sequence number to the
// host slot (mh_icp_wait then spins on it instead of synchronising the stream).  Worth it for one synchronous call,
// not for a pipelined batch: the system-scope fence it needs lengthens every K4 by ~2 us.
static int linearize_prepare(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                             const double * t_tgt, const double g_unit[3], mh_icp_result * out, bool want_flag, bool allow_timing,
                             mh::IcpArgs & a, mh::LocArgs & l, bool & timed)
{
  if (!icp || !R_src || !t_src || !g_unit || !out)
    return fail(icp ? icp->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_icp_linearize: NULL argument");
  mh_ctx * ctx = icp->ctx;
  if (icp->binary && (!R_tgt || !t_tgt)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_linearize: binary factor needs the target pose");
  if (icp->n_pending >= kMaxPending) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_linearize_async: too many calls in flight");
  MH_HIP(ctx, mh_enter(ctx));
  if (ctx->profiling && !icp->events_ready) {
    for (auto & ev : icp->events)
      for (auto & e : ev) MH_HIP(ctx, hipEventCreate(&e));
    icp->events_ready = true;
  }

  const size_t row_doubles = static_cast<size_t>(mh::linearize_grid_max(static_cast<int>(icp->n ? icp->n : 1))) * mh::kPartialStride;
  MH_HIP(ctx, icp->d_partials.reserve(row_doubles * sizeof(double), ctx->stream, false));
  // the record K3 writes for K4 (mh_internal.hpp): K4 follows on the same stream, one record serves every call
  a.rec = nullptr;
  a.rec_n = 0;
  if (icp->components && icp->n > 0) {
    MH_HIP(ctx, icp->d_rec.reserve(mh::loc_record_bytes(icp->n), ctx->stream, false));
    a.rec = static_cast<double *>(icp->d_rec.p);
    a.rec_n = static_cast<int>(icp->n);
  }
  a.map = map_view(icp->map);
  a.src = static_cast<const float4 *>(icp->d_src.p);
  a.n = static_cast<int>(icp->n);
  a.k = static_cast<int>(icp->cfg.num_corres_points);
  a.cold = icp->cold ? 1 : 0;
  a.use_huber = icp->cfg.use_huber;
  pose_inverse_compose(R_src, t_src, icp->binary ? R_tgt : nullptr, icp->binary ? t_tgt : nullptr, a.R, a.t);
  // config floats are promoted to double in the reference's expressions (geometric_config.hpp:17-33)
  a.da_thresh = static_cast<double>(icp->cfg.target_ivox_map_min_dist_in_voxel / 4);
  a.max_d2 = static_cast<double>(icp->cfg.max_corres_distance * icp->cfg.max_corres_distance);
  a.plane_valid = static_cast<double>(icp->cfg.plane_validity_distance);
  a.sigma = static_cast<double>(icp->cfg.lidar_point_noise_std_dev);
  a.inv_sigma = 1.0 / a.sigma;
  a.huber = static_cast<double>(icp->cfg.huber_threshold);
  a.inv_huber = 1.0 / a.huber;
  a.q_da = static_cast<double *>(icp->d_qda.p);
  a.mean = static_cast<double *>(icp->d_mean.p);
  a.normal = static_cast<double *>(icp->d_normal.p);
  a.status = static_cast<int32_t *>(icp->d_status.p);
  a.partials = static_cast<double *>(icp->d_partials.p);
  a.ticket = static_cast<unsigned int *>(icp->d_ticket.p);
  a.result = static_cast<mh::DeviceResult *>(icp->d_result.p);
  a.host_result = nullptr;  // set below once the slot is known
  a.dbg = static_cast<unsigned long long *>(icp->d_dbg.p);
  a.reps = 1;
#ifdef MH_TIMELINE
  if (const char * rp = std::getenv("MH_REPS")) a.reps = std::atoi(rp);
#endif

  l.src = a.src;
  l.host_result = nullptr;  // set below once the slot is known
  l.seq = 0;
  l.eig = nullptr;
  l.nv = icp->binary ? 13 : 7;
  l.n = a.n;
  l.k = a.k;
  l.chunks_per_block = 1;
  std::memcpy(l.R, a.R, sizeof(l.R));
  l.normal = a.normal;
  l.status = a.status;
  l.partials = a.partials;
  l.ticket = a.ticket + 1;
  l.result = a.result;
  l.rec = a.rec;
  l.rec_n = a.rec_n;
#ifdef MH_TIMELINE
  l.dbg = a.dbg ? a.dbg + static_cast<size_t>(mh::linearize_grid_max(static_cast<int>(icp->n ? icp->n : 1))) * 8 * 16 : nullptr;
#endif

  const int slot = icp->n_pending;
  PendingCall & pc = icp->pending[slot];
  pc.out = out;
  std::memcpy(pc.R, a.R, sizeof(pc.R));
  for (int i = 0; i < 3; ++i) pc.gz[i] = -g_unit[i];
  pc.linearize_count = ++icp->linearize_count;
  // each event record is a barrier + signal packet on the stream (~4 us): sampled calls only
  timed = allow_timing && ctx->profiling > 0 && (pc.linearize_count % ctx->profiling) == 0;
  if (timed) {
    for (int i = 0; i < 3; ++i) pc.ev[i] = icp->events[slot][i];
  } else {
    pc.ev[0] = pc.ev[1] = pc.ev[2] = nullptr;
  }
  pc.seq = 0;
  pc.components = icp->components;
  pc.seq_has_basis = false;  // set by the callers whose K4 publishes the eigenbases it projected on
  pc.loc_blocks = 0;
  pc.launched_k4 = false;
  a.seq = 0;
  a.tail = 1;
  a.ll = l.ll = nullptr;
  const int ppw = mh::linearize_class(a.n, a.k, false);  // (a call of its own; a window batch and the sharded path set their own)
  l.k3_blocks = mh::class_grid(a.n, ppw);
  (void)want_flag;  // every call is collected through its flagged words now: no completion flag to ask for
  if (a.n > 0) {
    // plain factors publish flagged words into the call's slot; the two-phase and sharded callers overwrite what they need
    pc.seq = a.seq = l.seq = next_call_seq();
    a.ll = l.ll = icp->d_h_ll + static_cast<size_t>(slot) * icp->ll_words;
    pc.loc_blocks = mh::class_loc_grid(a.n, ppw);
  }
  icp->n_pending++;
  icp->cold = false;
  return MH_OK;
}

static int linearize_enqueue(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                             const double * t_tgt, const double g_unit[3], mh_icp_result * out, bool want_flag)
{
  mh::IcpArgs a;
  mh::LocArgs l;
  bool timed = false;
  LinearizeTxn txn(icp);
  const int rc = linearize_prepare(icp, R_src, t_src, R_tgt, t_tgt, g_unit, out, want_flag, true, a, l, timed);
  if (rc != MH_OK) return rc;
  mh_ctx * ctx = icp->ctx;
  const int slot = icp->n_pending - 1;
  PendingCall & pc = icp->pending[slot];
  if (timed) MH_HIP(ctx, hipEventRecord(pc.ev[0], ctx->stream));
  if (a.n > 0) {
    a.tail = pc.components ? 0 : 1;  // K4 follows and folds K3's rows itself / K3 is the whole call: its last block folds and publishes
    // K3, then K4 on the context's stream, for synchronous and pipelined callers alike (mh_icp_linearize_async only skips the
    // wait).  Rounds 4-5 ran the K4 work of pipelined calls on a long-running "component server" kernel of a side stream: it
    // bought 6 us per step for a caller with <= 64 calls of ONE factor in flight — a pattern no caller of the reference has
    // (GTSAM linearizes one factor at a time, the smoother window goes through mh_icp_linearize_batch) — and cost a ring,
    // sign-off words, write-through records, a restart rule and a hazard with device-wide synchronisations: removed in round 6.
    const double tl0 = g_wt.on ? WaitTrace::now() : 0.0;
    MH_HIP(ctx, mh::launch_linearize(a, icp->binary, ctx->stream));
    if (g_wt.on) g_wt.k3launch += WaitTrace::now() - tl0;
    if (timed) MH_HIP(ctx, hipEventRecord(pc.ev[1], ctx->stream));
    if (pc.components) {
      MH_HIP(ctx, mh::launch_localizability(l, ctx->stream));
      pc.seq_has_basis = true;
      pc.launched_k4 = true;
    }
    if (timed) MH_HIP(ctx, hipEventRecord(pc.ev[2], ctx->stream));
  } else {
    if (timed) {
      MH_HIP(ctx, hipEventRecord(pc.ev[1], ctx->stream));
      MH_HIP(ctx, hipEventRecord(pc.ev[2], ctx->stream));
    }
    // nothing was launched (pc.seq == 0): the result of an empty cloud is all zero, assembled on the host
  }
  txn.commit();
  return MH_OK;
}

static int mh_icp_linearize_async_impl(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                           const double * t_tgt, const double g_unit[3], mh_icp_result * out)
{
  return linearize_enqueue(icp, R_src, t_src, R_tgt, t_tgt, g_unit, out, false);
}
int mh_icp_linearize_async(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                           const double * t_tgt, const double g_unit[3], mh_icp_result * out)
{
  return guarded(icp ? icp->ctx : nullptr, "mh_icp_linearize_async", [&]() -> int { return mh_icp_linearize_async_impl(icp, R_src, t_src, R_tgt, t_tgt, g_unit, out); });
}

// One flagged word (icp_device.hpp): two self-validating 8-byte halves {lo | seq << 32, hi | seq << 32}.
static inline bool ll_read_bits(const uint4 * p, unsigned int seq, unsigned long long & bits)
{
  const auto * q = reinterpret_cast<const unsigned long long *>(p);
  const unsigned long long a = __atomic_load_n(q, __ATOMIC_ACQUIRE), b = __atomic_load_n(q + 1, __ATOMIC_ACQUIRE);
  if (static_cast<unsigned int>(a >> 32) != seq || static_cast<unsigned int>(b >> 32) != seq) return false;
  bits = (a & 0xffffffffull) | (b << 32);
  return true;
}
static inline bool ll_read(const uint4 * p, unsigned int seq, double & v)
{
  unsigned long long bits;
  if (!ll_read_bits(p, seq, bits)) return false;
  std::memcpy(&v, &bits, sizeof(v));
  return true;
}

// Assemble a call's DeviceResult from its flagged words: sums + counters (from K4's workgroup 0, or from K3's last block when
// no K4 ran), the eigenbases K4 projected on, and K4's per-workgroup rows folded HERE in workgroup order (deterministic).
// spin_ns > 0: wait up to that long for words that have not arrived.  false = something is still missing.
static bool collect_call(const mh_icp * icp, int slot, const PendingCall & pc, long spin_ns, mh::DeviceResult & d)
{
  std::memset(&d, 0, sizeof(d));
  if (pc.seq == 0) return true;  // nothing was launched (empty cloud)
  const uint4 * base = icp->h_ll + static_cast<size_t>(slot) * icp->ll_words;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  unsigned spins = 0;
  auto get = [&](const uint4 * p, double & v) {
    while (!ll_read(p, pc.seq, v)) {
      if (spin_ns <= 0) return false;
      __builtin_ia32_pause();
      if ((++spins & 1023u) == 0u) {
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > spin_ns) return false;
      }
    }
    return true;
  };
  const int nent = icp->binary ? 91 : 28;
  double c4[4];
  // the words that are written LAST first (K4's rows, when it ran): once they are here the rest usually is
  if (pc.launched_k4) {
    double acc[6] = {0};
    unsigned long long hist[10] = {0};
    const uint4 * rows = base + (mh::kLlSums + mh::kLlEig);
    {
      // wait for the first word; K4's workgroups end within a microsecond of each other, so by then (nearly) everything has
      // landed: touch every cache line of the slot at once — the device wrote them over PCIe, each is a miss, and a dozen
      // misses in flight cost what one does
      double v0;
      const double tw0 = g_wt.on ? WaitTrace::now() : 0.0;
      if (!get(rows, v0)) return false;
      if (g_wt.on) g_wt.first += WaitTrace::now() - tw0;
      const char * lo = reinterpret_cast<const char *>(base);
      const char * hi = reinterpret_cast<const char *>(rows + static_cast<size_t>(pc.loc_blocks) * mh::kLlRow);
      for (const char * q = lo; q < hi; q += 64) __builtin_prefetch(q);
    }
    for (int b = 0; b < pc.loc_blocks; ++b) {
      const uint4 * row = rows + static_cast<size_t>(b) * mh::kLlRow;
      for (int i = 0; i < 6; ++i) {
        double v;
        if (!get(row + i, v)) return false;
        acc[i] += v;
      }
      for (int i = 0; i < 2; ++i) {  // counts 0..4 / 5..8, 12 bits each (icp_device.hpp: kLlRow)
        double raw;
        if (!get(row + 6 + i, raw)) return false;
        unsigned long long bits;
        std::memcpy(&bits, &raw, sizeof(bits));
        for (int h = 0; h < (i == 0 ? 5 : 4); ++h) hist[5 * i + h] += (bits >> (12 * h)) & 0xfffull;
      }
    }
    for (int i = 0; i < 6; ++i) d.loc_comp[i] = acc[i];
    for (int i = 0; i < 9; ++i) d.status_hist[i] = static_cast<unsigned int>(hist[i]);
    for (int i = 0; i < 18; ++i) {
      double v;
      if (!get(base + mh::kLlSums + i, v)) return false;
      (i < 9 ? d.eig_rot[i] : d.eig_trans[i - 9]) = v;
    }
  }
  for (int i = 0; i < nent; ++i)
    if (!get(base + i, d.sums[i])) return false;
  for (int i = 0; i < 4; ++i)
    if (!get(base + nent + i, c4[i])) return false;
  d.n_knn = static_cast<unsigned long long>(c4[0]);
  d.n_cand = static_cast<unsigned long long>(c4[1]);
  d.n_fallback = static_cast<unsigned long long>(c4[2]);
  d.n_scanned = static_cast<unsigned long long>(c4[3]);
  return true;
}

static int mh_icp_wait_impl(mh_icp * icp)
{
  const double tw_enter = g_wt.on ? WaitTrace::now() : 0.0;
  if (!icp) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_icp_wait: icp is NULL");
  mh_ctx * ctx = icp->ctx;
  MH_HIP(ctx, mh_enter(ctx));
  // Every value a call produces arrives in mapped pinned memory tagged with the call's sequence number: the host polls the
  // values themselves instead of paying the runtime's stream-synchronisation latency (or a device-side completion flag
  // behind a system-scope fence).  Anything that does not show up within the spin budget falls back to the stream.  A call
  // bracketed by HIP events is collected the same way; its last event — recorded behind K4, whose words have arrived — is
  // waited for alone (hipEventSynchronize on an event that has completed or is about to), not the whole stream.
  bool synced = false;
  for (int s = 0; s < icp->n_pending; ++s) {
    const PendingCall & pc = icp->pending[s];
    mh::DeviceResult d;
    if (!collect_call(icp, s, pc, synced ? 0L : 20000000L, d)) {  // 20 ms
      if (!synced) {
        MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        synced = true;
      }
      if (!collect_call(icp, s, pc, 2000000L, d)) {
        icp->n_pending = 0;
        return fail(ctx, MH_ERR_HIP, "mh_icp_wait: the stream drained without the call's results");
      }
    }
    const double tf0 = g_wt.on ? WaitTrace::now() : 0.0;
    finish_result(icp, d, pc, pc.out);
    if (g_wt.on) {
      const double tf1 = WaitTrace::now();
      g_wt.fin += tf1 - tf0;
      g_wt.fold += tf0 - tw_enter;  // (wait for the first word included: subtracted when printed)
      g_wt.n++;
    }
    if (pc.ev[0]) {
      if (!synced) (void)hipEventSynchronize(pc.ev[2]);
      (void)hipEventElapsedTime(&pc.out->gpu_ms_linearize, pc.ev[0], pc.ev[1]);
      (void)hipEventElapsedTime(&pc.out->gpu_ms_localizability, pc.ev[1], pc.ev[2]);
    } else {
      pc.out->gpu_ms_linearize = pc.out->gpu_ms_localizability = -1.0f;  // this call was not timed
    }
  }
  icp->n_pending = 0;
  return MH_OK;
}
int mh_icp_wait(mh_icp * icp)
{
  return guarded(icp ? icp->ctx : nullptr, "mh_icp_wait", [&]() -> int { return mh_icp_wait_impl(icp); });
}

static int mh_icp_linearize_impl(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                     const double * t_tgt, const double g_unit[3], mh_icp_result * out)
{
  const double t0 = g_wt.on ? WaitTrace::now() : 0.0;
  const int rc = linearize_enqueue(icp, R_src, t_src, R_tgt, t_tgt, g_unit, out, icp && icp->n_pending == 0);
  if (rc != MH_OK) return rc;
  if (g_wt.on) g_wt.enq += WaitTrace::now() - t0;
  return mh_icp_wait(icp);
}
int mh_icp_linearize(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                     const double * t_tgt, const double g_unit[3], mh_icp_result * out)
{
  return guarded(icp ? icp->ctx : nullptr, "mh_icp_linearize", [&]() -> int { return mh_icp_linearize_impl(icp, R_src, t_src, R_tgt, t_tgt, g_unit, out); });
}

// ---- all live factors of the sliding window in two launches ------------------------------------------------
// graph::Manager::defineNoLock (src/graph/manager.cpp:585-588): smoother_->update() + additional_update_iterations
// re-linearize EVERY live ICPFactor whose pose moved; GTSAM calls them one after the other.  Here one K3 grid
// and one K4 grid cover all of them (each factor keeps its own partial rows / ticket / fold: bit-identical to
// separate calls), so the machine sees sum(n_i) points at once instead of 10-25 k.
static int mh_icp_linearize_batch_impl(mh_icp * const * icps, size_t n_factors, const double * R_src, const double * t_src,
                           const double * R_tgt, const double * t_tgt, const double * g_unit, mh_icp_result * out)
{
  if (!icps || !n_factors || !R_src || !t_src || !g_unit || !out)
    return fail(nullptr, MH_ERR_INVALID_ARG, "mh_icp_linearize_batch: NULL argument");
  if (n_factors > static_cast<size_t>(kMaxBatch)) return fail(nullptr, MH_ERR_UNSUPPORTED, "mh_icp_linearize_batch: at most 64 factors per call");
  for (size_t f = 0; f < n_factors; ++f)
    if (!icps[f]) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_icp_linearize_batch: NULL factor");
  mh_ctx * ctx = icps[0]->ctx;
  for (size_t f = 0; f < n_factors; ++f) {
    const mh_icp * c = icps[f];
    if (c->ctx != ctx) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_linearize_batch: factors of different contexts");
    if (c->n_pending) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_linearize_batch: a factor has calls in flight");
    if (c->binary && (!R_tgt || !t_tgt)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_linearize_batch: binary factors need target poses");
    for (size_t g = 0; g < f; ++g)
      if (icps[g] == c) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_linearize_batch: the same factor twice");
  }
  MH_HIP(ctx, mh_enter(ctx));
  const double tb0 = g_wt.on ? WaitTrace::now() : 0.0;
  // Launch groups: the factors that share a kernel instantiation — workgroup size (256 threads up to 65 536 points, 512
  // above: so every factor reduces in exactly the order of a separate call), k == 5 or the generic k <= 8 path, neighbour
  // mode, unary / binary.  One K3b (+ one K4b) launch per non-empty group; a window of like factors — the usual case — is
  // one group.  Staging: [IcpArgs x 64 | LocArgs x 64 | grid prefixes], host-pinned + a device copy the kernels read.
  struct Group
  {
    int tpb, k, n_off;
    bool binary;
    std::vector<size_t> members;
    int first = 0, grid = 0, grid4 = 0;
  };
  std::vector<Group> groups;
  long long total_points = 0;  // the class of a small cloud depends on how full the machine is: the whole window's points
  for (size_t f = 0; f < n_factors; ++f) total_points += static_cast<long long>(icps[f]->n);
  // (more factors than ride in the kernel-argument segment: the staged launch form has the one-lane-per-point classes only)
  const bool maybe_staged = n_factors > static_cast<size_t>(mh::kBatchInline);
  for (size_t f = 0; f < n_factors; ++f) {
    const mh_icp * c = icps[f];
    if (c->n == 0) continue;
    const int k = c->cfg.num_corres_points == 5 ? 5 : 8, n_off = c->map->n_off;
    int tpb = mh::linearize_class(static_cast<int>(c->n), static_cast<int>(c->cfg.num_corres_points), false, total_points);
    if (maybe_staged && tpb < 256) tpb = 256;
    Group * g = nullptr;
    for (Group & q : groups)
      if (q.tpb == tpb && q.k == k && q.n_off == n_off && q.binary == c->binary) g = &q;
    if (!g) {
      groups.push_back(Group{tpb, k, n_off, c->binary, {}, 0, 0, 0});
      g = &groups.back();
    }
    g->members.push_back(f);
  }
  const size_t ab = sizeof(mh::IcpArgs) * kMaxBatch, lb = sizeof(mh::LocArgs) * kMaxBatch, sb = sizeof(int) * 4 * (kMaxBatch + 1);  // two prefix tables (K3b, K4b), each up to factors + groups entries
  const size_t total = ((ab + lb + sb + 255) & ~size_t(255)) + 256;
  if (!ctx->h_batch) MH_HIP(ctx, hipHostMalloc(&ctx->h_batch, total, hipHostMallocDefault));
  if (!ctx->d_batch) MH_HIP(ctx, hipMalloc(&ctx->d_batch, total));
  auto * h_a = reinterpret_cast<mh::IcpArgs *>(ctx->h_batch);
  auto * h_l = reinterpret_cast<mh::LocArgs *>(static_cast<char *>(ctx->h_batch) + ab);
  int * h_s = reinterpret_cast<int *>(static_cast<char *>(ctx->h_batch) + ab + lb);  // group g's prefix starts at h_s[first + g]
  size_t slot_of[kMaxBatch];
  {
    int pos = 0;
    for (Group & g : groups) {
      g.first = pos;
      for (size_t f : g.members) slot_of[f] = static_cast<size_t>(pos++);
    }
  }
  std::vector<LinearizeTxn> txns;  // every factor's handle goes back to what it was unless ALL launches went through
  txns.reserve(n_factors);
  for (size_t f = 0; f < n_factors; ++f) {
    bool timed = false;
    mh::IcpArgs a;
    mh::LocArgs l;
    txns.emplace_back(icps[f]);
    const int rc = linearize_prepare(icps[f], R_src + 9 * f, t_src + 3 * f, R_tgt ? R_tgt + 9 * f : nullptr,
                                     t_tgt ? t_tgt + 3 * f : nullptr, g_unit + 3 * f, out + f, true, false, a, l, timed);
    if (rc != MH_OK) return rc;
    if (icps[f]->n == 0) continue;  // (pc.seq == 0: an all-zero result is assembled at the wait)
    h_a[slot_of[f]] = a;
    h_l[slot_of[f]] = l;
  }
  // K4 is skipped when NO factor of the batch wants its components (the smoother's re-linearizations: nobody reads
  // them, geometric.cpp:205-214 is the only reader); a mixed batch runs it for all, the others report NaN all the same
  bool any_components = false;
  for (size_t f = 0; f < n_factors; ++f) any_components = any_components || (icps[f]->n && icps[f]->components);
  for (size_t f = 0; f < n_factors; ++f)
    if (icps[f]->n) h_a[slot_of[f]].tail = any_components ? 0 : 1;  // K4b follows and folds the rows / K3b is the whole call
  bool inline_args = true;
  int * h_s4 = h_s + 2 * (kMaxBatch + 1);  // the same prefixes for K4b's (smaller) per-factor grids
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    Group & g = groups[gi];
    int * start = h_s + g.first + static_cast<int>(gi);  // prefix of the group's grids, in slot order
    int * start4 = h_s4 + g.first + static_cast<int>(gi);
    int acc = 0, acc4 = 0;
    for (size_t i = 0; i < g.members.size(); ++i) {
      const int slot = g.first + static_cast<int>(i);
      start[i] = acc;
      start4[i] = acc4;
      acc += mh::class_grid(h_a[slot].n, g.tpb);
      acc4 += mh::class_loc_grid(h_a[slot].n, g.tpb);
      h_l[slot].k3_blocks = mh::class_grid(h_a[slot].n, g.tpb);
      h_l[slot].chunks_per_block = mh::kLocChunks;
      icps[g.members[i]]->pending[0].loc_blocks = mh::class_loc_grid(h_a[slot].n, g.tpb);
    }
    start[g.members.size()] = acc;
    start4[g.members.size()] = acc4;
    g.grid = acc;
    g.grid4 = acc4;
    inline_args = inline_args && static_cast<int>(g.members.size()) <= mh::kBatchInline;
  }
  const double tb1 = g_wt.on ? WaitTrace::now() : 0.0;
  if (!groups.empty()) {
    if (inline_args) {
      // small window: the argument blocks ride in the kernel-argument segment, nothing is copied before the launches
      for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group & g = groups[gi];
        const int * start = h_s + g.first + static_cast<int>(gi);
        mh::BatchInline<mh::IcpArgs> blk;
        std::memset(static_cast<void *>(&blk), 0, sizeof(blk));
        for (size_t i = 0; i < g.members.size(); ++i) blk.a[i] = h_a[g.first + static_cast<int>(i)];
        for (size_t i = 0; i <= g.members.size(); ++i) blk.start[i] = start[i];
        blk.n = static_cast<int>(g.members.size());
        MH_HIP(ctx, mh::launch_linearize_batch_inline(blk, g.grid, g.tpb, g.k, g.n_off, g.binary, ctx->stream));
      }
      for (size_t gi = 0; gi < groups.size() && any_components; ++gi) {
        const Group & g = groups[gi];
        const int * start4 = h_s4 + g.first + static_cast<int>(gi);
        mh::BatchInline<mh::LocArgs> blk;
        std::memset(static_cast<void *>(&blk), 0, sizeof(blk));
        for (size_t i = 0; i < g.members.size(); ++i) blk.a[i] = h_l[g.first + static_cast<int>(i)];
        for (size_t i = 0; i <= g.members.size(); ++i) blk.start[i] = start4[i];
        blk.n = static_cast<int>(g.members.size());
        MH_HIP(ctx, mh::launch_localizability_batch_inline(blk, g.grid4, g.tpb, ctx->stream));
      }
    } else {
      char * d = static_cast<char *>(ctx->d_batch);
      MH_HIP(ctx, hipMemcpyAsync(d, ctx->h_batch, ab + lb + sb, hipMemcpyHostToDevice, ctx->stream));
      for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group & g = groups[gi];
        const auto * da = reinterpret_cast<const mh::IcpArgs *>(d) + g.first;
        const int * ds = reinterpret_cast<const int *>(d + ab + lb) + g.first + static_cast<int>(gi);
        MH_HIP(ctx, mh::launch_linearize_batch(da, ds, static_cast<int>(g.members.size()), g.grid, g.tpb, g.k, g.n_off, g.binary, ctx->stream));
      }
      for (size_t gi = 0; gi < groups.size() && any_components; ++gi) {
        const Group & g = groups[gi];
        const auto * dl = reinterpret_cast<const mh::LocArgs *>(d + ab) + g.first;
        const int * ds4 = reinterpret_cast<const int *>(d + ab + lb) + 2 * (kMaxBatch + 1) + g.first + static_cast<int>(gi);
        MH_HIP(ctx, mh::launch_localizability_batch(dl, ds4, static_cast<int>(g.members.size()), g.grid4, g.tpb, ctx->stream));
      }
    }
  }
  if (any_components)
    for (size_t f = 0; f < n_factors; ++f)
      if (icps[f]->n) {
        icps[f]->pending[0].launched_k4 = true;  // its sums come from K4b's workgroup 0
        if (icps[f]->components) icps[f]->pending[0].seq_has_basis = true;  // ... and K4b published the bases it used
      }
  for (LinearizeTxn & t : txns) t.commit();
  const double tb2 = g_wt.on ? WaitTrace::now() : 0.0;
  int rc_all = MH_OK;
  for (size_t f = 0; f < n_factors; ++f) {
    const int rc = mh_icp_wait(icps[f]);
    if (rc != MH_OK) rc_all = rc;
  }
  static long b_seen = 0;
  if (g_wt.on && ++b_seen > 10) {  // (the first launches of a kernel instantiation load its code object: not the steady state)
    g_wt.b_prep += tb1 - tb0;
    g_wt.b_launch += tb2 - tb1;
    g_wt.b_wait += WaitTrace::now() - tb2;
    g_wt.b_n++;
  }
  return rc_all;
}
int mh_icp_linearize_batch(mh_icp * const * icps, size_t n_factors, const double * R_src, const double * t_src,
                           const double * R_tgt, const double * t_tgt, const double * g_unit, mh_icp_result * out)
{
  return guarded((icps && n_factors && icps[0]) ? icps[0]->ctx : nullptr, "mh_icp_linearize_batch", [&]() -> int { return mh_icp_linearize_batch_impl(icps, n_factors, R_src, t_src, R_tgt, t_tgt, g_unit, out); });
}

static int mh_icp_get_state_impl(const mh_icp * icp, int32_t * status, double * means, double * normals)
{
  if (!icp) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_icp_get_state: icp is NULL");
  mh_ctx * ctx = icp->ctx;
  MH_HIP(ctx, mh_enter(ctx));
  const size_t n = icp->n;
  if (n == 0) return MH_OK;
  if (icp->cold) {  // fresh factor / after mh_icp_reset: commonConstructor()'s zeros (geometric_factor.hpp:144-156)
    if (status) std::memset(status, 0, n * sizeof(int32_t));
    if (means) std::memset(means, 0, n * 3 * sizeof(double));
    if (normals) std::memset(normals, 0, n * 3 * sizeof(double));
    return MH_OK;
  }
  const int32_t * d_st = static_cast<const int32_t *>(icp->d_status.p);
  const double * d_mean = static_cast<const double *>(icp->d_mean.p);
  const double * d_nrm = static_cast<const double *>(icp->d_normal.p);
  DevTemp<int32_t> t_st;
  DevTemp<double> t_mean, t_nrm;
  if (icp->ordered) {  // back to the caller's point order
    if (status) MH_HIP(ctx, t_st.alloc(n * sizeof(int32_t)));
    if (means) MH_HIP(ctx, t_mean.alloc(n * 3 * sizeof(double)));
    if (normals) MH_HIP(ctx, t_nrm.alloc(n * 3 * sizeof(double)));
    MH_HIP(ctx, mh::launch_unpermute_state(static_cast<const uint32_t *>(icp->d_perm.p), static_cast<int>(n), d_st, d_mean,
                                           d_nrm, t_st.p, t_mean.p, t_nrm.p, ctx->stream));
    d_st = t_st.p;
    d_mean = t_mean.p;
    d_nrm = t_nrm.p;
  }
  if (status) MH_HIP(ctx, hipMemcpyAsync(status, d_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (means) MH_HIP(ctx, hipMemcpyAsync(means, d_mean, n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (normals) MH_HIP(ctx, hipMemcpyAsync(normals, d_nrm, n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int mh_icp_get_state(const mh_icp * icp, int32_t * status, double * means, double * normals)
{
  return guarded(icp ? icp->ctx : nullptr, "mh_icp_get_state", [&]() -> int { return mh_icp_get_state_impl(icp, status, means, normals); });
}

// ---- deskew / transforms -----------------------------------------------------------------------
static int mh_deskew_impl(mh_ctx * ctx, mh_point32 * pts, size_t n, const uint32_t * unique_ns, const float * Rt12, size_t n_groups,
              const float * R_B_L, const float * t_B_L)
{
  if (!ctx || (!pts && n) || (!unique_ns && n_groups) || (!Rt12 && n_groups))
    return fail(ctx, MH_ERR_INVALID_ARG, "mh_deskew: NULL argument");
  if ((R_B_L == nullptr) != (t_B_L == nullptr)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_deskew: R_B_L and t_B_L go together");
  if (n == 0) return MH_OK;
  MH_HIP(ctx, mh_enter(ctx));
  DevTemp<mh_point32> d_pts;
  DevTemp<uint32_t> d_ns;
  DevTemp<float> d_rt, d_body;
  MH_HIP(ctx, d_pts.alloc(n * sizeof(mh_point32)));
  MH_HIP(ctx, d_ns.alloc((n_groups ? n_groups : 1) * sizeof(uint32_t)));
  MH_HIP(ctx, d_rt.alloc((n_groups ? n_groups : 1) * 12 * sizeof(float)));
  MH_HIP(ctx, hipMemcpyAsync(d_pts, pts, n * sizeof(mh_point32), hipMemcpyHostToDevice, ctx->stream));
  if (n_groups) {
    MH_HIP(ctx, hipMemcpyAsync(d_ns, unique_ns, n_groups * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(d_rt, Rt12, n_groups * 12 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  }
  if (R_B_L) {
    float body[12];
    std::memcpy(body, R_B_L, 9 * sizeof(float));
    std::memcpy(body + 9, t_B_L, 3 * sizeof(float));
    MH_HIP(ctx, d_body.alloc(sizeof(body)));
    MH_HIP(ctx, hipMemcpyAsync(d_body, body, sizeof(body), hipMemcpyHostToDevice, ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `body` is a stack buffer
  }
  MH_HIP(ctx, mh::launch_deskew(d_pts, static_cast<int>(n), d_ns, d_rt, static_cast<int>(n_groups), d_body, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(pts, d_pts, n * sizeof(mh_point32), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int mh_deskew(mh_ctx * ctx, mh_point32 * pts, size_t n, const uint32_t * unique_ns, const float * Rt12, size_t n_groups,
              const float * R_B_L, const float * t_B_L)
{
  return guarded(ctx, "mh_deskew", [&]() -> int { return mh_deskew_impl(ctx, pts, n, unique_ns, Rt12, n_groups, R_B_L, t_B_L); });
}

static int mh_transform_f32_impl(mh_ctx * ctx, mh_point32 * pts, size_t n, const float R[9], const float t[3])
{
  if (!ctx || (!pts && n) || !R || !t) return fail(ctx, MH_ERR_INVALID_ARG, "mh_transform_f32: NULL argument");
  if (n == 0) return MH_OK;
  MH_HIP(ctx, mh_enter(ctx));
  DevTemp<mh_point32> d_pts;
  DevTemp<float> d_rt;
  float rt[12];
  std::memcpy(rt, R, 9 * sizeof(float));
  std::memcpy(rt + 9, t, 3 * sizeof(float));
  MH_HIP(ctx, d_pts.alloc(n * sizeof(mh_point32)));
  MH_HIP(ctx, d_rt.alloc(sizeof(rt)));
  MH_HIP(ctx, hipMemcpyAsync(d_pts, pts, n * sizeof(mh_point32), hipMemcpyHostToDevice, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(d_rt, rt, sizeof(rt), hipMemcpyHostToDevice, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MH_HIP(ctx, mh::launch_transform(d_pts, static_cast<int>(n), d_rt, ctx->stream));
  MH_HIP(ctx, hipMemcpyAsync(pts, d_pts, n * sizeof(mh_point32), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int mh_transform_f32(mh_ctx * ctx, mh_point32 * pts, size_t n, const float R[9], const float t[3])
{
  return guarded(ctx, "mh_transform_f32", [&]() -> int { return mh_transform_f32_impl(ctx, pts, n, R, t); });
}

}  // extern "C"

// ---- device-resident scan front end (scan_kernels.hip) ------------------------------------------

namespace
{
// pinned landing block of a scan: [ScanCounters | kUniqueCached timestamps | one flag word]
constexpr size_t kScanPinnedBytes = sizeof(mh::ScanCounters) + mh_scan::kUniqueCached * sizeof(uint32_t) + 16;
int scan_ensure_pinned(mh_scan * s)
{
  if (!s->h_c) {
    void * p = nullptr;
    MH_HIP(s->ctx, AllocCache::alloc_pinned(&p, kScanPinnedBytes));
    s->h_c = static_cast<mh::ScanCounters *>(p);
  }
  return MH_OK;
}
uint32_t * scan_pinned_flag(mh_scan * s)
{
  return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s->h_c) + sizeof(mh::ScanCounters) + mh_scan::kUniqueCached * sizeof(uint32_t));
}
int scan_fetch_counters(mh_scan * s, bool with_unique = false)
{
  mh_ctx * ctx = s->ctx;
  const int rcp = scan_ensure_pinned(s);
  if (rcp != MH_OK) return rcp;
  MH_HIP(ctx, hipMemcpyAsync(s->h_c, s->d_counters.p, sizeof(s->c), hipMemcpyDeviceToHost, ctx->stream));
  size_t n_copy = 0;
  if (with_unique) {  // the distinct timestamps ride along: the caller asks for them next (IMU propagation), one wait instead of two
    n_copy = std::min(mh_scan::kUniqueCached, s->d_unique.cap / sizeof(uint32_t));
    MH_HIP(ctx, hipMemcpyAsync(s->h_c + 1, s->d_unique.p, n_copy * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  s->c = *s->h_c;
  s->c.n_unique_ns += s->c.has_max_ns;  // the value 0xFFFFFFFF travels as a flag (scan_kernels.hip: input_scatter_kernel)
  if (with_unique) s->n_unique_cached = s->c.n_unique_ns <= n_copy ? s->c.n_unique_ns : 0;
  return MH_OK;
}
void scan_fill_info(const mh_scan * s, mh_scan_info * info)
{
  if (!info) return;
  std::memset(info, 0, sizeof(*info));
  info->n_in = s->n_in;
  info->n_full = s->c.n_full;
  info->n_geometric = s->c.n_geometric;
  info->n_unique_ns = s->c.n_unique_ns;
  info->n_body = s->n_body;
  info->n_downsampled = s->preprocessed ? s->c.n_downsampled : 0;
  info->last_point_ns = s->c.last_point_ns;
}
}  // namespace

extern "C" {

static int mh_scan_create_impl(mh_ctx * ctx, mh_scan ** out)
{
  if (!ctx || !out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_create: NULL argument");
  *out = nullptr;
  mh_scan * s = new (std::nothrow) mh_scan;
  if (!s) return fail(ctx, MH_ERR_OOM, "mh_scan_create: host allocation failed");
  s->ctx = ctx;
  *out = s;
  return MH_OK;
}
int mh_scan_create(mh_ctx * ctx, mh_scan ** out)
{
  return guarded(ctx, "mh_scan_create", [&]() -> int { return mh_scan_create_impl(ctx, out); });
}

void mh_scan_destroy(mh_scan * s)
{
  if (!s) return;
  (void)mh_enter(s->ctx);
  (void)hipStreamSynchronize(s->ctx->stream);
  // a staged upload (mh_scan_prefetch, on the context's copy stream) may still be WRITING d_raw: it is not "drained" before
  // that copy has finished
  if (s->copy_done) (void)hipEventSynchronize(s->copy_done);
  s->d_full_raw.release(true);
  for (DevBuf * b : {&s->d_raw, &s->d_full, &s->d_geo_idx, &s->d_unique, &s->d_body, &s->d_ds, &s->d_kept_idx, &s->d_counters,
                     &s->d_rt, &s->d_prep, &s->d_vox, &s->d_sensor})
    b->release(true);
  AllocCache::free_pinned(s->h_c, sizeof(mh::ScanCounters) + mh_scan::kUniqueCached * sizeof(uint32_t));
  if (s->copy_done) (void)hipEventDestroy(s->copy_done);
  if (s->compute_mark) (void)hipEventDestroy(s->compute_mark);
  if (s->h_stage) AllocCache::free_pinned(s->h_stage, s->h_stage_cap);
  if (s->rt_done) {
    (void)hipEventSynchronize(s->rt_done);
    (void)hipEventDestroy(s->rt_done);
  }
  if (s->h_rt) AllocCache::free_pinned(s->h_rt, s->h_rt_cap);
  delete s;
}

static int scan_prepare_common(mh_scan * s, const mh_ouster_point * raw, bool raw_on_device, size_t n, const mh_input_config * cfg,
                               mh_scan_info * info, const char * who, bool canonical = false, bool ring_filter = true)
{
  if (!s || !cfg || (!raw && n)) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, std::string(who) + ": NULL argument");
  mh_ctx * ctx = s->ctx;
  if (n > 0x3fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, std::string(who) + ": cloud too large");
  if (cfg->point_skip_divisor < 1 || cfg->ring_skip_divisor < 1)
    return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": skip divisors must be >= 1");
  MH_HIP(ctx, mh_enter(ctx));
  s->prepared = s->preprocessed = s->raw_valid = false;
  s->n_unique_cached = 0;
  s->n_in = n;
  s->n_body = 0;
  const size_t m = n ? n : 1;
  if (!raw_on_device && s->copy_done) {
    // a plain prepare behind a prefetch nobody consumed: the copy stream may still write d_raw — order this stream behind it,
    // and the staged cloud is gone
    MH_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->copy_done, 0));
    s->prefetch_valid = false;
    s->n_prefetched = 0;
  }
  if (!raw_on_device) MH_HIP(ctx, s->d_raw.reserve(m * sizeof(mh_ouster_point), ctx->stream, false));
  MH_HIP(ctx, s->d_full.reserve(m * sizeof(mh_point32), ctx->stream, false));
  MH_HIP(ctx, s->d_geo_idx.reserve(m * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, s->d_unique.reserve((m + 1) * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, s->d_counters.reserve(sizeof(mh::ScanCounters), ctx->stream, false));
  MH_HIP(ctx, s->d_prep.reserve(mh::prepare_layout(n).words * sizeof(uint32_t), ctx->stream, false));
  const mh_ouster_point * d_raw = raw;
  if (!raw_on_device) {
    if (n) MH_HIP(ctx, hipMemcpyAsync(s->d_raw.p, raw, n * sizeof(mh_ouster_point), hipMemcpyHostToDevice, ctx->stream));
    d_raw = static_cast<const mh_ouster_point *>(s->d_raw.p);
  }
  MH_HIP(ctx, mh::launch_prepare_input(d_raw, static_cast<uint32_t>(n), *cfg, static_cast<uint32_t *>(s->d_prep.p),
                                       static_cast<mh_point32 *>(s->d_full.p), static_cast<uint32_t *>(s->d_geo_idx.p),
                                       static_cast<uint32_t *>(s->d_unique.p), static_cast<mh::ScanCounters *>(s->d_counters.p),
                                       ctx->stream, canonical, ring_filter));
  const int rc = scan_fetch_counters(s, true);
  if (rc != MH_OK) return rc;
  s->prepared = true;
  scan_fill_info(s, info);
  return MH_OK;
}
int mh_scan_prepare_input(mh_scan * s, const mh_ouster_point * raw, size_t n, const mh_input_config * cfg, mh_scan_info * info)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_prepare_input",
                 [&]() -> int { return scan_prepare_common(s, raw, false, n, cfg, info, "mh_scan_prepare_input"); });
}
// Pipelined input: the NEXT cloud is staged (pinned buffer, host-to-device copy on the handle's own copy stream) while another
// scan is being processed; mh_scan_prepare_input_prefetched then makes the context stream wait for the copy on the device.
static int mh_scan_prefetch_impl(mh_scan * s, const mh_ouster_point * raw, size_t n)
{
  if (!s || (!raw && n)) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_prefetch: NULL argument");
  mh_ctx * ctx = s->ctx;
  if (n > 0x3fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_scan_prefetch: cloud too large");
  MH_HIP(ctx, mh_enter(ctx));
  hipStream_t copy_stream = nullptr;
  {
    std::lock_guard<std::mutex> g(ctx->copy_mu);
    if (!ctx->copy_stream) MH_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    copy_stream = ctx->copy_stream;
  }
  if (!s->copy_done) MH_HIP(ctx, hipEventCreateWithFlags(&s->copy_done, hipEventDisableTiming));
  if (!s->compute_mark) MH_HIP(ctx, hipEventCreateWithFlags(&s->compute_mark, hipEventDisableTiming));
  // d_raw is about to be overwritten (perhaps re-allocated) from the copy stream: everything the compute stream has enqueued
  // on this scan so far may read it, so the copy stream queues behind that point.  (This is what makes "freed on the stream
  // that last used it" — the allocation cache's hand-over rule — true for d_raw.)
  MH_HIP(ctx, hipEventRecord(s->compute_mark, ctx->stream));
  MH_HIP(ctx, hipStreamWaitEvent(copy_stream, s->compute_mark, 0));
  g_mh_stream = copy_stream;  // what this call allocates / frees is ordered by the copy stream
  const size_t bytes = (n ? n : 1) * sizeof(mh_ouster_point);
  if (bytes > s->h_stage_cap) {
    // pinned staging comes from the process-wide cache in 1 MiB classes (pinning 4 MiB costs ~2 ms: a front end that lives for
    // one sequence must not pay it again)
    MH_HIP(ctx, hipStreamSynchronize(copy_stream));  // an earlier copy may still read the old staging buffer
    if (s->h_stage) AllocCache::free_pinned(s->h_stage, s->h_stage_cap);
    s->h_stage = nullptr;
    s->h_stage_cap = 0;
    const size_t cap = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    MH_HIP(ctx, AllocCache::alloc_pinned(&s->h_stage, cap));
    s->h_stage_cap = cap;
  }
  s->prefetch_valid = false;
  MH_HIP(ctx, hipStreamSynchronize(copy_stream));  // the previous cloud staged here has left the buffer
  MH_HIP(ctx, s->d_raw.reserve(bytes, copy_stream, false));
  if (n) {
    std::memcpy(s->h_stage, raw, n * sizeof(mh_ouster_point));
    MH_HIP(ctx, hipMemcpyAsync(s->d_raw.p, s->h_stage, n * sizeof(mh_ouster_point), hipMemcpyHostToDevice, copy_stream));
  }
  MH_HIP(ctx, hipEventRecord(s->copy_done, copy_stream));
  s->n_prefetched = n;
  s->prefetch_valid = true;
  return MH_OK;
}
int mh_scan_prefetch(mh_scan * s, const mh_ouster_point * raw, size_t n)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_prefetch", [&]() -> int { return mh_scan_prefetch_impl(s, raw, n); });
}
int mh_scan_prepare_input_prefetched(mh_scan * s, const mh_input_config * cfg, mh_scan_info * info)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_prepare_input_prefetched", [&]() -> int {
    if (!s || !cfg) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_prepare_input_prefetched: NULL argument");
    mh_ctx * ctx = s->ctx;
    if (!s->prefetch_valid) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_prepare_input_prefetched: no mh_scan_prefetch before");
    MH_HIP(ctx, mh_enter(ctx));
    MH_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->copy_done, 0));  // on the device: the host does not wait for the copy
    s->prefetch_valid = false;
    return scan_prepare_common(s, static_cast<const mh_ouster_point *>(s->d_raw.p), true, s->n_prefetched, cfg, info, "mh_scan_prepare_input_prefetched");
  });
}

// Manager::prepareInput<PointT> for any of the reference's point types: decode into canonical records on the device, then
// the same filter / compaction / timestamp kernels as the PointOuster path.
static int mh_scan_prepare_input_layout_impl(mh_scan * s, const void * raw, size_t n, const mh_point_layout * L, uint32_t width, uint32_t height,
                                             int transpose, int organize_by_ring, double header_ts, const mh_input_config * cfg, mh_scan_info * info)
{
  const char * who = "mh_scan_prepare_input_layout";
  if (!s || !cfg || !L || (!raw && n)) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, std::string(who) + ": NULL argument");
  mh_ctx * ctx = s->ctx;
  if (n > 0x3fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, std::string(who) + ": cloud too large");
  if (static_cast<size_t>(width) * height != n) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": width * height must equal n");
  const uint32_t ends[] = {L->off_x + 4, L->off_y + 4, L->off_z + 4, L->off_intensity + (L->intensity_is_u16 ? 2u : 4u),
                           L->off_time + (L->time_kind == MH_TIME_U32_NS || L->time_kind == MH_TIME_F32_S ? 4u : 8u),
                           L->ring_kind == MH_RING_NONE ? 0u : L->off_ring + (L->ring_kind == MH_RING_U8 ? 1u : (L->ring_kind == MH_RING_U16 ? 2u : 4u)),
                           L->has_tag ? L->off_tag + 1u : 0u};
  for (const uint32_t e : ends)
    if (e > L->stride) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": a field lies outside the record");
  if (L->time_kind < MH_TIME_U32_NS || L->time_kind > MH_TIME_F32_S || L->ring_kind < MH_RING_NONE || L->ring_kind > MH_RING_F32)
    return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": unknown time / ring kind");
  if (L->ring_filter && L->ring_kind == MH_RING_NONE) return fail(ctx, MH_ERR_INVALID_ARG, std::string(who) + ": the ring filter needs a ring field");
  const uint32_t height_after = transpose ? width : height;  // the transposed cloud is height x width
  // :205-210 only an unorganised cloud is re-ordered, and only for the point types that carry a ring (compiled out for
  // PointLivox, PointLivoxFromCustom2 and PointOusterOdyssey there: the flag is ignored, not an error)
  const bool organize = organize_by_ring != 0 && height_after == 1 && L->ring_kind != MH_RING_NONE;
  MH_HIP(ctx, mh_enter(ctx));
  const size_t m = n ? n : 1, n_blk = (m + 255) / 256;
  const size_t raw_bytes = (m * L->stride + 255) & ~size_t(255), tmp_bytes = organize ? m * sizeof(mh_ouster_point) : 0;
  MH_HIP(ctx, s->d_raw.reserve(m * sizeof(mh_ouster_point), ctx->stream, false));  // the canonical records
  const size_t hist_bytes = organize ? (n_blk + 1) * 128 * sizeof(uint32_t) : 0;  // per-block ring counts + the ring starts
  MH_HIP(ctx, s->d_sensor.reserve(raw_bytes + tmp_bytes + hist_bytes + 256, ctx->stream, false));
  char * base = static_cast<char *>(s->d_sensor.p);
  auto * tmp = reinterpret_cast<mh_ouster_point *>(base + raw_bytes);
  auto * hist = reinterpret_cast<uint32_t *>(base + raw_bytes + tmp_bytes);
  uint32_t * bad_ring = reinterpret_cast<uint32_t *>(base + raw_bytes + tmp_bytes + hist_bytes);
  const int rcp = scan_ensure_pinned(s);
  if (rcp != MH_OK) return rcp;
  MH_HIP(ctx, hipMemsetAsync(bad_ring, 0, sizeof(uint32_t), ctx->stream));
  if (n) MH_HIP(ctx, hipMemcpyAsync(base, raw, n * L->stride, hipMemcpyHostToDevice, ctx->stream));
  MH_HIP(ctx, mh::launch_decode_points(base, static_cast<uint32_t>(n), *L, width, height, transpose != 0, organize, header_ts,
                                       static_cast<mh_ouster_point *>(s->d_raw.p), tmp, hist, bad_ring, ctx->stream));
  // the flag comes back with the counters of the filter pass: no synchronisation of its own (the placement kernel masks
  // the ring number, so a bad one cannot write out of bounds before it is reported)
  *scan_pinned_flag(s) = 0u;
  MH_HIP(ctx, hipMemcpyAsync(scan_pinned_flag(s), bad_ring, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  const int rc = scan_prepare_common(s, static_cast<const mh_ouster_point *>(s->d_raw.p), true, n, cfg, info, who, true, L->ring_filter != 0);
  if (rc == MH_OK && organize && *scan_pinned_flag(s)) {  // a ring number beyond the reference's 128-entry tables (undefined behaviour there)
    s->prepared = false;
    return fail(ctx, MH_ERR_UNSUPPORTED, std::string(who) + ": organize_by_ring with a ring number >= 128");
  }
  return rc;
}
int mh_scan_prepare_input_layout(mh_scan * s, const void * raw, size_t n, const mh_point_layout * layout, uint32_t width, uint32_t height,
                                 int transpose, int organize_by_ring, double header_ts, const mh_input_config * cfg, mh_scan_info * info)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_prepare_input_layout", [&]() -> int {
    return mh_scan_prepare_input_layout_impl(s, raw, n, layout, width, height, transpose, organize_by_ring, header_ts, cfg, info);
  });
}
int mh_scan_prepare_input_device(mh_scan * s, const mh_ouster_point * d_raw, size_t n, const mh_input_config * cfg, mh_scan_info * info)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_prepare_input_device",
                 [&]() -> int { return scan_prepare_common(s, d_raw, true, n, cfg, info, "mh_scan_prepare_input_device"); });
}

static int mh_scan_get_unique_ns_impl(const mh_scan * s, uint32_t * out, size_t capacity, size_t * n_out)
{
  if (!s || !n_out) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_get_unique_ns: NULL argument");
  mh_ctx * ctx = s->ctx;
  if (!s->prepared) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_get_unique_ns: no mh_scan_prepare_input before");
  *n_out = s->c.n_unique_ns;
  if (!out) return MH_OK;
  if (capacity < *n_out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_get_unique_ns: buffer too small");
  if (s->n_unique_cached == *n_out) {  // came back with the counters of mh_scan_prepare_input
    std::memcpy(out, s->h_c + 1, *n_out * sizeof(uint32_t));
    return MH_OK;
  }
  MH_HIP(ctx, mh_enter(ctx));
  if (*n_out) MH_HIP(ctx, hipMemcpyAsync(out, s->d_unique.p, *n_out * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int mh_scan_get_unique_ns(const mh_scan * s, uint32_t * out, size_t capacity, size_t * n_out)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_get_unique_ns", [&]() -> int { return mh_scan_get_unique_ns_impl(s, out, capacity, n_out); });
}

static int mh_scan_deskew_impl(mh_scan * s, const float * Rt12, size_t n_groups)
{
  if (!s || (!Rt12 && n_groups)) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_deskew: NULL argument");
  mh_ctx * ctx = s->ctx;
  if (!s->prepared) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_deskew: no mh_scan_prepare_input before");
  if (n_groups != s->c.n_unique_ns) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_deskew: one pose per unique timestamp");
  if (s->c.n_full == 0) return MH_OK;
  MH_HIP(ctx, mh_enter(ctx));
  if (s->keep_raw && !s->raw_valid) {  // points_raw_ = points_full_ before deskewing (lidar/manager.cpp:376-380)
    MH_HIP(ctx, s->d_full_raw.reserve(s->c.n_full * sizeof(mh_point32), ctx->stream, false));
    MH_HIP(ctx, mh::launch_copy16(s->d_full.p, s->d_full_raw.p, s->c.n_full * sizeof(mh_point32), ctx->stream));
    s->raw_valid = true;
  }
  MH_HIP(ctx, s->d_rt.reserve((n_groups + 1) * 12 * sizeof(float) + 16, ctx->stream, false));
  {
    // the poses leave the caller's buffer here, on the host: pinned block -> device in stream order, no wait
    const size_t bytes = n_groups * 12 * sizeof(float);
    if (s->rt_done) MH_HIP(ctx, hipEventSynchronize(s->rt_done));  // the previous call's copy out of the block (long done)
    if (bytes > s->h_rt_cap) {
      size_t cap = size_t(64) << 10;
      while (cap < bytes) cap <<= 1;
      if (s->h_rt) AllocCache::free_pinned(s->h_rt, s->h_rt_cap);
      s->h_rt = nullptr;
      s->h_rt_cap = 0;
      MH_HIP(ctx, AllocCache::alloc_pinned(&s->h_rt, cap));
      s->h_rt_cap = cap;
    }
    if (!s->rt_done) MH_HIP(ctx, hipEventCreateWithFlags(&s->rt_done, hipEventDisableTiming));
    std::memcpy(s->h_rt, Rt12, bytes);
    // a copy KERNEL reading the mapped block, not hipMemcpyAsync: a small host-to-device copy call was seen to block its caller
    // for 0.1-0.4 ms whenever another thread's 4 MiB upload (mh_scan_prefetch of the next cloud) was in flight
    void * d_src = nullptr;
    MH_HIP(ctx, hipHostGetDevicePointer(&d_src, s->h_rt, 0));
    MH_HIP(ctx, mh::launch_copy16(d_src, s->d_rt.p, (bytes + 15) & ~size_t(15), ctx->stream));
    MH_HIP(ctx, hipEventRecord(s->rt_done, ctx->stream));
  }
  MH_HIP(ctx, mh::launch_deskew(static_cast<mh_point32 *>(s->d_full.p), static_cast<int>(s->c.n_full),
                                static_cast<const uint32_t *>(s->d_unique.p), static_cast<const float *>(s->d_rt.p),
                                static_cast<int>(n_groups), nullptr, ctx->stream));
  s->preprocessed = false;
  return MH_OK;
}
int mh_scan_deskew(mh_scan * s, const float * Rt12, size_t n_groups)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_deskew", [&]() -> int { return mh_scan_deskew_impl(s, Rt12, n_groups); });
}

static int mh_scan_preprocess_geometric_impl(mh_scan * s, const float R_B_L[9], const float t_B_L[3], double leaf_size,
                                 int max_points_per_voxel, double min_dist_in_voxel, mh_scan_info * info)
{
  if (!s || !R_B_L || !t_B_L) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_preprocess_geometric: NULL argument");
  mh_ctx * ctx = s->ctx;
  if (!s->prepared) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_preprocess_geometric: no mh_scan_prepare_input before");
  if (!(leaf_size > 0.0)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_preprocess_geometric: leaf_size must be > 0");
  if (max_points_per_voxel < 1 || max_points_per_voxel > mh::kBucketStride)
    return fail(ctx, MH_ERR_UNSUPPORTED, "mh_scan_preprocess_geometric: max_points_per_voxel must be in 1..20");
  MH_HIP(ctx, mh_enter(ctx));
  const size_t n = s->c.n_geometric, m = n ? n : 1;
  s->n_body = n;
  MH_HIP(ctx, s->d_body.reserve(m * sizeof(mh_point32), ctx->stream, false));
  MH_HIP(ctx, s->d_ds.reserve(m * sizeof(mh_point32), ctx->stream, false));
  MH_HIP(ctx, s->d_kept_idx.reserve(m * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, s->d_vox.reserve(mh::voxel_layout(n).bytes, ctx->stream, false));
  mh::Rt12 rt;  // passed to the kernel by value: no staging copy
  std::memcpy(rt.v, R_B_L, 9 * sizeof(float));
  std::memcpy(rt.v + 9, t_B_L, 3 * sizeof(float));
  auto * cnt = static_cast<mh::ScanCounters *>(s->d_counters.p);
  MH_HIP(ctx, mh::launch_preprocess(static_cast<const mh_point32 *>(s->d_full.p), static_cast<const uint32_t *>(s->d_geo_idx.p),
                                    static_cast<uint32_t>(n), rt, leaf_size, static_cast<uint32_t>(max_points_per_voxel),
                                    min_dist_in_voxel, s->d_vox.p, static_cast<mh_point32 *>(s->d_body.p),
                                    static_cast<uint32_t *>(s->d_kept_idx.p), static_cast<mh_point32 *>(s->d_ds.p), cnt, ctx->stream));
  const int rc = scan_fetch_counters(s);
  if (rc != MH_OK) return rc;
  if (n == 0) s->c.n_downsampled = s->c.bad_coord = 0;
  if (s->c.bad_coord) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_scan_preprocess_geometric: a voxel coordinate exceeds +-2^20");
  s->preprocessed = true;
  scan_fill_info(s, info);
  return MH_OK;
}
int mh_scan_preprocess_geometric(mh_scan * s, const float R_B_L[9], const float t_B_L[3], double leaf_size,
                                 int max_points_per_voxel, double min_dist_in_voxel, mh_scan_info * info)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_preprocess_geometric", [&]() -> int { return mh_scan_preprocess_geometric_impl(s, R_B_L, t_B_L, leaf_size, max_points_per_voxel, min_dist_in_voxel, info); });
}

static int mh_scan_get_points_impl(const mh_scan * s, int which, mh_point32 * out, size_t capacity, size_t * n_out)
{
  if (!s || !n_out) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_get_points: NULL argument");
  mh_ctx * ctx = s->ctx;
  if (!s->prepared || (which != 0 && !s->preprocessed) || which < 0 || which > 2)
    return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_get_points: that stage has not run");
  const DevBuf & b = which == 0 ? s->d_full : (which == 1 ? s->d_body : s->d_ds);
  *n_out = which == 0 ? s->c.n_full : (which == 1 ? s->n_body : s->c.n_downsampled);
  if (!out) return MH_OK;
  if (capacity < *n_out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_get_points: buffer too small");
  MH_HIP(ctx, mh_enter(ctx));
  if (*n_out) MH_HIP(ctx, hipMemcpyAsync(out, b.p, *n_out * sizeof(mh_point32), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int mh_scan_get_points(const mh_scan * s, int which, mh_point32 * out, size_t capacity, size_t * n_out)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_get_points", [&]() -> int { return mh_scan_get_points_impl(s, which, out, capacity, n_out); });
}

int mh_scan_device_points(const mh_scan * s, int which, const mh_point32 ** d_points, size_t * n_out)
{
  if (!s || !d_points || !n_out) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_device_points: NULL argument");
  if (!s->prepared || (which != 0 && !s->preprocessed) || which < 0 || which > 2)
    return fail(s->ctx, MH_ERR_INVALID_ARG, "mh_scan_device_points: that stage has not run");
  const DevBuf & b = which == 0 ? s->d_full : (which == 1 ? s->d_body : s->d_ds);
  *d_points = static_cast<const mh_point32 *>(b.p);
  *n_out = which == 0 ? s->c.n_full : (which == 1 ? s->n_body : s->c.n_downsampled);
  return MH_OK;
}

static int mh_scan_get_indices_impl(const mh_scan * s, int which, uint32_t * out, size_t capacity, size_t * n_out)
{
  if (!s || !n_out) return fail(s ? s->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_scan_get_indices: NULL argument");
  mh_ctx * ctx = s->ctx;
  if (!s->prepared || (which == 1 && !s->preprocessed) || which < 0 || which > 1)
    return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_get_indices: that stage has not run");
  const DevBuf & b = which == 0 ? s->d_geo_idx : s->d_kept_idx;
  *n_out = which == 0 ? s->c.n_geometric : s->c.n_downsampled;
  if (!out) return MH_OK;
  if (capacity < *n_out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_scan_get_indices: buffer too small");
  MH_HIP(ctx, mh_enter(ctx));
  if (*n_out) MH_HIP(ctx, hipMemcpyAsync(out, b.p, *n_out * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int mh_scan_get_indices(const mh_scan * s, int which, uint32_t * out, size_t capacity, size_t * n_out)
{
  return guarded(s ? s->ctx : nullptr, "mh_scan_get_indices", [&]() -> int { return mh_scan_get_indices_impl(s, which, out, capacity, n_out); });
}

static int mh_icp_create_from_scan_impl(mh_ctx * ctx, mh_map * map, const mh_scan * s, const mh_reg_config * cfg, int is_binary,
                            mh_icp ** out)
{
  if (!ctx || !map || !s || !cfg || !out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_create_from_scan: NULL argument");
  if (!s->preprocessed) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_create_from_scan: no mh_scan_preprocess_geometric before");
  if (s->ctx->device != ctx->device) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_create_from_scan: scan lives on another device");
  // the scan's stream has been synchronised by mh_scan_preprocess_geometric: its buffers are complete.  The factor reads
  // them asynchronously on ITS context's stream: stream order protects them when that is the scan's stream too
  const int rc = icp_create_common(ctx, map, nullptr, static_cast<const mh_point32 *>(s->d_ds.p), s->c.n_downsampled, cfg, is_binary, out);
  if (rc == MH_OK && s->ctx != ctx) MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return rc;
}
int mh_icp_create_from_scan(mh_ctx * ctx, mh_map * map, const mh_scan * s, const mh_reg_config * cfg, int is_binary,
                            mh_icp ** out)
{
  return guarded(ctx, "mh_icp_create_from_scan", [&]() -> int { return mh_icp_create_from_scan_impl(ctx, map, s, cfg, is_binary, out); });
}

}  // extern "C"

// ---- contexts on a caller's stream, factors over device-resident clouds (the native sharded path and frameworks that own the stream)
extern "C" {

static int mh_init_on_stream_impl(int device, void * hip_stream, mh_ctx ** out)
{
  mh_ctx * ctx = nullptr;
  const int rc = mh_init(device, &ctx);
  if (rc != MH_OK) return rc;
  (void)hipStreamDestroy(ctx->stream);
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  ctx->owns_stream = false;
  *out = ctx;
  return MH_OK;
}
int mh_init_on_stream(int device, void * hip_stream, mh_ctx ** out)
{
  if (!out) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_init_on_stream: out is NULL");
  *out = nullptr;
  return guarded(nullptr, "mh_init_on_stream", [&]() -> int { return mh_init_on_stream_impl(device, hip_stream, out); });
}

static int mh_icp_create_from_device_impl(mh_ctx * ctx, mh_map * map, const mh_point32 * d_points, size_t n, const mh_reg_config * cfg, int is_binary,
                                          mh_icp ** out)
{
  if (!ctx || !map || !cfg || !out || (!d_points && n)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_icp_create_from_device: NULL argument");
  return icp_create_common(ctx, map, nullptr, n ? d_points : reinterpret_cast<const mh_point32 *>(map->d_vox.p), n, cfg, is_binary, out, true);
}
int mh_icp_create_from_device(mh_ctx * ctx, mh_map * map, const mh_point32 * d_points, size_t n, const mh_reg_config * cfg, int is_binary, mh_icp ** out)
{
  return guarded(ctx, "mh_icp_create_from_device", [&]() -> int { return mh_icp_create_from_device_impl(ctx, map, d_points, n, cfg, is_binary, out); });
}

}  // extern "C"

// ---- internals shared with shard_api.hip (declared in mh_internal.hpp) ---------------------------------------------
// MH_ALLOC_CHECK (mh_internal.hpp): words of a re-used cached block that are not the poison pattern any more
namespace mh
{
namespace
{
__global__ void alloc_verify_kernel(const unsigned int * p, size_t n_words, unsigned long long * violations)
{
  unsigned long long bad = 0;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n_words; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    bad += p[i] != kAllocPoison ? 1ull : 0ull;
  if (bad) atomicAdd_system(violations, bad);
}
}  // namespace
hipError_t launch_alloc_verify(const void * p, size_t bytes, unsigned long long * violations, hipStream_t stream)
{
  unsigned long long * d_v = nullptr;
  if (hipHostGetDevicePointer(reinterpret_cast<void **>(&d_v), violations, 0) != hipSuccess) return hipErrorInvalidValue;
  const size_t n = bytes / 4;
  const int grid = static_cast<int>(std::min<size_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(alloc_verify_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, static_cast<const unsigned int *>(p), n, d_v);
  return hipGetLastError();
}
}  // namespace mh

extern "C" int mh_alloc_check_stats(unsigned long long * blocks_verified, unsigned long long * words_overwritten)
{
  if (!AllocCache::checking() || !AllocCache::check_counters()) return MH_ERR_UNSUPPORTED;
  (void)hipDeviceSynchronize();
  if (blocks_verified) *blocks_verified = AllocCache::check_counters()[0];
  if (words_overwritten) *words_overwritten = AllocCache::check_counters()[1];
  return MH_OK;
}

namespace mhi
{
void pose_delta(const double * Rs, const double * ts, const double * Rt, const double * tt, double * R, double * t)
{
  pose_inverse_compose(Rs, ts, Rt, tt, R, t);
}
void finish(const mh_icp * icp, const mh::DeviceResult & d, const PendingCall & pc, mh_icp_result * out) { finish_result(icp, d, pc, out); }
int icp_create(mh_ctx * ctx, mh_map * map, const mh_point32 * source, const mh_point32 * d_source, size_t n, size_t capacity,
               const mh_reg_config * cfg, int is_binary, mh_icp ** out, bool no_order)
{
  return icp_create_common(ctx, map, source, d_source, n, cfg, is_binary, out, no_order, capacity);
}
int prepare(mh_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt, const double * t_tgt, const double g_unit[3],
            mh_icp_result * out, bool want_flag, mh::IcpArgs & a, mh::LocArgs & l)
{
  bool timed = false;
  return linearize_prepare(icp, R_src, t_src, R_tgt, t_tgt, g_unit, out, want_flag, false, a, l, timed);
}
}  // namespace mhi
