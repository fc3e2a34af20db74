// HIP kernels (gfx950 / CDNA4, wave64) for ICPFactor::linearize.
//
// Reference: include/mimosa/lidar/geometric_factor.hpp:231-562 (linearize), :176-229
// (estimatePlane), src/lidar/incremental_voxel_map.cpp:26-32 -> gtsam_points::iVox::knn_search.
// SURVEY.md Appendix A is the arithmetic spec; kernels K3 / K4 of SURVEY.md §2.3.
//
//   K3  icp_linearize_kernel   one thread per source point (two for small clouds), 512-thread workgroups (one per CU at
//       131 072 points; 256 threads below 65 536), fused:
//         pose transform (fp64) -> data-association cache test
//         A. neighbourhood lookup: one block-table probe + nine 12-byte loads from the block's halo'd cell
//            table; cell words to a per-lane LDS column, quad counts packed in registers
//         B. coarse scan of the packed 10-bit buckets (one 16-byte load = 4 candidates): centre voxel, exact
//            box pruning, then faces -> edges -> corners through a software-pipelined register-only cursor
//            with re-pruning; sorted-quad bitonic merge into a top-8 of 32-bit keys (distance | position)
//         C. exact tier: the 8 survivors re-ranked in fp64 in the reference's operation order, proof check,
//            wave-cooperative KnnResult::push for the rare lane the proof does not cover; mean / covariance /
//            Newton cubic eigen / plane gates -> residual, Huber weight, Jacobian row
//         D. every wave: LDS tile of its rows -> a lane owns one (entry, point-segment) of the 28 (unary) or 91 (binary)
//            sums of v v^T -> per-block partial row.  K4 follows and folds the rows; without K4 (components off) or for
//            a map-sharded factor: rows written through -> ticket -> the last-arriving block folds them in a fixed order.
//   K4  icp_localizability_kernel   second pass: component localizabilities in that eigenbasis
//       (geometric_factor.hpp:434-457) + status histogram (src/lidar/geometric.cpp:280-323).
//
// Gather / scan / reduce work bound by VALU issue and the memory system at 2 waves per SIMD, not a dense
// contraction: no MFMA.  At 131 072 points, where every SIMD has its two waves, one lane per query is the fastest form
// (round 1: cooperating sub-groups of 2/4/8 lanes per query with shuffle-min merges were 1.4x/2.5x/4.9x SLOWER, and
// finishing the heaviest lanes' voxels with the whole wave was 2x slower): there the wave-level
// cooperation lives in the reductions and in the exact fallback, not in the scan.  Launches of up to 32 768 points —
// the clouds the reference feeds the factor — leave most SIMDs without a wave and run TWO adjacent lanes per query
// ("several lanes per query" below: DPP quad merges of the lanes' top-8 lists; round 6).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "icp_device.hpp"

#ifndef MH_PRUNE_TRIPS
#define MH_PRUNE_TRIPS (trip == 1 || trip == 2)  // when the scan loop re-applies the box-distance pruning
#endif
#ifndef MH_PIPE
#define MH_PIPE 4  // quads in flight per lane in the neighbour scan (knn_query)
#endif
#ifndef MH_XCD_PIECES
#define MH_XCD_PIECES 2  // pieces of the Morton curve an XCD's chunks come from (icp_linearize_body: which chunk a wave takes)
#endif
#ifndef MH_PIPE_SMALL
#define MH_PIPE_SMALL 4  // the same for the 256-thread workgroup class (6 and 8 measured in round 5: 21.6 / 21.7 / 21.4 us at 24 576 points — no effect)
#endif
#include "map_device.hpp"
#include "math3.hpp"
#include "wave_dpp.hpp"

namespace mh
{
namespace
{
constexpr int kThreads = 512;
constexpr int kMaxOff = 27;
constexpr double kDblMax = 1.7976931348623157e308;

// Neighbour-voxel offsets in gtsam_points' generation order (mh::neighbor_offsets in voxel_map.hpp is the
// host twin) as compile-time packed codes (i+1) | (j+1) << 2 | (k+1) << 4.  Rows: mode 1, 7, 19, 27.
constexpr uint32_t kOffCode[4][kMaxOff] = {
  {21},
  {21, 22, 20, 25, 17, 37, 5},
  {16, 4, 20, 36, 24, 1, 17, 33, 5, 21, 37, 9, 25, 41, 18, 6, 22, 38, 26},
  {0, 16, 32, 4, 20, 36, 8, 24, 40, 1, 17, 33, 5, 21, 37, 9, 25, 41, 2, 18, 34, 6, 22, 38, 10, 26, 42}};
// Scan order of the neighbour voxels: centre, then the 6 faces, the 12 edges, the 8 corners (each class
// in generation order).  The coarse tier is order-independent (it keeps a SET of survivors), so it visits
// the voxels most likely to tighten the pruning bound first; the traversal rank o that KnnResult::push's
// tie rule needs is recovered per survivor.  ent[row][b] = offset code | o << 6 of scan position b,
// pos[row][o] = scan position of traversal rank o.
struct ScanTab
{
  uint32_t ent[4][kMaxOff];
  uint32_t pos[4][kMaxOff];
};
constexpr int kRowN[4] = {1, 7, 19, 27};
constexpr ScanTab make_scan_tab()
{
  ScanTab t{};
  for (int r = 0; r < 4; ++r) {
    int b = 0;
    for (int cls = 0; cls <= 3; ++cls)
      for (int o = 0; o < kRowN[r]; ++o) {
        const uint32_t c = kOffCode[r][o];
        const int nz = ((c & 3u) != 1u ? 1 : 0) + (((c >> 2) & 3u) != 1u ? 1 : 0) + (((c >> 4) & 3u) != 1u ? 1 : 0);
        if (nz == cls) {
          t.ent[r][b] = c | (static_cast<uint32_t>(o) << 6);
          t.pos[r][o] = static_cast<uint32_t>(b);
          ++b;
        }
      }
    for (; b < kMaxOff; ++b) t.ent[r][b] = 21u;
  }
  return t;
}
constexpr ScanTab kScan = make_scan_tab();
__constant__ ScanTab kScanDev = make_scan_tab();
constexpr int kScanLutWords = 4 * 32;  // LDS copy of one row, 16 B per scan position: voxel offset in grid units (3 x f32), traversal rank o

// Every block copies its mode's row into LDS once: the scan cursor indexes it with a per-lane value (one
// ds_read instead of a 12-instruction register LUT or a scattered constant-memory load).
template <int NOFF>
__device__ __forceinline__ void fill_scan_lut(uint32_t * lut)
{
  constexpr int row = NOFF == 7 ? 1 : (NOFF == 19 ? 2 : 3);
  if (threadIdx.x < kMaxOff) {
    const uint32_t e = kScanDev.ent[row][threadIdx.x];
    const float q = static_cast<float>(1 << kQuantBits);
    lut[4 * threadIdx.x + 0] = __float_as_uint((static_cast<float>(e & 3u) - 1.f) * q);
    lut[4 * threadIdx.x + 1] = __float_as_uint((static_cast<float>((e >> 2) & 3u) - 1.f) * q);
    lut[4 * threadIdx.x + 2] = __float_as_uint((static_cast<float>((e >> 4) & 3u) - 1.f) * q);
    lut[4 * threadIdx.x + 3] = e >> 6;
  }
}

// Squared distance exactly as the reference's CPU build evaluates it: no FMA contraction (baseline
// x86-64), Eigen's SSE2 Vector4d reduction order (dx2 + dz2) + (dy2 + 0).  Keeps the k-NN selection
// bit-identical to gtsam_points::FlatContainer::knn_search, including near-ties.
__device__ __forceinline__ double sq_dist3(double dx, double dy, double dz)
{
#pragma clang fp contract(off)
  const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return (xx + zz) + yy;
}

// XCD-aware blockIdx -> chunk mapping (cdna_hip_programming.md T1): workgroup b runs on XCD b % 8, so
// XCD x takes the contiguous chunk range [x * cpx, (x + 1) * cpx).  Consecutive chunks of a scan (or
// of a voxel-ordered down-sampled cloud) are spatial neighbours that read the same buckets: they now
// share one 4 MiB L2 instead of pulling the same lines into all eight.  The grid is padded to a
// multiple of 8; chunks past the cloud simply find qi >= n.  Placement is a speed matter only.
[[maybe_unused]] __device__ __forceinline__ int xcd_chunk(int b, int n_blocks)
{
  const int cpx = n_blocks >> 3;  // launchers round the grid up to a multiple of 8
  return (b & 7) * cpx + (b >> 3);
}

// Wave-wide sums by DPP (wave_dpp.hpp; all 64 lanes must be active): the total lands in lane 63.
__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t x)
{
  uint32_t v[1] = {x};
  wave_sum_to_lane63_u32<1>(v);
  return v[0];
}
// (six __shfl_xor rounds on doubles are 12 ds_bpermute each — the LDS pipeline of the CU, shared by its waves: 2.1 us for K4's
// six sums, round 4)
__device__ __forceinline__ void wave_sum6_f64_to_lane63(double (&v)[6]) { wave_sum_to_lane63_f64<6>(v); }
__device__ __forceinline__ void wave_sum3_to_lane63(uint32_t (&v)[3]) { wave_sum_to_lane63_u32<3>(v); }
// Broadcast of one lane's value to the wave (l must be wave-uniform): v_readlane_b32 -> SGPR.
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int l)
{
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l));
}
__device__ __forceinline__ double lane_get(double v, int l)
{
  return __hiloint2double(static_cast<int>(lane_get(static_cast<uint32_t>(__double2hiint(v)), l)),
                          static_cast<int>(lane_get(static_cast<uint32_t>(__double2loint(v)), l)));
}

// Diagnostic build only (-DMH_TIMELINE): per-wave s_memtime stamps at phase boundaries.
#ifdef MH_TIMELINE
#define MH_STAMP(ptr, i)                                                                       \
  do {                                                                                         \
    if ((ptr) && (threadIdx.x & 63) == 0)                                                      \
      (ptr)[(static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (i)] = \
        __builtin_amdgcn_s_memtime();                                                          \
  } while (0)
#else
#define MH_STAMP(ptr, i) do { } while (0)
#endif
// K4's waves: indexed by the factor's workgroup index
#ifdef MH_TIMELINE
#define MH_STAMP4(ptr, i)                                                                                          \
  do {                                                                                                             \
    if ((ptr) && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < static_cast<unsigned int>(NW))                    \
      (ptr)[(static_cast<size_t>(block_id) * NW + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#else
#define MH_STAMP4(ptr, i) do { } while (0)
#endif
#define MH_STAMP_C2(ptr, i) MH_STAMP(ptr, i)

// 32-bit literals used per candidate, held in VGPRs: (a & lit) | lit needs two instructions with literal operands
// (one literal per VOP3), one v_and_or_b32 with register operands.
struct KeyConsts
{
  uint32_t ymask, ymagic, kmask, xmask, xmagic;
};

// Coarse key of a candidate: packed word w (3 x 10-bit voxel-relative coordinates), voxel offsets in grid units, payload
// (scan position << 5 | slot).  y is decoded by OR-ing its bit field (bits 10-19) into the mantissa of 2^13 (whose mantissa
// bit 10 weighs 1, ulp 2^-10): one v_and_or + a subtract instead of extract + convert + add; the caller folds the 2^13 into
// the per-voxel offset `mfy` (rounding <= 2^-11 grid units, inside the error budget).  x (bits 0-9) goes through the mantissa
// of 2^23 the same way, but 2^23 — whose ulp is one whole grid unit — is subtracted again exactly before the sub-grid offset
// is added (coarse_dist2 below).
// compare-exchange: a <- min, b <- max
__device__ __forceinline__ void cmp_exch(uint32_t & a, uint32_t & b)
{
  const uint32_t lo = min(a, b);
  b = max(a, b);
  a = lo;
}

// Four candidates (one 16-byte load) into the sorted top-KK.  The scan is VALU-bound while both waves of a SIMD are in it
// (~135 instructions per quad in round 3), so the arithmetic is written for instruction count:
//   keys   two candidates per packed-f32 instruction (v_pk_add / v_pk_mul / v_pk_fma_f32: same IEEE results as the scalar
//          forms, same operation order dy^2 -> + dx^2 -> + dz^2), validity = slot index below `vcnt` (valid slots of THIS
//          quad, 0..4, worked out once by the cursor)
//   KK == 8  sort the quad (5 compare-exchanges); half-clean it against the sorted top-8 and run the first merge stage in
//          one step: with ck[i] <= ck[4+i], min(ck[i], min(ck[4+i], q)) = min(ck[i], q) and max(ck[i], min(ck[4+i], q)) =
//          med3(ck[i], ck[4+i], q) — 2 instructions per pair instead of 3; then the two remaining bitonic stages: 34
//          min / max / med3 ops instead of 4 x 15 for four serial insertions.
//   other KK (generic k <= 8 path): serial insertion.
// mfy: y offset already folded with the 2^13 magic; pb: payload of the quad's slot 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f32x2 coarse_dist2(const KeyConsts & kc, uint32_t wa, uint32_t wb, float ofx, float mfy, float ofz)
{
  // x: its bit field (bits 0-9) OR-ed into the mantissa of 2^23 (ulp 1) and 2^23 subtracted again, exactly — one v_and_or
  // and half a packed subtract instead of mask + convert
  const f32x2 xm = {__uint_as_float((wa & kc.xmask) | kc.xmagic), __uint_as_float((wb & kc.xmask) | kc.xmagic)};
  const f32x2 x = xm - 8388608.0f;
  const f32x2 y = {__uint_as_float((wa & kc.ymask) | kc.ymagic), __uint_as_float((wb & kc.ymask) | kc.ymagic)};
  // z: bits 20-29; a stored word has bits 30-31 clear (map_kernels.hip), an unused slot's key is discarded whatever it decodes to
  const f32x2 z = {static_cast<float>(wa >> 20), static_cast<float>(wb >> 20)};
  const f32x2 dx = x + ofx, dy = y - mfy, dz = z + ofz;
  f32x2 d = dy * dy;
  d = __builtin_elementwise_fma(dx, dx, d);
  d = __builtin_elementwise_fma(dz, dz, d);
  return d;
}
template <int KK>
__device__ __forceinline__ void merge_quad(uint32_t (&ck)[KK], const KeyConsts & kc, const uint4 qw, float ofx, float mfy, float ofz,
                                           uint32_t pb, uint32_t vcnt)
{
  const f32x2 da = coarse_dist2(kc, qw.x, qw.y, ofx, mfy, ofz), db = coarse_dist2(kc, qw.z, qw.w, ofx, mfy, ofz);
  uint32_t k0 = (__float_as_uint(da.x) & kc.kmask) | pb;
  uint32_t k1 = (__float_as_uint(da.y) & kc.kmask) | (pb | 1u);
  uint32_t k2 = (__float_as_uint(db.x) & kc.kmask) | (pb | 2u);
  uint32_t k3 = (__float_as_uint(db.y) & kc.kmask) | (pb | 3u);
  k0 = vcnt > 0u ? k0 : 0xFFFFFFFFu;
  k1 = vcnt > 1u ? k1 : 0xFFFFFFFFu;
  k2 = vcnt > 2u ? k2 : 0xFFFFFFFFu;
  k3 = vcnt > 3u ? k3 : 0xFFFFFFFFu;
  if constexpr (KK == 8) {
    cmp_exch(k0, k1); cmp_exch(k2, k3); cmp_exch(k0, k2); cmp_exch(k1, k3); cmp_exch(k1, k2);
    const uint32_t q[4] = {k3, k2, k1, k0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t hi = med3_u32(ck[i], ck[4 + i], q[i]);
      ck[i] = min(ck[i], q[i]);
      ck[4 + i] = hi;
    }
    cmp_exch(ck[0], ck[2]); cmp_exch(ck[1], ck[3]); cmp_exch(ck[4], ck[6]); cmp_exch(ck[5], ck[7]);
    cmp_exch(ck[0], ck[1]); cmp_exch(ck[2], ck[3]); cmp_exch(ck[4], ck[5]); cmp_exch(ck[6], ck[7]);
  } else {
    const uint32_t kq[4] = {k0, k1, k2, k3};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t t = kq[c];
#pragma unroll
      for (int i = 0; i < KK - 1; ++i) {
        const uint32_t lo = min(ck[i], t);
        t = max(ck[i], t);
        ck[i] = lo;
      }
      ck[KK - 1] = min(ck[KK - 1], t);
    }
  }
}

// Register-only (voxel, quad) cursor of the neighbour scan.  `rem` = scan positions not entered yet; quad counts
// per voxel are packed 3 bits each in qc0 (positions 0..20) / qc1 (21..26).  advance() steps to the next quad —
// entering the next listed voxel when the current one is exhausted — and issues that quad's loads: the cell word
// from the lane's LDS list, the voxel offset from the LDS table, the 16-byte packed quad from the map.  There is no
// select on a load index (a `live ? index : 0` makes the compiler sink the LDS read into a branch: four dependent
// LDS round trips behind s_waitcnt lgkmcnt(0) per trip): every list entry names a mapped voxel and the quad index
// is clamped, so a dead stage just loads a quad it never uses.
struct ScanStage
{
  uint4 quad;     // four packed candidates
  float4 ofs;     // voxel offset in grid units (x, y, z)
  uint32_t pb;    // payload of the quad's slot 0: scan position << 5 | first slot
  uint32_t vcnt;  // valid slots of this quad: 0 (dead stage) .. 4
};
template <int NOFF>
struct ScanCursor
{
  uint32_t rem;
  int o_cur = 0;
  uint32_t qd = 8u, nq_cur = 0u;  // "exhausted": the first advance() enters the first listed voxel
  uint64_t qc0, qc1;

  __device__ __forceinline__ ScanStage advance(const uint32_t * list, int lds_stride, const float4 * lut4, const uint4 * qbuckets)
  {
    const bool sw = (qd >= nq_cur) && rem != 0u;  // next listed voxel (each holds >= 1 point)
    o_cur = sw ? __builtin_ctz(rem) : o_cur;
    rem = sw ? (rem & (rem - 1u)) : rem;
    qd = sw ? 0u : qd;
    if constexpr (NOFF <= 21)
      nq_cur = static_cast<uint32_t>(qc0 >> (3 * o_cur)) & 7u;
    else
      nq_cur = static_cast<uint32_t>((o_cur < 21 ? qc0 : qc1) >> (3 * (o_cur < 21 ? o_cur : o_cur - 21))) & 7u;
    const uint32_t e = list[o_cur * lds_stride];  // off the cursor's dependency chain
    ScanStage st;
    st.ofs = lut4[o_cur];
    uint32_t qidx = (e >> 5) * (kBucketStride / 4) + min(qd, static_cast<uint32_t>(kBucketStride / 4 - 1));
    st.quad = qbuckets[qidx];
    // valid slots: the voxel's count minus the slots before this quad, clamped to 0..4 — a quad index at or past the
    // voxel's last (an exhausted cursor keeps counting) gives 0 by itself: no liveness select
    const uint32_t q4 = qd << 2;
    st.vcnt = static_cast<uint32_t>(min(max(static_cast<int>(e & 31u) - static_cast<int>(q4), 0), 4));
    st.pb = (static_cast<uint32_t>(o_cur) << 5) | q4;
    ++qd;
    return st;
  }
};

// Mask of the neighbour voxels (scan positions 1..NOFF-1) that may still hold a top-k point: a voxel whose BOX is
// strictly farther from q than a proven upper bound of the current k-th distance cannot (so not even ties are
// affected).  boxd[] = squared box distances in grid units; the bound undoes the 10-bit key truncation (<= 2^-13
// relative on d^2) and adds the coarse-tier error err_g; no k-th key yet = keep everything.
template <int K, int KK, int NOFF>
__device__ __forceinline__ uint32_t prune_keep_mask(const uint32_t (&ck)[KK], const float (&boxd)[NOFF], int k, float err_g,
                                                    uint32_t kth_given = 0xFFFFFFFFu)
{
  uint32_t kth = kth_given;  // a k-th key proven elsewhere (the other lanes of the query's lane group)
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (i == k - 1) kth = min(kth, ck[i]);
  const float kv = __uint_as_float(kth & ~0x3FFu) * (1.0f + 2.5e-4f);
  const float r_up = sqrtf(kv) * (1.0f + 2e-6f) + err_g;
  const float b_up = kth != 0xFFFFFFFFu ? r_up * r_up : 3.0e38f;
  uint32_t keep = ~0u;
#pragma unroll
  for (int b = 1; b < NOFF; ++b) keep &= boxd[b] > b_up ? ~(1u << b) : ~0u;
  return keep;
}

// Squared distances (grid units) from q (qg = its position inside the centre voxel, grid units) to the boxes of the
// neighbour voxels in scan order; the gaps to the faces of the centre voxel are shrunk by a margin that also covers a
// stored point sitting ~1 ulp outside its nominal box.
template <int NOFF>
__device__ __forceinline__ void box_dists(const float qg0, const float qg1, const float qg2, float (&boxd)[NOFF])
{
  constexpr int row = NOFF == 7 ? 1 : (NOFF == 19 ? 2 : 3);
  constexpr float kQ = static_cast<float>(1 << kQuantBits);
  const float marg = 1e-2f;
  const float gxm = fmaxf(qg0 - marg, 0.f), gxp = fmaxf(kQ - qg0 - marg, 0.f);
  const float gym = fmaxf(qg1 - marg, 0.f), gyp = fmaxf(kQ - qg1 - marg, 0.f);
  const float gzm = fmaxf(qg2 - marg, 0.f), gzp = fmaxf(kQ - qg2 - marg, 0.f);
  const float g2x[3] = {gxm * gxm, 0.f, gxp * gxp}, g2y[3] = {gym * gym, 0.f, gyp * gyp}, g2z[3] = {gzm * gzm, 0.f, gzp * gzp};
#pragma unroll
  for (int b = 0; b < NOFF; ++b) {
    const uint32_t ent = kScan.ent[row][b];
    boxd[b] = g2x[ent & 3u] + g2y[(ent >> 2) & 3u] + g2z[(ent >> 4) & 3u];
  }
}

// One trip of the neighbour scan: the kPipe quads in flight are merged into the top-KK, each stage is refilled from the
// cursor first.  Returns the number of live quads this lane consumed.
template <int KK, int NOFF, int PIPE>
__device__ __forceinline__ uint32_t scan_trip(ScanCursor<NOFF> & cur, ScanStage (&stage)[PIPE], uint32_t (&ck)[KK], const KeyConsts & kc,
                                              const uint32_t * list, int lds_stride, const float4 * lut4, const uint4 * qbuckets,
                                              const float cx0, const float cy1, const float cz2, uint32_t & n_scanned)
{
  uint32_t live_quads = 0u;
#pragma unroll
  for (int u = 0; u < PIPE; ++u) {
    const ScanStage st = stage[u];
    stage[u] = cur.advance(list, lds_stride, lut4, qbuckets);  // refill this stage
    n_scanned += st.vcnt;
    live_quads += st.vcnt ? 1u : 0u;
    merge_quad<KK>(ck, kc, st.quad, st.ofs.x + cx0, cy1 - st.ofs.y, st.ofs.z + cz2, st.pb, st.vcnt);
  }
  return live_quads;
}

// ---- several lanes per query (the small-cloud classes of K3) -------------------------------------------------------------
// QL = 2 or 4 ADJACENT lanes of a wave serve one query: they look the neighbourhood up together (the same addresses: one
// transaction), share the quads of the centre voxel and the neighbour voxels left by the pruning among themselves, each keeps a
// sorted top-KK of what it scanned, and the lists are merged through DPP quad permutes — every lane of the group ends with the
// top-KK of the union, sorted, exactly the list one lane scanning everything would hold (the coarse keys are unique per
// candidate: distance bits | scan position | slot).  For clouds that leave most of the machine without a wave the scan is a
// fraction of one wave's dependent chain instead of all of it.
template <int CTRL>
__device__ __forceinline__ uint32_t quad_pull(uint32_t v)
{
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
// own sorted top-8 + the partner's (lane ^ 1: CTRL 0xB1, lane ^ 2: CTRL 0x4E) -> the sorted top-8 of both, in both lanes:
// min(a[i], b[7 - i]) are the 8 smallest of the 16 as a bitonic sequence; three compare-exchange stages sort it.
template <int CTRL>
__device__ __forceinline__ void quad_merge8(uint32_t (&ck)[8])
{
  uint32_t o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = quad_pull<CTRL>(ck[7 - i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) ck[i] = min(ck[i], o[i]);
  cmp_exch(ck[0], ck[4]); cmp_exch(ck[1], ck[5]); cmp_exch(ck[2], ck[6]); cmp_exch(ck[3], ck[7]);
  cmp_exch(ck[0], ck[2]); cmp_exch(ck[1], ck[3]); cmp_exch(ck[4], ck[6]); cmp_exch(ck[5], ck[7]);
  cmp_exch(ck[0], ck[1]); cmp_exch(ck[2], ck[3]); cmp_exch(ck[4], ck[5]); cmp_exch(ck[6], ck[7]);
}
template <int QL>
__device__ __forceinline__ void group_merge8(uint32_t (&ck)[8])
{
  if constexpr (QL >= 2) quad_merge8<0xB1>(ck);
  if constexpr (QL >= 4) quad_merge8<0x4E>(ck);
}
template <int QL>
__device__ __forceinline__ uint32_t group_min(uint32_t v)
{
  if constexpr (QL >= 2) v = min(v, quad_pull<0xB1>(v));
  if constexpr (QL >= 4) v = min(v, quad_pull<0x4E>(v));
  return v;
}
template <int QL>
__device__ __forceinline__ uint32_t group_or(uint32_t v)
{
  if constexpr (QL >= 2) v |= quad_pull<0xB1>(v);
  if constexpr (QL >= 4) v |= quad_pull<0x4E>(v);
  return v;
}
// The set bits of x dealt round-robin: lane `sub` of the group keeps the bits whose index among the set bits is = sub (mod QL).
// Inclusive prefix parity by five shift-xor steps: bit i of y = parity of popcount(x & (2 << i) - 1).
__device__ __forceinline__ uint32_t prefix_parity(uint32_t x)
{
  uint32_t y = x;
  y ^= y << 1;
  y ^= y << 2;
  y ^= y << 4;
  y ^= y << 8;
  y ^= y << 16;
  return y;
}
template <int QL>
__device__ __forceinline__ uint32_t deal_bits(uint32_t x, uint32_t sub)
{
  if constexpr (QL == 1) return x;
  // first, third, ... set bit: odd inclusive count
  const uint32_t odd = x & prefix_parity(x);
  uint32_t m = (sub & 1u) ? (x & ~odd) : odd;
  if constexpr (QL == 4) {
    const uint32_t odd2 = m & prefix_parity(m);
    m = (sub & 2u) ? (m & ~odd2) : odd2;
  }
  return m;
}

// k-NN of q over the neighbour voxels of its centre voxel (IncrementalVoxelMapPCL::knn_search).
// bi[0..k-1] = bucket indices of the k nearest in ascending order (0xFFFFFFFF where not found), dk = squared
// distance of the k-th.  `list` is this lane's column of an LDS array [NOFF][stride] (cell words in scan order),
// `scan_lut` the LDS table of fill_scan_lut.  Returns the number of points in all occupied neighbour voxels (what
// the reference scans); n_scanned = what was actually scanned.  All lanes of a wave that call must call together
// (the exact fallback is wave-cooperative); lanes that do not call are simply not helpers.
//
// The scan does less work per point than the reference's, in three exact steps:
//   prune   the centre voxel is scanned first; a neighbour voxel whose BOX is farther from q than a proven upper
//           bound of the current k-th distance cannot hold a top-k point and is skipped (strictly farther, so not
//           even ties are affected); neighbours are visited faces -> edges -> corners and the test is repeated on
//           the voxels not entered yet after the first two trips: ~83 -> ~38 candidates per query
//   coarse  every scanned candidate: f32 squared distance on the packed 10-bit copy -> 32-bit key whose low 10 bits
//           carry (scan position, bucket slot) -> sorted-quad bitonic merge into the top-KK (v_min/max_u32)
//   exact   the KK survivors: fp64 distance in the reference's operation order, ordered by (distance, traversal
//           rank) exactly like KnnResult::push
// plus a proof check: every scanned non-survivor's key is >= the KK-th key, so if that bound (minus the f32 error
// budget) exceeds the exact k-th distance no non-survivor can belong to the answer; otherwise the wave re-runs
// KnnResult::push for that lane over the scanned voxels (counted in n_exact_fallback).  Either way the selection is
// bit-identical to the reference.
// FAST (K3's k = 5 instantiation): the caller wants the k nearest POINTS, not their order — the exact tier ranks the survivors by
// counting (28 independent comparisons of (distance, traversal rank) instead of 8 dependent insertions) and hands back the
// survivors' coordinates with a membership mask: no sorted index list, no second load of the chosen points.  The SET is the
// one KnnResult::push selects (same strict order, ties by traversal rank), dk the same k-th distance.
template <int KK>
struct KnnPoints
{
  float4 pt[KK];    // the survivors of the coarse tier (w unused)
  uint32_t member;  // bit u: pt[u] is one of the k nearest
};
constexpr int knn_survivors(int K) { return K + 3 + (K > 5 ? 1 : 0); }  // 8 for k = 5, 12 for the generic k <= 8 path
// QL > 1 (FAST only): QL adjacent lanes call with the SAME query and the same `list` column; `n_scanned` is then the calling
// lane's own share (the caller sums the lanes), the return value and every result are the same in all lanes of the group, and the
// exact fallback is run for the group's first lane only (the others' results are the caller's to ignore).
template <int K, int NOFF, bool FAST = false, int PIPE = MH_PIPE, int QL = 1>
__device__ __forceinline__ uint32_t knn_query(const MapView & map, const double q0, const double q1, const double q2,
                                              int k, uint32_t * list, int lds_stride, const uint32_t * scan_lut,
                                              uint32_t (&bi)[K], double & dk, bool & fell_back, uint32_t & n_scanned,
                                              unsigned long long * dbg = nullptr, KnnPoints<knn_survivors(K)> * fast = nullptr)
{
  (void)dbg;
  (void)fast;
  static_assert(QL == 1 || (FAST && knn_survivors(K) == 8), "several lanes per query: the k = 5 fast path's top-8 only");
  [[maybe_unused]] const uint32_t sub = QL > 1 ? (threadIdx.x & static_cast<uint32_t>(QL - 1)) : 0u;
  fell_back = false;
  constexpr int KK = knn_survivors(K);
  constexpr int row = NOFF == 7 ? 1 : (NOFF == 19 ? 2 : 3);
  // ---- A. neighbourhood lookup ------------------------------------------------------------------
  // Block tables carry a one-voxel halo (voxel_map.hpp): all 27 neighbours of the centre voxel are in
  // the table of ITS block.  One hash probe, then nine 12-byte loads (the z-triple of each (dx, dy)
  // column) — 11 per-lane L1 transactions with the source point instead of 28.
  const double u0 = q0 * map.inv_leaf, u1 = q1 * map.inv_leaf, u2 = q2 * map.inv_leaf;  // voxel units
  const int cx = fast_floor(u0), cy = fast_floor(u1), cz = fast_floor(u2);
  constexpr int m = kBlockDim - 1;
  const int bx = cx >> kBlockLog2, by = cy >> kBlockLog2, bz = cz >> kBlockLog2;
  int blk_id;
  {
    // table entry = {key lo, key hi, block id, -}: 3 x 21-bit packed block coordinate; hi word -1 = empty slot
    const uint64_t key = pack_coord_key(bx, by, bz);
    const int klo = static_cast<int>(static_cast<uint32_t>(key)), khi = static_cast<int>(static_cast<uint32_t>(key >> 32));
    uint32_t h = block_hash(bx, by, bz) & map.mask;
    int4 e = map.table[h];
    while (e.y != -1 && !(e.x == klo && e.y == khi)) {  // collision: linear probe (load <= 0.5)
      h = (h + 1) & map.mask;
      e = map.table[h];
    }
    constexpr int kB = 1 << (kVoxCoordBits - 1 - kBlockLog2);  // block coordinates the key can hold: [-kB, kB)
    const bool in_range = ((static_cast<uint32_t>(bx + kB) | static_cast<uint32_t>(by + kB) | static_cast<uint32_t>(bz + kB)) >> (kVoxCoordBits - kBlockLog2)) == 0u;
    blk_id = (e.y != -1 && in_range) ? e.z : -1;
  }
  MH_STAMP(dbg, 8);
  uint32_t col[9][3];
  {
    // centre word of the 3x3x3 neighbourhood inside the halo'd table; block 0 when the block is absent
    // (the cells array always holds >= 1 table) + select below
    const uint32_t * tab = map.cells + static_cast<size_t>(blk_id < 0 ? 0 : blk_id) * kCellsPerBlock +
                           halo_index(cx & m, cy & m, cz & m);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const uint32_t * p = tab + ((c / 3 - 1) * kHaloDim + (c % 3 - 1)) * kHaloDim - 1;  // (dx, dy) column, dz = -1
      col[c][0] = p[0];
      col[c][1] = p[1];
      col[c][2] = p[2];
    }
  }
  uint32_t cell[NOFF];  // scan order: cell[0] is the centre voxel
#pragma unroll
  for (int b = 0; b < NOFF; ++b) {
    const uint32_t ent = kScan.ent[row][b];
    const int ox = static_cast<int>(ent & 3u), oy = static_cast<int>((ent >> 2) & 3u), oz = static_cast<int>((ent >> 4) & 3u);
    cell[b] = (blk_id >= 0 && static_cast<int>(ent >> 6) < map.n_off) ? col[ox * 3 + oy][oz] : kEmptyCell;
  }
#ifdef MH_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // attribute the cell-load latency to this stamp
#endif
  MH_STAMP(dbg, 9);
  // Every cell word goes to LDS slot o (no compaction); which voxels get scanned is a bit mask, and
  // their point counts are packed into registers (5 bits each, 12 per word) so the scan cursor never
  // reads memory.
  uint32_t amask = 0u, total_ref = 0u;
  uint64_t qc0 = 0, qc1 = 0;  // quads per voxel, 3 bits each: scan positions 0..20 | 21..26
#pragma unroll
  for (int o = 0; o < NOFF; ++o) {
    static_assert(kEmptyCell == 0u, "an empty cell reads as (voxel 0, count 0): a mapped index and no candidates, without a select");
    list[o * lds_stride] = cell[o];
    const uint32_t c = cell[o] & 31u;
    amask |= (c ? 1u : 0u) << o;
    total_ref += c;
    const uint64_t nq = (c + 3u) >> 2;
    if (o < 21)
      qc0 |= nq << (3 * o);
    else
      qc1 |= nq << (3 * (o - 21));
  }
  MH_STAMP(dbg, 1);

  // Coarse tier works in grid units g = leaf / 1024 relative to the centre voxel's origin, on the
  // packed 3 x 10-bit copy of the buckets: ONE 16-byte load brings four candidates (the kernel is bound
  // by per-lane L1 requests, not bytes).  A decoded coordinate is the middle of its quantisation cell:
  // |error| <= 0.5 g per axis, so |r_coarse - r| <= 0.87 g + f32 round-off; kErrG covers it.
  // q inside its voxel, in grid units: frac(q * inv_leaf) * 1024 — the expression the map's packed copy was quantised with
  // (map_kernels.hip), no division.
  constexpr float kErrG = 0.9f;
  constexpr double kQ_d = static_cast<double>(1 << kQuantBits);
  const float qg0 = static_cast<float>((u0 - static_cast<double>(cx)) * kQ_d);
  const float qg1 = static_cast<float>((u1 - static_cast<double>(cy)) * kQ_d);
  const float qg2 = static_cast<float>((u2 - static_cast<double>(cz)) * kQ_d);
  uint32_t ck[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) ck[i] = 0xFFFFFFFFu;
  KeyConsts kc{0xFFC00u, 0x46000000u, ~0x3FFu, 0x3FFu, 0x4B000000u};
  asm volatile("" : "+v"(kc.ymask), "+v"(kc.ymagic), "+v"(kc.kmask), "+v"(kc.xmask), "+v"(kc.xmagic));  // opaque: keeps them in registers

  // ---- B1. centre voxel first: it supplies the pruning bound -------------------------------------
  n_scanned = 0;
  if constexpr (QL == 1) {
    if (amask & 1u) {
      const uint32_t cc = cell[0] & 31u;
      const uint4 * b = map.qbuckets + static_cast<size_t>(cell[0] >> 5) * (kBucketStride / 4);
      n_scanned += cc;
      const float ofx = 0.5f - qg0, ofy = 0.5f - qg1, ofz = 0.5f - qg2;  // centre voxel: offset (0,0,0)
      uint4 qw[kBucketStride / 4];
#pragma unroll
      for (int u = 0; u < kBucketStride / 4; ++u) qw[u] = b[static_cast<uint32_t>(4 * u) < cc ? u : 0];  // all issued together
#pragma unroll
      for (int u = 0; u < kBucketStride / 4; ++u) {
        if (static_cast<uint32_t>(4 * u) < cc) merge_quad<KK>(ck, kc, qw[u], ofx, 8192.0f - ofy, ofz, static_cast<uint32_t>(4 * u), min(cc - static_cast<uint32_t>(4 * u), 4u));
      }
    }
  } else {
    // the centre voxel's quads dealt to the group: lane `sub` takes quads sub, sub + QL, ... — every lane runs the same
    // ceil(5 / QL) merges (a quad past the voxel's last has no valid slot), then the lists are merged: every lane holds the
    // centre voxel's top-8 and with it the first pruning bound.  (An empty centre voxel reads quad 0 of voxel 0: mapped.)
    const uint32_t cc = cell[0] & 31u;
    const uint4 * b = map.qbuckets + static_cast<size_t>(cell[0] >> 5) * (kBucketStride / 4);
    const float ofx = 0.5f - qg0, ofy = 0.5f - qg1, ofz = 0.5f - qg2;
    constexpr int NJ = (kBucketStride / 4 + QL - 1) / QL;
    uint4 qw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) qw[j] = b[min(static_cast<uint32_t>(j * QL) + sub, static_cast<uint32_t>(kBucketStride / 4 - 1))];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint32_t u4 = 4u * (static_cast<uint32_t>(j * QL) + sub);
      const uint32_t vc = static_cast<uint32_t>(min(max(static_cast<int>(cc) - static_cast<int>(u4), 0), 4));
      n_scanned += vc;
      merge_quad<KK>(ck, kc, qw[j], ofx, 8192.0f - ofy, ofz, min(u4, 28u), vc);
    }
    group_merge8<QL>(ck);
  }
  MH_STAMP(dbg, 10);
  // ---- prune: a neighbour voxel whose BOX is farther from q than a proven upper bound of the current
  // k-th distance cannot hold a top-k point (strictly farther, so not even ties are affected).  The
  // squared box distances stay in registers: the scan re-applies the test as its bound tightens.
  float boxd[NOFF];
  box_dists<NOFF>(qg0, qg1, qg2, boxd);
  uint32_t rem = amask & ~1u;  // neighbour voxels the cursor has not entered yet
  rem &= prune_keep_mask<K, KK, NOFF>(ck, boxd, k, kErrG);
  // several lanes per query: the voxels that are left are dealt round-robin in scan order (faces, edges, corners: every lane
  // gets near and far ones); the group's first lane goes on with the centre voxel's list, the others start empty and prune
  // with the bound it proves (kth_group) until their own is tighter
  [[maybe_unused]] uint32_t kth_group = 0xFFFFFFFFu;
  if constexpr (QL > 1) {
#pragma unroll
    for (int i = 0; i < K; ++i)
      if (i == k - 1) kth_group = ck[i];
    // (Measured and not kept, 24 576 points, rocprofv3: dealing by weight — every voxel to the lane with fewer quads so far — 16.7 us
    // against 15.7; 2 or 3 quads in flight per lane instead of 4: 16.3 / 16.5 us.  What the dealing leaves unbalanced costs less
    // than the instructions that would balance it.)
    rem = deal_bits<QL>(rem, sub);
    if (sub != 0u) {
#pragma unroll
      for (int i = 0; i < KK; ++i) ck[i] = 0xFFFFFFFFu;
    }
  }
  uint32_t alive = rem;  // neighbour voxels never pruned: alive & ~rem (after the scan) = the voxels scanned

  MH_STAMP(dbg, 11);
  // ---- B2. remaining voxels: flattened, software-pipelined coarse scan over QUADS of candidates ---
  // The (voxel, quad) cursor walks the not-yet-entered mask in registers kPipe quads ahead of the
  // arithmetic; every load is unconditional (quad 0 of voxel 0 past the end), validity is a select.  The
  // wave loops while any lane still has a live quad in flight; after the first and second trip the
  // tightened bound prunes the voxels not entered yet (what matters for lanes whose centre voxel held
  // fewer than k points: their first bound is infinite).
  {
    constexpr int kPipe = PIPE;
    const float4 * lut4 = reinterpret_cast<const float4 *>(scan_lut);
    const float cx0 = 0.5f - qg0, cy1 = 8192.0f - 0.5f + qg1, cz2 = 0.5f - qg2;
    ScanCursor<NOFF> cur;
    cur.rem = rem;
    cur.qc0 = qc0;
    cur.qc1 = qc1;
    ScanStage stage[kPipe];
#pragma unroll
    for (int u = 0; u < kPipe; ++u) stage[u] = cur.advance(list, lds_stride, lut4, map.qbuckets);
    for (int trip = 0;; ++trip) {
      if (!__any(static_cast<int>(stage[0].vcnt))) break;  // a dead stage 0 means dead stages 1..3
      if (MH_PRUNE_TRIPS) {
        if constexpr (QL > 1) {
          // the tightest k-th key any lane of the group has proven (each is an upper bound of the answer's k-th distance).
          // (Merging the lists here instead, so that every lane prunes with the k-th key of everything the group has scanned:
          // measured, the same candidates scanned — 39.9 per query either way — and 0.2 us slower at 24 576 points.)
          uint32_t mine = kth_group;
#pragma unroll
          for (int i = 0; i < K; ++i)
            if (i == k - 1) mine = min(mine, ck[i]);
          kth_group = group_min<QL>(mine);
        }
        const uint32_t keep = prune_keep_mask<K, KK, NOFF>(ck, boxd, k, kErrG, kth_group);
        cur.rem &= keep;
        alive &= keep;
      }
      scan_trip<KK, NOFF, kPipe>(cur, stage, ck, kc, list, lds_stride, lut4, map.qbuckets, cx0, cy1, cz2, n_scanned);
    }
    rem = cur.rem;
  }
  if constexpr (QL > 1) {
    group_merge8<QL>(ck);                                  // every lane: the sorted top-8 of everything the group scanned
    alive = group_or<QL>(alive & ~rem);                    // ... and the neighbour voxels the group entered
    rem = 0u;
  }
  // (Measured, round 1: capping the per-lane scan at 16 quads and letting the 64 lanes scan the leftover
  // voxels together, one candidate per lane, was 2x SLOWER — each (lane, voxel) step is a dependent LDS ->
  // HBM -> ballot chain with nothing to overlap it.  The per-lane pipelined cursor stays.)
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t act = __ballot(1);
  const uint32_t nact = static_cast<uint32_t>(__popcll(act));
  const uint32_t rank = static_cast<uint32_t>(__popcll(act & ((1ull << lane) - 1ull)));
  // centre + every neighbour voxel the cursor entered
  const uint32_t scanned_mask = (amask & 1u) | (alive & ~rem);
  MH_STAMP(dbg, 2);

  // ---- exact tier: re-rank the survivors in fp64 by (distance, traversal rank) -------------------
  double bd[K];
  uint32_t br[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    bd[i] = kDblMax;
    bi[i] = 0xFFFFFFFFu;
    br[i] = 0xFFFFFFFFu;
  }
  {
    uint32_t sidx[KK], srank[KK];
    float4 sc[KK];
#pragma unroll
    for (int u = 0; u < KK; ++u) {  // all survivor loads issued together (index 0 for empty slots)
      const uint32_t p = ck[u] & 0x3FFu;
      const int b = min(static_cast<int>(p >> 5), NOFF - 1);  // scan position
      const uint32_t e = list[b * lds_stride];
      sidx[u] = (e >> 5) * kBucketStride + (p & 31u);  // empty survivor slot: slot 31 of a listed voxel (mapped, unused)
      sc[u] = map.buckets[sidx[u]];
      srank[u] = (scan_lut[4 * b + 3] << 5) | (p & 31u);  // (traversal rank of the voxel, slot): the order push() sees
    }
    if constexpr (FAST) {
      // rank by counting: survivor u's position in the order (distance, traversal rank) = the number of survivors before it.
      // An empty slot carries the largest double: it sorts behind every candidate, so no validity mask enters the comparisons.
      // A squared distance is a non-negative double, so its BIT PATTERN orders like its value: one 64-bit integer compare
      // per pair.  Exact ties are broken by the traversal rank only where it matters — a tie across the boundary of the answer
      // (the k-th and the (k+1)-th distance equal) — in a second, wave-uniform pass that almost no wave takes: a tie inside
      // the answer or outside it changes neither the set nor the k-th distance.
      unsigned long long d[KK];
      uint32_t rk[KK];
#pragma unroll
      for (int u = 0; u < KK; ++u) {
        const double du = sq_dist3(static_cast<double>(sc[u].x) - q0, static_cast<double>(sc[u].y) - q1, static_cast<double>(sc[u].z) - q2);
        d[u] = static_cast<unsigned long long>(__double_as_longlong(ck[u] != 0xFFFFFFFFu ? du : kDblMax));
        rk[u] = 0u;
      }
#pragma unroll
      for (int u = 0; u < KK; ++u)
#pragma unroll
        for (int v = u + 1; v < KK; ++v) {
          const bool lt = d[u] < d[v];  // u before v
          rk[v] += lt ? 1u : 0u;
          rk[u] += lt ? 0u : 1u;
        }
      unsigned long long dk_bits = static_cast<unsigned long long>(__double_as_longlong(kDblMax)), dn_bits = ~0ull;
#pragma unroll
      for (int u = 0; u < KK; ++u) {
        dk_bits = rk[u] == static_cast<uint32_t>(k - 1) ? d[u] : dk_bits;  // the k-th of the order (kDblMax: fewer than k candidates)
        dn_bits = rk[u] == static_cast<uint32_t>(k) ? d[u] : dn_bits;      // the (k+1)-th
      }
      uint32_t member = 0u;
#pragma unroll
      for (int u = 0; u < KK; ++u) {
        const bool valid = ck[u] != 0xFFFFFFFFu;
        member |= (valid && rk[u] < static_cast<uint32_t>(k)) ? (1u << u) : 0u;
        fast->pt[u] = sc[u];
      }
      if (__any(static_cast<int>(dk_bits == dn_bits && dk_bits != static_cast<unsigned long long>(__double_as_longlong(kDblMax))))) {
        // an exact tie across the boundary somewhere in this wave: the full order (distance, traversal rank) for everybody
        uint32_t rt[KK];
#pragma unroll
        for (int u = 0; u < KK; ++u) rt[u] = 0u;
#pragma unroll
        for (int u = 0; u < KK; ++u)
#pragma unroll
          for (int v = u + 1; v < KK; ++v) {
            const bool lt = d[u] < d[v] || (d[u] == d[v] && srank[u] < srank[v]);
            rt[v] += lt ? 1u : 0u;
            rt[u] += lt ? 0u : 1u;
          }
        uint32_t mt = 0u;
#pragma unroll
        for (int u = 0; u < KK; ++u) mt |= (ck[u] != 0xFFFFFFFFu && rt[u] < static_cast<uint32_t>(k)) ? (1u << u) : 0u;
        member = mt;
      }
      const double dkf = __longlong_as_double(static_cast<long long>(dk_bits));  // (the same value whichever tied survivor holds rank k - 1)
      fast->member = member;
      bd[K - 1] = dkf;  // (read back below as the k-th distance: K == k in this instantiation)
    } else {
#pragma unroll
    for (int u = 0; u < KK; ++u) {
      const uint32_t p = srank[u];
      const double d = sq_dist3(static_cast<double>(sc[u].x) - q0, static_cast<double>(sc[u].y) - q1,
                                static_cast<double>(sc[u].z) - q2);
      if (ck[u] != 0xFFFFFFFFu && (d < bd[K - 1] || (d == bd[K - 1] && p < br[K - 1]))) {
        bd[K - 1] = d;
        bi[K - 1] = sidx[u];
        br[K - 1] = p;
#pragma unroll
        for (int i = K - 1; i > 0; --i) {
          if (bd[i] < bd[i - 1] || (bd[i] == bd[i - 1] && br[i] < br[i - 1])) {
            const double td = bd[i];
            bd[i] = bd[i - 1];
            bd[i - 1] = td;
            const uint32_t ti = bi[i];
            bi[i] = bi[i - 1];
            bi[i - 1] = ti;
            const uint32_t tr = br[i];
            br[i] = br[i - 1];
            br[i - 1] = tr;
          }
        }
      }
    }
    }
  }
  MH_STAMP(dbg, 12);
  dk = kDblMax;
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (i == k - 1) dk = bd[i];
  // ---- proof check (only meaningful when there ARE non-survivors: the KK-th slot is filled) ------
  bool need_exact = false;
  if (ck[KK - 1] != 0xFFFFFFFFu && dk < kDblMax) {
    // Scanned non-survivors have coarse keys >= ck[KK-1]; clearing the payload bits only lowers the
    // bound; |r_coarse - r| <= kErrG grid units.  (Pruned voxels are farther than the k-th distance
    // by construction.)  Evaluated in f32 grid units with the roundings on the safe side: a lane flagged without need only
    // takes the exact pass, which returns the same answer.
    const float c8 = __uint_as_float(ck[KK - 1] & ~0x3FFu);                       // grid units^2, a lower bound of every non-survivor's key
    const float r_lo = sqrtf(c8) * (1.0f - 2e-6f) - kErrG;                         // grid units
    const double inv_g = map.inv_leaf * kQ_d;                                      // grid units per metre
    const float dk_g = static_cast<float>(dk * (inv_g * inv_g)) * (1.0f + 3e-7f);  // the exact k-th distance, rounded up
    need_exact = !(r_lo > 0.0f && dk_g < r_lo * r_lo * (1.0f - 3e-7f));
  }
  // ---- exact fallback, wave-cooperative: KnnResult::push verbatim for lane L — the scanned voxels in
  // traversal (offset-generation) order, slots in order, strict '<' so the earlier candidate wins ties —
  // with the 64 lanes computing one fp64 distance each and L inserting the few that beat its k-th.
  // (Pruned voxels are provably farther than the k-th distance, so skipping them changes nothing.)
  if constexpr (QL > 1) need_exact = need_exact && sub == 0u;  // one lane of the group answers for the query
  fell_back = need_exact;
  uint64_t fb = __ballot(need_exact);
  while (fb) {
    const int L = __builtin_ctzll(fb);
    fb &= fb - 1ull;
    const uint32_t mL = lane_get(scanned_mask, L);
    const double y0 = lane_get(q0, L), y1 = lane_get(q1, L), y2 = lane_get(q2, L);
    const int dl = QL > 1 ? L / QL - static_cast<int>(lane) / QL : L - static_cast<int>(lane);  // lane L's list column from this lane's
    double ed[K];
    uint32_t ei[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      ed[i] = kDblMax;
      ei[i] = 0xFFFFFFFFu;
    }
    double dlast = kDblMax;  // L's current k-th distance, wave-uniform
    for (int o = 0; o < NOFF; ++o) {
      const int b = static_cast<int>(kScanDev.pos[row][o]);
      if (!((mL >> b) & 1u)) continue;
      const uint32_t e = list[b * lds_stride + dl];
      const uint32_t cnt = e & 31u, vbase = (e >> 5) * kBucketStride;
      for (uint32_t j0 = 0; j0 < cnt; j0 += nact) {
        const uint32_t j = j0 + rank;
        const float4 c = map.buckets[vbase + min(j, static_cast<uint32_t>(kBucketStride - 1))];
        const double d = sq_dist3(static_cast<double>(c.x) - y0, static_cast<double>(c.y) - y1, static_cast<double>(c.z) - y2);
        uint64_t hm = __ballot(j < cnt && d < dlast);
        while (hm) {  // ascending lane = ascending slot: the order push() sees
          const int src = __builtin_ctzll(hm);
          hm &= hm - 1ull;
          const double ds = lane_get(d, src);
          const uint32_t is = vbase + lane_get(j, src);
          if (static_cast<int>(lane) == L && ds < ed[K - 1]) {
            ed[K - 1] = ds;
            ei[K - 1] = is;
#pragma unroll
            for (int i = K - 1; i > 0; --i) {
              if (ed[i] < ed[i - 1]) {
                const double td = ed[i];
                ed[i] = ed[i - 1];
                ed[i - 1] = td;
                const uint32_t ti = ei[i];
                ei[i] = ei[i - 1];
                ei[i - 1] = ti;
              }
            }
          }
        }
        double kth = kDblMax;  // found < k keeps the bound open, like an unfilled KnnResult
#pragma unroll
        for (int i = 0; i < K; ++i)
          if (i == k - 1) kth = ed[i];
        dlast = lane_get(kth, L);
      }
    }
    if (static_cast<int>(lane) == L) {
      dk = kDblMax;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        bi[i] = ei[i];
        if (i == k - 1) dk = ed[i];
      }
      if constexpr (FAST) {  // this lane's points come from the exact pass: the first k slots of the hand-over
        uint32_t member = 0u;
#pragma unroll
        for (int i = 0; i < K; ++i)
          if (i < k && ei[i] != 0xFFFFFFFFu) {
            fast->pt[i] = map.buckets[ei[i]];
            member |= 1u << i;
          }
        fast->member = member;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (i >= k) bi[i] = 0xFFFFFFFFu;
  return total_ref;
}

// null vector of M = A - lambda I (lambda ~ an eigenvalue): the largest of the three cross products of its rows
__device__ __forceinline__ void null_vector3(const double a00, const double a01, const double a02, const double a11, const double a12,
                                             const double a22, const double lambda, double (&v)[3])
{
  const double m00 = a00 - lambda, m11 = a11 - lambda, m22 = a22 - lambda;
  const double x0 = a01 * a12 - a02 * m11, y0 = a02 * a01 - m00 * a12, z0 = m00 * m11 - a01 * a01;  // r0 x r1
  const double x1 = a01 * m22 - a02 * a12, y1 = a02 * a02 - m00 * m22, z1 = m00 * a12 - a01 * a02;  // r0 x r2
  const double x2 = m11 * m22 - a12 * a12, y2 = a12 * a02 - a01 * m22, z2 = a01 * a12 - m11 * a02;  // r1 x r2
  const double n0 = x0 * x0 + y0 * y0 + z0 * z0, n1 = x1 * x1 + y1 * y1 + z1 * z1, n2 = x2 * x2 + y2 * y2 + z2 * z2;
  double vx = x0, vy = y0, vz = z0, nn = n0;
  if (n1 > nn) {
    vx = x1;
    vy = y1;
    vz = z1;
    nn = n1;
  }
  if (n2 > nn) {
    vx = x2;
    vy = y2;
    vz = z2;
    nn = n2;
  }
  if (nn > 0.0) {
    const double inv = mh_rsqrt(nn);
    v[0] = vx * inv;
    v[1] = vy * inv;
    v[2] = vz * inv;
  } else {
    v[0] = 1.0;
    v[1] = 0.0;
    v[2] = 0.0;
  }
}

// Eigen-decomposition of a symmetric PSD 3x3 (covariance of k points) for the plane fit:
// eigenvalues ascending + unit eigenvector of the smallest.  Replaces the
// Eigen::SelfAdjointEigenSolver call of estimatePlane (geometric_factor.hpp:196), which only consumes
// the three eigenvalues and eigenvector 0 (:202-215).
//   w0: Newton on the characteristic cubic p(x) = x^3 - c2 x^2 + c1 x - c0 from x = 0.  For a PSD
//       matrix p is increasing and concave on [0, w0], so the iterates rise monotonically to the
//       smallest root with no overshoot; convergence is quadratic (<= 6 steps from 0 in fp64).
//   w1, w2: the deflated quadratic.  v0: best-conditioned cross product of two rows of A - w0 I.
// fp64 throughout; |dw| ~ eps |A|, far inside the 1e-5 parity bar and the plane gates' margins.
// (The trigonometric closed form costs three fp64 transcendental calls per point: 2x this.)
__device__ __forceinline__ void plane_eigen(const double a00, const double a01, const double a02, const double a11,
                                            const double a12, const double a22, double (&w)[3], double (&v)[3])
{
  const double c2 = a00 + a11 + a22;
  const double c1 = (a00 * a11 - a01 * a01) + (a00 * a22 - a02 * a02) + (a11 * a22 - a12 * a12);
  const double c0 = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
  double x = 0.0;
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    const double p = ((x - c2) * x + c1) * x - c0;
    const double dp = (3.0 * x - 2.0 * c2) * x + c1;
    if (!(dp > 0.0)) break;
    const double step = p * mh_rcp1(dp);  // (a Newton step need not be divided exactly: the fixed point is p(x) = 0 either way)
    x -= step;
    if (fabs(step) <= 1e-16 * fabs(x)) break;
  }
  double w0 = x;
  null_vector3(a00, a01, a02, a11, a12, a22, w0, v);
  {
    // Two close smallest eigenvalues (a neighbourhood as thick in one in-plane direction as across the plane; about one
    // query in a thousand has them within 25 %).  Newton on the characteristic polynomial converges only linearly until it
    // has resolved the pair (8 steps do not), and even a converged root is good to ~eps w0 w1 / gap only, an error the
    // eigenvector inherits divided by the gap once more: 5 degrees off in the fuzz case that found this (gap 1.2 % of w1).
    // So, for those lanes only: finish the Newton iteration (from below it can only reach the smallest root), then
    // Rayleigh refinement on the MATRIX — lambda = v'Av is second-order accurate in v, so every round squares the error —
    // which ends at the eps |A| / gap an iterative solver like Eigen's delivers.  The usual lane skips all of it.
    // "w1 - w0 < 25 % of w1", i.e. w1 < (4/3) w0, without extracting w1: w1, w2 are the roots of t^2 - S0 t + P0, so the test
    // point t = (4/3) w0 lies at or above w1 iff the quadratic is <= 0 there or t is already past the roots' midpoint
    const double S0 = c2 - w0, P0 = c1 - w0 * S0, tq = (4.0 / 3.0) * w0;
    if (tq >= 0.5 * S0 || (tq - S0) * tq + P0 <= 0.0) {
      double prev = kDblMax;
#pragma unroll 1
      for (int it = 0; it < 56; ++it) {
        const double p = ((x - c2) * x + c1) * x - c0;
        const double dp = (3.0 * x - 2.0 * c2) * x + c1;
        if (!(dp > 0.0)) break;
        const double step = p * mh_rcp1(dp);
        if (!(fabs(step) < 0.75 * prev)) break;  // steps halve while the pair is unresolved, then collapse; noise does neither
        x -= step;
        prev = fabs(step);
        if (fabs(step) <= 1e-16 * fabs(x)) break;
      }
      w0 = x;
      null_vector3(a00, a01, a02, a11, a12, a22, w0, v);
#pragma unroll 1
      for (int r = 0; r < 3; ++r) {
        const double Av0 = a00 * v[0] + a01 * v[1] + a02 * v[2], Av1 = a01 * v[0] + a11 * v[1] + a12 * v[2],
                     Av2 = a02 * v[0] + a12 * v[1] + a22 * v[2];
        w0 = v[0] * Av0 + v[1] * Av1 + v[2] * Av2;
        null_vector3(a00, a01, a02, a11, a12, a22, w0, v);
      }
    }
  }
  const double S = c2 - w0;               // w1 + w2
  const double P = c1 - w0 * S;           // w1 * w2
  const double disc = sqrt(fmax(S * S - 4.0 * P, 0.0));
  w[0] = w0;
  w[1] = 0.5 * (S - disc);
  w[2] = 0.5 * (S + disc);
}

// Inter-workgroup hand-off of the per-block partial rows (cdna_hip_programming.md §6 G16,
// MI355X_MICROARCH.md "valid forms"): 8-byte agent-scope atomics on BOTH sides — write-through (sc1)
// producer stores drained with s_waitcnt vmcnt(0) before the ticket; the last arriver reads with
// agent-scope loads.  No release fence (a per-block buffer_wbl2 costs microseconds).
__device__ __forceinline__ void store_partial(double * p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_partial(const double * p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One double as a flagged word (icp_device.hpp): a single 16-byte store into mapped pinned memory.
__device__ __forceinline__ void ll_store(uint4 * p, double v, unsigned int seq)
{
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
  *p = make_uint4(static_cast<unsigned int>(b), seq, static_cast<unsigned int>(b >> 32), seq);
}

__device__ __forceinline__ bool arrive_is_last(unsigned int * ticket, unsigned int n_blocks, bool * s_last)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its own write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (prev == n_blocks - 1);
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    *s_last = last;
  }
  __syncthreads();
  return *s_last;
}

// Deterministic parallel fold of the per-block partial rows by the last-arriving block: thread
// (entry, lane-segment) sums blocks seg, seg+NSEG, ... with four independent accumulators (loads
// stay in flight), segments are then combined in index order.  Result in s_out[0..n_ent).
template <int EW, int TPB, bool PLAIN = false, int BATCH = 16>  // entries rounded up: 32 (unary / K4) or 96 (binary); TPB threads per workgroup
__device__ __forceinline__ void fold_rows(const double * partials, int n_blocks, int n_ent, double * s_seg,
                                          double * s_out)
{  // PLAIN: the rows were written by an EARLIER kernel (plain stores, visible at the kernel boundary): ordinary cached loads
  // The segmentation — and with it the order of the additions — is that of a 256-thread workgroup whatever TPB is: the same
  // rows fold to the same bits in K3's own last block, in a single call's K4 and in a window batch's.
  constexpr int NSEG = (TPB < 256 ? TPB : 256) / EW;
  const int ent = threadIdx.x % EW, seg = threadIdx.x / EW;
  if (seg < NSEG && ent < n_ent) {
    // all of this thread's rows are requested before the first one is consumed: ONE memory round trip for
    // the usual <= 16 rows per thread (the fold is the serial tail of the kernel; four dependent rounds
    // of write-through loads cost ~4 us)
    constexpr int kBatch = BATCH;  // (32: K4's 256 rows over 8 segments in one round trip, where the registers are there)
    double acc = 0.0;
    for (int b0 = seg; b0 < n_blocks; b0 += kBatch * NSEG) {
      double v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int b = b0 + u * NSEG;
        const double * src = partials + static_cast<size_t>(b < n_blocks ? b : seg) * kPartialStride + ent;
        v[u] = PLAIN ? *src : load_partial(src);
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) acc += (b0 + u * NSEG < n_blocks) ? v[u] : 0.0;  // fixed order: deterministic
    }
    s_seg[seg * EW + ent] = acc;
  } else if (seg < NSEG) {
    s_seg[seg * EW + ent] = 0.0;
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < n_ent) {
    double s = 0.0;
    for (int g = 0; g < NSEG; ++g) s += s_seg[g * EW + threadIdx.x];
    s_out[threadIdx.x] = s;
  }
  __syncthreads();
}

}  // namespace

// Unwhitened, normalised Jacobian directions of one point, as the component localizabilities project them
// (geometric_factor.hpp:343-352, :434-457): n_s = R^T n, jr = (n_s x p) normalised (Eigen normalized(): unchanged when the
// squared norm is zero), jt = -n_s.  One body for K3 (plain factors: it writes them into the call's record) and K4 (map-sharded
// and two-phase callers: from the stored normals), so both produce the same bits.
template <typename A>
__device__ __forceinline__ void loc_directions(const A & a, const double nx, const double ny, const double nz, const double px,
                                               const double py, const double pz, double (&jr)[3], double (&jt)[3])
{
  const double ns0 = a.R[0] * nx + (a.R[3] * ny + a.R[6] * nz);
  const double ns1 = a.R[1] * nx + (a.R[4] * ny + a.R[7] * nz);
  const double ns2 = a.R[2] * nx + (a.R[5] * ny + a.R[8] * nz);
  double r0 = ns1 * pz - ns2 * py, r1 = ns2 * px - ns0 * pz, r2 = ns0 * py - ns1 * px;
  const double nr2 = r0 * r0 + (r1 * r1 + r2 * r2);
  if (nr2 > 0.0) {
    const double inv = mh_rsqrt(nr2);
    r0 *= inv;
    r1 *= inv;
    r2 *= inv;
  }
  jr[0] = r0;
  jr[1] = r1;
  jr[2] = r2;
  jt[0] = -ns0;
  jt[1] = -ns1;
  jt[2] = -ns2;
}

// ------------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------------
// TPB = threads per workgroup: 512 for big clouds (one workgroup per CU at 131 072 points), 256 for clouds of up to
// 65 536 points — the down-sampled clouds the reference feeds the factor are 10-25 k points, and at 512 threads they
// would occupy a fifth of the CUs with two waves per SIMD; at 256 every wave has a SIMD to itself.
// block_id / n_blocks: this workgroup's index within ITS factor's (multiple-of-8) grid: the launch grid for the
// single-factor kernel, the factor's segment of the launch grid for the batched one.
// SHARD: the launch of a map-sharded factor (shard_api.hip) — the slot count is read from the device, slots whose status
// carries kShardSkip are passed over, the last block also writes its sums into the all-reduce vector.  The plain factor's
// instantiation carries none of it (round 3 had it in the one kernel: +1 us on every unsharded launch).
// QL: lanes per query (1; 2 or 4 for small clouds of plain k = 5 factors — "several lanes per query" above): the workgroup's
// TPB threads then serve TPB / QL consecutive points, the group's first lane owns the point (stores, row, counters), and the
// per-point results are bit-identical to the QL = 1 classes'.
template <int K, bool BINARY, int NOFF, int TPB, bool SHARD, int QL = 1>
__device__ __forceinline__ void icp_linearize_body(const IcpArgs & a, const int block_id, const int n_blocks)
{
  static_assert(QL == 1 || (!SHARD && K == 5), "several lanes per query: plain k = 5 factors only");
  constexpr int PPW = TPB / QL;  // points per workgroup
  [[maybe_unused]] const bool lead = QL == 1 || (threadIdx.x & static_cast<unsigned int>(QL - 1)) == 0u;
  constexpr int NV = BINARY ? 13 : 7;           // row vector v = [J_s(6) (, J_t(6)), e]
  constexpr int NENT = NV * (NV + 1) / 2;       // upper triangle of v v^T: 28 / 91 sums
  constexpr int EW = BINARY ? 96 : 32;
  constexpr int ROWW = NV;                      // 7 / 13 doubles: an odd row stride spreads the lanes' rows over the LDS banks
  // One LDS arena, reused: [k-NN] per-lane neighbour cell words; [last block] fold scratch.  The waves' row tiles and partial
  // sums (step 8) have memory of their own: a wave reduces its rows while the others are still in the k-NN arena.
  constexpr int kListWords = NOFF * PPW;
  constexpr int kFoldWords = (TPB / EW) * EW * 2 + EW * 2;
  constexpr int kArenaWords = kListWords > kFoldWords ? kListWords : kFoldWords;
  __shared__ __attribute__((aligned(16))) uint32_t s_arena[kArenaWords];
  __shared__ double s_tile[TPB * ROWW];                               // [wave][64 rows][ROWW]
  __shared__ double s_wsum[(TPB / 64) * (NENT <= 32 ? 2 : 1) * NENT];  // [wave][point segment][entry]
  __shared__ unsigned int s_cnt[4];  // n_knn, n_cand, exact-fallback count, candidates actually scanned
  __shared__ uint32_t s_scan[kScanLutWords];
  __shared__ bool s_last;

  uint32_t * s_list = s_arena + threadIdx.x / QL;                                     // [NOFF][PPW]: one column per point
  double * s_aux = reinterpret_cast<double *>(s_arena);                               // fold scratch of the last block

  // Which 64 points a wave takes.  A workgroup's waves take 64-point chunks that lie n_blocks / 8 chunks apart inside their XCD's
  // stretch of the (Morton-ordered) cloud instead of consecutive ones: the cost of a chunk varies slowly along the curve (a
  // dense corner is several thousand points long), there are exactly as many workgroups as CUs, and the kernel ends with its
  // slowest workgroup — eight consecutive chunks put a heavy stretch on ONE CU's eight waves, eight strided ones give every CU
  // a sample of its XCD's whole stretch (same L2 as before).  rocprofv3, cold calls on the configs[1] world: 32.2 -> 29.2 us at
  // 131 072 points, 21.3 -> 20.9 us at 24 576 (gpurun c19; -DMH_NO_INTERLEAVE restores the consecutive chunks).
#if !defined(MH_NO_INTERLEAVE)
  const int qi = [&] {
    if constexpr (QL > 1) return xcd_chunk(block_id, n_blocks) * PPW + static_cast<int>(threadIdx.x) / QL;  // (every wave has a SIMD to itself)
    constexpr int WPB = TPB / 64;
    const int cpx = n_blocks >> 3, x = block_id & 7, j = block_id >> 3, wv = static_cast<int>(threadIdx.x >> 6);
#if MH_XCD_PIECES > 1
    // An XCD's chunks come from MH_XCD_PIECES (2) separate pieces of the curve, the odd piece in mirrored XCD order, so that a heavy
    // eighth of the cloud is shared by two XCDs instead of ending the kernel on one: the same number of points and two compact
    // regions per L2 (29.3-29.7 -> 29.0 us at 131 072 points, 21.3 -> 20.8 us at 24 576; four pieces: 31.3 us — gpurun c26).
    constexpr int P = MH_XCD_PIECES;
    static_assert(WPB % P == 0, "pieces must divide the waves of a workgroup");
    const int S = cpx * (WPB / P), p = wv / (WPB / P), o = (wv % (WPB / P)) * cpx + j;
    return ((p * 8 + ((p & 1) ? 7 - x : x)) * S + o) * 64 + static_cast<int>(threadIdx.x & 63u);
#else
    return ((x * cpx * WPB) + wv * cpx + j) * 64 + static_cast<int>(threadIdx.x & 63u);
#endif
  }();
#else
  const int qi = xcd_chunk(block_id, n_blocks) * PPW + static_cast<int>(threadIdx.x) / QL;
#endif
  // the lane's source point is requested before the scan table is filled (a load of its own) and the barrier behind it: one
  // memory round trip less on every wave's chain
  const int k = (K == 5) ? 5 : a.k;  // compile-time in the fast instantiation: no `j < k` branches
  const int n_pts = SHARD ? static_cast<int>(*a.n_dev) : a.n;
  float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (qi < n_pts) sp = a.src[qi];
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
  fill_scan_lut<NOFF>(s_scan);
  __syncthreads();

  double row[NV];
  [[maybe_unused]] double rec_jr[3] = {0.0, 0.0, 0.0}, rec_jt[3] = {0.0, 0.0, 0.0};  // this point's entry of the call's record (IcpArgs::rec)
  [[maybe_unused]] int rec_st = -1;
  uint32_t cnt_pack = 0u;  // this lane's k-NN counters, reduced per wave after the per-point section
  bool did_knn = false, did_fall = false;
#ifdef MH_TIMELINE
  // diagnostic: repeat the per-point section (MH_REPS env) so the last pass runs with warm caches
  for (int rep = 0; rep < a.reps; ++rep) {
  if (rep > 0) {
    __syncthreads();
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
  }
#endif
  MH_STAMP(a.dbg, 0);
#pragma unroll
  for (int j = 0; j < NV; ++j) row[j] = 0.0;

  if (qi < n_pts) {
    const double px = sp.x, py = sp.y, pz = sp.z;
    // 1. q = R p + t (geometric_factor.hpp:276-277)
    const double q0 = (a.R[0] * px + (a.R[1] * py + a.R[2] * pz)) + a.t[0];
    const double q1 = (a.R[3] * px + (a.R[4] * py + a.R[5] * pz)) + a.t[1];
    const double q2 = (a.R[6] * px + (a.R[7] * py + a.R[8] * pz)) + a.t[2];

    // 2. data-association cache (:279-287).  cold == a freshly constructed factor: cached state
    //    reads as zero without touching memory.
    double qd0 = 0.0, qd1 = 0.0, qd2 = 0.0;
    int st = MH_UNPROCESSED;
    if (!a.cold) {
      qd0 = a.q_da[3 * qi + 0];
      qd1 = a.q_da[3 * qi + 1];
      qd2 = a.q_da[3 * qi + 2];
      st = a.status[qi];
    }
    bool gone = SHARD && (st & kShardSkip) != 0;  // map-sharded factors only: another rank's point in this call
    if constexpr (SHARD) {
      if (a.cold) {
        // a sharded factor after mh_shard_icp_reset: the association state reads as zero (as above), but WHICH slots are this
        // rank's is still in the status word — tombstones (-1) stay, a held-back mover (skip flag on a live status) is reset
        // in place, since no pass of this call will touch it
        const int raw = a.status[qi];
        gone = (raw & kShardSkip) != 0;
        if (gone && raw != -1) {
          a.status[qi] = kShardSkip;
          a.q_da[3 * qi + 0] = a.q_da[3 * qi + 1] = a.q_da[3 * qi + 2] = 0.0;
          a.mean[3 * qi + 0] = a.mean[3 * qi + 1] = a.mean[3 * qi + 2] = 0.0;
          a.normal[3 * qi + 0] = a.normal[3 * qi + 1] = a.normal[3 * qi + 2] = 0.0;
        }
      }
    }
    const double ddx = q0 - qd0, ddy = q1 - qd1, ddz = q2 - qd2;
    // (q - q_da).norm() > threshold (:281), compared as squares: both sides are non-negative
    const bool update = !gone && (ddx * ddx + (ddy * ddy + ddz * ddz)) > a.da_thresh * a.da_thresh;

    double mean[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
    bool go = false;
    if (update) {
      st = MH_UNPROCESSED;
      if (lead) {
        a.q_da[3 * qi + 0] = q0;
        a.q_da[3 * qi + 1] = q1;
        a.q_da[3 * qi + 2] = q2;
      }
      // 3. k-NN (:292-302)
      uint32_t bi[K];
      double dk;
      bool fell_back;
      uint32_t n_scanned;
      constexpr bool kFast = K == 5;  // k == K: the nearest points as a set (knn_query, FAST)
      constexpr int KS = knn_survivors(K);
      [[maybe_unused]] KnnPoints<KS> sel;
      // quads in flight per lane: the 256-thread class (clouds of up to 65 536 points: at most one wave per SIMD, the scan waits
      // on memory, registers are plentiful) runs a deeper pipeline than the 512-thread class (two waves per SIMD: VALU-bound)
      constexpr int kPipeK3 = TPB <= 256 ? MH_PIPE_SMALL : MH_PIPE;
      const uint32_t n_cand = knn_query<K, NOFF, kFast, kPipeK3, QL>(a.map, q0, q1, q2, k, s_list, PPW, s_scan, bi, dk, fell_back, n_scanned, a.dbg,
                                                                            kFast ? &sel : nullptr);
      cnt_pack = (lead ? n_cand : 0u) | (n_scanned << 16);  // each <= 27 x 20 = 540: the 64-lane sums fit 16 bits (several lanes per query: every lane its own share of the scan)
      did_knn = lead;
      did_fall = fell_back;
      if (!(dk < kDblMax)) {
        st = MH_INSUFFICIENT_CORRES_POINTS;  // found != k
      } else if (dk > a.max_d2) {
        st = MH_CORRES_MAX_DIST;
      } else {
        // 4. estimatePlane (:176-229)
        constexpr int NX = kFast ? KS : K;  // FAST: the survivors of the coarse tier, the k nearest marked in sel.member (the others count as zero)
        double X[NX][3];
        [[maybe_unused]] double dm[NX];  // FAST: 1.0 for a member, 0.0 otherwise (the two differ in the high word only: one select)
        double sx = 0, sy = 0, sz = 0;
        if constexpr (kFast) {
#pragma unroll
          for (int j = 0; j < NX; ++j) {
            const bool m = (sel.member >> j) & 1u;
            // the select on the f32 value (one instruction), then the conversion: a point that is not one of the k nearest is 0
            X[j][0] = static_cast<double>(m ? sel.pt[j].x : 0.0f);
            X[j][1] = static_cast<double>(m ? sel.pt[j].y : 0.0f);
            X[j][2] = static_cast<double>(m ? sel.pt[j].z : 0.0f);
            dm[j] = __hiloint2double(m ? 0x3FF00000 : 0, 0);
            sx += X[j][0];
            sy += X[j][1];
            sz += X[j][2];
          }
        } else {
#pragma unroll
          for (int j = 0; j < K; ++j) {
            X[j][0] = X[j][1] = X[j][2] = 0.0;
            if (j < k) {
              const float4 c = a.map.buckets[bi[j]];
              X[j][0] = c.x;
              X[j][1] = c.y;
              X[j][2] = c.z;
              sx += X[j][0];
              sy += X[j][1];
              sz += X[j][2];
            }
          }
        }
        MH_STAMP_C2(a.dbg, 13);
        const double kd = static_cast<double>(k);
        const double ikd = 1.0 / kd;  // (a compile-time constant in the k = 5 instantiation)
        mean[0] = sx * ikd;
        mean[1] = sy * ikd;
        mean[2] = sz * ikd;
        double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          // a point that is not one of the k nearest stays at zero: it adds nothing to the covariance and passes the plane gate
          if constexpr (kFast) {
            // X - dm * mean: the same rounding as X - mean for a member (dm = 1), exactly 0 otherwise (X = 0, dm = 0)
            X[j][0] = fma(-dm[j], mean[0], X[j][0]);
            X[j][1] = fma(-dm[j], mean[1], X[j][1]);
            X[j][2] = fma(-dm[j], mean[2], X[j][2]);
          } else {
            const bool m = j < k;
            X[j][0] = m ? X[j][0] - mean[0] : 0.0;
            X[j][1] = m ? X[j][1] - mean[1] : 0.0;
            X[j][2] = m ? X[j][2] - mean[2] : 0.0;
          }
          c00 += X[j][0] * X[j][0];
          c01 += X[j][0] * X[j][1];
          c02 += X[j][0] * X[j][2];
          c11 += X[j][1] * X[j][1];
          c12 += X[j][1] * X[j][2];
          c22 += X[j][2] * X[j][2];
        }
        const double ikm1 = 1.0 / (kd - 1.0);
        // the mean is cached before the gates (:191)
        if (lead) {
          a.mean[3 * qi + 0] = mean[0];
          a.mean[3 * qi + 1] = mean[1];
          a.mean[3 * qi + 2] = mean[2];
        }
        double w[3], v0[3];
        MH_STAMP_C2(a.dbg, 14);
        plane_eigen(c00 * ikm1, c01 * ikm1, c02 * ikm1, c11 * ikm1, c12 * ikm1, c22 * ikm1, w, v0);
        MH_STAMP_C2(a.dbg, 15);
        if (!(w[0] == w[0]) || !(w[2] == w[2])) {
          st = MH_EIGEN_SOLVER_FAIL;  // NaN input: Eigen would report NoConvergence (:197)
        } else if (w[0] < 1e-6) {
          st = MH_MIN_EIGEN_VALUE_LOW;
        } else if (w[2] > 3.0 * w[1]) {
          st = MH_LINE;
        } else {
          nrm[0] = v0[0];
          nrm[1] = v0[1];
          nrm[2] = v0[2];
          // normal faces the sensor origin o = t (:217-220)
          const double dp =
            nrm[0] * (a.t[0] - mean[0]) + (nrm[1] * (a.t[1] - mean[1]) + nrm[2] * (a.t[2] - mean[2]));
          if (dp < 0) {
            nrm[0] = -nrm[0];
            nrm[1] = -nrm[1];
            nrm[2] = -nrm[2];
          }
          if (lead) {
            a.normal[3 * qi + 0] = nrm[0];
            a.normal[3 * qi + 1] = nrm[1];
            a.normal[3 * qi + 2] = nrm[2];
          }
          bool plane_ok = true;
#pragma unroll
          for (int j = 0; j < NX; ++j) {
            if (kFast || j < k) {
              const double dj = X[j][0] * nrm[0] + (X[j][1] * nrm[1] + X[j][2] * nrm[2]);
              if (fabs(dj) > a.plane_valid) plane_ok = false;
            }
          }
          if (plane_ok)
            go = true;
          else
            st = MH_CORRES_PLANE_INVALID;
        }
      }
      if (a.cold && lead) {
        // a fresh factor has zero mean / normal wherever this pass did not write them
        if (st == MH_INSUFFICIENT_CORRES_POINTS || st == MH_CORRES_MAX_DIST)
          a.mean[3 * qi + 0] = a.mean[3 * qi + 1] = a.mean[3 * qi + 2] = 0.0;
        if (st >= MH_INSUFFICIENT_CORRES_POINTS && st <= MH_LINE)
          a.normal[3 * qi + 0] = a.normal[3 * qi + 1] = a.normal[3 * qi + 2] = 0.0;
      }
    } else {
      // :308-317 — reuse the cached plane only if the previous pass got past the plane gates
      if (!gone && st > MH_CORRES_PLANE_INVALID) {
        mean[0] = a.mean[3 * qi + 0];
        mean[1] = a.mean[3 * qi + 1];
        mean[2] = a.mean[3 * qi + 2];
        nrm[0] = a.normal[3 * qi + 0];
        nrm[1] = a.normal[3 * qi + 1];
        nrm[2] = a.normal[3 * qi + 2];
        go = true;
      } else if (a.cold && lead) {
        // lazily materialise the zero state of a fresh factor for points that never associate
        a.q_da[3 * qi + 0] = a.q_da[3 * qi + 1] = a.q_da[3 * qi + 2] = 0.0;
        a.mean[3 * qi + 0] = a.mean[3 * qi + 1] = a.mean[3 * qi + 2] = 0.0;
        a.normal[3 * qi + 0] = a.normal[3 * qi + 1] = a.normal[3 * qi + 2] = 0.0;
      }
    }

    if (go && lead) {
      // 5. residual + max-error gate (:319-328)
      double e = nrm[0] * (mean[0] - q0) + (nrm[1] * (mean[1] - q1) + nrm[2] * (mean[2] - q2));
      // s = 1 - 0.9 |e| / sqrt(range) < 0.9 (:322-326)  <=>  9 |e| > sqrt(range)  <=>  (81 e^2)^2 > range^2 = |p|^2: the gate
      // needs the comparison only, and that form has no square root in it (rounding differs from the reference's expression in
      // the last bits of the threshold, as every other gate's does)
      const double r2 = px * px + (py * py + pz * pz);
      const double e81 = 81.0 * (e * e);
      if (e81 * e81 > r2) {
        st = MH_MAX_ERROR;
      } else {
        // 6. Huber (:330-339)
        double sw = 1.0;
        if (a.use_huber) {
          const double we = e * a.inv_sigma;
          if (fabs(we) > a.huber) sw = mh_rsqrt(fabs(we) * a.inv_huber);  // sqrt(huber / |we|) (:334-336)
        }
        const double wgt = sw * a.inv_sigma;
        e *= wgt;
        // 7. Jacobian (:341-355): n_s = R^T n, J = [(n_s x p)^T, -n_s^T]
        const double ns0 = a.R[0] * nrm[0] + (a.R[3] * nrm[1] + a.R[6] * nrm[2]);
        const double ns1 = a.R[1] * nrm[0] + (a.R[4] * nrm[1] + a.R[7] * nrm[2]);
        const double ns2 = a.R[2] * nrm[0] + (a.R[5] * nrm[1] + a.R[8] * nrm[2]);
        row[0] = (ns1 * pz - ns2 * py) * wgt;
        row[1] = (ns2 * px - ns0 * pz) * wgt;
        row[2] = (ns0 * py - ns1 * px) * wgt;
        row[3] = -ns0 * wgt;
        row[4] = -ns1 * wgt;
        row[5] = -ns2 * wgt;
        if constexpr (BINARY) {
          // :368-376  J_t = [(q x n)^T, n^T]
          row[6] = (q1 * nrm[2] - q2 * nrm[1]) * wgt;
          row[7] = (q2 * nrm[0] - q0 * nrm[2]) * wgt;
          row[8] = (q0 * nrm[1] - q1 * nrm[0]) * wgt;
          row[9] = nrm[0] * wgt;
          row[10] = nrm[1] * wgt;
          row[11] = nrm[2] * wgt;
        }
        row[NV - 1] = e;
        st = MH_VALID;
      }
    }
    if (!gone && lead) a.status[qi] = st;
    if constexpr (!SHARD) {
      if (a.rec && lead) {
        if (st == MH_VALID) loc_directions(a, nrm[0], nrm[1], nrm[2], px, py, pz, rec_jr, rec_jt);
        rec_st = st;
      }
    }
  }
  {
    // k-NN counters: one LDS atomic per wave and counter (uniform control flow: every lane is active here)
    const uint32_t packed = wave_sum_to_lane63(cnt_pack);
    const unsigned long long mk = __ballot(did_knn), mf = __ballot(did_fall);
    if ((threadIdx.x & 63) == 63) {
      atomicAdd(&s_cnt[0], static_cast<unsigned int>(__popcll(mk)));
      atomicAdd(&s_cnt[1], packed & 0xFFFFu);
      atomicAdd(&s_cnt[2], static_cast<unsigned int>(__popcll(mf)));
      atomicAdd(&s_cnt[3], packed >> 16);
    }
    cnt_pack = 0u;
    did_knn = did_fall = false;
  }
#ifdef MH_TIMELINE
  }
#endif

  // the call's record: what K4 reads of this point (the kernel boundary makes it visible)
  [[maybe_unused]] auto store_record = [&]() {
    if constexpr (!SHARD) {
      const size_t rn = static_cast<size_t>(a.rec_n);
      int32_t * rst = reinterpret_cast<int32_t *>(a.rec + 6 * rn);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        a.rec[static_cast<size_t>(c) * rn + qi] = rec_jr[c];
        a.rec[static_cast<size_t>(3 + c) * rn + qi] = rec_jt[c];
      }
      rst[qi] = rec_st;
    }
  };
  // 8. H += J^T J, b += J^T e, f += e^2 (:363-382).  Every WAVE reduces its own 64 rows first, in its own time — a tile of rows in
  //    LDS that only this wave touches (no workgroup barrier: the k-NN arena of the slower waves is still in use), every lane
  //    owning one (entry, point-segment) of the upper triangle of sum v v^T — so that what is left behind the barrier, on the
  //    critical path of the workgroup's slowest wave, is a sum of 8 (16) wave partials per entry.  (Rounds 1-4 built one tile per
  //    workgroup behind the barrier: 4.8 k cycles = 2 us after the slowest wave of every workgroup.)
  MH_STAMP(a.dbg, 3);
  {
    constexpr int WSEG = NENT <= 32 ? 2 : 1;          // point segments per wave: lanes 0-31 / 32-63 (unary), all 64 points (binary)
    constexpr int WPTS = 64 / WSEG;
    const int lane = static_cast<int>(threadIdx.x & 63u), wv = static_cast<int>(threadIdx.x >> 6);
    double * tile = s_tile + static_cast<size_t>(wv) * 64 * ROWW;
#pragma unroll
    for (int j = 0; j < NV; ++j) tile[lane * ROWW + j] = row[j];
    // the wave's own stores, then its own loads: ordered by the LDS queue; the fences keep the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int ROUNDS = (NENT * WSEG + 63) / 64;   // 1 (unary: 56 lanes busy), 2 (binary: 91 entries)
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int item = rd * 64 + lane;
      const int ent = WSEG == 2 ? (lane & 31) : item, seg = WSEG == 2 ? (lane >> 5) : 0;
      if (ent < NENT) {
        int r = 0, rem = ent;  // ent -> (r, c), r <= c, row-major upper triangle
        while (rem >= NV - r) {
          rem -= NV - r;
          ++r;
        }
        const int c = r + rem;
        const double * pr = tile + seg * WPTS * ROWW;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent accumulators: a batch of LDS reads in flight together
#pragma unroll
        for (int p = 0; p < WPTS; p += 4) {
          s0 += pr[(p + 0) * ROWW + r] * pr[(p + 0) * ROWW + c];
          s1 += pr[(p + 1) * ROWW + r] * pr[(p + 1) * ROWW + c];
          s2 += pr[(p + 2) * ROWW + r] * pr[(p + 2) * ROWW + c];
          s3 += pr[(p + 3) * ROWW + r] * pr[(p + 3) * ROWW + c];
        }
        s_wsum[(wv * WSEG + seg) * NENT + ent] = (s0 + s1) + (s2 + s3);
      }
    }
    // the call's record (what K4 reads of this point), in the wave's own time
    if constexpr (!SHARD) {
      if (a.rec && qi < a.n && lead) store_record();
    }
  }
  __syncthreads();
  MH_STAMP(a.dbg, 4);
  // A plain factor whose K4 follows (a.tail == 0) ends here: the row is an ordinary store, K4's workgroups fold the rows
  // themselves after the kernel boundary — no write-through, no ticket, no last-block fold (3 us of serial tail with 255
  // CUs idle in round 3).  Otherwise (no K4 behind it, or a map-sharded factor, whose sums feed an all-reduce): rows
  // write-through, ticket, fold by the last block.
  const bool fold_here = SHARD || a.tail != 0;
  const bool through = fold_here;
  if (threadIdx.x < NENT) {
    constexpr int NPART = (TPB / 64) * (NENT <= 32 ? 2 : 1);  // wave partials per entry, summed in index order: deterministic
    double pv[NPART];
#pragma unroll
    for (int g = 0; g < NPART; ++g) pv[g] = s_wsum[g * NENT + threadIdx.x];  // all requested before the first is used
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < NPART; ++g) s += pv[g];
    double * dst = &a.partials[static_cast<size_t>(block_id) * kPartialStride + threadIdx.x];
    if (through)
      store_partial(dst, s);
    else
      *dst = s;
  }
  // the block's k-NN counters ride along as four more partial entries (exact in fp64): same-line
  // global atomics from every block would serialise at L2
  if (threadIdx.x >= 128 && threadIdx.x < 132) {
    double * dst = &a.partials[static_cast<size_t>(block_id) * kPartialStride + NENT + (threadIdx.x - 128)];
    const double c = static_cast<double>(s_cnt[threadIdx.x - 128]);
    if (through)
      store_partial(dst, c);
    else
      *dst = c;
  }

  MH_STAMP(a.dbg, 5);
  if (!fold_here) return;
  if (!arrive_is_last(a.ticket, static_cast<unsigned int>(n_blocks), &s_last)) return;
  MH_STAMP(a.dbg, 6);

  // ---- last block: fold the partial rows in fixed order, finalise --------------------------------
  double * s_sum = s_aux + (TPB / EW) * EW;
  fold_rows<EW, TPB>(a.partials, n_blocks, NENT + 4, s_aux, s_sum);
  if constexpr (SHARD) {
    // Results go to the device struct (the caller-driven two-phase form reads them there) and, when given, to a mapped host slot
#define MH_PUT(field, val)                          \
  do {                                              \
    a.result->field = (val);                        \
    if (a.host_result) a.host_result->field = (val); \
  } while (0)
    if (threadIdx.x < NENT) MH_PUT(sums[threadIdx.x], s_sum[threadIdx.x]);
    if (threadIdx.x == NENT) MH_PUT(n_knn, static_cast<unsigned long long>(s_sum[NENT]));
    if (threadIdx.x == NENT + 1) MH_PUT(n_cand, static_cast<unsigned long long>(s_sum[NENT + 1]));
    if (threadIdx.x == NENT + 2) MH_PUT(n_fallback, static_cast<unsigned long long>(s_sum[NENT + 2]));
    if (threadIdx.x == NENT + 3) MH_PUT(n_scanned, static_cast<unsigned long long>(s_sum[NENT + 3]));
#undef MH_PUT
    if (a.shard_out && threadIdx.x < NENT + 4) a.shard_out[threadIdx.x] = s_sum[threadIdx.x];  // what the shards all-reduce
  } else {
    // this kernel is the whole call: sums + counters straight to the host as flagged words (computeLocalizability of the
    // rot / trans blocks, :405-411, is the host epilogue's: finish_result)
    if (threadIdx.x < NENT + 4) ll_store(a.ll + threadIdx.x, s_sum[threadIdx.x], a.seq);
  }
  MH_STAMP(a.dbg, 7);
}

template <int K, bool BINARY, int NOFF, int TPB, bool SHARD, int QL = 1>
__global__ __launch_bounds__(TPB) void icp_linearize_kernel(const IcpArgs a)
{
  icp_linearize_body<K, BINARY, NOFF, TPB, SHARD, QL>(a, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
}

// A kernel-argument block read from memory into SGPRs: lane i of the wave loads dword i (one or two coalesced
// loads per wave), then every dword is broadcast with v_readlane (constant lane index -> SGPR), so the compiler
// treats the fields as wave-uniform scalars exactly like a by-value kernarg (plain global loads would put R, t
// and the thresholds into VGPRs of a kernel that has none to spare).  The buffer behind `p` must be readable up
// to the next multiple of 256 bytes.
template <typename T>
__device__ __forceinline__ T load_uniform(const T * p)
{
  static_assert(sizeof(T) % 4 == 0, "dword-sized argument blocks only");
  constexpr int NW = static_cast<int>(sizeof(T) / 4), NL = (NW + 63) / 64;
  union U {
    T v;
    uint32_t w[NW];
    __device__ U() {}
  } u;
  const uint32_t * s = reinterpret_cast<const uint32_t *>(p);
  const int lane = static_cast<int>(threadIdx.x & 63u);
  uint32_t chunk[NL];
#pragma unroll
  for (int c = 0; c < NL; ++c) chunk[c] = s[c * 64 + lane];
#pragma unroll
  for (int i = 0; i < NW; ++i) u.w[i] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(chunk[i / 64]), i % 64));
  return u.v;
}

// Which factor of a batch does workgroup b belong to?  start[] = exclusive prefix of the per-factor grids
// (n_factors + 1 entries, each grid a multiple of 8 so b % 8 — the XCD — is the same inside the segment).
__device__ __forceinline__ int batch_factor_of(const int * start, int n_factors, int b)
{
  int f = 0;
  for (int i = 1; i < n_factors; ++i) f += (b >= start[i]) ? 1 : 0;  // n_factors <= 64, uniform scalar loop
  return __builtin_amdgcn_readfirstlane(f);
}

// The sliding window's live factors in ONE launch (graph::Manager's smoother_->update re-linearizes every live
// ICPFactor, src/graph/manager.cpp:585-588): one grid over the concatenated per-factor grids; each workgroup
// picks up its factor's argument block and runs the same body; each factor keeps its own partial rows, ticket
// and last-block fold, so the results are bit-identical to separate launches.
template <int K, bool BINARY, int NOFF, int TPB>
__global__ __launch_bounds__(TPB) void icp_linearize_batch_kernel(const IcpArgs * args, const int * start, int n_factors)
{
  const int b = static_cast<int>(blockIdx.x);
  const int f = batch_factor_of(start, n_factors, b);
  const int s0 = __builtin_amdgcn_readfirstlane(start[f]), s1 = __builtin_amdgcn_readfirstlane(start[f + 1]);
  const IcpArgs a = load_uniform(args + f);
  icp_linearize_body<K, BINARY, NOFF, TPB, false>(a, b - s0, s1 - s0);
}

// The same with the argument blocks inside the kernel-argument segment.  They are read through the segment pointer:
// indexing the by-value parameter itself with a runtime index makes the compiler copy the whole struct to scratch.
template <int K, bool BINARY, int NOFF, int TPB, bool SHARD, int QL = 1>
__global__ __launch_bounds__(TPB) void icp_linearize_batch_inline_kernel(const BatchInline<IcpArgs> blk)
{
  (void)blk;
  const auto * p = (const BatchInline<IcpArgs> *)__builtin_amdgcn_kernarg_segment_ptr();
  const int b = static_cast<int>(blockIdx.x);
  const int f = batch_factor_of(p->start, p->n, b);
  const int s0 = __builtin_amdgcn_readfirstlane(p->start[f]), s1 = __builtin_amdgcn_readfirstlane(p->start[f + 1]);
  const IcpArgs a = load_uniform(p->a + f);
  icp_linearize_body<K, BINARY, NOFF, TPB, SHARD, QL>(a, b - s0, s1 - s0);
}

// ------------------------------------------------------------------------------------------------
// K4: component localizabilities (geometric_factor.hpp:434-457) + status histogram
// (src/lidar/geometric.cpp:280-323).  Recomputes the unwhitened Jacobian directions from the cached
// normal instead of storing two more per-point vectors.  6 sums by wave shuffles, 9 counts by
// ballot/popcount; per-block row of 15 -> same ticket + fold as K3.
// ------------------------------------------------------------------------------------------------
// A plain factor's K4 runs 256 point-carrying threads per workgroup whatever K3 ran (the batched body shares the shape, so a
// factor's rows fold — and its sums add — in the same order in a single call and in a window batch: the two agree to the bit).
__host__ __device__ constexpr int loc_tpb(int tpb, bool shard) { return (!shard && tpb > 256) ? 256 : tpb; }
__host__ __device__ constexpr int loc_threads(int tpb, bool shard) { return loc_tpb(tpb, shard) + (shard ? 0 : 64); }  // + the decomposition wave (XW)
// A K4 workgroup of a plain factor takes kLocChunks chunks of 256 points whatever K3's class was: 1024 points, i.e. 4 of K3's
// workgroups of the 256-thread class or MH_LOC_BLOCKS_512 = 2 of the 512-thread class (round 4 took 4 = 2048 points per workgroup:
// 64 workgroups for 131 072 points, a quarter of the CUs, each lane holding 8 chunks' record entries — the projection phase
// alone was 1.4 us; 128 workgroups halve it and the host folds 128 rows of two cache lines instead of 64 of three).
#ifndef MH_LOC_BLOCKS_512
#define MH_LOC_BLOCKS_512 2
#endif
__host__ __device__ constexpr int loc_ch(int tpb, bool shard) { return shard ? 1 : (tpb > 256 ? 2 * MH_LOC_BLOCKS_512 : kLocChunks); }

// XW: the workgroup has one wave MORE than TPB / 64 (the synchronous launch of a plain factor): that wave holds no points — it
// takes part in the fold's barriers and then decomposes H_rr and H_tt on two of its lanes AT ONCE (one SIMD pass for both), while
// the four point waves issue their 28 loads each; with the decompositions on lanes of two point waves those loads (1.7 k cycles
// of issue on a wave that has a SIMD to itself) sat in front of the 5 k-cycle decompositions on the kernel's critical path.  The
// bases are the same bits whichever lane works them out, so the batched body (no extra wave) agrees to the bit.
template <int TPB, bool SHARD, int CH, bool XW = false>
__device__ __forceinline__ void icp_localizability_body(const LocArgs & a, const int block_id, const int n_blocks)
{
  constexpr int NW = TPB / 64;
  const bool worker = !XW || threadIdx.x < static_cast<unsigned int>(TPB);  // (wave-uniform)
  __shared__ double s_w[NW][16];
  __shared__ double s_seg[TPB + 96 + 96];  // fold scratch: (TPB / EW) * EW segment sums + EW totals, EW = 32 or 96
  __shared__ bool s_last;

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  MH_STAMP4(a.dbg, 0);
  double v[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long hpack = 0ull;
  // The pass is a handful of memory round trips long and nothing else, so everything that does not depend on the eigenbases
  // happens BEFORE they are known: the status / point / normal of ALL of this workgroup's chunks are requested up front (one
  // round trip, in flight while the Hessian rows are folded), and the unwhitened Jacobian directions of the Valid points
  // (geometric_factor.hpp:343-352) are worked out while two lanes of the workgroup decompose H_rr and H_tt.
  // (a plain factor without a record: its component pass is switched off but a batch runs K4 for every member — nothing to
  // read, the host reports NaN for it)
  const int n_pts = !worker ? 0 : (SHARD ? static_cast<int>(*a.n_dev) : (a.rec ? a.n : 0));
  // The record entries (plain) / status, point and normal (map-sharded) of ALL of this workgroup's chunks: one round trip, in
  // flight while two lanes of the workgroup decompose H_rr and H_tt.  One uniform branch around each chunk's loads (a select
  // per loaded value made the compiler branch around every single load).
  int st_c[CH];
  double jr_c[CH][3], jt_c[CH][3];
  [[maybe_unused]] float4 sp_c[CH];
  [[maybe_unused]] double nx_c[CH], ny_c[CH], nz_c[CH];
  auto load_chunk = [&](const int ch) {
    const int i = (block_id * CH + ch) * TPB + static_cast<int>(threadIdx.x);
    const int i_ld = i < n_pts ? i : 0;
    st_c[ch] = -1;
    jr_c[ch][0] = jr_c[ch][1] = jr_c[ch][2] = jt_c[ch][0] = jt_c[ch][1] = jt_c[ch][2] = 0.0;
    if constexpr (SHARD) {
      sp_c[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
      nx_c[ch] = ny_c[ch] = nz_c[ch] = 0.0;
      if (n_pts > 0) {
        st_c[ch] = a.status[i_ld];
        sp_c[ch] = a.src[i_ld];
        nx_c[ch] = a.normal[3 * i_ld];
        ny_c[ch] = a.normal[3 * i_ld + 1];
        nz_c[ch] = a.normal[3 * i_ld + 2];
      }
    } else if (n_pts > 0) {
      // plain factor: the call's record (K3 wrote the directions, zero unless Valid, and the status)
      const size_t rn = static_cast<size_t>(a.rec_n);
      const int32_t * rst = reinterpret_cast<const int32_t *>(a.rec + 6 * rn);
      st_c[ch] = rst[i_ld];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        jr_c[ch][c] = a.rec[static_cast<size_t>(c) * rn + i_ld];
        jt_c[ch][c] = a.rec[static_cast<size_t>(3 + c) * rn + i_ld];
      }
    }
    if (i >= n_pts) st_c[ch] = -1;
  };
  auto load_all = [&]() {  // (behind the fold: requested inside the rows' round trip they delay the rows — K4 6.4 -> 7.3 us, gpurun c21)
    MH_STAMP4(a.dbg, 1);
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) load_chunk(ch);
  };
  // The Hessian sums the eigenbases come from — requested FIRST: the fold, the decompositions and everything behind them wait
  // for these rows, the record entries further down only for the projections.  Plain factor: K3 ended at its per-workgroup
  // rows; EVERY workgroup of this kernel folds them for itself (k3_blocks rows of <= 95 doubles, one round trip, the fixed
  // order of the old last-block fold — so every workgroup holds the same bits), workgroup 0 also publishes them.
  // Map-sharded factor: the all-reduced (global) sums are given.
  const double * sums = nullptr;
  double * s_h = s_seg + TPB;  // folded sums + counters (plain factors)
  if constexpr (SHARD) {
    sums = a.sums ? a.sums : a.result->sums;
  } else {
    const int n_ent = a.nv * (a.nv + 1) / 2 + 4;
    constexpr int FB = 32;  // rows in flight per thread: 256 rows over 8 segments in one round trip
    if (a.nv == 7)
      fold_rows<32, TPB, true, FB>(a.partials, a.k3_blocks, n_ent, s_seg, s_h);
    else
      fold_rows<96, TPB, true, FB>(a.partials, a.k3_blocks, n_ent, s_seg, s_h);
    sums = s_h;
  }
  load_all();
  MH_STAMP4(a.dbg, 2);
  // The two eigenbases: given (two-phase callers), or derived here — one lane per 3 x 3 block (computeLocalizability,
  // utils.hpp:308-313), every workgroup for itself, on the LAST two waves (the others go on to their points).
  __shared__ double s_E[18];
  if (a.eig) {
    if (threadIdx.x < 18) s_E[threadIdx.x] = a.eig[threadIdx.x];
  } else if (XW ? (threadIdx.x == TPB || threadIdx.x == TPB + 1) : (threadIdx.x == TPB - 64 || threadIdx.x == TPB - 128)) {
    const int NV = a.nv, o = (XW ? threadIdx.x == TPB + 1 : threadIdx.x == TPB - 64) ? 3 : 0;
    double Hb[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        const int rr = (r < c ? r : c) + o, cc = (r < c ? c : r) + o;
        Hb[3 * r + c] = sums[rr * NV - rr * (rr - 1) / 2 + (cc - rr)];
      }
    double loc[3], E[9];
    if (!sym_eigvec3_fast(Hb, E)) compute_localizability(Hb, loc, E);  // (the vectors only, verified; the full solver behind it)
    for (int q = 0; q < 9; ++q) s_E[(o ? 9 : 0) + q] = E[q];
  }
  MH_STAMP4(a.dbg, 3);
  // Jacobian directions of this thread's points (independent of the eigenbases): map-sharded / two-phase callers work them
  // out here from the stored normals, plain factors loaded them above
  if constexpr (SHARD) {
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
      loc_directions(a, nx_c[ch], ny_c[ch], nz_c[ch], static_cast<double>(sp_c[ch].x), static_cast<double>(sp_c[ch].y),
                     static_cast<double>(sp_c[ch].z), jr_c[ch], jt_c[ch]);
  }
  MH_STAMP4(a.dbg, 4);
  __syncthreads();
  MH_STAMP4(a.dbg, 5);
  double er[9], et[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    er[q] = s_E[q];
    et[q] = s_E[9 + q];
  }
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int st = st_c[ch];
    if (st == MH_VALID) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double tc = fabs(jt_c[ch][0] * et[c] + (jt_c[ch][1] * et[3 + c] + jt_c[ch][2] * et[6 + c]));
        const double rc = fabs(jr_c[ch][0] * er[c] + (jr_c[ch][1] * er[3 + c] + jr_c[ch][2] * er[6 + c]));
        v[c] += tc >= 0.5 ? tc : 0.0;      // trans components
        v[3 + c] += rc >= 0.5 ? rc : 0.0;  // rot components
      }
    }
    // status histogram (src/lidar/geometric.cpp:280-323): this lane's counts, 4 bits per status (<= 8 chunks per lane)
    hpack += st >= 0 && st < 9 ? (1ull << (4 * st)) : 0ull;
  }
  MH_STAMP4(a.dbg, 6);
  // wave sums by DPP (wave_sum_f64_to_lane63); the nine counts travel three to a 32-bit word (<= 64 lanes x 8 chunks = 512 each)
  // (all nine reductions in straight-line code, stores afterwards: a store under `lane == 63` between two of them is a branch
  // the scheduler does not interleave across, and each reduction alone is six dependent steps on an otherwise idle SIMD)
  wave_sum6_f64_to_lane63(v);
  uint32_t hw[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const uint32_t f0 = static_cast<uint32_t>(hpack >> (12 * g)) & 15u, f1 = static_cast<uint32_t>(hpack >> (12 * g + 4)) & 15u,
                   f2 = static_cast<uint32_t>(hpack >> (12 * g + 8)) & 15u;
    hw[g] = f0 | (f1 << 10) | (f2 << 20);
  }
  wave_sum3_to_lane63(hw);
  if (lane == 63 && worker) {
#pragma unroll
    for (int j = 0; j < 6; ++j) s_w[wv][j] = v[j];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      s_w[wv][6 + 3 * g + 0] = static_cast<double>(hw[g] & 1023u);
      s_w[wv][6 + 3 * g + 1] = static_cast<double>((hw[g] >> 10) & 1023u);
      s_w[wv][6 + 3 * g + 2] = static_cast<double>((hw[g] >> 20) & 1023u);
    }
  }
  MH_STAMP4(a.dbg, 7);
  __syncthreads();
  MH_STAMP4(a.dbg, 8);
  if constexpr (!SHARD) {
    // The workgroup's row of 15 goes to the host as flagged words and the HOST folds the rows (in workgroup order:
    // deterministic) — no partial-row store, ticket, fold or completion flag on the device.  Workgroup 0 adds what every
    // workgroup computed identically: the Hessian sums + counters and the eigenbases this pass projected on.
    uint4 * ll_rows = a.ll + (kLlSums + kLlEig) + static_cast<size_t>(block_id) * kLlRow;
    if (threadIdx.x < 6) {
      double s = 0.0;
      for (int w2 = 0; w2 < NW; ++w2) s += s_w[w2][threadIdx.x];
      ll_store(ll_rows + threadIdx.x, s, a.seq);
    } else if (threadIdx.x < 8) {  // histogram: counts 0..4 / 5..8, 12 bits each (a workgroup covers <= 2048 points), in one word
      const int h0 = threadIdx.x == 6 ? 0 : 5, hn = threadIdx.x == 6 ? 5 : 4;
      unsigned long long pk = 0ull;
      for (int h = 0; h < hn; ++h) {
        unsigned int c = 0u;
        for (int w2 = 0; w2 < NW; ++w2) c += static_cast<unsigned int>(s_w[w2][6 + h0 + h]);
        pk |= static_cast<unsigned long long>(c) << (12 * h);
      }
      ll_rows[threadIdx.x] = make_uint4(static_cast<unsigned int>(pk), a.seq, static_cast<unsigned int>(pk >> 32), a.seq);
    }
    if (block_id == 0) {
      const int n_ent = a.nv * (a.nv + 1) / 2 + 4;
      if (threadIdx.x >= 64 && static_cast<int>(threadIdx.x) < 64 + n_ent) ll_store(a.ll + (threadIdx.x - 64), s_h[threadIdx.x - 64], a.seq);
      if (threadIdx.x >= 192 && threadIdx.x < 192 + 18) ll_store(a.ll + kLlSums + (threadIdx.x - 192), s_E[threadIdx.x - 192], a.seq);
    }
    (void)n_blocks;
    (void)s_last;
#ifdef MH_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the flagged words have left (acknowledged) by this stamp
#endif
    MH_STAMP4(a.dbg, 9);
    return;
  } else {
  if (threadIdx.x < 15) {
    double s = 0.0;
    for (int w2 = 0; w2 < NW; ++w2) s += s_w[w2][threadIdx.x];
    store_partial(&a.partials[static_cast<size_t>(block_id) * kPartialStride + threadIdx.x], s);
  }
  if (!arrive_is_last(a.ticket, static_cast<unsigned int>(n_blocks), &s_last)) return;
  double * s_sum = s_seg + (TPB / 32) * 32;
  fold_rows<32, TPB>(a.partials, n_blocks, 15, s_seg, s_sum);
  if (threadIdx.x < 6) a.result->loc_comp[threadIdx.x] = s_sum[threadIdx.x];
  if (threadIdx.x >= 6 && threadIdx.x < 15)
    a.result->status_hist[threadIdx.x - 6] = static_cast<unsigned int>(s_sum[threadIdx.x]);
  if (a.shard_out && threadIdx.x < 16) a.shard_out[threadIdx.x] = threadIdx.x < 15 ? s_sum[threadIdx.x] : 0.0;
  // the eigenbases THIS pass projected on are what the caller is told (the host's own decomposition of the same sums may
  // pick another basis of a clustered eigenspace: no FMA there, other branches of sym_eigen3)
  if (threadIdx.x >= 32 && threadIdx.x < 50) {
    const int q = threadIdx.x - 32;
    double * dst = q < 9 ? &a.result->eig_rot[q] : &a.result->eig_trans[q - 9];
    *dst = s_E[q];
    if (a.host_result) *(q < 9 ? &a.host_result->eig_rot[q] : &a.host_result->eig_trans[q - 9]) = s_E[q];
  }
  // two-phase callers (mh_icp_linearize_finish): this kernel's outputs also go to a mapped pinned host slot; the end of the
  // kernel makes them visible to the host (they synchronise the stream)
  if (a.host_result) {
    if (threadIdx.x < 6) a.host_result->loc_comp[threadIdx.x] = s_sum[threadIdx.x];
    if (threadIdx.x >= 6 && threadIdx.x < 15)
      a.host_result->status_hist[threadIdx.x - 6] = static_cast<unsigned int>(s_sum[threadIdx.x]);
  }
  }
}

template <int TPB, bool SHARD>  // TPB: the class of the K3 launch (256 / 512 threads); this kernel runs loc_threads(TPB, SHARD) threads
__global__ __launch_bounds__(loc_threads(TPB, SHARD)) void icp_localizability_kernel(const LocArgs a)
{
  icp_localizability_body<loc_tpb(TPB, SHARD), SHARD, loc_ch(TPB, SHARD), !SHARD>(a, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
}
template <int TPB>
__global__ __launch_bounds__(loc_tpb(TPB, false)) void icp_localizability_batch_kernel(const LocArgs * args, const int * start, int n_factors)
{
  const int b = static_cast<int>(blockIdx.x);
  const int f = batch_factor_of(start, n_factors, b);
  const int s0 = __builtin_amdgcn_readfirstlane(start[f]), s1 = __builtin_amdgcn_readfirstlane(start[f + 1]);
  const LocArgs a = load_uniform(args + f);
  icp_localizability_body<loc_tpb(TPB, false), false, loc_ch(TPB, false)>(a, b - s0, s1 - s0);
}

template <int TPB, bool SHARD>
__global__ __launch_bounds__(loc_tpb(TPB, SHARD)) void icp_localizability_batch_inline_kernel(const BatchInline<LocArgs> blk)
{
  (void)blk;
  const auto * p = (const BatchInline<LocArgs> *)__builtin_amdgcn_kernarg_segment_ptr();
  const int b = static_cast<int>(blockIdx.x);
  const int f = batch_factor_of(p->start, p->n, b);
  const int s0 = __builtin_amdgcn_readfirstlane(p->start[f]), s1 = __builtin_amdgcn_readfirstlane(p->start[f + 1]);
  const LocArgs a = load_uniform(p->a + f);
  icp_localizability_body<loc_tpb(TPB, SHARD), SHARD, loc_ch(TPB, SHARD)>(a, b - s0, s1 - s0);
}

// ------------------------------------------------------------------------------------------------
// Batched k-NN (IncrementalVoxelMapPCL::knn_search for n queries) — parity / tooling entry point.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void map_knn_kernel(const MapView map, const double * q, int n, int k,
                                                           double * pts, double * sq, int32_t * found)
{
  constexpr int K = 8;
  __shared__ uint32_t s_list[kMaxOff * kThreads];
  __shared__ uint32_t s_scan[kScanLutWords];
  if (map.n_off <= 7)
    fill_scan_lut<7>(s_scan);
  else if (map.n_off == 19)
    fill_scan_lut<19>(s_scan);
  else
    fill_scan_lut<27>(s_scan);
  __syncthreads();
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const double q0 = q[3 * i], q1 = q[3 * i + 1], q2 = q[3 * i + 2];
  uint32_t bi[K];
  double dk;
  bool fell_back;
  uint32_t n_scanned;
  if (map.n_off <= 7)
    knn_query<K, 7>(map, q0, q1, q2, k, s_list + threadIdx.x, kThreads, s_scan, bi, dk, fell_back, n_scanned);
  else if (map.n_off == 19)
    knn_query<K, 19>(map, q0, q1, q2, k, s_list + threadIdx.x, kThreads, s_scan, bi, dk, fell_back, n_scanned);
  else
    knn_query<K, 27>(map, q0, q1, q2, k, s_list + threadIdx.x, kThreads, s_scan, bi, dk, fell_back, n_scanned);
  int f = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if (j < k) {
      const size_t o = static_cast<size_t>(i) * k + j;
      if (bi[j] != 0xFFFFFFFFu) {
        ++f;
        const float4 c = map.buckets[bi[j]];
        pts[o * 3 + 0] = c.x;
        pts[o * 3 + 1] = c.y;
        pts[o * 3 + 2] = c.z;
        sq[o] = sq_dist3(static_cast<double>(c.x) - q0, static_cast<double>(c.y) - q1, static_cast<double>(c.z) - q2);
      } else {
        pts[o * 3 + 0] = pts[o * 3 + 1] = pts[o * 3 + 2] = 0.0;
        sq[o] = kDblMax;
      }
    }
  }
  found[i] = f;
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
// The launch CLASS of a factor = the points one K3 workgroup serves:
//   512   512 threads, one lane per point     clouds above 65 536 points (one workgroup per CU at 131 072)
//   256   256 threads, one lane per point     up to 65 536 points, map-sharded factors, the generic k != 5 path
//   128   256 threads, 2 lanes per point      plain k = 5 factors of up to kQl2Max points
//    64   256 threads, 4 lanes per point      ... of up to kQl4Max points
// The down-sampled clouds the reference feeds the factor are 10-25 k points (config/enwide/params.yaml:79-80, geometric.cpp:170-172):
// at one lane per point they put a wave on 160-400 of the machine's 1024 SIMDs and the kernel is as long as that wave's dependent
// chain; with several lanes per point the scan — the longest link of the chain — is split across lanes that would have had
// nothing to do.  Per-point results are identical in every class; the order in which the rows are summed differs with the
// class (last digits of H, like the 256 / 512 classes before).  A window batch picks its class from the batch's TOTAL: above
// 32 768 points two lanes per point would put more than one wave on a SIMD, and then the lane groups' redundant lookups and
// plane fits cost more than the shorter chain returns (the machine is issue-bound, not latency-bound, from there on).
#ifndef MH_QL2_MAX
#define MH_QL2_MAX 32768
#endif
#ifndef MH_QL4_MAX
#define MH_QL4_MAX 0  // (measured, round 6: 19.6 us at 24 576 points against 16.1 with 2 lanes and 19.0 with 1 — a third and fourth lane add waves that share SIMDs and repeat the lookup and the plane fit; the class stays built for the measurement, off by default)
#endif
static int env_or(const char * name, int dflt)
{
  const char * e = std::getenv(name);
  return (e && *e) ? std::atoi(e) : dflt;
}
int linearize_class(int n, int k, bool shard, long long total)
{
  static const int ql2_max = env_or("MH_QL2_MAX", MH_QL2_MAX), ql4_max = env_or("MH_QL4_MAX", MH_QL4_MAX);  // (tuning: tools/k3_time.py)
  const long long tot = total > 0 ? total : n;  // what decides is how many waves the LAUNCH puts on the machine's 1024 SIMDs
  if (!shard && k == 5) {
    if (tot <= ql4_max) return 64;
    if (tot <= ql2_max) return 128;
  }
  // a window of more than 65 536 points fills the machine like one big cloud: the 512-thread class for every member, whose
  // waves take chunks a stride apart (K3b 29.9 us against 31.1 for 5 x 24 576 points, rocprofv3)
  if (!shard && tot > 65536) return kThreads;
  return n <= 65536 ? 256 : kThreads;
}
int class_grid(int n, int ppw) { return (((n + ppw - 1) / ppw) + 7) & ~7; }
int linearize_grid_max(int n) { return class_grid(n, 64); }
// K4's workgroups take 1024 points of a plain factor whatever K3's class was (the kernel's time does not depend on it — launch +
// round trips — but the host folds fewer rows), one K3 workgroup's points each of a map-sharded factor (its rows are folded on
// the device as before).
static int loc_chunks(bool shard, int ppw) { return shard ? 1 : (kLocChunks * 256) / ppw; }  // in K3 workgroups
int class_loc_grid(int n, int ppw, bool shard)
{
  const int c = loc_chunks(shard, ppw);
  return (class_grid(n, ppw) + c - 1) / c;
}

template <int K, bool BINARY, int NOFF, bool SHARD>
static void launch_linearize_k(const IcpArgs & a, int ppw, hipStream_t stream)
{
  const dim3 grid(class_grid(a.n, ppw));
  if (ppw == 512) {
    hipLaunchKernelGGL((icp_linearize_kernel<K, BINARY, NOFF, kThreads, SHARD>), grid, dim3(kThreads), 0, stream, a);
    return;
  }
  if constexpr (K == 5 && !SHARD) {
    if (ppw == 128) {
      hipLaunchKernelGGL((icp_linearize_kernel<K, BINARY, NOFF, 256, SHARD, 2>), grid, dim3(256), 0, stream, a);
      return;
    }
    if (ppw == 64) {
      hipLaunchKernelGGL((icp_linearize_kernel<K, BINARY, NOFF, 256, SHARD, 4>), grid, dim3(256), 0, stream, a);
      return;
    }
  }
  hipLaunchKernelGGL((icp_linearize_kernel<K, BINARY, NOFF, 256, SHARD>), grid, dim3(256), 0, stream, a);
}
template <int NOFF>
static void launch_linearize_n(const IcpArgs & a, bool binary, hipStream_t stream)
{
  const bool shard = a.n_dev != nullptr;  // a map-sharded factor's launch (shard_api.hip): the SHARD instantiation
  const int ppw = linearize_class(a.n, a.k, shard);
#define MH_K3_GO(KK, BIN)                                       \
  do {                                                          \
    if (shard)                                                  \
      launch_linearize_k<KK, BIN, NOFF, true>(a, ppw, stream);  \
    else                                                        \
      launch_linearize_k<KK, BIN, NOFF, false>(a, ppw, stream); \
  } while (0)
  if (a.k == 5) {
    if (binary)
      MH_K3_GO(5, true);
    else
      MH_K3_GO(5, false);
  } else {
    if (binary)
      MH_K3_GO(8, true);
    else
      MH_K3_GO(8, false);
  }
#undef MH_K3_GO
}

hipError_t launch_linearize(const IcpArgs & a, bool binary, hipStream_t stream)
{
  if (a.map.n_off <= 7)
    launch_linearize_n<7>(a, binary, stream);
  else if (a.map.n_off == 19)
    launch_linearize_n<19>(a, binary, stream);
  else
    launch_linearize_n<27>(a, binary, stream);
  return hipGetLastError();
}

hipError_t launch_localizability(const LocArgs & a0, hipStream_t stream)
{
  LocArgs a = a0;
  const bool shard = a.n_dev != nullptr;
  const int ppw = linearize_class(a.n, a.k, shard);
  a.chunks_per_block = loc_chunks(shard, ppw);
  const dim3 grid(class_loc_grid(a.n, ppw, shard));
  if (ppw <= 256) {
    if (shard)
      hipLaunchKernelGGL((icp_localizability_kernel<256, true>), grid, dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((icp_localizability_kernel<256, false>), grid, dim3(loc_threads(256, false)), 0, stream, a);
  } else {
    if (shard)
      hipLaunchKernelGGL((icp_localizability_kernel<kThreads, true>), grid, dim3(kThreads), 0, stream, a);
    else
      hipLaunchKernelGGL((icp_localizability_kernel<kThreads, false>), grid, dim3(loc_threads(kThreads, false)), 0, stream, a);
  }
  return hipGetLastError();
}

// ---- batched launches: all factors share (k == 5 or not, binary, neighbour mode, class) -----------------------
template <int NOFF, int TPB>
static void launch_linearize_batch_nt(const IcpArgs * d_args, const int * d_start, int n_factors, int total_grid, int k,
                                      bool binary, hipStream_t stream)
{
  const dim3 grid(total_grid), block(TPB);
  if (k == 5) {
    if (binary)
      hipLaunchKernelGGL((icp_linearize_batch_kernel<5, true, NOFF, TPB>), grid, block, 0, stream, d_args, d_start, n_factors);
    else
      hipLaunchKernelGGL((icp_linearize_batch_kernel<5, false, NOFF, TPB>), grid, block, 0, stream, d_args, d_start, n_factors);
  } else {
    if (binary)
      hipLaunchKernelGGL((icp_linearize_batch_kernel<8, true, NOFF, TPB>), grid, block, 0, stream, d_args, d_start, n_factors);
    else
      hipLaunchKernelGGL((icp_linearize_batch_kernel<8, false, NOFF, TPB>), grid, block, 0, stream, d_args, d_start, n_factors);
  }
}

// (windows of more than kBatchInline factors: the one-lane-per-point classes only — batch_class never hands them another)
hipError_t launch_linearize_batch(const IcpArgs * d_args, const int * d_start, int n_factors, int total_grid, int ppw, int k,
                                  int n_off, bool binary, hipStream_t stream)
{
#define MH_BATCH_TPB(NOFF)                                                                                  \
  do {                                                                                                      \
    if (ppw == 256)                                                                                         \
      launch_linearize_batch_nt<NOFF, 256>(d_args, d_start, n_factors, total_grid, k, binary, stream);      \
    else                                                                                                    \
      launch_linearize_batch_nt<NOFF, kThreads>(d_args, d_start, n_factors, total_grid, k, binary, stream); \
  } while (0)
  if (ppw != 256 && ppw != kThreads) return hipErrorInvalidValue;
  if (n_off <= 7)
    MH_BATCH_TPB(7);
  else if (n_off == 19)
    MH_BATCH_TPB(19);
  else
    MH_BATCH_TPB(27);
#undef MH_BATCH_TPB
  return hipGetLastError();
}

hipError_t launch_localizability_batch(const LocArgs * d_args, const int * d_start, int n_factors, int total_grid, int ppw,
                                       hipStream_t stream)
{
  if (ppw <= 256)
    hipLaunchKernelGGL(icp_localizability_batch_kernel<256>, dim3(total_grid), dim3(256), 0, stream, d_args, d_start, n_factors);
  else
    hipLaunchKernelGGL(icp_localizability_batch_kernel<kThreads>, dim3(total_grid), dim3(loc_tpb(kThreads, false)), 0, stream, d_args, d_start,
                       n_factors);
  return hipGetLastError();
}

// The inline form also serves the map-sharded factors' batch (shard_api.hip): shard = every argument block carries n_dev.
template <int K, bool BINARY, int NOFF, bool SHARD>
static void launch_linearize_batch_inline_k(const BatchInline<IcpArgs> & blk, int total_grid, int ppw, hipStream_t stream)
{
  const dim3 grid(total_grid);
  if (ppw == 512) {
    hipLaunchKernelGGL((icp_linearize_batch_inline_kernel<K, BINARY, NOFF, kThreads, SHARD>), grid, dim3(kThreads), 0, stream, blk);
    return;
  }
  if constexpr (K == 5 && !SHARD) {
    if (ppw == 128) {
      hipLaunchKernelGGL((icp_linearize_batch_inline_kernel<K, BINARY, NOFF, 256, SHARD, 2>), grid, dim3(256), 0, stream, blk);
      return;
    }
    if (ppw == 64) {
      hipLaunchKernelGGL((icp_linearize_batch_inline_kernel<K, BINARY, NOFF, 256, SHARD, 4>), grid, dim3(256), 0, stream, blk);
      return;
    }
  }
  hipLaunchKernelGGL((icp_linearize_batch_inline_kernel<K, BINARY, NOFF, 256, SHARD>), grid, dim3(256), 0, stream, blk);
}

hipError_t launch_linearize_batch_inline(const BatchInline<IcpArgs> & blk, int total_grid, int ppw, int k, int n_off, bool binary,
                                         hipStream_t stream, bool shard)
{
#define MH_BATCH_GO(KK, BIN, NOFF)                                                             \
  do {                                                                                         \
    if (shard)                                                                                 \
      launch_linearize_batch_inline_k<KK, BIN, NOFF, true>(blk, total_grid, ppw, stream);      \
    else                                                                                       \
      launch_linearize_batch_inline_k<KK, BIN, NOFF, false>(blk, total_grid, ppw, stream);     \
  } while (0)
#define MH_BATCH_KB(NOFF)          \
  do {                             \
    if (k == 5) {                  \
      if (binary)                  \
        MH_BATCH_GO(5, true, NOFF);  \
      else                         \
        MH_BATCH_GO(5, false, NOFF); \
    } else {                       \
      if (binary)                  \
        MH_BATCH_GO(8, true, NOFF);  \
      else                         \
        MH_BATCH_GO(8, false, NOFF); \
    }                              \
  } while (0)
  if (n_off <= 7)
    MH_BATCH_KB(7);
  else if (n_off == 19)
    MH_BATCH_KB(19);
  else
    MH_BATCH_KB(27);
#undef MH_BATCH_KB
#undef MH_BATCH_GO
  return hipGetLastError();
}

hipError_t launch_localizability_batch_inline(const BatchInline<LocArgs> & blk, int total_grid, int ppw, hipStream_t stream, bool shard)
{
  const dim3 grid(total_grid);
  if (ppw <= 256) {
    if (shard)
      hipLaunchKernelGGL((icp_localizability_batch_inline_kernel<256, true>), grid, dim3(256), 0, stream, blk);
    else
      hipLaunchKernelGGL((icp_localizability_batch_inline_kernel<256, false>), grid, dim3(256), 0, stream, blk);
  } else {
    if (shard)
      hipLaunchKernelGGL((icp_localizability_batch_inline_kernel<kThreads, true>), grid, dim3(kThreads), 0, stream, blk);
    else
      hipLaunchKernelGGL((icp_localizability_batch_inline_kernel<kThreads, false>), grid, dim3(loc_tpb(kThreads, false)), 0, stream, blk);
  }
  return hipGetLastError();
}

hipError_t launch_map_knn(const MapView & map, const double * q, int n, int k, double * pts, double * sq,
                          int32_t * found, hipStream_t stream)
{
  const int grid = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(map_knn_kernel, dim3(grid), dim3(kThreads), 0, stream, map, q, n, k, pts, sq, found);
  return hipGetLastError();
}

}  // namespace mh
