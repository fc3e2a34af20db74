// Kernel argument / device-result structs shared by the HIP kernels and the C-ABI implementation.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voxel_map.hpp"

namespace mh
{
constexpr int kPartialStride = 96;  // >= 91 sums of the binary factor + 4 counters

// Read-only view of the device-resident voxel map (see voxel_map.hpp for the layout).
struct MapView
{
  const int4 * table;
  const uint32_t * cells;
  const float4 * buckets;
  const uint4 * qbuckets;  // 5 x uint4 per voxel: 20 packed 3 x 10-bit points (coarse tier)
  double inv_leaf;
  uint32_t mask;
  int n_off;     // 1, 7, 19 or 27
  int mode_idx;  // row of the constant neighbour-offset table (0..3).  NOTE: no arrays in kernel-argument
                 // structs — a loop-indexed array member makes the compiler spill the whole by-value
                 // kernarg block to scratch and serialise every access behind s_waitcnt vmcnt(0).
};

// Everything linearize() returns from the device in one small D2H copy.
struct DeviceResult
{
  double sums[kPartialStride];  // upper triangle of sum v v^T, v = [J_s(6) (,J_t(6)), e]
  double loc_rot_final[3], eig_rot[9], loc_trans_final[3], eig_trans[9];
  double loc_comp[6];     // trans xyz, rot xyz
  unsigned long long n_knn, n_cand, n_fallback, n_scanned;
  unsigned int status_hist[9];
  unsigned int seq;  // host slot only: written LAST by K4 (system-scope release) = the call's sequence number
};

// Results of a plain (unsharded) factor reach the host as "flagged words": every double travels as ONE 16-byte store
// {lo, seq, hi, seq} into mapped pinned memory, each 8-byte half carrying the call's sequence number, so the host knows a
// value has arrived by looking at the value itself — no system-scope fence, no completion flag behind the data, no
// ordering between stores needed (the idea of NCCL's LL protocol).  Layout of one call's slot (LlSlot): the 28 / 91 Hessian
// sums + 4 counters, the two eigenbases K4 projected on, and K4's per-workgroup rows of 8 (6 component sums + 9 histogram
// counts, 12 bits each in two words), which the HOST folds in workgroup order — K4 has no ticket / last-block fold any more.
constexpr int kLlSums = 96;   // >= 91 + 4
constexpr int kLlEig = 32;    // 18 used
constexpr int kLlRow = 8;     // 6 component sums, then the 9 histogram counts 12 bits each: 0..4 in word 6, 5..8 in word 7 (128 B = two cache lines per row)
constexpr int kLocChunks = 4; // consecutive chunks of TPB points per K4 workgroup (plain factors): a quarter of K3's workgroups => a quarter of the rows (8: 14 us instead of 10, measured)
__host__ __device__ inline size_t ll_slot_words(int loc_grid_cap) { return static_cast<size_t>(kLlSums + kLlEig) + static_cast<size_t>(loc_grid_cap) * kLlRow; }

struct IcpArgs
{
  MapView map;
  const float4 * src;  // source cloud xyz (w unused), 16 B / point
  int n;
  int k;
  int cold;       // 1: treat the per-point state as freshly constructed (all zero)
  int use_huber;
  double R[9], t[3];  // delta pose  T_tgt^-1 * T_src
  double da_thresh, max_d2, plane_valid, sigma, huber;
  double inv_sigma;  // 1 / sigma, divided on the host: the kernel whitens by multiplication
  double inv_huber;  // 1 / huber, likewise (the Huber weight is one reciprocal square root of |we| / huber)
  double * q_da;
  double * mean;
  double * normal;
  int32_t * status;
  double * partials;
  unsigned int * ticket;
  DeviceResult * result;
  DeviceResult * host_result;  // mapped pinned host slot (may be null): the last block writes its part there too
  unsigned int seq;          // the call's sequence number (tags every flagged word of the call)
  int tail;                  // plain factors: 1 = no K4 follows (components switched off) — ticket, fold by the last block, sums + counters
                             // published as flagged words; 0 = K4 follows and folds the per-block rows itself: this kernel ends at its row
  uint4 * ll;                // plain factors: the call's slot in mapped pinned memory (device address)
  unsigned long long * dbg;  // MH_TIMELINE diagnostic build only, else null
  int reps;                  // MH_TIMELINE only: repeat the per-point section (warm-cache experiment)
  // map-sharded factors (shard_api.hip): the number of point slots lives on the device (arrivals are appended by a
  // kernel of the same stream, the host only knows an upper bound `n` that sizes the grid); slots whose status carries
  // kShardSkip (points that left for another rank, or could not be sent yet) are passed over untouched
  const uint32_t * n_dev = nullptr;    // null: n is exact
  double * shard_out = nullptr;        // non-null: the last block also writes the NENT sums + 4 counters here (all-reduce input)
  // plain factors whose K4 follows: what K4 needs of every point, written HERE into the factor's record (6 arrays of rec_n
  // doubles: the unwhitened, normalised Jacobian directions jr[3], jt[3] — zero unless the point is Valid — then rec_n
  // status words).  K4 follows on the same stream and reads nothing of the factor's association state.
  double * rec = nullptr;
  int rec_n = 0;
};
__host__ __device__ inline size_t loc_record_bytes(size_t n) { return n * (6 * sizeof(double) + sizeof(int32_t)); }
constexpr int32_t kShardSkip = 0x100;  // status flag bit: not this rank's point in this call

struct LocArgs
{
  DeviceResult * host_result;  // mapped pinned host slot: the last block writes loc_comp / status_hist there (no D2H copy node)
  unsigned int seq;            // published to host_result->seq after everything else: the host may spin on it
  const double * eig;          // 18 doubles: eig_rot (9) then eig_trans (9); null = derive them from result->sums (K3's Hessian sums)
  int nv;                      // 7 (unary) or 13 (binary): row length of the v v^T triangle in result->sums
  const float4 * src;
  int n;
  int chunks_per_block;  // set by the launcher
  int k;                 // num_corres_points (with n: the launch class of the single-call launcher)
  int k3_blocks;         // plain factors: K3's workgroups of this factor = rows in `partials` that every K4 workgroup folds for itself
  uint4 * ll;            // plain factors: the call's slot in mapped pinned memory
  double R[9];
  const double * normal;
  const int32_t * status;
  double * partials;
  unsigned int * ticket;
  DeviceResult * result;
  const uint32_t * n_dev = nullptr;  // as IcpArgs::n_dev
  const double * sums = nullptr;     // null: result->sums; map-sharded factors: the all-reduced (global) Hessian sums
  double * shard_out = nullptr;      // non-null: the last block also writes 6 component sums + 9 histogram counts here (16 doubles)
  const double * rec = nullptr;      // plain factors: the call's record written by K3 (IcpArgs::rec); src / normal / status are not read
  int rec_n = 0;
  unsigned long long * dbg = nullptr;  // MH_TIMELINE diagnostic build only, else null: per-wave stamps of this pass (16 words per wave)
};

// The launch class of a factor = points per K3 workgroup (icp_kernels.hip, "Launchers"): 512 / 256 (one lane per point), 128 / 64
// (2 / 4 lanes per point: small clouds of plain k = 5 factors).  total: the points of the whole window batch the factor is
// linearized in (0 = a call of its own).
int linearize_class(int n, int k, bool shard, long long total = 0);
int class_grid(int n, int ppw);                       // K3 workgroups of an n-point factor in that class (a multiple of 8)
int class_loc_grid(int n, int ppw, bool shard = false);  // K4 workgroups
int linearize_grid_max(int n);                        // the largest K3 grid any class gives n points (buffer sizing)
hipError_t launch_linearize(const IcpArgs & a, bool binary, hipStream_t stream);
hipError_t launch_localizability(const LocArgs & a, hipStream_t stream);
// Batched form: d_args / d_start live in device-visible memory (n_factors argument blocks, n_factors + 1 grid
// prefix entries); every factor's grid is class_grid(n, ppw) with one class for the launch group.
hipError_t launch_linearize_batch(const IcpArgs * d_args, const int * d_start, int n_factors, int total_grid, int tpb, int k,
                                  int n_off, bool binary, hipStream_t stream);
hipError_t launch_localizability_batch(const LocArgs * d_args, const int * d_start, int n_factors, int total_grid, int tpb,
                                       hipStream_t stream);
// Small windows (<= kBatchInline factors per launch): the argument blocks travel IN the kernel-argument segment — no
// staging copy before the launch.  start[] = exclusive prefix of the per-factor grids.
constexpr int kBatchInline = 8;
template <typename A>
struct BatchInline
{
  A a[kBatchInline];
  int start[kBatchInline + 1];
  int n;
  char pad[256];  // load_uniform reads whole 256-byte chunks: keep the last block's read inside the segment
};
// shard: every factor of the launch is a map-sharded factor's (IcpArgs::n_dev / LocArgs::n_dev set) — the SHARD instantiation.
hipError_t launch_linearize_batch_inline(const BatchInline<IcpArgs> & blk, int total_grid, int tpb, int k, int n_off, bool binary,
                                         hipStream_t stream, bool shard = false);
hipError_t launch_localizability_batch_inline(const BatchInline<LocArgs> & blk, int total_grid, int tpb, hipStream_t stream, bool shard = false);
hipError_t launch_map_knn(const MapView & map, const double * q, int n, int k, double * pts, double * sq,
                          int32_t * found, hipStream_t stream);

// order_kernels.hip
size_t order_temp_bytes(int n);
hipError_t launch_spatial_order(const float4 * xyz_in, int n, float cell, uint32_t * keys2, uint32_t * vals, void * temp,
                                size_t temp_bytes, uint32_t * perm, float4 * xyz_out, hipStream_t stream);
size_t source_order_scratch_bytes(int n);
hipError_t launch_source_order(const mh_point32 * d_pts, int n, float cell, void * scratch, uint32_t * perm, float4 * xyz_out,
                               uint32_t * zero_a, int n_zero_a, uint32_t * zero_b, int n_zero_b, hipStream_t stream);
hipError_t launch_unpermute_state(const uint32_t * perm, int n, const int32_t * st_in, const double * mean_in,
                                  const double * nrm_in, int32_t * st_out, double * mean_out, double * nrm_out,
                                  hipStream_t stream);

// deskew_kernels.hip
hipError_t launch_deskew(mh_point32 * pts, int n, const uint32_t * unique_ns, const float * Rt12, int n_groups,
                         const float * body_Rt12, hipStream_t stream);
hipError_t launch_transform(mh_point32 * pts, int n, const float * Rt12, hipStream_t stream);
hipError_t launch_pack_xyz(const mh_point32 * pts, int n, float4 * xyz, hipStream_t stream);
hipError_t launch_copy16(const void * src, void * dst, size_t bytes, hipStream_t stream);  // bytes: a multiple of 16

}  // namespace mh
