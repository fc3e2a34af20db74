// Spatial ordering of the source cloud, done once per factor (ICPFactor ctor,
// geometric_factor.hpp:119-156 copies the cloud; here the copy is also re-ordered).
//
// Why: icp_linearize_kernel runs one lane per point and is bound by L1/TA line throughput — a
// wave-wide load costs about one cycle per DISTINCT cache line it touches.  64 consecutive points of
// an Ouster scan row span ~8 m (~20 voxels, ~23 distinct lines per load, measured); after sorting by
// the Morton code of the point's 0.25 m cell the 64 lanes of a wave sit in a ~1.5 m patch, share
// voxels, walk the same buckets in lock-step and their identical addresses coalesce.  Per-point
// outputs are un-permuted in the getters, the Hessian sums are order-free.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "icp_device.hpp"

namespace mh
{
namespace
{
constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__global__ __launch_bounds__(kThreads) void morton_keys_kernel(const float4 * xyz, int n, float inv_cell, uint32_t * keys,
                                                               uint32_t * vals)
{
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const float4 p = xyz[i];
    const int qx = min(1023, max(0, static_cast<int>(floorf(p.x * inv_cell)) + 512));
    const int qy = min(1023, max(0, static_cast<int>(floorf(p.y * inv_cell)) + 512));
    const int qz = min(1023, max(0, static_cast<int>(floorf(p.z * inv_cell)) + 512));
    keys[i] = (spread10(qx) << 2) | (spread10(qy) << 1) | spread10(qz);
    vals[i] = static_cast<uint32_t>(i);
  }
}

// the same from the 32-byte point records (two float4 each): packs xyz on the way
__global__ __launch_bounds__(kThreads) void pack_morton_kernel(const float4 * pts2, int n, float inv_cell, float4 * xyz, uint32_t * keys,
                                                               uint32_t * vals, uint32_t * zero_a, int n_zero_a, uint32_t * zero_b, int n_zero_b)
{
  if (blockIdx.x == 0) {  // the factor's ticket / result block start at zero
    for (int i = threadIdx.x; i < n_zero_a; i += kThreads) zero_a[i] = 0u;
    for (int i = threadIdx.x; i < n_zero_b; i += kThreads) zero_b[i] = 0u;
  }
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const float4 p = pts2[2 * i];
    xyz[i] = p;
    const int qx = min(1023, max(0, static_cast<int>(floorf(p.x * inv_cell)) + 512));
    const int qy = min(1023, max(0, static_cast<int>(floorf(p.y * inv_cell)) + 512));
    const int qz = min(1023, max(0, static_cast<int>(floorf(p.z * inv_cell)) + 512));
    keys[i] = (spread10(qx) << 2) | (spread10(qy) << 1) | spread10(qz);
    vals[i] = static_cast<uint32_t>(i);
  }
}

__global__ __launch_bounds__(kThreads) void gather_xyz_kernel(const float4 * in, const uint32_t * perm, int n, float4 * out)
{
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) out[i] = in[perm[i]];
}

// out[perm[i]] = in[i]
__global__ __launch_bounds__(kThreads) void unpermute_state_kernel(const uint32_t * perm, int n, const int32_t * st_in,
                                                                   const double * mean_in, const double * nrm_in,
                                                                   int32_t * st_out, double * mean_out, double * nrm_out)
{
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const uint32_t o = perm[i];
    if (st_out) st_out[o] = st_in[i];
    if (mean_out) {
      mean_out[3 * o] = mean_in[3 * i];
      mean_out[3 * o + 1] = mean_in[3 * i + 1];
      mean_out[3 * o + 2] = mean_in[3 * i + 2];
    }
    if (nrm_out) {
      nrm_out[3 * o] = nrm_in[3 * i];
      nrm_out[3 * o + 1] = nrm_in[3 * i + 1];
      nrm_out[3 * o + 2] = nrm_in[3 * i + 2];
    }
  }
}

int grid_for(int n) { return max(1, min((n + kThreads - 1) / kThreads, 2048)); }
}  // namespace

size_t order_temp_bytes(int n)
{
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr),
                                  static_cast<uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr),
                                  static_cast<size_t>(n), 0, 30, hipStream_t(nullptr));
  return bytes;
}

// xyz_in (n points, original order) -> perm (sorted position -> original index), xyz_out (sorted).
// keys[2n], vals[n] and temp are caller-provided scratch.
hipError_t launch_spatial_order(const float4 * xyz_in, int n, float cell, uint32_t * keys2, uint32_t * vals, void * temp,
                                size_t temp_bytes, uint32_t * perm, float4 * xyz_out, hipStream_t stream)
{
  hipLaunchKernelGGL(morton_keys_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, xyz_in, n, 1.0f / cell, keys2, vals);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys2, keys2 + n, vals, perm, static_cast<size_t>(n), 0, 30,
                                           stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gather_xyz_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, xyz_in, perm, n, xyz_out);
  return hipGetLastError();
}

// Factor creation: d_pts (n 32-byte records, device) -> perm (sorted position -> original index) and xyz_out (packed,
// sorted).  scratch: source_order_scratch_bytes(n) bytes, stream-ordered.  zero_a / zero_b: dword ranges cleared on the way.
size_t source_order_scratch_bytes(int n)
{
  const size_t m = static_cast<size_t>(n);
  return ((m * sizeof(float4) + 255) & ~size_t(255)) + ((3 * m * sizeof(uint32_t) + 255) & ~size_t(255)) + order_temp_bytes(n) + 256;
}
hipError_t launch_source_order(const mh_point32 * d_pts, int n, float cell, void * scratch, uint32_t * perm, float4 * xyz_out,
                               uint32_t * zero_a, int n_zero_a, uint32_t * zero_b, int n_zero_b, hipStream_t stream)
{
  const size_t m = static_cast<size_t>(n);
  char * sc = static_cast<char *>(scratch);
  float4 * xyz_tmp = reinterpret_cast<float4 *>(sc);
  sc += (m * sizeof(float4) + 255) & ~size_t(255);
  uint32_t * keys2 = reinterpret_cast<uint32_t *>(sc), * vals = keys2 + 2 * m;
  sc += (3 * m * sizeof(uint32_t) + 255) & ~size_t(255);
  size_t tb = order_temp_bytes(n);
  hipLaunchKernelGGL(pack_morton_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, reinterpret_cast<const float4 *>(d_pts), n,
                     1.0f / cell, xyz_tmp, keys2, vals, zero_a, n_zero_a, zero_b, n_zero_b);
  hipError_t e = rocprim::radix_sort_pairs(sc, tb, keys2, keys2 + m, vals, perm, m, 0, 30, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gather_xyz_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, xyz_tmp, perm, n, xyz_out);
  return hipGetLastError();
}

hipError_t launch_unpermute_state(const uint32_t * perm, int n, const int32_t * st_in, const double * mean_in,
                                  const double * nrm_in, int32_t * st_out, double * mean_out, double * nrm_out,
                                  hipStream_t stream)
{
  hipLaunchKernelGGL(unpermute_state_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, perm, n, st_in, mean_in, nrm_in,
                     st_out, mean_out, nrm_out);
  return hipGetLastError();
}

}  // namespace mh
