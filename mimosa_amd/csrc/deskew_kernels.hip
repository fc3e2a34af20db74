// HIP kernels for the f32 rigid transforms of the scan (kernels K1 / K2 of SURVEY.md §2.3).
//
// Reference: Manager::deskewPoints hot loop src/lidar/manager.cpp:496-509 (per-timestamp-group
// pose, p <- R p + t in float), Geometric::preprocess body transform src/lidar/geometric.cpp:154-161,
// Geometric::updateMap world transform src/lidar/geometric.cpp:483-490.
//
// The reference is built for baseline x86-64 (no FMA) and Eigen evaluates the 3x3 * 3x1 float
// product coefficient-wise as r0*x + (r1*y + r2*z), then adds t.  This file is compiled with
// -ffp-contract=off so the results are bit-identical: a 1-ulp difference is harmless for the ICP
// residual but can move a point across a voxel boundary in the down-sampler / map insert.
// Pure streaming work: 32 B in, 12 B (xyz) out per point; the timestamp -> group lookup is a binary search over the
// <= 4096 distinct timestamps staged in LDS (16 KiB); the group's pose (48 B) comes through L1 / L2 (staging the whole
// pose table per block cost more than it saved: 48 KiB x 512 blocks of extra reads for a 4 MiB cloud).
#include <hip/hip_runtime.h>

#include "icp_device.hpp"

namespace mh
{
namespace
{
constexpr int kThreads = 256;
constexpr int kMaxGroupsLds = 4096;

__device__ __forceinline__ void xform(const float * P, float & x, float & y, float & z)
{
  const float px = x, py = y, pz = z;
  x = (P[0] * px + (P[1] * py + P[2] * pz)) + P[9];
  y = (P[3] * px + (P[4] * py + P[5] * pz)) + P[10];
  z = (P[6] * px + (P[7] * py + P[8] * pz)) + P[11];
}

// One thread per point.  A point record is two float4: {x,y,z,pad} {intensity,t,idx,range}.
__global__ __launch_bounds__(kThreads) void deskew_kernel(float4 * pts, int n, const uint32_t * unique_ns,
                                                           const float * Rt12, int n_groups, const float * body,
                                                           int use_lds)
{
  __shared__ uint32_t s_ns[kMaxGroupsLds];
  __shared__ float s_body[12];
  if (use_lds)
    for (int i = threadIdx.x; i < n_groups; i += kThreads) s_ns[i] = unique_ns[i];
  if (body && threadIdx.x < 12) s_body[threadIdx.x] = body[threadIdx.x];
  __syncthreads();
  const uint32_t * ns = use_lds ? s_ns : unique_ns;
  const float * poses = Rt12;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    float4 a = pts[2 * i];
    const float4 b = pts[2 * i + 1];
    const uint32_t t = __float_as_uint(b.y);
    // lower_bound(unique_ns, t)
    int lo = 0, hi = n_groups;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ns[mid] < t)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < n_groups && ns[lo] == t) xform(poses + 12 * lo, a.x, a.y, a.z);
    if (body) xform(s_body, a.x, a.y, a.z);
    pts[2 * i] = a;
  }
}

__global__ __launch_bounds__(kThreads) void transform_kernel(float4 * pts, int n, const float * Rt12)
{
  __shared__ float s_p[12];
  if (threadIdx.x < 12) s_p[threadIdx.x] = Rt12[threadIdx.x];
  __syncthreads();
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    float4 a = pts[2 * i];
    xform(s_p, a.x, a.y, a.z);
    pts[2 * i] = a;
  }
}

// mh_point32[n] -> float4 xyz[n] (the 16-byte-per-point source layout the ICP kernel reads)
__global__ __launch_bounds__(kThreads) void pack_xyz_kernel(const float4 * pts, int n, float4 * xyz)
{
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) xyz[i] = pts[2 * i];
}

int grid_for(int n) { return max(1, min((n + kThreads - 1) / kThreads, 2048)); }
// Device-to-device copy of 16-byte words as a kernel: stream-ordered like any launch.  (hipMemcpyAsync device-to-device goes
// through the runtime's copy path; in the pipelined replay — a second thread uploading 4 MiB on a copy stream at the same time —
// that call was seen to BLOCK the calling thread for 0.4 ms in about a third of the processes.)
__global__ __launch_bounds__(256) void copy16_kernel(const uint4 * __restrict__ src, uint4 * __restrict__ dst, size_t n16)
{
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

hipError_t launch_copy16(const void * src, void * dst, size_t bytes, hipStream_t stream)
{
  const size_t n16 = bytes / 16;  // callers copy whole 16-byte records (mh_point32 = 32 bytes)
  if (!n16) return hipSuccess;
  const size_t blocks = (n16 + 255) / 256;
  hipLaunchKernelGGL(copy16_kernel, dim3(static_cast<unsigned int>(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream,
                     static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16);
  return hipGetLastError();
}

hipError_t launch_deskew(mh_point32 * pts, int n, const uint32_t * unique_ns, const float * Rt12, int n_groups,
                         const float * body_Rt12, hipStream_t stream)
{
  hipLaunchKernelGGL(deskew_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, reinterpret_cast<float4 *>(pts), n,
                     unique_ns, Rt12, n_groups, body_Rt12, n_groups <= kMaxGroupsLds ? 1 : 0);
  return hipGetLastError();
}
hipError_t launch_transform(mh_point32 * pts, int n, const float * Rt12, hipStream_t stream)
{
  hipLaunchKernelGGL(transform_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, reinterpret_cast<float4 *>(pts), n,
                     Rt12);
  return hipGetLastError();
}
hipError_t launch_pack_xyz(const mh_point32 * pts, int n, float4 * xyz, hipStream_t stream)
{
  hipLaunchKernelGGL(pack_xyz_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream,
                     reinterpret_cast<const float4 *>(pts), n, xyz);
  return hipGetLastError();
}
}  // namespace mh
