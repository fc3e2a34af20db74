// Map-sharded factor: structs and launcher declarations shared by shard_kernels.hip and the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "icp_device.hpp"

namespace mh
{
struct ShardPose
{
  double R[9], t[3];  // delta pose T_tgt^-1 * T_src of the factor
};

// the per-point arrays of an ICP factor that travel with a point
struct ShardArrays
{
  float4 * src;
  double * q_da;
  double * mean;
  double * normal;
  int32_t * status;
  unsigned long long * origin;  // (rank that first held the point) << 32 | its index there
};

// what a point that changes owner takes along: 112 bytes
struct ShardRecord
{
  float4 src;
  double q_da[3], mean[3], normal[3];
  int32_t status;
  uint32_t pad;
  unsigned long long origin;
};
static_assert(sizeof(ShardRecord) == 112, "ShardRecord layout");

size_t shard_temp_bytes(size_t n);
hipError_t launch_shard_filter(const float * xyz, uint32_t n, uint32_t stride, double inv_leaf, uint32_t world, uint32_t rank, int log2,
                               uint32_t * flags, uint32_t * pos, float * out, uint32_t * n_out, void * temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_shard_plan(const ShardPose & P, const float4 * src, uint32_t n, double inv_leaf, uint32_t world, uint32_t rank, int log2,
                             uint32_t * keys_a, uint32_t * keys_b, uint32_t * idx_a, uint32_t * idx_b, uint32_t * counts, void * temp,
                             size_t temp_bytes, hipStream_t stream);
hipError_t launch_shard_origin(unsigned long long * origin, uint32_t n, uint32_t rank, hipStream_t stream);
hipError_t launch_shard_pack(const ShardArrays & in, const ShardArrays & out, const uint32_t * sorted_idx, uint32_t n, uint32_t n_movers,
                             ShardRecord * send, hipStream_t stream);
hipError_t launch_shard_unpack(const ShardArrays & a, uint32_t base, const ShardRecord * recv, uint32_t n_recv, hipStream_t stream);
// 28 Hessian sums + 4 counters of a unary factor's K3 as 32 doubles (what the all-reduce carries)
hipError_t launch_shard_pack_sums(const DeviceResult * r, double * out32, hipStream_t stream);
// eigenbases of the rot / trans blocks of the GLOBAL H (computeLocalizability, include/mimosa/utils.hpp:308-313) -> 18 doubles
hipError_t launch_shard_eig(const double * global32, double * eig18, hipStream_t stream);
// 6 component localizabilities + 9 histogram counts as 16 doubles
hipError_t launch_shard_pack_loc(const DeviceResult * r, double * out16, hipStream_t stream);

}  // namespace mh
