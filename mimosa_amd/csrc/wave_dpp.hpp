// Wave-wide sums by DPP (K3's counters, K4's component sums, the photometric factor's NCC sums).
//
// All 64 lanes must be active.  Two quad permutes, two row mirrors, then row_bcast15 / row_bcast31 carry the 16-lane row sums
// across rows; the total lands in lane 63 and is handed to every lane by v_readlane.  N independent chains per step for the
// scheduler to interleave.  (A butterfly of __shfl_xor on doubles is 12 ds_bpermute per sum through the CU's LDS pipeline.)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mh
{
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double wdpp_pull_f64(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int N, int CTRL, int ROW_MASK>
__device__ __forceinline__ void wdpp_step_f64(double (&v)[N])
{
  double t[N];
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = wdpp_pull_f64<CTRL, ROW_MASK>(v[j]);
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] += t[j];
}
// v[j] <- the sum of v[j] over the wave, in LANE 63 (the other lanes hold partial sums).  Each step moves the two halves of a
// double by DPP and adds in fp64 (lanes a step does not write add +0.0); the order of the additions is fixed by the lane
// pattern: deterministic.
template <int N>
__device__ __forceinline__ void wave_sum_to_lane63_f64(double (&v)[N])
{
  wdpp_step_f64<N, 0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  wdpp_step_f64<N, 0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  wdpp_step_f64<N, 0x141, 0xF>(v);  // row_half_mirror
  wdpp_step_f64<N, 0x140, 0xF>(v);  // row_mirror
  wdpp_step_f64<N, 0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
  wdpp_step_f64<N, 0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3
}
// ... in EVERY lane
template <int N>
__device__ __forceinline__ void wave_allsum_f64(double (&v)[N])
{
  wave_sum_to_lane63_f64<N>(v);
#pragma unroll
  for (int j = 0; j < N; ++j)
    v[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v[j]), 63), __builtin_amdgcn_readlane(__double2loint(v[j]), 63));
}
__device__ __forceinline__ double wave_allsum_f64(double x)
{
  double v[1] = {x};
  wave_allsum_f64<1>(v);
  return v[0];
}
// The same for N 32-bit words (total in lane 63).  (Leaving counters to atomicAdd on LDS makes the compiler aggregate with a 64-trip
// scalar v_readlane loop per counter: ~450 dependent SALU instructions each, 11 k of K3's 71 k cycles per wave in round 1.)
template <int N, int CTRL, int ROW_MASK>
__device__ __forceinline__ void wdpp_step_u32(uint32_t (&v)[N])
{
  uint32_t t[N];
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v[j]), CTRL, ROW_MASK, 0xF, false));
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] += t[j];
}
template <int N>
__device__ __forceinline__ void wave_sum_to_lane63_u32(uint32_t (&v)[N])
{
  wdpp_step_u32<N, 0xB1, 0xF>(v);
  wdpp_step_u32<N, 0x4E, 0xF>(v);
  wdpp_step_u32<N, 0x141, 0xF>(v);
  wdpp_step_u32<N, 0x140, 0xF>(v);
  wdpp_step_u32<N, 0x142, 0xA>(v);
  wdpp_step_u32<N, 0x143, 0xC>(v);
}
}  // namespace mh
