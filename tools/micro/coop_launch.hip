// Microbenchmark: what a cooperative launch + one grid-wide barrier costs on MI355X, against two ordinary back-to-back
// launches (the K3 -> K4 hand-off).  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/coop_launch.hip -o /tmp/coop_launch
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(512) void work_a(float * x, int spin)
{
  float v = x[blockIdx.x * 512 + threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  x[blockIdx.x * 512 + threadIdx.x] = v;
}
__global__ __launch_bounds__(512) void work_b(float * x, unsigned * ticket)
{
  float v = x[blockIdx.x * 512 + threadIdx.x] + 1.f;
  x[blockIdx.x * 512 + threadIdx.x] = v;
  if (threadIdx.x == 0) atomicAdd(ticket, 1u);
}
__global__ __launch_bounds__(512) void fused_cg(float * x, unsigned * ticket, int spin)
{
  float v = x[blockIdx.x * 512 + threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  cg::this_grid().sync();
  v += 1.f;
  x[blockIdx.x * 512 + threadIdx.x] = v;
  if (threadIdx.x == 0) atomicAdd(ticket, 1u);
}
// hand-rolled barrier on a generation counter (what the fused K3 tail would use: the ticket it already takes)
__global__ __launch_bounds__(512) void fused_manual(float * x, unsigned * bar, unsigned * ticket, int spin, unsigned gen)
{
  float v = x[blockIdx.x * 512 + threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen * gridDim.x) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  v += 1.f;
  x[blockIdx.x * 512 + threadIdx.x] = v;
  if (threadIdx.x == 0) atomicAdd(ticket, 1u);
}

int main()
{
  const int blocks = 256, iters = 400, spin = 4000;
  float * x;
  unsigned * t;
  (void)hipMalloc(&x, blocks * 512 * sizeof(float));
  (void)hipMalloc(&t, 64);
  (void)hipMemset(x, 0, blocks * 512 * sizeof(float));
  (void)hipMemset(t, 0, 64);
  hipStream_t s;
  (void)hipStreamCreate(&s);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int rep = 0; rep < 2; ++rep) {
    auto a = now();
    for (int i = 0; i < iters; ++i) {
      hipLaunchKernelGGL(work_a, dim3(blocks), dim3(512), 0, s, x, spin);
      hipLaunchKernelGGL(work_b, dim3(blocks), dim3(512), 0, s, x, t);
    }
    (void)hipStreamSynchronize(s);
    auto b = now();
    printf("two launches      : %.2f us / iteration\n", us(a, b) / iters);
    a = now();
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(work_a, dim3(blocks), dim3(512), 0, s, x, spin);
    (void)hipStreamSynchronize(s);
    b = now();
    printf("work_a alone      : %.2f us / iteration\n", us(a, b) / iters);
    a = now();
    int sp = spin;
    for (int i = 0; i < iters; ++i) {
      void * args[] = {&x, &t, &sp};
      hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<void *>(fused_cg), dim3(blocks), dim3(512), args, 0, s);
      if (e != hipSuccess) {
        printf("cooperative launch failed: %s\n", hipGetErrorString(e));
        return 1;
      }
    }
    (void)hipStreamSynchronize(s);
    b = now();
    printf("cooperative + sync: %.2f us / iteration\n", us(a, b) / iters);
    (void)hipMemset(t, 0, 64);
    a = now();
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fused_manual, dim3(blocks), dim3(512), 0, s, x, t + 8, t, spin, static_cast<unsigned>(i + 1));
    (void)hipStreamSynchronize(s);
    b = now();
    printf("manual barrier    : %.2f us / iteration (plain launch; safe only when all %d blocks are co-resident)\n", us(a, b) / iters, blocks);
  }
  return 0;
}
