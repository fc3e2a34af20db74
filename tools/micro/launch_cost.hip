// CPU cost of one kernel launch with a 456-byte argument block (K3's IcpArgs is that size), by launch API:
//   hipLaunchKernelGGL | hipModuleLaunchKernel with kernelParams | hipModuleLaunchKernel with HIP_LAUNCH_PARAM_BUFFER_POINTER
// and the time from the launch call to the kernel's first store being visible in mapped host memory (launch latency of an idle stream).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>

struct Big
{
  double d[56];
  unsigned int * flag;
  unsigned int seq;
};
__global__ void probe(const Big a)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned int * h;
  hipHostMalloc(reinterpret_cast<void **>(&h), 64, hipHostMallocMapped);
  unsigned int * d;
  hipHostGetDevicePointer(reinterpret_cast<void **>(&d), h, 0);
  Big a;
  std::memset(&a, 0, sizeof(a));
  a.flag = d;
  hipFunction_t fn;
  if (hipGetFuncBySymbol(&fn, reinterpret_cast<const void *>(probe)) != hipSuccess) { std::printf("hipGetFuncBySymbol failed\n"); return 1; }
  unsigned int seq = 0;
  for (int mode = 0; mode < 3; ++mode) {
    double call = 0, lat = 0;
    const int N = 2000;
    for (int i = 0; i < N + 50; ++i) {
      hipStreamSynchronize(s);  // idle stream, as for a synchronous call
      a.seq = ++seq;
      const double t0 = now_us();
      if (mode == 0) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, s, a);
      } else if (mode == 1) {
        void * params[] = {&a};
        hipModuleLaunchKernel(fn, 256, 1, 1, 512, 1, 1, 0, s, params, nullptr);
      } else {
        size_t sz = sizeof(a);
        void * extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        hipModuleLaunchKernel(fn, 256, 1, 1, 512, 1, 1, 0, s, nullptr, extra);
      }
      const double t1 = now_us();
      while (__atomic_load_n(h, __ATOMIC_ACQUIRE) != seq) {}
      const double t2 = now_us();
      if (i >= 50) {
        call += t1 - t0;
        lat += t2 - t0;
      }
    }
    std::printf("%-58s call %.2f us, launch -> first store seen by the host %.2f us\n",
                mode == 0 ? "hipLaunchKernelGGL" : (mode == 1 ? "hipModuleLaunchKernel (kernelParams)" : "hipModuleLaunchKernel (param buffer)"), call / N, lat / N);
  }
  // back to back (stream not idle): cost per call when 16 launches are queued before the wait
  for (int mode = 0; mode < 2; ++mode) {
    const int N = 200;
    double call = 0;
    for (int i = 0; i < N; ++i) {
      hipStreamSynchronize(s);
      const double t0 = now_us();
      for (int k = 0; k < 16; ++k) {
        a.seq = ++seq;
        if (mode == 0) {
          hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, s, a);
        } else {
          size_t sz = sizeof(a);
          void * extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
          hipModuleLaunchKernel(fn, 256, 1, 1, 512, 1, 1, 0, s, nullptr, extra);
        }
      }
      call += (now_us() - t0) / 16;
    }
    std::printf("%-58s %.2f us per call, 16 queued back to back\n", mode == 0 ? "hipLaunchKernelGGL" : "hipModuleLaunchKernel (param buffer)", call / N);
  }
  return 0;
}
