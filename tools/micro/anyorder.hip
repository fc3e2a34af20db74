// Does hipExtAnyOrderLaunch clear the barrier bit on gfx950?  Two independent spin kernels on ONE stream: serial = 2 T, overlapped = T.
// (hip_ext.h says the flag "is not supported on AMD GFX9xx boards": measured here rather than believed.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/anyorder.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long cycles, unsigned int * sink)
{
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1u;
}

int main()
{
  hipStream_t s;
  hipStreamCreate(&s);
  unsigned int * d;
  hipMalloc(&d, 4);
  const unsigned long long cyc = 100000ull;  // s_memtime / readcyclecounter ticks (100 MHz REFCLK: 1 ms; shader clock: ~45 us)
  void * args[] = {const_cast<unsigned long long *>(&cyc), &d};
  for (int mode = 0; mode < 3; ++mode) {
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipStreamSynchronize(s);
      const auto t0 = std::chrono::steady_clock::now();
      hipExtLaunchKernel(reinterpret_cast<const void *>(spin), dim3(64), dim3(64), args, 0, s, nullptr, nullptr, 0);
      if (mode > 0) hipExtLaunchKernel(reinterpret_cast<const void *>(spin), dim3(64), dim3(64), args, 0, s, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0);
      hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us < best) best = us;
    }
    std::printf("%s: %.1f us\n", mode == 0 ? "one kernel" : (mode == 1 ? "two kernels, in order" : "two kernels, second with hipExtAnyOrderLaunch"), best);
  }
  return 0;
}
