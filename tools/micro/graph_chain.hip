// Microbenchmark: a chain of 12 small dependent kernels (the shape of the photometric preprocess / the scan front end:
// 512 KiB images, 4-10 us each) launched one by one on a stream versus replayed as a hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/graph_chain.hip -o /tmp/graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ __launch_bounds__(256) void stage(const float * in, float * out, int n, int work)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = in[i] + in[(i + 1) % n];
  for (int k = 0; k < work; ++k) v = v * 1.0001f + 0.25f;
  out[i] = v;
}

int main()
{
  const int n = 128 * 1024, stages = 12, iters = 300;
  float *a, *b;
  (void)hipMalloc(&a, n * sizeof(float));
  (void)hipMalloc(&b, n * sizeof(float));
  (void)hipMemset(a, 0, n * sizeof(float));
  hipStream_t s;
  (void)hipStreamCreate(&s);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto t0, auto t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
  auto chain = [&](hipStream_t st) {
    for (int k = 0; k < stages; ++k) hipLaunchKernelGGL(stage, dim3(n / 256), dim3(256), 0, st, k % 2 ? b : a, k % 2 ? a : b, n, 40);
  };
  hipGraph_t g;
  hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  chain(s);
  (void)hipStreamEndCapture(s, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
    printf("graph instantiate failed\n");
    return 1;
  }
  for (int rep = 0; rep < 2; ++rep) {
    // latency of ONE chain, host waits for it (what a front-end call does)
    double t_stream = 0, t_graph = 0;
    for (int i = 0; i < iters; ++i) {
      auto t0 = now();
      chain(s);
      (void)hipStreamSynchronize(s);
      t_stream += us(t0, now());
    }
    for (int i = 0; i < iters; ++i) {
      auto t0 = now();
      (void)hipGraphLaunch(ge, s);
      (void)hipStreamSynchronize(s);
      t_graph += us(t0, now());
    }
    printf("one chain of %d kernels + sync: stream launches %.1f us, hipGraphLaunch %.1f us\n", stages, t_stream / iters, t_graph / iters);
    // throughput of back-to-back chains
    auto t0 = now();
    for (int i = 0; i < iters; ++i) chain(s);
    (void)hipStreamSynchronize(s);
    auto t1 = now();
    for (int i = 0; i < iters; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    auto t2 = now();
    printf("back to back: stream launches %.1f us per chain, hipGraphLaunch %.1f us per chain\n", us(t0, t1) / iters, us(t1, t2) / iters);
  }
  return 0;
}
