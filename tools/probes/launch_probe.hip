// Host cost of a kernel launch and the device-side gap between consecutive launches on one stream, for a kernel with an 8-byte and with a
// 256-byte argument block (the GEMM kernels take their GemmP struct by value).  build: hipcc -O3 --offload-arch=gfx950 -o launch_probe launch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { double v[32]; };
__global__ void k_small(double* p) { if (p && threadIdx.x == 1000) p[0] = 1.0; }
__global__ void k_big(Big b, double* p) { if (p && threadIdx.x == 1000) p[0] = b.v[3]; }
__global__ void k_work(double* p, int n) { double s = 0; for (int i = 0; i < n; ++i) s += p[(threadIdx.x + i) & 255]; if (s == 12345.678) p[0] = s; }
int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  double* d; (void)hipMalloc(&d, 4096); (void)hipMemset(d, 0, 4096);
  Big b{};
  const int N = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d);
    auto t1 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(st);
    auto t2 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, b, d);
    auto t3 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(st);
    auto t4 = std::chrono::steady_clock::now();
    // kernels of ~20 us each: does the host keep ahead?
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, st, d, 3000);
    auto t5 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(st);
    auto t6 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b2) { return std::chrono::duration<double, std::micro>(b2 - a).count(); };
    if (rep == 1)
      std::printf("empty kernel, 8-byte args: host %.2f us per launch, %.2f us per launch until drained; 256-byte args: host %.2f us, drained %.2f us; "
                  "working kernel: host %.2f us per launch, %.2f us per kernel end to end\n",
                  us(t0, t1) / N, us(t0, t2) / N, us(t2, t3) / N, us(t2, t4) / N, us(t4, t5) / N, us(t4, t6) / N);
  }
  return 0;
}
