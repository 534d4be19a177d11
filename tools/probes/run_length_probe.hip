// HBM read bandwidth against the length of the contiguous runs a workgroup reads: every wave reads `run` bytes (64 lanes x 8 bytes = 512 per
// instruction, as the per-bin kernels of the inner PCG do), then jumps `stride` bytes (a row of another vector / latent) - 75 such runs in flight
// per wave like pcg_cg_a_kernel; total 2 GB per launch, far beyond the caches.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/run_length_probe tools/probes/run_length_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PER>   // consecutive 512-byte pieces per run
__global__ __launch_bounds__(256) void rd(const double* __restrict__ base, size_t run_stride_d, int runs, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t w = (size_t)blockIdx.x * 4 + wave;
  // wave w owns runs w, w + W, ... (W = all waves): rows far apart, like (slot, latent) rows of different vectors
  const size_t W = (size_t)gridDim.x * 4;
  double acc = 0.0;
  for (int r = 0; r < runs; r += 16) {
    double v[16][PER > 4 ? 1 : 1];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double* p = base + ((size_t)(r + u) * W + w) * run_stride_d + lane;
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < PER; ++q) s += p[64 * q];
      v[u][0] = s;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u][0];
  }
  if (acc == 123.456) out[0] = acc;
}
int main() {
  const size_t total = (size_t)4 << 30;
  double* buf; double* out; hipMalloc(&buf, total + (1 << 20)); hipMemset(buf, 0, total); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 2048;                               // 8192 waves
  auto run = [&](int per, auto kern) {
    // each run = per * 512 bytes, padded to a row of per * 512 + 128 bytes so that runs are not adjacent
    const size_t run_stride_d = (size_t)per * 64 + 16;
    const size_t W = (size_t)blocks * 4;
    const int runs = (int)((total / 8 / run_stride_d / W) / 16 * 16);
    kern<<<blocks, 256>>>(buf, run_stride_d, 16, out);
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(buf, run_stride_d, runs, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)runs * W * per * 512.0;
    printf("run length %5d B: %.2f GB read in %.3f ms = %.0f GB/s\n", per * 512, bytes / 1e9, ms, bytes / ms / 1e6);
  };
  run(1, rd<1>); run(2, rd<2>); run(4, rd<4>); run(8, rd<8>); run(16, rd<16>); run(32, rd<32>);
  return 0;
}
