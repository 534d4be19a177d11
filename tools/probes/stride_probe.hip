// Does the stride between the 16 rows a matrix-core fragment load touches matter?  Every wave reads, per instruction pair, 16 rows x 128 bytes
// (lane (l15, l4): row l15, 32 bytes at offset 32 l4) and walks 4 KB along each row - the access pattern of thin.h / a fragment load straight
// from global memory - for row strides of 4096, 4096 + 128, ... bytes.  Working set per launch: rows x 4 KB (L2-resident when small).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stride_probe tools/probes/stride_probe.hip ; run: /tmp/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void walk(const double* __restrict__ base, size_t stride_d, int rows_total, int reps, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int tile = (blockIdx.x * 4 + wave) % (rows_total / 16);
  const double* p = base + (size_t)(tile * 16 + l15) * stride_d + 4 * l4;
  double4_t acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
#pragma unroll 8
    for (int k = 0; k < 32; ++k) acc += *reinterpret_cast<const double4_t*>(p + 16 * k);      // 32 x 128 B = 4 KB per row
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456) out[0] = acc[0];
}
// the same bytes with other lane -> address maps: MODE 1: 16 bytes per lane at 16 l4 (a row gives 64 contiguous bytes per instruction), two instructions
// 64 bytes apart; MODE 2: fully contiguous, 1 KB per instruction (lane * 16 bytes), the wave walks one 64-KB run; MODE 3: 8 bytes per lane, 16 lanes on
// one 128-byte run, 4 rows per instruction (the factor loads of thin_f_kernel)
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void walk2(const double* __restrict__ base, size_t stride_d, int rows_total, int reps, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int tile = (blockIdx.x * 4 + wave) % (rows_total / 16);
  double acc = 0.0;
  if (MODE == 1) {
    const double* p = base + (size_t)(tile * 16 + l15) * stride_d + 2 * l4;
    for (int r = 0; r < reps; ++r)
#pragma unroll 8
      for (int k = 0; k < 32; ++k) {
        const double2_t a = *reinterpret_cast<const double2_t*>(p + 16 * k), b = *reinterpret_cast<const double2_t*>(p + 16 * k + 8);
        acc += a[0] + a[1] + b[0] + b[1];
      }
  } else if (MODE == 2) {
    const double* p = base + (size_t)(tile * 16) * stride_d + 2 * lane;                  // (stride_d = 512: the 16 "rows" are one 64-KB run)
    for (int r = 0; r < reps; ++r)
#pragma unroll 8
      for (int k = 0; k < 64; ++k) { const double2_t a = *reinterpret_cast<const double2_t*>(p + 128 * k); acc += a[0] + a[1]; }
  } else {
    const double* p = base + (size_t)(tile * 16 + l4) * stride_d + l15;                   // 4 rows x 128 B per instruction
    for (int r = 0; r < reps; ++r)
#pragma unroll 8
      for (int k = 0; k < 128; ++k) acc += p[(size_t)(k & 3) * 4 * stride_d + 16 * (k >> 2)];
  }
  if (acc == 123.456) out[0] = acc;
}
int main() {
  const int rows = 4096;                                  // 4096 rows x 4 KB = 16 MB touched (stride decides the span)
  const size_t maxstride = 41088 / 8 + 64;
  double* buf; double* out;
  hipMalloc(&buf, rows * maxstride * 8 + 4096); hipMemset(buf, 0, rows * maxstride * 8 + 4096); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rows_used : {512, 4096})
    for (size_t sb : {4096, 4096 + 128, 4096 + 256, 4096 + 512, 8192, 40960, 40960 + 128, 41088 + 256}) {
      const int blocks = 2048, reps = 4;
      walk<<<blocks, 256>>>(buf, sb / 8, rows_used, 1, out);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) walk<<<blocks, 256>>>(buf, sb / 8, rows_used, reps, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 10.0 * blocks * 4 * reps * 16 * 4096.0;
      printf("rows %5d (%5.1f MB of lines) stride %6zu B: %7.1f GB/s delivered to the waves\n", rows_used, rows_used * 4096.0 / 1e6, sb, bytes / ms / 1e6);
    }
  for (int mode = 1; mode <= 3; ++mode)
    for (int rows_used : {512, 4096}) {
      const int blocks = 2048, reps = 4;
      const size_t sd = mode == 2 ? 512 : 4096 / 8 + 16;
      auto go = [&](int rp) {
        if (mode == 1) walk2<1><<<blocks, 256>>>(buf, sd, rows_used, rp, out);
        else if (mode == 2) walk2<2><<<blocks, 256>>>(buf, sd, rows_used, rp, out);
        else walk2<3><<<blocks, 256>>>(buf, sd, rows_used, rp, out);
      };
      go(1);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) go(reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 10.0 * blocks * 4 * reps * 16 * 4096.0;
      printf("mode %d rows %5d: %7.1f GB/s delivered to the waves\n", mode, rows_used, bytes / ms / 1e6);
    }
  return 0;
}
