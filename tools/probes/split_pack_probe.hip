// round trip of a two-half FP16 storage form of the split accumulation's correction D (tried in round 4, csrc/split.h) on the device: worst relative error of (hi + lo) 2^-11 against the double,
// and of hi alone.  build: hipcc -O3 --offload-arch=gfx950 -o split_pack_probe split_pack_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
namespace pgpfa {
constexpr float SPLIT_SCALE = 2048.0f;
__device__ __forceinline__ float split_pack(double d) {
  const float x = (float)d * SPLIT_SCALE;
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  const unsigned w = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
  return __builtin_bit_cast(float, w);
}
__device__ __forceinline__ float split_unpack(float packed) {
  const unsigned w = __builtin_bit_cast(unsigned, packed);
  const _Float16 h = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
  const _Float16 l = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
  return ((float)h + (float)l) * (1.0f / SPLIT_SCALE);
}
}  // namespace pgpfa
__global__ void k(const double* in, float* packed, float* back, float* hi, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w = pgpfa::split_pack(in[i]);
  packed[i] = w;
  back[i] = pgpfa::split_unpack(w);
  const unsigned u = __builtin_bit_cast(unsigned, w);
  hi[i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)) / 2048.0f;
}
int main() {
  const int n = 1 << 16;
  double* h = new double[n];
  for (int i = 0; i < n; ++i) h[i] = std::ldexp(std::sin(0.37 * i + 0.1), -(i % 24)) * ((i & 1) ? 1 : -1);
  double* d; float *p, *b, *hh;
  hipMalloc(&d, n * 8); hipMalloc(&p, n * 4); hipMalloc(&b, n * 4); hipMalloc(&hh, n * 4);
  hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, p, b, hh, n);
  float* hb = new float[n]; float* hhi = new float[n];
  hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hhi, hh, n * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst_hi = 0; int wi = 0;
  for (int i = 0; i < n; ++i) {
    if (std::fabs(h[i]) < 1e-6) continue;
    const double e = std::fabs(hb[i] - h[i]) / std::fabs(h[i]);
    if (e > worst) { worst = e; wi = i; }
    worst_hi = std::fmax(worst_hi, std::fabs(hhi[i] - h[i]) / std::fabs(h[i]));
  }
  std::printf("worst relative error of the round trip for |d| >= 1e-6: %.3e (d = %.6e -> %.6e), of hi alone: %.3e\n", worst, h[wi], (double)hb[wi], worst_hi);
  return 0;
}
