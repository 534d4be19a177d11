// Semantics probe of v_permlane16_swap / v_permlane32_swap on gfx950: the transpose-and-reduce over the four 16-lane rows of a wave and the
// broadcast back, as mstep_cd_hess_mfma_kernel uses them (expected output: every lane prints sum == expect, v0..v3 = the value of rows 0..3).
// hipcc -O3 --offload-arch=gfx950 -o permlane_swap_probe permlane_swap_probe.hip
#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ inline void swap16(double& a, double& b) {
  unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  u2 r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  u2 r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__device__ inline void swap32(double& a, double& b) {
  unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  u2 r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  u2 r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  double s0 = 1000 + lane, s1 = 2000 + lane, s2 = 3000 + lane, s3 = 4000 + lane;
  swap16(s0, s1); swap16(s2, s3);
  double t01 = s0 + s1, t23 = s2 + s3;
  swap32(t01, t23);
  out[lane] = t01 + t23;
  // broadcast back
  double x = out[lane], a = x, b = x;
  swap16(a, b);
  double a2 = a, b2 = b;
  swap32(a, a2); swap32(b, b2);
  out[64 + lane] = a; out[128 + lane] = b; out[192 + lane] = a2; out[256 + lane] = b2;
}
int main() {
  double* d; hipMalloc(&d, 320 * 8); k<<<1, 64>>>(d); double h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    const int l15 = l & 15, b = l >> 4;
    double expect = 0; for (int r = 0; r < 4; ++r) expect += 1000 * (b + 1) + (l15 + 16 * r);
    printf("lane %2d sum %g expect %g | v0 %g v1 %g v2 %g v3 %g\n", l, h[l], expect, h[64 + l], h[128 + l], h[192 + l], h[256 + l]);
  }
}
