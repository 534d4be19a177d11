// Layout and rate probe of v_mfma_f64_4x4x4_4b_f64 on gfx950 (4 independent 4x4x4 products per instruction).
// A lane passes ONE double of A, one of B, and gets one double of D.  Which (block, i, k) / (block, k, j) / (block, i, j) a lane holds
// is found by feeding one-hot operands: A = 1 at a single lane la, B = 1 at a single lane lb, and recording which lanes of D become 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void onehot(int la, int lb, double* out) {
  const int lane = threadIdx.x;
  double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[lane] = d;
}
template <int KIND>
__global__ void rate(double* out, int iters) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  if (KIND == 0) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 64 + lane] = s;
  } else {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + lane] = s;
  }
}
int main() {
  double* d; hipMalloc(&d, 1 << 20);
  std::vector<double> h(64);
  // D lanes hit by (la, lb): print for a few probes the list of lanes with value 1
  int amap_blk[64], amap_i[64], amap_k[64];
  printf("one-hot probes: for A lane la, the B lanes lb that produce any output, and the D lanes\n");
  for (int la = 0; la < 64; ++la) {
    printf("la %2d:", la);
    for (int lb = 0; lb < 64; ++lb) {
      hipLaunchKernelGGL(onehot, dim3(1), dim3(64), 0, 0, la, lb, d);
      hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l) if (h[l] != 0.0) printf(" (lb %d -> D lane %d)", lb, l);
    }
    printf("\n");
  }
  for (int kind = 0; kind < 2; ++kind) {
    const int iters = 20000, blocks = 256 * 4;     // one wave per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(64), 0, 0, d, iters);
      else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(64), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)iters * 8;          // instructions per wave
      printf("%s: %.3f ms for %g instr per wave, one wave per SIMD -> %.1f ns per instr = %.1f cycles at 2.4 GHz\n", kind == 0 ? "4x4x4_4b" : "16x16x4", ms, n,
             ms * 1e6 / n, ms * 1e6 / n * 2.4);
    }
  }
  return 0;
}
