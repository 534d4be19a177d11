// How fast the per-bin kernels of the inner solve (pcg.h: pcg_cg_a / b_kernel) can READ a slot's vectors, as a function of their LAYOUT in memory.
// A (slot, 64-bin tile) work item of kernel A reads 75 runs of 256 bytes (r, y: 10 latent rows each; the packed single-precision curvature: 55
// component rows), every run on another row of the slot - 2 KB apart today ([row][T]).  Laid out tile-major ([tile][row][64 bins]) the same
// bytes are ONE run of 19 200 bytes.  Same grid as the kernel (512 workgroups of 4 waves, a wave walks 4 slots), all loads of an item in flight
// together, a reduction so that nothing is optimised away.  Prints the rate of either layout over 1024 slots (1.5 GB apart per vector set: HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ROWS = 75, TILES = 8, TW = 512;      // rows of a slot, 64-bin tiles, row length in floats

template <bool TILE_MAJOR>
__global__ __launch_bounds__(256, 2) void read_kernel(const float* __restrict__ src, size_t slot_stride, int nslots, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x, group = blockIdx.y;
  float acc = 0.f;
  for (int si = group * 16 + wave; si < min(nslots, group * 16 + 16); si += 4) {
    const float* base = src + (size_t)si * slot_stride;
    float v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) v[r] = TILE_MAJOR ? base[((size_t)tile * ROWS + r) * 64 + lane] : base[(size_t)r * TW + tile * 64 + lane];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc += v[r];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave] = acc;
}

int main() {
  const int nslots = 1024;
  const size_t slot_stride = (size_t)ROWS * TW + 4096 * 75;       // floats: a slot's rows, then other vectors of the slot (as in the arena)
  float *src, *out;
  CK(hipMalloc(&src, nslots * slot_stride * sizeof(float)));
  CK(hipMalloc(&out, 8 * 64 * 4 * sizeof(float) * 4));
  CK(hipMemset(src, 0, nslots * slot_stride * sizeof(float)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(TILES, nslots / 16), block(256);
  const double bytes = (double)nslots * TILES * ROWS * 64 * 4;
  for (int layout = 0; layout < 2; ++layout) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipEventRecord(e0));
      if (layout) hipLaunchKernelGGL(read_kernel<true>, grid, block, 0, 0, src, slot_stride, nslots, out);
      else hipLaunchKernelGGL(read_kernel<false>, grid, block, 0, 0, src, slot_stride, nslots, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 3 && ms < best) best = ms;
    }
    std::printf("%-34s %7.1f us  %6.2f TB/s of %.0f MB\n", layout ? "tile-major [tile][row][64 bins]:" : "row-major [row][T] (today):", best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / 1e6);
  }
  return 0;
}
