// How many workgroups share a CU as a function of their LDS allocation (gfx950: 160 KB per CU)?  Each workgroup touches its dynamic LDS and then
// spins for a fixed time; N = 256 CUs x 8 workgroups are launched and the elapsed time / spin time is the number of rounds -> workgroups per CU.
// Next to it the figure hipOccupancyMaxActiveBlocksPerMultiprocessor reports.  usage: lds_occupancy_probe [threads per workgroup, default 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(unsigned long long ticks, int lds_doubles, double* out) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < lds_doubles; i += blockDim.x) lds[i] = (double)i;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (lds[(threadIdx.x * 7) % lds_doubles] == -1.0) out[0] = 1.0;
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 256;
  double* out; hipMalloc((void**)&out, 8);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::printf("CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, threads per workgroup %d\n", cus, prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor, threads);
  const unsigned long long ticks = 2000;                  // 20 us at the 100-MHz wall clock
  for (int kb : {8, 16, 24, 32, 40, 48, 53, 64, 80, 96, 128, 160}) {
    const size_t bytes = (size_t)kb * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { (void)hipGetLastError(); std::printf("%3d KB: not allowed\n", kb); continue; }
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin_kernel, threads, bytes);
    const int n = cus * 8;
    hipLaunchKernelGGL(spin_kernel, dim3(n), dim3(threads), bytes, 0, ticks, (int)(bytes / 8), out);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("%3d KB: launch failed (%s)\n", kb, hipGetErrorString(hipGetLastError())); continue; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin_kernel, dim3(n), dim3(threads), bytes, 0, ticks, (int)(bytes / 8), out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double rounds = ms * 1e3 / 20.0;
    std::printf("%3d KB of LDS per workgroup: %7.1f us for %d workgroups = %.2f rounds of 20 us -> %.1f workgroups per CU at a time (runtime's occupancy figure: %d)\n",
                kb, ms * 1e3, n, rounds, 8.0 / rounds, occ);
  }
  return 0;
}
