// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per access width (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the
// bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Every kernel below streams the SAME buffer of `bytes` bytes once (default 2 GiB: past the 256-MiB Infinity Cache, which the
// counters do not exclude) with one access shape and writes a 4-byte checksum per workgroup; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- tools/probes/pmc_width_probe
//   rocprofv3 --pmc WRITE_SIZE ...
// and divide the counter (KB) by the bytes: tools/pmc_calibrate.py prints the factor per shape, tools/pmc_summary.py applies it per kernel.
//   read_b32 / read_b64 / read_b128        : adjacent lanes, adjacent addresses, 4 / 8 / 16 bytes per lane (global_load_dword / x2 / x4)
//   read_b64_gather55                      : the (C,d) M-step's pattern (csrc/mstep.h): 8-byte loads of 55 of every 100 doubles (the lower triangle
//                                            of consecutive 10 x 10 blocks) - every line touched, 55 % of its bytes used
//   read_b64_rows                          : 8 bytes per lane, a wave reads 64 consecutive doubles of a row, rows 4000 bytes apart (the n-vectors of
//                                            the inner solve at T = 500: rows not line-aligned)
//   write_b32 / write_b64 / write_b128     : the same shapes as stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)

template <typename V>
__global__ __launch_bounds__(256) void read_kernel(const V* __restrict__ src, size_t n, unsigned* __restrict__ out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const V v = src[i];
    const unsigned* w = reinterpret_cast<const unsigned*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / 4); ++k) acc ^= w[k];
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;           // (never true in practice: keeps the loads alive, writes nothing)
}
__global__ __launch_bounds__(256) void read_b64_gather55(const double* __restrict__ src, size_t nblocks, unsigned* __restrict__ out) {
  // thread -> (block of 100 doubles, packed index c of its lower triangle), consecutive threads walk c then the block - as the staging plan of mstep.h does
  double acc = 0.0;
  const size_t total = nblocks * 55;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t b = e / 55; const int c = (int)(e - b * 55);
    int pa = 0;
    while ((pa + 1) * (pa + 2) / 2 <= c) ++pa;
    const int pb = c - pa * (pa + 1) / 2;
    acc += src[b * 100 + pa * 10 + pb];
  }
  if (acc == 0.12345) out[blockIdx.x] = 1u;
}
__global__ __launch_bounds__(256) void read_b64_rows(const double* __restrict__ src, size_t nrows, unsigned* __restrict__ out) {
  // a wave reads 64 consecutive doubles of a row; rows are 500 doubles (4000 bytes) long, eight 64-double tiles each (the last one short)
  double acc = 0.0;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  const int lane = threadIdx.x & 63;
  for (size_t w = wave; w < nrows * 8; w += nwaves) {
    const size_t row = w >> 3; const int t = (int)(w & 7) * 64 + lane;
    acc += src[row * 500 + (t < 500 ? t : 499)];
  }
  if (acc == 0.12345) out[blockIdx.x] = 1u;
}
template <typename V>
__global__ __launch_bounds__(256) void write_kernel(V* __restrict__ dst, size_t n) {
  V v;
  unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(V) / 4); ++k) w[k] = threadIdx.x + k;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)std::atoll(argv[1]) : (size_t)2048) << 20;      // MiB
  void* buf = nullptr; unsigned* out = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc((void**)&out, 65536 * sizeof(unsigned)));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipMemset(out, 0, 65536 * sizeof(unsigned)));
  const dim3 grid(8192), block(256);
  const size_t nblocks = bytes / 800;                      // blocks of 100 doubles
  const size_t nrows = bytes / 4000;                       // rows of 500 doubles
  for (int rep = 0; rep < 2; ++rep) {                      // (rep 0 warms code objects; the summary takes the mean over both)
    hipLaunchKernelGGL(read_kernel<unsigned>, grid, block, 0, 0, (const unsigned*)buf, bytes / 4, out);
    hipLaunchKernelGGL(read_kernel<uint2>, grid, block, 0, 0, (const uint2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(read_kernel<uint4>, grid, block, 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(read_b64_gather55, grid, block, 0, 0, (const double*)buf, nblocks, out);
    hipLaunchKernelGGL(read_b64_rows, grid, block, 0, 0, (const double*)buf, nrows, out);
    hipLaunchKernelGGL(write_kernel<unsigned>, grid, block, 0, 0, (unsigned*)buf, bytes / 4);
    hipLaunchKernelGGL(write_kernel<uint2>, grid, block, 0, 0, (uint2*)buf, bytes / 8);
    hipLaunchKernelGGL(write_kernel<uint4>, grid, block, 0, 0, (uint4*)buf, bytes / 16);
    CHECK(hipDeviceSynchronize());
  }
  std::printf("bytes %zu blocks100 %zu rows500 %zu\n", bytes, nblocks * 800, nrows * 4000);
  hipFree(buf); hipFree(out);
  return 0;
}
