// Batched FP64 "NT" GEMM for gfx950:  C[b] = beta*C[b] + alpha * A[b] * op(B[b])
// column-major everywhere; A is M x K (lda); B is N x K (ldb) when TRANSB=0 ("NT": C = A B^T)
// or K x N (ldb) when TRANSB=1 ("NN").
//
// This one kernel carries every O(n^3) step of the E-step (trailing SYRK updates of the blocked
// Cholesky, TRSM-as-GEMM against the inverted diagonal block, the triangular inverse, the
// selected products Sigma_kk = L^-T L^-1 restricted to diagonal blocks) - see chol.h.
//
// CDNA4 mapping: 256 threads = 4 waves per 128x128 C tile, each wave owns a 64x64 sub-tile as
// 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  A and B panels are staged k-major in
// LDS ([k][row], row stride 144 doubles so the two 16-lane halves of a ds_read_b64 group land on
// disjoint bank halves), double-buffered, one barrier per 16-deep K step.  The MFMA is issued as
// D = Bfrag x Afrag so that the lane index (lane&15) runs along the memory-contiguous row index
// of C: every accumulator store is 16 lanes x 8 B contiguous.
// FP64 MFMA on gfx950 runs at the vector FP64 rate (78.6 TFLOP/s chip peak), i.e. 64 cycles per
// 16x16x4 instruction per SIMD, so one LDS fragment read per MFMA is far below the LDS roof and the
// kernel is MFMA-issue bound by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "types.h"

namespace pgpfa {

// BT: tile size (BT x BT outputs per workgroup, 4 waves of (BT/2) x (BT/2)).  128: 4 x 4 accumulator tiles per wave, the form for
// products with many tiles.  64: 2 x 2 per wave - a quarter of the work per workgroup, 40 KB of LDS and ~80 registers, so 3-4 workgroups
// share a CU: thin products (multi-RHS vectors against block-diagonal factors, the r x r preconditioner, short panels) offer 4x the
// workgroups and hide each other's prologue and epilogue.  FP64 MFMA issues one 16x16x4 per 64 cycles per SIMD, so the extra LDS reads
// per MFMA of the small tile (1 instead of 1/2) are still far under the LDS roof.
// BF32 (TRANSB = 0, T = double only): the B operand is stored in single precision (strides sB, sB_hi, sBseg, ldb count floats) and widened
// while it is staged - products and accumulation stay FP64.
template <int TRANSB, typename T, int BT, bool BF32 = false>
__global__ __launch_bounds__(256, BT == 128 ? 2 : 3) void gemm_mfma_kernel_t(GemmP g) {
  using T2 = typename GemmVec<T>::v2;
  using T4 = typename GemmVec<T>::v4;
  constexpr int LS = BT + 16;                 // LDS row stride (elements): 2 * LS words = 32 mod 64 for both tile sizes
  constexpr int WT = BT / 2;                  // wave tile
  constexpr int MI = WT / 16;                 // 16 x 16 accumulator tiles per wave and dimension
  constexpr int NU = BT / 32;                 // staging units (pairs of elements) per thread and operand
  __shared__ __attribute__((aligned(16))) T As[2][GBK][LS];
  __shared__ __attribute__((aligned(16))) T Bs[2][GBK][LS];
  if (g.skip && *g.skip) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int b, tile;
  gemm_decode_block(g, blockIdx.x, b, tile);
  int ti, tj;
  gemm_decode_tile(g, tile, ti, tj);
  const T* A;
  const T* B;
  T* C;                            // may alias A (in-place TRSM): no restrict
  const int hi = gemm_bind_t<T>(g, b, A, B, C);
  const float* B32 = nullptr;
  if (BF32) {
    int lo2 = b, hi2 = 0;
    if (g.nb_lo > 0) { hi2 = b / g.nb_lo; lo2 = b - hi2 * g.nb_lo; }
    const long long slot2 = g.slots ? g.slots[lo2] : lo2;
    B32 = reinterpret_cast<const float*>(g.B) + slot2 * g.sB + hi2 * g.sB_hi;
  }

  int i0 = ti * BT, iend = g.M;
  const int j0 = tj * BT;
  if (g.n_dev) {                                 // (uniform over the workgroup: before any barrier)
    g.N = *g.n_dev;
    if (j0 >= g.N) return;
  }
  int kb = 0, ke = g.K;
  if (g.kflags & KF_BEGIN_ROW) kb = i0;
  if (g.kflags & KF_BEGIN_MAXRC) kb = (i0 > j0 ? i0 : j0);
  if (g.kflags & KF_END_ROW) ke = (i0 + BT < g.K ? i0 + BT : g.K);
  if (g.rtab) { i0 = g.rtab[4 * ti]; iend = g.rtab[4 * ti + 1]; kb = g.rtab[4 * ti + 2]; ke = g.rtab[4 * ti + 3]; }
  if (kb > ke) kb = ke;
  gemm_split_range(g.ksplit, hi, kb, ke);

  const bool a_vec = ((((size_t)A) & (2 * sizeof(T) - 1)) == 0) && ((g.lda & 1) == 0) && ((i0 & 1) == 0);
  const bool b_vec = ((((size_t)B) & (2 * sizeof(T) - 1)) == 0) && ((g.ldb & 1) == 0);

  // staging registers: NU pairs per operand per thread, two sets (the loads run two k chunks ahead of the matrix instructions, see the loop)
  T raA[2 * NU], rbA[2 * NU], raB[2 * NU], rbB[2 * NU];
  // (TRANSB) the B column each staging unit of this thread reads: j0 + nn, or its entry in the column list (clamped inside the list)
  // TRANSB staging map: 8 consecutive lanes walk the 16 k of a step of ONE column (a 128-byte line of that slot vector), a wave covers 8
  // columns - with lanes along the columns instead (the round-1 map) every load instruction touched 64 lines for 16 bytes each.  The
  // transposed LDS image would then be written 8-way conflicted (rows two apart are 320 words = 0 mod 64 apart), so its column index is
  // XOR-swizzled with ((k >> 1) & 7) << 3: the 64 (column, k pair) stores of a wave spread over all banks (2-way, the minimum for 8-byte
  // stores), and the fragment reads - 16 consecutive columns of rows kk + l4 - stay conflict-free because rows 2m, 2m + 1 share a swizzle.
  int bcol[NU];
#pragma unroll
  for (int s = 0; s < NU; ++s) {
    const int jj = j0 + (tid + 256 * s) / 8;
    bcol[s] = (TRANSB && g.cols) ? g.cols[jj < g.N ? jj : g.N - 1] : jj;
  }
  auto bswz = [](int k) { return TRANSB ? (((k >> 1) & 7) << 3) : 0; };

  // (vec: both operands allow 16-byte loads - decided once per workgroup OUTSIDE the loop: a branch around a load inside it makes the compiler wait for
  //  every outstanding load at the join)
  auto load_tiles = [&](auto vec, int k0, T (&ra)[2 * NU], T (&rb)[2 * NU]) {
    constexpr bool VEC = decltype(vec)::value;
#pragma unroll
    for (int s = 0; s < NU; ++s) {
      const int u = tid + 256 * s;
      {  // A: [k][row] ; unit -> k = u / (BT/2), rows 2*(u % (BT/2)), +1
        const int k = u / (BT / 2), r2 = (u % (BT / 2)) * 2;
        const T* src = A + gemm_koff(g.kseg, g.sAseg, g.lda, k0) + (size_t)k * g.lda + (i0 + r2);
        if constexpr (VEC) {
          const T2 v = *reinterpret_cast<const T2*>(src);
          ra[2 * s] = v.x; ra[2 * s + 1] = v.y;
        } else {
          ra[2 * s] = src[0]; ra[2 * s + 1] = src[1];
        }
      }
      if (TRANSB == 0 && BF32) {
        const int k = u / (BT / 2), r2 = (u % (BT / 2)) * 2;
        const float* src = B32 + gemm_koff(g.kseg, g.sBseg, g.ldb, k0) + (size_t)k * g.ldb + (j0 + r2);
        rb[2 * s] = (T)src[0]; rb[2 * s + 1] = (T)src[1];
      } else if (TRANSB == 0) {
        const int k = u / (BT / 2), r2 = (u % (BT / 2)) * 2;
        const T* src = B + gemm_koff(g.kseg, g.sBseg, g.ldb, k0) + (size_t)k * g.ldb + (j0 + r2);
        if constexpr (VEC) {
          const T2 v = *reinterpret_cast<const T2*>(src);
          rb[2 * s] = v.x; rb[2 * s + 1] = v.y;
        } else {
          rb[2 * s] = src[0]; rb[2 * s + 1] = src[1];
        }
      } else {  // B is K x N: unit -> k pair = u % 8, n = u / 8
        const int k2 = (u % 8) * 2;
        const T* src = B + (size_t)bcol[s] * g.ldb + (k0 + k2);
        if constexpr (VEC) {
          const T2 v = *reinterpret_cast<const T2*>(src);
          rb[2 * s] = v.x; rb[2 * s + 1] = v.y;
        } else {
          rb[2 * s] = src[0]; rb[2 * s + 1] = src[1];
        }
      }
    }
  };
  auto store_tiles = [&](int buf, const T (&ra)[2 * NU], const T (&rb)[2 * NU]) {
#pragma unroll
    for (int s = 0; s < NU; ++s) {
      const int u = tid + 256 * s;
      {
        const int k = u / (BT / 2), r2 = (u % (BT / 2)) * 2;
        *reinterpret_cast<T2*>(&As[buf][k][r2]) = T2{ra[2 * s], ra[2 * s + 1]};
      }
      if (TRANSB == 0) {
        const int k = u / (BT / 2), r2 = (u % (BT / 2)) * 2;
        *reinterpret_cast<T2*>(&Bs[buf][k][r2]) = T2{rb[2 * s], rb[2 * s + 1]};
      } else {
        const int nn = (u / 8) ^ bswz(u % 8 * 2), k2 = (u % 8) * 2;
        Bs[buf][k2][nn] = rb[2 * s];
        Bs[buf][k2 + 1][nn] = rb[2 * s + 1];
      }
    }
  };

  T4 acc[MI][MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < MI; ++ni) acc[mi][ni] = T4{(T)0, (T)0, (T)0, (T)0};

  // a wave whose sub-tile lies entirely outside C still stages and syncs, but skips MFMAs
  const bool wave_live = (i0 + wm * WT < iend) && (j0 + wn * WT < g.N) &&
                         !((g.mode == GEMM_LOWER) && (i0 + wm * WT + WT - 1 < j0 + wn * WT));

  const int nk = (ke - kb) / GBK;
  const int l15 = lane & 15, l4 = lane >> 4;
  auto multiply = [&](int buf) {
    if (!wave_live) return;
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 4) {
      T af[MI], bf[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = As[buf][kk + l4][wm * WT + mi * 16 + l15];
#pragma unroll
      for (int ni = 0; ni < MI; ++ni) bf[ni] = Bs[buf][kk + l4][(wn * WT + ni * 16 + l15) ^ bswz(kk + l4)];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < MI; ++ni)
          acc[mi][ni] = gemm_mfma16(bf[ni], af[mi], acc[mi][ni]);
    }
  };
  // The loads run TWO chunks ahead of the matrix instructions (a chunk is 16 - 64 of them per wave, 0.4 - 1.7 us: with one chunk of lead a
  // workgroup waited for memory in every trip and only its neighbours on the CU filled the gap).  Chunk c sits in LDS buffer c & 1; while it is
  // multiplied, chunk c + 1 waits in one register set and chunk c + 2 is being loaded into the other.  The loads of the steady loop stand under no
  // condition - the compiler then counts them (s_waitcnt vmcnt(n) leaves the newer set in flight; after a branch it waits for everything) - so the
  // last one to three chunks run in a tail of their own.
  // (the FP64 128 x 128 tile holds 128 accumulator registers: a second staging set makes it spill - it keeps one chunk of lead)
  constexpr bool DEEP = (BT == 64) || (sizeof(T) == 4);
  auto run = [&](auto vec) {
    if constexpr (!DEEP) {
      if (nk > 0) {
        load_tiles(vec, kb, raA, rbA);
        store_tiles(0, raA, rbA);
      }
      __syncthreads();
      for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        if (it + 1 < nk) load_tiles(vec, kb + (it + 1) * GBK, raA, rbA);
        multiply(buf);
        if (it + 1 < nk) store_tiles(buf ^ 1, raA, rbA);
        __syncthreads();
      }
      return;
    }
    if (nk > 0) {
      load_tiles(vec, kb, raA, rbA);
      if (nk > 1) load_tiles(vec, kb + GBK, raB, rbB);
      store_tiles(0, raA, rbA);
    }
    __syncthreads();
    int it = 0;
    for (; it + 3 < nk; it += 2) {
      load_tiles(vec, kb + (it + 2) * GBK, raA, rbA);
      multiply(0);
      store_tiles(1, raB, rbB);
      __syncthreads();
      load_tiles(vec, kb + (it + 3) * GBK, raB, rbB);
      multiply(1);
      store_tiles(0, raA, rbA);
      __syncthreads();
    }
    // here: chunk `it` in LDS buffer 0, chunk it + 1 (if any) in register set B, nothing else requested
    const int rem = nk - it;
    if (rem == 1) {
      multiply(0);
    } else if (rem == 2) {
      multiply(0);
      store_tiles(1, raB, rbB);
      __syncthreads();
      multiply(1);
    } else if (rem == 3) {
      load_tiles(vec, kb + (it + 2) * GBK, raA, rbA);
      multiply(0);
      store_tiles(1, raB, rbB);
      __syncthreads();
      multiply(1);
      store_tiles(0, raA, rbA);
      __syncthreads();
      multiply(0);
    }
  };
  if (a_vec && b_vec) run(std::true_type{}); else run(std::false_type{});

  if (!wave_live) return;
  // D[row][col = l15]: row <-> j (B index), col <-> i (A index)
  const bool mask_diag = (g.kflags & KF_MASK_DIAG) && (ti == tj);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int i = i0 + wm * WT + mi * 16 + l15;
    if (i >= iend) continue;
#pragma unroll
    for (int ni = 0; ni < MI; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // accumulator register r of a lane: row l4 + 4 r of the 16 x 16 tile for FP64, row 4 l4 + r for FP32 (the C/D map of
        // v_mfma_f64_16x16x4_f64 differs from every other shape / dtype on gfx950)
        const int j = j0 + wn * WT + ni * 16 + (sizeof(T) == 8 ? l4 + 4 * r : 4 * l4 + r);
        if (j >= g.N) continue;
        if (mask_diag && i < j) continue;
        T* dst = C + (size_t)((TRANSB && g.cols && !g.cols_c_off) ? g.cols[j] : j) * g.ldc + i;
        T v = (T)g.alpha * acc[mi][ni][r];
        if (g.beta != 0.0) v += (T)g.beta * (*dst);
        *dst = v;
      }
    }
  }
}

// Scalar check kernel: one thread per C element, same argument struct and masks.  Used only to
// validate the MFMA path (option use_mfma = 0); never the default.
template <int TRANSB>
__global__ void gemm_check_kernel(GemmP g) {
  if (g.skip && *g.skip) return;
  int b, tile;
  gemm_decode_block(g, blockIdx.x, b, tile);
  int ti, tj;
  gemm_decode_tile(g, tile, ti, tj);
  const double* A;
  const double* B;
  double* C;
  const int hi = gemm_bind(g, b, A, B, C);
  const int BT = g.bm;
  int i0 = ti * BT, iend = g.M;
  const int j0 = tj * BT;
  if (g.n_dev) {
    g.N = *g.n_dev;
    if (j0 >= g.N) return;
  }
  int kb = 0, ke = g.K;
  if (g.kflags & KF_BEGIN_ROW) kb = i0;
  if (g.kflags & KF_BEGIN_MAXRC) kb = (i0 > j0 ? i0 : j0);
  if (g.kflags & KF_END_ROW) ke = (i0 + BT < g.K ? i0 + BT : g.K);
  if (g.rtab) { i0 = g.rtab[4 * ti]; iend = g.rtab[4 * ti + 1]; kb = g.rtab[4 * ti + 2]; ke = g.rtab[4 * ti + 3]; }
  if (kb > ke) kb = ke;
  gemm_split_range(g.ksplit, hi, kb, ke);
  const bool mask_diag = (g.kflags & KF_MASK_DIAG) && (ti == tj);
  // compute first, store after a barrier: C may alias A (in-place TRSM)
  double vals[GBM * GBN / 256];
  int cnt = 0;
  for (int e = threadIdx.x; e < BT * BT; e += 256, ++cnt) {
    const int i = i0 + (e % BT), j = j0 + (e / BT);
    double s = 0.0;
    if (i < iend && j < g.N) {
      for (int k = kb; k < ke; ++k) {
        const double a = A[gemm_koff(g.kseg, g.sAseg, g.lda, k) + i];
        const double bb = TRANSB ? B[(size_t)(g.cols ? g.cols[j] : j) * g.ldb + k] : B[gemm_koff(g.kseg, g.sBseg, g.ldb, k) + j];
        s += a * bb;
      }
    }
    vals[cnt] = s;
  }
  __syncthreads();
  cnt = 0;
  const int WT = BT / 2;
  for (int e = threadIdx.x; e < BT * BT; e += 256, ++cnt) {
    const int i = i0 + (e % BT), j = j0 + (e / BT);
    if (i >= iend || j >= g.N) continue;
    if (mask_diag && i < j) continue;
    if (g.mode == GEMM_LOWER && ((i - i0) / WT) * WT + i0 + WT - 1 < ((j - j0) / WT) * WT + j0) continue;
    double* dst = C + (size_t)((TRANSB && g.cols && !g.cols_c_off) ? g.cols[j] : j) * g.ldc + i;
    double v = g.alpha * vals[cnt];
    if (g.beta != 0.0) v += g.beta * (*dst);
    *dst = v;
  }
}

// C[b] = beta*C[b] + sum_s part[s][b]  (part: [ksplit][nbatch][N][M] compact); grid = (ceil(M*N/256), nbatch)
inline __global__ void gemm_splitk_reduce_kernel(const double* __restrict__ part, int ksplit, int M, int N, int nbatch, double* __restrict__ C,
                                          long long sC, int ldc, const int* __restrict__ slots, double beta, const int* __restrict__ skip,
                                          const int* __restrict__ cols, const int* __restrict__ n_dev) {
  if (skip && *skip) return;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)M * N) return;
  if (n_dev && e / M >= (size_t)*n_dev) return;
  const int b = blockIdx.y;
  const long long slot = slots ? slots[b] : b;
  const size_t i = e % M, j = e / M;
  double s = 0.0;
  for (int k = 0; k < ksplit; ++k) s += part[((size_t)k * nbatch + b) * M * N + e];
  double* dst = C + slot * sC + (cols ? (size_t)cols[j] : j) * ldc + i;
  *dst = (beta != 0.0) ? beta * (*dst) + s : s;
}

// Register-only FP64 MFMA loop: measures the sustained v_mfma_f64_16x16x4_f64 rate of the chip (no
// memory traffic), i.e. the practical ceiling of gemm_mfma_kernel under the clock the chip holds.
inline __global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters) {
  double4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

inline int gemm_count_tiles(int mode, int tilesM, int tilesN) {
  if (mode == GEMM_FULL) return tilesM * tilesN;
  int n = 0;
  for (int tj = 0; tj < tilesN && tj < tilesM; ++tj) n += tilesM - tj;
  return n;
}

// Host launcher.  K, and every k range implied by kflags, must be a multiple of 16.  g.bm selects the tile size (0: 128).
hipError_t gemm_launch(hipStream_t st, bool use_mfma, bool transb, GemmP g, bool f32) {
  if (g.bm != 64) g.bm = 128;
  g.tilesM = g.rtab ? g.ntab : (g.M + g.bm - 1) / g.bm;
  g.tilesN = (g.N + g.bm - 1) / g.bm;
  g.ntiles = gemm_count_tiles(g.mode, g.tilesM, g.tilesN);
  if (g.ntiles <= 0 || g.nbatch <= 0 || g.M <= 0 || g.N <= 0) return hipSuccess;
  if (g.kseg != 0 && (transb || g.kseg % GBK != 0)) return hipErrorInvalidValue;
  if (g.rtab && g.mode != GEMM_FULL) return hipErrorInvalidValue;
  if (g.cols && !transb) return hipErrorInvalidValue;
  const long long blocks = (long long)g.ntiles * g.nbatch;
  dim3 grid((unsigned)blocks);
  if (f32) {
    if (g.kseg != 0) return hipErrorInvalidValue;        // (segmented K addresses are computed for FP64 operands only)
    if (g.bm == 64) {
      if (transb) hipLaunchKernelGGL((gemm_mfma_kernel_t<1, float, 64>), grid, dim3(256), 0, st, g);
      else hipLaunchKernelGGL((gemm_mfma_kernel_t<0, float, 64>), grid, dim3(256), 0, st, g);
    } else {
      if (transb) hipLaunchKernelGGL((gemm_mfma_kernel_t<1, float, 128>), grid, dim3(256), 0, st, g);
      else hipLaunchKernelGGL((gemm_mfma_kernel_t<0, float, 128>), grid, dim3(256), 0, st, g);
    }
  } else if (use_mfma) {
    if (g.b_f32) {
      if (transb || g.bm != 64) return hipErrorInvalidValue;
      hipLaunchKernelGGL((gemm_mfma_kernel_t<0, double, 64, true>), grid, dim3(256), 0, st, g);
    } else if (g.bm == 64) {
      if (transb) hipLaunchKernelGGL((gemm_mfma_kernel_t<1, double, 64>), grid, dim3(256), 0, st, g);
      else hipLaunchKernelGGL((gemm_mfma_kernel_t<0, double, 64>), grid, dim3(256), 0, st, g);
    } else {
      if (transb) hipLaunchKernelGGL((gemm_mfma_kernel_t<1, double, 128>), grid, dim3(256), 0, st, g);
      else hipLaunchKernelGGL((gemm_mfma_kernel_t<0, double, 128>), grid, dim3(256), 0, st, g);
    }
  } else {
    if (g.b_f32) return hipErrorInvalidValue;
    if (transb) hipLaunchKernelGGL(gemm_check_kernel<1>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_check_kernel<0>, grid, dim3(256), 0, st, g);
  }
  return hipGetLastError();
}

}  // namespace pgpfa
