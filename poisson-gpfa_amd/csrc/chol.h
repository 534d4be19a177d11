// Batched dense SPD factor / solve / transposed-triangular-inverse on top of gemm.h.
//
// Storage: one column-major (ld x ld) slab per slot; only the lower triangle is referenced by the
// factorisation.  npad (multiple of 128) rows/cols are active; rows >= n are identity padding.
//
//   chol_factor      blocked right-looking Cholesky, 128-wide diagonal blocks inside 512-wide
//                    super-panels: potrf_diag (one workgroup per slot, register-resident 128x128 block,
//                    also emits the block's triangular inverse) -> TRSM as GEMM against that inverse
//                    -> SYRK trailing update restricted to the super-panel; one K=512 SYRK update of
//                    the rest per super-panel keeps the trailing matrix traffic at n^3/(6*512)*16 B.
//   chol_solve       forward/backward substitution with the factor (Newton step), one workgroup per
//                    slot; HBM-bound on one read of L per sweep.
//   chol_inverse_t   Mt = L^-T (upper triangular) into a second slab, left-to-right over block
//                    columns with two NT GEMMs per column; Sigma blocks then follow as Mt Mt^T.
#pragma once
#include "types.h"

namespace pgpfa {


// --------------------------------------------------------------------------------------------------
// 128x128 diagonal block: Cholesky and triangular inverse.  512 threads; thread (r = tid & 127, cg = tid >> 7)
// owns row r, columns c = cg + 4m (m = 0..31) of the trailing matrix in 32 registers.  Step j broadcasts column j
// through a double-buffered LDS line (one barrier per step) and every thread applies the rank-1 update to its
// registers with no masks: registers of finished columns are dead (the finished column of L goes to an LDS copy),
// and the steps are instantiated per quarter of the block so that the register range still alive is static.
// The inverse runs in 32-wide blocks over the factor in LDS (potrf_diag_inverse).
// Writes L back in place and L^-1 to Dinv[slot][k0/128]; info[slot] = (k0 + j + 1) at the first bad pivot.
// --------------------------------------------------------------------------------------------------
// offset of L[r][j] (r >= j) in the packed column-major lower triangle
__device__ __forceinline__ int potrf_lidx(int j, int r) { return j * NB - (j * (j - 1)) / 2 + (r - j); }

template <typename T>
struct PotrfLds {
  T L[NB * (NB + 1) / 2];       // packed lower triangle of the factor, column by column (66 KB in FP64: two workgroups per CU)
  T line[2][NB];                // broadcast lines, permuted: pos(i) = (i & 3) * 32 + (i >> 2)
  T rd[2];                      // 1 / L[j][j] of the current step
  // the 2-D form (potrf_diag_chol_steps2): column j once per consumer - rowc[buf][h][2 tr + (i & 1)] = entry tr + 32 i (i = 2 h, 2 h + 1: a thread's
  // four rows in two 16-byte pieces, lane stride 16 bytes), colc[buf][8 tc + m] = entry tc + 16 m (a thread's eight columns contiguous)
  T rowc[2][2][64];
  T colc[2][NB];
};

template <typename T>
__device__ __forceinline__ T pick8(const T* a, int k) {
  const T v01 = (k & 1) ? a[1] : a[0], v23 = (k & 1) ? a[3] : a[2];
  const T v45 = (k & 1) ? a[5] : a[4], v67 = (k & 1) ? a[7] : a[6];
  const T v03 = (k & 2) ? v23 : v01, v47 = (k & 2) ? v67 : v45;
  return (k & 4) ? v47 : v03;
}

template <int Q, typename T>
__device__ __forceinline__ void potrf_diag_chol_steps(T (&a)[32], PotrfLds<T>& S, int r, int cg, int pos_r,
                                                       int* __restrict__ info, long long slot, int k0) {
#pragma unroll 1
  for (int j = 32 * Q; j < 32 * Q + 32; ++j) {
    const bool own = (j & 3) == cg;
    T* buf = S.line[j & 1];
    if (own && r >= j) {
      const T v = pick8(&a[8 * Q], (j >> 2) - 8 * Q);
      buf[pos_r] = v;                             // column j, unscaled
      if (r == j) {
        T rd = rsqrt(v);
        if (!(v > (T)0)) {                         // also catches NaN
          if (info[slot] == 0) info[slot] = k0 + j + 1;
          rd = (T)1;
        }
        S.rd[j & 1] = rd;
      }
    }
    __syncthreads();
    if (r >= j) {                                 // whole waves drop out as j advances
      const T rd = S.rd[j & 1];
      const T lr = buf[pos_r] * rd;               // L[r][j]  (r == j: a_jj / sqrt(a_jj))
      if (own) S.L[potrf_lidx(j, r)] = lr;
      const T nlr = -lr * rd;
      const T* bp = buf + cg * 32;
#pragma unroll
      for (int m = 8 * Q; m < 32; ++m) a[m] = fma(nlr, bp[m], a[m]);
    }
  }
}

// The same steps with the trailing matrix in 4 x 8 register blocks (round 5): thread (tr = tid & 31, tc = tid >> 5) owns rows tr + 32 i (i < 4)
// and columns tc + 16 m (m < 8).  In the form above a thread owns one row and 32 columns and reads 32 entries of column j from LDS per step -
// 16 wide reads per wave, 8 waves: ~1000 cycles of the LDS pipe per step.  A 4 x 8 block needs 4 + 8 entries: six 16-byte reads, the column side a
// broadcast.  Measured: 68 -> 62 us for the 128 steps of one block, 190 -> 180 us at batch 1024 - a step is mostly the latency of its dependent chain
// (pivot, FP64 reciprocal square root, LDS line, barrier, LDS reads, update), not the LDS bytes.  Steps are instantiated per 16 columns so that the live column range is
// static (m >= S); finished rows and the partly finished column block S are updated with numbers nobody reads again.
template <int S, typename T>
__device__ __forceinline__ void potrf_diag_chol_steps2(T (&a)[4][8], PotrfLds<T>& L, int tr, int tc, int* __restrict__ info, long long slot, int k0) {
  typedef T v2 __attribute__((ext_vector_type(2)));
#pragma unroll 1
  for (int j = 16 * S; j < 16 * S + 16; ++j) {
    const int b = j & 1;
    if (tc == (j & 15)) {                                 // column j = tc + 16 S: this thread's column m = S, its rows tr + 32 i
      *reinterpret_cast<v2*>(&L.rowc[b][0][2 * tr]) = v2{a[0][S], a[1][S]};
      *reinterpret_cast<v2*>(&L.rowc[b][1][2 * tr]) = v2{a[2][S], a[3][S]};
#pragma unroll
      for (int i = 0; i < 4; ++i) L.colc[b][(tr & 15) * 8 + 2 * i + (tr >> 4)] = a[i][S];    // entry c = tr + 32 i -> [8 (c & 15) + (c >> 4)]
      if (tr == (j & 31)) {
        const T v = (j >> 5) == 0 ? a[0][S] : (j >> 5) == 1 ? a[1][S] : (j >> 5) == 2 ? a[2][S] : a[3][S];
        T rd = rsqrt(v);
        if (!(v > (T)0)) {                                // also catches NaN
          if (info[slot] == 0) info[slot] = k0 + j + 1;
          rd = (T)1;
        }
        L.rd[b] = rd;
      }
    }
    __syncthreads();
    const T rd = L.rd[b];
    const v2 r01 = *reinterpret_cast<const v2*>(&L.rowc[b][0][2 * tr]), r23 = *reinterpret_cast<const v2*>(&L.rowc[b][1][2 * tr]);
    T lr[4] = {r01.x * rd, r01.y * rd, r23.x * rd, r23.y * rd};          // L[r][j] of this thread's rows (r >= j: the others are not used)
    if (tc == (j & 15)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = tr + 32 * i;
        if (r >= j) L.L[potrf_lidx(j, r)] = lr[i];
      }
    }
    T cv[8];
#pragma unroll
    for (int m = S & ~1; m < 8; m += 2) {
      const v2 c2 = *reinterpret_cast<const v2*>(&L.colc[b][8 * tc + m]);
      cv[m] = c2.x; cv[m + 1] = c2.y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const T nlr = -lr[i] * rd;
#pragma unroll
      for (int m = S; m < 8; ++m) a[i][m] = fma(nlr, cv[m], a[i][m]);
    }
  }
}

// X = L^-1 of the 128 x 128 block, in place over the packed factor in LDS, in 32-wide blocks:
//   1. the four diagonal blocks X_QQ = L_QQ^-1 at the same time, by rows (forward substitution, row i of every block broadcast through the
//      LDS line, 32 barrier steps instead of 128), in 8 registers per thread; then written over L_QQ;
//   2. block rows P = 1..3 in turn: W = L_P,Q..P-1 X_Q..P-1,Q for every block Q < P (X of the rows above is final and sits where their L
//      was), then X_PQ = -X_PP W: two small products on the matrix cores, 16 x 16 tiles over the 8 waves, fragments read from the packed
//      triangle in LDS (a scalar version of the same products was bound by its ~1500 LDS reads per thread: 65 us for the inverse).
template <typename T>
__device__ __forceinline__ void potrf_diag_inverse(PotrfLds<T>& S, int tid, int r, int cg) {
  const int Q = r >> 5, rl = r & 31;
  const T inv_lrr = (T)1 / S.L[potrf_lidx(r, r)];
  T t[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) t[m] = (T)0;
#pragma unroll 1
  for (int il = 0; il < 32; ++il) {
    T* xrow = S.line[il & 1] + Q * 32 + cg * 8;           // row 32 Q + il of X_QQ: columns cg + 4 m of the block at [cg * 8 + m]
    if (rl == il) {
#pragma unroll
      for (int m = 0; m < 8; ++m) xrow[m] = t[m];
      if ((il & 3) == cg) xrow[il >> 2] = inv_lrr;         // X[i][i]
    }
    __syncthreads();
    if (rl > il) {
      const T lri = -S.L[potrf_lidx(32 * Q + il, r)] * inv_lrr;
#pragma unroll
      for (int m = 0; m < 8; ++m) t[m] = fma(lri, xrow[m], t[m]);
    }
  }
  __syncthreads();                                        // every read of the L_QQ blocks is done
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int c = 32 * Q + cg + 4 * m;
    if (c < r) S.L[potrf_lidx(c, r)] = t[m];
  }
  if ((rl & 3) == cg) S.L[potrf_lidx(r, r)] = inv_lrr;
  __syncthreads();
  // Block rows on the matrix cores: 16 x 16 output tiles (2 row halves x 2 P column tiles of block row P), tile index = wave, wave + 8;
  // fragments straight from the packed triangle: X[i][c] (i >= c) sits at xoff(c) + i, L[r][i] / X_PP[r][i] (i <= r) at lrow(i) + r.
  using V4 = typename GemmVec<T>::v4;
  const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  auto lrow = [](int i) { return i * (NB - 1) - (i * (i - 1)) / 2; };
  auto xoff = [](int c) { return c * NB - (c * (c - 1)) / 2 - c; };
#pragma unroll 1
  for (int P = 1; P < 4; ++P) {
    const int ntile = 4 * P;
    V4 acc[2];
    // (a) W = L_P,0..P-1 X_0..P-1,.   (k < column: X is zero, those k steps are skipped / masked)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      acc[s2] = V4{(T)0, (T)0, (T)0, (T)0};
      const int tile = wave + 8 * s2;
      if (tile < ntile) {
        const int rt = tile & 1, ct = tile >> 1;
        const int row = 32 * P + 16 * rt + l15, col = 16 * ct + l15;
        const int xo = xoff(col);
        for (int k0 = 16 * ct; k0 < 32 * P; k0 += 4) {
          const int k = k0 + l4;
          const T av = S.L[lrow(k) + row];
          const T bv = S.L[xo + k];
          acc[s2] = gemm_mfma16(av, (k >= col) ? bv : (T)0, acc[s2]);
        }
      }
    }
    __syncthreads();                                      // L_P,. is dead now
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int tile = wave + 8 * s2;
      if (tile < ntile) {
        const int rt = tile & 1, ct = tile >> 1;
        const int xo = xoff(16 * ct + l15);
#pragma unroll
        for (int q = 0; q < 4; ++q) S.L[xo + 32 * P + 16 * rt + (sizeof(T) == 8 ? l4 + 4 * q : 4 * l4 + q)] = acc[s2][q];
      }
    }
    __syncthreads();
    // (b) X_P,. = -X_PP W   (X_PP lower triangular: k <= row)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      acc[s2] = V4{(T)0, (T)0, (T)0, (T)0};
      const int tile = wave + 8 * s2;
      if (tile < ntile) {
        const int rt = tile & 1, ct = tile >> 1;
        const int row = 32 * P + 16 * rt + l15;
        const int xo = xoff(16 * ct + l15);
        for (int k0 = 32 * P; k0 < 32 * P + 16 * (rt + 1); k0 += 4) {
          const int k = k0 + l4;
          const T av = S.L[lrow(k <= row ? k : row) + row];
          const T bv = S.L[xo + k];
          acc[s2] = gemm_mfma16((k <= row) ? -av : (T)0, bv, acc[s2]);
        }
      }
    }
    __syncthreads();                                      // W is dead now
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int tile = wave + 8 * s2;
      if (tile < ntile) {
        const int rt = tile & 1, ct = tile >> 1;
        const int xo = xoff(16 * ct + l15);
#pragma unroll
        for (int q = 0; q < 4; ++q) S.L[xo + 32 * P + 16 * rt + (sizeof(T) == 8 ? l4 + 4 * q : 4 * l4 + q)] = acc[s2][q];
      }
    }
    __syncthreads();
  }
}

// (PH: phases to run, for the phase timings of pgpfa_bench_potrf_diag: bit 0 = Cholesky steps, bit 1 = inverse; production is 3)
// ALG 2 (default): Cholesky steps on 4 x 8 register blocks (potrf_diag_chol_steps2); 1: a row and 32 columns per thread (kept for the A/B of
// pgpfa_bench_potrf_diag).  (Tried and dropped in round 5: panels of 16 columns factored inside ONE wave - pivot and column entries by
// v_readlane, no barrier - with the rank-16 trailing updates on the matrix cores: a column still costs ~1000 cycles - the FP64 reciprocal
// square root and the 30 lane broadcasts of a step are one dependent chain - 70 us against 62, and 300 against 180 at batch 1024, where the
// other seven waves of every workgroup wait.)
template <typename T, int PH = 3, int ALG = 2>
__global__ __launch_bounds__(512) void potrf_diag_kernel_t(T* __restrict__ H, long long sH, int ld, int k0,
                                                            T* __restrict__ Dinv, long long sD,
                                                            const int* __restrict__ slots, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) PotrfLds<T> S;
  const int tid = threadIdx.x;
  const int r = tid & (NB - 1);
  const int cg = tid >> 7;                      // 0..3, uniform per wave
  const int pos_r = (r & 3) * 32 + (r >> 2);
  const long long slot = slots ? slots[blockIdx.x] : blockIdx.x;
  T* Hs = H + slot * sH + (size_t)k0 * ld + k0;
  T* Ds = Dinv + slot * sD + (size_t)(k0 / NB) * NB * NB;

  if constexpr (ALG == 2) {
    const int tr = tid & 31, tc = tid >> 5;
    T a2[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < 8; ++m) a2[i][m] = Hs[(size_t)(tc + 16 * m) * ld + tr + 32 * i];   // entries above the diagonal are never used
    if constexpr (PH & 1) {
      potrf_diag_chol_steps2<0>(a2, S, tr, tc, info, slot, k0);
      potrf_diag_chol_steps2<1>(a2, S, tr, tc, info, slot, k0);
      potrf_diag_chol_steps2<2>(a2, S, tr, tc, info, slot, k0);
      potrf_diag_chol_steps2<3>(a2, S, tr, tc, info, slot, k0);
      potrf_diag_chol_steps2<4>(a2, S, tr, tc, info, slot, k0);
      potrf_diag_chol_steps2<5>(a2, S, tr, tc, info, slot, k0);
      potrf_diag_chol_steps2<6>(a2, S, tr, tc, info, slot, k0);
      potrf_diag_chol_steps2<7>(a2, S, tr, tc, info, slot, k0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 8; ++m)
          if (tc + 16 * m <= tr + 32 * i) S.L[potrf_lidx(tc + 16 * m, tr + 32 * i)] = a2[i][m];
    }
  } else {
    T a[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) a[m] = Hs[(size_t)(cg + 4 * m) * ld + r];   // entries above the diagonal are never used
    if constexpr (PH & 1) {
      potrf_diag_chol_steps<0>(a, S, r, cg, pos_r, info, slot, k0);
      potrf_diag_chol_steps<1>(a, S, r, cg, pos_r, info, slot, k0);
      potrf_diag_chol_steps<2>(a, S, r, cg, pos_r, info, slot, k0);
      potrf_diag_chol_steps<3>(a, S, r, cg, pos_r, info, slot, k0);
    } else {
#pragma unroll
      for (int m = 0; m < 32; ++m)
        if (cg + 4 * m <= r) S.L[potrf_lidx(cg + 4 * m, r)] = a[m];
    }
  }
  __syncthreads();
  for (int e = tid; e < NB * NB; e += 512) {
    const int rr = e & (NB - 1), cc = e >> 7;
    if (rr >= cc) Hs[(size_t)cc * ld + rr] = S.L[potrf_lidx(cc, rr)];
  }
  // ---- X = L^-1 over the factor in LDS, then out (Ds[c][r] = X[r][c], zero above the diagonal)
  if constexpr (PH & 2) potrf_diag_inverse(S, tid, r, cg);
  for (int e = tid; e < NB * NB; e += 512) {
    const int rr = e & (NB - 1), cc = e >> 7;
    Ds[(size_t)cc * NB + rr] = (rr >= cc) ? S.L[potrf_lidx(cc, rr)] : (T)0;
  }
}

// Mt[jblk, jblk] = Dinv[jb]^T for the diagonal block at k0 + blockIdx.y * NB (grid = (slots, blocks): every diagonal block of a factorisation in one
// launch - they only depend on the factor; a launch per block column was 21 us of per-workgroup latency each, 26 launches per E-step)
template <typename T>
__global__ void diag_transpose_kernel_t(T* __restrict__ Mt, long long sM, int ld, int k0,
                                        const T* __restrict__ Dinv, long long sD, const int* __restrict__ slots) {
  const long long slot = slots ? slots[blockIdx.x] : blockIdx.x;
  k0 += blockIdx.y * NB;
  T* M = Mt + slot * sM + (size_t)k0 * ld + k0;
  const T* D = Dinv + slot * sD + (size_t)(k0 / NB) * NB * NB;
  // 32 x 32 sub-tiles through LDS: reads and writes both run along the contiguous dimension (a direct transposed read touched one
  // 8-byte word per 1 KB row: 27 us per launch of 1024 blocks).  blockDim.x = 256 = 8 rows of 32 lanes.
  __shared__ T tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int sub = 0; sub < 16; ++sub) {
    const int br = (sub & 3) * 32, bc = (sub >> 2) * 32;     // block of D: rows br.., columns bc.. (D[r * NB + c], c contiguous)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = D[(br + ty + 8 * i) * NB + bc + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) M[(size_t)(bc + ty + 8 * i) * ld + br + tx] = tile[tx][ty + 8 * i];   // M[c][r] = D[r][c]
  }
}

// --------------------------------------------------------------------------------------------------
// Newton step solve: v <- -(L L^T)^-1 v for each slot, one workgroup (256 threads) per slot.
// Also returns dec[slot] = -g.delta (Newton decrement squared) and smax[slot] = max |delta|.
// v lives in global memory (ld doubles per slot); rows >= n hold zeros.
// --------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256) void chol_solve_kernel(const double* __restrict__ H, long long sH, int ld, int npad,
                                                          const double* __restrict__ Dinv, long long sD,
                                                          const double* __restrict__ G, double* __restrict__ V,
                                                          long long sV, const int* __restrict__ slots,
                                                          double* __restrict__ dec, double* __restrict__ smax, int n) {
  __shared__ double yk[NB];
  __shared__ double zk[NB];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const long long slot = slots ? slots[blockIdx.x] : blockIdx.x;
  const double* L = H + slot * sH;
  const double* D = Dinv + slot * sD;
  const double* g = G + slot * sV;
  double* v = V + slot * sV;
  const int nblk = npad / NB;

  for (int i = tid; i < npad; i += 256) v[i] = (i < n) ? g[i] : 0.0;
  __syncthreads();
  // forward: y = L^-1 g
  for (int kb = 0; kb < nblk; ++kb) {
    const int k0 = kb * NB;
    if (tid < NB) yk[tid] = v[k0 + tid];
    __syncthreads();
    if (tid < NB) {
      const double* Dk = D + (size_t)kb * NB * NB;
      double s = 0.0;
      for (int c = 0; c <= tid; ++c) s += Dk[c * NB + tid] * yk[c];
      zk[tid] = s;
      v[k0 + tid] = s;
    }
    __syncthreads();
    for (int i = k0 + NB + tid; i < npad; i += 256) {
      const double* Lp = L + (size_t)k0 * ld + i;
      double s = 0.0;
#pragma unroll 8
      for (int c = 0; c < NB; ++c) s += Lp[(size_t)c * ld] * zk[c];
      v[i] -= s;
    }
    __syncthreads();
  }
  // backward: x = L^-T y
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb * NB;
    for (int c = wave; c < NB; c += 4) {
      const double* Lc = L + (size_t)(k0 + c) * ld;
      double s = 0.0;
      for (int i = k0 + NB + lane; i < npad; i += 64) s += Lc[i] * v[i];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
      if (lane == 0) zk[c] = v[k0 + c] - s;
    }
    __syncthreads();
    if (tid < NB) {
      const double* Dk = D + (size_t)kb * NB * NB;
      double s = 0.0;
      for (int c = tid; c < NB; ++c) s += Dk[tid * NB + c] * zk[c];   // (Dinv^T)[tid][c] = Dinv[c][tid]
      v[k0 + tid] = s;
    }
    __syncthreads();
  }
  // delta = -x ; dec = -g.delta = g.x ; smax = max|x|
  double d = 0.0, m = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double x = v[i];
    d += g[i] * x;
    m = fmax(m, fabs(x));
    v[i] = -x;
  }
  for (int off = 32; off > 0; off >>= 1) {
    d += __shfl_down(d, off);
    m = fmax(m, __shfl_down(m, off));
  }
  if (lane == 0) { red[wave] = d; red[4 + wave] = m; }
  __syncthreads();
  if (tid == 0) {
    dec[slot] = red[0] + red[1] + red[2] + red[3];
    smax[slot] = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  }
}

// flops of one factorisation / one transposed inverse of an npad-sized system (for reporting)
inline double chol_factor_flops(double n) { return n * n * n / 3.0; }

}  // namespace pgpfa
