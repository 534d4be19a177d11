// thin.h - the two block-diagonal products of the low-rank preconditioner application  y = F Sb F^T t  (pgpfa.hip: shared_solve)
// as kernels of their own.
//
// Both have one short dimension: per latent l, u_l = F_l^T t_l is (r_l x T)(T x slots) with r_l ~ 50-120, y_l = F_l v_l is
// (T x r_l)(r_l x slots).  Through the general LDS-tiled product (gemm.h: 64 x 64 tiles, block-sparse row tables, split-K and a
// reduction launch for the first) they ran at 9-15 TFLOP/s (profiles/r04_c3_gemm_shapes.txt): 160 output tiles of a 16-step k loop
// each, i.e. prologue, staging barriers and epilogue with little between them.  Here a wave feeds v_mfma_f64_16x16x4_f64 straight
// from global memory, no LDS staging and no barrier inside the loop.
//
// What decides the speed of such fragment loads is the texture addresser, not the cache (tools/probes/stride_probe.hip, measured on
// MI355X, data L2-resident): a wave-wide load is processed four lanes per clock when ADJACENT lanes read adjacent addresses (1 KB per
// 16-byte-per-lane instruction in 16 clocks: 36 TB/s over the chip), one lane per clock when they do not - and in the natural
// fragment layout (lane l15 <-> operand row, lane group l4 <-> k) adjacent lanes sit in different rows: 9.6 TB/s whatever the stride.
// The first form of these kernels read both operands of F^T t that way and ran at 21 us where the matrix instructions need 8.5.
// The matrix instruction does not care which 16 rows of the operand a tile holds nor in which order k is summed, so:
//   * the FACTOR operand is read along its contiguous direction with the four 16-row tiles of a wave interleaved - lane l15 holds
//     rows 4 l15 + (0..3), one 32-byte load per lane, 512 contiguous bytes per 16 lanes: rank rows of the transposed copy (FTbig) for
//     F^T t, bins of F_l for F v;
//   * the VECTOR operand (slot-major, contiguous in k) keeps the natural layout - lane group l4 takes bins / rank columns
//     16 kb + 4 l4 + (0..3) as the four k steps of a block, one 32-byte load per lane, the slow kind, 1 of 5 loads;
//   * the result of a lane is then 4 consecutive rows of one slot: 32-byte stores.
// F^T t: a workgroup owns (latent, up to 64 rank rows, 16 slots); its eight waves split the bins and meet in LDS once at the end (fixed
// order: the result does not depend on the launch).  F v: a workgroup owns (latent, up to 256 bins, 16 slots), a wave 64 bins.
// Columns are the slots of a (device-side) list as in gemm.h: `cols` maps column position -> slot, `n_dev` holds how many there are.
// The slot tile is the SLOW grid index: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs and, inside an XCD, over its 4
// shader engines; with the slot tile as the fast index of a 64-wide grid the workgroups of a short live list (tiles 0..6, ids = 0..6 mod 64)
// all met on ONE engine of every XCD - a quarter of the CUs, two rounds of workgroups where one fits (tools/probes/thin_probe.hip prints
// the placement).
#pragma once
#include <hip/hip_runtime.h>
#include "types.h"

namespace pgpfa {

struct ThinP {
  const double* F; int Tf; int T;                // factors: element (t, j) of latent l at F[l Tf Tf + j Tf + t]; rows >= T and columns >= rank hold zeros
  const double* FT; int ldft;                    // transposed, all latents: element (t, j) of latent l at FT[(l T + t) ldft + roff_l + j], zero outside the blocks
  const int* tab;                                // work table, four ints per blockIdx.y: (latent, first rank row, rank rows, rank offset) | (latent, first bin, rank, rank offset)
  const double* X; long long ldx;                // input vectors, one column per slot
  double* Y; long long ldy;                      // output vectors
  int Tx;                                        // row stride between latents inside an n-vector (input of F^T t, output of F v): T, or T rounded up to 16
                                                 // in the inner solve's private layout (pcg.h: PcgCgP::Tl); a multiple of 4 where VEC4 is used
  const int* cols; const int* n_dev; int ncols;  // column list (null: identity), device-side count (null: ncols)
  const int* skip;                               // device stop flag of the inner solve (null: none)
#ifdef THIN_STAMPS
  unsigned long long* stamps;                    // tools/probes/thin_probe.hip: 8 clock stamps per workgroup
#endif
};
#ifdef THIN_STAMPS
#define THIN_STAMP(k) do { if (threadIdx.x == 0 && wgid < 4096) { a.stamps[8 * wgid + (k)] = wall_clock64(); if ((k) == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); a.stamps[8 * wgid + 7] = ((unsigned long long)xcc << 32) | hw; } } } while (0)
#else
#define THIN_STAMP(k) do { } while (0)
#endif

// u[roff_l + j, s] = sum_t F_l[t, j] x[l T + t, s].  grid = (entries of the (latent, row group) table, ceil(ncols / 16)), block = 512.
// VEC4: T % 4 == 0, so that the 4-bin runs of the vectors start on 32-byte boundaries (ldx is a multiple of 128).
// The eight waves of a workgroup split the bins; a wave fetches FOUR 16-bin blocks at a time - twenty 32-byte loads in flight - and then
// multiplies them: with the chip a third full (the live list of the inner solve shrinks) the kernel is a chain of memory round trips, and
// this makes it one or two of them (double-buffered single blocks: eight).  No branch encloses a load: the compiler counts outstanding loads
// per path and waits for ALL of them after a join.
#ifdef THIN_KB_OVERRIDE
constexpr int THIN_KB = THIN_KB_OVERRIDE;        // (tools/probes/Makefile builds the probe with 4 as well)
#else
constexpr int THIN_KB = 2;
#endif
// TX: storage type of the INPUT n-vectors (float: the inner solve's single-precision t, widened as it is loaded; the products stay FP64)
template <bool VEC4, typename TX = double>
__global__ __launch_bounds__(512, 2) void thin_ft_kernel(ThinP a) {
  if (a.skip && *a.skip) return;
  const int ncols = a.n_dev ? min(*a.n_dev, a.ncols) : a.ncols;
  const int s0 = blockIdx.y * 16;
  if (s0 >= ncols) return;
#ifdef THIN_STAMPS
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
#endif
  THIN_STAMP(0);
  __shared__ double red[8][4][4][64];            // [wave][row tile][accumulator register][lane]
  const int l = a.tab[4 * blockIdx.x], m0 = a.tab[4 * blockIdx.x + 1], rows = a.tab[4 * blockIdx.x + 2], r0 = a.tab[4 * blockIdx.x + 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  // (position clamped by the host's bound, not the device count: entries of a live list past its count are slots that left it - readable -
  //  and this load does not wait for the count)
  const int sc = min(s0 + l15, a.ncols - 1);
  const int slot = a.cols ? a.cols[sc] : sc;
  const TX* xp = reinterpret_cast<const TX*>(a.X) + (size_t)slot * a.ldx + (size_t)l * a.Tx;
  // row tile mi of this lane: rank row m0 + 4 l15 + mi (rows past the group's read neighbours' zeros or padding and are not stored)
  const double* fp = a.FT + (size_t)l * a.T * a.ldft + r0 + m0 + 4 * l15;
  const int nkb = (a.T + 15) >> 4;
  const int kb0 = (wave * nkb) >> 3, kb1 = ((wave + 1) * nkb) >> 3;
  double4_t acc[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) acc[mi] = double4_t{0.0, 0.0, 0.0, 0.0};
  THIN_STAMP(1);
#pragma unroll 1
  for (int kb = kb0; kb < kb1; kb += THIN_KB) {
    double4_t x[THIN_KB], f[THIN_KB][4];
    // bins past T, blocks past the wave's share: the factor row is clamped (finite), the vector entry is zero
#pragma unroll
    for (int u = 0; u < THIN_KB; ++u) {
      const int t0 = min(kb + u, kb1 - 1) * 16 + 4 * l4;
      const bool in = kb + u < kb1;
      if (VEC4) {
        if constexpr (sizeof(TX) == 4) {
          const float4_t xf = *reinterpret_cast<const float4_t*>(xp + min(t0, a.T - 4));
          x[u] = double4_t{(double)xf[0], (double)xf[1], (double)xf[2], (double)xf[3]};
        } else {
          x[u] = *reinterpret_cast<const double4_t*>(xp + min(t0, a.T - 4));
        }
        if (t0 >= a.T || !in) x[u] = double4_t{0.0, 0.0, 0.0, 0.0};
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const double v = (double)xp[min(t0 + i, a.T - 1)]; x[u][i] = (t0 + i < a.T && in) ? v : 0.0; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) f[u][i] = *reinterpret_cast<const double4_t*>(fp + (size_t)min(t0 + i, a.T - 1) * a.ldft);
    }
#ifdef THIN_STAMPS
    if (kb == kb0) { THIN_STAMP(2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); THIN_STAMP(3); }
#endif
#pragma unroll
    for (int u = 0; u < THIN_KB; ++u)
      if (kb + u < kb1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][i], f[u][i][mi], acc[mi], 0, 0, 0);
      }
  }
  THIN_STAMP(4);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][mi][r][lane] = acc[mi][r];
  __syncthreads();
  THIN_STAMP(5);
  // wave w < 4 finishes accumulator register w: slot position s0 + l4 + 4 w, rank rows m0 + 4 l15 + (0..3)
  const int sp = s0 + l4 + 4 * wave;
  if (wave >= 4 || sp >= ncols || 4 * l15 >= rows) return;
  double4_t v;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
    v[mi] = ((red[0][mi][wave][lane] + red[1][mi][wave][lane]) + (red[2][mi][wave][lane] + red[3][mi][wave][lane])) +
            ((red[4][mi][wave][lane] + red[5][mi][wave][lane]) + (red[6][mi][wave][lane] + red[7][mi][wave][lane]));
  const int so = a.cols ? a.cols[sp] : sp;
  *reinterpret_cast<double4_t*>(a.Y + (size_t)so * a.ldy + r0 + m0 + 4 * l15) = v;
  THIN_STAMP(6);
}

// y[l T + t, s] = sum_j F_l[t, j] v[roff_l + j, s].  grid = (entries of the (latent, 256-bin group) table, ceil(ncols / 16)), block = 256.
// A wave owns 64 bins, tb0 + 64 wave .. ; bin tile j holds bins 4 l15 + j of them.  Four 16-column k blocks are fetched at a time (see above).
// TY: storage type of the OUTPUT n-vectors
template <bool VEC4, typename TY = double>
__global__ __launch_bounds__(256, 2) void thin_f_kernel(ThinP a) {
  if (a.skip && *a.skip) return;
  const int ncols = a.n_dev ? min(*a.n_dev, a.ncols) : a.ncols;
  const int s0 = blockIdx.y * 16;
  if (s0 >= ncols) return;
  const int l = a.tab[4 * blockIdx.x], tb0 = a.tab[4 * blockIdx.x + 1], rk = a.tab[4 * blockIdx.x + 2], r0 = a.tab[4 * blockIdx.x + 3];
  const int nkb = rk >> 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int tw = tb0 + 64 * wave;                            // first bin of the wave
  if (tw >= a.T) return;                                     // (no barrier in this kernel)
#ifdef THIN_STAMPS
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
#endif
  THIN_STAMP(0);
  const int sc = min(s0 + l15, a.ncols - 1);
  const int slot = a.cols ? a.cols[sc] : sc;
  const double* vp = a.X + (size_t)slot * a.ldx + r0 + 4 * l4;
  // the lane's factor entries of k block kb, step i: column 16 kb + 4 l4 + i, bins tw + 4 l15 + (0..3).  Bins past T read rows of the slab that hold
  // zeros or - past Tf - its neighbours' (finite; the allocation has slack); they are not stored.
  const double* fp = a.F + (size_t)l * a.Tf * a.Tf + (size_t)(4 * l4) * a.Tf + tw + 4 * l15;
  double4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
  THIN_STAMP(1);
#pragma unroll 1
  for (int kb = 0; kb < nkb; kb += THIN_KB) {
    double4_t x[THIN_KB], f[THIN_KB][4];
#pragma unroll
    for (int u = 0; u < THIN_KB; ++u) {
      const int k = min(kb + u, nkb - 1);
      x[u] = *reinterpret_cast<const double4_t*>(vp + k * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) f[u][i] = *reinterpret_cast<const double4_t*>(fp + (size_t)(16 * k + i) * a.Tf);
    }
#ifdef THIN_STAMPS
    if (kb == 0) { THIN_STAMP(2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); THIN_STAMP(3); }
#endif
#pragma unroll
    for (int u = 0; u < THIN_KB; ++u)
      if (kb + u < nkb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][i], f[u][i][j], acc[j], 0, 0, 0);
      }
  }
  THIN_STAMP(4);
  THIN_STAMP(5);
  // accumulator register r of a lane: slot position s0 + l4 + 4 r; tile j: bin tw + 4 l15 + j
  const int t = tw + 4 * l15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int sp = s0 + l4 + 4 * r;
    if (sp >= ncols) continue;
    const int so = a.cols ? a.cols[sp] : sp;
    TY* yp = reinterpret_cast<TY*>(a.Y) + (size_t)so * a.ldy + (size_t)l * a.Tx;
    if (VEC4) {
      if constexpr (sizeof(TY) == 4) {
        if (t < a.T) *reinterpret_cast<float4_t*>(yp + t) = float4_t{(float)acc[0][r], (float)acc[1][r], (float)acc[2][r], (float)acc[3][r]};
      } else {
        if (t < a.T) *reinterpret_cast<double4_t*>(yp + t) = double4_t{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (t + j < a.T) yp[t + j] = (TY)acc[j][r];
    }
  }
  THIN_STAMP(6);
}

}  // namespace pgpfa
