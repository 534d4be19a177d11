// Yt = F L^-T and the mixing pass of the split covariance form in ONE kernel (round 5).
//
// Under the split form (split.h) Yt has a single consumer: the mixing pass, which reads every column of it once, writes the single-precision
// correction D = (I - G_t) y and sums post_vsm[t] = eps G_t + sum_b (G_t y_b)(G_t y_b)^T.  Written by p batched products with K = r_k (32..96:
// tiles that live for two dependent memory round trips and 1.7 us of matrix-core work) and read back by the mixing pass, Yt cost 6 + 6 ms per EM
// iteration at config 3 and 24 GB of HBM traffic for numbers nobody keeps.  Here a wave owns 16 bins of a slot and walks the columns of Yt in blocks
// of 16: for each block it forms the 16 x 16 tiles of all p latents on the FP64 matrix cores (v_mfma_f64_16x16x4: the lane (l15, l4) then holds bin
// l15 and columns l4 + 4 r of every latent - a bin's p-vector of one column sits in ONE lane), mixes them in registers against G_t (LDS, one image
// per workgroup), stores D and adds the pair sums of post_vsm.  Yt never exists in memory.
//
// Operands: the column block's panel of L^-T (rows 0 .. b0 + 15 - it is upper triangular - of 16 columns) is staged once per workgroup in LDS, in
// chunks of YTM_KC rows; the rows of F_k are read straight from global memory (F is shared by every slot: 1.2 MB at config 3, L2-resident), four
// K steps (one 16-row group) ahead of the matrix cores.
//
// Same arithmetic as gemm_mfma_kernel_t + mix_slot2_kernel up to the order of the sums over K and over the columns.
// grid = ceil(T / YTM_BINS) * nslots workgroups (1-D, remapped so that the workgroups of a slot share an XCD and its L2), block = YTM_THREADS,
// dynamic LDS = ytmix_lds(PW); p <= PW <= 10 (the widths dispatch_pw instantiates), ranks and offsets multiples of 16 (they are: build_lowrank),
// ract = roff[p].
#pragma once
#include <hip/hip_runtime.h>

#include "split.h"

namespace pgpfa {

constexpr int YTM_KC = 256;                  // rows of the L^-T panel staged at a time
constexpr int YTM_AS = YTM_KC + 2;           // its LDS row stride: 2 mod 32 doubles, the 16 x 2 lanes of half a fragment read hit 32 different bank pairs
constexpr int YTM_WAVES = 8;                 // waves (of 16 bins) per workgroup: they share one staged panel
constexpr int YTM_BINS = 16 * YTM_WAVES;
constexpr int YTM_THREADS = 64 * YTM_WAVES;

inline size_t ytmix_lds(int pw) { return ((size_t)(pw * (pw + 1) / 2) * YTM_BINS + 2 * 16 * (size_t)YTM_AS) * sizeof(double); }

struct YtMixArgs {
  const double* F; int Tp;                   // [p][Tp x Tp] column-major low-rank factors of the prior
  const double* Mts; long long sM; int rpad; // L^-T per slot (upper triangular, column-major, ld = rpad)
  float* D; long long sD; int ldd;           // correction per slot: D[b * ldd + k * ts + t]
  const double* G; long long sG;             // per-bin blocks (I + eps W)^-1 per slot: [T][p x p]
  int T, p, ract, nbx, nslots; double eps;     // p <= PW latents (7 and 9 run the 8- and 10-wide instantiations with empty latents behind)
  double* vsm;                               // post_vsm[(trial * T + t)][p x p]
  const int* slots; const int* trial_of_slot;
  const int* roff;                           // [p + 1] rank offsets of the ROWS the panel is walked in (16-aligned: roff16 under compact offsets)
  // (round 6) compact rank offsets (core.hip build_lowrank): the columns of L^-T - and of D - are in the compact order, the staged panel keeps the
  // padded rows the product loop walks: panel row i is row cmap[i] of the slab (-1: a padding row, staged as zeros), and block b0 has nrtab[b0 / 16]
  // panel rows that can hold something.  cmap = null: rows as they are, b0 + 16 of them.
  const int* cmap; const int* nrtab;
  int ncmap;                                 // entries of cmap (copied into LDS behind the panel images: a staged pair's row is then an LDS read away, not a dependent memory round trip)
  int ts;
  int dbg;                                   // timing experiments only (option yt_mix_dbg; results are wrong when set): 1 no F loads, 2 no products, 4 no mixing, 8 no D stores, 16 no panel staging, 32 no barriers, 64 no panel loads
};

template <int PW>
__global__ __launch_bounds__(YTM_THREADS, 512 / YTM_THREADS) void yt_mix_kernel(YtMixArgs a) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  constexpr int NPAIR = PW * (PW + 1) / 2;
  const int p = a.p, pp = p * p;
  extern __shared__ double ytm_lds[];
  double* Gs = ytm_lds;                      // [NPAIR][YTM_BINS]
  double* As = ytm_lds + NPAIR * YTM_BINS;   // [2][16][YTM_AS]
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups are handed to the XCDs round-robin: give XCD x the x-th eighth of the (slot, bin block) list, so a slot's panels stay in one L2
  int w = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) w = (w & 7) * (total >> 3) + (w >> 3);
  const int slot = a.slots[w / a.nbx];
  const int t0 = (w % a.nbx) * YTM_BINS;
  const int T = a.T;
  {
    const int bin = tid % YTM_BINS;
    const int part = __builtin_amdgcn_readfirstlane(tid / YTM_BINS);
    const int tb = t0 + bin < T ? t0 + bin : T - 1;
    const double* gsrc = a.G + (size_t)slot * a.sG + (size_t)tb * pp;
#pragma unroll
    for (int hi = 0; hi < PW; ++hi)
#pragma unroll
      for (int lo = 0; lo <= hi; ++lo) {
        const int idx = hi * (hi + 1) / 2 + lo;
        if ((idx & 3) == part) Gs[idx * YTM_BINS + bin] = hi < p ? gsrc[hi * p + lo] : 0.0;
      }
  }
  int ro[PW + 1];
#pragma unroll
  for (int k = 0; k <= PW; ++k) ro[k] = a.roff[k < p ? k : p];        // (latents p .. PW - 1: no rows)
  const int t = t0 + wave * 16 + l15;
  const int tc = t < T ? t : T - 1;
  const double* gl = Gs + wave * 16 + l15;
  const double* Ms = a.Mts + (size_t)slot * a.sM;
  const __amdgpu_buffer_rsrc_t d = wave_uniform_rsrc(a.D + (size_t)slot * a.sD, (size_t)a.ldd * a.ract * sizeof(float));   // (ract columns of ldd floats)
  const __amdgpu_buffer_rsrc_t fr = wave_uniform_rsrc(a.F, (size_t)p * a.Tp * a.Tp * sizeof(double));
  const unsigned flane = ((unsigned)l4 * (unsigned)a.Tp + (unsigned)tc) * 8u;
  const unsigned dlane = ((unsigned)l4 * (unsigned)a.ldd + (unsigned)tc) * 4u;
  double sums[NPAIR];
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) sums[i] = 0.0;
  // The panel of a column block goes through registers into one of TWO LDS images, in steps of YTM_KC rows: the waves that have finished with one
  // image request the next step's rows (panel_fetch: all of a thread's loads go out together), mask them and write the other image (panel_write) -
  // one barrier per step, behind that write.
  constexpr int TPC = YTM_THREADS / 16, NJ = YTM_KC / (2 * TPC);     // threads per column, row pairs per thread
  const int pcol = tid / TPC, prp = (tid % TPC) * 2;
  v2d tmp[NJ];
  int trow[NJ];                                               // slab row of the pair a thread has in flight (compact offsets: through cmap)
#pragma unroll
  for (int j = 0; j < NJ; ++j) { tmp[j] = v2d{0.0, 0.0}; trow[j] = 0; }
  int* const cmap_s = reinterpret_cast<int*>(As + 2 * 16 * YTM_AS);
  if (a.cmap)
    for (int e = tid; e < a.ncmap; e += YTM_THREADS) cmap_s[e] = a.cmap[e];
  // (ncmap = 0: the map did not fit LDS next to the images - very long rank totals - and is read where it lies)
  const int* const cmap = a.cmap ? (a.ncmap > 0 ? cmap_s : a.cmap) : nullptr;
  if (a.cmap && a.ncmap > 0) __syncthreads();
  auto nr_of = [&](int b0) { return cmap ? a.nrtab[b0 >> 4] : b0 + 16; };
  auto panel_fetch = [&](int b0, int kc0, int kc1) {
    const double* src = Ms + (size_t)(b0 + pcol) * a.rpad;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = kc0 + prp + 2 * TPC * j;
      // (pairs never split: real rows of a latent come in multiples of 4 from a multiple of 4)
      // (assigned on every path: left alone under the branch, the four pairs and their rows were carried around the block loop - through the
      //  mixing - as values that might still be needed: 20 registers)
      tmp[j] = v2d{0.0, 0.0};
      trow[j] = -1;
      if (i < kc1 && !(a.dbg & 64)) {
        const int ic = cmap ? cmap[i] : i;
        trow[j] = ic;
        tmp[j] = *reinterpret_cast<const v2d*>(src + (ic < 0 ? 0 : ic));
      }
    }
  };
  auto panel_write = [&](int buf, int b0, int kc0, int kc1) {
    double* dst = As + (buf * 16 + pcol) * YTM_AS - kc0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = kc0 + prp + 2 * TPC * j;
      if (i < kc1) {
        v2d v = tmp[j];
        const int ic = trow[j];
        if (ic < 0 || ic > b0 + pcol) v.x = 0.0;              // (padding row; below the diagonal: whatever an earlier use of the slab left there)
        if (ic < 0 || ic + 1 > b0 + pcol) v.y = 0.0;
        *reinterpret_cast<v2d*>(dst + i) = v;
      }
    }
  };
  const bool staged = !(a.dbg & 16);
  {
    const int nk0 = nr_of(0) < YTM_KC ? nr_of(0) : YTM_KC;
    if (staged) { panel_fetch(0, 0, nk0); panel_write(0, 0, 0, nk0); }
  }
  __syncthreads();                                            // (also: the G image is written)
  int buf = 0;
  for (int b0 = 0; b0 < a.ract; b0 += 16) {
    const int nr = nr_of(b0);                                 // rows of the panel that can hold something
    v4d acc[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) acc[k] = v4d{0.0, 0.0, 0.0, 0.0};
    // The rows of the block's panel are walked in groups of 16, latent after latent.  Two fragment sets of F (four K steps for the wave's bins each):
    // x holds the group about to be multiplied, y the one behind it; a set is requested again - for the group two ahead, whatever latent that row
    // belongs to - as soon as its products are issued, so a fragment has two groups of matrix-core time (2 x 256 cycles, and the other wave's) to
    // arrive from L2.  The loop body is a PAIR of groups so that no register is copied while its load is in flight; a latent with an odd number
    // of groups ends with one copy of the older set.  (One group ahead with a copy per group: the copy waited for the load just issued.)
    auto frow_of = [&](int row) {                             // row of F (all latents as one table of p * Tp rows) of panel row `row`
      int o = 0;
#pragma unroll
      for (int j = 1; j < PW; ++j) o = row >= ro[j] ? j * a.Tp - ro[j] : o;
      return o + row;
    };
    auto fetch4 = [&](int row, double& x0, double& x1, double& x2, double& x3) {
      // (unconditional: past the block's last group the last one is read again - behind a branch the compiler stops counting the loads in flight
      //  and waits for all of them at the head of the loop)
      const int fr0 = frow_of((row < nr && !(a.dbg & 1)) ? row : nr - 16);
      x0 = rsrc_load_f64(fr, flane, (unsigned)(fr0 * a.Tp) * 8u); x1 = rsrc_load_f64(fr, flane, (unsigned)((fr0 + 4) * a.Tp) * 8u);
      x2 = rsrc_load_f64(fr, flane, (unsigned)((fr0 + 8) * a.Tp) * 8u); x3 = rsrc_load_f64(fr, flane, (unsigned)((fr0 + 12) * a.Tp) * 8u);
    };
    double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0, y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
    if (!(a.dbg & 2)) {
      const int fr0 = frow_of(0), fr1 = frow_of(16 < nr ? 16 : 0);
      x0 = rsrc_load_f64(fr, flane, (unsigned)(fr0 * a.Tp) * 8u); x1 = rsrc_load_f64(fr, flane, (unsigned)((fr0 + 4) * a.Tp) * 8u);
      x2 = rsrc_load_f64(fr, flane, (unsigned)((fr0 + 8) * a.Tp) * 8u); x3 = rsrc_load_f64(fr, flane, (unsigned)((fr0 + 12) * a.Tp) * 8u);
      y0 = rsrc_load_f64(fr, flane, (unsigned)(fr1 * a.Tp) * 8u); y1 = rsrc_load_f64(fr, flane, (unsigned)((fr1 + 4) * a.Tp) * 8u);
      y2 = rsrc_load_f64(fr, flane, (unsigned)((fr1 + 8) * a.Tp) * 8u); y3 = rsrc_load_f64(fr, flane, (unsigned)((fr1 + 12) * a.Tp) * 8u);
    }
    for (int kc0 = 0; kc0 < nr; kc0 += YTM_KC) {
      const int kc1 = kc0 + YTM_KC < nr ? kc0 + YTM_KC : nr;
      const double* arow = As + (buf * 16 + l15) * YTM_AS + l4 - kc0;
      if (!(a.dbg & 2))
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const int kb = ro[k] > kc0 ? ro[k] : kc0, ke = ro[k + 1] < kc1 ? ro[k + 1] : kc1;
        int i = kb;
        for (; i + 32 <= ke; i += 32) {
          const double* ar = arow + i;
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[0], x0, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[4], x1, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[8], x2, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[12], x3, acc[k], 0, 0, 0);
          fetch4(i + 32, x0, x1, x2, x3);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[16], y0, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[20], y1, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[24], y2, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[28], y3, acc[k], 0, 0, 0);
          fetch4(i + 48, y0, y1, y2, y3);
        }
        if (i < ke) {
          const double* ar = arow + i;
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[0], x0, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[4], x1, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[8], x2, acc[k], 0, 0, 0);
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[12], x3, acc[k], 0, 0, 0);
          x0 = y0; x1 = y1; x2 = y2; x3 = y3;
          fetch4(i + 32, y0, y1, y2, y3);
        }
      }
      if (kc1 < nr) {                                         // this block's next chunk (nothing to hide it under)
        const int nk1 = kc1 + YTM_KC < nr ? kc1 + YTM_KC : nr;
        if (staged) { panel_fetch(b0, kc1, nk1); panel_write(buf ^ 1, b0, kc1, nk1); if (!(a.dbg & 32)) __syncthreads(); }
        buf ^= 1;
      }
    }
    const int nb0 = b0 + 16;                                  // the next column block and the first chunk of its panel
    const bool more = staged && nb0 < a.ract;
    const int nrn = more ? nr_of(nb0) : 16;
    const int nk1 = nrn < YTM_KC ? nrn : YTM_KC;
    // mixing: register r of a lane is column b0 + l4 + 4 r of bin l15, one p-vector per latent set.  TWO columns per read of G_t (round 6): the
    // phase was bound by those reads - one 8-byte LDS read per lane for two multiply-adds, i.e. 4 cycles of the CU's one LDS port against 8 of a
    // SIMD's vector pipe, four SIMDs to the port - not by its arithmetic; the second column's p-vector costs 2 PW registers at a point where the
    // fragment sets of F and the staging registers are dead.
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      double m[PW], n[PW];
#pragma unroll
      for (int k = 0; k < PW; ++k) { m[k] = 0.0; n[k] = 0.0; }
      if (!(a.dbg & 4))
#pragma unroll
      for (int hi = 0; hi < PW; ++hi)
#pragma unroll
        for (int lo = 0; lo <= hi; ++lo) {
          const double gg = gl[(hi * (hi + 1) / 2 + lo) * YTM_BINS];
          m[hi] += gg * acc[lo][r];
          n[hi] += gg * acc[lo][r + 1];
          if (lo != hi) { m[lo] += gg * acc[hi][r]; n[lo] += gg * acc[hi][r + 1]; }
        }
      if (t < T && !(a.dbg & 8)) {
#pragma unroll
        for (int k = 0; k < PW; ++k)
          if (k < p) {
            rsrc_store_f32(d, dlane, (unsigned)((b0 + 4 * r) * a.ldd + k * a.ts) * 4u, (float)(acc[k][r] - m[k]));
            rsrc_store_f32(d, dlane, (unsigned)((b0 + 4 * r + 4) * a.ldd + k * a.ts) * 4u, (float)(acc[k][r + 1] - n[k]));
          }
      }
#pragma unroll
      for (int hi = 0; hi < PW; ++hi)
#pragma unroll
        for (int lo = 0; lo <= hi; ++lo) sums[hi * (hi + 1) / 2 + lo] += m[hi] * m[lo];
#pragma unroll
      for (int hi = 0; hi < PW; ++hi)
#pragma unroll
        for (int lo = 0; lo <= hi; ++lo) sums[hi * (hi + 1) / 2 + lo] += n[hi] * n[lo];
      asm volatile("" ::: "memory");                          // (the reads of G_t stay inside their column pair: 55 values kept across the two pairs would not fit)
    }
    // the first chunk of the next block into the other image (requested under the mixing it bought nothing measurable for eight registers)
    if (more) { panel_fetch(nb0, 0, nk1); panel_write(buf ^ 1, nb0, 0, nk1); }
    if (staged && !(a.dbg & 32)) __syncthreads();
    buf ^= 1;
  }
  // the four lanes of a bin (l4 = 0..3) hold the sums over their columns
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) {
    sums[i] += __shfl_xor(sums[i], 16);
    sums[i] += __shfl_xor(sums[i], 32);
  }
  if (l4 != 0 || t >= T) return;
  double* vdst = a.vsm + ((size_t)a.trial_of_slot[slot] * T + t) * pp;
#pragma unroll
  for (int hi = 0; hi < PW; ++hi)
#pragma unroll
    for (int lo = 0; lo <= hi; ++lo) {
      if (hi >= p) continue;
      const double val = a.eps * gl[(hi * (hi + 1) / 2 + lo) * YTM_BINS] + sums[hi * (hi + 1) / 2 + lo];
      vdst[hi * p + lo] = val;
      vdst[lo * p + hi] = val;
    }
}

}  // namespace pgpfa
