// Sum over trials of the per-latent covariance blocks of the low-rank engine without the full-width FP64 product.
//
// With G_t = (I + eps W_t)^-1 = I - eps Wt_t (Wt = W G), the mixed slab is  Y~ = G Yt = Yt - D,  D = eps Wt Yt,  so
//
//   sum_r Y~_k Y~_k^T  =  F_k [sum_r A_rk A_rk^T] F_k^T  -  F_k [sum_r A_rk D_rk^T]  -  (.)^T  +  sum_r D_rk D_rk^T
//
// with A_rk = rows of latent k of L_r^-T (r_k x r) and Yt_k = F_k A_rk: EXACT algebra, no approximation.  What the split buys is where
// the work sits: the first term is r_k x r_k per trial (the trial-independent F_k leaves the sum), the cross term is T r^2 per trial
// (15 % of the full-width p T^2 r) and stays FP64 (its operand D is stored in single precision: it carries a factor eps ||Wt|| ~ 1e-2,
// so its 6e-8 rounding is 6e-10 of the result at worst, ~1e-11 in the root-mean-square), and the full-width term D D^T carries
// (eps ||Wt||)^2 ~ 1e-4: it runs on the FP16 matrix cores (16x the FP32 rate, 32x FP64) with every entry split in two halves,
// d * 2^11 = hi + lo, three products hi hi + hi lo + lo hi with FP32 accumulation (relative error of the term 2^-21 + FP32
// accumulation over <= 16 trials ~ 2e-6: below 1e-9 of the result), partial sums per group of trials reduced in FP64.
// The caller measures eps ||Wt|| per chunk and keeps the FP64 product above a threshold (pgpfa.hip).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

namespace pgpfa {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4v_t __attribute__((ext_vector_type(4)));

constexpr float SPLIT_SCALE = 2048.0f;          // |D| <= sqrt(p) (rows of Yt have norm <= 1, eps ||Wt|| < 1): 2^11 |D| < 65504 up to p = 1000

// (Tried in round 4 and dropped: storing D already split - hi | lo << 16 in one 32-bit word written by the mixing pass - so that the FP16
// kernel stops converting per tile.  The cross term then has to read d back as (hi + lo) 2^-11, and lo underflows into FP16 denormals for
// |d| < 1e-4: tools/probes/split_pack_probe.hip measures 1.4e-5 relative error at |d| = 1e-6, PautoSum lands at 2e-7 .. 2e-6 instead of 1e-9.
// Writing both forms costs the mixing pass what the FP16 kernel would save.)

// Mixing pass of the split form: like mix_vsm_kernel it accumulates post_vsm[t] = eps G_t + sum_b (G_t y_b)(G_t y_b)^T over the columns
// of the slab, but leaves Yt as it is and writes the correction D[(k,t), b] = y - G_t y (= eps Wt_t y) in single precision
// (column stride ldd floats, latent stride ts) - half the bytes of the in-place FP64 mix.
// grid = (ceil(T/64), nslots), block = 256, p <= PW <= 16.
template <int PW>
__global__ __launch_bounds__(256, 2) void mix_vsm_split_kernel(const double* __restrict__ Yt, long long sY, int ldy, float* __restrict__ D, long long sD,
                                                            int ldd, const double* __restrict__ G, long long sG, int T, int p, int rpad, double eps,
                                                            double* __restrict__ vsm, const int* __restrict__ slots,
                                                            const int* __restrict__ trial_of_slot, const int* __restrict__ roff, int col_tile, int ts) {
  constexpr int PP = PW * PW, LD = PP + 1, NPAIR = PW * (PW + 1) / 2;
  __shared__ double Gs[64 * LD];
  const int pp = p * p;
  const int slot = slots[blockIdx.y];
  const int t0 = blockIdx.x * 64;
  const int nt = min(64, T - t0);
  const double* gbase = G + (size_t)slot * sG + (size_t)t0 * pp;
  if (p < PW)
    for (int e = threadIdx.x; e < 64 * LD; e += 256) Gs[e] = 0.0;
  __syncthreads();
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const int t = e / pp, idx = e - t * pp, i = idx / p, j = idx - i * p;
    Gs[t * LD + i * PW + j] = gbase[e];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = lane < nt;
  double* g = Gs + lane * LD;
  double acc[NPAIR];
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) acc[i] = 0.0;
  int c0[PW];
#pragma unroll
  for (int k = 0; k < PW; ++k) c0[k] = (roff && k < p) ? (roff[k] / col_tile) * col_tile : 0;
  if (live) {
    const double* y = Yt + (size_t)slot * sY + t0 + lane;
    float* d = D + (size_t)slot * sD + t0 + lane;
    for (int b = wave; b < rpad; b += 4) {
      if constexpr (PW > 10) asm volatile("" ::: "memory");
      double v[PW], m[PW];
#pragma unroll
      for (int k = 0; k < PW; ++k) v[k] = (k < p && b >= c0[k]) ? y[(size_t)b * ldy + (size_t)k * ts] : 0.0;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        double s2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < PW; ++kk) s2 += g[(k >= kk) ? k * PW + kk : kk * PW + k] * v[kk];     // (G_t is symmetric: 55 values to keep, not 100)
        m[k] = s2;
        if (k < p) d[(size_t)b * ldd + (size_t)k * ts] = (float)(v[k] - s2);
      }
#pragma unroll
      for (int a = 0; a < PW; ++a)
#pragma unroll
        for (int c2 = 0; c2 <= a; ++c2) acc[a * (a + 1) / 2 + c2] += m[a] * m[c2];
    }
  }
  // fold the four waves' sums into the LDS block: wave 0 turns G into eps G + acc, the others add theirs
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w && live) {
#pragma unroll
      for (int a = 0; a < PW; ++a)
#pragma unroll
        for (int c2 = 0; c2 <= a; ++c2) {
          const double base = (w == 0) ? eps * g[a * PW + c2] : g[a * PW + c2];
          const double val = base + acc[a * (a + 1) / 2 + c2];
          g[a * PW + c2] = val;
          g[c2 * PW + a] = val;
        }
    }
  }
  __syncthreads();
  double* vbase = vsm + ((size_t)trial_of_slot[slot] * T + t0) * pp;
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const int t = e / pp, idx = e - t * pp, i = idx / p, j = idx - i * p;
    vbase[e] = Gs[t * LD + i * PW + j];
  }
}

// The same pass with a thread per bin and a workgroup per (slot, run of NTH consecutive bins): the workgroup walks the columns of its slot's
// slab together, so column b of latent k is ONE contiguous run of NTH doubles (2 KB at NTH = 256) instead of the 512-byte pieces eight
// workgroups of mix_vsm_split_kernel pick out of it at different times - HBM row locality is what that kernel runs at 2.2 TB/s on.  A bin
// belongs to one thread: its symmetric G_t (55 values at 10 latents) and the 55 pair sums stay in that thread's registers for the whole
// pass (round 4, second form: G_t in an LDS column of its thread), nothing is reduced across waves.  One wave per SIMD (the register budget of a 256-thread workgroup is 512
// per lane), U columns' loads in flight per trip.  Measured at config 3: 7.9 ms instead of 9.8 ms per E-step at the bench's ranks (3.9 TB/s
// of its algorithmic bytes); 4 columns per trip, 128- or 512-bin workgroups (the latter spills) and a software pipeline change nothing.  Entries of Yt left of a latent's first column tile were never written: their (uniform)
// addresses are clamped to the first written column and the values masked.
// grid = (ceil(T / NTH), nslots), block = NTH, p <= PW <= 10.
template <int PW, int NTH, int U>
__global__ __launch_bounds__(NTH, 1) void mix_slot_kernel(const double* __restrict__ Yt, long long sY, int ldy, float* __restrict__ D, long long sD, int ldd,
                                                          const double* __restrict__ G, long long sG, int T, int p, int ract, double eps,
                                                          double* __restrict__ vsm, const int* __restrict__ slots,
                                                          const int* __restrict__ trial_of_slot, const int* __restrict__ roff, int col_tile, int ts) {
  constexpr int NPAIR = PW * (PW + 1) / 2;
  // G_t of this thread's bin lives in LDS, one column per thread ([pair][thread]: conflict-free), the pair sums in registers: with both in
  // registers the compiler parks one of them in AGPRs and pays a v_accvgpr_read for every use - PMC: as many of those as FMAs, vector
  // instructions 49 % of the kernel's time at one wave per SIMD
  extern __shared__ double mix_slot_g[];
  const int pp = p * p;
  const int slot = slots[blockIdx.y];
  const int t = blockIdx.x * NTH + threadIdx.x;
  const bool live = t < T;
  const int tc = live ? t : T - 1;
  const double* gsrc = G + (size_t)slot * sG + (size_t)tc * pp;
  double* gs = mix_slot_g + threadIdx.x;
  double acc[NPAIR];
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) {
      gs[(a * (a + 1) / 2 + c2) * NTH] = gsrc[(a < p ? a : 0) * p + (a < p ? c2 : 0)];
      acc[a * (a + 1) / 2 + c2] = 0.0;
    }
  int c0[PW], cmax = 0;
#pragma unroll
  for (int k = 0; k < PW; ++k) {
    c0[k] = (roff && k < p) ? (roff[k] / col_tile) * col_tile : 0;
    cmax = c0[k] > cmax ? c0[k] : cmax;
  }
  const double* y = Yt + (size_t)slot * sY + tc;
  float* d = D + (size_t)slot * sD + tc;
  for (int b0 = 0; b0 < ract; b0 += U) {
    double v[U][PW], m[U][PW];
    // entries of Yt left of a latent's first column were never written: their (uniform) addresses are clamped to the first written column
    // and the values masked - only while the trip touches such columns at all
    const bool edge = b0 < cmax;
    asm volatile("" ::: "memory");                      // (the LDS reads of G_t belong inside the trip: hoisted, they are the registers this form exists to free)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = (b0 + u < ract) ? b0 + u : ract - 1;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const int kc = k < p ? k : 0;
        const int bc = (edge && b < c0[kc]) ? c0[kc] : b;
        v[u][k] = y[(size_t)bc * ldy + (size_t)kc * ts];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        if (k >= p || (edge && b0 + u < c0[k < p ? k : 0])) v[u][k] = 0.0;
        m[u][k] = 0.0;
      }
#pragma unroll
    for (int a = 0; a < PW; ++a)
#pragma unroll
      for (int c2 = 0; c2 <= a; ++c2) {
        const double gg = gs[(a * (a + 1) / 2 + c2) * NTH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          m[u][a] += gg * v[u][c2];
          if (c2 != a) m[u][c2] += gg * v[u][a];
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u;
      if (b < ract) {
#pragma unroll
        for (int k = 0; k < PW; ++k)
          if (k < p && live) d[(size_t)b * ldd + (size_t)k * ts] = (float)(v[u][k] - m[u][k]);
#pragma unroll
        for (int a = 0; a < PW; ++a)
#pragma unroll
          for (int c2 = 0; c2 <= a; ++c2) acc[a * (a + 1) / 2 + c2] += m[u][a] * m[u][c2];
      }
    }
  }
  if (!live) return;
  double* vdst = vsm + ((size_t)trial_of_slot[slot] * T + t) * pp;
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) {
      if (a < p) {
        const double val = eps * gs[(a * (a + 1) / 2 + c2) * NTH] + acc[a * (a + 1) / 2 + c2];
        vdst[a * p + c2] = val;
        vdst[c2 * p + a] = val;
      }
    }
}
inline size_t mix_slot_lds(int pw, int nth) { return (size_t)(pw * (pw + 1) / 2) * nth * sizeof(double); }

// The same pass with the columns of the NEXT trip requested before this trip's arithmetic (two register sets), for p == PW and ract a multiple of 2 U:
// the kernel runs one wave per SIMD (its G_t column fills LDS), so nothing else hides a memory round trip - the first form spent one per trip, 2.5 us for
// 0.5 us of arithmetic.  No branch in the loop: loads at clamped addresses, masks as selects, and a thread past T repeats the arithmetic of bin T - 1 and
// stores the same numbers to the same place (so that the stores stand under no condition either: the compiler counts loads AND stores in flight).
template <int PW, int NTH, int U>
__global__ __launch_bounds__(NTH, 1) void mix_slot2_kernel(const double* __restrict__ Yt, long long sY, int ldy, float* __restrict__ D, long long sD, int ldd,
                                                           const double* __restrict__ G, long long sG, int T, int ract, double eps,
                                                           double* __restrict__ vsm, const int* __restrict__ slots,
                                                           const int* __restrict__ trial_of_slot, const int* __restrict__ roff, int col_tile, int ts) {
  constexpr int NPAIR = PW * (PW + 1) / 2, p = PW, pp = PW * PW;
  extern __shared__ double mix_slot_g[];
  const int slot = slots[blockIdx.y];
  const int t = blockIdx.x * NTH + threadIdx.x;
  const int tc = t < T ? t : T - 1;
  const double* gsrc = G + (size_t)slot * sG + (size_t)tc * pp;
  double* gs = mix_slot_g + threadIdx.x;
  double acc[NPAIR];
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) {
      gs[(a * (a + 1) / 2 + c2) * NTH] = gsrc[a * p + c2];
      acc[a * (a + 1) / 2 + c2] = 0.0;
    }
  int c0[PW], cmax = 0;
#pragma unroll
  for (int k = 0; k < PW; ++k) {
    c0[k] = roff ? (roff[k] / col_tile) * col_tile : 0;
    cmax = c0[k] > cmax ? c0[k] : cmax;
  }
  const double* y = Yt + (size_t)slot * sY + tc;
  float* d = D + (size_t)slot * sD + tc;
  auto fetch = [&](int b0, double (&v)[U][PW]) {
    const int bb = b0 < ract ? b0 : ract - U;                 // (past the end: the last trip again)
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const int b = bb + u;
        const int bc = b < c0[k] ? c0[k] : b;                  // (left of a latent's first column nothing was written: read that column, mask below)
        v[u][k] = y[(size_t)bc * ldy + (size_t)k * ts];
      }
  };
  auto process = [&](int b0, double (&v)[U][PW]) {
    double m[U][PW];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        v[u][k] = (b0 < cmax && b0 + u < c0[k]) ? 0.0 : v[u][k];
        m[u][k] = 0.0;
      }
#pragma unroll
    for (int a = 0; a < PW; ++a)
#pragma unroll
      for (int c2 = 0; c2 <= a; ++c2) {
        const double gg = gs[(a * (a + 1) / 2 + c2) * NTH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          m[u][a] += gg * v[u][c2];
          if (c2 != a) m[u][c2] += gg * v[u][a];
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < PW; ++k) d[(size_t)(b0 + u) * ldd + (size_t)k * ts] = (float)(v[u][k] - m[u][k]);
#pragma unroll
      for (int a = 0; a < PW; ++a)
#pragma unroll
        for (int c2 = 0; c2 <= a; ++c2) acc[a * (a + 1) / 2 + c2] += m[u][a] * m[u][c2];
    }
  };
  double vA[U][PW], vB[U][PW];
  fetch(0, vA);
  for (int b0 = 0; b0 < ract; b0 += 2 * U) {
    fetch(b0 + U, vB);
    asm volatile("" ::: "memory");                      // (the LDS reads of G_t stay inside the trip, behind the requests of the next one)
    process(b0, vA);
    fetch(b0 + 2 * U, vA);
    asm volatile("" ::: "memory");
    process(b0 + U, vB);
  }
  if (t >= T) return;
  double* vdst = vsm + ((size_t)trial_of_slot[slot] * T + t) * pp;
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) {
      const double val = eps * gs[(a * (a + 1) / 2 + c2) * NTH] + acc[a * (a + 1) / 2 + c2];
      vdst[a * p + c2] = val;
      vdst[c2 * p + a] = val;
    }
}

// A buffer descriptor over [ptr, ptr + bytes) built from wave-uniform values only (the halves of the pointer go through readfirstlane so that the
// compiler keeps the four words in scalar registers): raw buffer loads and stores through it take "descriptor + scalar byte offset + 32-bit lane
// byte offset" and need no 64-bit address per access in vector registers.
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_uniform_rsrc(T* ptr, size_t bytes) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)(bytes < 0xfffffff0ull ? bytes : 0xfffffff0ull));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)n, 0x00020000);
}
__device__ __forceinline__ double rsrc_load_f64(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_bytes, (int)uniform_bytes, 0);
  return __hiloint2double((int)w.y, (int)w.x);
}
__device__ __forceinline__ void rsrc_store_f32(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes, float x) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, (int)lane_bytes, (int)uniform_bytes, 0);
}

// Round 5: the same pass with TWO waves per SIMD.  mix_slot2_kernel keeps one bin per thread and 256 bins per workgroup: its G_t image fills 112 KB of
// LDS, one workgroup = one wave per SIMD, and with 35 % of the SIMD's time in vector instructions (PMC) and the rest waiting for the loads of the next
// trip nothing overlaps.  Here a workgroup is NB = 128 bins x 2 column halves: both halves read the
// SAME G_t column of their bin from LDS (56 KB: two workgroups per CU, eight waves), each keeps its own pair sums, and the halves meet once at the end
// through the LDS image.  Thread (bin, h) walks the trips b0 = (2 i + h) U.  Same arithmetic per column as mix_slot2_kernel; post_vsm sums its columns in a different order (two partial sums per bin).
// grid = (ceil(T / NB), nslots), block = 2 NB, dynamic LDS = mix_slot_lds(PW, NB); p == PW, ract a multiple of 2 U.
template <int PW, int NB, int U>
__global__ __launch_bounds__(2 * NB, 2) void mix_slot3_kernel(const double* __restrict__ Yt, long long sY, int ldy, float* __restrict__ D, long long sD, int ldd,
                                                              const double* __restrict__ G, long long sG, int T, int ract, double eps,
                                                              double* __restrict__ vsm, const int* __restrict__ slots,
                                                              const int* __restrict__ trial_of_slot, const int* __restrict__ roff, int col_tile, int ts) {
  constexpr int NPAIR = PW * (PW + 1) / 2, p = PW, pp = PW * PW;
  extern __shared__ double mix_slot_g[];
  const int slot = slots[blockIdx.y];
  const int bin = threadIdx.x % NB;
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NB);   // (wave-uniform: the trip index stays in scalar registers)
  const int t = blockIdx.x * NB + bin;
  const int tc = t < T ? t : T - 1;
  const double* gsrc = G + (size_t)slot * sG + (size_t)tc * pp;
  double* gs = mix_slot_g + bin;
  double acc[NPAIR];
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) {
      if (half == 0) gs[(a * (a + 1) / 2 + c2) * NB] = gsrc[a * p + c2];
      acc[a * (a + 1) / 2 + c2] = 0.0;
    }
  __syncthreads();
  int c0[PW], cmax = 0;
#pragma unroll
  for (int k = 0; k < PW; ++k) {
    c0[k] = roff ? (roff[k] / col_tile) * col_tile : 0;
    cmax = c0[k] > cmax ? c0[k] : cmax;
  }
  // (wave-uniform descriptors over the slot's two images; the lane's bin is the 32-bit offset of every access, the column a scalar one)
  const __amdgpu_buffer_rsrc_t y = wave_uniform_rsrc(Yt + (size_t)slot * sY, (size_t)sY * sizeof(double));
  const __amdgpu_buffer_rsrc_t d = wave_uniform_rsrc(D + (size_t)slot * sD, (size_t)ldd * ract * sizeof(float));   // (ract columns of ldd floats)
  const unsigned voff = (unsigned)tc;
  auto fetch = [&](int b0, double (&v)[U][PW]) {
    const int bb = b0 < ract ? b0 : ract - U;                 // (past the end: the last trip again)
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const int b = bb + u;
        const int bc = b < c0[k] ? c0[k] : b;                  // (left of a latent's first column nothing was written: read that column, mask below)
        v[u][k] = rsrc_load_f64(y, voff * 8u, (unsigned)(bc * ldy + k * ts) * 8u);
      }
  };
  // Two sets of column registers, each requested a whole trip ahead: a set's columns are dead once D is stored, its next loads go out there and the
  // pair sums (2 x 55 products) and the whole trip of the other set run under them.  Four waves of a workgroup x two workgroups keep 2 x 20 loads of
  // 512 B each in flight per wave.
  auto trip = [&](int b0, double (&v)[U][PW], int bnext) {
    double m[U][PW];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        v[u][k] = (b0 < cmax && b0 + u < c0[k]) ? 0.0 : v[u][k];
        m[u][k] = 0.0;
      }
#pragma unroll
    for (int a = 0; a < PW; ++a)
#pragma unroll
      for (int c2 = 0; c2 <= a; ++c2) {
        const double gg = gs[(a * (a + 1) / 2 + c2) * NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          m[u][a] += gg * v[u][c2];
          if (c2 != a) m[u][c2] += gg * v[u][a];
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) rsrc_store_f32(d, voff * 4u, (unsigned)((b0 + u) * ldd + k * ts) * 4u, (float)(v[u][k] - m[u][k]));
    asm volatile("" ::: "memory");                      // (the LDS reads of G_t stay inside the trip)
    if (bnext < ract) fetch(bnext, v);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < PW; ++a)
#pragma unroll
        for (int c2 = 0; c2 <= a; ++c2) acc[a * (a + 1) / 2 + c2] += m[u][a] * m[u][c2];
  };
  double vA[U][PW], vB[U][PW];
  const int first = half * U;
  if (first < ract) fetch(first, vA);
  if (first + 2 * U < ract) fetch(first + 2 * U, vB);
  for (int b0 = first; b0 < ract; b0 += 4 * U) {
    trip(b0, vA, b0 + 4 * U);
    if (b0 + 2 * U < ract) trip(b0 + 2 * U, vB, b0 + 6 * U);
  }
  // the halves meet: half 0 turns its sums into eps G_t + sums while the image still holds G_t, half 1 then hands its sums over through the image
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) acc[i] += eps * gs[i * NB];
  }
  __syncthreads();
  if (half == 1) {
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) gs[i * NB] = acc[i];
  }
  __syncthreads();
  if (half != 0 || t >= T) return;
  double* vdst = vsm + ((size_t)trial_of_slot[slot] * T + t) * pp;
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int c2 = 0; c2 <= a; ++c2) {
      const double val = acc[a * (a + 1) / 2 + c2] + gs[(a * (a + 1) / 2 + c2) * NB];
      vdst[a * p + c2] = val;
      vdst[c2 * p + a] = val;
    }
}

// part[(k * ngroups + g)][T x T] (column-major, ld = T, lower 64 x 64 wave tiles) = sum over the slots of group g, over columns b < ract,
// of D_k[:, b] D_k[:, b]^T, with D_k[t, b] = D[slot][(k ts + t) + b ldd] (single precision), evaluated on the FP16 matrix cores as
// described at the top of this file.  Group g holds slots [g sps, min((g + 1) sps, nslots)).
// Workgroup = one 128 x 128 tile (ti >= tj) of one (latent, group): 4 waves of 64 x 64 = 4 x 4 accumulator tiles of
// v_mfma_f32_16x16x32_f16.  Per 32-column step the two 128 x 32 operand tiles are read coalesced over t (lanes), split into hi / lo
// halves and stored in LDS as [row][32 k] (chunks of 8 halves XOR-swizzled by the row: conflict-free 16-byte fragment reads, see lidx below);
// the next step's global loads are in flight while the current one is multiplied.
// grid = (tiles * ngroups * p) in the XCD-aware order of the GEMM (8 consecutive ids = 8 batch entries, one per XCD), block = 256.
struct SyrkF16Args {
  const float* D; long long sD; int ldd, ts;
  double* part;
  int T, p, ract, nslots, sps, ngroups, tiles, ntiles;
  int dbg;                                   // timing experiments (option syrk_dbg; results wrong when set): 1 no conversions, 2 no products, 4 no global loads (slow form only), 8 no barriers
};

inline __global__ __launch_bounds__(256, 2) void syrk_f16x2_kernel(SyrkF16Args a) {
  constexpr int BT = 128, KS = 32, LS = 32;
  // LDS image [row][4 chunks of 8 halves], the chunk index XOR-swizzled with f((row >> 2) & 3), f = (0, 3, 2, 1): the 16-byte fragment reads are
  // served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) - for each of them the 16 lanes then cover
  // all 64 banks once (a padded row stride of 40 halves was conflict-free for 16 CONSECUTIVE lanes only: 2-way in the real groups), and the
  // staging stores (8 consecutive lanes = rows 4 apart, one chunk) are 2-way instead of 4-way.  No padding: 32 KB per workgroup.
  auto lidx = [](int row, int chunk) { return row * LS + ((chunk ^ ((4 - ((row >> 2) & 3)) & 3)) << 3); };
  __shared__ __attribute__((aligned(16))) _Float16 Ah[2][BT * LS], Al[2][BT * LS], Bh[2][BT * LS], Bl[2][BT * LS];     // two stages: 64 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // block id -> (batch entry, tile): entries grouped by 8 so that the blocks resident on one XCD share a (latent, group) pair
  const int nbatch = a.ngroups * a.p;
  int b, tile;
  {
    const int bid = blockIdx.x;
    const int full = nbatch >> 3, per_group = a.ntiles << 3, grp = bid / per_group;
    if (grp < full) { const int r = bid - grp * per_group; tile = r >> 3; b = (grp << 3) + (r & 7); }
    else { const int m = nbatch - (full << 3); const int r = bid - full * per_group; tile = r / m; b = (full << 3) + (r - tile * m); }
  }
  int tj = 0, rem = tile;
  while (rem >= a.tiles - tj) { rem -= a.tiles - tj; ++tj; }
  const int ti = tj + rem;
  const int k = b / a.ngroups, g = b - k * a.ngroups;
  const int i0 = ti * BT, j0 = tj * BT;
  const bool diag = (ti == tj);
  const int s_begin = g * a.sps, s_end = min(a.nslots, s_begin + a.sps);
  const bool wave_live = (i0 + wm * 64 < a.T) && (j0 + wn * 64 < a.T) && !(i0 + wm * 64 + 63 < j0 + wn * 64);

  float4v_t acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = float4v_t{0.f, 0.f, 0.f, 0.f};

  // staging: threads 0..127 take the A tile, 128..255 the B tile (idle on diagonal tiles, where B is A); a thread owns four
  // consecutive rows (t: contiguous in memory, one 16-byte load per column) and eight of the step's 32 columns
  const bool is_b = tid >= 128;
  const int rq = tid & 31, co = (tid >> 5) & 3;
  const int row0 = (is_b ? j0 : i0) + 4 * rq;
  const bool stage = !(is_b && diag);
  // 16-byte loads need rows on 4-float boundaries inside the slab (ts, ldd multiples of 4); rows at or past T are read too when the
  // latent stride leaves room for them (ts >= round_up(T, 4)): those rows only reach outputs that are never stored
  // (uniform over the workgroup: the two forms of the loop are separate code paths with barriers inside)
  const bool vec = ((a.ts & 3) == 0) && ((a.ldd & 3) == 0) && (i0 + BT <= a.ts) && (j0 + BT <= a.ts) && ((((size_t)a.D) & 15) == 0) && ((a.sD & 3) == 0);
  // Two staging register sets and two LDS stages: while step s is multiplied out of stage s & 1, step s + 1 waits in one register set (it is
  // split into halves and stored into the other stage after the matrix instructions have been issued - the conversion's vector work then runs
  // under them) and step s + 2 is being loaded into the other set: one barrier per step, the loads two steps ahead.  Every load is unconditional -
  // threads without a tile to stage (the B side of diagonal tiles) read one fixed line, columns past ract a clamped one - so that the compiler counts
  // the loads in flight instead of waiting for all of them after a branch.  (Round 3's form: one stage, two barriers per step, loads one step
  // ahead: matrix cores 30 %, vector ALU 37 %, LDS 28 % of the time - they added up.)
  float4v_t vA[8], vB[8];
  const int steps_per_slot = (a.ract + KS - 1) / KS;
  const int nsteps = (s_end - s_begin) * steps_per_slot;
  auto load = [&](auto vecc, int step, float4v_t (&v4)[8]) {
    constexpr bool VEC = decltype(vecc)::value;
    const int st = min(step, nsteps - 1);
    const int s = s_begin + st / steps_per_slot;
    const int c0 = (st % steps_per_slot) * KS + co * 8;
    const float* base = stage ? a.D + (size_t)s * a.sD + (size_t)k * a.ts : a.D;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const bool in = stage && c < a.ract;
      const size_t off = (size_t)(in ? c : 0) * a.ldd;
      float4v_t x;
      if (a.dbg & 4) {
        x = float4v_t{1.f, 2.f, 3.f, 4.f};
      } else if constexpr (VEC) {
        x = *reinterpret_cast<const float4v_t*>(base + off + (stage ? row0 : 0));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = base[off + (stage ? min(row0 + r, a.T - 1) : 0)];
      }
      v4[j] = in ? x : float4v_t{0.f, 0.f, 0.f, 0.f};
    }
  };
  // The same loads with nothing but scalar arithmetic per step (round 5; the rows, strides and base must allow 16-byte loads: `vec`).  Timing builds
  // of this kernel showed where its time went: with loads, conversions and products all switched off it still took 4.3 of 6.8 ms - per step two
  // integer divisions and eight 64-bit address chains per thread.  Here the step is a cursor of two scalars (slot, 32-column step of the slot), a
  // slot is a buffer descriptor over its ract columns, a load is descriptor + scalar byte offset (column) + the thread's constant 32-bit offset (its
  // column group and rows), and columns at or past ract fall outside the descriptor: the hardware returns zeros for them - no masks.
  int cur_s = s_begin, cur_c = 0;                             // load cursor: slot, column step (wave-uniform; advanced by every load_fast)
  const unsigned voff = (unsigned)(stage ? (co * 8) * a.ldd + row0 : 0) * 4u;
  auto load_fast = [&](float4v_t (&v4)[8]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = wave_uniform_rsrc(a.D + (size_t)cur_s * a.sD, (size_t)a.ract * a.ldd * sizeof(float));
    const unsigned soff = (unsigned)(k * a.ts + cur_c * KS * a.ldd) * 4u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)(soff + (unsigned)(j * a.ldd) * 4u), 0);
      v4[j] = float4v_t{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
    }
    if (cur_c + 1 < steps_per_slot) ++cur_c;
    else if (cur_s + 1 < s_end) { ++cur_s; cur_c = 0; }       // (past the last step: the last one again, its values are not used)
  };
  auto store = [&](int buf, const float4v_t (&v4)[8]) {
    if (!stage) return;
    _Float16* Hh = is_b ? Bh[buf] : Ah[buf];
    _Float16* Hl = is_b ? Bl[buf] : Al[buf];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      half8_t h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = v4[j][r] * SPLIT_SCALE;
        const _Float16 hx = (a.dbg & 1) ? (_Float16)0 : (_Float16)x;
        h[j] = hx;
        l[j] = (a.dbg & 1) ? (_Float16)0 : (_Float16)(x - (float)hx);
      }
      *reinterpret_cast<half8_t*>(&Hh[lidx(4 * rq + r, co)]) = h;
      *reinterpret_cast<half8_t*>(&Hl[lidx(4 * rq + r, co)]) = l;
    }
  };
  const int l15 = lane & 15, l4 = lane >> 4;
  auto multiply = [&](int buf) {
    if (!wave_live || (a.dbg & 2)) return;
    const _Float16* Bhs = diag ? Ah[buf] : Bh[buf];
    const _Float16* Bls = diag ? Al[buf] : Bl[buf];
    half8_t ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int r = lidx(wm * 64 + mi * 16 + l15, l4);
      ah[mi] = *reinterpret_cast<const half8_t*>(&Ah[buf][r]);
      al[mi] = *reinterpret_cast<const half8_t*>(&Al[buf][r]);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int r = lidx(wn * 64 + ni * 16 + l15, l4);
      bh[ni] = *reinterpret_cast<const half8_t*>(&Bhs[r]);
      bl[ni] = *reinterpret_cast<const half8_t*>(&Bls[r]);
    }
    // issued as (B fragment) x (A fragment): the lane index then runs along i, the contiguous index of the column-major output
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], ah[mi], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], al[mi], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ni], ah[mi], acc[mi][ni], 0, 0, 0);
      }
  };
  auto run = [&](auto vecc) {
    if (nsteps <= 0) return;
    constexpr bool FAST = decltype(vecc)::value;
    auto ld = [&](int step, float4v_t (&v4)[8]) { if constexpr (FAST) load_fast(v4); else load(vecc, step, v4); };
    ld(0, vA);
    ld(1, vB);
    store(0, vA);
    if (!(a.dbg & 8)) __syncthreads();
    for (int step = 0; step < nsteps; step += 2) {
      ld(step + 2, vA);
      multiply(0);
      if (step + 1 < nsteps) store(1, vB);
      if (!(a.dbg & 8)) __syncthreads();
      if (step + 1 >= nsteps) break;
      ld(step + 3, vB);
      multiply(1);
      if (step + 2 < nsteps) store(0, vA);
      if (!(a.dbg & 8)) __syncthreads();
    }
  };
  if (vec) run(std::true_type{}); else run(std::false_type{});
  if (!wave_live) return;
  // accumulator register r of a lane: output row (of the issued product) 4 l4 + r  <->  j, column l15  <->  i
  double* C = a.part + (size_t)(k * a.ngroups + g) * a.T * a.T;
  const double inv = 1.0 / ((double)SPLIT_SCALE * (double)SPLIT_SCALE);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int i = i0 + wm * 64 + mi * 16 + l15;
    if (i >= a.T) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + wn * 64 + ni * 16 + 4 * l4 + r;
        if (j >= a.T) continue;
        C[(size_t)j * a.T + i] = (double)acc[mi][ni][r] * inv;
      }
  }
}

// The same sums on 256 x 256 workgroup tiles (round 5).  With its loads freed of address arithmetic the 128-tile kernel above is bound by the stream
// of D: a (latent, group) panel has T / 128 row blocks and every tile reads two of them - 16 block reads for 4 blocks at T = 500, of which the
// XCD's L2 catches a quarter (PMC: 3.0 x the operand bytes at 5.8 TB/s).  Tiles of 256 read 4 blocks of 256 rows for 2.  Eight waves of 64 x 128
// (4 x 8 accumulator tiles = 128 registers), ONE staging register set (the loads of step s + 1 in flight while step s is multiplied), B fragments
// in two halves of four; two LDS stages of 64 KB.  Fast path only: rows, strides and base allow 16-byte loads (the host checks, syrk_f16x2_kernel
// otherwise).  grid = tiles256 (tiles256 + 1) / 2 * ngroups * p in the same XCD-aware order, block = 512, dynamic LDS = 128 KB.
inline __global__ __launch_bounds__(512, 1) void syrk256_f16x2_kernel(SyrkF16Args a) {
  constexpr int BT = 256, KS = 32, LS = 32;
  auto lidx = [](int row, int chunk) { return row * LS + ((chunk ^ ((4 - ((row >> 2) & 3)) & 3)) << 3); };
  extern __shared__ __attribute__((aligned(16))) _Float16 syrk256_lds[];
  _Float16* Ah = syrk256_lds;                       // [2][BT * LS] each
  _Float16* Al = Ah + 2 * BT * LS;
  _Float16* Bh = Al + 2 * BT * LS;
  _Float16* Bl = Bh + 2 * BT * LS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;               // rows 64 wm .., columns 128 wn ..
  const int nbatch = a.ngroups * a.p;
  int b, tile;
  {
    const int bid = blockIdx.x;
    const int full = nbatch >> 3, per_group = a.ntiles << 3, grp = bid / per_group;
    if (grp < full) { const int r = bid - grp * per_group; tile = r >> 3; b = (grp << 3) + (r & 7); }
    else { const int m = nbatch - (full << 3); const int r = bid - full * per_group; tile = r / m; b = (full << 3) + (r - tile * m); }
  }
  int tj = 0, rem = tile;
  while (rem >= a.tiles - tj) { rem -= a.tiles - tj; ++tj; }
  const int ti = tj + rem;
  const int k = b / a.ngroups, g = b - k * a.ngroups;
  const int i0 = ti * BT, j0 = tj * BT;
  const bool diag = (ti == tj);
  const int s_begin = g * a.sps, s_end = min(a.nslots, s_begin + a.sps);
  // a wave's 64 x 128 block is issued unless it lies outside T or strictly above the diagonal
  const bool wave_live = (i0 + wm * 64 < a.T) && (j0 + wn * 128 < a.T) && !(i0 + wm * 64 + 63 < j0 + wn * 128);

  float4v_t acc[4][8];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = float4v_t{0.f, 0.f, 0.f, 0.f};

  // staging: threads 0..255 take the A tile, 256..511 the B tile (idle on diagonal tiles); four consecutive rows and eight of the 32 columns each
  const bool is_b = tid >= 256;
  const int rq = tid & 63, co = (tid >> 6) & 3;
  const int row0 = (is_b ? j0 : i0) + 4 * rq;
  const bool stage = !(is_b && diag);
  const int steps_per_slot = (a.ract + KS - 1) / KS;
  const int nsteps = (s_end - s_begin) * steps_per_slot;
  int cur_s = s_begin, cur_c = 0;
  const unsigned voff = (unsigned)(stage ? (co * 8) * a.ldd + row0 : 0) * 4u;
  float4v_t v[8];
  auto load_fast = [&]() {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = wave_uniform_rsrc(a.D + (size_t)cur_s * a.sD, (size_t)a.ract * a.ldd * sizeof(float));
    const unsigned soff = (unsigned)(k * a.ts + cur_c * KS * a.ldd) * 4u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)(soff + (unsigned)(j * a.ldd) * 4u), 0);
      v[j] = float4v_t{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
    }
    if (cur_c + 1 < steps_per_slot) ++cur_c;
    else if (cur_s + 1 < s_end) { ++cur_s; cur_c = 0; }
  };
  auto store = [&](int buf) {
    if (!stage) return;
    _Float16* Hh = (is_b ? Bh : Ah) + buf * BT * LS;
    _Float16* Hl = (is_b ? Bl : Al) + buf * BT * LS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      half8_t h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = v[j][r] * SPLIT_SCALE;
        const _Float16 hx = (_Float16)x;
        h[j] = hx;
        l[j] = (_Float16)(x - (float)hx);
      }
      *reinterpret_cast<half8_t*>(&Hh[lidx(4 * rq + r, co)]) = h;
      *reinterpret_cast<half8_t*>(&Hl[lidx(4 * rq + r, co)]) = l;
    }
  };
  const int l15 = lane & 15, l4 = lane >> 4;
  auto multiply = [&](int buf) {
    if (!wave_live) return;
    const _Float16* As_h = Ah + buf * BT * LS;
    const _Float16* As_l = Al + buf * BT * LS;
    const _Float16* Bs_h = (diag ? Ah : Bh) + buf * BT * LS;
    const _Float16* Bs_l = (diag ? Al : Bl) + buf * BT * LS;
    half8_t ah[4], al[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int r = lidx(wm * 64 + mi * 16 + l15, l4);
      ah[mi] = *reinterpret_cast<const half8_t*>(&As_h[r]);
      al[mi] = *reinterpret_cast<const half8_t*>(&As_l[r]);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      half8_t bh[4], bl[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int r = lidx(wn * 128 + half * 64 + ni * 16 + l15, l4);
        bh[ni] = *reinterpret_cast<const half8_t*>(&Bs_h[r]);
        bl[ni] = *reinterpret_cast<const half8_t*>(&Bs_l[r]);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float4v_t c = acc[mi][half * 4 + ni];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], ah[mi], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], al[mi], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ni], ah[mi], c, 0, 0, 0);
          acc[mi][half * 4 + ni] = c;
        }
    }
  };
  if (nsteps > 0) {
    load_fast();
    store(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
      if (step + 1 < nsteps) load_fast();
      multiply(step & 1);
      if (step + 1 < nsteps) store((step + 1) & 1);
      __syncthreads();
    }
  }
  if (!wave_live) return;
  double* C = a.part + (size_t)(k * a.ngroups + g) * a.T * a.T;
  const double inv = 1.0 / ((double)SPLIT_SCALE * (double)SPLIT_SCALE);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int i = i0 + wm * 64 + mi * 16 + l15;
    if (i >= a.T) continue;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + wn * 128 + ni * 16 + 4 * l4 + r;
        if (j >= a.T) continue;
        C[(size_t)j * a.T + i] = (double)acc[mi][ni][r] * inv;
      }
  }
}

// Cross term of the split form for one latent:  X[g] (rk x T, column-major, ld = rk) = sum over the slots s of group g of
//   A_s (rk x kw) D_s^T (kw x T),   A_s[i][b] = A[s sM + i + b lda]  (FP64: rows of latent k of L_s^-T right of its first column),
//                                    D_s[b][t] = D[s sD + b ldd + t]  (the single-precision correction of latent k, its first kw columns used).
// The general GEMM kernel walks this product in 64-row tiles; a latent's rank is a multiple of 16 between 16 and 128 here, so its
// second row tile is mostly padding and the waves assigned to it idle (PMC: matrix cores busy half of the time).  This kernel gives a
// workgroup ALL rk rows of a 64-column tile: wave w owns the 16 columns w of the tile and NTR = rk / 16 accumulator tiles, the k loop
// runs over the group's slots and their kw columns in chunks of 16 staged in LDS (A rows contiguous, D widened to FP64 while it is
// staged), double-buffered through registers.  Issued matrix-core work = the useful work.  grid = (ceil(T / 64), ngroups), block = 256.
struct CrossArgs {
  const double* A; long long sM; int lda;
  const void* D; long long sD; int ldd;       // the column-side operand: float (the correction D) or double (A itself: the first term's S_k = sum A A^T)
  double* C; long long sC;
  int rk, T, kw, nslots, sps;
  int row0, ldc;                                // this launch's first row of A / X and the row count of X (ranks above 128: one launch per 128 rows)
};

// TB = float: the cross term as described above.  TB = double, LOWER: the same loop with D := A (column n of the result <-> row n of A), i.e.
// S[g] = sum_s A_s A_s^T (rk x rk); the 16 x 16 tiles above the diagonal are not issued (row tile mi < the wave's column tile), their
// entries are left untouched - the consumer reads the lower 32 x 32 blocks and mirrors the rest.
template <int NTR, typename TB, bool LOWER>
__global__ __launch_bounds__(256) void cross_term_kernel(CrossArgs a) {
  constexpr int GK = 16;
  constexpr int RS = NTR * 16 + ((NTR & 1) ? 0 : 16);          // row stride of the A image: = 16 mod 32 (conflict-free 8-byte fragment reads)
  constexpr int NA = (GK * NTR * 16 + 255) / 256;               // staged A elements per thread
  __shared__ double As[2][GK][RS];
  __shared__ double Bs[2][GK][64 + 16];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int t0 = blockIdx.x * 64;
  const int g = blockIdx.y;
  const int s_begin = g * a.sps, s_end = min(a.nslots, s_begin + a.sps);
  const int cps = a.kw / GK;                                    // chunks per slot (kw is a multiple of 16)
  const int nchunks = (s_end - s_begin) * cps;
  const int rk = a.rk;
  const TB* Dall = reinterpret_cast<const TB*>(a.D);
  const int ct = blockIdx.x * 4 + wave;                         // this wave's column tile of 16
  const bool wave_live = t0 + wave * 16 < a.T;

  // Two register sets: the loads of chunk ch + 2 are issued while chunk ch is multiplied (one chunk is 16 matrix instructions per wave, 0.4 us - less than
  // a memory round trip; with a single set the matrix cores were busy half of the time, PMC).  All loads are unconditional from clamped addresses and
  // masked when they are stored: a load under a condition is waited for on its own.
  double raA[NA], raB[NA];
  TB rbA[4], rbB[4];
  // (the chunks are loaded in order: the chunk is a cursor of two scalars - slot, chunk of the slot - advanced by every call, and a thread's element
  //  offsets inside a chunk are computed once: no division and no index decoding per chunk)
  int cur_s = s_begin, cur_c = 0;
  size_t aoff[NA], doff[4];
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    const int e = tid + 256 * u;
    const int kk = min(e / (NTR * 16), GK - 1), i = min(e % (NTR * 16), rk - 1);
    aoff[u] = (size_t)kk * a.lda + i;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    const int kk = e >> 6, t = min(t0 + (e & 63), a.T - 1) - t0;
    doff[u] = (size_t)kk * a.ldd + t;
  }
  auto load = [&](int, double (&ra)[NA], TB (&rb)[4]) {
    const int c0 = cur_c * GK;
    const double* Ap = a.A + (size_t)cur_s * a.sM + (size_t)c0 * a.lda + a.row0;
    const TB* Dp = Dall + (size_t)cur_s * a.sD + (size_t)c0 * a.ldd + t0;
#pragma unroll
    for (int u = 0; u < NA; ++u) ra[u] = Ap[aoff[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) rb[u] = Dp[doff[u]];
    if (cur_c + 1 < cps) ++cur_c;
    else if (cur_s + 1 < s_end) { ++cur_s; cur_c = 0; }       // (past the last chunk: the last one again, its values are not used)
  };
  auto store = [&](int buf, const double (&ra)[NA], const TB (&rb)[4]) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + 256 * u;
      const int kk = e / (NTR * 16), i = e - kk * (NTR * 16);
      if (kk < GK) As[buf][kk][i] = (i < rk) ? ra[u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      Bs[buf][e >> 6][e & 63] = (t0 + (e & 63) < a.T) ? (double)rb[u] : 0.0;
    }
  };
  mdouble4 acc[NTR];
#pragma unroll
  for (int mi = 0; mi < NTR; ++mi) acc[mi] = mdouble4{0.0, 0.0, 0.0, 0.0};
  auto multiply = [&](int buf) {
    if (!wave_live) return;
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      const double bf = Bs[buf][kk + l4][wave * 16 + l15];
#pragma unroll
      for (int mi = 0; mi < NTR; ++mi)
        if (!LOWER || mi >= ct) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf, As[buf][kk + l4][mi * 16 + l15], acc[mi], 0, 0, 0);
    }
  };
  if (nchunks > 0) {
    load(0, raA, rbA);
    load(1, raB, rbB);
    store(0, raA, rbA);
  }
  __syncthreads();
  // chunk ch sits in LDS buffer ch & 1; its successor's registers are set B for even ch, set A for odd ch
  for (int ch = 0; ch < nchunks; ch += 2) {
    load(ch + 2, raA, rbA);
    multiply(0);
    if (ch + 1 < nchunks) store(1, raB, rbB);
    __syncthreads();
    if (ch + 1 >= nchunks) break;
    load(ch + 3, raB, rbB);
    multiply(1);
    if (ch + 2 < nchunks) store(0, raA, rbA);
    __syncthreads();
  }
  // issued as (column-side fragment) x (A fragment): result row l4 + 4 r  <->  column t, result column l15  <->  row i
  if (!wave_live) return;
  double* C = a.C + (size_t)g * a.sC + a.row0;
#pragma unroll
  for (int mi = 0; mi < NTR; ++mi) {
    const int i = mi * 16 + l15;
    if (i >= rk || (LOWER && mi < ct)) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + wave * 16 + l4 + 4 * r;
      if (t < a.T) C[(size_t)t * a.ldc + i] = acc[mi][r];
    }
  }
}

template <typename TB, bool LOWER>
inline void cross_term_launch(const CrossArgs& ca, dim3 grid, hipStream_t st) {
  switch (ca.rk / 16) {
    case 1: hipLaunchKernelGGL((cross_term_kernel<1, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
    case 2: hipLaunchKernelGGL((cross_term_kernel<2, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
    case 3: hipLaunchKernelGGL((cross_term_kernel<3, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
    case 4: hipLaunchKernelGGL((cross_term_kernel<4, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
    case 5: hipLaunchKernelGGL((cross_term_kernel<5, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
    case 6: hipLaunchKernelGGL((cross_term_kernel<6, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
    case 7: hipLaunchKernelGGL((cross_term_kernel<7, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
    default: hipLaunchKernelGGL((cross_term_kernel<8, TB, LOWER>), grid, dim3(256), 0, st, ca); break;
  }
}

// out[M x N] (ld = M) = sum over g < ngroups of part[g][M x N]; with lower > 0 (M == N) the parts hold only the wave tiles
// (i / lower) >= (j / lower) of a symmetric matrix (GEMM_LOWER on 64 x 64 workgroup tiles skips 32 x 32 wave tiles) and the rest is
// mirrored.  grid = ceil(M N / 64), block = 256: 64 entries x 4 group lanes - lane q of an entry sums the groups g = q mod 4 (eight loads in
// flight, eight partial sums: one running sum made it a chain of memory round trips), the four meet in LDS in a fixed order.  (A thread per
// entry walking all groups: 32 workgroups and chains of 256 loads for a latent's r_k x r_k sums - 35 us per launch, twenty launches per E-step.)
inline __global__ __launch_bounds__(256) void sum_groups_kernel(const double* __restrict__ part, int ngroups, int M, int N, int lower, double* __restrict__ out) {
  __shared__ double red[4][64];
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + el;
  const bool in = e < (size_t)M * N;
  const size_t ec = in ? e : 0;
  const int i = (int)(ec % M), j = (int)(ec / M);
  const size_t src = (lower > 0 && (i / lower) < (j / lower)) ? (size_t)i * M + j : ec;
  const size_t gs = (size_t)M * N;
  double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int g = q;
  for (; g + 28 < ngroups; g += 32) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(g + 4 * u) * gs + src];
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += v[u];
  }
  for (; g < ngroups; g += 4) s8[0] += part[(size_t)g * gs + src];
  red[q][el] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  __syncthreads();
  if (q == 0 && in) out[e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// Pacc[k][T x T] (ld = Tp, full symmetric) += the split sum of latent k:
//   eps sum_slots G_t[k][k] on the diagonal  +  T1_k  -  X_k - X_k^T  +  sum_groups DD[(k, g)]
// with T1 = F_k S_k F_k^T and X = F_k [sum_r A_rk D_rk^T] given as full T x T matrices (ld = T) and DD as lower 64 x 64 wave tiles of the
// FP16 kernel.  A block owns the pair of tiles (ti, tj), ti >= tj, of latent k (every such tile lies inside the stored wave tiles): it sums the group parts of the stored tile
// (coalesced), stages the two X tiles in LDS so that X^T comes from LDS rather than from stride-T global reads, and writes both the
// tile and - from the LDS copy of the sums - its mirror image, every global access running along the contiguous index.
// (The first form - one block per column, the mirror half read at stride T - moved 9.5 GB for 1.3 GB of parts: PMC.)
// grid = (ntile (ntile + 1) / 2, p), block = 256; tiles of 32 x 32 (256-byte runs; a 64 x 64 form left one block per CU and ran 1.06 ms).
constexpr int PACC_TS = 32;
inline __global__ __launch_bounds__(256) void pacc_split_reduce_kernel(const double* __restrict__ T1, const double* __restrict__ X, const double* __restrict__ DD, int ngroups,
                                                                const double* __restrict__ G, long long sG, int nslots, double eps, int T, int Tp, int p,
                                                                double* __restrict__ Pacc) {
  constexpr int TS = PACC_TS;
  __shared__ double Xa[TS][TS + 1], Xb[TS][TS + 1], Ss[TS][TS + 1];
  __shared__ double gd[TS];
  const int k = blockIdx.y;
  int tj = 0, rem = blockIdx.x;
  const int ntile = (T + TS - 1) / TS;
  while (rem >= ntile - tj) { rem -= ntile - tj; ++tj; }
  const int ti = tj + rem;
  const int i0 = ti * TS, j0 = tj * TS;
  const size_t tt = (size_t)T * T;
  const double* t1 = T1 + (size_t)k * tt;
  const double* x = X + (size_t)k * tt;
  const double* dd = DD + (size_t)k * ngroups * tt;
  double* out = Pacc + (size_t)k * Tp * Tp;
  const int tid = threadIdx.x;
  // tiles of X: Xa[c][r] = X[i0 + r][j0 + c] (the stored orientation is column-major: element (i, j) at j T + i), Xb[c][r] = X[j0 + r][i0 + c]
  for (int e = tid; e < TS * TS; e += 256) {
    const int r = e % TS, cc = e / TS;
    Xa[cc][r] = (i0 + r < T && j0 + cc < T) ? x[(size_t)(j0 + cc) * T + i0 + r] : 0.0;
    Xb[cc][r] = (j0 + r < T && i0 + cc < T) ? x[(size_t)(i0 + cc) * T + j0 + r] : 0.0;
    double sum = 0.0;
    if (i0 + r < T && j0 + cc < T) {
      const size_t e2 = (size_t)(j0 + cc) * T + i0 + r;
      for (int g2 = 0; g2 < ngroups; ++g2) sum += dd[(size_t)g2 * tt + e2];
    }
    Ss[cc][r] = sum;
  }
  // diagonal term of a diagonal tile: eps sum over slots of G_t[k][k], t = i0 .. i0 + 63 (four threads per bin, fixed order)
  if (ti == tj) {
    constexpr int PER = 256 / TS;                                  // threads per bin (a power of two, lanes of one wave)
    const int t = tid / PER, part = tid % PER;
    double gs = 0.0;
    if (i0 + t < T)
      for (int sl = part; sl < nslots; sl += PER) gs += G[(size_t)sl * sG + (size_t)(i0 + t) * p * p + (size_t)k * p + k];
#pragma unroll
    for (int o = 1; o < PER; o <<= 1) gs += __shfl_xor(gs, o);
    if (part == 0) gd[t] = eps * gs;
  }
  __syncthreads();
  // the tile itself: (i, j) = (i0 + r, j0 + c):  T1 - X[i][j] - X[j][i] + S;  X[j][i] = Xb[r][c]
  for (int e = tid; e < TS * TS; e += 256) {
    const int r = e % TS, cc = e / TS;
    const int i = i0 + r, j = j0 + cc;
    if (i < T && j < T) {
      double v = t1[(size_t)j * T + i] - Xa[cc][r] - Xb[r][cc] + Ss[cc][r];
      if (ti == tj && r == cc) v += gd[r];
      out[(size_t)j * Tp + i] += v;
    }
  }
  // its mirror image (off-diagonal tile pairs only): (i, j) = (j0 + r, i0 + c):  T1[i][j] - X[i][j] - X[j][i] + S^T = T1 - Xb[c][r] - Xa[r][c] + Ss[r][c]
  if (ti != tj) {
    for (int e = tid; e < TS * TS; e += 256) {
      const int r = e % TS, cc = e / TS;
      const int i = j0 + r, j = i0 + cc;
      if (i < T && j < T) out[(size_t)j * Tp + i] += t1[(size_t)j * T + i] - Xb[cc][r] - Xa[r][cc] + Ss[r][cc];
    }
  }
}

}  // namespace pgpfa
