// (C,d) M-step cost / gradient pass on the FP64 matrix cores (learning.MStepObservationCost(_grad), learning.py:20-91).
//
// With theta_n = (c_n, d_n), V_t = post_vsm[t], m_t = post_mean[:,t]:
//   hh = c_n.m_t + d_n ; rho = c_n^T V_t c_n ; yhat = exp(hh + rho/2)
//   cost_n = sum_t (y hh - yhat) ; dC_n = sum_t (y - yhat) m_t - yhat V_t c_n ; dd_n = sum_t (y - yhat)
// Everything that involves the counts is LINEAR in theta and independent of it otherwise:
//   sum_t y hh = c_n . YM_n + d_n YS_n ,  YM_n = sum_t y_nt m_t ,  YS_n = sum_t y_nt     (cd_ym_kernel, once per E-step)
// so one evaluation only needs yhat, and both of its contractions are GEMMs over the (trial, bin) axis:
//   E[t][n]    = sum_c Phi[t][c] Theta[c][n]      Phi[t] = [ V_t[a][b], a >= b | m_t ] , Theta[.][n] = [ (1 or 1/2) c_na c_nb | c_n ]
//   Out[c'][n] = sum_t Phi'[c'][t] yhat[t][n]     Phi' = [ V_t pairs | m_t | 1 ]  ->  A_n = sum_t yhat V_t, sum_t yhat m_t, sum_t yhat
//   dC_n = YM_n - sum_t yhat m_t - A_n c_n ; dd_n = YS_n - sum_t yhat ; cost_n = c_n.YM_n + d_n YS_n - sum_t yhat
// v_mfma_f64_16x16x4_f64 tiles: 16 bins x 16 neurons.  The accumulator layout of E (register r, lane (l15, l4) = bin 4r + l4,
// neuron l15) IS the B-fragment layout of the second product (k = bin), so yhat never leaves registers.  A workgroup covers
// ALL neurons (one wave per 16) of a staged Phi tile, so post_vsm is read from HBM once per evaluation; Theta lives in
// registers for the whole kernel; Phi tiles are staged in LDS (row stride = 2 mod 32 doubles: the per-bin fragment reads of
// the first product are conflict-free) with the next tile's global loads in flight in registers while the current one is
// multiplied.  Per (16 bins, 16 neurons): KS + 4 NT MFMAs (p = 10: 17 + 20) against ~140 FP64 FMAs per (bin, neuron) of
// the vector form - the same flops, but operands are reused from registers instead of being broadcast from LDS per FMA.
#pragma once

namespace pgpfa {

template <int PW>
struct CdM {
  static constexpr int NP = PW * (PW + 1) / 2;          // pairs a >= b
  static constexpr int NC = NP + PW + 1;                // Phi row: pairs | mean | 1
  static constexpr int KS = (NP + PW + 3) / 4;          // k steps of the first product (d enters as accumulator init)
  static constexpr int NT = (NC + 15) / 16;             // row tiles of the second product
  static constexpr int S = ((NC - 2 + 31) / 32) * 32 + 2;   // LDS row stride (doubles), = 2 mod 32
  static constexpr int BT = (PW <= 10) ? 64 : (PW <= 12) ? 32 : 16;   // bins per staged tile
  static constexpr int MAXPF = (BT * NC + 511) / 512;   // prefetch registers per thread (blocks have 512 threads)
  static constexpr int LDS_DOUBLES = BT * S + 32;       // one stage; the kernel double-buffers
};

// YM[k][n] = sum over the listed trials and bins of y_nt m_kt (k < p), YM[p][n] = sum y_nt: partial sums per block,
// part[blockIdx.x][(p+1)][q].  grid = (nblocks), block = 256; a block walks trials, per trial bin tiles of 64: the count tile
// [q][64] (coalesced 64-byte rows) and the mean tile [p][64] are staged in LDS, thread = neuron (q <= 256 per pass).
// (Yhi: plane of the counts' high bytes, NULL when every count fits one byte - a second pass over the tile with weight 256)
inline __global__ __launch_bounds__(256) void cd_ym_kernel(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, const double* __restrict__ mean,
                                                    const int* __restrict__ trials, int ntr, int q, int p, int T, double* __restrict__ part) {
  __shared__ double ms[32][64];
  __shared__ unsigned yt[256][17];                      // 64 counts of a neuron as 16 words (+1: bank spread)
  double* out = part + (size_t)blockIdx.x * (p + 1) * q;
  for (int n0 = 0; n0 < q; n0 += 256) {
    const int n = n0 + threadIdx.x;
    const int nrow = min(256, q - n0);
    double acc[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) acc[k] = 0.0;
    for (int i = blockIdx.x; i < ntr; i += gridDim.x) {
      const size_t r = trials[i];
      for (int t0 = 0; t0 < T; t0 += 64) {
        const int tn = min(64, T - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < p * 64; e += 256) {
          const int k = e >> 6, t = e & 63;
          ms[k][t] = (t < tn) ? mean[(r * p + k) * T + t0 + t] : 0.0;
        }
        for (int plane = 0; plane < (Yhi ? 2 : 1); ++plane) {
        const uint8_t* Yp = plane ? Yhi : Y;
        const double ysc = plane ? 256.0 : 1.0;
        if (plane) __syncthreads();                    // the low plane's tile is fully consumed
        if ((T & 3) == 0) {
          // rows of counts start on 4-byte boundaries: one word (4 bins) per load
          for (int e = threadIdx.x; e < nrow * 16; e += 256) {
            const int row = e >> 4, w4 = e & 15;
            unsigned w = 0u;
            if (4 * w4 < tn) {
              w = *reinterpret_cast<const unsigned*>(Yp + (r * q + n0 + row) * T + t0 + 4 * w4);
              const int left = tn - 4 * w4;                       // bins of this word inside the trial (T % 4 == 0: 4, always)
              if (left < 4) w &= (1u << (8 * left)) - 1u;
            }
            yt[row][w4] = w;
          }
        } else {
          for (int e = threadIdx.x; e < nrow * 64; e += 256) {
            const int row = e >> 6, t = e & 63;
            const unsigned v = (t < tn) ? Yp[(r * q + n0 + row) * T + t0 + t] : 0u;
            // pack 4 counts per word: lanes t, t+1, t+2, t+3 of a quad
            unsigned w = v << (8 * (t & 3));
            w |= __shfl_xor(w, 1);
            w |= __shfl_xor(w, 2);
            if ((t & 3) == 0) yt[row][t >> 2] = w;
          }
        }
        __syncthreads();
        if (n < q) {
#pragma unroll 4
          for (int w4 = 0; w4 < 16; ++w4) {
            const unsigned w = yt[threadIdx.x][w4];
            if (w == 0u) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const double yv = ysc * (double)((w >> (8 * b)) & 255u);
              const int t = 4 * w4 + b;
#pragma unroll
              for (int k = 0; k < 32; ++k)
                if (k < p) acc[k] += yv * ms[k][t];
              acc[32] += yv;
            }
          }
        }
        }
      }
    }
    if (n < q) {
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k < p) out[(size_t)k * q + n] = acc[k];
      out[(size_t)p * q + n] = acc[32];
    }
  }
}

// The same sums on the FP64 matrix cores (round 5; p + 1 <= 16, q <= 256 per pass).  The vector form above reads one LDS value per multiply-add
// (10.25 broadcast reads per 11 products and thread): 0.74 ms per EM iteration at config 3 for 143 MB of input.  As a product
// YM[k][n] = sum_(trial, bin) M[k][bin] Y[n][bin] it is 13 neuron tiles x (bins / 4) instructions of v_mfma_f64_16x16x4: first operand the means
// (lane l15 <-> k, row p = ones for sum y, rows beyond zero), second operand the counts (lane l15 <-> neuron, one byte of a staged word), K = four
// bins per step; result lane (l15 = neuron, l4), register r = row k = l4 + 4 r: stores run along the neurons.  Staging as above (count words
// coalesced, 64 bins per tile); the mean tile is kept bin-major with stride 16 - fragment reads conflict-free.  Wave w takes neuron tiles w, w + 4, ...
// Same part layout: part[blockIdx.x][(p+1)][q].  grid = (nblocks), block = 256.
inline __global__ __launch_bounds__(256) void cd_ym_mfma_kernel(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, const double* __restrict__ mean,
                                                                const int* __restrict__ trials, int ntr, int q, int p, int T, double* __restrict__ part) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  __shared__ double ms[64][16];                         // [bin][k]
  __shared__ unsigned yt[256][17], yh[256][17];         // 64 counts of a neuron as 16 words (+1: bank spread): low and high byte planes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  double* out = part + (size_t)blockIdx.x * (p + 1) * q;
  for (int e = tid; e < 64 * 16; e += 256) ms[e >> 4][e & 15] = 0.0;       // rows k > p stay zero
  for (int n0 = 0; n0 < q; n0 += 256) {
    const int nrow = min(256, q - n0);
    const int ntile = (nrow + 15) / 16;
    v4d acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int i = blockIdx.x; i < ntr; i += gridDim.x) {
      const size_t r = trials[i];
      for (int t0 = 0; t0 < T; t0 += 64) {
        const int tn = min(64, T - t0);
        __syncthreads();
        for (int e = tid; e < (p + 1) * 64; e += 256) {
          const int k = e >> 6, t = e & 63;
          ms[t][k] = (t < tn) ? (k < p ? mean[(r * p + k) * T + t0 + t] : 1.0) : 0.0;
        }
        for (int plane = 0; plane < (Yhi ? 2 : 1); ++plane) {
          const uint8_t* Yp = plane ? Yhi : Y;
          unsigned (*dst)[17] = plane ? yh : yt;
          if ((T & 3) == 0) {
            for (int e = tid; e < nrow * 16; e += 256) {
              const int row = e >> 4, w4 = e & 15;
              dst[row][w4] = (4 * w4 < tn) ? *reinterpret_cast<const unsigned*>(Yp + (r * q + n0 + row) * T + t0 + 4 * w4) : 0u;
            }
          } else {
            for (int e = tid; e < nrow * 64; e += 256) {
              const int row = e >> 6, t = e & 63;
              const unsigned v = (t < tn) ? Yp[(r * q + n0 + row) * T + t0 + t] : 0u;
              unsigned w = v << (8 * (t & 3));
              w |= __shfl_xor(w, 1);
              w |= __shfl_xor(w, 2);
              if ((t & 3) == 0) dst[row][t >> 2] = w;
            }
          }
        }
        __syncthreads();
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
          const double mk = ms[4 * s4 + l4][l15];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int tile = wave + 4 * j;
            if (tile < ntile) {                                             // (uniform over the wave)
              const int row = tile * 16 + l15;
              unsigned cnt = (yt[row][s4] >> (8 * l4)) & 255u;
              if (Yhi) cnt += ((yh[row][s4] >> (8 * l4)) & 255u) << 8;
              acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(mk, (row < nrow) ? (double)cnt : 0.0, acc[j], 0, 0, 0);
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tile = wave + 4 * j, n = n0 + tile * 16 + l15;
      if (tile < ntile && n < q) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int k = l4 + 4 * rr;
          if (k <= p) out[(size_t)k * q + n] = acc[j][rr];
        }
      }
    }
  }
}

// sums[(p+2)][q] (rows: -(sum yhat m + A c), -sum yhat, -sum yhat as written by mstep_cd_mfma_kernel and reduced over blocks)
// += the count terms: rows k < p: YM[k][n]; row p: YS[n]; row p+1: c_n.YM_n + d_n YS_n
inline __global__ void cd_add_ym_kernel(double* __restrict__ sums, const double* __restrict__ ym, const double* __restrict__ vec, int q, int p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= q) return;
  double lin = vec[(size_t)p * q + n] * ym[(size_t)p * q + n];
  for (int k = 0; k < p; ++k) {
    const double v = ym[(size_t)k * q + n];
    sums[(size_t)k * q + n] += v;
    lin += vec[(size_t)k * q + n] * v;
  }
  sums[(size_t)p * q + n] += ym[(size_t)p * q + n];
  sums[(size_t)(p + 1) * q + n] += lin;
}

// grid = (nby, groups), block = (64, 8): the neuron tiles (16 neurons each, one per wave) are dealt to `groups`
// workgroups of 8 waves (waves without a tile only help staging), so a wave may use 256 registers and two waves share a
// SIMD's matrix pipe.  The item index is the fast grid dimension: with nby a multiple of 8 the workgroups that walk the same
// (trial, tile) items for different neuron groups get block ids nby apart, i.e. the same XCD, and meet in its L2.  Phi tiles are double-buffered in LDS: one barrier per tile.
// The last, partly filled 4-column k step of the first product and (when at most 4) the rows left over after the full
// 16-row tiles of the second product are done on the vector ALU instead of a mostly empty MFMA (p = 10: 16 + 16 MFMAs
// per 16 bins instead of 17 + 20).
template <int PW>
struct CdK {
  using M = CdM<PW>;
  static constexpr int KSM = (M::NP + PW) / 4;                       // full k steps
  static constexpr int KV = (M::NP + PW) - 4 * KSM;                   // leftover columns (vector ALU)
  static constexpr int NV0 = M::NC - 16 * (M::NC / 16);
  static constexpr int NTM = (NV0 <= 4) ? M::NC / 16 : M::NC / 16 + 1;   // row tiles on the matrix cores
  static constexpr int NV = (NV0 <= 4) ? NV0 : 0;                     // leftover rows (vector ALU)
};

template <int PW>
__global__ __launch_bounds__(512) void mstep_cd_mfma_kernel(CdArgs a, int tiles_per_group) {
  using M = CdM<PW>;
  using K = CdK<PW>;
  constexpr int NP = M::NP, NC = M::NC, S = M::S, BT = M::BT, MAXPF = M::MAXPF;
  constexpr int KSM = K::KSM, KV = K::KV, NTM = K::NTM, NV = K::NV;
  extern __shared__ double lds[];                       // Phi tile [BT][S] (+ slack)
  const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nthreads = 64 * blockDim.y, tid = wave * 64 + lane;
  const int p = a.p, q = a.q, T = a.T, pp = p * p;
  const int n0 = (blockIdx.y * tiles_per_group + wave) * 16;
  const bool active_wave = wave < tiles_per_group && n0 < q;      // other waves only help staging
  const int n = n0 + l15;
  const bool live = active_wave && n < q;
  const int nc = (n < q) ? n : q - 1;

  // Theta fragments (B operand of the first product): th[kk] = Theta[c = l4 + 4 kk][n]; thv: the leftover columns
  double th[KSM > 0 ? KSM : 1], thv[KV > 0 ? KV : 1];
  {
    double c[PW];
#pragma unroll
    for (int l = 0; l < PW; ++l) c[l] = (l < p) ? a.vec[(size_t)l * q + nc] : 0.0;
    auto theta = [&](int col) {
      double v = 0.0;
      if (col < NP) {
        int pa = 0;
        while ((pa + 1) * (pa + 2) / 2 <= col) ++pa;
        const int pb = col - pa * (pa + 1) / 2;
        double ca = 0.0, cb = 0.0;
#pragma unroll
        for (int l = 0; l < PW; ++l) { ca = (l == pa) ? c[l] : ca; cb = (l == pb) ? c[l] : cb; }
        v = (pa == pb ? 0.5 : 1.0) * ca * cb;
      } else if (col - NP < PW) {
#pragma unroll
        for (int l = 0; l < PW; ++l) v = (l == col - NP) ? c[l] : v;
      }
      return v;
    };
#pragma unroll
    for (int kk = 0; kk < KSM; ++kk) th[kk] = theta(l4 + 4 * kk);
#pragma unroll
    for (int i = 0; i < KV; ++i) thv[i] = theta(4 * KSM + i);
  }
  const double dn = a.vec[(size_t)p * q + nc];

  // per-thread staging plan: element e = tid + i * nthreads of the [BT][NC] tile -> source kind / offset, LDS offset
  int soff[MAXPF], loff[MAXPF];
#pragma unroll
  for (int i = 0; i < MAXPF; ++i) {
    const int e = tid + i * nthreads;
    soff[i] = -3; loff[i] = 0;
    if (e < BT * NC) {
      const int t = e / NC, col = e - t * NC;
      loff[i] = t * S + col;
      if (col < NP) {
        int pa = 0;
        while ((pa + 1) * (pa + 2) / 2 <= col) ++pa;
        const int pb = col - pa * (pa + 1) / 2;
        soff[i] = (pa < p) ? (t * pp + pa * p + pb) : -2;                       // >= 0: post_vsm element (bin t relative to the tile)
      } else if (col < NP + PW) {
        const int k = col - NP;
        soff[i] = (k < p) ? -(16 + k * 64 + t) : -2;                             // <= -16: mean[k][t0 + t]   (t < 64)
      } else {
        soff[i] = -1;                                                            // the column of ones
      }
    }
  }
  mdouble4 acc[NTM > 0 ? NTM : 1];
#pragma unroll
  for (int tl = 0; tl < NTM; ++tl) acc[tl] = mdouble4{0.0, 0.0, 0.0, 0.0};
  double accv[NV > 0 ? NV : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) accv[i] = 0.0;

  const int ntt = (T + BT - 1) / BT;
  const int nitems = a.ntr * ntt;
  double pf[MAXPF];
  // (every load is issued unconditionally at a clamped, always valid address and masked afterwards: conditional loads make
  // the compiler wait for each one before issuing the next, which serialises the memory latencies of a tile)
  auto prefetch = [&](int itrial, int itile) {
    const size_t r = a.trials[itrial];
    const int t0 = itile * BT;
    const int tn = (T - t0 < BT) ? T - t0 : BT;
    const double* vsm = a.vsm + (r * T + t0) * pp;
    const double* mean = a.mean + r * p * T + t0;
    double raw[MAXPF];
    bool ok[MAXPF];
#pragma unroll
    for (int i = 0; i < MAXPF; ++i) {
      const int s = soff[i];
      const int kt = -s - 16, k = kt >> 6, t = kt & 63;
      const bool is_v = s >= 0, is_m = s <= -16;
      ok[i] = is_v ? (s < tn * pp) : (is_m && t < tn);
      const double* src = is_v ? vsm : mean;
      const long long off = is_v ? (long long)s : (long long)k * T + t;
      raw[i] = src[ok[i] ? off : 0];
    }
#pragma unroll
    for (int i = 0; i < MAXPF; ++i) pf[i] = ok[i] ? raw[i] : (soff[i] == -1 ? 1.0 : 0.0);
  };
  // zero both stages once: the gap columns NC..S-1 of a row and the slack behind the last row are never staged
  for (int e = tid; e < 2 * M::LDS_DOUBLES; e += nthreads) lds[e] = 0.0;
  // items = (trial, tile) pairs, walked with stride gridDim.x; the pair of the current and of the next item is kept in
  // scalar counters (no integer division in the loop)
  const int step_tr = gridDim.x / ntt, step_ti = gridDim.x % ntt;
  int item = blockIdx.x;
  int c_tr = item / ntt, c_ti = item % ntt;
  if (item < nitems) prefetch(c_tr, c_ti);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXPF; ++i)
    if (soff[i] != -3) lds[loff[i]] = pf[i];
  __syncthreads();
  int cur = 0;
  for (; item < nitems; item += gridDim.x) {
    const int t0 = c_ti * BT;
    const int tn = (T - t0 < BT) ? T - t0 : BT;
    const bool more = item + (int)gridDim.x < nitems;
    c_tr += step_tr; c_ti += step_ti;
    if (c_ti >= ntt) { c_ti -= ntt; c_tr += 1; }
    if (more && !(a.dbg & 8)) prefetch(c_tr, c_ti);     // global loads in flight while this stage is multiplied
    const double* st = lds + cur * M::LDS_DOUBLES;
    if (active_wave) {
#pragma unroll
      for (int sub = 0; sub < BT / 16; ++sub) {
        if (sub * 16 >= tn) break;
        const double* row = st + (sub * 16 + l15) * S + l4;
        // two partial accumulators: consecutive MFMAs do not wait for each other's result
        mdouble4 h = {dn, dn, dn, dn}, h2 = {0.0, 0.0, 0.0, 0.0};
        if (!(a.dbg & 4))
#pragma unroll
        for (int kk = 0; kk + 1 < KSM; kk += 2) {
          h = __builtin_amdgcn_mfma_f64_16x16x4f64(row[4 * kk], th[kk], h, 0, 0, 0);
          h2 = __builtin_amdgcn_mfma_f64_16x16x4f64(row[4 * kk + 4], th[kk + 1], h2, 0, 0, 0);
        }
        if constexpr (KSM & 1) h = __builtin_amdgcn_mfma_f64_16x16x4f64(row[4 * (KSM - 1)], th[KSM - 1], h, 0, 0, 0);
        double yh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double* bin = st + (sub * 16 + 4 * r + l4) * S;        // this lane's bin of register r
          double hv = h[r] + h2[r];
#pragma unroll
          for (int i = 0; i < KV; ++i) hv += bin[4 * KSM + i] * thv[i];
          const double e = (live && sub * 16 + 4 * r + l4 < tn) ? ((a.dbg & 1) ? 1.0 + 1e-3 * hv : exp(hv)) : 0.0;
          yh[r] = e;
#pragma unroll
          for (int i = 0; i < NV; ++i) accv[i] += bin[16 * NTM + i] * e;
        }
        if (!(a.dbg & 2))
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double* col = st + (sub * 16 + 4 * r + l4) * S + l15;
#pragma unroll
          for (int tl = 0; tl < NTM; ++tl) acc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(col[16 * tl], yh[r], acc[tl], 0, 0, 0);
        }
      }
    }
    if (more && !(a.dbg & 8)) {
      // the other stage was fully consumed before the barrier that ended the previous iteration
      double* nx = lds + (cur ^ 1) * M::LDS_DOUBLES;
#pragma unroll
      for (int i = 0; i < MAXPF; ++i)
        if (soff[i] != -3) nx[loff[i]] = pf[i];
    }
    __syncthreads();
    cur ^= 1;
  }
  // epilogue: Out[c'][n] sits in the accumulators of the 4 lanes (l4 = 0..3) that share a neuron: c' = 16 tl + 4 r + l4
  // (matrix-core rows), or as per-lane partial sums over the lane's bins (vector rows).  Every lane folds its entries into
  // u = sum_t yhat m + A_n c_n (each pair entry feeds two components) and sum_t yhat with compile-time register indices,
  // the 4 lanes are summed by two xor-shuffles (fixed order), lane l4 = 0 stores.
  if (active_wave) {
    double c[PW], u[PW];
#pragma unroll
    for (int l = 0; l < PW; ++l) { c[l] = (l < p) ? a.vec[(size_t)l * q + nc] : 0.0; u[l] = 0.0; }
    double sy = 0.0;
    auto fold = [&](int cr, double v) {
      if (cr < NP) {
        int pa = 0;
        while ((pa + 1) * (pa + 2) / 2 <= cr) ++pa;
        const int pb = cr - pa * (pa + 1) / 2;
        double ca = 0.0, cb = 0.0;
#pragma unroll
        for (int l = 0; l < PW; ++l) { ca = (l == pa) ? c[l] : ca; cb = (l == pb) ? c[l] : cb; }
        const double va = v * cb, vb = (pa != pb) ? v * ca : 0.0;
#pragma unroll
        for (int l = 0; l < PW; ++l) u[l] += ((l == pa) ? va : 0.0) + ((l == pb) ? vb : 0.0);
      } else if (cr < NP + PW) {
#pragma unroll
        for (int l = 0; l < PW; ++l) u[l] += (l == cr - NP) ? v : 0.0;
      } else if (cr == NP + PW) {
        sy += v;
      }
    };
#pragma unroll
    for (int tl = 0; tl < NTM; ++tl)
#pragma unroll
      for (int r = 0; r < 4; ++r) fold(16 * tl + 4 * r + l4, acc[tl][r]);
#pragma unroll
    for (int i = 0; i < NV; ++i) fold(16 * NTM + i, accv[i]);
#pragma unroll
    for (int l = 0; l < PW; ++l) {
      double w = u[l];
      w += __shfl_xor(w, 16);
      w += __shfl_xor(w, 32);
      u[l] = w;
    }
    sy += __shfl_xor(sy, 16);
    sy += __shfl_xor(sy, 32);
    if (live && l4 == 0) {
      double* part = a.part + (size_t)blockIdx.x * (p + 2) * q;
#pragma unroll
      for (int l = 0; l < PW; ++l)
        if (l < p) part[(size_t)l * q + n] = -u[l];
      part[(size_t)p * q + n] = -sy;
      part[(size_t)(p + 1) * q + n] = -sy;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// (C,d) Newton pass on the matrix cores: per neuron cost, gradient and the packed (p+1) x (p+1) Hessian
//   H_n = sum_t yhat_nt ( [w; 1][w; 1]^T + [V_t 0; 0 0] ),  w = m_t + V_t c_n,  yhat = exp(d_n + c_n.m_t + c_n^T V_t c_n / 2)
// in two matrix-core stages per (16 neurons = one wave) x (4 bins):
//   1. per bin, V_t c_n for the 16 neurons as one 16 x 16 x p product (A = V_t read through its packed pairs, B = the loadings):
//      D[i][n] = (V_t c_n)_i.  The four bins' chains are independent and issued interleaved.  The exponent is a sum over components, i.e.
//      over registers and over the 4 lanes that share a neuron: the four bins' partial sums are transposed-and-reduced across those lanes
//      with three row swaps (v_permlane16_swap / v_permlane32_swap: lane (neuron, l4) ends up with the sum of bin l4), so ONE exp and one
//      sqrt serve the four bins, and sqrt(yhat) goes back to the four lanes with three more swaps;  u = sqrt(yhat) [w; 1].
//   2. per neuron, H_n += sum over the 4 bins of u u^T on v_mfma_f64_4x4x4_4b: four independent 4 x 4 x 4 products per instruction
//      (19 cycles against 68 for the 16 x 16 x 4 shape, measured: tools/probes/mfma_4x4x4_probe.hip), operand lane (k = lane >> 4,
//      block = (lane >> 2) & 3, row = lane & 3) for A and B alike, result lane (row = lane >> 4, block, column = lane & 3).  The four blocks of
//      an instruction are four neurons, the instruction index runs over the lower 4 x 4 block pairs (I >= J) of the Hessian: its A and B
//      operands are the SAME three registers per neuron quad (rows 4I..4I+3 of u over the 4 bins), 24 instructions and 24 accumulator
//      registers per 16 neurons where the 16 x 16 x 4 shape took 16 instructions of four times the length and 64 registers.  The hand-over
//      from stage 1 (lane = neuron, registers = components) goes through a wave-private LDS tile [4 bins][16 components][16 neurons]
//      (component stride 20, bin stride 336 doubles: writes and reads are conflict-free);
//   and sum_t yhat V_t as 4 more 16 x 16 x 4 MFMAs per 4 bins (A = the pair columns of the staged tile, B = yhat: as in mstep_cd_mfma_kernel).
// Staging, launch shape and the hoisted count terms are those of mstep_cd_mfma_kernel; part layout of mstep_cd_hess_kernel:
// [NH][q], NH = 1 + (p+1) + (p+1)(p+2)/2: cost | gradient | packed lower Hessian - WITHOUT the count terms (cd_hess_add_ym_kernel).
// PW <= 10 (p + 1 <= 12 components: three 4-row blocks).
constexpr int CDH_NW = 8;
template <int PW>
struct CdH {
  static constexpr int KSJ = (PW + 3) / 4;                          // k steps of stage 1
  static constexpr int NBK = (PW + 1 + 3) / 4;                      // 4-row blocks of the (p+1)-dim Hessian
  static constexpr int NPB = NBK * (NBK + 1) / 2;                   // lower block pairs = instructions per neuron quad
  static constexpr int NHH = (PW + 1) * (PW + 2) / 2;               // packed Hessian entries (template width)
  static constexpr int CS = 20, BS = 16 * CS + 16;                  // component / bin stride of the wave tile (doubles; BS = 16 mod 32)
  static constexpr int UW = (4 * BS > 16 * NHH) ? 4 * BS : 16 * NHH;   // doubles of the wave-private tile (also the epilogue's scratch)
  static constexpr int SH = CdM<PW>::S + 2;                         // row stride of the staged Phi tile: two gap columns that stay zero
  static constexpr int ZC = CdM<PW>::NC;                            // the first of them
  static constexpr int BT = 32;                                     // bins per staged tile (the next tile waits in 5 registers per thread, not 9)
  static constexpr int LDS_DOUBLES = BT * SH + 32;                  // one stage
};

typedef unsigned cdh_u2 __attribute__((ext_vector_type(2)));
// rows = the four 16-lane groups of a wave.  swap16: (a, b) <- a keeps its even rows and takes b's even rows into its odd rows, b takes a's odd rows into
// its even rows and keeps its odd rows;  swap32: a <- [a.lo, b.lo], b <- [a.hi, b.hi] over the 32-lane halves.
__device__ __forceinline__ void cdh_swap16(double& a, double& b) {
  const cdh_u2 r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const cdh_u2 r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)r1[0], (int)r0[0]); b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void cdh_swap32(double& a, double& b) {
  const cdh_u2 r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const cdh_u2 r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)r1[0], (int)r0[0]); b = __hiloint2double((int)r1[1], (int)r0[1]);
}

template <int PW>
__global__ __launch_bounds__(64 * CDH_NW) void mstep_cd_hess_mfma_kernel(CdArgs a, int tiles_per_group) {
  using M = CdM<PW>;
  using H = CdH<PW>;
  constexpr int NP = M::NP, NC = M::NC, S = H::SH, ZC = H::ZC, BT = H::BT, KSJ = H::KSJ, NBK = H::NBK, NPB = H::NPB, NHH = H::NHH, UW = H::UW, CS = H::CS, BS = H::BS;
  constexpr int MAXPF = (BT * NC + 64 * CDH_NW - 1) / (64 * CDH_NW);
  constexpr int NTP = (NP + 15) / 16;                               // row tiles of the pair product
  extern __shared__ double lds[];                                   // Phi tile [2][BT][S] (+ slack) | wave tiles [CDH_NW][UW]
  const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int l15 = lane & 15, l4 = lane >> 4, x4 = lane & 3, blk = (lane >> 2) & 3;
  const int nthreads = 64 * blockDim.y, tid = wave * 64 + lane;
  const int p = a.p, q = a.q, T = a.T, pp = p * p;
  const int n0 = (blockIdx.y * tiles_per_group + wave) * 16;
  const bool active_wave = wave < tiles_per_group && n0 < q;        // other waves only help staging
  const int n = n0 + l15;
  const bool live = active_wave && n < q;
  const int nc = (n < q) ? n : q - 1;
  double* Ub = lds + 2 * H::LDS_DOUBLES + (size_t)wave * UW;

  // loadings of this lane's neuron: cq[r] = c_n[4 r + l4] - the B fragments of stage 1 (k = latent) and the weights of the exponent
  double cq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) cq[r] = (4 * r + l4 < p) ? a.vec[(size_t)(4 * r + l4) * q + nc] : 0.0;
  const double dn = a.vec[(size_t)p * q + nc];
  // stage-1 A fragments: V_t[i = l15][j = 4 kk + l4] sits in the pair column max(i,j)(max(i,j)+1)/2 + min(i,j) of the staged row
  // (entries outside the p x p block read a gap column of the staged row, which is zero: no selects, no branches in the loop)
  int poff[KSJ];
#pragma unroll
  for (int kk = 0; kk < KSJ; ++kk) {
    const int i = l15, j = 4 * kk + l4;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    poff[kk] = (i < p && j < p) ? hi * (hi + 1) / 2 + lo : ZC;
  }
  // what is added to (V_t c_n)_i for this lane's components i = 4 r + l4: the mean (i < p), the column of ones (i = p: the offset's
  // component of [w; 1]), the zero column beyond - [w; 1; 0..] comes out of one addition
  int moff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + l4;
    moff[r] = (i < p) ? NP + i : (i == p ? NP + PW : ZC);
  }
  int pcol[NTP];                                                    // pair columns of the last product's A fragments (past the pairs: zero column)
#pragma unroll
  for (int tl = 0; tl < NTP; ++tl) pcol[tl] = (16 * tl + l15 < NP) ? 16 * tl + l15 : ZC;
  // per-thread staging plan (as mstep_cd_mfma_kernel)
  int soff[MAXPF], loff[MAXPF];
#pragma unroll
  for (int i = 0; i < MAXPF; ++i) {
    const int e = tid + i * nthreads;
    soff[i] = -3; loff[i] = 0;
    if (e < BT * NC) {
      const int t = e / NC, col = e - t * NC;
      loff[i] = t * S + col;
      if (col < NP) {
        int pa = 0;
        while ((pa + 1) * (pa + 2) / 2 <= col) ++pa;
        const int pb = col - pa * (pa + 1) / 2;
        soff[i] = (pa < p) ? (t * pp + pa * p + pb) : -2;
      } else if (col < NP + PW) {
        const int k = col - NP;
        soff[i] = (k < p) ? -(16 + k * 64 + t) : -2;
      } else {
        soff[i] = -1;
      }
    }
  }
  double acc[4][NPB];
  mdouble4 pacc[NTP];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NPB; ++j) acc[i][j] = 0.0;
#pragma unroll
  for (int i = 0; i < NTP; ++i) pacc[i] = mdouble4{0.0, 0.0, 0.0, 0.0};

  const int ntt = (T + BT - 1) / BT;
  const int nitems = a.ntr * ntt;
  double pf[MAXPF];
  auto prefetch = [&](int itrial, int itile) {
    const size_t r = a.trials[itrial];
    const int t0 = itile * BT;
    const int tn = (T - t0 < BT) ? T - t0 : BT;
    const double* vsm = a.vsm + (r * T + t0) * pp;
    const double* mean = a.mean + r * p * T + t0;
    double raw[MAXPF];
    bool ok[MAXPF];
#pragma unroll
    for (int i = 0; i < MAXPF; ++i) {
      const int s = soff[i];
      const int kt = -s - 16, k = kt >> 6, t = kt & 63;
      const bool is_v = s >= 0, is_m = s <= -16;
      ok[i] = is_v ? (s < tn * pp) : (is_m && t < tn);
      const double* src = is_v ? vsm : mean;
      const long long off = is_v ? (long long)s : (long long)k * T + t;
      raw[i] = src[ok[i] ? off : 0];
    }
#pragma unroll
    for (int i = 0; i < MAXPF; ++i) pf[i] = ok[i] ? raw[i] : (soff[i] == -1 ? 1.0 : 0.0);
  };
  for (int e = tid; e < 2 * H::LDS_DOUBLES; e += nthreads) lds[e] = 0.0;
  const int step_tr = gridDim.x / ntt, step_ti = gridDim.x % ntt;
  int item = blockIdx.x;
  int c_tr = item / ntt, c_ti = item % ntt;
  if (item < nitems) prefetch(c_tr, c_ti);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXPF; ++i)
    if (soff[i] != -3) lds[loff[i]] = pf[i];
  __syncthreads();
  int cur = 0;
  for (; item < nitems; item += gridDim.x) {
    const int t0 = c_ti * BT;
    const int tn = (T - t0 < BT) ? T - t0 : BT;
    const bool more = item + (int)gridDim.x < nitems;
    c_tr += step_tr; c_ti += step_ti;
    if (c_ti >= ntt) { c_ti -= ntt; c_tr += 1; }
    if (more) prefetch(c_tr, c_ti);
    const double* st = lds + cur * H::LDS_DOUBLES;
    if (active_wave) {
      for (int tb0 = 0; tb0 < tn; tb0 += 4) {
        // stage 1: h[b][r] = (V_t c_n)_{4 r + l4} for the bins b = tb0 .. tb0 + 3 (rows past tn are zero-filled stage rows: yhat is masked below)
        mdouble4 h[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b] = mdouble4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KSJ; ++kk)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            h[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(st[(size_t)(tb0 + b) * S + poff[kk]], cq[kk], h[b], 0, 0, 0);
          }
        double wv[4][4], sb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          double s = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double mi = st[(size_t)(tb0 + b) * S + moff[r]];
            wv[b][r] = h[b][r] + mi;
            s += cq[r] * (0.5 * h[b][r] + mi);
          }
          sb[b] = s;
        }
        // lane (neuron, l4) <- the exponent of bin tb0 + l4
        cdh_swap16(sb[0], sb[1]);
        cdh_swap16(sb[2], sb[3]);
        double t01 = sb[0] + sb[1], t23 = sb[2] + sb[3];
        cdh_swap32(t01, t23);
        const bool on = live && tb0 + l4 < tn;
        const double yh = exp(on ? dn + (t01 + t23) : -1000.0);      // (exp(-1000) = 0: masked bins and neurons drop out of every sum)
        const double sq = sqrt(yh);
        double sqb[4];
        sqb[0] = sq; sqb[1] = sq;
        cdh_swap16(sqb[0], sqb[1]);
        sqb[2] = sqb[0]; sqb[3] = sqb[1];
        cdh_swap32(sqb[0], sqb[2]);
        cdh_swap32(sqb[1], sqb[3]);                                  // sqb[b] = sqrt(yhat) of bin tb0 + b
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) Ub[b * BS + (4 * r + l4) * CS + l15] = sqb[b] * wv[b][r];
        asm volatile("" ::: "memory");                               // (LDS operations of a wave execute in order: only the compiler must keep it)
        // stage 2: per neuron quad, the lower block pairs of sum over the 4 bins of u u^T
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
          double v[NBK];
#pragma unroll
          for (int I = 0; I < NBK; ++I) v[I] = Ub[l4 * BS + (4 * I + x4) * CS + 4 * mm + blk];
          int pr = 0;
#pragma unroll
          for (int I = 0; I < NBK; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {
              acc[mm][pr] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[I], v[J], acc[mm][pr], 0, 0, 0);
              ++pr;
            }
        }
        // sum_t yhat V_t: A = pair columns of the 4 bins, B = yhat of bin tb0 + l4 (this lane's own)
        const double* prow = st + (size_t)(tb0 + l4) * S;
#pragma unroll
        for (int tl = 0; tl < NTP; ++tl) pacc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(prow[pcol[tl]], yh, pacc[tl], 0, 0, 0);
        asm volatile("" ::: "memory");                               // the tile is rewritten by the next group of bins
      }
    }
    if (more) {
      double* nx = lds + (cur ^ 1) * H::LDS_DOUBLES;
#pragma unroll
      for (int i = 0; i < MAXPF; ++i)
        if (soff[i] != -3) nx[loff[i]] = pf[i];
    }
    __syncthreads();
    cur ^= 1;
  }
  // epilogue: per neuron the packed lower Hessian in the wave's tile, Hs[nn][idx(i,j)], idx = i(i+1)/2 + j over components 0..p
  // (component p = the offset d): first the u u^T sums (lane (l4, blk, x4) of pair (I, J) of quad mm = entry (4I + l4, 4J + x4) of neuron
  // 4 mm + blk), then the pair sums (lane = neuron)
  if (active_wave) {
    double* Hs = Ub;
    for (int e = lane; e < 16 * NHH; e += 64) Hs[e] = 0.0;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      int pr = 0;
#pragma unroll
      for (int I = 0; I < NBK; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
          const int i = 4 * I + l4, j = 4 * J + x4, nn = 4 * mm + blk;
          if (i <= p && j <= i) Hs[nn * NHH + i * (i + 1) / 2 + j] = acc[mm][pr];
          ++pr;
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int tl = 0; tl < NTP; ++tl)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pr = 16 * tl + 4 * r + l4;
        if (pr < NP) {
          int pa = 0;
          while ((pa + 1) * (pa + 2) / 2 <= pr) ++pa;
          if (pa < p) Hs[l15 * NHH + pr] += pacc[tl][r];            // (pair index = packed index: both are pa(pa+1)/2 + pb)
        }
      }
    asm volatile("" ::: "memory");
    const int D = p + 1, nha = D * (D + 1) / 2;
    double* part = a.part + (size_t)blockIdx.x * (1 + D + nha) * q;
    for (int e = lane; e < 16 * nha; e += 64) {
      const int nn = e & 15, idx = e >> 4;
      if (n0 + nn < q) part[(size_t)(1 + D + idx) * q + n0 + nn] = Hs[nn * NHH + idx];
    }
    // gradient rows: sum yhat w_i = H[p][i], sum yhat = H[p][p]; cost row: -sum yhat  (count terms: cd_hess_add_ym_kernel)
    for (int e = lane; e < 16 * (D + 1); e += 64) {
      const int nn = e & 15, i = e >> 4;                             // i = 0: cost, 1..D: gradient component i - 1
      if (n0 + nn < q) {
        const double syh = Hs[nn * NHH + p * (p + 1) / 2 + p];
        part[(size_t)i * q + n0 + nn] = (i == 0) ? -syh : Hs[nn * NHH + p * (p + 1) / 2 + (i - 1)];
      }
    }
  }
}

template <int PW>
constexpr size_t cd_hess_mfma_lds_bytes() { return (2 * (size_t)CdH<PW>::LDS_DOUBLES + CDH_NW * (size_t)CdH<PW>::UW) * sizeof(double); }

// count terms of the Newton pass: sums [NH][q] as reduced from mstep_cd_hess_mfma_kernel;  row 0 += c_n.YM_n + d_n YS_n,
// gradient rows -= YM / YS  (the part layout of mstep_cd_hess_kernel: cost = sum (y hh - yhat), grad = -(sum y m - yhat w), -(sum y - yhat))
inline __global__ void cd_hess_add_ym_kernel(double* __restrict__ sums, const double* __restrict__ ym, const double* __restrict__ vec, int q, int p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= q) return;
  double lin = vec[(size_t)p * q + n] * ym[(size_t)p * q + n];
  for (int k = 0; k < p; ++k) {
    const double v = ym[(size_t)k * q + n];
    sums[(size_t)(1 + k) * q + n] -= v;
    lin += vec[(size_t)k * q + n] * v;
  }
  sums[(size_t)(1 + p) * q + n] -= ym[(size_t)p * q + n];
  sums[(size_t)n] += lin;
}

template <int PW>
constexpr size_t cd_mfma_lds_bytes() { return 2 * (size_t)CdM<PW>::LDS_DOUBLES * sizeof(double); }

}  // namespace pgpfa
