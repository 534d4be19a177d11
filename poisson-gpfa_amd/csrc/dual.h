// Dual-variational evaluation (inference.py:188-256): the contractions over neurons as GEMMs on the matrix cores.
//   W_t = C^T diag(lambda_t) C      ->  Wp (T x pairs)  = Lambda^T (T x q) . TBL[:, pairs]        TBL[n][pair(a,b)] = c_na c_nb
//   v   = C_big (lambda - y)        ->  V  (T x p)      = (Lambda - Y)^T (T x q) . TBL[:, latents] TBL[n][NPd + k]   = c_nk
//   -1/2 c_n^T Sigma_t c_n          ->  G  (T x q)      = -1/2 Sp (T x pairs) . TBL[:, pairs]^T    Sp[t][pair] = w_ab Sigma_t[a][b]
//   C_big^T K v                     ->  G += KV (T x p) . TBL[:, latents]^T
// (one thread per bin looping over neurons and latent pairs - the form these replace - costs 30 ms per evaluation of 16 trials at
// 500 neurons x 20 latents x 1000 bins; the products are ~3e8 flops per trial).  Element-wise kernels around them below.
#pragma once

namespace pgpfa {

// TBL[n][ncol] (n < qpad): columns [0, NP) pair products (a >= b at a(a+1)/2 + b), [NPd, NPd + p) the loadings, zeros elsewhere
inline __global__ void dual_table_kernel(const double* __restrict__ C, int q, int p, int ncol, int npd, double* __restrict__ tbl) {
  const int n = blockIdx.x;
  const int np = p * (p + 1) / 2;
  for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
    double v = 0.0;
    if (n < q) {
      if (c < np) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= c) ++a;
        const int b = c - a * (a + 1) / 2;
        v = C[(size_t)n * p + a] * C[(size_t)n * p + b];
      } else if (c >= npd && c < npd + p) {
        v = C[(size_t)n * p + (c - npd)];
      }
    }
    tbl[(size_t)n * ncol + c] = v;
  }
}

// lmy = lambda - y for the slots [0, nslots); partial sums per (slot, 64-bin tile): sum d_n lmy and sum lambda (log lambda - 1).
// grid = (ceil(T/64), nslots), block = 256 (lanes = bins, the 4 waves take interleaved neurons)
inline __global__ __launch_bounds__(256) void dual_pre_kernel(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, const double* __restrict__ d, const double* __restrict__ lam,
                                                       double* __restrict__ lmy, double* __restrict__ part, const int* __restrict__ trial_of_slot,
                                                       int q, int T) {
  __shared__ double red[2][4];
  const size_t slot = blockIdx.y;
  const size_t trial = trial_of_slot[slot];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  double sB = 0.0, sD = 0.0;
  if (t < T) {
    for (int n = wave; n < q; n += 4) {
      const size_t e = slot * (size_t)q * T + (size_t)n * T + t;
      const double l = lam[e];
      const double v = l - (double)count_at(Y, Yhi, (trial * q + n) * T + t);
      lmy[e] = v;
      sB += d[n] * v;
      sD += l * (log(l) - 1.0);
    }
  }
  for (int off = 32; off > 0; off >>= 1) { sB += __shfl_down(sB, off); sD += __shfl_down(sD, off); }
  if (lane == 0) { red[0][wave] = sB; red[1][wave] = sD; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[(slot * gridDim.x + blockIdx.x) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[(slot * gridDim.x + blockIdx.x) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// Rates pass of the GEMM form of the Poisson pass (latent widths beyond the matrix-core kernel's 16): for the listed slots
//   h = C x + d [+ off],  lam = exp(h),  lmy = lam - y,  partial objective sum_n,t lam - y h per (slot, 64-bin tile)
// - the q x p x T product as a vector kernel (2 % of the flops of the pass: W_t = C^T diag(lam_t) C and C^T (lam - y) follow as the GEMMs of the dual
// evaluation against the pair / loading table).  grid = (ceil(T/64), nslots), block = 256 (lanes = bins, the 4 waves take interleaved neurons).
inline __global__ __launch_bounds__(256) void rates_wide_kernel(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, const double* __restrict__ C,
                                                         const double* __restrict__ d, const double* __restrict__ X, long long sX,
                                                         const double* __restrict__ off, double* __restrict__ lam, double* __restrict__ lmy,
                                                         double* __restrict__ fpart, const int* __restrict__ slots,
                                                         const int* __restrict__ trial_of_slot, int q, int p, int T) {
  extern __shared__ double rw_x[];                 // [p][64]
  __shared__ double red[4];
  const size_t slot = slots[blockIdx.y];
  const size_t trial = trial_of_slot[slot];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = blockIdx.x * 64, t = t0 + lane;
  const bool valid = t < T;
  for (int e = threadIdx.x; e < p * 64; e += 256) {
    const int l = e >> 6, tt = t0 + (e & 63);
    rw_x[e] = tt < T ? X[slot * sX + (size_t)l * T + tt] : 0.0;
  }
  __syncthreads();
  double facc = 0.0;
  if (valid) {
    for (int n = wave; n < q; n += 4) {
      const size_t e = slot * (size_t)q * T + (size_t)n * T + t;
      double h = d[n] + (off ? off[e] : 0.0);
      const double* Cn = C + (size_t)n * p;
      for (int l = 0; l < p; ++l) h += Cn[l] * rw_x[l * 64 + lane];
      const double ev = exp(h);
      const double y = (double)count_at(Y, Yhi, (trial * q + n) * T + t);
      // (stored finite: the products that follow read Lambda^T with K = q rounded up, i.e. also the first rows of the NEXT slot against zero
      // table rows - an Inf or NaN rate of a far trial point would turn into NaN in its neighbour's curvature; the slot's own objective keeps
      // the true value, which is what rejects such a point)
      const double evs = (ev < 1e300) ? ev : 1e300;
      lam[e] = evs;
      lmy[e] = evs - y;
      facc += ev - y * h;
    }
  }
  for (int o = 32; o > 0; o >>= 1) facc += __shfl_down(facc, o);
  if (lane == 0) red[wave] = facc;
  __syncthreads();
  if (threadIdx.x == 0) fpart[slot * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// W[slot][t][a][b] = W[slot][t][b][a] = Wp[slot][pair(a,b)][t]: a transpose between the pair-major image of the GEMMs and the per-bin blocks.
// Through an LDS tile of 32 bins x all pairs (round 5): both sides move in contiguous runs - 256-byte rows of Wp in, whole p x p blocks out (the
// thread-per-entry form wrote every 8-byte entry into a cache line of its own: 0.76 ms per call at config 5, 88 calls per E-step pair).
// grid = (ceil(T / nbt), nslots), block = 256, dynamic LDS = np * (nbt + 1) doubles (nbt = dual_tile_bins(...): 32 up to 21 latents);
// slots (may be NULL): list of the slots
inline int dual_tile_bins(size_t doubles_per_bin) { int n = 32; while (n > 1 && doubles_per_bin * (n + 1) * sizeof(double) > 60000) n >>= 1; return n; }
inline __global__ __launch_bounds__(256) void dual_unpack_w_kernel(const double* __restrict__ Wp, long long sWp, double* __restrict__ W, long long sW, int T, int p,
                                                                  int nbt, const int* __restrict__ slots = nullptr) {
  extern __shared__ double duw_tile[];                // [np][nbt + 1]
  const int np = p * (p + 1) / 2, pp = p * p, ldt = nbt + 1;
  const int t0 = blockIdx.x * nbt, nt = min(nbt, T - t0);
  const size_t slot = slots ? (size_t)slots[blockIdx.y] : (size_t)blockIdx.y;
  const double* src = Wp + slot * sWp + t0;
  for (int e = threadIdx.x; e < np * nbt; e += 256) {
    const int c = e / nbt, tt = e - c * nbt;
    duw_tile[c * ldt + tt] = (tt < nt) ? src[(size_t)c * T + tt] : 0.0;
  }
  __syncthreads();
  double* dst = W + slot * sW + (size_t)t0 * pp;
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const int tt = e / pp, idx = e - tt * pp, a = idx / p, b = idx - a * p;
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    dst[e] = duw_tile[(hi * (hi + 1) / 2 + lo) * ldt + tt];
  }
}

// Sp[slot][pair][t] = (a == b ? 1 : 2) * Sigma_t[a][b] of the slot's trial, zero rows up to npd: the transpose the other way, through the same kind
// of LDS tile (whole p x p blocks in, rows of Sp out).  grid = (ceil(T / nbt), nslots), block = 256, dynamic LDS = nbt * (p p + 1) doubles
inline __global__ __launch_bounds__(256) void dual_pack_sigma_kernel(const double* __restrict__ vsm, const int* __restrict__ trial_of_slot, double* __restrict__ Sp,
                                                                    long long sSp, int T, int p, int npd, int nbt) {
  extern __shared__ double dps_tile[];                // [nbt][pp + 1]
  const int np = p * (p + 1) / 2, pp = p * p, ldt = pp + 1;
  const int t0 = blockIdx.x * nbt, nt = min(nbt, T - t0);
  const double* src = vsm + ((size_t)trial_of_slot[blockIdx.y] * T + t0) * pp;
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const int tt = e / pp, idx = e - tt * pp;
    dps_tile[tt * ldt + idx] = src[e];
  }
  __syncthreads();
  double* dst = Sp + (size_t)blockIdx.y * sSp + t0;
  for (int e = threadIdx.x; e < npd * nbt; e += 256) {
    const int c = e / nbt, tt = e - c * nbt;
    if (tt >= nt) continue;
    double v = 0.0;
    if (c < np) {
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= c) ++a;
      const int b = c - a * (a + 1) / 2;
      v = dps_tile[tt * ldt + a * p + b] * (a == b ? 1.0 : 2.0);
    }
    dst[(size_t)c * T + tt] = v;
  }
}

// The reference inverts  postPrecision + 1e-6 diag(diag(postPrecision))  (VIPostCov, inference.py:190).  The diagonal of the precision
// K^-1 + scatter(W_t) at (latent k, bin t) is (K_k^-1)_tt + W_t[k][k], so the jitter is a diagonal addition to the per-bin blocks and
// nothing else:  W_t[k][k] <- (1 + jit) W_t[k][k] + jit (K_k^-1)_tt.  Every later step (per-bin blocks, r x r system, log det, Sigma_t) then
// evaluates the reference's jittered matrix exactly.  grid = (ceil(T*p/256), nslots)
inline __global__ void dual_jitter_kernel(double* __restrict__ W, long long sW, const double* __restrict__ Kinv, int Tp, int T, int p, double jit) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= T * p) return;
  const int t = e / p, k = e - t * p;
  double* w = W + (size_t)blockIdx.y * sW + (size_t)t * p * p + k * p + k;
  *w = (1.0 + jit) * (*w) + jit * Kinv[(size_t)k * Tp * Tp + (size_t)t * Tp + t];
}

// grad[slot][n][t] += log(lambda) - d_n.  grid = (ceil(q*T/256), nslots)
inline __global__ void dual_grad_finish_kernel(double* __restrict__ grad, const double* __restrict__ lam, const double* __restrict__ d, int q, int T) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)q * T) return;
  const size_t o = (size_t)blockIdx.y * q * T + e;
  grad[o] += log(lam[o]) - d[e / T];
}

}  // namespace pgpfa
