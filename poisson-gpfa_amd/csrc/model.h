// Model-specific streaming kernels of the Poisson-GPFA EM hot path (gfx950).
// Each kernel cites the reference expression it evaluates (file:line under the reference tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double mdouble4 __attribute__((ext_vector_type(4)));

namespace pgpfa {

// --------------------------------------------------------------------------------------------------
// Gram matrices.  util.makeK_big util.py:599-619:
//   K[k][i][j] = (1-eps) * exp(-0.5 * ((i*bin - j*bin)^2 / (tau_k*1000)^2)) + eps*[i==j]
// written into a (Tp x Tp) slab per latent with identity padding (rows/cols >= T).
// --------------------------------------------------------------------------------------------------
inline __global__ void gram_tau_kernel(double* __restrict__ K, int Tp, int T, const double* __restrict__ tau, double bin, double eps) {
  const int k = blockIdx.y;
  const int j = blockIdx.x;
  double* Kk = K + (size_t)k * Tp * Tp + (size_t)j * Tp;
  const double den = (tau[k] * 1000.0) * (tau[k] * 1000.0);
  for (int i = threadIdx.x; i < Tp; i += blockDim.x) {
    double v;
    if (i < T && j < T) {
      const double dt = (double)i * bin - (double)j * bin;
      v = (1.0 - eps) * exp(-0.5 * ((dt * dt) / den));
      if (i == j) v += eps;
    } else {
      v = (i == j) ? 1.0 : 0.0;
    }
    Kk[i] = v;
  }
}

// learning.MStepGPtimescaleCost learning.py:183-185 (gamma = exp(p), lags in bins):
//   temp = (1-eps)*exp(-exp(p)/2 * difSq) ; K = temp + eps*I ; dKdgamma = -0.5*temp*difSq
inline __global__ void gram_gamma_kernel(double* __restrict__ K, double* __restrict__ M, int Tp, int T, double logp, double eps) {
  const int j = blockIdx.x;
  const double g = exp(logp);
  for (int i = threadIdx.x; i < Tp; i += blockDim.x) {
    double kv, mv;
    if (i < T && j < T) {
      const double dd = (double)(i - j);
      const double dsq = dd * dd;
      const double temp = (1.0 - eps) * exp(-g / 2.0 * dsq);
      kv = temp + (i == j ? eps : 0.0);
      mv = -0.5 * temp * dsq;
    } else {
      kv = (i == j) ? 1.0 : 0.0;
      mv = 0.0;
    }
    K[(size_t)j * Tp + i] = kv;
    M[(size_t)j * Tp + i] = mv;
  }
}

// batched over latents: K/M slabs of latent blockIdx.y at log-gamma logp[blockIdx.y]
inline __global__ void gram_gamma_batch_kernel(double* __restrict__ K, double* __restrict__ M, int Tp, int T, const double* __restrict__ logp, double eps) {
  const int j = blockIdx.x;
  const size_t off = (size_t)blockIdx.y * Tp * Tp;
  const double g = exp(logp[blockIdx.y]);
  for (int i = threadIdx.x; i < Tp; i += blockDim.x) {
    double kv, mv;
    if (i < T && j < T) {
      const double dd = (double)(i - j);
      const double dsq = dd * dd;
      const double temp = (1.0 - eps) * exp(-g / 2.0 * dsq);
      kv = temp + (i == j ? eps : 0.0);
      mv = -0.5 * temp * dsq;
    } else {
      kv = (i == j) ? 1.0 : 0.0;
      mv = 0.0;
    }
    K[off + (size_t)j * Tp + i] = kv;
    M[off + (size_t)j * Tp + i] = mv;
  }
}

// out[blockIdx.x] = 2 * sum_i log L_ii of slab blockIdx.x
inline __global__ void logdet_batch_kernel(const double* __restrict__ L, long long sL, int ld, int n, double* __restrict__ out) {
  __shared__ double red[256];
  const double* Ls = L + (size_t)blockIdx.x * sL;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += log(Ls[(size_t)i * ld + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = 2.0 * red[0];
}

// batched dot of equally laid out slabs: part[blockIdx.y][blockIdx.x]; then sum_part_batch
// (bmod > 0: B is indexed by blockIdx.y % bmod - several batch entries share one B slab)
inline __global__ void dot_part_batch_kernel(const double* __restrict__ A, long long sA, const double* __restrict__ B, long long sB, long long n,
                                      double* __restrict__ part, int bmod) {
  __shared__ double red[256];
  const double* a = A + (size_t)blockIdx.y * sA;
  const double* b = B + (size_t)(bmod > 0 ? blockIdx.y % bmod : blockIdx.y) * sB;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += a[i] * b[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
inline __global__ void sum_part_batch_kernel(const double* __restrict__ part, int nper, double* __restrict__ out, int nbatch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbatch) return;
  double s = 0.0;
  for (int i = 0; i < nper; ++i) s += part[(size_t)b * nper + i];
  out[b] = s;
}

// sum_i log(diag(L)) * 2 for one slab
inline __global__ void logdet_kernel(const double* __restrict__ L, int ld, int n, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += log(L[(size_t)i * ld + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = 2.0 * red[0];
}

// deterministic two-stage dot product of two equally laid out arrays: part[blockIdx.x]
inline __global__ void dot_part_kernel(const double* __restrict__ A, const double* __restrict__ B, long long n, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += A[i] * B[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
inline __global__ void sum_part_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += part[i];
    out[0] = s;
  }
}

// --------------------------------------------------------------------------------------------------
// Poisson likelihood pass (inference.py:22-26, 38-43, 56-63 in structured form), per trial:
//   h = C X + d ; e = exp(h) ; f_lik = sum(e - y*h) ; G = C^T (e - y)  (p x T)
//   W[t] = C^T diag(e[:,t]) C  (p x p per bin)
// Block = 64 bins (lanes) x KY latents.  Phase A: the block's threads share out a chunk of 32
// neurons and leave e and e-y in LDS; phase B: thread (t,k) accumulates G[k][t] and row k of W[t]
// with wave-uniform (scalar) reads of C.  Spike counts are read as packed uint8, coalesced over t.
// --------------------------------------------------------------------------------------------------
// Spike counts live in HBM as one byte per (trial, neuron, bin), coalesced over bins; counts above 255 (long bins of fast units - the
// reference keeps counts as float64 / int64 of any size, util.py:741,750) put their high byte into a second plane of the same layout
// that exists only when the tensor holds such a count (NULL otherwise: a wave-uniform branch).
__device__ __forceinline__ unsigned count_at(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, size_t i) {
  unsigned v = Y[i];
  if (Yhi) v |= (unsigned)Yhi[i] << 8;
  return v;
}

struct PoissonArgs {
  const uint8_t* Y;      // [R][q][T]
  const uint8_t* Yhi;    // high bytes of the counts, NULL when every count fits one byte
  const double* C;       // [q][p]
  const double* d;       // [q]
  const double* X; long long sX;   // per slot [p][T]
  double* G; long long sG;         // per slot [p][T] (likelihood part)
  double* W; long long sW;         // per slot [T][p][p]
  double* fpart;                   // [slot][ntile]
  const int* slots;                // list of slots to process
  const int* trial_of_slot;
  const int* mask;                 // optional per slot: index of a neuron left out of the likelihood (NULL / -1: none)
  // optional per slot [q][T]: an offset added to the log rate, h = C x + d + off (the variance term 1/2 c_n^T Sigma_t c_n of the
  // variational fixed point, pgpfa_dual_fixed_point), and an output for the rates exp(h) themselves
  const double* off; long long sOff;
  double* lam_out; long long sLam;
  int q, p, T, ntile, full;
};

constexpr int PNC = 32;

template <int PMAX>
__global__ __launch_bounds__(PMAX == 20 ? 640 : 1024) void poisson_pass_kernel(PoissonArgs a) {
  constexpr int NK = (PMAX > 16) ? 2 : 1;
  __shared__ double E[PNC][64];
  __shared__ double Rr[PNC][64];
  __shared__ double xs[PMAX][64];
  __shared__ double fred[16];
  const int tx = threadIdx.x, ty = threadIdx.y, KY = blockDim.y;
  const int slot = a.slots[blockIdx.y];
  const int trial = a.trial_of_slot[slot];
  const int held_out = a.mask ? a.mask[slot] : -1;
  const int t = blockIdx.x * 64 + tx;
  const bool valid = t < a.T;
  const int p = a.p, q = a.q, T = a.T;
  const double* X = a.X + (size_t)slot * a.sX;
  const uint8_t* Y = a.Y + (size_t)trial * q * T;
  const uint8_t* Yh = a.Yhi ? a.Yhi + (size_t)trial * q * T : nullptr;

  for (int l = ty; l < p; l += KY) xs[l][tx] = valid ? X[(size_t)l * T + t] : 0.0;
  __syncthreads();

  double g[NK];
  double w[NK][PMAX];
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    g[j] = 0.0;
#pragma unroll
    for (int l = 0; l < PMAX; ++l) w[j][l] = 0.0;
  }
  double facc = 0.0;

  for (int n0 = 0; n0 < q; n0 += PNC) {
    for (int nn = ty; nn < PNC; nn += KY) {
      const int n = n0 + nn;
      double e = 0.0, r = 0.0;
      if (n < q && valid && n != held_out) {
        double h = a.d[n];
        if (a.off) h += a.off[(size_t)slot * a.sOff + (size_t)n * T + t];
        const double* Cn = a.C + (size_t)n * p;
        for (int l = 0; l < p; ++l) h += Cn[l] * xs[l][tx];
        e = exp(h);
        if (a.lam_out) a.lam_out[(size_t)slot * a.sLam + (size_t)n * T + t] = e;
        const double y = (double)count_at(Y, Yh, (size_t)n * T + t);
        r = e - y;
        facc += e - y * h;
      }
      E[nn][tx] = e;
      Rr[nn][tx] = r;
    }
    __syncthreads();
    if (a.full) {
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        const int k = __builtin_amdgcn_readfirstlane(ty + j * KY);
        if (k < p) {
          const int nmax = (q - n0 < PNC) ? q - n0 : PNC;
          for (int nn = 0; nn < nmax; ++nn) {
            const double* Cn = a.C + (size_t)(n0 + nn) * p;
            const double cnk = Cn[k];
            g[j] += cnk * Rr[nn][tx];
            const double ce = cnk * E[nn][tx];
#pragma unroll
            for (int l = 0; l < PMAX; ++l)
              if (l <= k) w[j][l] += ce * Cn[l];
          }
        }
      }
    }
    __syncthreads();
  }

  if (a.full && valid) {
    double* G = a.G + (size_t)slot * a.sG;
    double* W = a.W + (size_t)slot * a.sW + (size_t)t * p * p;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int k = ty + j * KY;
      if (k < p) {
        G[(size_t)k * T + t] = g[j];
#pragma unroll
        for (int l = 0; l < PMAX; ++l)
          if (l <= k) {
            W[k * p + l] = w[j][l];
            W[l * p + k] = w[j][l];
          }
      }
    }
  }
  // deterministic block reduction of the objective partial: lanes, then waves in order
  for (int off = 32; off > 0; off >>= 1) facc += __shfl_down(facc, off);
  if (tx == 0) fred[ty] = facc;
  __syncthreads();
  if (tx == 0 && ty == 0) {
    double s = 0.0;
    for (int i = 0; i < KY; ++i) s += fred[i];
    a.fpart[(size_t)slot * a.ntile + blockIdx.x] = s;
  }
}

// --------------------------------------------------------------------------------------------------
// The same pass on the FP64 matrix cores (p <= 16).  Per wave one tile of 16 bins; neurons in tiles of 16:
//   H tile   (neuron x bin)  = C16 . X + d           3 MFMAs (K = latents, padded to 4s)
//   e = exp(H), r = e - y, f += e - y*h              element-wise on the accumulator registers
//   W[bin][pair] += sum_n e[n][bin] * CCu[n][pair]   NT MFMAs per 4 neurons (pairs a >= b of C[n][a] C[n][b])
//   G[bin][k]    += sum_n r[n][bin] * C16[n][k]      1 MFMA per 4 neurons
// The accumulator layout of the H tile (lane = bin, register r = neurons 4r..4r+3 over the lane quads) IS the
// A-fragment layout of the following products, so e and r never leave registers.  CCu / C16 are zero-padded
// tables built once per parameter set (poisson_tables_kernel) and read from L2.  W tiles are staged in LDS and
// leave as contiguous runs.  grid = (ceil(T/64), nslots), block = 256 (4 waves x 16 bins).
// --------------------------------------------------------------------------------------------------
inline __global__ void poisson_tables_kernel(const double* __restrict__ C, int q, int p, int qpad, int ncol, double* __restrict__ CCu,
                                      double* __restrict__ C16) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= c) ++a;
    const int b = c - a * (a + 1) / 2;
    CCu[(size_t)n * ncol + c] = (n < q && a < p) ? C[(size_t)n * p + a] * C[(size_t)n * p + b] : 0.0;
  }
  for (int l = threadIdx.x; l < 16; l += blockDim.x) C16[(size_t)n * 16 + l] = (n < q && l < p) ? C[(size_t)n * p + l] : 0.0;
}

// NBT (round 5): bin tiles of 16 per wave.  The table fragments of a neuron tile (23 doubles per lane: CCu, C16) are the same for every wave, slot
// and bin tile and came from L2 once per (wave, 16 bins, 16 neurons): 11.8 KB per 23 matrix instructions, 5 GB of L2 -> CU traffic per launch at
// config 3 for 0.58 GB of HBM bytes - the pass sat under neither roof (0.79 ms).  With NBT tiles per wave a fragment set serves NBT x 16 bins:
// 0.58 ms at NBT = 2.  (One copy of the neuron tile's tables per WORKGROUP in LDS on top of that - 10 KB, double-buffered, one barrier per neuron
// tile, every fragment read conflict-free - ran 0.60 ms: at two tiles per wave the fragments are no longer what bounds the pass.  Dropped.)
// grid = (ceil(T / (64 NBT)), nslots); fpart holds gridDim.x partial sums per slot.
template <int PW, int NBT>
__global__ __launch_bounds__(256, 2) void poisson_mfma_kernel(PoissonArgs a, const double* __restrict__ CCu, const double* __restrict__ C16, int qpad) {
  constexpr int NP = PW * (PW + 1) / 2, NT = (NP + 15) / 16, NC = NT * 16, KS = (PW + 3) / 4;
  __shared__ double Wl[4][16 * PW * PW];
  __shared__ double fred[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int slot = a.slots[blockIdx.y];
  const int trial = a.trial_of_slot[slot];
  const int held_out = a.mask ? a.mask[slot] : -1;
  const int p = a.p, q = a.q, T = a.T, pp = p * p;
  const int sbase0 = blockIdx.x * 64 * NBT + wave * 16 * NBT;       // this wave's bins: NBT consecutive tiles of 16
  const double* X = a.X + (size_t)slot * a.sX;
  const uint8_t* Y = a.Y + (size_t)trial * q * T;
  const uint8_t* Yh = a.Yhi ? a.Yhi + (size_t)trial * q * T : nullptr;

  double xb[NBT][KS];
  int tcl[NBT];
  bool vt[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    const int t = sbase0 + 16 * bt + l15;
    vt[bt] = t < T;
    tcl[bt] = vt[bt] ? t : T - 1;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int l = l4 + 4 * kk;
      xb[bt][kk] = (l < p && vt[bt]) ? X[(size_t)l * T + t] : 0.0;
    }
  }
  // LDS offsets of this lane's pair columns (c = tile*16 + l15 -> (pa, pb), pa >= pb); -1: padding column
  int off1[NT], off2[NT];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    const int c = tl * 16 + l15;
    int pa = 0;
    while ((pa + 1) * (pa + 2) / 2 <= c) ++pa;
    const int pb = c - pa * (pa + 1) / 2;
    off1[tl] = (pa < p) ? pa * p + pb : -1;
    off2[tl] = pb * p + pa;
  }
  mdouble4 accW[NBT][NT], accG[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    accG[bt] = mdouble4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) accW[bt][tl] = mdouble4{0.0, 0.0, 0.0, 0.0};
  }
  double facc = 0.0;

  if (sbase0 < T) {
    // every global load of a neuron tile (offsets, counts, table fragments) is issued up front with clamped,
    // branch-free addresses; invalid rows/bins are masked afterwards
    for (int nb0 = 0; nb0 < qpad; nb0 += 16) {
      double dn[4];
      double dv[NBT][4];
      unsigned yv[NBT][4];
      double ch[KS];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb0 + l4 + 4 * r;
        const int nc = n < q ? n : q - 1;
        dn[r] = a.d[nc];
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {
          dv[bt][r] = dn[r];
          if (a.off) dv[bt][r] += a.off[(size_t)slot * a.sOff + (size_t)nc * T + tcl[bt]];
          yv[bt][r] = count_at(Y, Yh, (size_t)nc * T + tcl[bt]);
        }
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) ch[kk] = C16[(size_t)(nb0 + l15) * 16 + l4 + 4 * kk];
      double bw[4][NT], bg[4];
      if (a.full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t nrow = (size_t)(nb0 + 4 * r + l4);
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) bw[r][tl] = CCu[nrow * NC + tl * 16 + l15];
          bg[r] = C16[nrow * 16 + l15];
        }
      }
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) {
        if (sbase0 + 16 * bt >= T) continue;                    // (uniform over the wave)
        mdouble4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (nb0 + l4 + 4 * r < q) ? dv[bt][r] : 0.0;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) h = __builtin_amdgcn_mfma_f64_16x16x4f64(ch[kk], xb[bt][kk], h, 0, 0, 0);
        double e[4], rr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb0 + l4 + 4 * r;
          const bool ok = (n < q) && vt[bt] && (n != held_out);
          const double y = ok ? (double)yv[bt][r] : 0.0;
          const double ev = ok ? exp(h[r]) : 0.0;
          e[r] = ev;
          rr[r] = ev - y;
          facc += ok ? ev - y * h[r] : 0.0;
          if (a.lam_out && (n < q) && vt[bt]) a.lam_out[(size_t)slot * a.sLam + (size_t)n * T + sbase0 + 16 * bt + l15] = ev;
        }
        if (a.full) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) accW[bt][tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(e[r], bw[r][tl], accW[bt][tl], 0, 0, 0);
            accG[bt] = __builtin_amdgcn_mfma_f64_16x16x4f64(rr[r], bg[r], accG[bt], 0, 0, 0);
          }
        }
      }
    }
  }
  if (a.full) {
    // accumulators: bin = l4 + 4 r, column = l15; the wave's tiles leave one after the other through its LDS tile
    double* Ws = Wl[wave];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      const int sbase = sbase0 + 16 * bt;
      if (sbase < T) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
          if (off1[tl] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Ws[(l4 + 4 * r) * pp + off1[tl]] = accW[bt][tl][r];
              Ws[(l4 + 4 * r) * pp + off2[tl]] = accW[bt][tl][r];
            }
          }
      }
      __syncthreads();
      if (sbase < T) {
        const int nbins = min(16, T - sbase);
        double* W = a.W + (size_t)slot * a.sW + (size_t)sbase * pp;
        for (int e2 = lane; e2 < nbins * pp; e2 += 64) W[e2] = Ws[e2];
        if (l15 < p) {
          double* G = a.G + (size_t)slot * a.sG + (size_t)l15 * T + sbase;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (sbase + l4 + 4 * r < T) G[l4 + 4 * r] = accG[bt][r];
        }
      }
      if (bt + 1 < NBT) __syncthreads();
    }
  }
  for (int off = 32; off > 0; off >>= 1) facc += __shfl_down(facc, off);
  if (lane == 0) fred[wave] = facc;
  __syncthreads();
  if (threadIdx.x == 0) a.fpart[(size_t)slot * a.ntile + blockIdx.x] = (fred[0] + fred[1]) + (fred[2] + fred[3]);
}

// Leave-one-neuron-out prediction (util.leaveOneOutPrediction util.py:328-329): rate of the held-out neuron at the
// mode found without it, yp[slot][t] = exp(c_n . x_t + d_n), and err[slot] = sum_t (y_nt - yp_t)^2.
// grid = nslots, block = 256.
inline __global__ void loo_predict_kernel(const double* __restrict__ X, long long sX, const double* __restrict__ C, const double* __restrict__ d,
                                   const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, const int* __restrict__ trial_of_slot,
                                   const int* __restrict__ mask, int q, int p, int T, double* __restrict__ yp, long long sP, double* __restrict__ err) {
  __shared__ double red[256];
  const int slot = blockIdx.x;
  const int n = mask[slot];
  const double* x = X + (size_t)slot * sX;
  const uint8_t* y = Y + ((size_t)trial_of_slot[slot] * q + n) * T;
  const uint8_t* yh = Yhi ? Yhi + ((size_t)trial_of_slot[slot] * q + n) * T : nullptr;
  double s = 0.0;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    double h = d[n];
    for (int l = 0; l < p; ++l) h += C[(size_t)n * p + l] * x[(size_t)l * T + t];
    const double v = exp(h);
    yp[(size_t)slot * sP + t] = v;
    const double r = (double)count_at(y, yh, t) - v;
    s += r * r;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) err[slot] = red[0];
}

// flik[slot] = sum_tile fpart[slot][tile]
inline __global__ void sum_tiles_kernel(const double* __restrict__ fpart, int ntile, const int* __restrict__ slots, int nslots, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslots) return;
  const int slot = slots[i];
  double s = 0.0;
  for (int b = 0; b < ntile; ++b) s += fpart[(size_t)slot * ntile + b];
  out[slot] = s;
}

// --------------------------------------------------------------------------------------------------
// Prior mat-vec: out[slot][k][t] = sum_s Kinv[k][t][s] * in[slot][k][s]   (K^-1 x, inference.py:27,44)
// Kinv slabs are (Tp x Tp) symmetric; the read runs down a column so lanes (t) are contiguous.
// grid = (p, nslots), block = 256.
// --------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256) void prior_matvec_kernel(const double* __restrict__ Kinv, int Tp, int T, int p,
                                                            const double* __restrict__ in, long long sIn,
                                                            double* __restrict__ out, long long sOut,
                                                            const int* __restrict__ slots) {
  extern __shared__ double xin[];
  const int k = blockIdx.x;
  const int slot = slots[blockIdx.y];
  const double* x = in + (size_t)slot * sIn + (size_t)k * T;
  for (int s = threadIdx.x; s < T; s += 256) xin[s] = x[s];
  __syncthreads();
  const double* Kk = Kinv + (size_t)k * Tp * Tp;
  for (int t = threadIdx.x; t < T; t += 256) {
    double acc = 0.0;
    for (int s = 0; s < T; ++s) acc += Kk[(size_t)s * Tp + t] * xin[s];
    out[(size_t)slot * sOut + (size_t)k * T + t] = acc;
  }
}

// three dot products per slot: r[slot] = {a.b, c.b', c.d'} -> used for x^T K^-1 x, delta^T K^-1 x, delta^T K^-1 delta
inline __global__ __launch_bounds__(256) void dots3_kernel(const double* __restrict__ X, long long sX, const double* __restrict__ KX, long long sKX,
                                                     const double* __restrict__ D, long long sD, const double* __restrict__ KD, long long sKD,
                                                     int n, const int* __restrict__ slots, double* __restrict__ qxx, double* __restrict__ qdx,
                                                     double* __restrict__ qdd) {
  __shared__ double red[3][4];
  const int slot = slots[blockIdx.x];
  const double* x = X + (size_t)slot * sX;
  const double* kx = KX + (size_t)slot * sKX;
  const double* dd = D ? D + (size_t)slot * sD : nullptr;
  const double* kd = KD ? KD + (size_t)slot * sKD : nullptr;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    s0 += x[i] * kx[i];
    if (dd) { s1 += dd[i] * kx[i]; s2 += dd[i] * kd[i]; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    qxx[slot] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    if (dd) {
      qdx[slot] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      qdd[slot] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
  }
}

// total gradient for the solve: Gt[slot][i] = Gl[slot][i] + KX[slot][i]
inline __global__ void grad_total_kernel(const double* __restrict__ Gl, long long sG, const double* __restrict__ KX, long long sKX,
                                  double* __restrict__ Gt, long long sGt, int n, const int* __restrict__ slots) {
  const int slot = slots[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Gt[(size_t)slot * sGt + i] = Gl[(size_t)slot * sG + i] + KX[(size_t)slot * sKX + i];
}

// Xt = X + alpha[slot] * delta
inline __global__ void make_try_kernel(const double* __restrict__ X, long long sX, const double* __restrict__ D, long long sD,
                                const double* __restrict__ alpha, double* __restrict__ Xt, long long sXt, int n,
                                const int* __restrict__ slots) {
  const int slot = slots[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Xt[(size_t)slot * sXt + i] = X[(size_t)slot * sX + i] + alpha[slot] * D[(size_t)slot * sD + i];
}

// accepted slots: X <- Xt ; KX <- KX + alpha*KD ; Gl <- Glt ; W <- Wt
inline __global__ void commit_kernel(double* __restrict__ X, const double* __restrict__ Xt, double* __restrict__ KX,
                              const double* __restrict__ KD, double* __restrict__ Gl, const double* __restrict__ Glt,
                              long long sV, double* __restrict__ W, const double* __restrict__ Wt, long long sW,
                              const double* __restrict__ alpha, int n, int nw, const int* __restrict__ slots) {
  const size_t slot = slots[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    X[slot * sV + i] = Xt[slot * sV + i];
    KX[slot * sV + i] += alpha[slot] * KD[slot * sV + i];
    Gl[slot * sV + i] = Glt[slot * sV + i];
  }
  for (int j = i; j < nw; j += gridDim.x * blockDim.x) W[slot * sW + j] = Wt[slot * sW + j];
}

// --------------------------------------------------------------------------------------------------
// Hessian assembly (inference.py:50-65, structured): lower triangle of
//   H[(k,t),(l,s)] = [k==l] Kinv[k][t][s] + [t==s] W[t][k][l]     (latent-major index i = k*T + t)
// plus identity padding for rows/cols >= n.  grid = (npad, nslots): one block per column.
// --------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256) void assemble_h_kernel(double* __restrict__ H, long long sH, int ld, int npad, int n, int T, int Tp, int p,
                                                          const double* __restrict__ Kinv, const double* __restrict__ W, long long sW,
                                                          const int* __restrict__ slots, double diag_scale) {
  const int j = blockIdx.x;
  const size_t slot = slots[blockIdx.y];
  double* col = H + slot * sH + (size_t)j * ld;
  if (j >= n) {
    for (int i = j + threadIdx.x; i < npad; i += 256) col[i] = (i == j) ? 1.0 : 0.0;
    return;
  }
  const int kj = j / T, tj = j - kj * T;
  const double* Kcol = Kinv + (size_t)kj * Tp * Tp + (size_t)tj * Tp;   // Kinv[kj][:, tj] (symmetric)
  const double* Wt = W + slot * sW + (size_t)tj * p * p;
  for (int i = j + threadIdx.x; i < npad; i += 256) {
    double v = 0.0;
    if (i < n) {
      const int ki = i / T, ti = i - ki * T;
      if (ki == kj) v = Kcol[ti];
      if (ti == tj) v += Wt[ki * p + kj];
      if (i == j) v *= diag_scale;        // VIPostCov jitter (inference.py:190); 1.0 for Laplace
    }
    col[i] = v;
  }
}

// dense symmetric H (n x n, row-major == column-major) for one slot -> host-visible buffer (a6 getter)
inline __global__ void dense_h_kernel(double* __restrict__ out, int n, int T, int Tp, int p, const double* __restrict__ Kinv,
                               const double* __restrict__ W) {
  const int j = blockIdx.x;
  const int kj = j / T, tj = j - kj * T;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int ki = i / T, ti = i - ki * T;
    double v = 0.0;
    if (ki == kj) v = Kinv[(size_t)kj * Tp * Tp + (size_t)tj * Tp + ti];
    if (ti == tj) v += W[(size_t)tj * p * p + ki * p + kj];
    out[(size_t)j * n + i] = v;
  }
}

// --------------------------------------------------------------------------------------------------
// post_vsm (inference.py:169-172): vsm[t][k][l] = Sigma[(k,t),(l,t)] = sum_i Mt[(k,t), i] Mt[(l,t), i]
// with Mt = L^-T (upper triangular).  Block = 64 bins x KY latents (KY = post_vsm_rows(p) of pgpfa.hip; two rows per thread
// beyond 16 latents), columns i streamed in chunks through LDS; output indexed by trial.
// --------------------------------------------------------------------------------------------------
template <int PMAX, typename TIN = double>
__global__ __launch_bounds__(PMAX == 24 ? 768 : 1024) void post_vsm_kernel(const TIN* __restrict__ Mt, long long sM, int ld, int npad, int T, int p,
                                double* __restrict__ vsm, const int* __restrict__ slots, const int* __restrict__ trial_of_slot,
                                int full_range, int ts) {      // ts: row stride between latents in the panel (T, or the padded stride of the low-rank slab)
  constexpr int NK = (PMAX > 16) ? 2 : 1;
  constexpr int VIC = (PMAX <= 8) ? 16 : (PMAX <= 16 ? 8 : 4);   // 64 KB of LDS at most (49 KB at PMAX = 24)
  __shared__ double A[VIC][PMAX][64];
  const int tx = threadIdx.x, ty = threadIdx.y, KY = blockDim.y;
  const int slot = slots[blockIdx.y];
  const int trial = trial_of_slot[slot];
  const int t0 = blockIdx.x * 64;
  const int t = t0 + tx;
  const bool valid = t < T;
  const TIN* M = Mt + (size_t)slot * sM;
  double acc[NK][PMAX];
#pragma unroll
  for (int j = 0; j < NK; ++j)
#pragma unroll
    for (int l = 0; l < PMAX; ++l) acc[j][l] = 0.0;

  const int istart = full_range ? 0 : (t0 / VIC) * VIC;   // triangular Mt: row (0,t0) is the first with entries at column t0
  for (int i0 = istart; i0 < npad; i0 += VIC) {
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int k = ty + j * KY;
      if (k < p) {
        const int row = k * ts + t;
#pragma unroll
        for (int ii = 0; ii < VIC; ++ii) A[ii][k][tx] = valid ? (double)M[(size_t)(i0 + ii) * ld + row] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int k = ty + j * KY;
      if (k < p) {
#pragma unroll
        for (int ii = 0; ii < VIC; ++ii) {
          const double ak = A[ii][k][tx];
#pragma unroll
          for (int l = 0; l < PMAX; ++l)
            if (l <= k) acc[j][l] += ak * A[ii][l][tx];
        }
      }
    }
    __syncthreads();
  }
  if (valid) {
    double* out = vsm + ((size_t)trial * T + t) * p * p;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int k = ty + j * KY;
      if (k < p) {
#pragma unroll
        for (int l = 0; l < PMAX; ++l)
          if (l <= k) { out[k * p + l] = acc[j][l]; out[l * p + k] = acc[j][l]; }
      }
    }
  }
}

// The same on the FP64 matrix cores for 10 < p <= 32 (the dual evaluation at config 5's 20 latents spent a quarter of its time in the
// vector form above: one LDS read per multiply-add).  Per bin the block is the Gram matrix of the p x ncol panel Y_t = rows (., t) of Mt:
// with a = b = Y_t[lane & 15][b0 + (lane >> 4)] one v_mfma_f64_16x16x4 adds four columns to a 16 x 16 tile - the A and B fragments of
// a Gram product are the same register - so p <= 16 takes one LDS read and one MFMA per four columns, p <= 32 two reads and three
// MFMAs (tiles (0,0), (1,0), (1,1) of the lower triangle).  A block owns 32 bins (128-byte runs of the single-precision panel) of one
// slot, 8 waves x 4 bins; column chunks of CB are staged in LDS as doubles ([column][latent][bin], bin stride 33), the next chunk's
// loads in flight in registers during the products.  NRT = row tiles (1 or 2).
// grid = (ceil(T/32), nslots), block = 512, dynamic LDS = CB * p * 33 elements of the panel's type with CB = post_vsm_mfma_cb(p, f32).
// (the chunk is staged in the panel's own precision and widened when a fragment is read: a single-precision panel takes chunks 1.5x as deep -
// 12 columns, 32 KB of LDS at 20 latents; 16 would put the staging map and prefetch registers into scratch)
inline int post_vsm_mfma_cb(int p, bool f32) { return p <= 24 ? (f32 ? 12 : 8) : (f32 ? 8 : 4); }
template <int NRT, typename TIN>
__global__ __launch_bounds__(512) void post_vsm_mfma_kernel(const TIN* __restrict__ Mt, long long sM, int ld, int ncol, int T, int p,
                                                            double* __restrict__ vsm, const int* __restrict__ slots,
                                                            const int* __restrict__ trial_of_slot, int full_range, int CB,
                                                            const int* __restrict__ roff, int col_tile, int ts) {
  // roff (may be null): rank offsets of the latents; rows (k, .) of the panel are identically zero - and were not written - left of
  // column (roff[k] / col_tile) * col_tile (see the Yt product of the low-rank engine)
  constexpr int LT = 33, MAXPF = sizeof(TIN) == 4 ? 18 : 12, NACC = NRT == 1 ? 1 : 3;   // MAXPF >= CB p / 16: (12, 24) or (8, 32) in FP32
  extern __shared__ double sm_raw[];
  TIN* sm = reinterpret_cast<TIN*>(sm_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int slot = slots[blockIdx.y];
  const int trial = trial_of_slot[slot];
  const int t0 = blockIdx.x * 32;
  const int nt = min(32, T - t0);
  const TIN* M = Mt + (size_t)slot * sM + t0;
  const int per_chunk = CB * p * 32;
  // staging map, the same for every chunk: element e = tid + 512 j -> (column b, latent k, bin tt), packed as b | k << 8 | first written
  // column tile of latent k << 16 (-1: no element)
  const int tt = tid & 31;
  const int ttc = tt < nt ? tt : nt - 1;
  int map[MAXPF];
#pragma unroll
  for (int j = 0; j < MAXPF; ++j) {
    const int e = tid + 512 * j;
    const int row = min(e, per_chunk - 1) >> 5;               // b * p + k
    const int b = row / p, k = row - b * p;
    map[j] = (e < per_chunk) ? (b | (k << 8) | ((roff ? roff[k] / col_tile : 0) << 16)) : -1;
  }
  // (an element outside the written part of the panel is read from one fixed written location instead: the loads stay
  // unconditional - issued back to back - and the unwritten columns cost no traffic)
  const size_t off_dummy = (size_t)(ncol - 1) * ld + (size_t)(p - 1) * ts;
  TIN pf[MAXPF];
  auto issue = [&](int i0) {
#pragma unroll
    for (int j = 0; j < MAXPF; ++j) {
      const int mj = map[j] < 0 ? 0 : map[j];
      const int col = i0 + (mj & 255), k = (mj >> 8) & 255, c0 = (mj >> 16) * col_tile;
      const bool in = col < ncol && col >= c0;
      pf[j] = M[in ? (size_t)col * ld + (size_t)k * ts + ttc : off_dummy];
    }
  };
  auto commit = [&](int i0) {
#pragma unroll
    for (int j = 0; j < MAXPF; ++j) {
      const int mj = map[j];
      if (mj >= 0) {
        const int b = mj & 255, k = (mj >> 8) & 255, col = i0 + b, c0 = (mj >> 16) * col_tile;
        sm[(b * p + k) * LT + tt] = (tt < nt && col < ncol && col >= c0) ? pf[j] : (TIN)0;
      }
    }
  };
  double4_t acc[4][NACC];
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[bb][a] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int li = lane & 15, l4 = lane >> 4;
  const bool r0 = li < p, r1 = 16 + li < p;
  const int istart = full_range ? 0 : (t0 / CB) * CB;         // triangular Mt: rows (., t0..) vanish left of column t0
  const int c0_second = (roff && p > 16) ? (roff[16] / col_tile) * col_tile : 0;
  if (istart < ncol) issue(istart);
  for (int i0 = istart; i0 < ncol; i0 += CB) {
    __syncthreads();
    commit(i0);
    __syncthreads();
    if (i0 + CB < ncol) issue(i0 + CB);
    const bool second = NRT == 2 && i0 + CB > c0_second;      // (uniform) the second row tile is identically zero left of its first column
    for (int ks = 0; ks < CB; ks += 4) {
      const TIN* base = sm + (size_t)(ks + l4) * p * LT + wave * 4;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const double y0 = r0 ? (double)base[li * LT + bb] : 0.0;
        acc[bb][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, y0, acc[bb][0], 0, 0, 0);
        if constexpr (NRT == 2) {
          if (second) {
            const double y1 = r1 ? (double)base[(16 + li) * LT + bb] : 0.0;
            acc[bb][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, y0, acc[bb][1], 0, 0, 0);
            acc[bb][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, y1, acc[bb][2], 0, 0, 0);
          }
        }
      }
    }
  }
  // D[row = (lane >> 4) + 4 r][col = lane & 15] of every tile -> vsm[t][k][l]
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    const int t = t0 + wave * 4 + bb;
    if (t >= T) continue;
    double* out = vsm + ((size_t)trial * T + t) * p * p;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = l4 + 4 * r;
      if (row < p && li < p) out[row * p + li] = acc[bb][0][r];
      if constexpr (NRT == 2) {
        if (16 + row < p && li < p) {
          out[(16 + row) * p + li] = acc[bb][1][r];
          out[li * p + 16 + row] = acc[bb][1][r];
        }
        if (16 + row < p && 16 + li < p) out[(16 + row) * p + 16 + li] = acc[bb][2][r];
      }
    }
  }
}

// The same Gram matrices on v_mfma_f64_4x4x4_4b for 10 < p <= 20 (round 5).  The 16 x 16 x 4 shape pads 20 latents to two 16-row tiles: three
// instructions of 68 cycles per bin and four columns, 27 % of whose products are wanted.  The 4 x 4 x 4 shape runs FOUR independent 4 x 4 x 4
// products in 19 cycles (tools/probes/mfma_4x4x4_probe.hip): here the four blocks of an instruction are the wave's four BINS and the instruction
// index runs over the lower pairs (I >= J) of NBK = ceil(p / 4) blocks of four latents - operand lane (k = lane >> 4, bin = (lane >> 2) & 3,
// latent = 4 I + (lane & 3)), the A fragment of pair (I, J) is block I's register and the B fragment block J's: NBK LDS reads and
// NBK (NBK + 1) / 2 instructions per four bins and four columns (5 and 15 x 19 = 285 cycles at 20 latents, against 12 x 68 = 816), result
// lane (row = lane >> 4, bin, column = lane & 3).  A pair whose block I has no written column yet (rows of latent k vanish left of roff[k]) is
// not issued.  Staging, chunking and launch shape are post_vsm_mfma_kernel's; the LDS image has latent stride 36 and column stride 36 p + 8
// words so that the 64 lanes of a fragment read (4 columns x 4 latents x 4 bins) fall on 64 different words of the 32 x 2 banks.
// VEC: 64 bins per workgroup (two quads of bins per wave) staged with 16-byte loads - 256-byte runs of the panel instead of 128-byte ones (the
// pass reads the whole panel once: 45 GB per pass at config 5, and ran at 1.3 TB/s on 128-byte runs); needs T, ts, ld and the slab stride
// multiples of the vector width.
// dynamic LDS = CB * post_vsm_b4_cs(p, VEC) elements of the panel's type.
__host__ __device__ inline int post_vsm_b4_cs(int p, bool vec) { return (vec ? 68 : 36) * p + 8; }
template <int NBK, typename TIN, bool VEC>
__global__ __launch_bounds__(512) void post_vsm_b4_kernel(const TIN* __restrict__ Mt, long long sM, int ld, int ncol, int T, int p,
                                                          double* __restrict__ vsm, const int* __restrict__ slots,
                                                          const int* __restrict__ trial_of_slot, int full_range, int CB,
                                                          const int* __restrict__ roff, int col_tile, int ts) {
  constexpr int BINS = VEC ? 64 : 32, NQ = BINS / 32, LT = VEC ? 68 : 36, NPR = NBK * (NBK + 1) / 2;
  constexpr int VW = VEC ? 16 / (int)sizeof(TIN) : 1, UPR = BINS / VW;                     // elements per load, loads per (column, latent) row
  constexpr int MAXPF = VEC ? (sizeof(TIN) == 4 ? 8 : 10) : (sizeof(TIN) == 4 ? 18 : 12);  // >= CB p UPR / 512
  typedef TIN vec_t __attribute__((ext_vector_type(VW)));
  extern __shared__ double sm_raw[];
  TIN* sm = reinterpret_cast<TIN*>(sm_raw);
  const int CS = post_vsm_b4_cs(p, VEC);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int slot = slots[blockIdx.y];
  const int trial = trial_of_slot[slot];
  const int t0 = blockIdx.x * BINS;
  const int nt = min(BINS, T - t0);
  const TIN* M = Mt + (size_t)slot * sM + t0;
  const int per_chunk = CB * p * UPR;
  // staging map, the same for every chunk: load e = tid + 512 j -> (column b, latent k), its bins u0 .. u0 + VW - 1 (512 is a multiple of UPR: u0 is
  // the thread's own), packed as b | k << 8 | first written column tile of latent k << 16 (-1: no load)
  const int u0 = (tid % UPR) * VW;
  const int uc = VEC ? u0 : (u0 < nt ? u0 : nt - 1);          // (scalar form: bins past T read the last one; vector form: a load is all in or all out)
  const bool uin = u0 < nt;
  int map[MAXPF];
#pragma unroll
  for (int j = 0; j < MAXPF; ++j) {
    const int e = tid + 512 * j;
    const int row = min(e, per_chunk - 1) / UPR;              // b * p + k
    const int b = row / p, k = row - b * p;
    map[j] = (e < per_chunk) ? (b | (k << 8) | ((roff ? roff[k] / col_tile : 0) << 16)) : -1;
  }
  // (a load outside the written part of the panel reads one fixed written location instead: the loads stay unconditional - issued back to
  // back - and the unwritten columns cost no traffic)
  const size_t off_dummy = VEC ? 0 : (size_t)(ncol - 1) * ld + (size_t)(p - 1) * ts;
  vec_t pf[MAXPF];
  auto issue = [&](int i0) {
#pragma unroll
    for (int j = 0; j < MAXPF; ++j) {
      const int mj = map[j] < 0 ? 0 : map[j];
      const int col = i0 + (mj & 255), k = (mj >> 8) & 255, c0 = (mj >> 16) * col_tile;
      const bool in = col < ncol && col >= c0 && (!VEC || uin);
      pf[j] = *reinterpret_cast<const vec_t*>(M + (in ? (size_t)col * ld + (size_t)k * ts + uc : off_dummy));
    }
  };
  auto commit = [&](int i0) {
#pragma unroll
    for (int j = 0; j < MAXPF; ++j) {
      const int mj = map[j];
      if (mj >= 0) {
        const int b = mj & 255, k = (mj >> 8) & 255, col = i0 + b, c0 = (mj >> 16) * col_tile;
        vec_t z;
#pragma unroll
        for (int x = 0; x < VW; ++x) z[x] = (TIN)0;
        *reinterpret_cast<vec_t*>(sm + b * CS + k * LT + u0) = (uin && col < ncol && col >= c0) ? pf[j] : z;
      }
    }
  };
  double acc[NQ][NPR];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int a = 0; a < NPR; ++a) acc[q][a] = 0.0;
  const int l4 = lane >> 4, blk = (lane >> 2) & 3, x4 = lane & 3;
  // first column in which block I has something written (uniform)
  int cfirst[NBK];
#pragma unroll
  for (int I = 0; I < NBK; ++I) cfirst[I] = (roff && 4 * I < p) ? (roff[4 * I] / col_tile) * col_tile : 0;
  const int istart = full_range ? 0 : (t0 / CB) * CB;         // triangular Mt: rows (., t0..) vanish left of column t0
  if (istart < ncol) issue(istart);
  for (int i0 = istart; i0 < ncol; i0 += CB) {
    __syncthreads();
    commit(i0);
    __syncthreads();
    if (i0 + CB < ncol) issue(i0 + CB);
    for (int ks = 0; ks < CB; ks += 4) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const TIN* base = sm + (size_t)(ks + l4) * CS + q * 32 + wave * 4 + blk;
        double v[NBK];
#pragma unroll
        for (int I = 0; I < NBK; ++I) v[I] = (4 * I + x4 < p) ? (double)base[(4 * I + x4) * LT] : 0.0;
        int pr = 0;
#pragma unroll
        for (int I = 0; I < NBK; ++I) {
          const bool on = i0 + CB > cfirst[I];
#pragma unroll
          for (int J = 0; J <= I; ++J) {
            if (on) acc[q][pr] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[I], v[J], acc[q][pr], 0, 0, 0);
            ++pr;
          }
        }
      }
    }
  }
  // lane (l4, blk, x4) of pair (I, J): entry (4 I + l4, 4 J + x4) of bin q * 32 + wave * 4 + blk
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int t = t0 + q * 32 + wave * 4 + blk;
    if (t >= T) continue;
    double* out = vsm + ((size_t)trial * T + t) * p * p;
    int pr = 0;
#pragma unroll
    for (int I = 0; I < NBK; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) {
        const int r = 4 * I + l4, cc = 4 * J + x4;
        if (r < p && cc < p) {
          out[r * p + cc] = acc[q][pr];
          if (I != J) out[cc * p + r] = acc[q][pr];
        }
        ++pr;
      }
  }
}

// (T,T,p) reference layout of post_vsmGP (inference.py:164-167) from the device layout [p][T][T]
inline __global__ void vsmgp_to_ref_kernel(const double* __restrict__ src, double* __restrict__ dst, int T, int p) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)T * T * p;
  if (e >= tot) return;
  const int k = e % p;
  const size_t ab = e / p;          // a*T + b
  dst[e] = src[(size_t)k * T * T + ab];
}
inline __global__ void vsmgp_from_ref_kernel(const double* __restrict__ src, double* __restrict__ dst, int T, int p) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)T * T * p;
  if (e >= tot) return;
  const int k = e % p;
  const size_t ab = e / p;
  dst[(size_t)k * T * T + ab] = src[e];
}

// learning.makePrecomp learning.py:162-166: PautoSum[k] = sum_r (Sigma_r^{kk} + m_rk m_rk^T), into (Tp x Tp) slabs
inline __global__ void pautosum_kernel(const double* __restrict__ vsmgp, const double* __restrict__ mean, const int* __restrict__ trials,
                                int ntr, int T, int Tp, int p, double* __restrict__ P) {
  const int k = blockIdx.y;
  const int b = blockIdx.x;               // column
  for (int a = threadIdx.x; a < Tp; a += blockDim.x) {
    double s = 0.0;
    if (a < T && b < T) {
      for (int i = 0; i < ntr; ++i) {
        const size_t r = trials[i];
        const double* m = mean + (r * p + k) * T;
        s += vsmgp[((r * p + k) * T + b) * T + a] + m[a] * m[b];
      }
    }
    P[(size_t)k * Tp * Tp + (size_t)b * Tp + a] = s;
  }
}

// Sum-only form of the above for the low-rank engine (no per-trial T x T blocks are stored): the E-step accumulates
//   Pacc[k] += sum_slots Ymix_k Ymix_k^T + eps * diag(sum_slots G_t[k][k])
// from split-K partial products part[k][split] (lower triangle valid, ld = T); grid = (T, p), block = 128
inline __global__ void pacc_reduce_kernel(const double* __restrict__ part, int nsplit, const double* __restrict__ G, long long sG, int nslots,
                                   double eps, int T, int Tp, int p, double* __restrict__ Pacc) {
  const int k = blockIdx.y, b = blockIdx.x;
  double* P = Pacc + (size_t)k * Tp * Tp;
  const double* src = part + (size_t)k * nsplit * T * T + (size_t)b * T;
  // diagonal term sum_slots G_b[k][k]: the slots are spread over the block (one thread walking ~1000 slabs in turn set the time of the
  // whole launch), summed in a fixed order: lanes, then waves
  __shared__ double gred[2];
  double g = 0.0;
  for (int sl = threadIdx.x; sl < nslots; sl += blockDim.x) g += G[(size_t)sl * sG + (size_t)b * p * p + (size_t)k * p + k];
  for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off);
  if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = g;
  __syncthreads();
  const double gdiag = gred[0] + gred[1];
  for (int a = b + threadIdx.x; a < T; a += blockDim.x) {
    double s = 0.0;
    for (int i = 0; i < nsplit; ++i) s += src[(size_t)i * T * T + a];
    if (a == b) s += eps * gdiag;
    P[(size_t)b * Tp + a] += s;
    if (a != b) P[(size_t)a * Tp + b] += s;
  }
}

// PautoSum[k] = Pacc[k] + sum_r m_rk m_rk^T  (same output layout as pautosum_kernel); grid = (Tp, p)
inline __global__ void pauto_from_acc_kernel(const double* __restrict__ Pacc, const double* __restrict__ mean, const int* __restrict__ trials,
                                      int ntr, int T, int Tp, int p, double* __restrict__ P) {
  const int k = blockIdx.y;
  const int b = blockIdx.x;
  for (int a = threadIdx.x; a < Tp; a += blockDim.x) {
    double s = 0.0;
    if (a < T && b < T) {
      s = Pacc[(size_t)k * Tp * Tp + (size_t)b * Tp + a];
      for (int i = 0; i < ntr; ++i) {
        const double* m = mean + ((size_t)trials[i] * p + k) * T;
        s += m[a] * m[b];
      }
    }
    P[(size_t)k * Tp * Tp + (size_t)b * Tp + a] = s;
  }
}

// --------------------------------------------------------------------------------------------------
// (C,d) M-step pass, learning.MStepObservationCost(_grad) learning.py:20-91:
//   hh = c_n.m_t + d_n ; u = V_t c_n ; rho = c_n.u ; yhat = exp(hh + rho/2)
//   cost_n = sum (y*hh - yhat) ; dd_n = sum (y - yhat) ; dC_n = sum (y - yhat) m_t - yhat u
// Lanes are NEURONS: every per-neuron sum stays in one thread's registers (no cross-lane reduction).
// A block owns 64 neurons and walks (trial, bin-tile) items; per item the V_t blocks, the means and the
// packed counts of the tile are staged in LDS with coalesced loads (V_t re-laid out to a PW x PW stride
// and zero padded, so the inner loops are fully unrolled with compile-time offsets), then wave ty handles
// bins ty, ty+8, ... of the tile reading V_t / m_t as LDS broadcasts.
// --------------------------------------------------------------------------------------------------
struct CdArgs {
  const uint8_t* Y; const uint8_t* Yhi;     // counts [R][q][T] (+ plane of high bytes, NULL when none: count_at)
  const double* mean; const double* vsm; const double* vec;   // vecCd
  const int* trials; int ntr;
  double* part;          // [gridDim.y][p+2][q]
  int q, p, T;
  int dbg;               // timing experiments only (option cd_debug): bit 0 no exp, bit 1 no second product, bit 2 no first product, bit 3 no staging
};
constexpr int CD_KY = 8;
// waves per workgroup of mstep_cd_kernel: 8 up to 16 latents (128-register budget), 4 beyond (c, acc, u of 20+ doubles each need the 256 budget)
template <int PW> struct CdKy { static constexpr int v = (PW <= 16) ? 8 : 4; };
template <int PW> struct CdTile { static constexpr int TT = (PW <= 10) ? 64 : (PW <= 16) ? 32 : (PW <= 20) ? 16 : 8; };

template <int PW>
__global__ __launch_bounds__(64 * CdKy<PW>::v) void mstep_cd_kernel(CdArgs a) {
  constexpr int KYW = CdKy<PW>::v;
  constexpr int TT = CdTile<PW>::TT;
  constexpr int YS = TT + 4;                      // byte row stride of the count tile (bank spread)
  __shared__ __attribute__((aligned(16))) double Vt[TT][PW * PW];
  __shared__ double Mt[PW][TT];
  __shared__ uint16_t Yt[64 * YS];
  const int lane = threadIdx.x;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = ty * 64 + lane;
  const int n0 = blockIdx.x * 64;
  const int n = n0 + lane;
  const bool live = n < a.q;
  const int p = a.p, q = a.q, T = a.T;
  double c[PW], acc[PW];
#pragma unroll
  for (int l = 0; l < PW; ++l) {
    c[l] = (live && l < p) ? a.vec[(size_t)l * q + n] : 0.0;
    acc[l] = 0.0;
  }
  const double dn = live ? a.vec[(size_t)p * q + n] : 0.0;
  double cost = 0.0, dd = 0.0;

  const int ntt = (T + TT - 1) / TT;
  const int nitems = a.ntr * ntt;
  for (int item = blockIdx.y; item < nitems; item += gridDim.y) {
    const size_t r = a.trials[item / ntt];
    const int t0 = (item % ntt) * TT;
    const int tn = (T - t0 < TT) ? T - t0 : TT;
    const double* mean = a.mean + r * p * T;
    const double* vsm = a.vsm + (r * T + t0) * p * p;
    const uint8_t* Y = a.Y + r * q * T;
    const uint8_t* Yh = a.Yhi ? a.Yhi + r * q * T : nullptr;
    __syncthreads();                               // previous tile fully consumed
    for (int e = tid; e < TT * PW * PW; e += 64 * KYW) {
      const int t = e / (PW * PW), kl = e - t * (PW * PW);
      const int k = kl / PW, l = kl - k * PW;
      Vt[t][kl] = (t < tn && k < p && l < p) ? vsm[(size_t)t * p * p + k * p + l] : 0.0;
    }
    for (int e = tid; e < PW * TT; e += 64 * KYW) {
      const int k = e / TT, t = e - k * TT;
      Mt[k][t] = (k < p && t < tn) ? mean[(size_t)k * T + t0 + t] : 0.0;
    }
    for (int e = tid; e < 64 * TT; e += 64 * KYW) {
      const int nn = e / TT, t = e - nn * TT;
      Yt[nn * YS + t] = (n0 + nn < q && t < tn) ? (uint16_t)count_at(Y, Yh, (size_t)(n0 + nn) * T + t0 + t) : (uint16_t)0;
    }
    __syncthreads();
    for (int t = ty; t < tn; t += KYW) {
      // u = V_t c from the lower triangle only (V_t is symmetric): p(p+1)/2 LDS reads instead of p^2
      double u[PW];
      double hh = dn, rho = 0.0;
#pragma unroll
      for (int k = 0; k < PW; ++k) u[k] = 0.0;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
#pragma unroll
        for (int l = 0; l < k; ++l) {
          const double v = Vt[t][k * PW + l];
          u[k] += v * c[l];
          u[l] += v * c[k];
        }
        u[k] += Vt[t][k * PW + k] * c[k];
      }
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        rho += c[k] * u[k];
        hh += c[k] * Mt[k][t];
      }
      const double yh = exp(hh + 0.5 * rho);
      const double y = (double)Yt[lane * YS + t];
      const double rs = y - yh;
      if (live) {
        cost += y * hh - yh;
        dd += rs;
#pragma unroll
        for (int k = 0; k < PW; ++k) acc[k] += rs * Mt[k][t] - yh * u[k];
      }
    }
  }
  // fixed-order combine over the KY waves of the block, one output row at a time
  __shared__ double red[KYW][64];
  double* part = a.part + (size_t)blockIdx.y * (p + 2) * q;
#pragma unroll
  for (int k = 0; k < PW + 2; ++k) {
    const double v = (k < PW) ? acc[k < PW ? k : 0] : (k == PW ? dd : cost);
    __syncthreads();
    red[ty][lane] = v;
    __syncthreads();
    if (ty == (k % KYW) && live && (k >= PW || k < p)) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < KYW; ++w) s += red[w][lane];
      const int row = (k < PW) ? k : p + (k - PW);
      part[(size_t)row * q + n] = s;
    }
  }
}

// --------------------------------------------------------------------------------------------------
// (C,d) M-step pass WITH per-neuron Hessians for the device Newton solver.  The cost of learning.py:20-49
// is separable over neurons: q independent convex problems in theta_n = (c_n, d_n), dimension p+1.
// With w = m_t + V_t c_n and yhat as above (per trial and bin, before the 1/R factor):
//   grad_c += -(y m - yhat w) ; grad_d += -(y - yhat)
//   H_cc  += yhat (w w^T + V_t) ; H_cd += yhat w ; H_dd += yhat
// Same tiling as mstep_cd_kernel; per lane 1 + (p+1) + (p+1)(p+2)/2 accumulators.
// part layout per block: [NH][q], NH = 1 + (p+1) + (p+1)(p+2)/2 : cost | grad (c.., d) | packed lower Hessian
// --------------------------------------------------------------------------------------------------
constexpr int CDH_KY = 4;      // 4 waves per block: the whole 512-register file per lane for the (p+1)(p+2)/2 Hessian accumulators
template <int PW>
__global__ __launch_bounds__(64 * CDH_KY) void mstep_cd_hess_kernel(CdArgs a) {
  constexpr int TT = CdTile<PW>::TT;
  constexpr int YS = TT + 4;
  constexpr int D = PW + 1;
  constexpr int NHW = D * (D + 1) / 2;
  __shared__ __attribute__((aligned(16))) double Vt[TT][PW * PW];
  __shared__ double Mt[PW][TT];
  __shared__ uint16_t Yt[64 * YS];
  const int lane = threadIdx.x;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = ty * 64 + lane;
  const int n0 = blockIdx.x * 64;
  const int n = n0 + lane;
  const bool live = n < a.q;
  const int p = a.p, q = a.q, T = a.T;
  double c[PW], gacc[D], hacc[NHW];
#pragma unroll
  for (int l = 0; l < PW; ++l) c[l] = (live && l < p) ? a.vec[(size_t)l * q + n] : 0.0;
#pragma unroll
  for (int l = 0; l < D; ++l) gacc[l] = 0.0;
#pragma unroll
  for (int l = 0; l < NHW; ++l) hacc[l] = 0.0;
  const double dn = live ? a.vec[(size_t)p * q + n] : 0.0;
  double cost = 0.0;

  const int ntt = (T + TT - 1) / TT;
  const int nitems = a.ntr * ntt;
  for (int item = blockIdx.y; item < nitems; item += gridDim.y) {
    const size_t r = a.trials[item / ntt];
    const int t0 = (item % ntt) * TT;
    const int tn = (T - t0 < TT) ? T - t0 : TT;
    const double* mean = a.mean + r * p * T;
    const double* vsm = a.vsm + (r * T + t0) * p * p;
    const uint8_t* Y = a.Y + r * q * T;
    const uint8_t* Yh = a.Yhi ? a.Yhi + r * q * T : nullptr;
    __syncthreads();
    for (int e = tid; e < TT * PW * PW; e += 64 * CDH_KY) {
      const int t = e / (PW * PW), kl = e - t * (PW * PW);
      const int k = kl / PW, l = kl - k * PW;
      Vt[t][kl] = (t < tn && k < p && l < p) ? vsm[(size_t)t * p * p + k * p + l] : 0.0;
    }
    for (int e = tid; e < PW * TT; e += 64 * CDH_KY) {
      const int k = e / TT, t = e - k * TT;
      Mt[k][t] = (k < p && t < tn) ? mean[(size_t)k * T + t0 + t] : 0.0;
    }
    for (int e = tid; e < 64 * TT; e += 64 * CDH_KY) {
      const int nn = e / TT, t = e - nn * TT;
      Yt[nn * YS + t] = (n0 + nn < q && t < tn) ? (uint16_t)count_at(Y, Yh, (size_t)(n0 + nn) * T + t0 + t) : (uint16_t)0;
    }
    __syncthreads();
    for (int t = ty; t < tn; t += CDH_KY) {
      double w[D];
      double hh = dn, rho = 0.0;
      // u = V_t c from the lower triangle only (V_t symmetric), accumulated in w
#pragma unroll
      for (int k = 0; k < PW; ++k) w[k] = 0.0;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
#pragma unroll
        for (int l = 0; l < k; ++l) {
          const double v = Vt[t][k * PW + l];
          w[k] += v * c[l];
          w[l] += v * c[k];
        }
        w[k] += Vt[t][k * PW + k] * c[k];
      }
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        rho += c[k] * w[k];
        const double mk = Mt[k][t];
        hh += c[k] * mk;
        w[k] += mk;
      }
      w[PW] = 1.0;
      const double yh = exp(hh + 0.5 * rho);
      const double y = (double)Yt[lane * YS + t];
      if (live) {
        cost += y * hh - yh;
#pragma unroll
        for (int k = 0; k < PW; ++k) gacc[k] -= y * Mt[k][t] - yh * w[k];
        gacc[PW] -= y - yh;
        int idx = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const double ywi = yh * w[i];
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            double add = ywi * w[j];
            if (i < PW) add += yh * Vt[t][i * PW + j];
            hacc[idx] += add;
            ++idx;
          }
        }
      }
    }
  }
  // fixed-order combine over the KY waves, one output row at a time
  __shared__ double red[CDH_KY][64];
  const int NH = 1 + (p + 1) + (p + 1) * (p + 2) / 2;
  double* part = a.part + (size_t)blockIdx.y * NH * q;
  auto emit = [&](double v, int row, bool keep) {
    __syncthreads();
    red[ty][lane] = v;
    __syncthreads();
    if (ty == (row & (CDH_KY - 1)) && live && keep) {
      double s = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < CDH_KY; ++w2) s += red[w2][lane];
      part[(size_t)row * q + n] = s;
    }
  };
  emit(cost, 0, true);
#pragma unroll
  for (int k = 0; k < D; ++k) {            // grad rows: c_0..c_{p-1}, d
    const bool keep = (k < p) || (k == PW);
    const int row = 1 + (k == PW ? p : k);
    emit(gacc[k], keep ? row : 0, keep);
  }
  {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const bool keep = ((i < p) || (i == PW)) && ((j < p) || (j == PW));
        const int ii = (i == PW) ? p : i, jj = (j == PW) ? p : j;       // index in the (p+1)-dim problem
        const int row = 1 + (p + 1) + ii * (ii + 1) / 2 + jj;
        emit(hacc[idx], keep ? row : 0, keep);
        ++idx;
      }
  }
}

// The same pass for wider latent states (12 < p <= 32), where the (p+1)(p+2)/2 Hessian accumulators no longer fit one
// lane's registers: the rows of the packed Hessian are dealt round-robin to NG row groups, blockIdx.z selects the group,
// and every group repeats the (cheap) per-bin preamble.  Group 0 also carries the cost and the gradient.  Same staging,
// same part layout (each group writes its own rows).  grid = (ceil(q/64), nby, NG), block = (64, CDH_KY).
// (20 latents - config 5 - in 4 groups: with 3 the 256 architectural registers were 4 short and the compiler parked values in AGPRs)
template <int PW> struct CdGroups { static constexpr int NG = (PW <= 12) ? 1 : (PW <= 16) ? 2 : (PW <= 20) ? 4 : 8; };
constexpr int cd_group_entries(int D, int NG, int G) {
  int n = 0;
  for (int i = 0; i < D; ++i)
    if (i % NG == G) n += i + 1;
  return n;
}

template <int PW, int NG, int G>
__device__ __forceinline__ void cd_hess_rows_body(const CdArgs& a, double (*Vt)[PW * PW], double (*Mt)[CdTile<PW>::TT], uint16_t* Yt,
                                                  double (*red)[64]) {
  constexpr int TT = CdTile<PW>::TT;
  constexpr int YS = TT + 4;
  constexpr int D = PW + 1;
  constexpr int NHG = cd_group_entries(D, NG, G);
  const int lane = threadIdx.x;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = ty * 64 + lane;
  const int n0 = blockIdx.x * 64;
  const int n = n0 + lane;
  const bool live = n < a.q;
  const int p = a.p, q = a.q, T = a.T;
  double c[PW], gacc[G == 0 ? D : 1], hacc[NHG];
#pragma unroll
  for (int l = 0; l < PW; ++l) c[l] = (live && l < p) ? a.vec[(size_t)l * q + n] : 0.0;
#pragma unroll
  for (int l = 0; l < (G == 0 ? D : 1); ++l) gacc[l] = 0.0;
#pragma unroll
  for (int l = 0; l < NHG; ++l) hacc[l] = 0.0;
  const double dn = live ? a.vec[(size_t)p * q + n] : 0.0;
  double cost = 0.0;

  const int ntt = (T + TT - 1) / TT;
  const int nitems = a.ntr * ntt;
  for (int item = blockIdx.y; item < nitems; item += gridDim.y) {
    const size_t r = a.trials[item / ntt];
    const int t0 = (item % ntt) * TT;
    const int tn = (T - t0 < TT) ? T - t0 : TT;
    const double* mean = a.mean + r * p * T;
    const double* vsm = a.vsm + (r * T + t0) * p * p;
    const uint8_t* Y = a.Y + r * q * T;
    const uint8_t* Yh = a.Yhi ? a.Yhi + r * q * T : nullptr;
    __syncthreads();
    for (int e = tid; e < TT * PW * PW; e += 64 * CDH_KY) {
      const int t = e / (PW * PW), kl = e - t * (PW * PW);
      const int k = kl / PW, l = kl - k * PW;
      Vt[t][kl] = (t < tn && k < p && l < p) ? vsm[(size_t)t * p * p + k * p + l] : 0.0;
    }
    for (int e = tid; e < PW * TT; e += 64 * CDH_KY) {
      const int k = e / TT, t = e - k * TT;
      Mt[k][t] = (k < p && t < tn) ? mean[(size_t)k * T + t0 + t] : 0.0;
    }
    for (int e = tid; e < 64 * TT; e += 64 * CDH_KY) {
      const int nn = e / TT, t = e - nn * TT;
      Yt[nn * YS + t] = (n0 + nn < q && t < tn) ? (uint16_t)count_at(Y, Yh, (size_t)(n0 + nn) * T + t0 + t) : (uint16_t)0;
    }
    __syncthreads();
    for (int t = ty; t < tn; t += CDH_KY) {
      double w[D];
      double hh = dn, rho = 0.0;
#pragma unroll
      for (int k = 0; k < PW; ++k) w[k] = 0.0;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
#pragma unroll
        for (int l = 0; l < k; ++l) {
          const double v = Vt[t][k * PW + l];
          w[k] += v * c[l];
          w[l] += v * c[k];
        }
        w[k] += Vt[t][k * PW + k] * c[k];
      }
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        rho += c[k] * w[k];
        const double mk = Mt[k][t];
        hh += c[k] * mk;
        w[k] += mk;
      }
      w[PW] = 1.0;
      const double yh = exp(hh + 0.5 * rho);
      const double y = (double)Yt[lane * YS + t];
      if (live) {
        if (G == 0) {
          cost += y * hh - yh;
#pragma unroll
          for (int k = 0; k < PW; ++k) gacc[k] -= y * Mt[k][t] - yh * w[k];
          gacc[G == 0 ? PW : 0] -= y - yh;
        }
        int idx = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          if (i % NG != G) continue;
          const double ywi = yh * w[i];
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            double add = ywi * w[j];
            if (i < PW) add += yh * Vt[t][i * PW + j];
            hacc[idx] += add;
            ++idx;
          }
        }
      }
    }
  }
  const int NH = 1 + (p + 1) + (p + 1) * (p + 2) / 2;
  double* part = a.part + (size_t)blockIdx.y * NH * q;
  auto emit = [&](double v, int row, bool keep) {
    __syncthreads();
    red[ty][lane] = v;
    __syncthreads();
    if (ty == (row & (CDH_KY - 1)) && live && keep) {
      double s = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < CDH_KY; ++w2) s += red[w2][lane];
      part[(size_t)row * q + n] = s;
    }
  };
  if (G == 0) {
    emit(cost, 0, true);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const bool keep = (k < p) || (k == PW);
      const int row = 1 + (k == PW ? p : k);
      emit(gacc[G == 0 ? k : 0], keep ? row : 0, keep);
    }
  }
  {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (i % NG != G) continue;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const bool keep = ((i < p) || (i == PW)) && ((j < p) || (j == PW));
        const int ii = (i == PW) ? p : i, jj = (j == PW) ? p : j;
        const int row = 1 + (p + 1) + ii * (ii + 1) / 2 + jj;
        emit(hacc[idx], keep ? row : 0, keep);
        ++idx;
      }
    }
  }
}

template <int PW, int NG>
__global__ __launch_bounds__(64 * CDH_KY) void mstep_cd_hess_rows_kernel(CdArgs a) {
  constexpr int TT = CdTile<PW>::TT;
  __shared__ __attribute__((aligned(16))) double Vt[TT][PW * PW];
  __shared__ double Mt[PW][TT];
  __shared__ uint16_t Yt[64 * (TT + 4)];
  __shared__ double red[CDH_KY][64];
  const int g = blockIdx.z;
  if constexpr (NG == 2) {
    if (g == 0) cd_hess_rows_body<PW, NG, 0>(a, Vt, Mt, Yt, red);
    else cd_hess_rows_body<PW, NG, 1>(a, Vt, Mt, Yt, red);
  } else if constexpr (NG == 3) {
    if (g == 0) cd_hess_rows_body<PW, NG, 0>(a, Vt, Mt, Yt, red);
    else if (g == 1) cd_hess_rows_body<PW, NG, 1>(a, Vt, Mt, Yt, red);
    else cd_hess_rows_body<PW, NG, 2>(a, Vt, Mt, Yt, red);
  } else if constexpr (NG == 4) {
    if (g == 0) cd_hess_rows_body<PW, NG, 0>(a, Vt, Mt, Yt, red);
    else if (g == 1) cd_hess_rows_body<PW, NG, 1>(a, Vt, Mt, Yt, red);
    else if (g == 2) cd_hess_rows_body<PW, NG, 2>(a, Vt, Mt, Yt, red);
    else cd_hess_rows_body<PW, NG, 3>(a, Vt, Mt, Yt, red);
  } else {
    static_assert(NG == 8, "row groups: 2, 3, 4 or 8");
    switch (g) {
      case 0: cd_hess_rows_body<PW, NG, 0>(a, Vt, Mt, Yt, red); break;
      case 1: cd_hess_rows_body<PW, NG, 1>(a, Vt, Mt, Yt, red); break;
      case 2: cd_hess_rows_body<PW, NG, 2>(a, Vt, Mt, Yt, red); break;
      case 3: cd_hess_rows_body<PW, NG, 3>(a, Vt, Mt, Yt, red); break;
      case 4: cd_hess_rows_body<PW, NG, 4>(a, Vt, Mt, Yt, red); break;
      case 5: cd_hess_rows_body<PW, NG, 5>(a, Vt, Mt, Yt, red); break;
      case 6: cd_hess_rows_body<PW, NG, 6>(a, Vt, Mt, Yt, red); break;
      default: cd_hess_rows_body<PW, NG, 7>(a, Vt, Mt, Yt, red); break;
    }
  }
}

// per-neuron Newton step: solve H delta = -g (dimension p+1, Cholesky with a tiny ridge), one thread per neuron.
// sums: [NH][q] as emitted above (not yet divided by R); prior: + inv_s2*(theta - center) on the gradient and
// + inv_s2 on the Hessian diagonal.  Writes delta[(p+1)][q] (vecCd layout) and dec[q] = -g.delta.
// rtot: device address of the (all-reduced) trial count R, read here so that the host need not fetch it before the launch;
// pack[q (p+3) + 1] = [row 0 of sums (per-neuron cost sums) | delta | dec | R]: everything the host reads back, one copy.
inline __global__ void cd_newton_step_kernel(const double* __restrict__ sums, int q, int p, const double* __restrict__ rtot, const double* __restrict__ vec,
                                      const double* __restrict__ center, double inv_s2, double* __restrict__ pack) {
  extern __shared__ double sm[];
  const int D = p + 1;
  double* A = sm + (size_t)threadIdx.x * (D * D + 2 * D);
  double* g = A + D * D;
  double* x = g + D;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const double invR = 1.0 / rtot[0];
  double* delta = pack + q;
  double* dec = pack + (size_t)q * (D + 1);
  if (n == 0) pack[(size_t)q * (D + 2)] = rtot[0];
  if (n >= q) return;
  pack[n] = sums[n];
  for (int i = 0; i < D; ++i) {
    double gi = sums[(size_t)(1 + i) * q + n] * invR;
    if (center) gi += inv_s2 * (vec[(size_t)i * q + n] - center[(size_t)i * q + n]);
    g[i] = gi;
    for (int j = 0; j <= i; ++j) {
      double h = sums[(size_t)(1 + D + i * (i + 1) / 2 + j) * q + n] * invR;
      if (i == j) h += (center ? inv_s2 : 0.0);
      A[i * D + j] = h;
    }
  }
  // Cholesky (lower) with a relative ridge if a pivot is not positive
  for (int j = 0; j < D; ++j) {
    double dj = A[j * D + j];
    for (int m = 0; m < j; ++m) dj -= A[j * D + m] * A[j * D + m];
    if (!(dj > 1e-300)) dj = 1e-12 * fabs(A[j * D + j]) + 1e-300;
    dj = sqrt(dj);
    A[j * D + j] = dj;
    for (int i = j + 1; i < D; ++i) {
      double v = A[i * D + j];
      for (int m = 0; m < j; ++m) v -= A[i * D + m] * A[j * D + m];
      A[i * D + j] = v / dj;
    }
  }
  for (int i = 0; i < D; ++i) {                      // forward: L y = -g
    double v = -g[i];
    for (int m = 0; m < i; ++m) v -= A[i * D + m] * x[m];
    x[i] = v / A[i * D + i];
  }
  for (int i = D - 1; i >= 0; --i) {                 // backward: L^T delta = y
    double v = x[i];
    for (int m = i + 1; m < D; ++m) v -= A[m * D + i] * x[m];
    x[i] = v / A[i * D + i];
  }
  double dd = 0.0;
  for (int i = 0; i < D; ++i) { delta[(size_t)i * q + n] = x[i]; dd -= g[i] * x[i]; }
  dec[n] = dd;
}

// chord pass: put a fresh cost/gradient (layout of mstep_cd_kernel: rows 0..p-1 dC, p dd, p+1 cost; signs of the
// maximised sum) into the cost/gradient rows of the Newton sums (rows 0 cost, 1..p+1 gradient of the minimised
// function); the Hessian rows of the last full pass stay.  One thread per (row, neuron).
inline __global__ void cd_chord_merge_kernel(const double* __restrict__ cg, int q, int p, double* __restrict__ sums) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (p + 2) * q) return;
  const int row = e / q, n = e - row * q;
  if (row == 0) sums[n] = cg[(size_t)(p + 1) * q + n];
  else sums[(size_t)row * q + n] = -cg[(size_t)(row - 1) * q + n];
}

// out[e] = sum_b part[b][e]
// A block owns 32 consecutive elements; its 8 groups of 32 threads each sum every 8th part (a serial walk over all nb
// parts by one thread is a chain of nb dependent-latency loads, ~1 us each), then the 8 group sums are added in a
// fixed order through LDS - the result does not depend on scheduling.  grid = ceil(len/32), block = 256.
inline __global__ __launch_bounds__(256) void reduce_parts_kernel(const double* __restrict__ part, int nb, int len, double* __restrict__ out) {
  __shared__ double red[8][33];
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  double s = 0.0;
  if (e < len)
    for (int b = grp; b < nb; b += 8) s += part[(size_t)b * len + e];
  red[grp][el] = s;
  __syncthreads();
  if (grp == 0 && e < len) {
    double t = red[0][el];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][el];
    out[e] = t;
  }
}

// --------------------------------------------------------------------------------------------------
// Dual variational E-step (inference.py:188-256), structured: with lmy = lambda - y,
//   v = C_big lmy  ->  v[k][t] = sum_n C[n][k] lmy[n][t]            (p x T)
//   W[t] = C^T diag(lambda[:,t]) C                                   (posterior precision blocks)
//   partial sums  sB = sum d_n lmy[n][t] ,  sD = sum lambda (log lambda - 1)
// grid = (ceil(T/64), nslots), block = 64 threads (one bin each).
// --------------------------------------------------------------------------------------------------
inline __global__ void dual_prep_kernel(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, const double* __restrict__ C, const double* __restrict__ d,
                                 const double* __restrict__ lam, long long sLam, double* __restrict__ V, long long sV,
                                 double* __restrict__ W, long long sW, double* __restrict__ part, int ntile,
                                 const int* __restrict__ slots, const int* __restrict__ trial_of_slot, int q, int p, int T) {
  const int slot = slots[blockIdx.y];
  const size_t trial = trial_of_slot[slot];
  const int t = blockIdx.x * 64 + threadIdx.x;
  double sB = 0.0, sD = 0.0;
  if (t < T) {
    const double* L = lam + (size_t)slot * sLam;
    const uint8_t* Yr = Y + trial * q * T;
    const uint8_t* Yh = Yhi ? Yhi + trial * q * T : nullptr;
    double* Wt = W + (size_t)slot * sW + (size_t)t * p * p;
    for (int k = 0; k < p; ++k) {
      double vk = 0.0;
      for (int n = 0; n < q; ++n) vk += C[(size_t)n * p + k] * (L[(size_t)n * T + t] - (double)count_at(Yr, Yh, (size_t)n * T + t));
      V[(size_t)slot * sV + (size_t)k * T + t] = vk;
      for (int l = 0; l <= k; ++l) {
        double w = 0.0;
        for (int n = 0; n < q; ++n) w += C[(size_t)n * p + k] * C[(size_t)n * p + l] * L[(size_t)n * T + t];
        Wt[k * p + l] = w;
        Wt[l * p + k] = w;
      }
    }
    for (int n = 0; n < q; ++n) {
      const double l = L[(size_t)n * T + t];
      sB += d[n] * (l - (double)count_at(Yr, Yh, (size_t)n * T + t));
      sD += l * (log(l) - 1.0);
    }
  }
  for (int off = 32; off > 0; off >>= 1) { sB += __shfl_down(sB, off); sD += __shfl_down(sD, off); }
  if (threadIdx.x == 0) {
    part[((size_t)slot * ntile + blockIdx.x) * 2 + 0] = sB;
    part[((size_t)slot * ntile + blockIdx.x) * 2 + 1] = sD;
  }
}

// dualProblem_grad (inference.py:218): g[n][t] = sum_k C[n][k] (Kv)[k][t] - d_n + log(lambda) - 0.5 c_n^T Sigma_t c_n
inline __global__ void dual_grad_kernel(const double* __restrict__ C, const double* __restrict__ d, const double* __restrict__ lam,
                                 const double* __restrict__ KV, const double* __restrict__ vsm_t, double* __restrict__ grad,
                                 int q, int p, int T) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int n = blockIdx.y;
  if (t >= T) return;
  const double* S = vsm_t + (size_t)t * p * p;
  const double* Cn = C + (size_t)n * p;
  double lin = 0.0, quad = 0.0;
  for (int k = 0; k < p; ++k) {
    lin += Cn[k] * KV[(size_t)k * T + t];
    double u = 0.0;
    for (int l = 0; l < p; ++l) u += S[k * p + l] * Cn[l];
    quad += Cn[k] * u;
  }
  grad[(size_t)n * T + t] = lin - d[n] + log(lam[(size_t)n * T + t]) - 0.5 * quad;
}

// the same for the slots [0, nslots) at once: grid = (ceil(T/64), q, nslots); lam / grad are [slot][q*T], KV [slot][ld],
// the per-bin blocks Sigma_t of slot s sit in vsm[trial_of_slot[s]]
inline __global__ void dual_grad_batch_kernel(const double* __restrict__ C, const double* __restrict__ d, const double* __restrict__ lam,
                                       const double* __restrict__ KV, long long sKV, const double* __restrict__ vsm,
                                       const int* __restrict__ trial_of_slot, double* __restrict__ grad, int q, int p, int T) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int n = blockIdx.y;
  const size_t slot = blockIdx.z;
  if (t >= T) return;
  const double* S = vsm + ((size_t)trial_of_slot[slot] * T + t) * p * p;
  const double* Cn = C + (size_t)n * p;
  const double* kv = KV + slot * sKV;
  double lin = 0.0, quad = 0.0;
  for (int k = 0; k < p; ++k) {
    lin += Cn[k] * kv[(size_t)k * T + t];
    double u = 0.0;
    for (int l = 0; l < p; ++l) u += S[k * p + l] * Cn[l];
    quad += Cn[k] * u;
  }
  const size_t e = slot * (size_t)q * T + (size_t)n * T + t;
  grad[e] = lin - d[n] + log(lam[e]) - 0.5 * quad;
}

// quad[slot][n][t] = 1/2 c_n^T Sigma_t c_n from the per-bin covariance blocks of the slot's trial (the vector form of the GEMM path in
// dual_gradient).  grid = (ceil(T/64), q, nslots), block = 64
inline __global__ void var_quad_kernel(const double* __restrict__ C, const double* __restrict__ vsm, const int* __restrict__ trial_of_slot,
                                double* __restrict__ quad, int q, int p, int T) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int n = blockIdx.y;
  const size_t slot = blockIdx.z;
  if (t >= T) return;
  const double* S = vsm + ((size_t)trial_of_slot[slot] * T + t) * p * p;
  const double* Cn = C + (size_t)n * p;
  double acc = 0.0;
  for (int k = 0; k < p; ++k) {
    double u = 0.0;
    for (int l = 0; l < p; ++l) u += S[k * p + l] * Cn[l];
    acc += Cn[k] * u;
  }
  quad[slot * (size_t)q * T + (size_t)n * T + t] = 0.5 * acc;
}

// lam <- exp(lam) in place (the array holds rho on entry), or lam <- fill; flag[0] set when an entry of rho is not finite or its exp is not
// positive and finite.  grid-stride over n entries
inline __global__ void var_exp_kernel(double* __restrict__ lam, size_t n, int use_fill, double fill, int* __restrict__ flag) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double r = lam[i];
    const double v = use_fill ? fill : exp(r);
    if (!use_fill && (!isfinite(r) || !(v > 0.0) || !isfinite(v))) flag[0] = 1;
    lam[i] = v;
  }
}
// out <- log(lam)
inline __global__ void var_log_kernel(const double* __restrict__ lam, double* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = log(lam[i]);
}

// maximum that keeps a NaN (fmax drops it: a slot of non-finite offsets would report a change of zero and pass for converged)
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : fmax(a, b)); }

// v <- v + damp (vnew - v) over the m entries of every slot; delta[slot] = max |vnew - v| (before the update; NaN if any entry is).  grid = nslots, block = 256
inline __global__ __launch_bounds__(256) void var_update_kernel(double* __restrict__ v, const double* __restrict__ vnew, size_t m, const double* __restrict__ damp,
                                                         double* __restrict__ delta) {
  __shared__ double red[4];
  const size_t o = (size_t)blockIdx.x * m;
  const double s = damp[blockIdx.x];
  double d = 0.0;
  for (size_t i = threadIdx.x; i < m; i += 256) {
    const double a = v[o + i], b = vnew[o + i];
    d = nanmax(d, fabs(b - a));
    v[o + i] = a + s * (b - a);
  }
  for (int off = 32; off > 0; off >>= 1) d = nanmax(d, __shfl_down(d, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) delta[blockIdx.x] = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
}

// ---- small batched vector kernels of the device L-BFGS (one optimisation per slot, vectors [slot][m]) ----------------
// out[slot] = a[slot] . b[slot]; grid = nslots, block = 256 (fixed reduction tree: deterministic)
inline __global__ __launch_bounds__(256) void bdot_kernel(const double* __restrict__ A, const double* __restrict__ B, size_t m, double* __restrict__ out) {
  __shared__ double red[256];
  const double* a = A + (size_t)blockIdx.x * m;
  const double* b = B + (size_t)blockIdx.x * m;
  double s = 0.0;
  for (size_t i = threadIdx.x; i < m; i += 256) s += a[i] * b[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
// out[slot] = max_i |a[slot][i]|
inline __global__ __launch_bounds__(256) void bmaxabs_kernel(const double* __restrict__ A, size_t m, double* __restrict__ out) {
  __shared__ double red[256];
  const double* a = A + (size_t)blockIdx.x * m;
  double s = 0.0;
  for (size_t i = threadIdx.x; i < m; i += 256) s = fmax(s, fabs(a[i]));
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
// y[slot] = beta[slot] * y[slot] + alpha[slot] * x[slot]   (beta NULL: 1); grid = (ceil(m/256), nslots)
inline __global__ void baxpby_kernel(const double* __restrict__ alpha, const double* __restrict__ X, const double* __restrict__ beta,
                              double* __restrict__ Y, size_t m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const size_t e = (size_t)blockIdx.y * m + i;
  Y[e] = (beta ? beta[blockIdx.y] : 1.0) * Y[e] + alpha[blockIdx.y] * X[e];
}
// z[slot] = x[slot] + t[slot] * d[slot]
inline __global__ void bstep_kernel(const double* __restrict__ X, const double* __restrict__ D, const double* __restrict__ t, double* __restrict__ Z,
                             size_t m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const size_t e = (size_t)blockIdx.y * m + i;
  Z[e] = X[e] + t[blockIdx.y] * D[e];
}
// dst[slot] = a[slot] - b[slot]  where take[slot] != 0 (other slots untouched)
inline __global__ void bdiff_kernel(const double* __restrict__ A, const double* __restrict__ B, const int* __restrict__ take, double* __restrict__ dst,
                             size_t m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || !take[blockIdx.y]) return;
  const size_t e = (size_t)blockIdx.y * m + i;
  dst[e] = A[e] - B[e];
}
// dst[slot] = src[slot] where take[slot] != 0
inline __global__ void bcopy_kernel(const double* __restrict__ src, const int* __restrict__ take, double* __restrict__ dst, size_t m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || !take[blockIdx.y]) return;
  const size_t e = (size_t)blockIdx.y * m + i;
  dst[e] = src[e];
}
// lam = exp(rho)
inline __global__ void exp_kernel(const double* __restrict__ rho, double* __restrict__ lam, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lam[i] = exp(rho[i]);
}
// g_rho = g_lambda * lambda  (dualProblemRho_grad, inference.py:251-256)
inline __global__ void chain_kernel(const double* __restrict__ glam, const double* __restrict__ lam, double* __restrict__ grho, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) grho[i] = glam[i] * lam[i];
}

// out[row] = sum of the `len` consecutive entries of row `row`; grid = rows, block = 256
inline __global__ __launch_bounds__(256) void sum_rows_kernel(const double* __restrict__ A, int len, double* __restrict__ out) {
  __shared__ double red[256];
  const double* a = A + (size_t)blockIdx.x * len;
  double s = 0.0;
  for (int i = threadIdx.x; i < len; i += 256) s += a[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
// x = -Kv  (VIPostMean, inference.py:193-194)
inline __global__ void negate_rows_kernel(const double* __restrict__ src, long long sSrc, double* __restrict__ dst, long long sDst, int n,
                                   const int* __restrict__ slots) {
  const int slot = slots[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(size_t)slot * sDst + i] = -src[(size_t)slot * sSrc + i];
}

// --------------------------------------------------------------------------------------------------
// Size of the blocks B[i] (p x p, i < nblocks, contiguous): v_i = scale * ||B_i||_inf; out_bits <- max_i v_i (atomicMax on the float's
// bit pattern: non-negative floats order like their bits), *out_sq += sum_i v_i^2.  Used on the blocks Wt = W (I + eps W)^-1 of a
// chunk: eps ||Wt_t|| is the relative size of the mixing correction D = eps Wt y of the low-rank covariance engine, its root mean
// square over (trial, bin) what the precision of the split accumulation depends on.  grid = ceil(nblocks/256), block = 256.
// One thread per ROW of a block (consecutive threads read consecutive 8p-byte rows: whole cache lines; a thread per block read its
// own p^2 doubles at stride 8p^2 and ran at 0.55 TB/s), row sums through LDS, the first row's thread of every block takes the maximum.
// A workgroup walks chunks of 256 rows with stride gridDim.x and issues ONE pair of atomics at the end (a pair per wave per chunk was
// 160 000 atomics on two addresses: 1.9 ms).  grid = min(ceil(nblocks p / 256), 2048), block = 256.
inline __global__ __launch_bounds__(256) void block_norm_max_kernel(const double* __restrict__ B, long long nblocks, int p, double scale, unsigned* __restrict__ out_bits,
                                                             double* __restrict__ out_sq) {
  __shared__ double rs[256 + 32];
  __shared__ float wv[4];
  __shared__ double wsq[4];
  const long long nrows = nblocks * p;
  const long long nchunks = (nrows + 255) / 256;
  float vmax = 0.f;
  double sq = 0.0;
  for (long long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const long long row = ch * 256 + threadIdx.x;                                 // global row index = block * p + r
    double s = 0.0;
    if (row < nrows) {
      const double* b = B + row * p;
      for (int c2 = 0; c2 < p; ++c2) s += fabs(b[c2]);
    }
    __syncthreads();                                                              // the previous chunk's sums have been read
    rs[threadIdx.x] = s;
    // the rows of a block that starts in this chunk may run into the next one: those (at most p - 1 <= 31) rows are read again here
    if (threadIdx.x < 32) {
      const long long row2 = (ch + 1) * 256 + threadIdx.x;
      double s2 = 0.0;
      if (threadIdx.x < p - 1 && row2 < nrows) {
        const double* b = B + row2 * p;
        for (int c2 = 0; c2 < p; ++c2) s2 += fabs(b[c2]);
      }
      rs[256 + threadIdx.x] = s2;
    }
    __syncthreads();
    if (row < nrows && row % p == 0) {
      double worst = 0.0;
      for (int r = 0; r < p; ++r) worst = fmax(worst, rs[threadIdx.x + r]);
      const float v = (float)(scale * worst);
      vmax = fmaxf(vmax, v);
      sq += (double)v * (double)v;
    }
  }
  for (int off = 32; off > 0; off >>= 1) { vmax = fmaxf(vmax, __shfl_down(vmax, off)); sq += __shfl_down(sq, off); }
  if ((threadIdx.x & 63) == 0) { wv[threadIdx.x >> 6] = vmax; wsq[threadIdx.x >> 6] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = fmaxf(fmaxf(wv[0], wv[1]), fmaxf(wv[2], wv[3]));
    if (v > 0.f) atomicMax(out_bits, __float_as_uint(v));
    if (out_sq) atomicAdd(out_sq, (wsq[0] + wsq[1]) + (wsq[2] + wsq[3]));
  }
}

// Shared-preconditioner Newton-PCG pieces.
// Wbar[t] = mean over the listed slots of W[slot][t]   (p x p per bin) -> written into slot `dst` of Wdst
// --------------------------------------------------------------------------------------------------
inline __global__ void mean_w_kernel(const double* __restrict__ W, long long sW, const int* __restrict__ slots, int nslots, int len,
                              double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  // (eight independent partial sums: the slots' loads are in flight together; one running sum made the launch latency-bound)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
  int i = 0;
  for (; i + 7 < nslots; i += 8) {
    s0 += W[(size_t)slots[i] * sW + e];
    s1 += W[(size_t)slots[i + 1] * sW + e];
    s2 += W[(size_t)slots[i + 2] * sW + e];
    s3 += W[(size_t)slots[i + 3] * sW + e];
    s4 += W[(size_t)slots[i + 4] * sW + e];
    s5 += W[(size_t)slots[i + 5] * sW + e];
    s6 += W[(size_t)slots[i + 6] * sW + e];
    s7 += W[(size_t)slots[i + 7] * sW + e];
  }
  for (; i < nslots; ++i) s0 += W[(size_t)slots[i] * sW + e];
  out[e] = (((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))) / (double)nslots;
}

// q = W p  added onto q (which already holds Kinv p):  q[(k,t)] += sum_l W[t][k][l] p[(l,t)], and the partial
// products pqpart[slot][tile] = sum over the tile's bins of p.q (summed in tile order by the consumer: deterministic).
// A block owns 64 bins of one slot: the W_t blocks are staged in LDS with coalesced reads (odd row stride: each lane
// then reads its own block conflict-free); lanes = bins, the 4 waves take the output rows k = w, w+4, ...
// grid = (ceil(T/64), nslots), block = 256, p <= PW.
template <int PW>
__global__ __launch_bounds__(256) void pcg_hessvec_dot_kernel(const double* __restrict__ W, long long sW, const double* __restrict__ P,
                                                              double* __restrict__ Q, long long sV, int T, int p,
                                                              const int* __restrict__ slots, double* __restrict__ pqpart,
                                                              const int* __restrict__ nlive = nullptr) {
  if (nlive && (int)blockIdx.y >= *nlive) return;                 // (slots is a device-side live list of that length)
  constexpr int PP = PW * PW, LD = PP + 1;
  __shared__ double Ws[64 * LD];
  __shared__ double red[4];
  const int pp = p * p;
  const size_t slot = slots[blockIdx.y];
  const int t0 = blockIdx.x * 64;
  const int nt = min(64, T - t0);
  const double* wbase = W + slot * sW + (size_t)t0 * pp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the lane's slice of p and its rows of q are requested first: those loads then travel together with the W tile's
  // instead of starting after the barrier (a block is one short chain of dependent global accesses, so their
  // latencies add up)
  constexpr int NK = (PW + 3) / 4;
  const bool live = lane < nt;
  const double* pv = P + slot * sV + t0 + lane;
  double* q = Q + slot * sV + t0 + lane;
  double v[PW], qk[NK];
#pragma unroll
  for (int l = 0; l < PW; ++l) v[l] = (live && l < p) ? pv[(size_t)l * T] : 0.0;
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int k = wave + 4 * i;
    qk[i] = (live && k < p) ? q[(size_t)k * T] : 0.0;
  }
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const int t = e / pp, idx = e - t * pp;
    Ws[t * LD + idx] = wbase[e];                               // block of bin t keeps its p x p layout (stride LD)
  }
  __syncthreads();
  double acc = 0.0;
  if (live) {
    const double* wt = Ws + lane * LD;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int k = wave + 4 * i;
      if (k < p) {
        double s2 = qk[i];
#pragma unroll
        for (int l = 0; l < PW; ++l)
          if (l < p) s2 += wt[k * p + l] * v[l];
        q[(size_t)k * T] = s2;
        double vk = 0.0;
#pragma unroll
        for (int l = 0; l < PW; ++l) vk = (l == k) ? v[l] : vk;
        acc += s2 * vk;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) pqpart[slot * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The same for wide latent states (p > 16: the staged tile would not fit in LDS): block per slot, thread per bin,
// W_t read straight from memory; one partial product per slot (ntile = 1 for the consumer).
template <int PW>
__global__ __launch_bounds__(256) void pcg_hessvec_dot_wide_kernel(const double* __restrict__ W, long long sW, const double* __restrict__ P,
                                                                   double* __restrict__ Q, long long sV, int T, int p,
                                                                   const int* __restrict__ slots, double* __restrict__ pq) {
  __shared__ double red[4];
  const size_t slot = slots[blockIdx.x];
  const double* w = W + slot * sW;
  const double* pv = P + slot * sV;
  double* q = Q + slot * sV;
  double acc = 0.0;
  for (int t = threadIdx.x; t < T; t += 256) {
    const double* wt = w + (size_t)t * p * p;
    double v[PW];
#pragma unroll
    for (int l = 0; l < PW; ++l) v[l] = (l < p) ? pv[(size_t)l * T + t] : 0.0;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      if (k < p) {
        double s2 = q[(size_t)k * T + t];
#pragma unroll
        for (int l = 0; l < PW; ++l)
          if (l < p) s2 += wt[k * p + l] * v[l];
        q[(size_t)k * T + t] = s2;
        acc += s2 * v[k];
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) pq[slot] = red[0] + red[1] + red[2] + red[3];
}

// alpha = rz/pq (pq = sum of the ntile partial products) ; x += alpha p ; r -= alpha q   (block per slot)
inline __global__ __launch_bounds__(256) void pcg_update_xr_kernel(double* __restrict__ X, double* __restrict__ R, const double* __restrict__ P,
                                                            const double* __restrict__ Q, long long sV, int n, const int* __restrict__ slots,
                                                            const double* __restrict__ rz, const double* __restrict__ pqpart, int ntile) {
  const size_t slot = slots[blockIdx.x];
  double d = 0.0;
  for (int i = 0; i < ntile; ++i) d += pqpart[slot * ntile + i];
  const double alpha = (d > 0.0) ? rz[slot] / d : 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    X[slot * sV + i] += alpha * P[slot * sV + i];
    R[slot * sV + i] -= alpha * Q[slot * sV + i];
  }
}

// rz_new = r.z ; beta = rz_new/rz (0 on the first call: first != 0) ; p = z + beta p ; rz = rz_new ; also rr = r.r
inline __global__ __launch_bounds__(256) void pcg_update_p_kernel(const double* __restrict__ R, const double* __restrict__ Z, double* __restrict__ P,
                                                           long long sV, int n, const int* __restrict__ slots, double* __restrict__ rz,
                                                           double* __restrict__ rr, int first) {
  __shared__ double red[2][4];
  __shared__ double beta_s;
  const size_t slot = slots[blockIdx.x];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double r = R[slot * sV + i];
    a += r * Z[slot * sV + i];
    b += r * r;
  }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double rzn = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double old = rz[slot];
    beta_s = (first || !(old > 0.0)) ? 0.0 : rzn / old;
    rz[slot] = rzn;
    rr[slot] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
  __syncthreads();
  const double beta = beta_s;
  for (int i = threadIdx.x; i < n; i += 256) P[slot * sV + i] = Z[slot * sV + i] + beta * P[slot * sV + i];
}

// r = -g ; x = 0  (rows >= n of every vector are kept at zero)
inline __global__ void pcg_init_kernel(const double* __restrict__ G, double* __restrict__ R, double* __restrict__ X, long long sV, int n, int npad,
                                const int* __restrict__ slots) {
  const size_t slot = slots[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) {
    R[slot * sV + i] = (i < n) ? -G[slot * sV + i] : 0.0;
    X[slot * sV + i] = 0.0;
  }
}

// out[slot] = |Gl + KX|^2 : squared norm of the total gradient at the committed point (block per listed slot)
inline __global__ __launch_bounds__(256) void grad_norm2_kernel(const double* __restrict__ Gl, const double* __restrict__ KX, long long sV, int n,
                                                         const int* __restrict__ slots, double* __restrict__ out) {
  __shared__ double red[4];
  const size_t slot = slots[blockIdx.x];
  double d = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double g = Gl[slot * sV + i] + KX[slot * sV + i];
    d += g * g;
  }
  for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) out[slot] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dec = -g.x ; smax = max|x|   (block per slot)
inline __global__ __launch_bounds__(256) void step_stats_kernel(const double* __restrict__ G, const double* __restrict__ X, long long sV, int n,
                                                         const int* __restrict__ slots, double* __restrict__ dec, double* __restrict__ smax) {
  __shared__ double red[2][4];
  const size_t slot = slots[blockIdx.x];
  double d = 0.0, m = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double x = X[slot * sV + i];
    d -= G[slot * sV + i] * x;
    m = fmax(m, fabs(x));
  }
  for (int off = 32; off > 0; off >>= 1) { d += __shfl_down(d, off); m = fmax(m, __shfl_down(m, off)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d; red[1][threadIdx.x >> 6] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dec[slot] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    smax[slot] = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
  }
}

// The same for a step the inner solve kept in single precision on its private row stride (pcg.h: PcgCgP::X32): x[k Tl + t] is widened into the
// caller's FP64 step vector X[k T + t] on the way.
inline __global__ __launch_bounds__(256) void step_stats_x32_kernel(const double* __restrict__ G, const float* __restrict__ X32, double* __restrict__ X, long long sV,
                                                             int T, int Tl, int p, const int* __restrict__ slots, double* __restrict__ dec, double* __restrict__ smax) {
  __shared__ double red[2][4];
  const size_t slot = slots[blockIdx.x];
  double d = 0.0, m = 0.0;
  for (int k = 0; k < p; ++k)
    for (int t = threadIdx.x; t < T; t += 256) {
      const double x = (double)X32[slot * sV + (size_t)k * Tl + t];
      const size_t o = slot * sV + (size_t)k * T + t;
      X[o] = x;
      d -= G[o] * x;
      m = fmax(m, fabs(x));
    }
  for (int off = 32; off > 0; off >>= 1) { d += __shfl_down(d, off); m = fmax(m, __shfl_down(m, off)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d; red[1][threadIdx.x >> 6] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dec[slot] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    smax[slot] = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
  }
}

// --------------------------------------------------------------------------------------------------
// Low-rank covariance engine.  K_k = eps*I + F_k F_k^T with F_k (T x r_k) from a pivoted Cholesky of
// the RBF part (numerically low rank: eigenvalues decay like a Gaussian), so with W = blockdiag_t(W_t),
//   G = (I + eps W)^-1 ,  Wt = W G   (p x p per bin, symmetric)
//   Sigma = H^-1 = eps G + G F (I_r + F^T Wt F)^-1 F^T G                    (Woodbury, exact)
// and every O(n^3) step of the covariance phase becomes O(T r^2) / O(r^3), r = sum_k r_k << n = pT.
// --------------------------------------------------------------------------------------------------

// Pivoted Cholesky of (1-eps)*RBF_k, one workgroup of NT threads per latent.  F: [p][Tf x Tf] column-major slabs
// (column j of latent k at F + k*Tf*Tf + j*Tf), rows >= T and unused columns are zero.  rank[k] out.
// A step = pivot search (wave shuffles, one barrier), row `piv` of the factor so far into LDS, then every bin's new entry
// v_t = (K[t][piv] - sum_m F[m][t] F[m][piv]) / sqrt(d_piv): the columns of F are read back from memory (L2), eight independent
// partial sums so that sixteen loads are in flight per thread - the step is bound by that latency, not by its j T multiply-adds.
// dynamic LDS = (T + rmax + NT / 64) doubles + NT / 64 ints.
template <int NT, int NS = 1>
__global__ __launch_bounds__(NT * NS) void rbf_pivchol_kernel(double* __restrict__ F, int Tf, int T, const double* __restrict__ tau, double bin,
                                                              double eps, double tol, int rmax, int* __restrict__ rank) {
  // NT threads cover the rows (bins); NS groups of NT threads share the dot products of a step: group g takes the 8-column blocks
  // g, g + NS, ... of the columns found so far (the step is bound by how fast ONE workgroup pulls the factor out of L2: two groups
  // have twice the loads in flight), partial sums meet in LDS
  constexpr int NW = NT * NS / 64;
  extern __shared__ double sh[];              // d[T] | frow[rmax] | wave maxima [NW] | partial sums [NS - 1][T] ; then int wave argmax [NW]
  double* d = sh;
  double* frow = sh + T;
  double* rv = frow + rmax;
  double* psum = rv + NW;
  int* ri = reinterpret_cast<int*>(psum + (size_t)(NS - 1) * T);
  const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / NT, rt = tid - grp * NT;                    // dot-product group, row thread
  constexpr int NTT = NT * NS;
  double* Fk = F + (size_t)k * Tf * Tf;
  const double den = (tau[k] * 1000.0) * (tau[k] * 1000.0);
  for (size_t e = tid; e < (size_t)Tf * Tf; e += NTT) Fk[e] = 0.0;
  for (int t = tid; t < T; t += NTT) d[t] = 1.0 - eps;
  __syncthreads();
  int j = 0;
  for (; j < rmax; ++j) {
    // largest remaining diagonal entry, lowest index on ties
    double best = -1.0; int bi = 0x7fffffff;
    for (int t = tid; t < T; t += NTT) if (d[t] > best) { best = d[t]; bi = t; }
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_down(best, off);
      const int oi = __shfl_down(bi, off);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
    __syncthreads();
    double dp = rv[0]; int piv = ri[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
      if (rv[w] > dp || (rv[w] == dp && ri[w] < piv)) { dp = rv[w]; piv = ri[w]; }
    if (!(dp > tol)) break;                                         // (uniform: every thread reads the same LDS values)
    for (int m = tid; m < j; m += NTT) frow[m] = Fk[(size_t)m * Tf + piv];
    __syncthreads();
    const double rs = 1.0 / sqrt(dp);
    for (int t0 = 0; t0 < T; t0 += NT) {                          // (uniform trip count: the barrier below is for every thread)
      const int t = t0 + rt;
      double part = 0.0;
      if (t < T) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
        const double* col = Fk + t;
        int m = 8 * grp;
#pragma unroll 2
        for (; m + 7 < j; m += 8 * NS) {
          s0 += col[(size_t)m * Tf] * frow[m];
          s1 += col[(size_t)(m + 1) * Tf] * frow[m + 1];
          s2 += col[(size_t)(m + 2) * Tf] * frow[m + 2];
          s3 += col[(size_t)(m + 3) * Tf] * frow[m + 3];
          s4 += col[(size_t)(m + 4) * Tf] * frow[m + 4];
          s5 += col[(size_t)(m + 5) * Tf] * frow[m + 5];
          s6 += col[(size_t)(m + 6) * Tf] * frow[m + 6];
          s7 += col[(size_t)(m + 7) * Tf] * frow[m + 7];
        }
        // the last, partly filled block of 8 columns belongs to the group whose turn it is
        for (int mm = m; mm < j && mm < m + 8; ++mm) s0 += col[(size_t)mm * Tf] * frow[mm];
        part = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        if (NS > 1 && grp > 0) psum[(size_t)(grp - 1) * T + t] = part;
      }
      if (NS > 1) __syncthreads();
      if (t < T && grp == 0) {
        double tot = part;
#pragma unroll
        for (int g2 = 1; g2 < NS; ++g2) tot += psum[(size_t)(g2 - 1) * T + t];
        const double dt = (double)t * bin - (double)piv * bin;
        double v = (1.0 - eps) * exp(-0.5 * ((dt * dt) / den));
        v -= tot;
        v *= rs;
        Fk[(size_t)j * Tf + t] = v;
        d[t] = (t == piv) ? 0.0 : d[t] - v * v;
      }
      if (NS > 1 && t0 + NT < T) __syncthreads();                  // (psum is rewritten by the next trip)
    }
    __syncthreads();
  }
  if (tid == 0) rank[k] = j;
}

// The same factorisation with TWO bins per row thread (16-byte loads of the factor columns) and NS = 4 column groups: a step's chain of
// dependent memory round trips - what it is made of, the work is negligible - is half as long (64 columns per round instead of 32), the
// pivot search runs on the new diagonal while it is still in registers (one barrier and one LDS sweep less per step).  Same pivots (largest
// remaining diagonal entry, lowest index on ties); the sums are grouped differently, so the factor agrees with rbf_pivchol_kernel's to rounding.
// dynamic LDS = pivchol2_lds(T, rmax, NT * NS, NS).  Tf even (rows come in aligned pairs; it is a multiple of 64).
constexpr size_t pivchol2_lds(int T, int rmax, int threads, int ns) {
  return ((size_t)((T + 1) & ~1) * ns + rmax + threads / 64 + 2) * sizeof(double) + (size_t)(threads / 64 + 2) * sizeof(int);
}
template <int NT, int NS>
__global__ __launch_bounds__(NT * NS) void rbf_pivchol2_kernel(double* __restrict__ F, int Tf, int T, const double* __restrict__ tau, double bin,
                                                               double eps, double tol, int rmax, int* __restrict__ rank) {
  constexpr int NTT = NT * NS, NWR = NT / 64;           // threads; waves of group 0 (the ones that own the diagonal)
  const int T2 = (T + 1) & ~1;
  extern __shared__ double sh[];                        // d[T2] | frow[rmax] | wave maxima [NWR] | partial sums [NS - 1][T2] ; int wave argmax [NWR]
  double* d = sh;
  double* frow = d + T2;
  double* rv = frow + rmax;
  double* psum = rv + NTT / 64 + (rmax & 1);          // (16-byte aligned: its rows are read in pairs)
  int* ri = reinterpret_cast<int*>(psum + (size_t)(NS - 1) * T2);
  const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / NT, rt = tid - grp * NT;
  double* Fk = F + (size_t)k * Tf * Tf;
  const double den = (tau[k] * 1000.0) * (tau[k] * 1000.0);
  for (size_t e = tid; e < (size_t)Tf * Tf; e += NTT) Fk[e] = 0.0;
  for (int t = tid; t < T2; t += NTT) d[t] = t < T ? 1.0 - eps : -1.0;
  if (tid < NWR) { rv[tid] = tid == 0 ? 1.0 - eps : -1.0; ri[tid] = tid == 0 ? 0 : 0x7fffffff; }     // the first pivot: bin 0
  __syncthreads();
  int j = 0;
  for (; j < rmax; ++j) {
    double dp = rv[0]; int piv = ri[0];
#pragma unroll
    for (int w = 1; w < NWR; ++w)
      if (rv[w] > dp || (rv[w] == dp && ri[w] < piv)) { dp = rv[w]; piv = ri[w]; }
    if (!(dp > tol)) break;                                         // (uniform: every thread reads the same LDS values)
    for (int m = tid; m < j; m += NTT) frow[m] = Fk[(size_t)m * Tf + piv];
    __syncthreads();
    const double rs = 1.0 / sqrt(dp);
    double best = -1.0; int bi = 0x7fffffff;
    for (int t0 = 0; t0 < T2; t0 += 2 * NT) {                     // (uniform trip count: the barriers below are for every thread)
      const int t = t0 + 2 * rt;
      double2_t part = {0.0, 0.0};
      if (t < T2) {
        double2_t s0 = {0.0, 0.0}, s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0, s6 = s0, s7 = s0;
        const double* col = Fk + t;
        int m = 8 * grp;
#pragma unroll 2
        for (; m + 7 < j; m += 8 * NS) {
          s0 += *reinterpret_cast<const double2_t*>(col + (size_t)m * Tf) * frow[m];
          s1 += *reinterpret_cast<const double2_t*>(col + (size_t)(m + 1) * Tf) * frow[m + 1];
          s2 += *reinterpret_cast<const double2_t*>(col + (size_t)(m + 2) * Tf) * frow[m + 2];
          s3 += *reinterpret_cast<const double2_t*>(col + (size_t)(m + 3) * Tf) * frow[m + 3];
          s4 += *reinterpret_cast<const double2_t*>(col + (size_t)(m + 4) * Tf) * frow[m + 4];
          s5 += *reinterpret_cast<const double2_t*>(col + (size_t)(m + 5) * Tf) * frow[m + 5];
          s6 += *reinterpret_cast<const double2_t*>(col + (size_t)(m + 6) * Tf) * frow[m + 6];
          s7 += *reinterpret_cast<const double2_t*>(col + (size_t)(m + 7) * Tf) * frow[m + 7];
        }
        // the last, partly filled block of 8 columns belongs to the group whose turn it is
        for (int mm = m; mm < j && mm < m + 8; ++mm) s0 += *reinterpret_cast<const double2_t*>(col + (size_t)mm * Tf) * frow[mm];
        part = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        if (grp > 0) *reinterpret_cast<double2_t*>(psum + (size_t)(grp - 1) * T2 + t) = part;
      }
      __syncthreads();
      if (t < T2 && grp == 0) {
        double2_t tot = part;
#pragma unroll
        for (int g2 = 1; g2 < NS; ++g2) tot += *reinterpret_cast<const double2_t*>(psum + (size_t)(g2 - 1) * T2 + t);
        double2_t v;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const double dt = (double)(t + i) * bin - (double)piv * bin;
          double vi = (1.0 - eps) * exp(-0.5 * ((dt * dt) / den));
          vi -= tot[i];
          vi *= rs;
          v[i] = (t + i < T) ? vi : 0.0;
        }
        if (t + 1 < T) *reinterpret_cast<double2_t*>(Fk + (size_t)j * Tf + t) = v;
        else if (t < T) Fk[(size_t)j * Tf + t] = v[0];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (t + i < T) {
            const double dn = (t + i == piv) ? 0.0 : d[t + i] - v[i] * v[i];
            d[t + i] = dn;
            if (dn > best) { best = dn; bi = t + i; }                // (ascending bins: the lowest index wins a tie)
          }
        }
      }
      if (t0 + 2 * NT < T2) __syncthreads();                        // (psum is rewritten by the next trip)
    }
    if (grp == 0) {
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off);
        const int oi = __shfl_down(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
    }
    __syncthreads();
  }
  if (tid == 0) rank[k] = j;
}

// per (slot, bin): G = (I + eps W)^-1 and Wt = W G.  One thread per matrix, matrices in dynamic LDS.
// (ldet, optional: ldet[item] = log det(I + eps W_t), item = list position * T + t)
inline __global__ void bin_blocks_kernel(const double* __restrict__ W, long long sW, double* __restrict__ G, double* __restrict__ Wt, long long sO,
                                  int T, int p, double eps, const int* __restrict__ slots, int nslots, double* __restrict__ ldet) {
  extern __shared__ double sm[];
  const int pp = p * p, stride = 2 * pp + 1;
  double* A = sm + (size_t)threadIdx.x * stride;   // A: I + eps W -> L -> L^-1 ; Gm = result
  double* Gm = A + pp;
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= (long long)nslots * T) return;
  const int slot = slots[item / T];
  const int t = (int)(item % T);
  const double* w = W + (size_t)slot * sW + (size_t)t * pp;
  for (int i = 0; i < p; ++i)
    for (int j2 = 0; j2 < p; ++j2) A[i * p + j2] = eps * w[i * p + j2] + (i == j2 ? 1.0 : 0.0);
  // Cholesky (lower, in place)
  double logdet = 0.0;
  for (int j2 = 0; j2 < p; ++j2) {
    double dj = A[j2 * p + j2];
    for (int m = 0; m < j2; ++m) dj -= A[j2 * p + m] * A[j2 * p + m];
    logdet += log(dj);
    dj = sqrt(dj);
    A[j2 * p + j2] = dj;
    for (int i = j2 + 1; i < p; ++i) {
      double v = A[i * p + j2];
      for (int m = 0; m < j2; ++m) v -= A[i * p + m] * A[j2 * p + m];
      A[i * p + j2] = v / dj;
    }
  }
  if (ldet) ldet[item] = logdet;
  // invert L in place (lower)
  for (int j2 = 0; j2 < p; ++j2) {
    A[j2 * p + j2] = 1.0 / A[j2 * p + j2];
    for (int i = j2 + 1; i < p; ++i) {
      double v = 0.0;
      for (int m = j2; m < i; ++m) v -= A[i * p + m] * A[m * p + j2];
      A[i * p + j2] = v / A[i * p + i];
    }
  }
  // (column j2 reads L[i][m], m >= j2, and L[i][i], i > j2, which are still un-inverted at that point)
  // G = L^-T L^-1
  double* g = G + (size_t)slot * sO + (size_t)t * pp;
  double* wt = Wt + (size_t)slot * sO + (size_t)t * pp;
  for (int i = 0; i < p; ++i)
    for (int j2 = 0; j2 <= i; ++j2) {
      double v = 0.0;
      for (int m = i; m < p; ++m) v += A[m * p + i] * A[m * p + j2];
      Gm[i * p + j2] = v;
      Gm[j2 * p + i] = v;
    }
  for (int i = 0; i < p; ++i)
    for (int j2 = 0; j2 < p; ++j2) {
      g[i * p + j2] = Gm[i * p + j2];
      double v = 0.0;
      for (int m = 0; m < p; ++m) v += w[i * p + m] * Gm[m * p + j2];
      wt[i * p + j2] = v;
    }
}

// The same for 10 < p <= PMAX <= 32 with 32 lanes per matrix (two matrices per wave) instead of one thread per matrix - the
// thread-per-matrix form above fits 7 threads per workgroup at 20 latents.  Lane c owns row c during the factorisation (in LDS,
// odd row stride: the lanes' rows start in distinct banks) and column c afterwards: x = column c of L^-1 by forward substitution
// (the L[i][k] reads are the same address for every lane: LDS broadcasts), then column c of G = L^-T L^-1 against the LDS copy of
// L^-1, then column c of W G.  Fully unrolled over PMAX so that x and g stay in registers.
// block = 64 * nwaves, 2 * nwaves matrices per block, dynamic LDS = 2 * nwaves * bin_blocks_coop_doubles(p) doubles.
inline size_t bin_blocks_coop_doubles(int p) { return (size_t)2 * p * (p | 1) + (size_t)p * p; }
template <int PMAX>
__global__ __launch_bounds__(256) void bin_blocks_coop_kernel(const double* __restrict__ W, long long sW, double* __restrict__ G,
                                                              double* __restrict__ Wt, long long sO, int T, int p, double eps,
                                                              const int* __restrict__ slots, int nslots, double* __restrict__ ldet) {
  extern __shared__ double sm[];
  const int pp = p * p, ldp = p | 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
  const int per_block = (blockDim.x >> 6) * 2;
  const long long item = (long long)blockIdx.x * per_block + wave * 2 + sub;
  const bool have = item < (long long)nslots * T;
  const long long it2 = have ? item : 0;                     // (every lane walks the barriers; stores are guarded)
  const size_t slot = slots[it2 / T];
  const int t = (int)(it2 % T);
  double* A = sm + (size_t)(wave * 2 + sub) * (2 * p * ldp + pp);
  double* X = A + p * ldp;
  double* Wc = X + p * ldp;
  const double* w = W + slot * sW + (size_t)t * pp;
  for (int e = c; e < pp; e += 32) {
    const double v = w[e];
    const int i = e / p, j = e - i * p;
    Wc[e] = v;
    A[i * ldp + j] = eps * v + (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  // Cholesky, right-looking: lane c scales its entry of column j and updates its row
  double logdet = 0.0;
  for (int j = 0; j < p; ++j) {
    const double ajj = A[j * ldp + j];
    logdet += log(ajj);
    const double d = sqrt(ajj);
    __syncthreads();
    if (c == j) A[j * ldp + j] = d;
    else if (c > j && c < p) A[c * ldp + j] /= d;
    __syncthreads();
    if (c > j && c < p) {
      const double lcj = A[c * ldp + j];
      for (int k = j + 1; k <= c; ++k) A[c * ldp + k] -= lcj * A[k * ldp + j];
    }
    __syncthreads();
  }
  // x = column c of L^-1 (entries i >= c)
  double x[PMAX];
#pragma unroll
  for (int i = 0; i < PMAX; ++i) {
    x[i] = 0.0;
    if (i < p) {
      double s2 = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s2 -= A[i * ldp + k] * x[k];
      x[i] = (i >= c) ? s2 / A[i * ldp + i] : 0.0;
    }
  }
  if (c < p) {
#pragma unroll
    for (int i = 0; i < PMAX; ++i)
      if (i < p) X[c * ldp + i] = x[i];
  }
  __syncthreads();
  // g = column c of G = L^-T L^-1:  G[a][c] = sum_i Linv[i][a] Linv[i][c]
  double g[PMAX];
#pragma unroll
  for (int a = 0; a < PMAX; ++a) {
    g[a] = 0.0;
    if (a < p) {
      double s2 = 0.0;
#pragma unroll
      for (int i = 0; i < PMAX; ++i)
        if (i < p) s2 += X[a * ldp + i] * x[i];
      g[a] = s2;
    }
  }
  if (have && c < p) {
    double* go = G + slot * sO + (size_t)t * pp;
    double* wt = Wt + slot * sO + (size_t)t * pp;
#pragma unroll
    for (int a = 0; a < PMAX; ++a) {
      if (a < p) {
        go[a * p + c] = g[a];
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < PMAX; ++k)
          if (k < p) s2 += Wc[a * p + k] * g[k];
        wt[a * p + c] = s2;
      }
    }
    if (ldet && c == 0) ldet[item] = logdet;
  }
}

// The same in registers for p <= PW <= 10: the LDS version above keeps 2 p^2 + 1 doubles per thread in LDS, which at p = 10
// leaves 29 threads per workgroup and 3 waves per CU.  Here every loop is unrolled over the packed lower triangles, so
// the factor, its inverse and G live in VGPRs (static indices only).
// One thread per (slot, bin), one wave per workgroup.  The wave's 64 blocks are contiguous runs of p^2 doubles (per slot): they are
// copied into LDS with consecutive lanes on consecutive addresses, every thread then takes its own block from LDS (row stride
// p^2 + 1: conflict-free), and Wt and G go back out through the same LDS image.  (Threads used to read and write their blocks in
// global memory directly, 64 different cache lines per instruction: 1.26 ms and twice the algorithmic bytes at config 3.)
// grid = ceil(nslots*T / BBR_MPB), block = 64.
constexpr int BBR_TPB = 64;     // threads per workgroup (one wave: all of them copy)
constexpr int BBR_MPB = 32;     // blocks per workgroup (the first BBR_MPB lanes compute): 26 KB of LDS, six workgroups per CU keep enough loads in flight
template <int PW>
__global__ __launch_bounds__(BBR_TPB) void bin_blocks_reg_kernel(const double* __restrict__ W, long long sW, double* __restrict__ G,
                                                                 double* __restrict__ Wt, long long sO, int T, int p, double eps,
                                                                 const int* __restrict__ slots, int nslots, double* __restrict__ ldet) {
  constexpr int NP = PW * (PW + 1) / 2;
  constexpr int LD = PW * PW + 1;
  __shared__ double buf[BBR_MPB * LD];
  __shared__ long long in_off[BBR_MPB], out_off[BBR_MPB];
  const int tid = threadIdx.x;
  const long long total = (long long)nslots * T;
  const long long item0 = (long long)blockIdx.x * BBR_MPB;
  const int nitems = (int)((total - item0 < BBR_MPB) ? total - item0 : BBR_MPB);
  const long long item = item0 + tid;
  const int pp = p * p;
  if (tid < nitems) {
    const int slot = slots[item / T];
    const int t = (int)(item % T);
    in_off[tid] = (long long)slot * sW + (long long)t * pp;
    out_off[tid] = (long long)slot * sO + (long long)t * pp;
  }
  __syncthreads();
  // element e of the tile = entry idx of block m: (m, idx) advance without divisions
  auto copy_in = [&](const double* src, const long long* off) {
    int m = 0, idx = tid;
    while (idx >= pp) { idx -= pp; ++m; }
    for (; m < nitems;) {
      buf[m * LD + idx] = src[off[m] + idx];
      idx += BBR_TPB;
      while (idx >= pp) { idx -= pp; ++m; }
    }
  };
  auto copy_out = [&](double* dst, const long long* off) {
    int m = 0, idx = tid;
    while (idx >= pp) { idx -= pp; ++m; }
    for (; m < nitems;) {
      dst[off[m] + idx] = buf[m * LD + idx];
      idx += BBR_TPB;
      while (idx >= pp) { idx -= pp; ++m; }
    }
  };
  copy_in(W, in_off);
  __syncthreads();
  const bool live = tid < nitems;
  double* w = buf + (tid < BBR_MPB ? tid : 0) * LD;                     // this thread's block (W, then Wt row by row, then G)
  double L[NP];                                   // A = I + eps W (lower) -> Cholesky factor -> its inverse
#pragma unroll
  for (int i = 0; i < PW; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = ((live && i < p && j < p) ? eps * w[i * p + j] : 0.0) + (i == j ? 1.0 : 0.0);
  double logdet = 0.0;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    double dj = L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int m = 0; m < j; ++m) dj -= L[j * (j + 1) / 2 + m] * L[j * (j + 1) / 2 + m];
    logdet += log(dj);
    dj = sqrt(dj);
    L[j * (j + 1) / 2 + j] = dj;
#pragma unroll
    for (int i = j + 1; i < PW; ++i) {
      double v = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int m = 0; m < j; ++m) v -= L[i * (i + 1) / 2 + m] * L[j * (j + 1) / 2 + m];
      L[i * (i + 1) / 2 + j] = v / dj;
    }
  }
  if (ldet && live) ldet[item] = logdet;
  // invert L in place, column by column (column j reads L[i][m], m >= j, and L[i][i], i > j: still un-inverted then)
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    L[j * (j + 1) / 2 + j] = 1.0 / L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int i = j + 1; i < PW; ++i) {
      double v = 0.0;
#pragma unroll
      for (int m = j; m < i; ++m) v -= L[i * (i + 1) / 2 + m] * L[m * (m + 1) / 2 + j];
      L[i * (i + 1) / 2 + j] = v / L[i * (i + 1) / 2 + i];
    }
  }
  // G = L^-T L^-1 (lower)
  double Gm[NP];
#pragma unroll
  for (int i = 0; i < PW; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double v = 0.0;
#pragma unroll
      for (int m = i; m < PW; ++m) v += L[m * (m + 1) / 2 + i] * L[m * (m + 1) / 2 + j];
      Gm[i * (i + 1) / 2 + j] = v;
    }
  // Wt = W G, row by row over this thread's W (row i of W is not needed once row i of Wt is there)
  if (live) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      double wrow[PW];
#pragma unroll
      for (int m = 0; m < PW; ++m) wrow[m] = (i < p && m < p) ? w[i * p + m] : 0.0;
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < PW; ++m) v += wrow[m] * ((m >= j) ? Gm[m * (m + 1) / 2 + j] : Gm[j * (j + 1) / 2 + m]);
        if (i < p && j < p) w[i * p + j] = v;
      }
    }
  }
  __syncthreads();
  copy_out(Wt, out_off);
  __syncthreads();
  if (live) {
#pragma unroll
    for (int i = 0; i < PW; ++i)
#pragma unroll
      for (int j = 0; j < PW; ++j)
        if (i < p && j < p) w[i * p + j] = (j <= i) ? Gm[i * (i + 1) / 2 + j] : Gm[j * (j + 1) / 2 + i];
  }
  __syncthreads();
  copy_out(G, out_off);
}

// B = I + F^T Wt F (lower triangle) for the low-rank engine.  B is cut into 16-wide blocks that never straddle a
// latent (ranks are padded to 16): block (bi, bj) is sum_t F_ka[t][a] Wt_t[ka][kb] F_kb[t][b].
// One workgroup (4 waves, 2x2 MFMA tiles each) per 64x64 tile of B: 32-bin chunks of the two F column panels and
// of the 4x4 table of per-bin weights are staged in LDS, the weight scales the A fragment in registers.  The A
// operand is the column side, so the accumulator's lanes run along rows of B and the stores are contiguous
// (Wt_t is symmetric).  grid = (tile pairs, slots), block = 256.
constexpr int AB_TK = 32;     // bins per LDS chunk
constexpr int AB_LD = 81;     // LDS row stride of a staged panel (64 columns + pad; odd*... keeps writes conflict-free)
constexpr int AB_SLOTS = 2;   // slots per workgroup: the F panels are the same for every slot, only the weights differ,
                              // so one staged chunk (one exposed global-load latency) feeds AB_SLOTS x 32 MFMAs per wave
                              // (measured 2: -23 %; 3 and 4 lose resident workgroups to the extra accumulators)
// (TB = double, or float for the mixed-precision dual evaluation: F32 panels, weights rounded on load, FP32 matrix cores - twice the rate -
// and B leaves as floats, ready for the single-precision factorisation)
template <typename TB>
__global__ __launch_bounds__(256) void assemble_b_kernel_t(TB* __restrict__ Bm, long long sB, int ldb, int nblk,
                                                           const TB* __restrict__ F, int Tf, int T, int p,
                                                         const int* __restrict__ blk_lat, const int* __restrict__ blk_col,
                                                         const double* __restrict__ Wt, long long sW, const int* __restrict__ slots,
                                                         int nslots, const int* __restrict__ cmap = nullptr) {
  // cmap (round 6, compact rank offsets: core.hip build_lowrank): the tiles are walked in the PADDED index space (16-blocks that never straddle a
  // latent); entry (row, col) is stored at (cmap[row], cmap[col]) of the compact r x r system, padding rows and columns (cmap < 0) are not stored
  // (pad_identity_kernel writes the identity behind the last latent)
  using V4 = typename GemmVec<TB>::v4;
  __shared__ TB FR[AB_TK * AB_LD];                   // row-side panel  [bin][column]
  __shared__ TB FC[AB_TK * AB_LD];                   // column-side panel
  __shared__ TB WL[AB_SLOTS][AB_TK * 16];            // weights [slot][bin][row block * 4 + column block]
  __shared__ int lat_r[4], lat_c[4], col_r[4], col_c[4];
  int bi = 0, rem = blockIdx.x;
  while (rem > bi) { rem -= bi + 1; ++bi; }
  const int bj = rem;                                    // bj <= bi, in units of 64
  const int s_first = blockIdx.y * AB_SLOTS;
  const int ns = min(AB_SLOTS, nslots - s_first);        // block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  if (tid < 4) {
    lat_r[tid] = blk_lat[bi * 4 + tid]; col_r[tid] = blk_col[bi * 4 + tid];
    lat_c[tid] = blk_lat[bj * 4 + tid]; col_c[tid] = blk_col[bj * 4 + tid];
  }
  __syncthreads();
  const int lt = tid & 31, lc = tid >> 5;                // staging: bin lt, columns lc + 8 i
  const TB* srcR[8];
  const TB* srcC[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cc = lc + 8 * i, blk = cc >> 4;
    srcR[i] = lat_r[blk] < 0 ? nullptr : F + (size_t)lat_r[blk] * Tf * Tf + (size_t)(col_r[blk] + (cc & 15)) * Tf + lt;
    srcC[i] = lat_c[blk] < 0 ? nullptr : F + (size_t)lat_c[blk] * Tf * Tf + (size_t)(col_c[blk] + (cc & 15)) * Tf + lt;
  }
  // weights: element e = tid + 256 i of a slot's [bin][16] table; the same (bin, block pair) for every slot
  long long woff[2];
  int wt_bin[2];
  bool wok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i, pr = e & 15;
    const int la = lat_r[pr >> 2], lb = lat_c[pr & 3];
    wt_bin[i] = e >> 4;
    wok[i] = !(la < 0 || lb < 0);
    woff[i] = wok[i] ? (long long)wt_bin[i] * p * p + (long long)la * p + lb : 0;
  }
  const double* wbase[AB_SLOTS];
#pragma unroll
  for (int sl = 0; sl < AB_SLOTS; ++sl) wbase[sl] = Wt + (size_t)slots[s_first + (sl < ns ? sl : 0)] * sW;
  V4 acc[AB_SLOTS][2][2];
#pragma unroll
  for (int sl = 0; sl < AB_SLOTS; ++sl)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[sl][i][j] = V4{(TB)0, (TB)0, (TB)0, (TB)0};

  // register staging of the next chunk: its global loads are in flight while the current chunk is multiplied
  TB pr[8], pc[8], pw[AB_SLOTS][2];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      pr[i] = srcR[i] ? srcR[i][t0] : (TB)0;
      pc[i] = srcC[i] ? srcC[i][t0] : (TB)0;
    }
#pragma unroll
    for (int sl = 0; sl < AB_SLOTS; ++sl)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        pw[sl][i] = (sl < ns && wok[i] && t0 + wt_bin[i] < T) ? (TB)wbase[sl][woff[i] + (long long)t0 * p * p] : (TB)0;
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += AB_TK) {                // panel rows T..Tf are zero, weights are guarded
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      FR[lt * AB_LD + lc + 8 * i] = pr[i];
      FC[lt * AB_LD + lc + 8 * i] = pc[i];
    }
#pragma unroll
    for (int sl = 0; sl < AB_SLOTS; ++sl)
#pragma unroll
      for (int i = 0; i < 2; ++i) WL[sl][tid + 256 * i] = pw[sl][i];
    __syncthreads();
    if (t0 + AB_TK < T) fetch(t0 + AB_TK);
#pragma unroll
    for (int kk = 0; kk < AB_TK / 4; ++kk) {
      const int tt = kk * 4 + l4;
      const TB r0 = FR[tt * AB_LD + wr * 32 + l15], r1 = FR[tt * AB_LD + wr * 32 + 16 + l15];
      const TB c0 = FC[tt * AB_LD + wc * 32 + l15], c1 = FC[tt * AB_LD + wc * 32 + 16 + l15];
#pragma unroll
      for (int sl = 0; sl < AB_SLOTS; ++sl) {
        if (sl >= ns) break;
        const TB* wl = WL[sl] + tt * 16 + (2 * wr) * 4 + 2 * wc;
        acc[sl][0][0] = gemm_mfma16(c0 * wl[0], r0, acc[sl][0][0]);
        acc[sl][0][1] = gemm_mfma16(c1 * wl[1], r0, acc[sl][0][1]);
        acc[sl][1][0] = gemm_mfma16(c0 * wl[4], r1, acc[sl][1][0]);
        acc[sl][1][1] = gemm_mfma16(c1 * wl[5], r1, acc[sl][1][1]);
      }
    }
    __syncthreads();
  }
  // accumulator: i (A side, column of B) = l4 + 4 r in FP64, 4 l4 + r in FP32 (see gemm.h) ; j (B side, row of B) = l15
#pragma unroll
  for (int sl = 0; sl < AB_SLOTS; ++sl) {
    if (sl >= ns) break;
    TB* out = Bm + (size_t)slots[s_first + sl] * sB;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = bi * 64 + wr * 32 + ri * 16 + l15;
          const int col = bj * 64 + wc * 32 + ci * 16 + (sizeof(TB) == 8 ? l4 + 4 * r : 4 * l4 + r);
          if (cmap) {
            const int rc = cmap[row], cc = cmap[col];
            if (row >= col && rc >= 0 && cc >= 0) out[(size_t)cc * ldb + rc] = acc[sl][ri][ci][r] + (row == col ? (TB)1 : (TB)0);
          } else if (row >= col) out[(size_t)col * ldb + row] = acc[sl][ri][ci][r] + (row == col ? (TB)1 : (TB)0);
        }
  }
}

// The identity behind the last latent of the compact r x r system (rows [r0, r1) of the lower triangle; beyond `nact` - rows the blocked
// factorisation never updates - only inside their own diagonal block of `nb` rows).  Column-major: a workgroup takes 4 columns, its threads run
// along the rows.  grid = (ceil(r1 / 4), slots), block = 256.
template <typename TB>
__global__ __launch_bounds__(256) void pad_identity_kernel(TB* __restrict__ Bm, long long sB, int ldb, int r0, int r1, int nact, int nb, const int* __restrict__ slots) {
  TB* out = Bm + (size_t)slots[blockIdx.y] * sB;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= r1) return;
  for (int row = max(r0, col) + (threadIdx.x & 63); row < r1; row += 64)
    if (row < nact || col >= (row / nb) * nb) out[(size_t)col * ldb + row] = (col == row) ? (TB)1 : (TB)0;
}

// vsm[t] <- eps*G_t + G_t * Bt_t * G_t in place (Bt_t already in vsm, trial indexed).  One thread per matrix ROW:
// a block stages the V and G blocks of 256/PW consecutive bins of one slot in LDS (coalesced reads, re-laid out to
// a zero-padded PW x PW stride so the loops are unrolled without guards), thread (bin, i) forms row i of Bt G, then
// row i of eps G + G (Bt G), and writes its row back.  grid = (ceil(T / (256/PW)), nslots), block = 256, p <= PW.
template <int PW>
__global__ __launch_bounds__(256) void vsm_finish_kernel(double* __restrict__ vsm, const double* __restrict__ G, long long sG, int T, int p,
                                                         double eps, const int* __restrict__ slots, const int* __restrict__ trial_of_slot) {
  constexpr int PP = PW * PW, BT = 256 / PW;
  __shared__ double Vs[BT * PP];
  __shared__ double Gs[BT * PP];
  const int pp = p * p;
  const int slot = slots[blockIdx.y];
  const int t0 = blockIdx.x * BT;
  const int nt = min(BT, T - t0);
  const double* gbase = G + (size_t)slot * sG + (size_t)t0 * pp;
  double* vbase = vsm + ((size_t)trial_of_slot[slot] * T + t0) * pp;
  if (p < PW)
    for (int e = threadIdx.x; e < BT * PP; e += 256) { Vs[e] = 0.0; Gs[e] = 0.0; }
  __syncthreads();
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const int t = e / pp, idx = e - t * pp, i = idx / p, j = idx - i * p;
    Vs[t * PP + i * PW + j] = vbase[e];
    Gs[t * PP + i * PW + j] = gbase[e];
  }
  __syncthreads();
  const int bt = threadIdx.x / PW, i = threadIdx.x - bt * PW;
  const bool live = bt < nt && i < p;
  const double* g = Gs + bt * PP;
  double* v = Vs + bt * PP;
  double row[PW], acc[PW];
  if (live) {
#pragma unroll
    for (int m = 0; m < PW; ++m) row[m] = v[i * PW + m];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      double sacc = 0.0;
#pragma unroll
      for (int m = 0; m < PW; ++m) sacc += row[m] * g[m * PW + j];
      acc[j] = sacc;
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) v[i * PW + j] = acc[j];       // row i of Bt G (only this thread reads row i of V)
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int m = 0; m < PW; ++m) row[m] = g[i * PW + m];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      double sacc = eps * row[j];
#pragma unroll
      for (int m = 0; m < PW; ++m) sacc += row[m] * v[m * PW + j];
      if (j < p) vbase[(size_t)bt * pp + i * p + j] = sacc;
    }
  }
}

// Mixing pass over Yt (n x rpad, ld = ldy) of the low-rank engine: every p-vector y = Yt[(.,t), b] is replaced by G_t y
// - afterwards rows (k,.) of the slab ARE Ymix_k, the GEMM operand of post_vsmGP_k = eps diag + Ymix_k Ymix_k^T - AND
// the per-bin blocks post_vsm[t] = eps G_t + sum_b (G_t y_b)(G_t y_b)^T are accumulated on the way (G_t symmetric, so
// this is eps G + G (Y_t^T Y_t) G: what post_vsm_kernel + vsm_finish_kernel compute from the unmixed slab).
// A block owns 64 bins (lanes) of one slot and ALL columns (its 4 waves take interleaved columns); G_t sits in LDS
// (odd stride: each lane reads its own block conflict-free), the p(p+1)/2 accumulators in registers; the waves'
// partial sums are folded into the LDS copy of G in turn (-> eps G + sum), which then leaves as one contiguous run.
// grid = (ceil(T/64), nslots), block = 256, p <= PW <= 16.
template <int PW>
__global__ __launch_bounds__(256) void mix_vsm_kernel(double* __restrict__ Yt, long long sY, int ldy, const double* __restrict__ G, long long sG,
                                                      int T, int p, int rpad, double eps, double* __restrict__ vsm,
                                                      const int* __restrict__ slots, const int* __restrict__ trial_of_slot,
                                                      const int* __restrict__ roff, int col_tile, int ts) {
  // roff (may be null): rank offsets of the latents; rows (k,.) of Yt are identically zero - and were not written - left of
  // column (roff[k] / col_tile) * col_tile (the factor L^-T is upper triangular).  ts: row stride between latents in the slab (>= T)
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
    double* y = Yt + (size_t)slot * sY + t0 + lane;
    for (int b = wave; b < rpad; b += 4) {
      // Beyond 10 latents the LDS reads of G stay inside the trip: hoisted out of the loop (what the compiler does by itself) they
      // take 2 PW^2 registers, i.e. scratch.  Up to 10 the hoisted form is kept - the pass runs at the rate of the slab's
      // read + write either way (measured: more waves per SIMD or more loads in flight per wave change nothing).
      if constexpr (PW > 10) asm volatile("" ::: "memory");
      double v[PW], m[PW];
#pragma unroll
      for (int k = 0; k < PW; ++k) v[k] = (k < p && b >= c0[k]) ? y[(size_t)b * ldy + (size_t)k * ts] : 0.0;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        double s2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < PW; ++kk) s2 += g[k * PW + kk] * v[kk];
        m[k] = s2;
        if (k < p) y[(size_t)b * ldy + (size_t)k * ts] = s2;
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

// vsmGP scatter for the low-rank engine: dst = mirror(src) + eps*G_t[k][k] on the diagonal
inline __global__ void scatter_vsmgp_lr_kernel(const double* __restrict__ src, long long sSrc, int lds, double* __restrict__ dst, int T, int p, int k,
                                        const double* __restrict__ G, long long sG, double eps, const int* __restrict__ trial_of_slot) {
  const int slot = blockIdx.y;
  const size_t r = trial_of_slot[slot];
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (size_t)T * T) {
    const size_t a = e % T, b = e / T;
    double v = (a >= b) ? src[(size_t)slot * sSrc + b * lds + a] : src[(size_t)slot * sSrc + a * lds + b];
    if (a == b) v += eps * G[(size_t)slot * sG + a * p * p + (size_t)k * p + k];
    dst[(r * p + k) * T * T + e] = v;
  }
}

// block-diagonal F (n x rpad, ld) and F^T (rpad x n, ldt = rpad) from the per-latent factors; grid = (rk_max, p)
inline __global__ void build_fbig_kernel(const double* __restrict__ F, int Tf, int T, const int* __restrict__ roff, double* __restrict__ Fbig,
                                  int ld, double* __restrict__ FTbig, int ldt) {
  const int k = blockIdx.y, a = blockIdx.x;
  const int r0 = roff[k], r1 = roff[k + 1];
  if (a >= r1 - r0) return;
  const double* col = F + (size_t)k * Tf * Tf + (size_t)a * Tf;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const double v = col[t];
    Fbig[(size_t)(r0 + a) * ld + (size_t)k * T + t] = v;
    FTbig[((size_t)k * T + t) * ldt + r0 + a] = v;
  }
}

// out[slot][(k,t)] = sum_k' Gb[t][k][k'] * (scale * a[slot][(k',t)] + b[slot][(k',t)])   (b may be null)
// Gb is ONE set of per-bin p x p blocks shared by all slots: a block stages the blocks of 64 bins in LDS once
// (zero-padded PW x PW, odd stride) and its 4 waves walk APPLY_BIN_SLOTS slots with lanes = bins.
// grid = (ceil(T/64), ceil(nslots / APPLY_BIN_SLOTS)), block = 256, p <= PW.
constexpr int APPLY_BIN_SLOTS = 16;
template <int PW>
__global__ __launch_bounds__(256) void apply_bin_kernel(const double* __restrict__ Gb, const double* __restrict__ A, const double* __restrict__ B2,
                                                        double scale, double* __restrict__ out, long long sV, int T, int p, int nslots) {
  constexpr int PP = PW * PW, LD = PP + 1;
  __shared__ double Gs[64 * LD];
  const int pp = p * p;
  const int t0 = blockIdx.x * 64;
  const int nt = min(64, T - t0);
  if (p < PW)
    for (int e = threadIdx.x; e < 64 * LD; e += 256) Gs[e] = 0.0;
  __syncthreads();
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const int t = e / pp, idx = e - t * pp, i = idx / p, j = idx - i * p;
    Gs[t * LD + i * PW + j] = Gb[(size_t)t0 * pp + e];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane >= nt) return;
  const int t = t0 + lane;
  const double* g = Gs + lane * LD;
  const int s_end = min(nslots, (int)(blockIdx.y + 1) * APPLY_BIN_SLOTS);
  for (int sl = blockIdx.y * APPLY_BIN_SLOTS + wave; sl < s_end; sl += 4) {
    const double* a = A + (size_t)sl * sV + t;
    const double* b = B2 ? B2 + (size_t)sl * sV + t : nullptr;
    double* o = out + (size_t)sl * sV + t;
    double v[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      v[k] = 0.0;
      if (k < p) v[k] = scale * a[(size_t)k * T] + (b ? b[(size_t)k * T] : 0.0);
    }
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int kk = 0; kk < PW; ++kk) acc += g[k * PW + kk] * v[kk];
      if (k < p) o[(size_t)k * T] = acc;
    }
  }
}

// ---- wide latent states (16 < p <= 32) of the low-rank engine ---------------------------------------------------
// The per-bin p x p blocks no longer fit 64 bins at a time in LDS and p^2 accumulators no longer fit a lane, so the
// three per-bin kernels above get a second shape: a block owns `bins` (8 or 4) consecutive bins, 32 threads per bin,
// thread (bin, k) produces component / row k; the p-vectors being multiplied travel through LDS.  Runtime p, no
// template; dynamic LDS = wide_lds_bytes(p, bins, arrays).
constexpr int WIDE_MAX = 32;
inline int wide_bins(int p) { return p <= 22 ? 8 : 4; }
inline size_t wide_lds_bytes(int p, int bins, int blocks_per_bin) {
  return ((size_t)bins * blocks_per_bin * (p * p + 1) + 2 * (size_t)bins * WIDE_MAX) * sizeof(double);
}

// apply_bin_kernel for wide p.  grid = (ceil(T/bins), ceil(nslots / APPLY_BIN_SLOTS)), block = bins*32
inline __global__ void apply_bin_wide_kernel(const double* __restrict__ Gb, const double* __restrict__ A, const double* __restrict__ B2, double scale,
                                      double* __restrict__ out, long long sV, int T, int p, int nslots, int bins) {
  extern __shared__ double wsm[];
  const int pp = p * p, LD = pp + 1;
  double* Gs = wsm;
  double* vs = wsm + (size_t)bins * LD;
  const int t0 = blockIdx.x * bins;
  const int nt = min(bins, T - t0);
  for (int e = threadIdx.x; e < nt * pp; e += blockDim.x) {
    const int t = e / pp;
    Gs[t * LD + (e - t * pp)] = Gb[(size_t)t0 * pp + e];
  }
  __syncthreads();
  const int bt = threadIdx.x >> 5, k = threadIdx.x & 31;
  const bool live = bt < nt && k < p;
  const int t = t0 + bt;
  const double* g = Gs + bt * LD + k * p;
  const int s_end = min(nslots, (int)(blockIdx.y + 1) * APPLY_BIN_SLOTS);
  for (int sl = blockIdx.y * APPLY_BIN_SLOTS; sl < s_end; ++sl) {
    if (live) {
      const size_t off = (size_t)sl * sV + (size_t)k * T + t;
      vs[bt * WIDE_MAX + k] = scale * A[off] + (B2 ? B2[off] : 0.0);
    }
    __syncthreads();
    if (live) {
      double acc = 0.0;
      for (int kk = 0; kk < p; ++kk) acc += g[kk] * vs[bt * WIDE_MAX + kk];
      out[(size_t)sl * sV + (size_t)k * T + t] = acc;
    }
    __syncthreads();
  }
}

// apply_bin_wide_kernel with the lanes along the bins (see mix_vsm_wide2_kernel below: the same layout): 64 bins x 4 waves, wave w owns rows a = w + 4 i of the
// shared blocks (in registers), the slots' vectors are exchanged through LDS.  grid = (ceil(T / 64), ceil(nslots / APPLY_BIN_SLOTS)), block = 256, p <= PW = 20.
template <int PW>
__global__ __launch_bounds__(256, 1) void apply_bin_wide2_kernel(const double* __restrict__ Gb, const double* __restrict__ A, const double* __restrict__ B2, double scale,
                                                                 double* __restrict__ out, long long sV, int T, int p, int nslots, double* __restrict__ sink) {
  constexpr int R = PW / 4;
  __shared__ double vs[2][PW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const int tc = t < T ? t : T - 1;
  const int pp = p * p;
  const double* gsrc = Gb + (size_t)tc * pp;
  double g[R][PW];
  size_t roff[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int a = wave + 4 * i;
    const int ac = a < p ? a : 0;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const double v = gsrc[ac * p + (k < p ? k : 0)];
      g[i][k] = (a < p && k < p) ? v : 0.0;
    }
    roff[i] = (size_t)ac * T + tc;
  }
  for (int e = threadIdx.x; e < 2 * PW * 64; e += 256) (&vs[0][0][0])[e] = 0.0;
  __syncthreads();
  const int s0 = blockIdx.y * APPLY_BIN_SLOTS, s_end = min(nslots, s0 + APPLY_BIN_SLOTS);
  auto request = [&](int sl, double (&y)[R]) {
    const size_t base = (size_t)(sl < s_end ? sl : s_end - 1) * sV;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const double av = A[base + roff[i]];
      const double bv = B2 ? B2[base + roff[i]] : 0.0;
      y[i] = scale * av + bv;
    }
  };
  double yv[R], yn[R];
  request(s0, yv);
#pragma unroll 1
  for (int sl = s0; sl < s_end; ++sl) {
    const int par = (sl - s0) & 1;
    request(sl + 1, yn);
#pragma unroll
    for (int i = 0; i < R; ++i) vs[par][wave + 4 * i][lane] = (wave + 4 * i < p) ? yv[i] : 0.0;
    __syncthreads();                                   // (one barrier per slot: the buffers alternate)
    double m[R];
#pragma unroll
    for (int i = 0; i < R; ++i) m[i] = 0.0;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const double yk = vs[par][k][lane];
#pragma unroll
      for (int i = 0; i < R; ++i) m[i] += g[i][k] * yk;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      double* dst = (wave + 4 * i < p) ? out + (size_t)sl * sV + roff[i] : sink + lane;
      *dst = m[i];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) yv[i] = yn[i];
  }
}

// mix_vsm_kernel for wide p: y <- G_t y for every column of the slab, post_vsm[t] = eps G_t + sum_b (G_t y_b)(G_t y_b)^T.
// Thread (bin, a) owns row a of the accumulated block (columns c <= a).  grid = (ceil(T/bins), nslots), block = bins*32
inline __global__ void mix_vsm_wide_kernel(double* __restrict__ Yt, long long sY, int ldy, const double* __restrict__ G, long long sG, int T, int p,
                                    int rpad, double eps, double* __restrict__ vsm, const int* __restrict__ slots,
                                    const int* __restrict__ trial_of_slot, int bins, int ts) {
  extern __shared__ double wsm[];
  const int pp = p * p, LD = pp + 1;
  double* Gs = wsm;
  double* vs = wsm + (size_t)bins * LD;
  double* ms = vs + (size_t)bins * WIDE_MAX;
  const int slot = slots[blockIdx.y];
  const int t0 = blockIdx.x * bins;
  const int nt = min(bins, T - t0);
  const double* gbase = G + (size_t)slot * sG + (size_t)t0 * pp;
  for (int e = threadIdx.x; e < nt * pp; e += blockDim.x) {
    const int t = e / pp;
    Gs[t * LD + (e - t * pp)] = gbase[e];
  }
  __syncthreads();
  const int bt = threadIdx.x >> 5, a = threadIdx.x & 31;
  const bool live = bt < nt && a < p;
  const double* g = Gs + bt * LD + a * p;
  double acc[WIDE_MAX];
#pragma unroll
  for (int c2 = 0; c2 < WIDE_MAX; ++c2) acc[c2] = 0.0;
  double* y = Yt + (size_t)slot * sY + (size_t)a * ts + t0 + bt;
  for (int b = 0; b < rpad; ++b) {
    if (live) vs[bt * WIDE_MAX + a] = y[(size_t)b * ldy];
    __syncthreads();
    double m = 0.0;
    if (live) {
      for (int kk = 0; kk < p; ++kk) m += g[kk] * vs[bt * WIDE_MAX + kk];
      y[(size_t)b * ldy] = m;
      ms[bt * WIDE_MAX + a] = m;
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int c2 = 0; c2 < WIDE_MAX; ++c2)
        if (c2 <= a) acc[c2] += m * ms[bt * WIDE_MAX + c2];
    }
  }
  if (live) {
    double* v = vsm + ((size_t)trial_of_slot[slot] * T + t0 + bt) * pp;
#pragma unroll
    for (int c2 = 0; c2 < WIDE_MAX; ++c2)
      if (c2 <= a) {
        const double val = eps * g[c2] + acc[c2];
        v[a * p + c2] = val;
        v[c2 * p + a] = val;
      }
  }
}

// The same pass with lanes along the BINS (the contiguous index of the slab's rows): mix_vsm_wide_kernel puts the latents on the lanes, so every 8-byte access of a
// wave is a line of its own, and it meets two barriers per column with 8 bins per workgroup - 199 ms for the finalize pass of config 5 (82 GB moved at 0.4 TB/s).
// Here a workgroup is 64 bins x 4 waves; wave w owns rows a = w + 4 i of G_t (in registers: PW / 4 rows of PW entries) and of the accumulated block; per column
// every wave reads its rows of y (512-byte runs), the 4 waves exchange y and m = G_t y through LDS (double-buffered by column parity), the next column's loads
// are in flight meanwhile.  No branch around a load or a store in the loop: clamped addresses, bins past T repeat bin T - 1, rows past p write to `sink`
// (>= 64 doubles of scratch).  grid = (ceil(T / 64), nslots), block = 256, PW in {20, 24, 32} >= p.
// SPLIT: the pass of the split accumulation (split.h) instead - Yt stays, the correction y - G_t y goes to D in single precision (row (a, t) at a ts + t, column
// stride ldd, slot stride sD floats); `sink` then takes the loads of the rows past p.
template <int PW, bool SPLIT>
__global__ __launch_bounds__(256, 1) void mix_vsm_wide2_kernel(double* __restrict__ Yt, long long sY, int ldy, const double* __restrict__ G, long long sG, int T,
                                                               int p, int ncol, double eps, double* __restrict__ vsm, const int* __restrict__ slots,
                                                               const int* __restrict__ trial_of_slot, int ts, double* __restrict__ sink,
                                                               float* __restrict__ D, long long sD, int ldd) {
  constexpr int R = PW / 4;
  __shared__ double vs[2][PW][64];
  __shared__ double ms[2][PW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = slots[blockIdx.y];
  const int t = blockIdx.x * 64 + lane;
  const int tc = t < T ? t : T - 1;
  const int pp = p * p;
  const double* gsrc = G + (size_t)slot * sG + (size_t)tc * pp;
  double g[R][PW];
  double* yp[R];
  float* dp[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int a = wave + 4 * i;
    const int ac = a < p ? a : 0;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const double v = gsrc[ac * p + (k < p ? k : 0)];
      g[i][k] = (a < p && k < p) ? v : 0.0;
    }
    yp[i] = (a < p) ? Yt + (size_t)slot * sY + (size_t)a * ts + tc : sink + lane;
    dp[i] = (SPLIT && a < p) ? D + (size_t)slot * sD + (size_t)a * ts + tc : reinterpret_cast<float*>(sink) + lane;
  }
  // rows >= p of the exchange buffers are read (against zeros of g, and as m of rows that are never stored): keep them finite
  for (int e = threadIdx.x; e < 2 * PW * 64; e += 256) { (&vs[0][0][0])[e] = 0.0; (&ms[0][0][0])[e] = 0.0; }
  // acc[i][c]: row a = wave + 4 i against columns c < 4 (i + 1) (c <= a needs no more)
  double acc[R][4 * R];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int c2 = 0; c2 < 4 * R; ++c2) acc[i][c2] = 0.0;
  __syncthreads();
  // the next column of y is requested while this one is mixed (two register sets; a ring of 2 or 4 unrolled columns spills: G_t's rows hold 200 registers)
  auto request = [&](int b, double (&y)[R]) {
    const size_t o = (size_t)(b < ncol ? b : ncol - 1) * ldy;                  // (past the end: the last column again)
#pragma unroll
    for (int i = 0; i < R; ++i) y[i] = yp[i][(wave + 4 * i < p) ? o : 0];     // (address select, not a branch: the sink has one column)
  };
  auto column = [&](int b, const double (&yv)[R]) {
    const int par = b & 1;
#pragma unroll
    for (int i = 0; i < R; ++i) vs[par][wave + 4 * i][lane] = yv[i];
    __syncthreads();
    double m[R];
#pragma unroll
    for (int i = 0; i < R; ++i) m[i] = 0.0;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const double yk = vs[par][k][lane];
#pragma unroll
      for (int i = 0; i < R; ++i) m[i] += g[i][k] * yk;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int a = wave + 4 * i;
      if constexpr (SPLIT) dp[i][(a < p) ? (size_t)b * ldd : 0] = (float)(yv[i] - m[i]);
      else yp[i][(a < p) ? (size_t)b * ldy : 0] = m[i];
      ms[par][a][lane] = m[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int c2 = 0; c2 < 4 * (i + 1); ++c2) acc[i][c2] += m[i] * ms[par][c2][lane];
  };
  double yv[R], yn[R];
  request(0, yv);
#pragma unroll 1
  for (int b = 0; b < ncol; ++b) {
    request(b + 1, yn);
    column(b, yv);
#pragma unroll
    for (int i = 0; i < R; ++i) yv[i] = yn[i];
  }
  if (t >= T) return;
  double* v = vsm + ((size_t)trial_of_slot[slot] * T + t) * pp;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int a = wave + 4 * i;
    if (a >= p) continue;
#pragma unroll
    for (int c2 = 0; c2 < 4 * (i + 1); ++c2)
      if (c2 <= a) {
        const double val = eps * g[i][c2] + acc[i][c2];
        v[a * p + c2] = val;
        v[c2 * p + a] = val;
      }
  }
}

// vsm_finish_kernel for wide p: vsm[t] <- eps G_t + G_t Bt_t G_t in place.  Thread (bin, i) owns row i.
// grid = (ceil(T/bins), nslots), block = bins*32; LDS holds the V and the G blocks of the block's bins.
inline __global__ void vsm_finish_wide_kernel(double* __restrict__ vsm, const double* __restrict__ G, long long sG, int T, int p, double eps,
                                       const int* __restrict__ slots, const int* __restrict__ trial_of_slot, int bins) {
  extern __shared__ double wsm[];
  const int pp = p * p, LD = pp + 1;
  double* Gs = wsm;
  double* Vs = wsm + (size_t)bins * LD;
  const int slot = slots[blockIdx.y];
  const int t0 = blockIdx.x * bins;
  const int nt = min(bins, T - t0);
  const double* gbase = G + (size_t)slot * sG + (size_t)t0 * pp;
  double* vbase = vsm + ((size_t)trial_of_slot[slot] * T + t0) * pp;
  for (int e = threadIdx.x; e < nt * pp; e += blockDim.x) {
    const int t = e / pp;
    Gs[t * LD + (e - t * pp)] = gbase[e];
    Vs[t * LD + (e - t * pp)] = vbase[e];
  }
  __syncthreads();
  const int bt = threadIdx.x >> 5, i = threadIdx.x & 31;
  const bool live = bt < nt && i < p;
  const double* g = Gs + bt * LD;
  double* v = Vs + bt * LD;
  double row[WIDE_MAX];
  if (live) {
    // row i of Bt G (only this thread reads or writes row i of V)
#pragma unroll
    for (int j = 0; j < WIDE_MAX; ++j) {
      double sacc = 0.0;
      if (j < p)
        for (int m = 0; m < p; ++m) sacc += v[i * p + m] * g[m * p + j];
      row[j] = sacc;
    }
#pragma unroll
    for (int j = 0; j < WIDE_MAX; ++j)
      if (j < p) v[i * p + j] = row[j];
  }
  __syncthreads();
  if (live) {
    for (int j = 0; j < p; ++j) {
      double sacc = eps * g[i * p + j];
      for (int m = 0; m < p; ++m) sacc += g[i * p + m] * v[m * p + j];
      vbase[(size_t)bt * pp + i * p + j] = sacc;
    }
  }
}

// First and second moments of the spike counts over all (trial, bin) samples of the listed trials, EXACT in integer
// arithmetic: sum[i] = sum y_i, cross[i][j] = sum y_i y_j (what np.mean / np.cov of the concatenated raster are built
// from: Poisson-PCA initialiser util.py:528-533, spike-count diagnostics engine.py:487-492).
// A block owns one trial and one pair of 32-neuron tiles (ti >= tj); rows are packed 4 bins per 32-bit word in LDS
// and every 4-bin product sum is one v_dot4_u32_u8; per-trial partial sums (< 2^32 for T < 66000) are added to the
// 64-bit totals with integer atomics (order independent, so the result is deterministic).
// grid = (tile pairs, trials), block = 256.
constexpr int CM_TILE = 32, CM_BINS = 512, CM_LD = CM_BINS / 4 + 1;
// (counts above 255: y = lo + 256 hi with the bytes in two planes, so sum y_i y_j is four such passes - plane Ya on the row side, Yb on the
// column side, the 64-bit contribution scaled by cross_scale = 1 / 256 / 256 / 65536 and the row sums by sum_scale = 1 / 0 / 0 / 256)
inline __global__ __launch_bounds__(256) void count_moments_kernel(const uint8_t* __restrict__ Ya, const uint8_t* __restrict__ Yb, const int* __restrict__ trials,
                                                            int q, int T, unsigned long long* __restrict__ sum, unsigned long long* __restrict__ cross,
                                                            unsigned long long cross_scale, unsigned long long sum_scale) {
  __shared__ unsigned Wi[CM_TILE * CM_LD];
  __shared__ unsigned Wj[CM_TILE * CM_LD];
  int ti = 0, rem = blockIdx.x;
  while (rem > ti) { rem -= ti + 1; ++ti; }
  const int tj = rem;
  const uint8_t* Yr = Ya + (size_t)trials[blockIdx.y] * q * T;
  const uint8_t* Yc = Yb + (size_t)trials[blockIdx.y] * q * T;
  const int a = threadIdx.x >> 3, bg = (threadIdx.x & 7) * 4;
  unsigned acc[4] = {0u, 0u, 0u, 0u};
  unsigned acc_s = 0u;
  for (int t0 = 0; t0 < T; t0 += CM_BINS) {
    const int nb = min(CM_BINS, T - t0), nw = (nb + 3) / 4;
    __syncthreads();
    for (int e = threadIdx.x; e < CM_TILE * nw; e += 256) {
      const int r = e / nw, w = e - r * nw;
      unsigned vi = 0u, vj = 0u;
      const int ni = ti * CM_TILE + r, nj = tj * CM_TILE + r;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int t = 4 * w + b;
        if (t < nb) {
          if (ni < q) vi |= (unsigned)Yr[(size_t)ni * T + t0 + t] << (8 * b);
          if (nj < q) vj |= (unsigned)Yc[(size_t)nj * T + t0 + t] << (8 * b);
        }
      }
      Wi[r * CM_LD + w] = vi;
      Wj[r * CM_LD + w] = vj;
    }
    __syncthreads();
    for (int w = 0; w < nw; ++w) {
      const unsigned wa = Wi[a * CM_LD + w];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_udot4(wa, Wj[(bg + k) * CM_LD + w], acc[k], false);
      if (bg == 0) acc_s = __builtin_amdgcn_udot4(wa, 0x01010101u, acc_s, false);
    }
  }
  const int i = ti * CM_TILE + a;
  if (i >= q) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = tj * CM_TILE + bg + k;
    if (j < q && acc[k]) atomicAdd(&cross[(size_t)i * q + j], (unsigned long long)acc[k] * cross_scale);
  }
  if (ti == tj && bg == 0 && acc_s && sum_scale) atomicAdd(&sum[i], (unsigned long long)acc_s * sum_scale);
}

// counts: double (or uint16) [R][q][T] -> byte planes with validation (non-negative integers <= 65535): low bytes to lo, high bytes to hi
// when that plane exists; flags[0] = an invalid entry was seen, flags[1] = a count above 255 was seen (the caller then allocates the
// second plane and repeats the pass)
template <typename TS>
__global__ void pack_counts_kernel(const TS* __restrict__ src, uint8_t* __restrict__ lo, uint8_t* __restrict__ hi, size_t n, int* __restrict__ flags) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = (double)src[i];
  if (!(v >= 0.0) || v > 65535.0 || v != floor(v)) { atomicExch(flags, 1); lo[i] = 0; if (hi) hi[i] = 0; return; }
  const unsigned u = (unsigned)v;
  if (u > 255u) atomicExch(flags + 1, 1);
  lo[i] = (uint8_t)(u & 255u);
  if (hi) hi[i] = (uint8_t)(u >> 8);
}

// p[slot * stride + i] = v for i < n; grid = (ceil(n/1024), nslots), block = 256, 4 elements per thread
// ---- single-precision views of the factor slabs (mixed-precision dual-variational evaluation) ----------------------------
// dst[slot] (float, n x n, ld = n) <- lower triangle of src[slot] (double, same shape); the upper triangle is zeroed.
// grid = (ceil(n*n/1024), nslots), block = 256
inline __global__ void cvt_lower_f32_kernel(const double* __restrict__ src, long long sS, float* __restrict__ dst, long long sD, int n) {
  const double* a = src + (size_t)blockIdx.y * sS;
  float* b = dst + (size_t)blockIdx.y * sD;
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t e = base + 256 * u;
    if (e < (size_t)n * n) {
      const size_t i = e % n, j = e / n;
      b[e] = (i >= j) ? (float)a[e] : 0.0f;
    }
  }
}
inline __global__ void cvt_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
inline __global__ void fill_slabs_f32_kernel(float* __restrict__ p, long long stride, size_t n, float v) {
  float* dst = p + (size_t)blockIdx.y * stride;
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t i = base + 256 * u;
    if (i < n) dst[i] = v;
  }
}
// log det from a single-precision Cholesky factor, accumulated in double: out[b] = 2 sum_i log L[b][i][i]
inline __global__ void logdet_batch_f32_kernel(const float* __restrict__ L, long long sL, int ld, int n, double* __restrict__ out) {
  __shared__ double red[256];
  const float* Ls = L + (size_t)blockIdx.x * sL;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += log((double)Ls[(size_t)i * ld + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = 2.0 * red[0];
}

inline __global__ void fill_slabs_kernel(double* __restrict__ p, long long stride, size_t n, double v) {
  double* dst = p + (size_t)blockIdx.y * stride;
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t i = base + 256 * u;
    if (i < n) dst[i] = v;
  }
}
// Zero what the consumers of the low-rank covariance engine read BELOW the diagonal of a slot's L^-T slab: they take rows [roff_k, roff_k+1) of latent k and
// columns from c0_k = roff_k rounded down to `ctile` on (the Yt product; the cross / first term of the split form start at roff_k itself), the inverse writes
// the diagonal 128 x 128 blocks (zeros included) and everything above them, so only the strictly lower entries of these p rectangles can be stale.
// ~ sum r_k^2 entries per slot instead of rpad^2.  grid = (p, slots), block = 256.
inline __global__ void clear_lower_reads_kernel(double* __restrict__ Mt, long long sM, int ld, const int* __restrict__ roff, int ctile,
                                         const int* __restrict__ slots) {
  const int k = blockIdx.x;
  double* M = Mt + (size_t)(slots ? slots[blockIdx.y] : blockIdx.y) * sM;
  // (rows: the latent's count rounded up to 16 - under compact offsets the products take that many from r0 on, into the next latent's first rows)
  const int r0 = roff[k];
  const int nrow = min(((roff[k + 1] - r0 + 15) / 16) * 16, ld - r0);
  const int c0 = (r0 / ctile) * ctile;
  const int ncol = r0 + nrow - c0;
  for (int e = threadIdx.x; e < nrow * ncol; e += blockDim.x) {
    const int i = e % nrow, c = e / nrow;
    const int row = r0 + i, col = c0 + c;
    if (col < row) M[(size_t)col * ld + row] = 0.0;
  }
}
inline __global__ void fill_kernel(double* __restrict__ p, size_t n, double v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// gather / scatter of per-trial vectors between persistent [R][len] storage and slot storage [B][stride]
inline __global__ void gather_rows_kernel(const double* __restrict__ src, int len, double* __restrict__ dst, long long sDst,
                                   const int* __restrict__ trial_of_slot, int zero) {
  const int slot = blockIdx.y;
  const size_t r = trial_of_slot[slot];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) dst[(size_t)slot * sDst + i] = zero ? 0.0 : src[r * len + i];
}
// start point of a slot: how[slot] = 0 zero (cold), 1 the resident mode m, 2 the linear extrapolation m + beta (m - m_prev)
// over the trial's last two E-steps (over EM iterations the parameters drift smoothly and so do the modes).
inline __global__ void gather_start_kernel(const double* __restrict__ mode, const double* __restrict__ prev, int len, double* __restrict__ dst,
                                    long long sDst, const int* __restrict__ trial_of_slot, const int* __restrict__ how, double beta) {
  const int slot = blockIdx.y;
  const size_t r = trial_of_slot[slot];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  const int h = how[slot];
  double v = 0.0;
  if (h) {
    v = mode[r * len + i];
    if (h == 2) v += beta * (v - prev[r * len + i]);
  }
  dst[(size_t)slot * sDst + i] = v;
}
// mode <- new point; prev <- the mode it replaces where rotate[slot] != 0
inline __global__ void scatter_rotate_kernel(const double* __restrict__ src, long long sSrc, int len, double* __restrict__ mode, double* __restrict__ prev,
                                      const int* __restrict__ trial_of_slot, const int* __restrict__ rotate) {
  const int slot = blockIdx.y;
  const size_t r = trial_of_slot[slot];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  if (rotate[slot]) prev[r * len + i] = mode[r * len + i];
  mode[r * len + i] = src[(size_t)slot * sSrc + i];
}
inline __global__ void scatter_rows_kernel(const double* __restrict__ src, long long sSrc, int len, double* __restrict__ dst,
                                    const int* __restrict__ trial_of_slot) {
  const int slot = blockIdx.y;
  const size_t r = trial_of_slot[slot];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) dst[r * len + i] = src[(size_t)slot * sSrc + i];
}

}  // namespace pgpfa
