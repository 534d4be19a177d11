// Inner PCG iteration of the shared-preconditioner Newton phase without host round trips.
//
// Every slot is its own CG solve with its own forcing term; the slots advance in lockstep but LEAVE the iteration as they reach their
// targets: the list of live slots and its length live on the device (PcgCtl::nlive, rebuilt by pcg_check_kernel after every iteration),
// every kernel of the following iterations - the GEMMs included (GemmP::cols / GemmP::n_dev) - walks that list, and workgroups beyond
// its end return at once.  The solve stops when the list is empty.  (Measured at config 3: when the worst slot of 1024 reaches a
// relative residual of 1e-2 the median slot is at 1e-4 - the lockstep form ran every slot for the worst one's iteration count.)
//
// The stopping test of the inner solve (worst relative residual over the active slots <= eta) runs on the device:
// pcg_update_p2_kernel folds every slot's ratio into one word with atomicMax, pcg_check_kernel turns it into a stop flag that
// every kernel of the following iterations (the GEMMs included: GemmP::skip) reads first and returns on.  The host enqueues
// iterations ahead without synchronising and only peeks at a copy of the flag in host-mapped memory to stop enqueuing.
// Fused passes (low-rank preconditioner, p <= 16):
//   pcg_xr_apply_kernel    x += alpha p, r -= alpha q, and the first half of the preconditioner, xt = Gb r, in one pass
//   pcg_apply2_dots_kernel z = Gb (eps r + F Sb F^T Gb r) with the partial sums of r.z and r.r of the tile
//   pcg_update_p2_kernel   beta from the partial sums, p = z + beta p (one pass over z and p), residual ratio
// and the curvature blocks of the slots are read as packed single-precision lower triangles in the Hessian-vector product
// (pack_w32_kernel / pcg_hessvec32_dot_kernel): the inner solves then use H~ = K^-1 + fl32(W), an inexact Newton matrix whose
// relative error (6e-8) is far below the forcing terms; gradients, objective and the covariance phase keep the FP64 blocks.
#pragma once

namespace pgpfa {

struct PcgCtl { int stop; int iters; unsigned worst_bits; int nlive; unsigned long long slot_iters; };

// One block: close an iteration.  ctl->iters counts executed iterations, ctl->slot_iters the slot-iterations (sum of the live counts).
// The live list is compacted in place: a slot stays while its residual ratio (ratio[slot], written by pcg_update_p2_kernel) is above
// its own target eta[slot] - or, before inner_min iterations, always - and the order of the survivors is kept.  stop is raised when
// nobody is left.  host (mapped, may be null) receives {stop, iters}.  block = 256 threads.
__global__ __launch_bounds__(256) void pcg_check_kernel(PcgCtl* __restrict__ ctl, volatile int* __restrict__ host, int* __restrict__ live,
                                                        const float* __restrict__ ratio, const float* __restrict__ eta, int inner_min) {
  __shared__ int keep_s[256];
  __shared__ int base_s, n_s, it_s;
  if (threadIdx.x == 0) {
    n_s = ctl->stop ? 0 : ctl->nlive;
    it_s = ctl->iters + (ctl->stop ? 0 : 1);
    base_s = 0;
  }
  __syncthreads();
  const int n = n_s, it = it_s;
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int i = c0 + threadIdx.x;
    int slot = -1, keep = 0;
    if (i < n) {
      slot = live[i];
      const float r = ratio[slot];
      keep = (it < inner_min || !(r <= eta[slot])) ? 1 : 0;       // (a NaN ratio keeps iterating: the outer loop deals with it)
    }
    keep_s[threadIdx.x] = keep;
    __syncthreads();
    // exclusive prefix sum of the keep flags of this chunk (256 entries: a serial scan by one thread is ~1 us)
    if (threadIdx.x == 0) {
      int run = base_s;
      for (int j = 0; j < 256; ++j) { const int k = keep_s[j]; keep_s[j] = run; run += k; }
      base_s = run;
    }
    __syncthreads();
    const int pos = keep_s[threadIdx.x];
    __syncthreads();                                               // everyone has read its slot and position before anyone writes
    if (keep) live[pos] = slot;                                    // pos <= i: never overwrites an entry of a later chunk
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (!ctl->stop) {
      ctl->iters = it;
      ctl->slot_iters += (unsigned long long)n;
      ctl->nlive = base_s;
      if (base_s == 0) ctl->stop = 1;
    }
    ctl->worst_bits = 0u;
    if (host) {
      host[1] = ctl->iters;
      __threadfence_system();
      host[0] = ctl->stop;
      __threadfence_system();
    }
  }
}

// W[slot][t][p][p] (double) -> Wp[slot][t][NP] (float, lower triangle a >= b at a(a+1)/2 + b) for the listed slots.
// grid = (ceil(T*NP/256), nslots)
__global__ void pack_w32_kernel(const double* __restrict__ W, long long sW, float* __restrict__ Wp, long long sWp, int T, int p,
                                const int* __restrict__ slots) {
  const size_t slot = slots[blockIdx.y];
  const int np = p * (p + 1) / 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= T * np) return;
  const int t = e / np, c = e - t * np;
  int a = 0;
  while ((a + 1) * (a + 2) / 2 <= c) ++a;
  const int b = c - a * (a + 1) / 2;
  Wp[slot * sWp + e] = (float)W[slot * sW + (size_t)t * p * p + a * p + b];
}

// q += W p (W from the packed single-precision triangles), partial p.q per 64-bin tile; same launch shape and outputs as
// pcg_hessvec_dot_kernel.  grid = (ceil(T/64), nslots), block = 256.
template <int PW>
__global__ __launch_bounds__(256) void pcg_hessvec32_dot_kernel(const float* __restrict__ Wp, long long sWp, const double* __restrict__ P,
                                                                double* __restrict__ Q, long long sV, int T, int p,
                                                                const int* __restrict__ slots, double* __restrict__ pqpart,
                                                                const int* __restrict__ skip, const PcgCtl* __restrict__ ctl) {
  if (skip && *skip) return;
  if (ctl && (int)blockIdx.y >= ctl->nlive) return;              // (slots: the live list; workgroups past its end have nothing to do)
  constexpr int NPW = PW * (PW + 1) / 2, LD = NPW | 1;          // odd row stride: lanes (bins) hit distinct banks
  __shared__ float Ws[64 * LD];
  __shared__ double red[4];
  const int np = p * (p + 1) / 2;
  const size_t slot = slots[blockIdx.y];
  const int t0 = blockIdx.x * 64;
  const int nt = min(64, T - t0);
  const float* wbase = Wp + slot * sWp + (size_t)t0 * np;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  for (int e = threadIdx.x; e < nt * np; e += 256) {
    const int t = e / np, idx = e - t * np;
    Ws[t * LD + idx] = wbase[e];
  }
  __syncthreads();
  double acc = 0.0;
  if (live) {
    const float* wt = Ws + lane * LD;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int k = wave + 4 * i;
      if (k < p) {
        double s2 = qk[i];
        double vk = 0.0;
#pragma unroll
        for (int l = 0; l < PW; ++l) {
          if (l < p) {
            const int hi = k > l ? k : l, lo = k > l ? l : k;
            s2 += (double)wt[hi * (hi + 1) / 2 + lo] * v[l];
          }
          vk = (l == k) ? v[l] : vk;
        }
        q[(size_t)k * T] = s2;
        acc += s2 * vk;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) pqpart[slot * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int PCG_SLOTS = 16;       // slots of the active list walked by one block of the per-bin kernels

// alpha = rz / sum(pqpart) ; x += alpha p ; r -= alpha q ; xt = Gb r   for the active slots list[0..na).
// grid = (ceil(T/64), ceil(na / PCG_SLOTS)), block = 256 (lanes = bins, waves = slots), p <= PW.
template <int PW>
__global__ __launch_bounds__(256) void pcg_xr_apply_kernel(const double* __restrict__ Gb, double* __restrict__ X, double* __restrict__ R,
                                                           const double* __restrict__ P, const double* __restrict__ Q,
                                                           double* __restrict__ Xt, long long sV, int T, int p, const int* __restrict__ list,
                                                           int na, const double* __restrict__ rz, const double* __restrict__ pqpart,
                                                           int ntile, const int* __restrict__ skip, const PcgCtl* __restrict__ ctl) {
  if (skip && *skip) return;
  if (ctl) na = ctl->nlive;                                        // (list: the live list)
  if ((int)blockIdx.y * PCG_SLOTS >= na) return;
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
  const int s_end = min(na, (int)(blockIdx.y + 1) * PCG_SLOTS);
  // U slots per trip: the loads of all of them are issued before the first use (the wave is latency-bound otherwise), and the
  // block's LDS copy of Gb is read once for the U matrix-vector products.  A trip past the end of the list re-reads its
  // first slot and stores nothing.
  constexpr int U = PW <= 10 ? 2 : 1;
  for (int si0 = blockIdx.y * PCG_SLOTS + wave; si0 < s_end; si0 += 4 * U) {
    asm volatile("" ::: "memory");                        // keep the LDS reads of Gb inside the trip (hoisted, they cost 2 PW^2 registers)
    size_t base[U];
    double alpha[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int si = si0 + 4 * u;
      ok[u] = si < s_end;
      const size_t sl = list[ok[u] ? si : si0];
      base[u] = sl * sV + t;
      double d = 0.0;
      for (int i = 0; i < ntile; ++i) d += pqpart[sl * ntile + i];
      alpha[u] = (d > 0.0) ? rz[sl] / d : 0.0;
    }
    double xv[U][PW], pv[U][PW], rv[U][PW], qv[U][PW];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const size_t o = base[u] + (size_t)(k < p ? k : 0) * T;
        xv[u][k] = X[o]; pv[u][k] = P[o]; rv[u][k] = R[o]; qv[u][k] = Q[o];
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const double rn = rv[u][k] - alpha[u] * qv[u][k];
        rv[u][k] = (k < p) ? rn : 0.0;
        if (k < p && ok[u]) {
          const size_t o = base[u] + (size_t)k * T;
          X[o] = xv[u][k] + alpha[u] * pv[u][k];
          R[o] = rn;
        }
      }
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      double acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = 0.0;
#pragma unroll
      for (int kk = 0; kk < PW; ++kk) {
        const double gk = g[k * PW + kk];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += gk * rv[u][kk];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k < p && ok[u]) Xt[base[u] + (size_t)k * T] = acc[u];
    }
  }
}

// z = Gb (eps r + y) with y = F Sb F^T Gb r already in Y2 ; partial sums of r.z and r.r per (slot, tile):
// part[(slot * ntile + tile) * 2 + {0, 1}].  Same launch shape as pcg_xr_apply_kernel.
template <int PW>
__global__ __launch_bounds__(256) void pcg_apply2_dots_kernel(const double* __restrict__ Gb, const double* __restrict__ R,
                                                              const double* __restrict__ Y2, double eps, double* __restrict__ Z,
                                                              long long sV, int T, int p, const int* __restrict__ list, int na,
                                                              double* __restrict__ part, const int* __restrict__ skip,
                                                              const PcgCtl* __restrict__ ctl) {
  if (skip && *skip) return;
  if (ctl) na = ctl->nlive;
  if ((int)blockIdx.y * PCG_SLOTS >= na) return;
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
  const bool live = lane < nt;
  const int t = t0 + (live ? lane : 0);
  const double* g = Gs + (live ? lane : 0) * LD;
  const int s_end = min(na, (int)(blockIdx.y + 1) * PCG_SLOTS);
  constexpr int U = PW <= 10 ? 2 : 1;                        // (see pcg_xr_apply_kernel)
  for (int si0 = blockIdx.y * PCG_SLOTS + wave; si0 < s_end; si0 += 4 * U) {
    asm volatile("" ::: "memory");                        // keep the LDS reads of Gb inside the trip (hoisted, they cost 2 PW^2 registers)
    size_t sl[U], base[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int si = si0 + 4 * u;
      ok[u] = si < s_end;
      sl[u] = list[ok[u] ? si : si0];
      base[u] = sl[u] * sV + t;
    }
    double v[U][PW], rv[U][PW];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const size_t o = base[u] + (size_t)(k < p ? k : 0) * T;
        rv[u][k] = R[o]; v[u][k] = Y2[o];
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const bool in = k < p && live;
        rv[u][k] = in ? rv[u][k] : 0.0;
        v[u][k] = in ? eps * rv[u][k] + v[u][k] : 0.0;
      }
    double a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { a[u] = 0.0; b[u] = 0.0; }
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      double acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = 0.0;
#pragma unroll
      for (int kk = 0; kk < PW; ++kk) {
        const double gk = g[k * PW + kk];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += gk * v[u][kk];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (k < p && live && ok[u]) Z[base[u] + (size_t)k * T] = acc[u];
        a[u] += rv[u][k] * acc[u];
        b[u] += rv[u][k] * rv[u][k];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double au = a[u], bu = b[u];
      for (int off = 32; off > 0; off >>= 1) { au += __shfl_down(au, off); bu += __shfl_down(bu, off); }
      if (lane == 0 && ok[u]) {
        part[(sl[u] * gridDim.x + blockIdx.x) * 2] = au;
        part[(sl[u] * gridDim.x + blockIdx.x) * 2 + 1] = bu;
      }
    }
  }
}

// rz_new, rr from the tile partial sums (tile order: deterministic) ; beta = rz_new / rz ; p = z + beta p ; rz = rz_new ;
// rr0 on the first call of a solve ; the slot's residual ratio sqrt(rr / rr0) goes to ratio_out[slot] (and into ctl->worst_bits).
// grid = (na), block = 256; with ctl the list is the live list and workgroups past ctl->nlive return.
__global__ __launch_bounds__(256) void pcg_update_p2_kernel(const double* __restrict__ Z, double* __restrict__ P, long long sV, int n,
                                                            const int* __restrict__ list, const double* __restrict__ part, int ntile,
                                                            double* __restrict__ rz, double* __restrict__ rr, double* __restrict__ rr0,
                                                            int first, PcgCtl* __restrict__ ctl, float* __restrict__ ratio_out) {
  if (ctl && (ctl->stop || (int)blockIdx.x >= ctl->nlive)) return;
  __shared__ double beta_s;
  const size_t slot = list[blockIdx.x];
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < ntile; ++i) { a += part[(slot * ntile + i) * 2]; b += part[(slot * ntile + i) * 2 + 1]; }
    const double old = rz[slot];
    beta_s = (first || !(old > 0.0)) ? 0.0 : a / old;
    rz[slot] = a;
    rr[slot] = b;
    double b0 = rr0[slot];
    if (first) { rr0[slot] = b; b0 = b; }
    if (ctl) {
      const float ratio = (b0 > 0.0) ? (float)sqrt(b / b0) : 0.0f;
      if (ratio_out) ratio_out[slot] = ratio;
      // (finite non-negative floats order like their bit patterns; a NaN maps above every finite value and keeps iterating)
      atomicMax(&ctl->worst_bits, __float_as_uint(ratio));
    }
  }
  __syncthreads();
  const double beta = beta_s;
  for (int i = threadIdx.x; i < n; i += 256) P[slot * sV + i] = Z[slot * sV + i] + beta * P[slot * sV + i];
}

}  // namespace pgpfa
