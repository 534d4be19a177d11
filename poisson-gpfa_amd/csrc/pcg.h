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
#include "types.h"

namespace pgpfa {


// One block: close an iteration.  ctl->iters counts executed iterations, ctl->slot_iters the slot-iterations (sum of the live counts).
// The live list is compacted in place: a slot stays while its residual ratio (ratio[slot], written by pcg_update_p2_kernel) is above
// its own target eta[slot] - or, before inner_min iterations, always - and the order of the survivors is kept.  stop is raised when
// nobody is left.  host (mapped, may be null) receives {stop, iters}.  block = 256 threads.
inline __global__ __launch_bounds__(256) void pcg_check_kernel(PcgCtl* __restrict__ ctl, volatile int* __restrict__ host, int* __restrict__ live,
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
inline __global__ void pack_w32_kernel(const double* __restrict__ W, long long sW, float* __restrict__ Wp, long long sWp, int T, int p,
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
inline __global__ __launch_bounds__(256) void pcg_update_p2_kernel(const double* __restrict__ Z, double* __restrict__ P, long long sV, int n,
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

// ---------------------------------------------------------------------------------------------------------------------------------
// The PCG step without the prior mat-vec, as TWO tile-parallel kernels with ONE reduction (round 4).
//
// The preconditioner is the exact inverse of  P = Kt^-1 + Wb  with Kt = eps I + F F^T the low-rank form of the prior and Wb the mean
// curvature blocks, so z = P^-1 r satisfies  Kt^-1 z = r - Wb z : a per-bin product instead of p products with T x T matrices.  In the
// Chronopoulos-Gear form of PCG the matrix is applied to z, not to the search direction:
//   A (pcg_cg_a_kernel):  z = Gb (eps r + y),  s = H~ z = (r - Wb z) + fl32(W) z,  partial sums of gamma = r.z, delta = z.s, r.r per
//       (slot, tile of 64 bins)                                               [y = F Sb F^T Gb r comes from the three thin products]
//   B (pcg_cg_b_kernel):  gamma, delta, r.r from the partial sums (every tile of a slot adds the same numbers in the same order), the
//       slot's own stopping test, beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old),  p = z + beta p,
//       q = s + beta q (= H~ p),  x += alpha p,  r -= alpha q,  t = Gb r  (input of the next products)
// so a step is A, B, a one-thread closing kernel and the products: 4 + 11 + 2 passes over an n-vector per live slot, against 20 and a
// 5-GFLOP product with K^-1 in the round-3 form.  H~ = Kt^-1 + fl32(W) differs from H by the truncation of the pivoted Cholesky seen
// through K^-1 (|Kt^-1 - K^-1| <= |E| / eps^2 where K^-1 ~ 1/eps, i.e. ~1e3 |E| relative to H there) and by single-precision curvature
// (Wb is read in single precision too): an inexact Newton matrix.  Gradient, objective and line search keep the exact K^-1 (one product
// per OUTER iteration), and the outer loop measures the contraction of the TRUE gradient over every accepted step (estep_impl) instead of
// trusting the inner residual.
// A workgroup is (64 bins) x (16 slots): the packed triangles of Gb / Wb of its bins sit in LDS once for the 16 slots; the slots' own
// curvature comes component-major ([c][T] floats: a wave reads 64 consecutive bins of one component) straight from memory.  All loads are
// unconditional from clamped addresses and masked afterwards - predicated loads get an s_waitcnt vmcnt(0) each (measured: 106 -> 82 us).
// A slot that has reached its forcing term does not append itself to the next live list (atomic counter: the order of the list is not
// deterministic, the arithmetic of a slot does not depend on it).  Scalars carried between steps (gamma, alpha) are double-buffered by
// step parity: the tiles of one slot run in different workgroups, and the one that writes must not be seen by the others of the same step.
//
// Tried first and dropped: the whole step as ONE kernel with workgroup = slot, thread = bin (both reductions inside the workgroup, 12
// vector passes): 170 us per step at 1024 live slots against 82 + 103 us here, but 40 us even for a single live slot (one workgroup's
// chain of dependent loads and two barriers), 512 bins at most, and with the loads hoisted it needs more than 256 registers.
struct PcgCgP {
  const double* GbT; const float* WbT;        // [NP][T] packed lower triangles (component-major; Wb in single precision)
  const float* W32T; long long sW32; int Tw;  // [slot][NP][Tw], Tw = T rounded up to 32 floats (rows start on 128-byte lines)
  double *X, *R, *P, *Q, *Z, *S, *Y; long long sV;
  double* part;                                // [slot][ntile][3]
  double *gam, *alp;                           // [2][B]: gamma, alpha of the previous step at [par], of this step at [par ^ 1]
  double *rr, *rr0; const float* eta;
  PcgCtl* ctl; int* live0; int* live1;
  double eps; int T, p, par, first, inner_min, ntile, B;
  int xcd_map;                                 // 1: workgroup -> (bin tile, slot group) by pcg_cg_wg; 0: grid indices as they are
  int spw;                                     // slots per workgroup (a multiple of 4, at most PCG_SLOTS): the host picks it by the live count it last saw
  // Row strides of a latent inside a slot vector: Tl for the vectors PRIVATE to the solve (R, P, Q, Z, S, Y) - round 5: T rounded up to 16 doubles,
  // so that every latent's row and every 64-bin tile of it start on a 128-byte line (at T = 500 a row started 32 bytes into a line and a tile's
  // 512 bytes touched 5 lines for 4) - and Tx for X, the step the rest of the E-step reads in its compact layout (k T + t).  Form 1: Tl = Tx = T.
  int Tl, Tx;
  int step;                                    // index of this step within the solve
  // (round 5, second half) vec32: z, s, p, q and the preconditioner's t / y are STORED in single precision (same element offsets in the same
  // buffers); x and r - the vectors that accumulate - and every product and dot stay FP64.  After preconditioning |s| = |H~ z| ~ |r| and
  // |q| ~ |r|, so a rounding of 6e-8 relative in those vectors perturbs the residual recurrence by ~1e-7 |r| per step - the forcing terms of the
  // solves never go below 7e-6 (e = sqrt(xtol / 20) minimises max(e, xtol / 20 e)), and the outer loop measures the true gradient anyway.  A
  // slot-step moves 13.8 n-vector equivalents instead of 19.8.
  int vec32;
  int fold_close;                              // 1: kernel A of step i closes step i - 1 (no closing launch per step; pcg_iter_close_kernel once after the last)
  volatile int* host;                          // host-mapped {stop, steps, live} mirror (written by whoever closes a step)
  const double *Gl, *KX; double* Gt;           // pcg_cg_start_kernel: likelihood gradient and K^-1 x (compact), their sum out
  // (round 6) rx32: the residual r and the step x of the solve are STORED in single precision as well (kernels instantiated with TX = float; up to 10
  // latents).  r then lives in R's buffer at the same element offsets, x in X32 on the private row stride Tl (the caller widens it into its FP64
  // step vector once, after the solve).  Every product, dot and update is still computed in FP64 from the widened values.  x = sum alpha_i p_i is
  // a sum of vectors that are already rounded to single precision when they are stored (p), and the forcing terms never go below 7e-6: the rounding
  // of r and x (6e-8 relative per step) stays two orders of magnitude under what the solve is asked for, and the outer loop measures the true
  // gradient of every accepted step.  A slot-step moves 68 bytes per entry of an n-vector instead of 88 (+ 22 of packed curvature).
  float* X32;
};

// Closes step `par`'s bookkeeping (one thread): counts, stop flag when the next list is empty, the list just consumed is reset for the step after
// the next, host mirror {stop, steps, live}.
__device__ __forceinline__ void pcg_close_step(PcgCtl* __restrict__ ctl, int par, volatile int* __restrict__ host) {
  if (!ctl->stop) {
    const int nnext = ctl->nl[par ^ 1];
    ctl->iters += 1;
    ctl->slot_iters += (unsigned long long)nnext;          // CG steps taken: the slots that went on (a retiring slot only ran the test)
    ctl->nlive = nnext;
    if (nnext == 0) ctl->stop = 1;
  }
  ctl->nl[par] = 0;
  ctl->pad_[0] += 1;                                         // steps closed
  if (host) {
    host[2] = ctl->stop ? 0 : ctl->nlive;                  // (the live count only falls during a solve: a stale value is an upper bound)
    host[1] = ctl->iters;
    __threadfence_system();
    host[0] = ctl->stop;
    __threadfence_system();
  }
}

// out = M v for the symmetric p x p matrix of this thread's bin, rows in two groups of about half the packed entries: the loads of a group
// are in flight together, the compiler barrier keeps the second group's behind the first group's arithmetic (left alone the compiler
// hoists every load of every product of a slot, runs out of registers and spills).  M: packed lower triangle, component stride `cs`
// (global component-major arrays: cs = T; an LDS tile row: cs = 1); v is zero beyond p; loads are unconditional - a component index past
// the matrix (clamp = true: global arrays sized for p) is clamped to 0 and meets a zero of v.
template <int PW, bool CLAMP, typename TM>
__device__ __forceinline__ void pcg_sym_mv(const TM* __restrict__ M, size_t cs, int p, const double (&v)[PW], double (&out)[PW]) {
#pragma unroll
  for (int k = 0; k < PW; ++k) out[k] = 0.0;
  constexpr int HSPLIT = (PW * 7 + 9) / 10;
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
    constexpr int NPK = PW * (PW + 1) / 2;
    TM g[NPK];
#pragma unroll
    for (int hi = 0; hi < PW; ++hi)
#pragma unroll
      for (int lo = 0; lo <= hi; ++lo) {
        if ((grp == 0) != (hi < HSPLIT)) continue;
        const int c = hi * (hi + 1) / 2 + lo;
        g[c] = M[(size_t)((!CLAMP || hi < p) ? c : 0) * cs];
      }
#pragma unroll
    for (int hi = 0; hi < PW; ++hi)
#pragma unroll
      for (int lo = 0; lo <= hi; ++lo) {
        if ((grp == 0) != (hi < HSPLIT)) continue;
        const double gg = (double)g[hi * (hi + 1) / 2 + lo];
        out[hi] += gg * v[lo];
        if (lo != hi) out[lo] += gg * v[hi];
      }
    asm volatile("" ::: "memory");
  }
  if (CLAMP) {
#pragma unroll
    for (int k = 0; k < PW; ++k) out[k] = (k < p) ? out[k] : 0.0;
  }
}

// out = M v with the packed lower triangle of M already in registers (entries of components past p are whatever the clamped loads brought:
// they meet zeros of v, and out is zeroed past p)
template <int PW>
__device__ __forceinline__ void pcg_sym_mv_reg(const float (&m)[PW * (PW + 1) / 2], int p, const double (&v)[PW], double (&out)[PW]) {
#pragma unroll
  for (int k = 0; k < PW; ++k) out[k] = 0.0;
#pragma unroll
  for (int hi = 0; hi < PW; ++hi)
#pragma unroll
    for (int lo = 0; lo <= hi; ++lo) {
      const double gg = (double)m[hi * (hi + 1) / 2 + lo];
      out[hi] += gg * v[lo];
      if (lo != hi) out[lo] += gg * v[hi];
    }
#pragma unroll
  for (int k = 0; k < PW; ++k) out[k] = (k < p) ? out[k] : 0.0;
}

// LDS tile [64 bins][LD] of a component-major packed matrix; components past np and bins past nt are zero
template <int NPW, typename TM, typename TL>
__device__ __forceinline__ void pcg_stage_sym(const TM* __restrict__ MT, int T, int t0, int nt, int np, TL* __restrict__ dst, int LD) {
  for (int e = threadIdx.x; e < NPW * 64; e += 256) {
    const int c = e >> 6, t = e & 63;
    dst[t * LD + c] = (t < nt && c < np) ? (TL)MT[(size_t)c * T + t0 + t] : (TL)0;
  }
}

constexpr int pcg_cg_ld(int pw) { return (pw * (pw + 1) / 2) | 1; }
inline size_t pcg_cg_a_lds(int pw) { return (size_t)64 * pcg_cg_ld(pw) * (sizeof(double) + sizeof(float)); }
inline size_t pcg_cg_b_lds(int pw) { return (size_t)64 * pcg_cg_ld(pw) * sizeof(double); }

// grid = (ceil(T/64), ceil(live bound / spw) rounded up to 8: pcg_cg_wg), block = 256 (lanes = bins, waves = slots); dynamic LDS = pcg_cg_a_lds(PW)
// (two waves per SIMD: left alone the compiler hoists every load and LDS read of a slot, takes all 256 registers and one workgroup fills a CU)
// (bin tile, slot group) of a workgroup.  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with an L2 of its own: with the bin
// tile as the fast grid index the eight tiles of a slot group sat on eight XCDs, and since a latent's row of an n-vector is not line-aligned (T = 500: a 64-bin
// run touches 5 lines for 4) every boundary line was fetched into two L2s.  Here eight consecutive ids take eight different slot groups and an XCD walks the
// bin tiles of its groups in turn: a slot's tiles, its boundary lines and its scalars meet in ONE L2.  Live groups still fill a prefix of the ids (in blocks of
// eight groups).  grid = (ntile, slot groups rounded up to a multiple of 8).
__device__ __forceinline__ void pcg_cg_wg(int ntile, int xcd_map, int& tile, int& group) {
  if (!xcd_map) { tile = blockIdx.x; group = blockIdx.y; return; }
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int xcd = id & 7, m = id >> 3;
  tile = m % ntile;
  group = xcd + 8 * (m / ntile);
}

template <int PW, typename TV = double, typename TX = double>
__global__ __launch_bounds__(256, 2) void pcg_cg_a_kernel(PcgCgP a) {
  const TX* const Rp = reinterpret_cast<const TX*>(a.R);        // TX: storage type of r (and of x in kernel B): PcgCgP::X32
  // TV: storage type of the vectors private to the solve that carry no accumulated state - z, s, p, q and the preconditioner's t / y (see PcgCgP::vec32)
  TV* const Zp = reinterpret_cast<TV*>(a.Z); TV* const Sp = reinterpret_cast<TV*>(a.S); TV* const Pp = reinterpret_cast<TV*>(a.P);
  TV* const Qp = reinterpret_cast<TV*>(a.Q); TV* const Yp = reinterpret_cast<TV*>(a.Y);
  (void)Zp; (void)Sp; (void)Pp; (void)Qp; (void)Yp;
  constexpr int NP = PW * (PW + 1) / 2, LD = pcg_cg_ld(PW);
  extern __shared__ double pcg_cg_smem[];
  double* Gs = pcg_cg_smem;
  float* Ws = reinterpret_cast<float*>(pcg_cg_smem + 64 * LD);
  PcgCtl* ctl = a.ctl;
  if (ctl->stop) return;
  const int na = ctl->nl[a.par];
  // (fold_close: the first workgroup closes the step before this one - its list is nl[par ^ 1], which nobody in this launch reads; the other
  //  workgroups do not wait for it: they take their count from nl[par], final since kernel B of that step ended.  A step whose list came out
  //  empty makes every workgroup return on na = 0; `stop` is for the products' skip flag and the host.)
  if (a.fold_close && a.step > 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && ctl->pad_[0] == a.step - 1) pcg_close_step(ctl, a.par ^ 1, a.host);
  int wg_tile, wg_group;
  pcg_cg_wg(a.ntile, a.xcd_map, wg_tile, wg_group);
  if (wg_group * a.spw >= na) return;
  const int T = a.T, p = a.p, np = p * (p + 1) / 2;
  const int Tl = a.Tl;
  const int t0 = wg_tile * 64;
  const int nt = min(64, T - t0);
  pcg_stage_sym<NP>(a.GbT, T, t0, nt, np, Gs, LD);
  pcg_stage_sym<NP>(a.WbT, T, t0, nt, np, Ws, LD);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool in = lane < nt;
  const int t = t0 + (in ? lane : 0);
  const double* g = Gs + lane * LD;
  const float* wb = Ws + lane * LD;
  const int* live = a.par ? a.live1 : a.live0;
  const int s_end = min(na, (wg_group + 1) * a.spw);
  for (int si = wg_group * a.spw + wave; si < s_end; si += 4) {
    const size_t slot = (size_t)live[si];
    const size_t base = slot * a.sV + t;
    double r[PW], v[PW], z[PW];
    float w32[NP];
    // every global load of the slot in ONE round - r, y and the packed single-precision curvature, 2 PW + NP loads in flight - before any
    // arithmetic: with the curvature read inside its product (two row groups behind the two LDS products) a slot was three memory round
    // trips in a row, and a wave walks four slots
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const size_t o = base + (size_t)(k < p ? k : 0) * Tl;
      r[k] = (double)Rp[o];
      v[k] = (double)Yp[o];
    }
    {
      const float* wp = a.W32T + slot * a.sW32 + t;
#pragma unroll
      for (int hi = 0; hi < PW; ++hi)
#pragma unroll
        for (int lo = 0; lo <= hi; ++lo) {
          const int c = hi * (hi + 1) / 2 + lo;
          w32[c] = wp[(size_t)(hi < p ? c : 0) * a.Tw];
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const bool ok = in && k < p;
      r[k] = ok ? r[k] : 0.0;
      v[k] = ok ? a.eps * r[k] + v[k] : 0.0;
    }
    pcg_sym_mv<PW, false>(g, 1, p, v, z);
    // (round 6) s = (r - Wb z) + fl32(W) z = r + (fl32(W) - fl32(Wb)) z: ONE product with the difference of the two single-precision triangles
    // (formed in single precision: its rounding is 6e-8 of the DIFFERENCE) where there were two - a third of the kernel's multiply-adds
#pragma unroll
    for (int c = 0; c < NP; ++c) w32[c] -= wb[c];
    pcg_sym_mv_reg<PW>(w32, p, z, v);                                                // v <- (fl32(W) - fl32(Wb)) z
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const double sk = r[k] + v[k];                                 // s = H~ z
      s0 += r[k] * z[k]; s1 += z[k] * sk; s2 += r[k] * r[k];
      if (in && k < p) {
        const size_t o = base + (size_t)k * Tl;
        Zp[o] = (TV)z[k];
        Sp[o] = (TV)sk;
      }
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if (lane == 0) {
      double* pp = a.part + (slot * a.ntile + wg_tile) * 3;
      pp[0] = s0; pp[1] = s1; pp[2] = s2;
    }
  }
}

// same launch shape; dynamic LDS = pcg_cg_b_lds(PW)
template <int PW, typename TV = double, typename TX = double>
__global__ __launch_bounds__(256, 3) void pcg_cg_b_kernel(PcgCgP a) {
  constexpr bool X32 = sizeof(TX) == 4;
  TX* const Rp = reinterpret_cast<TX*>(a.R);
  // TV: storage type of the vectors private to the solve that carry no accumulated state - z, s, p, q and the preconditioner's t / y (see PcgCgP::vec32)
  TV* const Zp = reinterpret_cast<TV*>(a.Z); TV* const Sp = reinterpret_cast<TV*>(a.S); TV* const Pp = reinterpret_cast<TV*>(a.P);
  TV* const Qp = reinterpret_cast<TV*>(a.Q); TV* const Yp = reinterpret_cast<TV*>(a.Y);
  (void)Zp; (void)Sp; (void)Pp; (void)Qp; (void)Yp;
  constexpr int NP = PW * (PW + 1) / 2, LD = pcg_cg_ld(PW);
  extern __shared__ double pcg_cg_smem[];
  double* Gs = pcg_cg_smem;
  PcgCtl* ctl = a.ctl;
  if (ctl->stop) return;
  const int na = ctl->nl[a.par];
  int wg_tile, wg_group;
  pcg_cg_wg(a.ntile, a.xcd_map, wg_tile, wg_group);
  if (wg_group * a.spw >= na) return;
  const int it = a.fold_close ? a.step : ctl->iters;       // (steps executed before this one: the closing of step - 1 may still be in flight in kernel A's first workgroup... it is not - A has ended - but a.step is the same number without the read)
  const int T = a.T, p = a.p, np = p * (p + 1) / 2;
  const int Tl = a.Tl, Tx = a.Tx;
  const int t0 = wg_tile * 64;
  const int nt = min(64, T - t0);
  pcg_stage_sym<NP>(a.GbT, T, t0, nt, np, Gs, LD);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool in = lane < nt;
  const int t = t0 + (in ? lane : 0);
  const double* g = Gs + lane * LD;
  const int* live = a.par ? a.live1 : a.live0;
  int* live_next = a.par ? a.live0 : a.live1;
  const double* gam_old = a.gam + (size_t)a.par * a.B;
  const double* alp_old = a.alp + (size_t)a.par * a.B;
  double* gam_new = a.gam + (size_t)(a.par ^ 1) * a.B;
  double* alp_new = a.alp + (size_t)(a.par ^ 1) * a.B;
  const int s_end = min(na, (wg_group + 1) * a.spw);
  for (int si = wg_group * a.spw + wave; si < s_end; si += 4) {
    const int sloti = live[si];
    const size_t slot = (size_t)sloti;
    double gamma = 0.0, delta = 0.0, rrn = 0.0;
    for (int i = 0; i < a.ntile; ++i) {
      const double* pp = a.part + (slot * a.ntile + i) * 3;
      gamma += pp[0]; delta += pp[1]; rrn += pp[2];
    }
    const double b0 = a.first ? rrn : a.rr0[slot];
    const float ratio = (b0 > 0.0) ? (float)sqrt(rrn / b0) : 0.0f;
    const bool keep = (it < a.inner_min || !(ratio <= a.eta[slot]));      // (a NaN ratio keeps iterating: the outer loop deals with it)
    const double g_old = gam_old[slot], a_old = alp_old[slot];
    const double beta = (a.first || !(g_old > 0.0)) ? 0.0 : gamma / g_old;
    const double den = (a.first || !(a_old > 0.0)) ? delta : delta - beta * gamma / a_old;
    const double alpha = (den > 0.0) ? gamma / den : 0.0;
    if (wg_tile == 0 && lane == 0) {
      if (a.first) a.rr0[slot] = rrn;
      a.rr[slot] = rrn;
      if (keep) {
        gam_new[slot] = gamma;
        alp_new[slot] = alpha;
        const int pos = atomicAdd(&ctl->nl[a.par ^ 1], 1);
        live_next[pos] = sloti;
      }
    }
    if (!keep) continue;
    const size_t base = slot * a.sV + t;
    double r[PW], pv[PW], tv[PW];
    double zk[PW], sk[PW], po[PW], qo[PW], xo[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const size_t o = base + (size_t)(k < p ? k : 0) * Tl;
      zk[k] = (double)Zp[o]; sk[k] = (double)Sp[o]; r[k] = (double)Rp[o];
      if constexpr (X32) xo[k] = (double)a.X32[o]; else xo[k] = a.X[base + (size_t)(k < p ? k : 0) * Tx];
      po[k] = 0.0; qo[k] = 0.0;
    }
    if (!a.first) {
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        const size_t o = base + (size_t)(k < p ? k : 0) * Tl;
        po[k] = (double)Pp[o]; qo[k] = (double)Qp[o];
      }
    }
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const bool ok = in && k < p;
      const size_t o = base + (size_t)(k < p ? k : 0) * Tl;
      pv[k] = zk[k] + beta * po[k];
      const double qn = sk[k] + beta * qo[k];
      r[k] = ok ? (double)(TX)(r[k] - alpha * qn) : 0.0;            // (the stored value: t = Gb r below and kernel A's r are then the same vector)
      if (ok) {
        Pp[o] = (TV)pv[k];
        Qp[o] = (TV)qn;
        if constexpr (X32) a.X32[o] = (float)(xo[k] + alpha * pv[k]); else a.X[base + (size_t)k * Tx] = xo[k] + alpha * pv[k];
        Rp[o] = (TX)r[k];
      }
    }
    pcg_sym_mv<PW, false>(g, 1, p, r, tv);
#pragma unroll
    for (int k = 0; k < PW; ++k)
      if (in && k < p) Yp[base + (size_t)k * Tl] = (TV)tv[k];
  }
}

// Closes a step (one thread): counts, stop flag when the next list is empty, the list just consumed is reset for the step after the
// next, host mirror {stop, steps}.  (A separate launch, not a "last workgroup" inside B: a device-scope release there makes every
// workgroup write its XCD's L2 back.)
// (step >= 0: only when that step is the next one to close - with fold_close kernel A of the following step may already have done it)
inline __global__ void pcg_iter_close_kernel(PcgCtl* __restrict__ ctl, int par, volatile int* __restrict__ host, int step = -1) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (step >= 0 && ctl->pad_[0] != step) return;
  pcg_close_step(ctl, par, host);
}

// First kernel of a solve (round 5; one launch where grad_total_kernel, pcg_init_kernel and the per-bin application of the shared preconditioner
// were three): for the slots of the solve's first live list  g = Gl + K^-1 x -> Gt (compact: the outer loop reads it),  r = -g -> R,  x = 0 -> X,
// t = Gb r -> Y (input of the products y = F Sb F^T t).  Same launch shape and LDS as kernel B.
template <int PW, typename TV = double, typename TX = double>
__global__ __launch_bounds__(256, 3) void pcg_cg_start_kernel(PcgCgP a) {
  constexpr bool X32 = sizeof(TX) == 4;
  TX* const Rp = reinterpret_cast<TX*>(a.R);
  // TV: storage type of the vectors private to the solve that carry no accumulated state - z, s, p, q and the preconditioner's t / y (see PcgCgP::vec32)
  TV* const Zp = reinterpret_cast<TV*>(a.Z); TV* const Sp = reinterpret_cast<TV*>(a.S); TV* const Pp = reinterpret_cast<TV*>(a.P);
  TV* const Qp = reinterpret_cast<TV*>(a.Q); TV* const Yp = reinterpret_cast<TV*>(a.Y);
  (void)Zp; (void)Sp; (void)Pp; (void)Qp; (void)Yp;
  constexpr int NP = PW * (PW + 1) / 2, LD = pcg_cg_ld(PW);
  extern __shared__ double pcg_cg_smem[];
  double* Gs = pcg_cg_smem;
  const int na = a.ctl->nl[0];
  int wg_tile, wg_group;
  pcg_cg_wg(a.ntile, a.xcd_map, wg_tile, wg_group);
  if (wg_group * a.spw >= na) return;
  const int T = a.T, p = a.p, np = p * (p + 1) / 2;
  const int Tl = a.Tl, Tx = a.Tx;
  const int t0 = wg_tile * 64;
  const int nt = min(64, T - t0);
  pcg_stage_sym<NP>(a.GbT, T, t0, nt, np, Gs, LD);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool in = lane < nt;
  const int t = t0 + (in ? lane : 0);
  const double* g = Gs + lane * LD;
  const int s_end = min(na, (wg_group + 1) * a.spw);
  for (int si = wg_group * a.spw + wave; si < s_end; si += 4) {
    const size_t base = (size_t)a.live0[si] * a.sV + t;
    double r[PW], tv[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const size_t o = base + (size_t)(k < p ? k : 0) * Tx;
      r[k] = a.Gl[o] + a.KX[o];
    }
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const bool ok = in && k < p;
      if (ok) {
        a.Gt[base + (size_t)k * Tx] = r[k];
        if constexpr (X32) a.X32[base + (size_t)k * Tl] = 0.f; else a.X[base + (size_t)k * Tx] = 0.0;
        Rp[base + (size_t)k * Tl] = (TX)(-r[k]);
      }
      // (the residual the iteration goes on with is the STORED one: rounded when TX is float)
      r[k] = ok ? (double)(TX)(-r[k]) : 0.0;
    }
    pcg_sym_mv<PW, false>(g, 1, p, r, tv);
#pragma unroll
    for (int k = 0; k < PW; ++k)
      if (in && k < p) Yp[base + (size_t)k * Tl] = (TV)tv[k];
  }
}

// ---- the same step for 11 .. 20 latents (round 5: config 5's mode searches ran the split kernels of round 3 with K^-1 p as a product) -----------
// At 20 latents the packed triangle of a bin has 210 entries: the shared Gb / Wb tiles of 64 bins no longer fit LDS (161 KB) and the slot's own
// curvature no longer fits a lane's registers.  Here a workgroup tile is 32 bins (Gb 54 KB + Wb 27 KB), its 256 threads are 32 bins x 8 slots in
// flight, and every symmetric product walks the packed triangle in row chunks: two rows at a time out of LDS, five rows (at most 90 single-precision
// entries, all requested before any is used) of the slot's own curvature out of memory.  Vectors, scalars, lists and the closing of a step are those
// of the kernels above (PcgCgP; ntile counts 32-bin tiles).
constexpr int PCGW_TB = 32;
constexpr int PCGW_SL = 256 / PCGW_TB;          // slots a workgroup has in flight

// out += sign * M v over the packed lower triangle M (entry (hi, lo) at M[(hi (hi + 1) / 2 + lo) * cs]), rows in chunks of CH; v is zero and out
// ignored beyond p.  Chunks that start at or past row p are skipped; inside the last chunk rows >= p ARE read (an address clamped by the runtime p is
// an address register per entry: 210 of them, hoisted out of the slot loop, spilled) and their values dropped by a select - the LDS tiles hold zeros
// there, the packed curvature array has a slot's worth of slack behind it
template <int PW, int CH, int H0, typename TM>
__device__ __forceinline__ void pcgw_sym_mv_chunk(const TM* __restrict__ M, size_t cs, unsigned voff, int p, const double (&v)[PW], double (&out)[PW], double sign) {
  // (one instantiation per row chunk: the compiler gave up unrolling a loop over the chunks at 20 latents and put the entries into scratch)
  constexpr int H1 = (H0 + CH < PW) ? H0 + CH : PW;
  constexpr int BASE = H0 * (H0 + 1) / 2, NE = H1 * (H1 + 1) / 2 - BASE;
  if (H0 < p) {
    TM g[NE];
#pragma unroll
    for (int hi = H0; hi < H1; ++hi)
#pragma unroll
      for (int lo = 0; lo <= hi; ++lo) g[hi * (hi + 1) / 2 + lo - BASE] = (M + (size_t)(hi * (hi + 1) / 2 + lo) * cs)[voff];
#pragma unroll
    for (int hi = H0; hi < H1; ++hi)
#pragma unroll
      for (int lo = 0; lo <= hi; ++lo) {
        const double gg = (hi < p) ? sign * (double)g[hi * (hi + 1) / 2 + lo - BASE] : 0.0;
        out[hi] += gg * v[lo];
        if (lo != hi) out[lo] += gg * v[hi];
      }
    asm volatile("" ::: "memory");
  }
  if constexpr (H1 < PW) pcgw_sym_mv_chunk<PW, CH, H1>(M, cs, voff, p, v, out, sign);
}
// (M and cs wave-uniform, voff this lane's 32-bit element offset: the address of an entry is then a scalar base + a per-lane 32-bit offset - one
// register per lane for the whole chunk instead of a 64-bit address per entry: with two slots per wave the per-lane part cannot go into the base)
template <int PW, int CH, typename TM>
__device__ __forceinline__ void pcgw_sym_mv_acc(const TM* __restrict__ M, size_t cs, unsigned voff, int p, const double (&v)[PW], double (&out)[PW], double sign) {
  pcgw_sym_mv_chunk<PW, CH, 0>(M, cs, voff, p, v, out, sign);
}

template <int NPW, typename TM, typename TL>
__device__ __forceinline__ void pcgw_stage_sym(const TM* __restrict__ MT, int T, int t0, int nt, int np, TL* __restrict__ dst, int LD) {
  for (int e = threadIdx.x; e < NPW * PCGW_TB; e += 256) {
    const int c = e / PCGW_TB, t = e % PCGW_TB;
    dst[t * LD + c] = (t < nt && c < np) ? (TL)MT[(size_t)c * T + t0 + t] : (TL)0;
  }
}
inline size_t pcgw_a_lds(int pw) { return (size_t)PCGW_TB * pcg_cg_ld(pw) * (sizeof(double) + sizeof(float)); }
inline size_t pcgw_b_lds(int pw) { return (size_t)PCGW_TB * pcg_cg_ld(pw) * sizeof(double); }

// sum over the 32 lanes of a slot's half wave (lane 0 of the half holds the result)
__device__ __forceinline__ double pcgw_sum32(double x) {
  for (int off = PCGW_TB / 2; off > 0; off >>= 1) x += __shfl_down(x, off, PCGW_TB);
  return x;
}

// grid = (ceil(T / 32), slot groups of spw rounded up to 8), block = 256, dynamic LDS = pcgw_a_lds(PW)
template <int PW, typename TV = double>
__global__ __launch_bounds__(256, 1) void pcgw_a_kernel(PcgCgP a) {
  // TV: storage type of the vectors private to the solve that carry no accumulated state - z, s, p, q and the preconditioner's t / y (see PcgCgP::vec32)
  TV* const Zp = reinterpret_cast<TV*>(a.Z); TV* const Sp = reinterpret_cast<TV*>(a.S); TV* const Pp = reinterpret_cast<TV*>(a.P);
  TV* const Qp = reinterpret_cast<TV*>(a.Q); TV* const Yp = reinterpret_cast<TV*>(a.Y);
  (void)Zp; (void)Sp; (void)Pp; (void)Qp; (void)Yp;
  constexpr int NP = PW * (PW + 1) / 2, LD = pcg_cg_ld(PW);
  extern __shared__ double pcg_cg_smem[];
  double* Gs = pcg_cg_smem;
  float* Ws = reinterpret_cast<float*>(pcg_cg_smem + PCGW_TB * LD);
  PcgCtl* ctl = a.ctl;
  if (ctl->stop) return;
  const int na = ctl->nl[a.par];
  if (a.fold_close && a.step > 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && ctl->pad_[0] == a.step - 1) pcg_close_step(ctl, a.par ^ 1, a.host);
  int wg_tile, wg_group;
  pcg_cg_wg(a.ntile, a.xcd_map, wg_tile, wg_group);
  if (wg_group * a.spw >= na) return;
  const int T = a.T, p = a.p, np = p * (p + 1) / 2, Tl = a.Tl;
  const int t0 = wg_tile * PCGW_TB;
  const int nt = min(PCGW_TB, T - t0);
  pcgw_stage_sym<NP>(a.GbT, T, t0, nt, np, Gs, LD);
  pcgw_stage_sym<NP>(a.WbT, T, t0, nt, np, Ws, LD);
  __syncthreads();
  const int lane = threadIdx.x % PCGW_TB, sl = threadIdx.x / PCGW_TB;
  const bool in = lane < nt;
  const int t = t0 + (in ? lane : 0);
  const int* live = a.par ? a.live1 : a.live0;
  const int s_end = min(na, (wg_group + 1) * a.spw);
  // (every lane of a half wave walks the same slots: the shuffles below are executed by all of them)
  for (int sb = wg_group * a.spw; sb < s_end; sb += PCGW_SL) {
    const int si = sb + sl;
    const bool have = si < s_end;
    const size_t slot = (size_t)live[have ? si : s_end - 1];
    const size_t base = slot * a.sV + t;
    double r[PW], v[PW], z[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const size_t o = base + (size_t)(k < p ? k : 0) * Tl;
      r[k] = a.R[o];
      v[k] = (double)Yp[o];
    }
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const bool ok = in && k < p;
      r[k] = ok ? r[k] : 0.0;
      v[k] = ok ? a.eps * r[k] + v[k] : 0.0;
      z[k] = 0.0;
    }
    pcgw_sym_mv_acc<PW, 2>(Gs, 1, (unsigned)(lane * LD), p, v, z, 1.0);               // z = Gb (eps r + y)
#pragma unroll
    for (int k = 0; k < PW; ++k) v[k] = 0.0;
    pcgw_sym_mv_acc<PW, 2>(Ws, 1, (unsigned)(lane * LD), p, z, v, -1.0);             // v = -Wb z
    pcgw_sym_mv_acc<PW, 5>(a.W32T, (size_t)a.Tw, (unsigned)(slot * a.sW32 + t), p, z, v, 1.0);   // v += fl32(W) z  (offsets below 2^32 elements: the host checks)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const double sk = r[k] + v[k];                                 // s = H~ z = (r - Wb z) + W z
      s0 += r[k] * z[k]; s1 += z[k] * sk; s2 += r[k] * r[k];
      if (have && in && k < p) {
        const size_t o = base + (size_t)k * Tl;
        Zp[o] = (TV)z[k];
        Sp[o] = (TV)sk;
      }
    }
    s0 = pcgw_sum32(s0); s1 = pcgw_sum32(s1); s2 = pcgw_sum32(s2);
    if (have && lane == 0) {
      double* pp = a.part + (slot * a.ntile + wg_tile) * 3;
      pp[0] = s0; pp[1] = s1; pp[2] = s2;
    }
  }
}

// t = Gb r for this lane's bin; r is zero beyond p / past the tile
template <int PW, typename TV>
__device__ __forceinline__ void pcgw_apply_store(const double* __restrict__ g, int p, const double (&r)[PW], TV* __restrict__ Y, size_t base, int Tl, bool ok) {
  double tv[PW];
#pragma unroll
  for (int k = 0; k < PW; ++k) tv[k] = 0.0;
  pcgw_sym_mv_acc<PW, 2>(g, 1, 0u, p, r, tv, 1.0);
#pragma unroll
  for (int k = 0; k < PW; ++k)
    if (ok && k < p) Y[base + (size_t)k * Tl] = (TV)tv[k];
}

// same launch shape; dynamic LDS = pcgw_b_lds(PW)
template <int PW, typename TV = double>
__global__ __launch_bounds__(256, 2) void pcgw_b_kernel(PcgCgP a) {
  // TV: storage type of the vectors private to the solve that carry no accumulated state - z, s, p, q and the preconditioner's t / y (see PcgCgP::vec32)
  TV* const Zp = reinterpret_cast<TV*>(a.Z); TV* const Sp = reinterpret_cast<TV*>(a.S); TV* const Pp = reinterpret_cast<TV*>(a.P);
  TV* const Qp = reinterpret_cast<TV*>(a.Q); TV* const Yp = reinterpret_cast<TV*>(a.Y);
  (void)Zp; (void)Sp; (void)Pp; (void)Qp; (void)Yp;
  constexpr int NP = PW * (PW + 1) / 2, LD = pcg_cg_ld(PW);
  extern __shared__ double pcg_cg_smem[];
  double* Gs = pcg_cg_smem;
  PcgCtl* ctl = a.ctl;
  if (ctl->stop) return;
  const int na = ctl->nl[a.par];
  int wg_tile, wg_group;
  pcg_cg_wg(a.ntile, a.xcd_map, wg_tile, wg_group);
  if (wg_group * a.spw >= na) return;
  const int it = a.fold_close ? a.step : ctl->iters;
  const int T = a.T, p = a.p, np = p * (p + 1) / 2, Tl = a.Tl, Tx = a.Tx;
  const int t0 = wg_tile * PCGW_TB;
  const int nt = min(PCGW_TB, T - t0);
  pcgw_stage_sym<NP>(a.GbT, T, t0, nt, np, Gs, LD);
  __syncthreads();
  const int lane = threadIdx.x % PCGW_TB, sl = threadIdx.x / PCGW_TB;
  const bool in = lane < nt;
  const int t = t0 + (in ? lane : 0);
  const double* g = Gs + lane * LD;
  const int* live = a.par ? a.live1 : a.live0;
  int* live_next = a.par ? a.live0 : a.live1;
  const double* gam_old = a.gam + (size_t)a.par * a.B;
  const double* alp_old = a.alp + (size_t)a.par * a.B;
  double* gam_new = a.gam + (size_t)(a.par ^ 1) * a.B;
  double* alp_new = a.alp + (size_t)(a.par ^ 1) * a.B;
  const int s_end = min(na, (wg_group + 1) * a.spw);
  for (int si = wg_group * a.spw + sl; si < s_end; si += PCGW_SL) {
    const int sloti = live[si];
    const size_t slot = (size_t)sloti;
    double gamma = 0.0, delta = 0.0, rrn = 0.0;
    for (int i = 0; i < a.ntile; ++i) {
      const double* pp = a.part + (slot * a.ntile + i) * 3;
      gamma += pp[0]; delta += pp[1]; rrn += pp[2];
    }
    const double b0 = a.first ? rrn : a.rr0[slot];
    const float ratio = (b0 > 0.0) ? (float)sqrt(rrn / b0) : 0.0f;
    const bool keep = (it < a.inner_min || !(ratio <= a.eta[slot]));
    const double g_old = gam_old[slot], a_old = alp_old[slot];
    const double beta = (a.first || !(g_old > 0.0)) ? 0.0 : gamma / g_old;
    const double den = (a.first || !(a_old > 0.0)) ? delta : delta - beta * gamma / a_old;
    const double alpha = (den > 0.0) ? gamma / den : 0.0;
    if (wg_tile == 0 && lane == 0) {
      if (a.first) a.rr0[slot] = rrn;
      a.rr[slot] = rrn;
      if (keep) {
        gam_new[slot] = gamma;
        alp_new[slot] = alpha;
        const int pos = atomicAdd(&ctl->nl[a.par ^ 1], 1);
        live_next[pos] = sloti;
      }
    }
    if (!keep) continue;
    const size_t base = slot * a.sV + t;
    double r[PW];
    // the vector updates five latents at a time (thirty loads in flight), the new residual stays in registers for t = Gb r
#pragma unroll
    for (int k0 = 0; k0 < PW; k0 += 5) {
      double zk[5], sk[5], po[5], qo[5], xo[5], ro[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int k = k0 + j;
        const size_t o = base + (size_t)(k < p ? k : 0) * Tl;
        zk[j] = (double)Zp[o]; sk[j] = (double)Sp[o]; ro[j] = a.R[o]; xo[j] = a.X[base + (size_t)(k < p ? k : 0) * Tx];
        po[j] = 0.0; qo[j] = 0.0;
      }
      if (!a.first) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const size_t o = base + (size_t)(k0 + j < p ? k0 + j : 0) * Tl;
          po[j] = (double)Pp[o]; qo[j] = (double)Qp[o];
        }
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int k = k0 + j;
        const bool ok = in && k < p;
        const size_t o = base + (size_t)(k < p ? k : 0) * Tl;
        const double pn = zk[j] + beta * po[j];
        const double qn = sk[j] + beta * qo[j];
        if (k < PW) r[k < PW ? k : 0] = ok ? ro[j] - alpha * qn : 0.0;
        if (ok) {
          Pp[o] = (TV)pn;
          Qp[o] = (TV)qn;
          a.X[base + (size_t)k * Tx] = xo[j] + alpha * pn;
          a.R[o] = ro[j] - alpha * qn;
        }
      }
    }
    pcgw_apply_store<PW>(g, p, r, Yp, base, Tl, in);
  }
}

// first kernel of a solve (see pcg_cg_start_kernel); launch shape and LDS of pcgw_b_kernel
template <int PW, typename TV = double>
__global__ __launch_bounds__(256, 2) void pcgw_start_kernel(PcgCgP a) {
  // TV: storage type of the vectors private to the solve that carry no accumulated state - z, s, p, q and the preconditioner's t / y (see PcgCgP::vec32)
  TV* const Zp = reinterpret_cast<TV*>(a.Z); TV* const Sp = reinterpret_cast<TV*>(a.S); TV* const Pp = reinterpret_cast<TV*>(a.P);
  TV* const Qp = reinterpret_cast<TV*>(a.Q); TV* const Yp = reinterpret_cast<TV*>(a.Y);
  (void)Zp; (void)Sp; (void)Pp; (void)Qp; (void)Yp;
  constexpr int NP = PW * (PW + 1) / 2, LD = pcg_cg_ld(PW);
  extern __shared__ double pcg_cg_smem[];
  double* Gs = pcg_cg_smem;
  const int na = a.ctl->nl[0];
  int wg_tile, wg_group;
  pcg_cg_wg(a.ntile, a.xcd_map, wg_tile, wg_group);
  if (wg_group * a.spw >= na) return;
  const int T = a.T, p = a.p, np = p * (p + 1) / 2, Tl = a.Tl, Tx = a.Tx;
  const int t0 = wg_tile * PCGW_TB;
  const int nt = min(PCGW_TB, T - t0);
  pcgw_stage_sym<NP>(a.GbT, T, t0, nt, np, Gs, LD);
  __syncthreads();
  const int lane = threadIdx.x % PCGW_TB, sl = threadIdx.x / PCGW_TB;
  const bool in = lane < nt;
  const int t = t0 + (in ? lane : 0);
  const double* g = Gs + lane * LD;
  const int s_end = min(na, (wg_group + 1) * a.spw);
  for (int si = wg_group * a.spw + sl; si < s_end; si += PCGW_SL) {
    const size_t base = (size_t)a.live0[si] * a.sV + t;
    double r[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const size_t o = base + (size_t)(k < p ? k : 0) * Tx;
      r[k] = a.Gl[o] + a.KX[o];
    }
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const bool ok = in && k < p;
      if (ok) {
        a.Gt[base + (size_t)k * Tx] = r[k];
        a.X[base + (size_t)k * Tx] = 0.0;
        a.R[base + (size_t)k * Tl] = -r[k];
      }
      r[k] = ok ? -r[k] : 0.0;
    }
    pcgw_apply_store<PW>(g, p, r, Yp, base, Tl, in);
  }
}

// M[t][p][p] (double, symmetric) -> out[c][T], c = a(a+1)/2 + b over the lower triangle a >= b.  grid = ceil(T*NP/256)
template <typename TO>
__global__ void pack_sym_t_kernel(const double* __restrict__ M, TO* __restrict__ out, int T, int p) {
  const int np = p * (p + 1) / 2;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= T * np) return;
  const int c = e / T, t = e - c * T;
  int a = 0;
  while ((a + 1) * (a + 2) / 2 <= c) ++a;
  const int b = c - a * (a + 1) / 2;
  out[e] = (TO)M[(size_t)t * p * p + a * p + b];
}

// W[slot][t][p][p] (double) -> Wp[slot][c][T] (float, c over the lower triangle) for the listed slots: 64 bins per workgroup, packed
// and transposed through LDS so that reads walk a bin's block and writes walk the bins.  grid = (ceil(T/64), nslots), block = 256,
// dynamic LDS = NP * 65 floats
inline __global__ __launch_bounds__(256) void pack_w32t_kernel(const double* __restrict__ W, long long sW, float* __restrict__ Wp, long long sWp, int Tw,
                                                        int T, int p, const int* __restrict__ slots) {
  extern __shared__ float w32t_tile[];
  const size_t slot = slots[blockIdx.y];
  const int np = p * (p + 1) / 2, pp = p * p;
  const int t0 = blockIdx.x * 64, nt = min(64, T - t0);
  for (int e = threadIdx.x; e < nt * np; e += 256) {
    const int t = e / np, c = e - t * np;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= c) ++a;
    const int b = c - a * (a + 1) / 2;
    w32t_tile[c * 65 + t] = (float)W[slot * sW + (size_t)(t0 + t) * pp + a * p + b];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < np * 64; e += 256) {
    const int c = e >> 6, t = e & 63;
    if (t < nt) Wp[slot * sWp + (size_t)c * Tw + t0 + t] = w32t_tile[c * 65 + t];
  }
}

// Commit of an accepted line-search step, curvature part: W <- Wt for the listed slots AND its packed single-precision form (pack_w32t_kernel's
// output) in the same pass over Wt.  grid = (ceil(T/64), nslots), block = 256, dynamic LDS = NP * 65 floats
inline __global__ __launch_bounds__(256) void commit_w_pack_kernel(const double* __restrict__ Wt, double* __restrict__ W, long long sW, float* __restrict__ Wp,
                                                            long long sWp, int Tw, int T, int p, const int* __restrict__ slots) {
  extern __shared__ float w32t_tile[];
  const size_t slot = slots[blockIdx.y];
  const int np = p * (p + 1) / 2, pp = p * p;
  const int t0 = blockIdx.x * 64, nt = min(64, T - t0);
  const size_t base = slot * sW + (size_t)t0 * pp;
  for (int e = threadIdx.x; e < nt * pp; e += 256) {
    const double v = Wt[base + e];
    W[base + e] = v;
    const int t = e / pp, idx = e - t * pp, a = idx / p, b = idx - a * p;
    if (b <= a) w32t_tile[(a * (a + 1) / 2 + b) * 65 + t] = (float)v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < np * 64; e += 256) {
    const int c = e >> 6, t = e & 63;
    if (t < nt) Wp[slot * sWp + (size_t)c * Tw + t0 + t] = w32t_tile[c * 65 + t];
  }
}

}  // namespace pgpfa
