// Synthetic population on the device (reference recipe util.py:705-750: x_k ~ N(0, K(tau_k)), y_nt ~ Poisson(exp(c_n.x_t + d_n))).
// Latents are drawn through the resident low-rank form of the Gram matrices, K_k = eps I + F_k F_k^T (pivoted Cholesky of the
// RBF part to 1e-13): x_k = F_k z1 + sqrt(eps) z2 with independent standard normals z1 (r_k), z2 (T) has covariance K_k.  No
// (xdim*T)^2 matrix, no SVD per trial, and the counts land directly in the context's packed uint8 tensor.
// Random numbers: Philox4x32-10 (counter-based: element index and a stream tag in the counter, the seed in the key), so a
// sample is a pure function of (seed, trial, latent / neuron, bin) - independent of launch geometry and of the order of calls.
// This is NOT NumPy's legacy stream: funs.util.dataset(sampler='reference') keeps that one (host) for the golden fixtures.
#pragma once

namespace pgpfa {

struct Philox {
  unsigned k0, k1;
  __device__ void round(unsigned (&c)[4], unsigned ka, unsigned kb) const {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
    c[0] = h1 ^ c[1] ^ ka; c[1] = l1; c[2] = h0 ^ c[3] ^ kb; c[3] = l0;
  }
  __device__ void operator()(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned (&out)[4]) const {
    unsigned c[4] = {c0, c1, c2, c3};
    unsigned ka = k0, kb = k1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, ka, kb);
      ka += 0x9E3779B9u; kb += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
};

// 53-bit uniform in (0, 1) from two words
__device__ inline double u01(unsigned a, unsigned b) {
  const unsigned long long m = (((unsigned long long)a << 32) | b) >> 11;       // 53 bits
  return ((double)m + 0.5) * (1.0 / 9007199254740992.0);
}

// two standard normals from four words (Box-Muller)
__device__ inline void normal2(const unsigned (&w)[4], double& n0, double& n1) {
  const double u = u01(w[0], w[1]), v = u01(w[2], w[3]);
  const double rad = sqrt(-2.0 * log(u));
  double s, c;
  sincos(6.283185307179586476925 * v, &s, &c);
  n0 = rad * c; n1 = rad * s;
}

// X[trial][k][t] = sum_j F_k[t][j] z1[j] + sqrt(eps) z2[t].  grid = (p, ntrials), block = 256, dynamic LDS = rmax doubles.
// Counter layout: (index, trial, latent, stream) with stream 1 = z1 pairs, 2 = z2 pairs.
inline __global__ __launch_bounds__(256) void sample_latents_kernel(const double* __restrict__ F, int Tf, int T, int p, const int* __restrict__ rank,
                                                             double eps, unsigned long long seed, const int* __restrict__ trials,
                                                             double* __restrict__ X) {
  extern __shared__ double z1[];
  const int k = blockIdx.x;
  const unsigned trial = (unsigned)trials[blockIdx.y];
  const int r = rank[k];
  const Philox rng{(unsigned)seed, (unsigned)(seed >> 32)};
  for (int j2 = threadIdx.x; 2 * j2 < r; j2 += 256) {
    unsigned w[4];
    rng((unsigned)j2, trial, (unsigned)k, 1u, w);
    double a, b;
    normal2(w, a, b);
    z1[2 * j2] = a;
    if (2 * j2 + 1 < r) z1[2 * j2 + 1] = b;
  }
  __syncthreads();
  const double* Fk = F + (size_t)k * Tf * Tf;
  double* x = X + ((size_t)trial * p + k) * T;
  const double se = sqrt(eps);
  for (int t = threadIdx.x; t < T; t += 256) {
    unsigned w[4];
    rng((unsigned)(t >> 1), trial, (unsigned)k, 2u, w);
    double a, b;
    normal2(w, a, b);
    double acc = se * ((t & 1) ? b : a);
    for (int j = 0; j < r; ++j) acc += Fk[(size_t)j * Tf + t] * z1[j];
    x[t] = acc;
  }
}

// Poisson(lam) from uniforms of the stream (n, t, trial): sequential inversion below 12 (expected lam + 1 steps), Hoermann's
// transformed rejection (PTRS, 1993) above; every uniform pair comes from its own counter (attempt index in the low word).
__device__ inline unsigned poisson_draw(double lam, const Philox& rng, unsigned c1, unsigned c2) {
  unsigned w[4];
  if (!(lam > 0.0)) return 0u;
  if (lam < 12.0) {
    rng(0u, c1, c2, 3u, w);
    const double u = u01(w[0], w[1]);
    double pk = exp(-lam), cdf = pk;
    unsigned kk = 0;
    while (u > cdf && kk < 1000u) {
      ++kk;
      pk *= lam / (double)kk;
      cdf += pk;
    }
    return kk;
  }
  const double slam = sqrt(lam), loglam = log(lam);
  const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
  const double inv_alpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  for (unsigned attempt = 1; attempt < 4096u; ++attempt) {
    rng(attempt, c1, c2, 3u, w);
    const double U = u01(w[0], w[1]) - 0.5, V = u01(w[2], w[3]);
    const double us = 0.5 - fabs(U);
    const double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return (unsigned)kf;
    if (kf < 0.0 || (us < 0.013 && V > us)) continue;
    if (log(V) + log(inv_alpha) - log(a / (us * us) + b) <= -lam + kf * loglam - lgamma(kf + 1.0)) return (unsigned)kf;
  }
  return (unsigned)(lam + 0.5);
}

// Y[trial][n][t] ~ Poisson(exp(c_n . x_t + d_n)); low bytes to Y, high bytes to Yhi when that plane exists - otherwise counts above 255
// are flagged (overflow[0]) and the caller repeats the launch with the plane allocated (a draw is a pure function of its counters);
// counts above 65535 are flagged in overflow[1].
// grid = (ceil(T/64), q, ntrials), block = 64.
inline __global__ __launch_bounds__(64) void sample_counts_kernel(const double* __restrict__ X, const double* __restrict__ C, const double* __restrict__ d, int q, int p,
                                     int T, unsigned long long seed, const int* __restrict__ trials, uint8_t* __restrict__ Y,
                                     uint8_t* __restrict__ Yhi, int* __restrict__ overflow) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int n = blockIdx.y;
  const unsigned trial = (unsigned)trials[blockIdx.z];
  if (t >= T) return;
  const double* x = X + (size_t)trial * p * T + t;
  double h = d[n];
  for (int k = 0; k < p; ++k) h += C[(size_t)n * p + k] * x[(size_t)k * T];
  const Philox rng{(unsigned)seed, (unsigned)(seed >> 32)};
  const unsigned v = poisson_draw(exp(h), rng, trial, (unsigned)n * 65536u + (unsigned)t);
  if (v > 255u && !Yhi) atomicAdd(overflow, 1);
  if (v > 65535u) atomicAdd(overflow + 1, 1);
  Y[((size_t)trial * q + n) * T + t] = (uint8_t)(v & 255u);
  if (Yhi) Yhi[((size_t)trial * q + n) * T + t] = (uint8_t)((v >> 8) & 255u);
}

}  // namespace pgpfa
