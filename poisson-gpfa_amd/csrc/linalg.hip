// libpgpfa_hip.so - linalg.hip (one translation unit of the C-ABI library; shared declarations: ctx.h)
#include "ctx.h"
#include "model.h"
#include "gemm.h"
#include "chol.h"

using namespace pgpfa;

// algorithmic flops of one GEMM launch (useful multiply-adds x2, triangular structure respected)
double gemm_flops(const GemmP& g) {
  if (g.flops_hint > 0.0) return g.flops_hint;
  const double M = g.M, N = g.N, K = g.K;
  double per;
  if (g.mode == GEMM_LOWER) {
    // trapezoid i >= j, j < N <= M
    const double elems = N * (M - N) + N * (N + 1) / 2.0;
    per = 2.0 * K * elems;
  } else if (g.kflags & KF_BEGIN_ROW) {
    per = 2.0 * N * (M * K - M * (M - 1) / 2.0);         // row i uses k >= i
  } else if (g.kflags & KF_BEGIN_MAXRC) {
    per = 0.0;
    // sum_{i,j} (K - max(i,j)) : for M == N == K this is ~ K^3/3 * 2
    const double m = std::min(M, N);
    per = 2.0 * (M * N * K - (m * (m - 1) * (m + 1) / 3.0 + (M > N ? N * (M - N) * (M + N - 1) / 2.0 : M * (N - M) * (M + N - 1) / 2.0)));
  } else {
    per = 2.0 * M * N * K;
  }
  return per * g.nbatch;
}

// (f32: operands are single precision - pointers carried as double*, strides in elements - on the FP32 matrix cores; no split-K)
int gemm(pgpfa_ctx* c, bool transb, GemmP g, bool f32) {
  // (products over a device-side live list: how many columns a launch really has is only known after the solve - their launches are
  // recorded with their time and no flops, the flops are added per shape once the slot-iterations are known: prof_live_flops)
  const bool live_cols = g.cols && c->cur_ndev;
  prof_begin(c, TAG_GEMM, live_cols ? 0.0 : gemm_flops(g));
  if (c->prof.on && c->prof.open) {
    char key[176];
    std::snprintf(key, sizeof key, "%s %s M=%d N=%s%d K=%d%s batch=%d%s%s%s", f32 ? "f32" : "f64", transb ? "NN" : "NT", g.M, live_cols ? "<=" : "", g.N,
                  g.K, g.kseg ? " (segmented)" : "", std::max(g.nbatch, 1), g.mode == GEMM_LOWER ? " lower" : "",
                  g.kflags ? " triangular-k" : "", g.rtab ? " block-sparse" : "");
    c->prof.recs.back().shape = key;
    if (live_cols && c->live_gemm_collect) c->live_gemms.emplace_back(key, gemm_flops(g) / std::max(g.N, 1));
  }
  // Tile size: products that offer few 128 x 128 tiles (multi-RHS vectors against the block-diagonal factors and the r x r
  // preconditioner, K^-1 p, the short panels of the r x r factorisations) run on 64 x 64 tiles - four times the workgroups, 3-4 of
  // them resident per CU; everything with a row-tile table is laid out for 64-row tiles.
  if (g.bm == 0) {
    const long long t128 = (long long)((g.M + GBM - 1) / GBM) * ((g.N + GBN - 1) / GBN) * std::max(g.nbatch, 1);
    g.bm = ((!f32 || c->f32_tile64) && c->mfma && (g.rtab || t128 < c->small_tile_below)) ? 64 : 128;
  }
  if (g.rtab) g.bm = 64;
  if (g.cols && c->cur_ndev) g.n_dev = c->cur_ndev;
  // in-place products (the TRSM of the factorisation writes its own A panel: one 128-wide column tile reads all of it before it stores)
  // must keep the tile that covers the whole panel
  if ((const double*)g.C == g.A || (const double*)g.C == g.B) g.bm = 128;
  // Few output tiles and a long k loop: the launch would occupy a fraction of the 256 CUs for the length of one k loop.  Cut k into
  // parts run as extra batch entries, sum the partial products afterwards.
  const int tiles = (g.rtab ? g.ntab : (g.M + g.bm - 1) / g.bm) * ((g.N + g.bm - 1) / g.bm) * std::max(g.nbatch, 1);
  int ksplit = 1;
  // (block-sparse operands: the k loop a tile really runs is the one implied by the flop count)
  const double k_eff = g.k_loop_hint > 0 ? (double)g.k_loop_hint
                       : (g.flops_hint > 0.0 && g.M > 0 && g.N > 0) ? g.flops_hint / (2.0 * g.M * g.N * std::max(g.nbatch, 1)) : (double)g.K;
  const int split_below = g.bm == 64 ? c->splitk_below64 : 384;
  if (!f32 && c->gemm_part && g.mode == GEMM_FULL && g.kflags == 0 && g.nb_lo == 0 && g.kseg == 0 && tiles > 0 && tiles < split_below && k_eff >= 128.0) {
    // (a lone workgroup per CU walks its k loop at the latency of one global load per 16-wide step: with a handful of
    // tiles even a 128-long loop is worth cutting, down to parts of two steps)
    ksplit = std::min(std::min(8, (int)(k_eff / (tiles < 64 ? 32.0 : 64.0))), (c->splitk_target + tiles - 1) / tiles);
    while (ksplit > 1 && (size_t)ksplit * g.nbatch * g.M * g.N > c->gemm_part_len) --ksplit;
  }
  hipError_t e;
  if (ksplit > 1) {
    GemmP s = g;
    s.C = c->gemm_part; s.sC = (long long)g.M * g.N; s.ldc = g.M; s.beta = 0.0;
    s.nb_lo = g.nbatch; s.nbatch = g.nbatch * ksplit; s.sA_hi = 0; s.sB_hi = 0; s.sC_hi = (long long)g.nbatch * g.M * g.N;
    s.ksplit = ksplit;
    s.c_by_pos = 1;                      // partial products are indexed by batch position, the operands by slot
    s.cols_c_off = 1;                    // ... and by column position: the reduction applies the column list
    e = gemm_launch(c->st, c->mfma, transb, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)(((size_t)g.M * g.N + 255) / 256), g.nbatch), dim3(256), 0, c->st, c->gemm_part,
                         ksplit, g.M, g.N, g.nbatch, g.C, g.sC, g.ldc, g.slots, g.beta, g.skip, g.cols, g.n_dev);
      e = hipGetLastError();
    }
  } else {
    e = gemm_launch(c->st, c->mfma, transb, g, f32);
  }
  prof_end(c);
  if (e != hipSuccess) return fail("gemm launch failed: %s", hipGetErrorString(e));
  return 0;
}

// chol_factor / chol_inverse_t with per-launch profiling (same sequence as chol.h's plain versions)
// (f32: the slabs of w hold single-precision matrices - same pointers reinterpreted, strides in elements)
int factor(pgpfa_ctx* c, const CholWS& w, const int* slots, int nb, bool f32) {
  auto at = [f32](double* base, size_t off) { return f32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(base) + off) : base + off; };
  const int np = w.npad, ld = w.ld;
  const int na = (w.nact > 0 && w.nact <= np) ? w.nact : np;   // rows >= na are identity padding: never updated
  for (int c0 = 0; c0 < np; c0 += NSUP) {
    const int c1 = std::min(c0 + NSUP, np);
    for (int k0 = c0; k0 < c1; k0 += NB) {
      prof_begin(c, TAG_POTRF, 2.0 * nb * (double)NB * NB * NB / 3.0);
      if (f32)
        hipLaunchKernelGGL(potrf_diag_kernel_t<float>, dim3(nb), dim3(512), 0, c->st, reinterpret_cast<float*>(w.H), w.sH, ld, k0,
                           reinterpret_cast<float*>(w.Dinv), w.sD, slots, w.info);
      else
        hipLaunchKernelGGL(potrf_diag_kernel_t<double>, dim3(nb), dim3(512), 0, c->st, w.H, w.sH, ld, k0, w.Dinv, w.sD, slots, w.info);
      prof_end(c);
      const int r0 = k0 + NB;
      if (r0 >= np) break;
      GemmP g{};
      g.A = at(w.H, (size_t)k0 * ld + r0); g.sA = w.sH; g.lda = ld;
      g.B = at(w.Dinv, (size_t)(k0 / NB) * NB * NB); g.sB = w.sD; g.ldb = NB;
      g.C = at(w.H, (size_t)k0 * ld + r0); g.sC = w.sH; g.ldc = ld;
      g.M = na - r0; g.N = NB; g.K = NB; g.alpha = 1.0; g.beta = 0.0;
      g.slots = slots; g.nbatch = nb; g.mode = GEMM_FULL; g.kflags = 0;
      if (g.M > 0) CHK(gemm(c, false, g, f32));
      if (r0 < c1 && r0 < na) {
        GemmP s{};
        s.A = at(w.H, (size_t)k0 * ld + r0); s.sA = w.sH; s.lda = ld;
        s.B = s.A; s.sB = w.sH; s.ldb = ld;
        s.C = at(w.H, (size_t)r0 * ld + r0); s.sC = w.sH; s.ldc = ld;
        s.M = na - r0; s.N = std::min(c1, na) - r0; s.K = NB; s.alpha = -1.0; s.beta = 1.0;
        s.slots = slots; s.nbatch = nb; s.mode = GEMM_LOWER; s.kflags = KF_MASK_DIAG;
        CHK(gemm(c, false, s, f32));
      }
    }
    if (c1 < na) {
      GemmP s{};
      s.A = at(w.H, (size_t)c0 * ld + c1); s.sA = w.sH; s.lda = ld;
      s.B = s.A; s.sB = w.sH; s.ldb = ld;
      s.C = at(w.H, (size_t)c1 * ld + c1); s.sC = w.sH; s.ldc = ld;
      s.M = na - c1; s.N = na - c1; s.K = c1 - c0; s.alpha = -1.0; s.beta = 1.0;
      s.slots = slots; s.nbatch = nb; s.mode = GEMM_LOWER; s.kflags = KF_MASK_DIAG;
      CHK(gemm(c, false, s, f32));
    }
  }
  HIPC(hipGetLastError());
  return 0;
}

int inverse_t(pgpfa_ctx* c, const CholWS& w, const int* slots, int nb, bool f32) {
  auto at = [f32](double* base, size_t off) { return f32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(base) + off) : base + off; };
  const int np = w.npad, ld = w.ld;
  const int na = (w.nact > 0 && w.nact <= np) ? w.nact : np;   // rows >= na of Mt are identity padding
  if (f32)
    hipLaunchKernelGGL(diag_transpose_kernel_t<float>, dim3(nb, np / NB), dim3(256), 0, c->st, reinterpret_cast<float*>(w.Mt), w.sM, ld, 0,
                       reinterpret_cast<const float*>(w.Dinv), w.sD, slots);
  else
    hipLaunchKernelGGL(diag_transpose_kernel_t<double>, dim3(nb, np / NB), dim3(256), 0, c->st, w.Mt, w.sM, ld, 0, w.Dinv, w.sD, slots);
  for (int j0 = NB; j0 < np; j0 += NB) {
    // columns of this block that are not identity padding (na is a multiple of 64): the padding columns of L^-T stay zero above the
    // diagonal (the slab was cleared), so a half-padded last block costs half
    const int nbw = std::min(NB, na - j0);
    if (nbw <= 0) continue;
    GemmP a{};
    a.A = w.Mt; a.sA = w.sM; a.lda = ld;
    a.B = at(w.H, j0); a.sB = w.sH; a.ldb = ld;
    a.C = w.P; a.sC = w.sP; a.ldc = np;
    a.M = std::min(j0, na); a.N = nbw; a.K = j0; a.alpha = 1.0; a.beta = 0.0;
    a.slots = slots; a.nbatch = nb; a.mode = GEMM_FULL; a.kflags = KF_BEGIN_ROW;
    CHK(gemm(c, false, a, f32));
    GemmP b{};
    b.A = w.P; b.sA = w.sP; b.lda = np;
    b.B = at(w.Dinv, (size_t)(j0 / NB) * NB * NB); b.sB = w.sD; b.ldb = NB;
    b.C = at(w.Mt, (size_t)j0 * ld); b.sC = w.sM; b.ldc = ld;
    b.M = std::min(j0, na); b.N = nbw; b.K = nbw; b.alpha = -1.0; b.beta = 0.0;
    b.slots = slots; b.nbatch = nb; b.mode = GEMM_FULL; b.kflags = 0;
    CHK(gemm(c, false, b, f32));
  }
  HIPC(hipGetLastError());
  return 0;
}


// ---- test / bench hooks ----------------------------------------------------------------------------------------
int pgpfa_test_potrf(pgpfa_ctx* c, int batch, int n, const double* A, double* L, double* inv) {
  if (!c || !A || !L) return fail("null argument");
  if (batch < 1 || n < 1) return fail("invalid sizes");
  HIPC(hipSetDevice(c->device));
  const int np = round_up(n, NB);
  CholWS w{};
  const size_t mark = c->allocs.size();
  CHK(alloc_cholws(c, &w, batch, np, true));
  const size_t slab = (size_t)np * np;
  std::vector<double> h(slab * batch, 0.0);
  for (int b = 0; b < batch; ++b) {
    double* s = h.data() + slab * b;
    for (int j = 0; j < np; ++j)
      for (int i = 0; i < np; ++i) s[(size_t)j * np + i] = (i < n && j < n) ? A[((size_t)b * n + i) * n + j] : (i == j ? 1.0 : 0.0);
  }
  int rc = upload(c, w.H, h.data(), slab * batch);
  if (!rc) rc = factor(c, w, nullptr, batch);
  if (!rc) rc = download(c, h.data(), w.H, slab * batch);
  if (!rc) {
    for (int b = 0; b < batch; ++b)
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) L[((size_t)b * n + i) * n + j] = (j <= i) ? h[slab * b + (size_t)j * np + i] : 0.0;
  }
  if (!rc && inv) {
    rc = inverse_t(c, w, nullptr, batch);
    GemmP g{};
    g.A = w.Mt; g.sA = w.sM; g.lda = np; g.B = w.Mt; g.sB = w.sM; g.ldb = np;
    g.C = w.H; g.sC = w.sH; g.ldc = np; g.M = np; g.N = np; g.K = np; g.alpha = 1.0; g.beta = 0.0;
    g.slots = nullptr; g.nbatch = batch; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
    if (!rc) rc = gemm(c, false, g);
    if (!rc) rc = download(c, h.data(), w.H, slab * batch);
    if (!rc)
      for (int b = 0; b < batch; ++b)
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) inv[((size_t)b * n + i) * n + j] = h[slab * b + (size_t)j * np + i];
  }
  std::vector<int> info(batch, 0);
  if (!rc) {
    hipMemcpy(info.data(), w.info, sizeof(int) * batch, hipMemcpyDeviceToHost);
    for (int b = 0; b < batch; ++b)
      if (info[b] != 0) rc = fail("matrix %d is not positive definite (pivot %d)", b, info[b]);
  }
  hipStreamSynchronize(c->st);
  while (c->allocs.size() > mark) { hipFree(c->allocs.back()); c->allocs.pop_back(); }
  return rc;
}

static int test_gemm(pgpfa_ctx* c, bool transb, int M, int N, int K, double alpha, const double* A, const double* B, double beta, double* C) {
  if (!c || !A || !B || !C) return fail("null argument");
  if (K % 16 != 0) return fail("K must be a multiple of 16");
  HIPC(hipSetDevice(c->device));
  const int Mp = round_up(M, 128), Np = round_up(N, 128);
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  const size_t mark = c->allocs.size();
  CHK(dmalloc(c, &dA, (size_t)Mp * K, true));
  CHK(dmalloc(c, &dB, (size_t)Np * K, true));
  CHK(dmalloc(c, &dC, (size_t)Mp * N + 16, true));
  HIPC(hipMemcpy2DAsync(dA, (size_t)Mp * 8, A, (size_t)M * 8, (size_t)M * 8, K, hipMemcpyHostToDevice, c->st));
  if (!transb) HIPC(hipMemcpy2DAsync(dB, (size_t)Np * 8, B, (size_t)N * 8, (size_t)N * 8, K, hipMemcpyHostToDevice, c->st));
  else HIPC(hipMemcpyAsync(dB, B, (size_t)K * N * 8, hipMemcpyHostToDevice, c->st));       // K x N column-major, ldb = K
  HIPC(hipMemcpy2DAsync(dC, (size_t)Mp * 8, C, (size_t)M * 8, (size_t)M * 8, N, hipMemcpyHostToDevice, c->st));
  GemmP g{};
  g.A = dA; g.lda = Mp; g.B = dB; g.ldb = transb ? K : Np; g.C = dC; g.ldc = Mp; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.nbatch = 1; g.mode = GEMM_FULL;
  int rc = gemm(c, transb, g);
  if (!rc) {
    hipError_t e = hipMemcpy2DAsync(C, (size_t)M * 8, dC, (size_t)Mp * 8, (size_t)M * 8, N, hipMemcpyDeviceToHost, c->st);
    if (e != hipSuccess) rc = fail("copy back: %s", hipGetErrorString(e));
  }
  hipStreamSynchronize(c->st);
  while (c->allocs.size() > mark) { hipFree(c->allocs.back()); c->allocs.pop_back(); }
  return rc;
}

// The same products through the single-precision instantiation of the MFMA kernel (operands rounded to float on the way in).
static int test_gemm_f32(pgpfa_ctx* c, bool transb, int M, int N, int K, double alpha, const double* A, const double* B, double beta, double* C) {
  if (!c || !A || !B || !C) return fail("null argument");
  if (M < 1 || N < 1 || K < 1) return fail("invalid sizes");
  HIPC(hipSetDevice(c->device));
  const int Kp = round_up(K, 16);
  const size_t nA = (size_t)M * Kp + 256 * (size_t)Kp, nB = (size_t)N * Kp + 256 * (size_t)Kp, nC = (size_t)M * N;
  std::vector<float> hA(nA, 0.f), hB(nB, 0.f), hC(nC);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < M; ++i) hA[(size_t)k * M + i] = (float)A[(size_t)k * M + i];          // column-major M x K, lda = M
  if (transb) { for (int j = 0; j < N; ++j) for (int k = 0; k < K; ++k) hB[(size_t)j * Kp + k] = (float)B[(size_t)j * K + k]; }   // K x N, ldb = Kp
  else { for (int k = 0; k < K; ++k) for (int j = 0; j < N; ++j) hB[(size_t)k * N + j] = (float)B[(size_t)k * N + j]; }          // N x K, ldb = N
  for (size_t i = 0; i < nC; ++i) hC[i] = (float)C[i];
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  HIPC(hipMalloc((void**)&dA, (nA + 4096) * sizeof(float)));
  HIPC(hipMalloc((void**)&dB, (nB + 4096) * sizeof(float)));
  HIPC(hipMalloc((void**)&dC, (nC + 4096) * sizeof(float)));
  hipMemsetAsync(dA, 0, (nA + 4096) * sizeof(float), c->st); hipMemsetAsync(dB, 0, (nB + 4096) * sizeof(float), c->st);
  hipMemcpyAsync(dA, hA.data(), nA * sizeof(float), hipMemcpyHostToDevice, c->st);
  hipMemcpyAsync(dB, hB.data(), nB * sizeof(float), hipMemcpyHostToDevice, c->st);
  hipMemcpyAsync(dC, hC.data(), nC * sizeof(float), hipMemcpyHostToDevice, c->st);
  GemmP g{};
  g.A = reinterpret_cast<const double*>(dA); g.sA = 0; g.lda = M;
  g.B = reinterpret_cast<const double*>(dB); g.sB = 0; g.ldb = transb ? Kp : N;
  g.C = reinterpret_cast<double*>(dC); g.sC = 0; g.ldc = M;
  g.M = M; g.N = N; g.K = Kp; g.alpha = alpha; g.beta = beta; g.slots = nullptr; g.nbatch = 1; g.mode = GEMM_FULL; g.kflags = 0;
  int rc = gemm(c, transb, g, true);
  if (!rc) {
    hipMemcpyAsync(hC.data(), dC, nC * sizeof(float), hipMemcpyDeviceToHost, c->st);
    if (hipStreamSynchronize(c->st) != hipSuccess) rc = fail("f32 gemm failed");
    for (size_t i = 0; i < nC; ++i) C[i] = hC[i];
  }
  hipFree(dA); hipFree(dB); hipFree(dC);
  return rc;
}
int pgpfa_test_gemm_nt_f32(pgpfa_ctx* c, int M, int N, int K, double alpha, const double* A, const double* B, double beta, double* C) {
  return test_gemm_f32(c, false, M, N, K, alpha, A, B, beta, C);
}
int pgpfa_test_gemm_nn_f32(pgpfa_ctx* c, int M, int N, int K, double alpha, const double* A, const double* B, double beta, double* C) {
  return test_gemm_f32(c, true, M, N, K, alpha, A, B, beta, C);
}
int pgpfa_test_gemm_nt(pgpfa_ctx* c, int M, int N, int K, double alpha, const double* A, const double* B, double beta, double* C) {
  return test_gemm(c, false, M, N, K, alpha, A, B, beta, C);
}
int pgpfa_test_gemm_nn(pgpfa_ctx* c, int M, int N, int K, double alpha, const double* A, const double* B, double beta, double* C) {
  return test_gemm(c, true, M, N, K, alpha, A, B, beta, C);
}

int pgpfa_bench_mfma_peak(pgpfa_ctx* c, int iters, double* tflops) {
  if (!c || !tflops || iters < 1) return fail("invalid argument");
  HIPC(hipSetDevice(c->device));
  const int blocks = 256 * 8;                      // 8 waves per SIMD worth of blocks in flight
  double* out = nullptr;
  HIPC(hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, c->st, out, iters);
  hipEventRecord(e0, c->st);
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, c->st, out, iters);
  hipEventRecord(e1, c->st);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(out);
  HIPC(hipGetLastError());
  const double flops = (double)blocks * 4.0 * iters * 8.0 * 2048.0;   // 4 waves/block, 8 MFMAs/iter, 2*16*16*4 flops
  *tflops = flops / (ms * 1e-3) / 1e12;
  return 0;
}

int pgpfa_bench_syrk(pgpfa_ctx* c, int batch, int n, int k, int reps, double* ms_per_launch, double* flops_per_launch) {
  if (!c || !ms_per_launch || !flops_per_launch) return fail("null argument");
  if (n % 128 != 0 || k % 16 != 0 || batch < 1 || reps < 1) return fail("n must be a multiple of 128 and k of 16");
  HIPC(hipSetDevice(c->device));
  const size_t mark = c->allocs.size();
  double *dC = nullptr, *dA = nullptr;
  CHK(dmalloc(c, &dC, (size_t)n * n * batch));
  CHK(dmalloc(c, &dA, (size_t)n * k * batch));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((size_t)n * n * batch + 255) / 256)), dim3(256), 0, c->st, dC, (size_t)n * n * batch, 1.0);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((size_t)n * k * batch + 255) / 256)), dim3(256), 0, c->st, dA, (size_t)n * k * batch, 1e-3);
  GemmP s{};
  s.A = dA; s.sA = (long long)n * k; s.lda = n; s.B = dA; s.sB = s.sA; s.ldb = n;
  s.C = dC; s.sC = (long long)n * n; s.ldc = n; s.M = n; s.N = n; s.K = k; s.alpha = -1e-6; s.beta = 1.0;
  s.nbatch = batch; s.mode = GEMM_LOWER; s.kflags = KF_MASK_DIAG;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int rc = 0;
  for (int i = 0; i < 2 && !rc; ++i) rc = gemm_launch(c->st, c->mfma, false, s) != hipSuccess;
  hipEventRecord(e0, c->st);
  for (int i = 0; i < reps && !rc; ++i) rc = gemm_launch(c->st, c->mfma, false, s) != hipSuccess;
  hipEventRecord(e1, c->st);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  *ms_per_launch = ms / reps;
  *flops_per_launch = gemm_flops(s);
  while (c->allocs.size() > mark) { hipFree(c->allocs.back()); c->allocs.pop_back(); }
  if (rc) return fail("syrk bench launch failed");
  return 0;
}

// GEMM launches since the profile was switched on (option "profile" = 1 or 2), grouped by operand shape: one text line per shape, longest
// total first - launches, total ms, algorithmic GFLOP, TFLOP/s.  Returns the number of bytes the full report needs (incl. the terminator).
int pgpfa_gemm_shape_report(pgpfa_ctx* c, char* buf, int len) {
  if (!c || (len > 0 && !buf)) { fail("null argument"); return -1; }
  prof_collect(c);
  std::vector<std::pair<std::string, Prof::Shape>> v(c->prof.shapes.begin(), c->prof.shapes.end());
  std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.ms > b.second.ms; });
  std::string out;
  char line[256];
  for (const auto& kv : v) {
    std::snprintf(line, sizeof line, "%-74s n=%6.0f  %9.2f ms  %10.1f GFLOP  %6.1f TFLOP/s\n", kv.first.c_str(), kv.second.count, kv.second.ms,
                  kv.second.flops * 1e-9, kv.second.ms > 0.0 ? kv.second.flops / kv.second.ms * 1e-9 : 0.0);
    out += line;
  }
  if (len > 0) {
    const size_t n = std::min(out.size(), (size_t)len - 1);
    std::memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int)out.size() + 1;
}

// Phase timings of the diagonal-block kernel: `batch` well-conditioned 128 x 128 blocks, phases = 0 (load / store only), 1 (+ Cholesky
// steps), 3 (+ inverse: the production kernel).  us_per_launch = HIP-event time over `reps` launches.
int pgpfa_bench_potrf_diag(pgpfa_ctx* c, int batch, int reps, int phases, double* us_per_launch) {
  if (!c || !us_per_launch) return fail("null argument");
  if (batch < 1 || reps < 1 || (phases != 0 && phases != 1 && phases != 3 && phases != 5 && phases != 7)) return fail("batch, reps >= 1; phases 0, 1 or 3 (+ 4: the round-1 form of the Cholesky steps)");
  HIPC(hipSetDevice(c->device));
  const size_t mark = c->allocs.size();
  double *dH = nullptr, *dD = nullptr;
  int* dinfo = nullptr;
  const size_t blk = (size_t)NB * NB;
  CHK(dmalloc(c, &dH, blk * batch));
  CHK(dmalloc(c, &dD, blk * batch));
  CHK(dmalloc(c, &dinfo, (size_t)batch, true));
  std::vector<double> h(blk);
  for (int j = 0; j < NB; ++j)
    for (int i = 0; i < NB; ++i) h[(size_t)j * NB + i] = (i == j ? 2.0 : 0.0) + 1.0 / (1.0 + std::abs(i - j));
  for (int b = 0; b < batch; ++b) HIPC(hipMemcpyAsync(dH + blk * b, h.data(), blk * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPC(hipStreamSynchronize(c->st));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    // (the factor overwrites its input: later launches factor the factor's lower triangle - still SPD-like, diagonal > 1 - same work)
    if (phases == 7)
      hipLaunchKernelGGL((potrf_diag_kernel_t<double, 3, 1>), dim3(batch), dim3(512), 0, c->st, dH, (long long)blk, NB, 0, dD, (long long)blk, (const int*)nullptr, dinfo);
    else if (phases == 5)
      hipLaunchKernelGGL((potrf_diag_kernel_t<double, 1, 1>), dim3(batch), dim3(512), 0, c->st, dH, (long long)blk, NB, 0, dD, (long long)blk, (const int*)nullptr, dinfo);
    else if (phases == 3)
      hipLaunchKernelGGL((potrf_diag_kernel_t<double, 3>), dim3(batch), dim3(512), 0, c->st, dH, (long long)blk, NB, 0, dD, (long long)blk, (const int*)nullptr, dinfo);
    else if (phases == 1)
      hipLaunchKernelGGL((potrf_diag_kernel_t<double, 1>), dim3(batch), dim3(512), 0, c->st, dH, (long long)blk, NB, 0, dD, (long long)blk, (const int*)nullptr, dinfo);
    else
      hipLaunchKernelGGL((potrf_diag_kernel_t<double, 0>), dim3(batch), dim3(512), 0, c->st, dH, (long long)blk, NB, 0, dD, (long long)blk, (const int*)nullptr, dinfo);
  };
  launch();
  hipEventRecord(e0, c->st);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, c->st);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  *us_per_launch = 1e3 * ms / reps;
  while (c->allocs.size() > mark) { hipFree(c->allocs.back()); c->allocs.pop_back(); }
  HIPC(hipGetLastError());
  return 0;
}



