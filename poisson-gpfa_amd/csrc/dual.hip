// libpgpfa_hip.so - dual.hip (one translation unit of the C-ABI library; shared declarations: ctx.h)
#include "ctx.h"
#include "model.h"
#include "dual.h"

using namespace pgpfa;

// ---- dual variational E-step (inference.py:188-432) ----------------------------------------------------
int ensure_lambda(pgpfa_ctx* c) {
  if (c->lamd) return 0;
  // (slack: the GEMM form reads Lambda^T as a T x qpad operand and whole 128-row tiles)
  const size_t lam_slack = (size_t)16 * c->T + 4096;
  CHK(dmalloc(c, &c->lamd, (size_t)c->B * c->q * c->T + lam_slack, true));
  CHK(dmalloc(c, &c->dgrad, (size_t)c->B * c->q * c->T + lam_slack, true));
  CHK(dmalloc(c, &c->voff, (size_t)c->B * c->q * c->T + lam_slack, true));
  CHK(dmalloc(c, &c->dpart, (size_t)c->B * ((c->T + 63) / 64) * 2 + 16));
  CHK(dmalloc(c, &c->ldet_buf, (size_t)c->B * c->T + 16));
  c->dual_sscr = (long long)c->T * std::max(c->dual_npd, c->p * c->p);
  CHK(dmalloc(c, &c->dual_scr, (size_t)c->B * c->dual_sscr + (size_t)256 * c->T + 4096, true));
  return 0;
}

// lambda of the slots [0,nb) (already on device) -> v (into Xt), W, Kv (into KD); returns per-slot scalars
int dual_common(pgpfa_ctx* c, int nb, std::vector<double>* sB, std::vector<double>* sD, std::vector<double>* vKv) {
  const int q = c->q, p = c->p, T = c->T, ntile = (T + 63) / 64;
  const long long ld = c->ld;
  if (c->dual_gemm && c->mfma) {
    const long long sW = (long long)T * p * p;
    const int np = p * (p + 1) / 2;
    hipLaunchKernelGGL(dual_pre_kernel, dim3(ntile, nb), dim3(256), 0, c->st, c->Y, c->Yhi, c->d, c->lamd, c->dgrad, c->dpart, c->trial_of_slot, q, T);
    GemmP w{};                                               // Wp (T x pairs) = Lambda^T . TBL[:, pairs]
    w.A = c->lamd; w.sA = (long long)q * T; w.lda = T;
    w.B = c->dual_tbl; w.sB = 0; w.ldb = c->dual_ncol;
    w.C = c->dual_scr; w.sC = c->dual_sscr; w.ldc = T;
    w.M = T; w.N = np; w.K = c->qpad; w.alpha = 1.0; w.beta = 0.0; w.slots = c->ident; w.nbatch = nb; w.mode = GEMM_FULL; w.kflags = 0;
    CHK(gemm(c, false, w));
    GemmP v = w;                                             // V (T x p) = (Lambda - Y)^T . TBL[:, latents]   -> c->Xt
    v.A = c->dgrad; v.B = c->dual_tbl + c->dual_npd; v.C = c->Xt; v.sC = ld; v.N = p;
    CHK(gemm(c, false, v));
    const int nbu = dual_tile_bins((size_t)np);
    hipLaunchKernelGGL(dual_unpack_w_kernel, dim3((unsigned)((T + nbu - 1) / nbu), nb), dim3(256), (size_t)np * (nbu + 1) * sizeof(double), c->st, c->dual_scr, c->dual_sscr,
                       c->W, sW, T, p, nbu);
  } else {
    hipLaunchKernelGGL(dual_prep_kernel, dim3(ntile, nb), dim3(64), 0, c->st, c->Y, c->Yhi, c->C, c->d, c->lamd, (long long)q * T, c->Xt, ld, c->W,
                       (long long)T * p * p, c->dpart, ntile, c->ident, c->trial_of_slot, q, p, T);
  }
  CHK(prior_mv(c, c->ident, nb, c->Xt, c->KD, c->Kpad));            // K v
  hipLaunchKernelGGL(dots3_kernel, dim3(nb), dim3(256), 0, c->st, c->Xt, ld, c->KD, ld, (const double*)nullptr, 0LL, (const double*)nullptr, 0LL,
                     c->n, c->ident, c->sc_qxx, c->sc_qdx, c->sc_qdd);
  HIPC(hipGetLastError());
  std::vector<double> part((size_t)nb * ntile * 2);
  CHK(download(c, part.data(), c->dpart, part.size()));
  vKv->resize(nb);
  CHK(download(c, vKv->data(), c->sc_qxx, nb));
  sB->assign(nb, 0.0);
  sD->assign(nb, 0.0);
  for (int s = 0; s < nb; ++s)
    for (int b = 0; b < ntile; ++b) {
      (*sB)[s] += part[((size_t)s * ntile + b) * 2];
      (*sD)[s] += part[((size_t)s * ntile + b) * 2 + 1];
    }
  return 0;
}

int pgpfa_dual_costgrad(pgpfa_ctx* c, int trial, const double* lam, double* cost, double* grad) {
  if (c && c->have_counts && c->have_params && c->dual_lowrank && want_lowrank(c)) {
    // the low-rank engine is the batched evaluation with one trial
    const int32_t t = trial;
    return pgpfa_dual_costgrad_batch(c, 1, &t, lam, cost, grad);
  }
  CHK(ready(c));
  if (!lam || !cost) return fail("null argument");
  if (trial < 0 || trial >= c->R) return fail("trial %d out of range", trial);
  CHK(ensure_lambda(c));
  const int q = c->q, p = c->p, T = c->T;
  for (size_t i = 0; i < (size_t)q * T; ++i)
    if (!(lam[i] > 0.0)) return fail("lambda must be positive (entry %zu = %g)", i, lam[i]);
  std::vector<int> tr{trial};
  CHK(upload_list(c, c->trial_of_slot, tr));
  CHK(upload(c, c->lamd, lam, (size_t)q * T));
  std::vector<double> sB, sD, vKv;
  CHK(dual_common(c, 1, &sB, &sD, &vKv));
  HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int), c->st));
  CHK(assemble(c, c->ident, 1, 1.0 + 1e-6));                          // inference.py:190
  CHK(factor(c, c->ws, c->ident, 1));
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, c->st, c->ws.H, c->ld, c->npad, c->tscal + 8);
  double logdetH = 0.0;
  int info = 0;
  CHK(dl_enqueue(c, &logdetH, c->tscal + 8, sizeof(double)));
  CHK(dl_enqueue(c, &info, c->ws.info, sizeof(int)));
  CHK(dl_flush(c));
  if (info != 0) return fail("dual problem: posterior precision not positive definite (pivot %d)", info);
  // A + B + C + D of inference.py:203-213 ; C = 0.5*logdet(Sigma) = -0.5*logdet(precision + jitter)
  *cost = 0.5 * vKv[0] - sB[0] - 0.5 * logdetH + sD[0];
  if (grad) {
    CHK(ensure_mt_clean(c));
    CHK(inverse_t(c, c->ws, c->ident, 1));
    CHK(post_vsm_from_mt(c, 1));
    hipLaunchKernelGGL(dual_grad_kernel, dim3((T + 63) / 64, q), dim3(64), 0, c->st, c->C, c->d, c->lamd, c->KD,
                       c->vsm + (size_t)trial * T * p * p, c->dgrad, q, p, T);
    HIPC(hipGetLastError());
    CHK(download(c, grad, c->dgrad, (size_t)q * T));
  }
  return 0;
}

// VIPostMean (inference.py:193-194): -K_big C_big (lambda - y) for one trial, latent-major [p*T].
int pgpfa_dual_post_mean(pgpfa_ctx* c, int trial, const double* lam, double* mean) {
  CHK(ready(c));
  if (!lam || !mean) return fail("null argument");
  if (trial < 0 || trial >= c->R) return fail("trial %d out of range", trial);
  CHK(ensure_lambda(c));
  const int q = c->q, T = c->T;
  std::vector<int> tr{trial};
  CHK(upload_list(c, c->trial_of_slot, tr));
  CHK(upload(c, c->lamd, lam, (size_t)q * T));
  std::vector<double> sB, sD, vKv;
  CHK(dual_common(c, 1, &sB, &sD, &vKv));                              // KD <- K v,  v = C_big (lambda - y)
  CHK(download(c, mean, c->KD, (size_t)c->n));
  for (int i = 0; i < c->n; ++i) mean[i] = -mean[i];
  return 0;
}

// VIPostCov (inference.py:188-191): prec = K_big^-1 + C_big diag(lambda) C_big^T (dense, latent-major; may be NULL) and
// cov = (prec + 1e-6 diag(diag(prec)))^-1 for one trial.
// (lambda of the trial bound to slot 0 is already in c->lamd)
static int dual_post_cov_dev(pgpfa_ctx* c, double* cov, double* prec) {
  std::vector<double> sB, sD, vKv;
  CHK(dual_common(c, 1, &sB, &sD, &vKv));                              // W <- C^T diag(lambda_t) C
  if (prec) {
    hipLaunchKernelGGL(dense_h_kernel, dim3(c->n), dim3(256), 0, c->st, c->ws.H, c->n, c->T, c->Tp, c->p, c->Kinv, c->W);
    HIPC(hipGetLastError());
    CHK(download(c, prec, c->ws.H, (size_t)c->n * c->n));
  }
  HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int), c->st));
  CHK(ensure_mt_clean(c));
  CHK(assemble(c, c->ident, 1, 1.0 + 1e-6));
  CHK(factor(c, c->ws, c->ident, 1));
  CHK(inverse_t(c, c->ws, c->ident, 1));
  GemmP g{};
  g.A = c->ws.Mt; g.sA = c->ws.sM; g.lda = c->ld;
  g.B = c->ws.Mt; g.sB = c->ws.sM; g.ldb = c->ld;
  g.C = c->ws.H; g.sC = c->ws.sH; g.ldc = c->ld;
  g.M = c->npad; g.N = c->npad; g.K = c->npad; g.alpha = 1.0; g.beta = 0.0;
  g.slots = c->ident; g.nbatch = 1; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
  CHK(gemm(c, false, g));
  int info = 0;
  HIPC(hipMemcpyAsync(&info, c->ws.info, sizeof(int), hipMemcpyDeviceToHost, c->st));
  HIPC(hipMemcpy2DAsync(cov, (size_t)c->n * sizeof(double), c->ws.H, (size_t)c->ld * sizeof(double), (size_t)c->n * sizeof(double), c->n,
                        hipMemcpyDeviceToHost, c->st));
  HIPC(hipStreamSynchronize(c->st));
  if (info != 0) return fail("VIPostCov: posterior precision not positive definite (pivot %d)", info);
  return 0;
}

int pgpfa_dual_post_cov(pgpfa_ctx* c, int trial, const double* lam, double* cov, double* prec) {
  CHK(ready(c));
  if (!lam || !cov) return fail("null argument");
  if (trial < 0 || trial >= c->R) return fail("trial %d out of range", trial);
  CHK(ensure_lambda(c));
  const int q = c->q, T = c->T;
  for (size_t i = 0; i < (size_t)q * T; ++i)
    if (!std::isfinite(lam[i])) return fail("lambda entry %zu is not finite", i);
  std::vector<int> tr{trial};
  CHK(upload_list(c, c->trial_of_slot, tr));
  CHK(upload(c, c->lamd, lam, (size_t)q * T));
  return dual_post_cov_dev(c, cov, prec);
}

// post_cov of a trial whose resident posterior is the dual-variational one: VIPostCov at the lambda kept by pgpfa_dual_finalize
int post_cov_dual_impl(pgpfa_ctx* c, int trial, double* out) {
  CHK(ready(c));
  CHK(ensure_lambda(c));
  std::vector<int> tr{trial};
  CHK(upload_list(c, c->trial_of_slot, tr));
  const size_t m = (size_t)c->q * c->T;
  HIPC(hipMemcpyAsync(c->lamd, c->lam_keep + (size_t)trial * m, m * sizeof(double), hipMemcpyDeviceToDevice, c->st));
  return dual_post_cov_dev(c, out, nullptr);
}

// the reference's 1e-6 relative jitter on the diagonal of the posterior precision, applied to the W blocks of the slots [0, nb) in place
int dual_jitter(pgpfa_ctx* c, int nb) {
  hipLaunchKernelGGL(dual_jitter_kernel, dim3((unsigned)((c->T * c->p + 255) / 256), nb), dim3(256), 0, c->st, c->W, (long long)c->T * c->p * c->p,
                     c->Kinv, c->Tp, c->T, c->p, 1e-6);
  HIPC(hipGetLastError());
  return 0;
}

// dualProblem_grad (inference.py:218) of the slots [0, nb) into c->dgrad from K v (c->KD) and the per-bin covariance blocks in c->vsm
static int dual_gradient(pgpfa_ctx* c, int nb) {
  const int q = c->q, p = c->p, T = c->T;
  if (c->dual_gemm && c->mfma) {
    {
      int nbp = 32;
      while (nbp > 1 && (size_t)nbp * (p * p + 1) * sizeof(double) > 60000) nbp >>= 1;
      hipLaunchKernelGGL(dual_pack_sigma_kernel, dim3((unsigned)((T + nbp - 1) / nbp), nb), dim3(256), (size_t)nbp * (p * p + 1) * sizeof(double), c->st, c->vsm,
                         c->trial_of_slot, c->dual_scr, c->dual_sscr, T, p, c->dual_npd, nbp);
    }
    GemmP g{};                                               // G (T x q) = -1/2 Sp . TBL[:, pairs]^T
    g.A = c->dual_scr; g.sA = c->dual_sscr; g.lda = T;
    g.B = c->dual_tbl; g.sB = 0; g.ldb = c->dual_ncol;       // K x N column-major: element (pair, n) at n * ncol + pair
    g.C = c->dgrad; g.sC = (long long)q * T; g.ldc = T;
    g.M = T; g.N = q; g.K = c->dual_npd; g.alpha = -0.5; g.beta = 0.0; g.slots = c->ident; g.nbatch = nb; g.mode = GEMM_FULL; g.kflags = 0;
    CHK(gemm(c, true, g));
    GemmP l = g;                                             // G += K v (T x p) . TBL[:, latents]^T
    l.A = c->KD; l.sA = c->ld; l.B = c->dual_tbl + c->dual_npd; l.K = round_up(p, 16); l.alpha = 1.0; l.beta = 1.0;
    CHK(gemm(c, true, l));
    hipLaunchKernelGGL(dual_grad_finish_kernel, dim3((unsigned)(((size_t)q * T + 255) / 256), nb), dim3(256), 0, c->st, c->dgrad, c->lamd, c->d, q, T);
  } else {
    hipLaunchKernelGGL(dual_grad_batch_kernel, dim3((T + 63) / 64, q, nb), dim3(64), 0, c->st, c->C, c->d, c->lamd, c->KD, (long long)c->ld, c->vsm,
                       c->trial_of_slot, c->dgrad, q, p, T);
  }
  HIPC(hipGetLastError());
  return 0;
}

// Dual cost (and gradient with respect to lambda, into c->dgrad) of the slots [0, nb) whose lambda is already in c->lamd and
// whose trials are bound in c->trial_of_slot: the arithmetic of dualProblem / dualProblem_grad (inference.py:196-219) with
// the dense factorisations of the chunk batched.
// (tolerate: a slot whose precision is not positive definite or whose cost is not finite - a line-search trial point far
// out - gets cost = +inf instead of failing the call)
int dual_eval_slots(pgpfa_ctx* c, int nb, const std::vector<int>& tos, bool want_grad, double* cost, bool tolerate) {
  const int p = c->p, T = c->T;
  std::vector<double> sB, sD, vKv, logdet(nb);
  std::vector<int> info(nb);
  CHK(dual_common(c, nb, &sB, &sD, &vKv));
  if (c->plan_lowrank) {
    // low-rank engine: log det through the r x r system, Sigma_t blocks from the per-bin pass over Yt.  The reference's jitter
    // (diagonal of the precision scaled by 1 + 1e-6, inference.py:190) is a diagonal addition to the per-bin blocks W_t (dual.h):
    // with it the engine evaluates the reference's function - cost, log det and gradient follow inference.py:188-219.
    CHK(dual_jitter(c, nb));
    CHK(posterior_blocks_lowrank(c, nb, false, false, logdet.data()));
    CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
    CHK(dl_flush(c));
    for (int s2 = 0; s2 < nb; ++s2) {
      cost[s2] = 0.5 * vKv[s2] - sB[s2] - 0.5 * logdet[s2] + sD[s2];
      if (info[s2] != 0 || !std::isfinite(cost[s2])) {
        if (!tolerate) return fail("dual problem: posterior precision of trial %d not positive definite (pivot %d)", tos[s2], info[s2]);
        cost[s2] = std::numeric_limits<double>::infinity();
      }
    }
    if (want_grad) {
      CHK(dual_gradient(c, nb));
    }
    return 0;
  }
  HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int) * nb, c->st));
  CHK(ensure_mt_clean(c));
  CHK(assemble(c, c->ident, nb, 1.0 + 1e-6));                           // inference.py:190
  CHK(factor(c, c->ws, c->ident, nb));
  hipLaunchKernelGGL(logdet_batch_kernel, dim3(nb), dim3(256), 0, c->st, c->ws.H, (long long)c->ws.sH, c->ld, c->npad, c->sc_f);
  CHK(download(c, logdet.data(), c->sc_f, nb));
  CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
  CHK(dl_flush(c));
  for (int s2 = 0; s2 < nb; ++s2) {
    // A + B + C + D of inference.py:203-213 ; C = 0.5*logdet(Sigma) = -0.5*logdet(precision + jitter)
    cost[s2] = 0.5 * vKv[s2] - sB[s2] - 0.5 * logdet[s2] + sD[s2];
    if (info[s2] != 0 || !std::isfinite(cost[s2])) {
      if (!tolerate) return fail("dual problem: posterior precision of trial %d not positive definite (pivot %d)", tos[s2], info[s2]);
      cost[s2] = std::numeric_limits<double>::infinity();
    }
  }
  if (want_grad) {
    CHK(inverse_t(c, c->ws, c->ident, nb));
    CHK(post_vsm_from_mt(c, nb));
    CHK(dual_gradient(c, nb));
  }
  return 0;
}

// out[slot][n][t] = 1/2 c_n^T Sigma_t c_n of the slots [0, nb) from the per-bin covariance blocks in c->vsm (the variance term of the
// reference's dual gradient, inference.py:218)
int var_offsets(pgpfa_ctx* c, int nb, double* out) {
  const int q = c->q, p = c->p, T = c->T;
  if (c->dual_gemm && c->mfma && c->dual_tbl) {
    {
      int nbp = 32;
      while (nbp > 1 && (size_t)nbp * (p * p + 1) * sizeof(double) > 60000) nbp >>= 1;
      hipLaunchKernelGGL(dual_pack_sigma_kernel, dim3((unsigned)((T + nbp - 1) / nbp), nb), dim3(256), (size_t)nbp * (p * p + 1) * sizeof(double), c->st, c->vsm,
                         c->trial_of_slot, c->dual_scr, c->dual_sscr, T, p, c->dual_npd, nbp);
    }
    GemmP g{};                                               // (T x q) = 1/2 Sp . TBL[:, pairs]^T
    g.A = c->dual_scr; g.sA = c->dual_sscr; g.lda = T;
    g.B = c->dual_tbl; g.sB = 0; g.ldb = c->dual_ncol;
    g.C = out; g.sC = (long long)q * T; g.ldc = T;
    g.M = T; g.N = q; g.K = c->dual_npd; g.alpha = 0.5; g.beta = 0.0; g.slots = c->ident; g.nbatch = nb; g.mode = GEMM_FULL; g.kflags = 0;
    CHK(gemm(c, true, g));
  } else {
    hipLaunchKernelGGL(var_quad_kernel, dim3((T + 63) / 64, q, nb), dim3(64), 0, c->st, c->C, c->vsm, c->trial_of_slot, out, q, p, T);
  }
  HIPC(hipGetLastError());
  return 0;
}

// The optimum of the dual problem (inference.py:196-219) of a list of trials by a fixed point instead of a quasi-Newton run in lambda.
// At the optimum  log lambda = d + C m + v  with  m = -K C_big (lambda - y)  (VIPostMean) and  v = 1/2 diag(C Sigma C^T)  (VIPostCov, jitter
// included).  Given v, the first two say that m is the mode of the Laplace objective with the log rates shifted by v - found by the same
// batched Newton-PCG as the Laplace E-step, warm-started - and lambda = exp(C m + d + v); given lambda, v follows from the covariance
// blocks.  The map v -> v contracts by about half the largest posterior variance of a log rate (its Jacobian is
// -1/2 (C Sigma C^T)o(C Sigma C^T) diag(lambda) (I - C Sigma C^T diag(lambda)), rows sum to at most 1/2 c_n^T Sigma_t c_n), i.e. a
// digit or more per pass, where L-BFGS in rho needs thousands of evaluations (the dual's Hessian carries C K C^T: condition > 1e4).
// Stops per trial when max |v_new - v| <= tol: that IS the max-norm of the reference's dual gradient at the returned lambda.
// rho[n][q*T]: log lambda, start in (start = 1, 2) / optimum out; start: 0 cold (lambda = 0.5, rho not read), 1 rho is the start, 2 rho is a
// previous optimum (the mode search starts at its variational mean instead of zero); lam_out (may be NULL): the optimal lambda itself - it
// also stays on the device for pgpfa_dual_finalize(lam = NULL); fopt[n]: dual cost there; outer[n] (may be NULL): passes; vstatus[n]: 0 converged,
// 1 iteration cap, 2 not contracting (posterior variances too large for the plain fixed point: hand the trial to pgpfa_dual_lbfgs).
int pgpfa_dual_fixed_point(pgpfa_ctx* c, int n, const int32_t* idx, double* rho, int start, int max_outer, double tol, double* fopt, int32_t* outer,
                           int32_t* vstatus, double* lam_out) {
  if (!c) return fail("null context");
  if (!fopt || !vstatus) return fail("null argument");
  if (max_outer < 1 || !(tol > 0.0)) return fail("max_outer and tol must be positive");
  if (start < 0 || start > 3) return fail("start must be 0 (cold), 1 (rho is the start), 2 (rho is a previous optimum) or 3 (the resident optimum)");
  if (!rho && (start == 1 || start == 2)) return fail("start = %d reads rho", start);
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  CHK(check_distinct(tr.v));
  HIPC(hipSetDevice(c->device));
  const int N = (int)tr.v.size();
  if (start == 3) {
    if (!c->lam_keep) return fail("start = 3 needs a resident dual optimum (pgpfa_dual_fixed_point or pgpfa_dual_finalize)");
    for (int t : tr.v)
      if (!c->lam_valid[t]) return fail("trial %d has no resident dual optimum (start = 3)", t);
  }
  VarJob job{rho, max_outer, tol, fopt, outer, vstatus, start, lam_out};
  std::vector<int32_t> it1(N), st1(N);
  double obj = 0.0;
  CHK(estep_impl(c, tr, 0, c->dual_lowrank, &obj, it1.data(), st1.data(), nullptr, &job));
  double ev = 0.0;
  for (int i = 0; i < N; ++i) ev += (outer ? outer[i] : 0) + 1.0;
  c->info["last_dual_evaluations"] = ev;          // covariance passes (one per outer pass + the start), the unit the L-BFGS driver counts too
  return 0;
}

int check_distinct(const std::vector<int>& v) {
  std::vector<int> s(v);
  std::sort(s.begin(), s.end());
  for (size_t i = 1; i < s.size(); ++i)
    if (s[i] == s[i - 1]) return fail("trial %d listed twice (the per-trial covariance blocks are scratch space of this call)", s[i]);
  return 0;
}

// dualProblem / dualProblem_grad for a LIST of trials at once (each trial at its own lambda).  The per-trial scipy
// L-BFGS-B runs of inference.dualVariational (DUAL_SOLVER = 'scipy') are driven concurrently so that one round of their
// requests is one call of this.
int pgpfa_dual_costgrad_batch(pgpfa_ctx* c, int n, const int32_t* idx, const double* lam, double* cost, double* grad) {
  PhaseRange range_phase("pgpfa.dual_costgrad_batch");
  if (!c) return fail("null context");
  if (!lam || !cost) return fail("null argument");
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  const int N = (int)tr.v.size();
  c->want_slots = std::max(c->want_slots, std::min(N, c->R));
  CHK(ready_estep(c, c->dual_lowrank));
  CHK(ensure_lambda(c));
  const size_t m = (size_t)c->q * c->T;
  for (size_t i = 0; i < (size_t)N * m; ++i)
    if (!(lam[i] > 0.0)) return fail("lambda must be positive (trial %d, entry %zu = %g)", tr.v[i / m], i % m, lam[i]);
  CHK(check_distinct(tr.v));
  for (int c0 = 0; c0 < N; c0 += c->B) {
    const int nb = std::min(c->B, N - c0);
    std::vector<int> tos(tr.v.begin() + c0, tr.v.begin() + c0 + nb);
    CHK(upload_list(c, c->trial_of_slot, tos));
    CHK(upload(c, c->lamd, lam + (size_t)c0 * m, (size_t)nb * m));
    CHK(dual_eval_slots(c, nb, tos, grad != nullptr, cost + c0));
    if (grad) CHK(download(c, grad + (size_t)c0 * m, c->dgrad, (size_t)nb * m));
  }
  return 0;
}

// The whole dual optimisation of a list of trials on the device: one L-BFGS run per trial in rho = log(lambda) (the
// unconstrained form of the reference's optimizeLogLambda=True path, inference.py:222-256, 391-396), all runs of a chunk in
// lockstep - every iteration is one batched dual evaluation plus per-slot two-loop recursions on device-resident vectors.
// Backtracking (Armijo) line search; stops per trial on scipy's L-BFGS-B criteria: relative decrease <= factr * eps or
// max |gradient| <= pgtol.  rho[n][q*T]: start in, optimum out; fopt[n]: dual optimum; iters[n] (may be NULL).
int pgpfa_dual_lbfgs(pgpfa_ctx* c, int n, const int32_t* idx, double* rho, int max_iter, double factr, double pgtol, double* fopt,
                     int32_t* iters) {
  PhaseRange range_phase("pgpfa.dual_lbfgs");
  if (!c) return fail("null context");
  if (!rho || !fopt) return fail("null argument");
  if (max_iter < 1) return fail("max_iter must be positive");
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  CHK(check_distinct(tr.v));
  const int N = (int)tr.v.size();
  c->want_slots = std::max(c->want_slots, std::min(N, c->R));
  CHK(ready_estep(c, c->dual_lowrank));
  CHK(ensure_lambda(c));
  const size_t m = (size_t)c->q * c->T;
  constexpr int HIST = 10;                                          // scipy's default m = 10 corrections
  const int Bc = std::min(c->B, N);
  // device vectors of this call (freed on return): X, G, D, Xn, Gn and the correction pairs
  std::vector<double*> owned;
  auto dalloc = [&](double** ptr, size_t count) -> int {
    if (hipMalloc((void**)ptr, count * sizeof(double)) != hipSuccess) return fail("out of device memory for the L-BFGS state (%zu bytes)", count * sizeof(double));
    owned.push_back(*ptr);
    return 0;
  };
  struct Freer { std::vector<double*>& v; ~Freer() { for (double* q2 : v) hipFree(q2); } } freer{owned};
  double *X, *G, *D, *Xn, *Gn, *S, *Yh, *scal;
  const size_t vec = (size_t)Bc * m;
  CHK(dalloc(&X, vec)); CHK(dalloc(&G, vec)); CHK(dalloc(&D, vec)); CHK(dalloc(&Xn, vec)); CHK(dalloc(&Gn, vec));
  CHK(dalloc(&S, vec * HIST)); CHK(dalloc(&Yh, vec * HIST)); CHK(dalloc(&scal, 4 * (size_t)Bc));
  int* take = nullptr;
  HIPC(hipMalloc((void**)&take, sizeof(int) * Bc));
  struct FreeI { int* q2; ~FreeI() { hipFree(q2); } } freei{take};
  const dim3 vgrid((unsigned)((m + 255) / 256), 1);
  const double eps = 2.220446049250313e-16;
  double n_eval = 0.0;                                              // batched dual evaluations of this call

  for (int c0 = 0; c0 < N; c0 += Bc) {
    const int nb0 = std::min(Bc, N - c0);
    int nb = nb0;                                                     // live slots: finished trials are retired (compaction below)
    std::vector<int> tos(tr.v.begin() + c0, tr.v.begin() + c0 + nb);
    std::vector<int> orig(nb);                                        // slot -> position in this chunk's trial list
    for (int s2 = 0; s2 < nb; ++s2) orig[s2] = s2;
    CHK(upload_list(c, c->trial_of_slot, tos));
    dim3 grid(vgrid.x, nb);
    auto bdot = [&](const double* A, const double* B2, std::vector<double>& out) -> int {
      hipLaunchKernelGGL(bdot_kernel, dim3(nb), dim3(256), 0, c->st, A, B2, m, scal);
      return download(c, out.data(), scal, nb);
    };
    auto upload_scal = [&](const std::vector<double>& v, int slot) -> int { return upload(c, scal + (size_t)slot * Bc, v.data(), nb); };
    // f and the gradient with respect to rho at the device vector Xin (lambda = exp(rho) goes to c->lamd)
    auto evaluate = [&](const double* Xin, double* Gout, std::vector<double>& f) -> int {
      n_eval += 1.0;
      hipLaunchKernelGGL(exp_kernel, dim3((unsigned)((nb * m + 255) / 256)), dim3(256), 0, c->st, Xin, c->lamd, nb * m);
      CHK(dual_eval_slots(c, nb, tos, true, f.data(), /*tolerate=*/true));
      hipLaunchKernelGGL(chain_kernel, dim3((unsigned)((nb * m + 255) / 256)), dim3(256), 0, c->st, c->dgrad, c->lamd, Gout, nb * m);
      HIPC(hipGetLastError());
      return 0;
    };
    CHK(upload(c, X, rho + (size_t)c0 * m, (size_t)nb * m));
    std::vector<double> f(nb), fn(nb), gd(nb), t(nb), tmp(nb), gmax(nb);
    std::vector<std::vector<double>> rho_h(HIST, std::vector<double>(nb, 0.0)), alpha(HIST, std::vector<double>(nb, 0.0));
    std::vector<int> nhist(nb, 0), head(nb, 0), done(nb, 0), its(nb, 0), flags(nb);
    CHK(evaluate(X, G, f));
    hipLaunchKernelGGL(bmaxabs_kernel, dim3(nb), dim3(256), 0, c->st, G, m, scal);
    CHK(download(c, gmax.data(), scal, nb));
    for (int s2 = 0; s2 < nb; ++s2) done[s2] = (gmax[s2] <= pgtol) ? 1 : 0;
    int global_hist = 0;                                              // pairs are pushed in lockstep; per-slot validity via rho_h > 0

    for (int it = 0; it < max_iter; ++it) {
      bool any = false;
      for (int s2 = 0; s2 < nb; ++s2) any = any || !done[s2];
      if (!any) break;
      // ---- direction D = -H G (two-loop recursion over the stored pairs; invalid pairs have rho_h = 0: no-ops)
      HIPC(hipMemcpyAsync(D, G, (size_t)nb * m * sizeof(double), hipMemcpyDeviceToDevice, c->st));
      const int used = std::min(global_hist, HIST);
      for (int j = 0; j < used; ++j) {                              // newest -> oldest
        const int i = (global_hist - 1 - j) % HIST;
        CHK(bdot(S + (size_t)i * vec, D, tmp));
        for (int s2 = 0; s2 < nb; ++s2) { alpha[i][s2] = rho_h[i][s2] * tmp[s2]; tmp[s2] = -alpha[i][s2]; }
        CHK(upload_scal(tmp, 0));
        hipLaunchKernelGGL(baxpby_kernel, grid, dim3(256), 0, c->st, scal, Yh + (size_t)i * vec, (const double*)nullptr, D, m);
      }
      if (used > 0) {                                               // initial scaling gamma = s.y / y.y of the newest pair
        const int i = (global_hist - 1) % HIST;
        CHK(bdot(Yh + (size_t)i * vec, Yh + (size_t)i * vec, tmp));
        std::vector<double> gam(nb), zero(nb, 0.0);
        for (int s2 = 0; s2 < nb; ++s2) gam[s2] = (rho_h[i][s2] > 0.0 && tmp[s2] > 0.0) ? 1.0 / (rho_h[i][s2] * tmp[s2]) : 1.0;
        CHK(upload_scal(gam, 1));
        CHK(upload_scal(zero, 0));
        hipLaunchKernelGGL(baxpby_kernel, grid, dim3(256), 0, c->st, scal, D, scal + Bc, D, m);       // D <- gamma D
      }
      for (int j = used - 1; j >= 0; --j) {                         // oldest -> newest
        const int i = (global_hist - 1 - j) % HIST;
        CHK(bdot(Yh + (size_t)i * vec, D, tmp));
        for (int s2 = 0; s2 < nb; ++s2) tmp[s2] = alpha[i][s2] - rho_h[i][s2] * tmp[s2];
        CHK(upload_scal(tmp, 0));
        hipLaunchKernelGGL(baxpby_kernel, grid, dim3(256), 0, c->st, scal, S + (size_t)i * vec, (const double*)nullptr, D, m);
      }
      {                                                             // D <- -D
        std::vector<double> zero(nb, 0.0), neg(nb, -1.0);
        CHK(upload_scal(zero, 0));
        CHK(upload_scal(neg, 1));
        hipLaunchKernelGGL(baxpby_kernel, grid, dim3(256), 0, c->st, scal, D, scal + Bc, D, m);
      }
      CHK(bdot(G, D, gd));
      bool reset = false;
      for (int s2 = 0; s2 < nb; ++s2)
        if (!done[s2] && !(gd[s2] < 0.0)) reset = true;             // not a descent direction (stale pairs): restart from steepest descent
      if (reset) {
        global_hist = 0;
        for (auto& r : rho_h) std::fill(r.begin(), r.end(), 0.0);
        std::vector<double> zero(nb, 0.0), neg(nb, -1.0);
        CHK(upload_scal(neg, 0));
        CHK(upload_scal(zero, 1));
        hipLaunchKernelGGL(baxpby_kernel, grid, dim3(256), 0, c->st, scal, G, scal + Bc, D, m);        // D <- -G
        CHK(bdot(G, D, gd));
      }
      // ---- backtracking line search, all slots in lockstep (finished slots take t = 0)
      hipLaunchKernelGGL(bmaxabs_kernel, dim3(nb), dim3(256), 0, c->st, D, m, scal + 2 * (size_t)Bc);
      CHK(download(c, tmp.data(), scal + 2 * (size_t)Bc, nb));
      std::vector<int> pending;
      for (int s2 = 0; s2 < nb; ++s2) {
        t[s2] = 0.0;
        if (done[s2]) continue;
        t[s2] = (global_hist == 0 && tmp[s2] > 1.0) ? 1.0 / tmp[s2] : 1.0;      // first step: at most unit length in the max norm
        pending.push_back(s2);
      }
      std::vector<int> accepted(nb, 0);
      for (int ls = 0; ls < 30 && !pending.empty(); ++ls) {
        CHK(upload_scal(t, 0));
        hipLaunchKernelGGL(bstep_kernel, grid, dim3(256), 0, c->st, X, D, scal, Xn, m);
        CHK(evaluate(Xn, Gn, fn));
        std::vector<int> rej;
        for (int s2 : pending) {
          if (std::isfinite(fn[s2]) && fn[s2] <= f[s2] + 1e-4 * t[s2] * gd[s2] + 1e-14 * (1.0 + std::fabs(f[s2]))) accepted[s2] = 1;
          else { t[s2] *= 0.5; rej.push_back(s2); }
        }
        if (rej.empty()) break;
        // slots accepted in this round keep their point: freeze it by re-deriving the same Xn next round (t unchanged)
        pending.swap(rej);
      }
      for (int s2 : pending)
        if (!accepted[s2]) { t[s2] = 0.0; done[s2] = 1; }            // search exhausted: stay (cannot improve at this precision)
      if (!pending.empty() && std::any_of(pending.begin(), pending.end(), [&](int s2) { return !accepted[s2]; })) {
        CHK(upload_scal(t, 0));
        hipLaunchKernelGGL(bstep_kernel, grid, dim3(256), 0, c->st, X, D, scal, Xn, m);
        CHK(evaluate(Xn, Gn, fn));
      }
      // ---- new correction pair s = Xn - X, y = Gn - G (slots that moved), convergence tests, commit
      const int i_new = global_hist % HIST;
      for (int s2 = 0; s2 < nb; ++s2) flags[s2] = (accepted[s2] && t[s2] > 0.0) ? 1 : 0;
      HIPC(hipMemcpyAsync(take, flags.data(), sizeof(int) * nb, hipMemcpyHostToDevice, c->st));
      HIPC(hipMemsetAsync(S + (size_t)i_new * vec, 0, (size_t)nb * m * sizeof(double), c->st));
      HIPC(hipMemsetAsync(Yh + (size_t)i_new * vec, 0, (size_t)nb * m * sizeof(double), c->st));
      hipLaunchKernelGGL(bdiff_kernel, grid, dim3(256), 0, c->st, Xn, X, take, S + (size_t)i_new * vec, m);
      hipLaunchKernelGGL(bdiff_kernel, grid, dim3(256), 0, c->st, Gn, G, take, Yh + (size_t)i_new * vec, m);
      CHK(bdot(S + (size_t)i_new * vec, Yh + (size_t)i_new * vec, tmp));
      for (int s2 = 0; s2 < nb; ++s2) rho_h[i_new][s2] = (flags[s2] && tmp[s2] > 1e-300) ? 1.0 / tmp[s2] : 0.0;
      global_hist += 1;
      hipLaunchKernelGGL(bcopy_kernel, grid, dim3(256), 0, c->st, Xn, take, X, m);
      hipLaunchKernelGGL(bcopy_kernel, grid, dim3(256), 0, c->st, Gn, take, G, m);
      hipLaunchKernelGGL(bmaxabs_kernel, dim3(nb), dim3(256), 0, c->st, G, m, scal);
      CHK(download(c, gmax.data(), scal, nb));
      for (int s2 = 0; s2 < nb; ++s2) {
        if (done[s2] || !flags[s2]) continue;
        its[s2] = it + 1;
        const double dec = f[s2] - fn[s2];
        const double den = std::max(std::max(std::fabs(f[s2]), std::fabs(fn[s2])), 1.0);
        f[s2] = fn[s2];
        if (dec / den <= factr * eps || gmax[s2] <= pgtol) done[s2] = 1;
      }
      // ---- retire finished trials: once an eighth of the live slots are done, their results leave and the last live
      // slots move into the holes (X, G and the stored pairs), so that every later evaluation only pays for trials
      // that are still being optimised (the slowest trial takes several times the iterations of the median one)
      int ndone = 0;
      for (int s2 = 0; s2 < nb; ++s2) ndone += done[s2] ? 1 : 0;
      if (ndone > 0 && ndone < nb && ndone >= std::max(1, nb / 8)) {
        auto retire = [&](int s2) -> int {
          CHK(download(c, rho + (size_t)(c0 + orig[s2]) * m, X + (size_t)s2 * m, m));
          fopt[c0 + orig[s2]] = f[s2];
          if (iters) iters[c0 + orig[s2]] = its[s2];
          return 0;
        };
        int last = nb - 1;
        for (int s2 = 0; s2 <= last; ++s2) {
          if (!done[s2]) continue;
          CHK(retire(s2));
          while (last > s2 && done[last]) { CHK(retire(last)); --last; }
          if (last > s2) {                                            // move live slot `last` into position s2
            const size_t bytes = m * sizeof(double);
            HIPC(hipMemcpyAsync(X + (size_t)s2 * m, X + (size_t)last * m, bytes, hipMemcpyDeviceToDevice, c->st));
            HIPC(hipMemcpyAsync(G + (size_t)s2 * m, G + (size_t)last * m, bytes, hipMemcpyDeviceToDevice, c->st));
            for (int h = 0; h < HIST; ++h) {
              HIPC(hipMemcpyAsync(S + (size_t)h * vec + (size_t)s2 * m, S + (size_t)h * vec + (size_t)last * m, bytes, hipMemcpyDeviceToDevice, c->st));
              HIPC(hipMemcpyAsync(Yh + (size_t)h * vec + (size_t)s2 * m, Yh + (size_t)h * vec + (size_t)last * m, bytes, hipMemcpyDeviceToDevice, c->st));
              rho_h[h][s2] = rho_h[h][last];
            }
            f[s2] = f[last]; its[s2] = its[last]; done[s2] = 0; orig[s2] = orig[last]; tos[s2] = tos[last];
          }
          --last;
        }
        nb = last + 1;
        tos.resize(nb);
        CHK(upload_list(c, c->trial_of_slot, tos));
        grid = dim3(vgrid.x, nb);
      }
    }
    for (int s2 = 0; s2 < nb; ++s2) {
      CHK(download(c, rho + (size_t)(c0 + orig[s2]) * m, X + (size_t)s2 * m, m));
      fopt[c0 + orig[s2]] = f[s2];
      if (iters) iters[c0 + orig[s2]] = its[s2];
    }
  }
  c->info["last_dual_evaluations"] = n_eval;
  return 0;
}

// the dual variables resident for the listed trials (the optimum of the last pgpfa_dual_fixed_point, or what pgpfa_dual_finalize was given)
int pgpfa_get_dual_lambda(pgpfa_ctx* c, int n, const int32_t* idx, double* out) {
  if (!c || !out) return fail("null argument");
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr));
  HIPC(hipSetDevice(c->device));
  if (!c->lam_keep) return fail("no dual variables are resident");
  const size_t m = (size_t)c->q * c->T;
  for (int t : tr.v)
    if (!c->lam_valid[t]) return fail("trial %d has no resident dual variables", t);
  for (size_t i = 0; i < tr.v.size(); ++i) {
    CHK(dl_enqueue(c, out + i * m, c->lam_keep + (size_t)tr.v[i] * m, m * sizeof(double)));
  }
  return dl_flush(c);
}

int pgpfa_dual_finalize(pgpfa_ctx* c, int n, const int32_t* idx, const double* lam, double* nlp_sum) {
  PhaseRange range_phase("pgpfa.dual_finalize");
  if (!c) return fail("null context");
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  const int N = (int)tr.v.size();
  if (!lam) {
    // the optimum the last pgpfa_dual_fixed_point left on the device for these trials
    if (!c->lam_keep) return fail("lam = NULL needs the resident optimum of pgpfa_dual_fixed_point");
    for (int t : tr.v)
      if (!c->lam_resident[t]) return fail("trial %d has no resident optimum of pgpfa_dual_fixed_point (lam = NULL)", t);
  }
  c->want_slots = std::max(c->want_slots, std::min(N, c->R));
  CHK(ready_estep(c, c->dual_lowrank));
  CHK(ensure_lambda(c));
  // under the low-rank plan the reference's 1e-6 diagonal jitter enters through the per-bin blocks (dual_jitter), and - as in the
  // Laplace E-step - only the sum over trials of post_vsmGP is accumulated unless keep_trial_vsmgp is set
  const bool sum_only = c->plan_lowrank && !c->keep_trial_vsmgp;
  c->pacc_used = false; c->pacc_valid = false;
  c->info["last_eps_wt_norm"] = 0.0; c->info["last_eps_wt_rms"] = 0.0;      // maxima over the chunks of THIS call
  HIPC(hipMemsetAsync(c->Pacc, 0, (size_t)c->Tp * c->Tp * c->p * sizeof(double), c->st));
  snapshot_params(c, tr.v);
  if (!c->lam_keep) {
    const size_t bytes = (size_t)c->R * c->q * c->T * sizeof(double);
    if (hipMalloc((void**)&c->lam_keep, bytes) != hipSuccess) { (void)hipGetLastError(); c->lam_keep = nullptr; return fail("hipMalloc(%zu bytes) for the resident dual variables failed", bytes); }
    c->bytes += bytes;
  }
  const int q = c->q;
  const long long ld = c->ld;
  double total = 0.0;
  std::vector<double> f(c->B), qxx(c->B);
  std::vector<int> info(c->B);
  for (int c0 = 0; c0 < N; c0 += c->B) {
    const int nb = std::min(c->B, N - c0);
    std::vector<int> tos(tr.v.begin() + c0, tr.v.begin() + c0 + nb);
    CHK(upload_list(c, c->trial_of_slot, tos));
    if (lam) CHK(upload(c, c->lamd, lam + (size_t)c0 * q * c->T, (size_t)nb * q * c->T));
    for (int s = 0; s < nb; ++s) {
      const size_t mq = (size_t)q * c->T;
      if (lam) { CHK(copy_dev(c, c->lam_keep + (size_t)tos[s] * mq, c->lamd + (size_t)s * mq, mq * sizeof(double))); c->lam_resident[tos[s]] = 0; c->lam_valid[tos[s]] = 1; }
      else CHK(copy_dev(c, c->lamd + (size_t)s * mq, c->lam_keep + (size_t)tos[s] * mq, mq * sizeof(double)));
      c->trial_dual[tos[s]] = 1;
    }
    HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int) * nb, c->st));
    std::vector<double> sB, sD, vKv;
    CHK(dual_common(c, nb, &sB, &sD, &vKv));
    // posterior mean -K C_big (lambda - y) (inference.py:194) and covariance blocks (inference.py:188-191)
    hipLaunchKernelGGL(negate_rows_kernel, dim3((c->n + 255) / 256, nb), dim3(256), 0, c->st, c->KD, ld, c->Xc, ld, c->n, c->ident);
    if (c->plan_lowrank) { CHK(dual_jitter(c, nb)); CHK(posterior_blocks(c, nb, 1.0, true, sum_only)); }
    else CHK(posterior_blocks(c, nb, 1.0 + 1e-6, true));
    for (int t : tos) c->vsmgp_ok[t] = sum_only ? 0 : 1;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((c->n + 255) / 256, nb), dim3(256), 0, c->st, c->Xc, ld, c->n, c->Xmode, c->trial_of_slot);
    for (int t_ : tos) c->mode_serial[t_] = -10;
    // negLogPosteriorUnNorm at the VI mean (inference.py:333)
    CHK(prior_mv(c, c->ident, nb, c->Xc, c->KX));
    hipLaunchKernelGGL(dots3_kernel, dim3(nb), dim3(256), 0, c->st, c->Xc, ld, c->KX, ld, (const double*)nullptr, 0LL, (const double*)nullptr, 0LL,
                       c->n, c->ident, c->sc_qxx, c->sc_qdx, c->sc_qdd);
    CHK(poisson(c, c->ident, nb, c->Xc, c->Gl, c->Wt, c->sc_f, 0));
    CHK(download(c, f.data(), c->sc_f, nb));
    CHK(download(c, qxx.data(), c->sc_qxx, nb));
    CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
    CHK(dl_flush(c));
    for (int s = 0; s < nb; ++s) {
      if (info[s] != 0) return fail("dual finalize: posterior precision of trial %d not positive definite", tos[s]);
      total += f[s] + 0.5 * qxx[s];
    }
  }
  CHK(remember_trials(c, tr.v));
  c->pacc_valid = c->pacc_used;
  if (nlp_sum) *nlp_sum = total;
  return 0;
}


