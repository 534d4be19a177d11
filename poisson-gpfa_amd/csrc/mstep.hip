// libpgpfa_hip.so - mstep.hip (one translation unit of the C-ABI library; shared declarations: ctx.h)
#include "ctx.h"
#include "model.h"
#include "mstep.h"

using namespace pgpfa;

// ---- M-step ------------------------------------------------------------------------------------------
// One (C,d) cost / gradient sweep over the trials of the last E-step at the parameters in c->vec: c->cdout <- per-neuron sums
// [(p+2)][q] (rows 0..p-1: sum (y - yhat) m - yhat V c, row p: sum (y - yhat), row p+1: sum (y hh - yhat)).  Matrix-core kernel
// up to 20 latents (mstep.h), the vector kernel beyond that or with option cd_mfma = 0.
// count terms of the (C,d) cost, linear in (c_n, d_n): c->cdym[(p+1)][q] = sum_t y m_t | sum_t y over the trials of the last E-step
static int ensure_cdym(pgpfa_ctx* c) {
  if (c->cdym_valid) return 0;
  const int q = c->q, p = c->p, ntr = (int)c->last_trials_h.size();
  const int nbk = std::max(1, std::min(1024, ntr));
  if (c->mfma && c->cd_mfma && p + 1 <= 16)
    hipLaunchKernelGGL(cd_ym_mfma_kernel, dim3(nbk), dim3(256), 0, c->st, c->Y, c->Yhi, c->Xmode, c->last_trials, ntr, q, p, c->T, c->cdym_part);
  else
    hipLaunchKernelGGL(cd_ym_kernel, dim3(nbk), dim3(256), 0, c->st, c->Y, c->Yhi, c->Xmode, c->last_trials, ntr, q, p, c->T, c->cdym_part);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(((p + 1) * q + 31) / 32), dim3(256), 0, c->st, c->cdym_part, nbk, (p + 1) * q, c->cdym);
  HIPC(hipGetLastError());
  c->cdym_valid = true;
  return 0;
}

static int cd_sweep(pgpfa_ctx* c) {
  const int q = c->q, p = c->p, T = c->T;
  const int len = (p + 2) * q;
  CdArgs a{};
  a.Y = c->Y; a.Yhi = c->Yhi; a.mean = c->Xmode; a.vsm = c->vsm; a.vec = c->vec;
  a.trials = c->last_trials; a.ntr = (int)c->last_trials_h.size();
  a.part = c->cdpart; a.q = q; a.p = p; a.T = T; a.dbg = c->cd_debug;
  const double flops = (double)a.ntr * q * T * (2.0 * p * p + 8.0 * p);
  if (c->mfma && c->cd_mfma && p <= 10) {
    CHK(ensure_cdym(c));
    int nby = 1;
    prof_begin(c, TAG_CD, flops);
    dispatch_pw(p, [&](auto pw) {
      constexpr int PW = decltype(pw)::value;
      if constexpr (PW <= 10) {
        const int ntt = (T + CdM<PW>::BT - 1) / CdM<PW>::BT;
        const int tiles = (q + 15) / 16, groups = (tiles + 7) / 8, tpg = (tiles + groups - 1) / groups;
        nby = std::max(1, std::min(a.ntr * ntt, std::max(64, 512 / groups)));      // one resident workgroup per CU: about one round of blocks
        const int waves = 8;
        hipLaunchKernelGGL(mstep_cd_mfma_kernel<PW>, dim3(nby, groups), dim3(64, waves), cd_mfma_lds_bytes<PW>(), c->st, a, tpg);
      }
    });
    prof_end(c);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 31) / 32), dim3(256), 0, c->st, c->cdpart, nby, len, c->cdout);
    hipLaunchKernelGGL(cd_add_ym_kernel, dim3((q + 127) / 128), dim3(128), 0, c->st, c->cdout, c->cdym, c->vec, q, p);
    HIPC(hipGetLastError());
    return 0;
  }
  const int nchunk = (q + 63) / 64;
  const int nby = std::max(1, std::min(a.ntr * 4, std::max(64, 1024 / nchunk)));
  prof_begin(c, TAG_CD, flops);
  dispatch_pw(p, [&](auto pw) {
    hipLaunchKernelGGL(mstep_cd_kernel<decltype(pw)::value>, dim3((q + 63) / 64, nby), dim3(64, CdKy<decltype(pw)::value>::v), 0, c->st, a);
  });
  prof_end(c);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 31) / 32), dim3(256), 0, c->st, c->cdpart, nby, len, c->cdout);
  HIPC(hipGetLastError());
  return 0;
}

int pgpfa_mstep_cd_costgrad(pgpfa_ctx* c, const double* vecCd, const double* prior_center, double inv_s2, double* cost, double* grad) {
  PhaseRange range_phase("pgpfa.mstep_cd_costgrad");
  if (!c || !vecCd || !cost || !grad) return fail("null argument");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  if (!c->have_post) return fail("no E-step result resident: run an E-step or pgpfa_set_posterior first");
  HIPC(hipSetDevice(c->device));
  const int q = c->q, p = c->p;
  const int len = (p + 2) * q;
  CHK(upload_nosync(c, c->vec, vecCd, (size_t)q * (p + 1) * sizeof(double)));
  CHK(cd_sweep(c));
  const int ntr_local = (int)c->last_trials_h.size();
  // append the local trial count, all-reduce [sums | count] over ranks
  const double cnt = (double)ntr_local;
  CHK(upload_nosync(c, c->cdout + len, &cnt, sizeof(double)));
  CHK(allreduce_dev(c, c->cdout, (size_t)len + 1));
  CHK(ensure_hbuf(c, (size_t)len + 1));
  CHK(download(c, c->hbuf, c->cdout, (size_t)len + 1));
  const double Rtot = c->hbuf[len];
  c->n_trials_global = Rtot;
  double fsum = 0.0;
  for (int nn = 0; nn < q; ++nn) fsum += c->hbuf[(size_t)(p + 1) * q + nn];
  double cst = -fsum / Rtot;
  for (int i = 0; i < q * (p + 1); ++i) grad[i] = -c->hbuf[i] / Rtot;
  if (prior_center) {
    double s = 0.0;
    for (int i = 0; i < q * (p + 1); ++i) {
      const double dv = vecCd[i] - prior_center[i];
      s += dv * dv;
      grad[i] += inv_s2 * dv;
    }
    cst += 0.5 * inv_s2 * s;
  }
  *cost = cst;
  return 0;
}


// Tail of both Newton-pass variants: the q independent (p+1)-dim Newton steps on the reduced sums in c->cdhout (trial count at rtot_dev, read
// by the kernel), then ONE read-back of [cost sums | delta | dec | R] through pinned memory.
static int cd_newton_finish(pgpfa_ctx* c, const double* rtot_dev, const double* vecCd, const double* prior_center, double inv_s2, double* cost_n,
                            double* delta, double* dec) {
  const int q = c->q, p = c->p, D = p + 1;
  const int th = std::max(1, std::min(32, (int)(48 * 1024 / ((D * D + 2 * D) * sizeof(double)))));
  hipLaunchKernelGGL(cd_newton_step_kernel, dim3((q + th - 1) / th), dim3(th), (size_t)th * (D * D + 2 * D) * sizeof(double), c->st, c->cdhout, q, p,
                     rtot_dev, c->vec, prior_center ? c->cdcenter : nullptr, inv_s2, c->cdpack);
  HIPC(hipGetLastError());
  const size_t np = (size_t)q * (D + 2) + 1;
  CHK(ensure_hbuf(c, np));
  CHK(download(c, c->hbuf, c->cdpack, np));
  const double Rtot = c->hbuf[np - 1];
  c->n_trials_global = Rtot;
  std::memcpy(delta, c->hbuf + q, (size_t)q * D * sizeof(double));
  std::memcpy(dec, c->hbuf + (size_t)q * (D + 1), (size_t)q * sizeof(double));
  for (int n = 0; n < q; ++n) {
    double cn = -c->hbuf[n] / Rtot;                          // row 0 of the sums: sum (y*hh - yhat) per neuron
    if (prior_center) {
      double s2 = 0.0;
      for (int i = 0; i < D; ++i) { const double dv = vecCd[(size_t)i * q + n] - prior_center[(size_t)i * q + n]; s2 += dv * dv; }
      cn += 0.5 * inv_s2 * s2;
    }
    cost_n[n] = cn;
  }
  return 0;
}

// One pass of the device Newton solver for the (C,d) M-step: cost, gradient and per-neuron Hessians at vecCd
// (mstep_cd_hess_kernel), all-reduced over ranks, then the q independent (p+1)-dim Newton steps on device.
int pgpfa_mstep_cd_newton_pass(pgpfa_ctx* c, const double* vecCd, const double* prior_center, double inv_s2, double* cost_n,
                               double* delta, double* dec) {
  PhaseRange range_phase("pgpfa.mstep_cd_newton_pass");
  if (!c || !vecCd || !cost_n || !delta || !dec) return fail("null argument");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  if (!c->have_post) return fail("no E-step result resident: run an E-step or pgpfa_set_posterior first");
  if (c->p > 32) return fail("device Newton M-step supports up to 32 latents (p=%d): use a scipy method", c->p);
  HIPC(hipSetDevice(c->device));
  const int q = c->q, p = c->p, T = c->T, D = p + 1;
  const int NH = 1 + D + D * (D + 1) / 2;
  CHK(upload_nosync(c, c->vec, vecCd, (size_t)q * D * sizeof(double)));
  if (prior_center) CHK(upload_nosync(c, c->cdcenter, prior_center, (size_t)q * D * sizeof(double)));
  CdArgs a{};
  a.Y = c->Y; a.Yhi = c->Yhi; a.mean = c->Xmode; a.vsm = c->vsm; a.vec = c->vec;
  a.trials = c->last_trials; a.ntr = (int)c->last_trials_h.size();
  a.part = c->cdhpart; a.q = q; a.p = p; a.T = T; a.dbg = c->cd_debug;
  int nby = std::max(1, std::min(a.ntr * 4, 128));
  const bool on_mfma = c->mfma && c->cd_mfma && c->cd_hess_mfma && p <= 10;     // two-stage matrix-core form (mstep.h)
  if (on_mfma) CHK(ensure_cdym(c));
  prof_begin(c, TAG_CD, (double)a.ntr * q * T * (3.0 * p * p + 12.0 * p));
  dispatch_pw(p, [&](auto pw) {
    constexpr int PW = decltype(pw)::value;
    if constexpr (PW <= 10) {
      if (on_mfma) {
        const int ntt = (T + CdH<PW>::BT - 1) / CdH<PW>::BT;
        const int tiles = (q + 15) / 16, groups = (tiles + CDH_NW - 1) / CDH_NW, tpg = (tiles + groups - 1) / groups;
        nby = std::max(1, std::min(a.ntr * ntt, std::min(128, std::max(64, 512 / groups))));
        // (per launch: the attribute belongs to the function object of the current device, and contexts of one process may sit on different devices)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mstep_cd_hess_mfma_kernel<PW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)cd_hess_mfma_lds_bytes<PW>());
        hipLaunchKernelGGL(mstep_cd_hess_mfma_kernel<PW>, dim3(nby, groups), dim3(64, CDH_NW), cd_hess_mfma_lds_bytes<PW>(), c->st, a, tpg);
        return;
      }
    }
    if constexpr (PW <= 12) {
      hipLaunchKernelGGL(mstep_cd_hess_kernel<PW>, dim3((q + 63) / 64, nby), dim3(64, CDH_KY), 0, c->st, a);
    } else {
      constexpr int NG = CdGroups<PW>::NG;               // Hessian rows dealt to NG row groups (blockIdx.z)
      hipLaunchKernelGGL((mstep_cd_hess_rows_kernel<PW, NG>), dim3((q + 63) / 64, nby, NG), dim3(64, CDH_KY), 0, c->st, a);
    }
  });
  prof_end(c);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((NH * q + 31) / 32), dim3(256), 0, c->st, c->cdhpart, nby, NH * q, c->cdhout);
  if (on_mfma) hipLaunchKernelGGL(cd_hess_add_ym_kernel, dim3((q + 127) / 128), dim3(128), 0, c->st, c->cdhout, c->cdym, c->vec, q, p);
  HIPC(hipGetLastError());
  const double cnt = (double)a.ntr;
  CHK(upload_nosync(c, c->cdhout + (size_t)NH * q, &cnt, sizeof(double)));
  CHK(allreduce_dev(c, c->cdhout, (size_t)NH * q + 1));
  c->cd_hess_valid = true;
  c->cd_hess_ntr = a.ntr;
  return cd_newton_finish(c, c->cdhout + (size_t)NH * q, vecCd, prior_center, inv_s2, cost_n, delta, dec);
}

// Chord variant of the pass above: cost and gradient are evaluated at vecCd (the cheap kernel), the per-neuron
// Hessians are the ones of the last pgpfa_mstep_cd_newton_pass (still a descent direction: they are SPD).
int pgpfa_mstep_cd_chord_pass(pgpfa_ctx* c, const double* vecCd, const double* prior_center, double inv_s2, double* cost_n,
                              double* delta, double* dec) {
  PhaseRange range_phase("pgpfa.mstep_cd_chord_pass");
  if (!c || !vecCd || !cost_n || !delta || !dec) return fail("null argument");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  if (!c->have_post) return fail("no E-step result resident: run an E-step or pgpfa_set_posterior first");
  if (!c->cd_hess_valid) return fail("no per-neuron Hessians resident: call pgpfa_mstep_cd_newton_pass first");
  HIPC(hipSetDevice(c->device));
  const int q = c->q, p = c->p, D = p + 1;
  const int NH = 1 + D + D * (D + 1) / 2;
  const int len = (p + 2) * q;
  CHK(upload_nosync(c, c->vec, vecCd, (size_t)q * D * sizeof(double)));
  if (prior_center) CHK(upload_nosync(c, c->cdcenter, prior_center, (size_t)q * D * sizeof(double)));
  CHK(cd_sweep(c));
  const double cnt = (double)c->last_trials_h.size();
  CHK(upload_nosync(c, c->cdout + len, &cnt, sizeof(double)));
  CHK(allreduce_dev(c, c->cdout, (size_t)len + 1));
  hipLaunchKernelGGL(cd_chord_merge_kernel, dim3((len + 255) / 256), dim3(256), 0, c->st, c->cdout, q, p, c->cdhout);
  (void)NH;
  return cd_newton_finish(c, c->cdout + len, vecCd, prior_center, inv_s2, cost_n, delta, dec);
}

// per-neuron values of the (C,d) cost (for the Newton line search); same kernel as pgpfa_mstep_cd_costgrad
int pgpfa_mstep_cd_cost_per_neuron(pgpfa_ctx* c, const double* vecCd, const double* prior_center, double inv_s2, double* cost_n) {
  if (!c || !vecCd || !cost_n) return fail("null argument");
  std::vector<double> grad((size_t)c->q * (c->p + 1));
  double total = 0.0;
  CHK(pgpfa_mstep_cd_costgrad(c, vecCd, nullptr, 0.0, &total, grad.data()));
  const int q = c->q, p = c->p;
  const double Rtot = c->n_trials_global;
  for (int n = 0; n < q; ++n) {
    double cn = -c->hbuf[(size_t)(p + 1) * q + n] / Rtot;   // hbuf still holds the reduced sums of that call
    if (prior_center) {
      double s2 = 0.0;
      for (int i = 0; i <= p; ++i) { const double dv = vecCd[(size_t)i * q + n] - prior_center[(size_t)i * q + n]; s2 += dv * dv; }
      cn += 0.5 * inv_s2 * s2;
    }
    cost_n[n] = cn;
  }
  return 0;
}

int pgpfa_mstep_precomp(pgpfa_ctx* c, double* num_trials) {
  PhaseRange range_phase("pgpfa.mstep_precomp");
  if (!c) return fail("null context");
  if (!c->have_post) return fail("no E-step result resident");
  if (c->tau_inflight) return fail("a timescale pass is in flight (pgpfa_mstep_tau_costgrad_multi_begin): collect it first");
  HIPC(hipSetDevice(c->device));
  const int ntr = (int)c->last_trials_h.size();
  bool contiguous = ntr > 0 && ntr % 16 == 0;
  for (int i = 1; i < ntr && contiguous; ++i) contiguous = (c->last_trials_h[i] == c->last_trials_h[0] + i);
  if (c->pacc_valid && contiguous) {
    // PautoSum[k] = Pacc[k] + M_k M_k^T with M_k = [m_rk]_r (T x ntr, the trials' mean rows side by side in Xmode): one GEMM
    const size_t len = (size_t)c->Tp * c->Tp * c->p;
    CHK(copy_dev(c, c->Pauto, c->Pacc, len * sizeof(double)));
    GemmP g{};
    g.A = c->Xmode + (size_t)c->last_trials_h[0] * c->n; g.sA = c->T; g.lda = c->n;
    g.B = g.A; g.sB = c->T; g.ldb = c->n;
    g.C = c->Pauto; g.sC = (long long)c->Tp * c->Tp; g.ldc = c->Tp;
    g.M = c->T; g.N = c->T; g.K = ntr; g.alpha = 1.0; g.beta = 1.0;
    g.slots = nullptr; g.nbatch = c->p; g.mode = GEMM_FULL; g.kflags = 0;
    CHK(gemm(c, false, g));
  } else if (c->pacc_valid) {
    hipLaunchKernelGGL(pauto_from_acc_kernel, dim3(c->Tp, c->p), dim3(128), 0, c->st, c->Pacc, c->Xmode, c->last_trials, ntr, c->T, c->Tp, c->p,
                       c->Pauto);
  } else {
    CHK(ensure_vsmgp_buffer(c));
    CHK(ensure_trial_vsmgp(c, c->last_trials_h));
    hipLaunchKernelGGL(pautosum_kernel, dim3(c->Tp, c->p), dim3(128), 0, c->st, c->vsmgp, c->Xmode, c->last_trials, ntr, c->T, c->Tp, c->p, c->Pauto);
  }
  HIPC(hipGetLastError());
  const size_t len = (size_t)c->Tp * c->Tp * c->p;
  CHK(allreduce_dev(c, c->Pauto, len));
  double cnt = (double)ntr;
  if (c->comm) {
    CHK(upload_nosync(c, c->tscal, &cnt, sizeof(double)));
    CHK(allreduce_dev(c, c->tscal, 1));
    CHK(download(c, &cnt, c->tscal, 1));
  }
  HIPC(hipStreamSynchronize(c->st));
  c->n_trials_global = cnt;
  c->have_precomp = true;
  if (num_trials) *num_trials = cnt;
  return 0;
}

int pgpfa_get_pautosum(pgpfa_ctx* c, double* out) {
  if (!c || !out) return fail("null argument");
  if (!c->have_precomp) return fail("pgpfa_mstep_precomp has not been called");
  HIPC(hipSetDevice(c->device));
  return get_slabs(c, c->Pauto, out);
}

static int dot_slabs(pgpfa_ctx* c, const double* A, const double* B, size_t n, double* out_dev) {
  const int nbk = 256;
  hipLaunchKernelGGL(dot_part_kernel, dim3(nbk), dim3(256), 0, c->st, A, B, (long long)n, c->tpart);
  hipLaunchKernelGGL(sum_part_kernel, dim3(1), dim3(64), 0, c->st, c->tpart, nbk, out_dev);
  HIPC(hipGetLastError());
  return 0;
}

int pgpfa_mstep_tau_costgrad(pgpfa_ctx* c, int k, double logp, double* cost, double* grad) {
  if (!c || !cost || !grad) return fail("null argument");
  if (!c->have_precomp) return fail("pgpfa_mstep_precomp has not been called");
  if (k < 0 || k >= c->p) return fail("latent %d out of range", k);
  if (!std::isfinite(logp)) return fail("log-gamma is not finite");
  HIPC(hipSetDevice(c->device));
  const int Tp = c->Tp;
  const size_t slab = (size_t)Tp * Tp;
  const double* P = c->Pauto + (size_t)k * slab;
  hipLaunchKernelGGL(gram_gamma_kernel, dim3(Tp), dim3(256), 0, c->st, c->tK, c->tM, Tp, c->T, logp, c->eps);
  CHK(copy_dev(c, c->kws.H, c->tK, slab * sizeof(double)));
  HIPC(hipMemsetAsync(c->kws.info, 0, sizeof(int), c->st));
  CHK(factor(c, c->kws, nullptr, 1));
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, c->st, c->kws.H, Tp, Tp, c->tscal + 0);
  CHK(inverse_t(c, c->kws, nullptr, 1));
  GemmP g{};
  g.A = c->kws.Mt; g.sA = 0; g.lda = Tp; g.B = c->kws.Mt; g.sB = 0; g.ldb = Tp;
  g.C = c->tK; g.sC = 0; g.ldc = Tp;                       // tK <- Kinv
  g.M = Tp; g.N = Tp; g.K = Tp; g.alpha = 1.0; g.beta = 0.0; g.slots = nullptr; g.nbatch = 1; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
  CHK(gemm(c, false, g));
  GemmP a1 = g;                                            // A1 = Kinv * M   (M symmetric)
  a1.A = c->tK; a1.B = c->tM; a1.C = c->tA1; a1.kflags = 0;
  CHK(gemm(c, false, a1));
  GemmP a2 = g;                                            // A2 = P * Kinv   (Kinv symmetric)
  a2.A = P; a2.B = c->tK; a2.C = c->tA2; a2.kflags = 0;
  CHK(gemm(c, false, a2));
  CHK(dot_slabs(c, c->tK, P, slab, c->tscal + 1));         // tr(Kinv P)
  CHK(dot_slabs(c, c->tK, c->tM, slab, c->tscal + 2));     // tr(Kinv M)
  CHK(dot_slabs(c, c->tA1, c->tA2, slab, c->tscal + 3));   // tr(Kinv M Kinv P)
  double h[4];
  int info = 0;
  CHK(dl_enqueue(c, h, c->tscal, 4 * sizeof(double)));
  CHK(dl_enqueue(c, &info, c->kws.info, sizeof(int)));
  CHK(dl_flush(c));
  if (info != 0) return fail("timescale Gram matrix not positive definite at log-gamma=%g (pivot %d)", logp, info);
  const double R = c->n_trials_global;
  *cost = 0.5 * R * h[0] + 0.5 * h[1];                                    // learning.py:212-214
  const double dE = -0.5 * R * h[2] + 0.5 * h[3];                          // learning.py:253
  *grad = -dE * std::exp(logp);                                            // learning.py:255
  return 0;
}


// m candidate points per latent in ONE batched pass (queries ordered candidate-major: j = cand * p + latent).  The
// pass is latency bound (a chain of ~30 small launches on T x T matrices), so evaluating 4 p matrices costs about
// the same as p: the host-side root finder uses that to bracket and interpolate instead of stepping serially.
// the launches of one batched pass on c->st: logp already at dlogp; results to dres [4][nq], factor flags to c->kws.info
static int tau_pass_enqueue(pgpfa_ctx* c, int nq, double* dlogp, double* dres) {
  const int Tp = c->Tp, p = c->p;
  const size_t slab = (size_t)Tp * Tp;
  hipLaunchKernelGGL(gram_gamma_batch_kernel, dim3(Tp, nq), dim3(256), 0, c->st, c->tK, c->tM, Tp, c->T, dlogp, c->eps);
  CHK(copy_dev(c, c->kws.H, c->tK, slab * nq * sizeof(double)));
  HIPC(hipMemsetAsync(c->kws.info, 0, sizeof(int) * nq, c->st));
  CHK(factor(c, c->kws, nullptr, nq));
  hipLaunchKernelGGL(logdet_batch_kernel, dim3(nq), dim3(256), 0, c->st, c->kws.H, (long long)slab, Tp, Tp, dres);
  CHK(inverse_t(c, c->kws, nullptr, nq));
  GemmP g{};
  g.A = c->kws.Mt; g.sA = slab; g.lda = Tp; g.B = c->kws.Mt; g.sB = slab; g.ldb = Tp;
  g.C = c->tK; g.sC = slab; g.ldc = Tp;                       // tK <- Kinv
  g.M = Tp; g.N = Tp; g.K = Tp; g.alpha = 1.0; g.beta = 0.0; g.slots = nullptr; g.nbatch = nq; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
  CHK(gemm(c, false, g));
  GemmP a1 = g;                                              // A1 = Kinv * M
  a1.A = c->tK; a1.B = c->tM; a1.C = c->tA1; a1.kflags = 0;
  CHK(gemm(c, false, a1));
  GemmP a2 = g;                                              // A2 = P * Kinv: P is per latent (lo index), Kinv / A2 per query
  a2.A = c->Pauto; a2.B = c->tK; a2.C = c->tA2; a2.kflags = 0;
  a2.nb_lo = p; a2.sA_hi = 0; a2.sB_hi = (long long)p * slab; a2.sC_hi = (long long)p * slab;
  CHK(gemm(c, false, a2));
  const int nbk = 64;
  auto bdot = [&](const double* A, const double* B, double* out, int bmod) {
    hipLaunchKernelGGL(dot_part_batch_kernel, dim3(nbk, nq), dim3(256), 0, c->st, A, (long long)slab, B, (long long)slab, (long long)slab, c->tpart, bmod);
    hipLaunchKernelGGL(sum_part_batch_kernel, dim3(1), dim3(64), 0, c->st, c->tpart, nbk, out, nq);
  };
  bdot(c->tK, c->Pauto, dres + nq, p);
  bdot(c->tK, c->tM, dres + 2 * nq, 0);
  bdot(c->tA1, c->tA2, dres + 3 * nq, 0);
  HIPC(hipGetLastError());
  return 0;
}

static int tau_pass_finish(pgpfa_ctx* c, int nq, const double* logp, const double* h, const int* info, double* cost, double* grad) {
  const double R = c->n_trials_global;
  for (int k = 0; k < nq; ++k) {
    if (info[k] != 0) return fail("timescale Gram matrix of latent %d not positive definite at log-gamma=%g", k % c->p, logp[k]);
    cost[k] = 0.5 * R * h[k] + 0.5 * h[nq + k];
    const double dE = -0.5 * R * h[2 * nq + k] + 0.5 * h[3 * nq + k];
    grad[k] = -dE * std::exp(logp[k]);
  }
  return 0;
}

static int tau_pass_check(pgpfa_ctx* c, int m, const double* logp) {
  if (m < 1 || m > TAU_MULTI_MAX) return fail("between 1 and %d candidates per latent (m=%d)", TAU_MULTI_MAX, m);
  if (!c->have_precomp) return fail("pgpfa_mstep_precomp has not been called");
  for (int k = 0; k < m * c->p; ++k)
    if (!std::isfinite(logp[k])) return fail("log-gamma[%d] is not finite", k % c->p);
  return 0;
}

int pgpfa_mstep_tau_costgrad_multi(pgpfa_ctx* c, int m, const double* logp, double* cost, double* grad) {
  PhaseRange range_phase("pgpfa.mstep_tau_costgrad");
  if (!c || !logp || !cost || !grad) return fail("null argument");
  CHK(tau_pass_check(c, m, logp));
  if (c->tau_inflight) return fail("a timescale pass started with pgpfa_mstep_tau_costgrad_multi_begin has not been collected");
  HIPC(hipSetDevice(c->device));
  const int nq = m * c->p;
  double* dlogp = c->tscal + 16;                     // [nq]
  double* dres = c->tscal + 16 + nq;                 // [4][nq]: logdet, tr(KinvP), tr(KinvM), tr(KinvMKinvP)
  CHK(upload(c, dlogp, logp, nq));
  CHK(tau_pass_enqueue(c, nq, dlogp, dres));
  std::vector<double> h(4 * (size_t)nq);
  std::vector<int> info(nq);
  CHK(dl_enqueue(c, h.data(), dres, 4 * nq * sizeof(double)));
  CHK(dl_enqueue(c, info.data(), c->kws.info, sizeof(int) * nq));
  CHK(dl_flush(c));
  return tau_pass_finish(c, nq, logp, h.data(), info.data(), cost, grad);
}

// The same pass split in two (round 6): _begin enqueues it on the context's SIDE stream and returns, _end waits for it and hands out the numbers.
// Between the two the caller is free to run other work of the same context on the main stream - the (C,d) passes of the M-step: the timescale
// pass is a latency-bound chain of ~45 small launches on 40 matrices of T x T (1.3 ms during which most of the chip idles), a (C,d) pass one or
// two compute-bound launches; they share nothing but the device (the pass reads PautoSum, left by pgpfa_mstep_precomp, and its own scratch).
// Same arithmetic in the same order as pgpfa_mstep_tau_costgrad_multi: the results are the same bits.  One pass in flight per context.
int pgpfa_mstep_tau_costgrad_multi_begin(pgpfa_ctx* c, int m, const double* logp) {
  PhaseRange range_phase("pgpfa.mstep_tau_costgrad_begin");
  if (!c || !logp) return fail("null argument");
  CHK(tau_pass_check(c, m, logp));
  if (c->tau_inflight) return fail("a timescale pass is already in flight: collect it with pgpfa_mstep_tau_costgrad_multi_end first");
  if (!c->st2) return fail("no side stream in this context");
  HIPC(hipSetDevice(c->device));
  const int nq = m * c->p;
  if (!c->tau_pin) {
    // pinned block: [logp (TAU_MULTI_MAX p) | results 4 x (TAU_MULTI_MAX p) | flags]
    const size_t cap = (size_t)TAU_MULTI_MAX * c->p;
    if (hipHostMalloc((void**)&c->tau_pin, (5 * cap) * sizeof(double) + cap * sizeof(int), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError(); c->tau_pin = nullptr; return fail("hipHostMalloc for the asynchronous timescale pass failed");
    }
    if (hipEventCreateWithFlags(&c->ev_tau_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_tau_done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError(); return fail("hipEventCreate for the asynchronous timescale pass failed");
    }
  }
  const size_t cap = (size_t)TAU_MULTI_MAX * c->p;
  double* pin_logp = c->tau_pin; double* pin_res = c->tau_pin + cap; int* pin_info = reinterpret_cast<int*>(c->tau_pin + 5 * cap);
  std::memcpy(pin_logp, logp, sizeof(double) * nq);
  double* dlogp = c->tscal + 16;
  double* dres = c->tscal + 16 + nq;
  // everything the main stream has been given so far (PautoSum among it) comes first; then the pass runs beside whatever follows there
  HIPC(hipEventRecord(c->ev_tau_fork, c->st));
  HIPC(hipStreamWaitEvent(c->st2, c->ev_tau_fork, 0));
  int rc = 0;
  {
    struct StreamSwap { pgpfa_ctx* c; hipStream_t keep; ~StreamSwap() { c->st = keep; } } swap{c, c->st};
    c->st = c->st2;                                 // (single-threaded by contract: the helpers below launch on c->st)
    if (hipMemcpyAsync(dlogp, pin_logp, sizeof(double) * nq, hipMemcpyHostToDevice, c->st) != hipSuccess) rc = fail("hipMemcpyAsync: %s", hipGetErrorString(hipGetLastError()));
    if (!rc) rc = tau_pass_enqueue(c, nq, dlogp, dres);
    if (!rc && hipMemcpyAsync(pin_res, dres, 4 * sizeof(double) * nq, hipMemcpyDeviceToHost, c->st) != hipSuccess) rc = fail("hipMemcpyAsync: %s", hipGetErrorString(hipGetLastError()));
    if (!rc && hipMemcpyAsync(pin_info, c->kws.info, sizeof(int) * nq, hipMemcpyDeviceToHost, c->st) != hipSuccess) rc = fail("hipMemcpyAsync: %s", hipGetErrorString(hipGetLastError()));
    if (!rc && hipEventRecord(c->ev_tau_done, c->st) != hipSuccess) rc = fail("hipEventRecord: %s", hipGetErrorString(hipGetLastError()));
  }
  if (rc) { (void)hipStreamSynchronize(c->st2); return rc; }
  c->tau_inflight = m;
  return 0;
}

int pgpfa_mstep_tau_costgrad_multi_end(pgpfa_ctx* c, double* cost, double* grad) {
  PhaseRange range_phase("pgpfa.mstep_tau_costgrad_end");
  if (!c || !cost || !grad) return fail("null argument");
  if (!c->tau_inflight) return fail("no timescale pass in flight");
  HIPC(hipSetDevice(c->device));
  const int m = c->tau_inflight, nq = m * c->p;
  c->tau_inflight = 0;
  HIPC(hipEventSynchronize(c->ev_tau_done));
  const size_t cap = (size_t)TAU_MULTI_MAX * c->p;
  return tau_pass_finish(c, nq, c->tau_pin, c->tau_pin + cap, reinterpret_cast<const int*>(c->tau_pin + 5 * cap), cost, grad);
}

int pgpfa_mstep_tau_costgrad_batch(pgpfa_ctx* c, const double* logp, double* cost, double* grad) {
  return pgpfa_mstep_tau_costgrad_multi(c, 1, logp, cost, grad);
}


