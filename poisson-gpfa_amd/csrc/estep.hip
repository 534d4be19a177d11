// libpgpfa_hip.so - estep.hip (one translation unit of the C-ABI library; shared declarations: ctx.h)
#include "ctx.h"
#include "model.h"
#include "chol.h"
#include "pcg.h"
#include "thin.h"
#include "dual.h"

using namespace pgpfa;

// Poisson pass over the slots in d_list (nl of them): X source -> G/W destinations, flik per slot
int poisson(pgpfa_ctx* c, const int* d_list, int nl, const double* X, double* G, double* W, double* flik, int full) {
  PoissonArgs a{};
  a.Y = c->Y; a.Yhi = c->Yhi; a.C = c->C; a.d = c->d;
  a.X = X; a.sX = c->ld; a.G = G; a.sG = c->ld; a.W = W; a.sW = (long long)c->T * c->p * c->p;
  a.fpart = c->fpart; a.slots = d_list; a.trial_of_slot = c->trial_of_slot;
  a.mask = c->mask_active ? c->mask_of_slot : nullptr;
  a.off = c->var_active ? c->voff : nullptr; a.sOff = (long long)c->q * c->T;
  a.lam_out = c->lam_out_active ? c->lamd : nullptr; a.sLam = (long long)c->q * c->T;
  a.q = c->q; a.p = c->p; a.T = c->T; a.ntile = (c->T + 63) / 64; a.full = full;
  const int KY = poisson_rows(c->p);
  dim3 grid(a.ntile, nl), block(64, KY);
  const double fl = (double)nl * c->q * c->T * (4.0 * c->p + (full ? c->p * (c->p + 1.0) : 0.0));
  // Latent widths beyond the matrix-core kernel: the neuron contractions as the GEMMs of the dual evaluation (dual.h) once the dual scratch exists
  // (the variational fixed point allocates it: config 5, 20 latents - the vector kernel below took a third of its time)
  // (the choice follows from the configuration alone - the scratch is allocated here when no dual call has done so yet: a Laplace E-step at these
  // widths must not depend on what ran before it in the context)
  const bool gemm_form = !(c->CCu && c->mfma) && c->mfma && c->dual_gemm && c->dual_tbl && !c->mask_active && !c->lam_out_active;
  if (gemm_form) {
    CHK(ensure_lambda(c));
    prof_begin(c, TAG_POISSON, fl);
    hipLaunchKernelGGL(rates_wide_kernel, grid, dim3(256), (size_t)c->p * 64 * sizeof(double), c->st, c->Y, c->Yhi, c->C, c->d, X, (long long)c->ld,
                       a.off, c->lamd, c->dgrad, c->fpart, d_list, c->trial_of_slot, c->q, c->p, c->T);
    prof_end(c);
    if (full) {
      const int np = c->p * (c->p + 1) / 2;
      GemmP w{};                                               // Wp (T x pairs) = Lambda^T . TBL[:, pairs]
      w.A = c->lamd; w.sA = (long long)c->q * c->T; w.lda = c->T;
      w.B = c->dual_tbl; w.sB = 0; w.ldb = c->dual_ncol;
      w.C = c->dual_scr; w.sC = c->dual_sscr; w.ldc = c->T;
      w.M = c->T; w.N = np; w.K = c->qpad; w.alpha = 1.0; w.beta = 0.0; w.slots = d_list; w.nbatch = nl; w.mode = GEMM_FULL; w.kflags = 0;
      CHK(gemm(c, false, w));
      GemmP v = w;                                             // G (T x p, i.e. [p][T]) = (Lambda - Y)^T . TBL[:, latents]
      v.A = c->dgrad; v.B = c->dual_tbl + c->dual_npd; v.C = G; v.sC = c->ld; v.N = c->p;
      CHK(gemm(c, false, v));
      const int nbu = dual_tile_bins((size_t)np);
      hipLaunchKernelGGL(dual_unpack_w_kernel, dim3((unsigned)((c->T + nbu - 1) / nbu), nl), dim3(256), (size_t)np * (nbu + 1) * sizeof(double), c->st, c->dual_scr,
                         c->dual_sscr, W, (long long)c->T * c->p * c->p, c->T, c->p, nbu, d_list);
    }
    hipLaunchKernelGGL(sum_tiles_kernel, dim3((nl + 255) / 256), dim3(256), 0, c->st, c->fpart, a.ntile, d_list, nl, flik);
    HIPC(hipGetLastError());
    return 0;
  }
  prof_begin(c, TAG_POISSON, fl);
  if (c->CCu && c->mfma) {
    // bin tiles per wave (option poisson_tiles): the table fragments of a neuron tile are read from L2 once per wave and serve that many tiles
    const int nbt = (c->poisson_tiles >= 2 && c->p <= 10) ? 2 : 1;
    a.ntile = (c->T + 64 * nbt - 1) / (64 * nbt);
    const dim3 gridm(a.ntile, nl);
    dispatch_pw(c->p, [&](auto pm) {
      constexpr int PW = decltype(pm)::value;
      if constexpr (PW <= 10) {
        if (nbt == 2) hipLaunchKernelGGL((poisson_mfma_kernel<PW, 2>), gridm, dim3(256), 0, c->st, a, c->CCu, c->C16, c->qpad);
        else hipLaunchKernelGGL((poisson_mfma_kernel<PW, 1>), gridm, dim3(256), 0, c->st, a, c->CCu, c->C16, c->qpad);
      } else if constexpr (PW <= 16) {
        hipLaunchKernelGGL((poisson_mfma_kernel<PW, 1>), gridm, dim3(256), 0, c->st, a, c->CCu, c->C16, c->qpad);
      }
    });
  } else {
    dispatch_pw(c->p, [&](auto pm) { hipLaunchKernelGGL(poisson_pass_kernel<decltype(pm)::value>, grid, block, 0, c->st, a); });
  }
  prof_end(c);
  hipLaunchKernelGGL(sum_tiles_kernel, dim3((nl + 255) / 256), dim3(256), 0, c->st, c->fpart, a.ntile, d_list, nl, flik);
  HIPC(hipGetLastError());
  return 0;
}

int prior_mv(pgpfa_ctx* c, const int* d_list, int nl, const double* in, double* out, const double* mat) {
  hipLaunchKernelGGL(prior_matvec_kernel, dim3(c->p, nl), dim3(256), c->T * sizeof(double), c->st, mat ? mat : c->Kinv, c->Tp, c->T, c->p,
                     in, (long long)c->ld, out, (long long)c->ld, d_list);
  HIPC(hipGetLastError());
  return 0;
}

// out[slot][k] = mat_k * in[slot][k] for ALL slots [0,nb) as one batched MFMA GEMM (batch = latents, N = slots)
// (cols / ncols: only the listed slots - the live ones of a Newton-PCG solve; the product then has ncols columns)
int prior_mv_all(pgpfa_ctx* c, int nb, const double* in, double* out, const double* mat, const int* skip,
                 const int* cols, int ncols) {
  GemmP g{};
  g.skip = skip;
  if (cols) { g.cols = cols; nb = ncols; }
  g.A = mat ? mat : c->Kinv; g.sA = (long long)c->Tp * c->Tp; g.lda = c->Tp;
  g.B = in; g.sB = c->T; g.ldb = c->ld;                 // latent k: rows k*T.. of every slot vector (K x N column-major)
  g.C = out; g.sC = c->T; g.ldc = c->ld;
  g.M = c->T; g.N = nb; g.K = round_up(c->T, 16); g.alpha = 1.0; g.beta = 0.0;
  g.slots = nullptr; g.nbatch = c->p; g.mode = GEMM_FULL; g.kflags = 0;
  return gemm(c, true, g);
}

int assemble(pgpfa_ctx* c, const int* d_list, int nl, double diag_scale) {
  prof_begin(c, TAG_ASSEMBLE, 0.0);
  hipLaunchKernelGGL(assemble_h_kernel, dim3(c->npad, nl), dim3(256), 0, c->st, c->ws.H, c->ws.sH, c->ld, c->npad, c->n, c->T, c->Tp,
                     c->p, c->Kinv, c->W, (long long)c->T * c->p * c->p, d_list, diag_scale);
  prof_end(c);
  HIPC(hipGetLastError());
  return 0;
}


// load X[nb][p][T] host points into the chunk slots and bind slot -> trial
static int load_points(pgpfa_ctx* c, const std::vector<int>& trials, int c0, int nb, const double* X) {
  std::vector<int> tos(trials.begin() + c0, trials.begin() + c0 + nb);
  CHK(upload_list(c, c->trial_of_slot, tos));
  HIPC(hipMemcpy2DAsync(c->Xc, (size_t)c->ld * sizeof(double), X + (size_t)c0 * c->n, (size_t)c->n * sizeof(double),
                        (size_t)c->n * sizeof(double), nb, hipMemcpyHostToDevice, c->st));
  return 0;
}

int pgpfa_laplace_eval(pgpfa_ctx* c, int n, const int32_t* idx, const double* X, double* f, double* grad) {
  CHK(ready_estep(c, c->B > 0 ? c->plan_lowrank : true));   // vectors only: any workspace plan serves
  if (!X || !f) return fail("null argument");
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr));
  const int N = (int)tr.v.size();
  for (int c0 = 0; c0 < N; c0 += c->B) {
    const int nb = std::min(c->B, N - c0);
    CHK(load_points(c, tr.v, c0, nb, X));
    CHK(prior_mv(c, c->ident, nb, c->Xc, c->KX));
    hipLaunchKernelGGL(dots3_kernel, dim3(nb), dim3(256), 0, c->st, c->Xc, (long long)c->ld, c->KX, (long long)c->ld, (const double*)nullptr,
                       0LL, (const double*)nullptr, 0LL, c->n, c->ident, c->sc_qxx, c->sc_qdx, c->sc_qdd);
    CHK(poisson(c, c->ident, nb, c->Xc, c->Gl, c->W, c->sc_f, 1));
    CHK(ensure_hbuf(c, 2 * (size_t)nb));
    HIPC(hipMemcpyAsync(c->hbuf, c->sc_f, nb * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPC(hipMemcpyAsync(c->hbuf + nb, c->sc_qxx, nb * sizeof(double), hipMemcpyDeviceToHost, c->st));
    if (grad) {
      hipLaunchKernelGGL(grad_total_kernel, dim3((c->n + 255) / 256, nb), dim3(256), 0, c->st, c->Gl, (long long)c->ld, c->KX,
                         (long long)c->ld, c->Gt, (long long)c->ld, c->n, c->ident);
      HIPC(hipMemcpy2DAsync(grad + (size_t)c0 * c->n, (size_t)c->n * sizeof(double), c->Gt, (size_t)c->ld * sizeof(double),
                            (size_t)c->n * sizeof(double), nb, hipMemcpyDeviceToHost, c->st));
    }
    HIPC(hipStreamSynchronize(c->st));
    for (int s = 0; s < nb; ++s) f[c0 + s] = c->hbuf[s] + 0.5 * c->hbuf[nb + s];
  }
  return 0;
}

int pgpfa_laplace_hessian(pgpfa_ctx* c, int trial, const double* X, double* H) {
  CHK(ready(c));
  if (!X || !H) return fail("null argument");
  if (trial < 0 || trial >= c->R) return fail("trial %d out of range", trial);
  std::vector<int> tr{trial};
  CHK(load_points(c, tr, 0, 1, X));
  CHK(poisson(c, c->ident, 1, c->Xc, c->Gl, c->W, c->sc_f, 1));
  // dense n x n into the (unused) Mt slab of slot 0 would break its zero pattern: use the H slab
  hipLaunchKernelGGL(dense_h_kernel, dim3(c->n), dim3(256), 0, c->st, c->ws.H, c->n, c->T, c->Tp, c->p, c->Kinv, c->W);
  HIPC(hipGetLastError());
  return download(c, H, c->ws.H, (size_t)c->n * c->n);
}


// Z <- P^-1 R for the nb slot vectors at once.  Dense plan: the shared preconditioner is ONE matrix, its explicit
// inverse is formed once per chunk and applied with a single multi-RHS GEMM.  Low-rank plan: the same matrix in
// Woodbury form, P^-1 v = Gb (eps v + F Sb F^T Gb v), with Gb the per-bin blocks of the mean curvature and Sb the
// inverse of the r x r system: two per-bin kernels and three thin GEMMs, no n x n matrix anywhere.
// (skip: device stop flag of the inner PCG loop; final_apply = false leaves the last per-bin application to the caller, with
// y = F Sb F^T Gb R in c->Xt; first_apply = false: the caller has already put Gb R into c->Xt)
// (vec_row > 0: the n-vectors c->Xt holds - input t and output y - are in the inner solve's private layout, latent rows vec_row apart (pcg.h:
// PcgCgP::Tl); only with the thin kernels and without the per-bin applications)
static int shared_solve(pgpfa_ctx* c, int nb, const double* R, double* Z, const int* skip = nullptr, bool first_apply = true,
                        bool final_apply = true, const int* cols = nullptr, int ncols = 0, int vec_row = 0, bool vec_f32 = false) {
  const int ng = cols ? ncols : nb;                // columns of the multi-RHS products: all slots, or the listed (live) ones
  if (c->plan_lowrank) {
    const long long ld = c->ld;
    const int rpad = c->rpad;
    auto apply_bin = [&](const double* a, const double* b2, double scale, double* o) {
      dispatch_pw(c->p, [&](auto pw) {
        constexpr int PW = decltype(pw)::value;
        if constexpr (PW <= 16) {
          hipLaunchKernelGGL(apply_bin_kernel<PW>, dim3((c->T + 63) / 64, (nb + APPLY_BIN_SLOTS - 1) / APPLY_BIN_SLOTS), dim3(256), 0, c->st, c->Gbar,
                             a, b2, scale, o, ld, c->T, c->p, nb);
        } else if (c->mix_wide && c->p <= 20) {
          hipLaunchKernelGGL(apply_bin_wide2_kernel<20>, dim3((c->T + 63) / 64, (nb + APPLY_BIN_SLOTS - 1) / APPLY_BIN_SLOTS), dim3(256), 0, c->st, c->Gbar, a, b2,
                             scale, o, ld, c->T, c->p, nb, c->sink);
        } else {
          const int bins = wide_bins(c->p);
          hipLaunchKernelGGL(apply_bin_wide_kernel, dim3((c->T + bins - 1) / bins, (nb + APPLY_BIN_SLOTS - 1) / APPLY_BIN_SLOTS), dim3(bins * 32),
                             wide_lds_bytes(c->p, bins, 1), c->st, c->Gbar, a, b2, scale, o, ld, c->T, c->p, nb, bins);
        }
      });
    };
    if (first_apply) apply_bin(R, nullptr, 1.0, c->Xt);
    // the two block-diagonal products as kernels of their own (thin.h) where the matrix cores are in use; the general product otherwise
    const bool thin = c->thin_products && c->mfma && c->T >= 4 && (size_t)c->rpad <= (size_t)c->ld;
    if ((vec_row > 0 || vec_f32) && (!thin || first_apply || final_apply)) return fail("internal: padded vector rows need the thin products and no per-bin application");
    ThinP tp{};
    tp.F = c->Flr; tp.Tf = c->Tp; tp.T = c->T; tp.Tx = vec_row > 0 ? vec_row : c->T; tp.FT = c->FTbig; tp.ldft = c->rpad;
    tp.cols = cols; tp.n_dev = (cols && c->cur_ndev) ? c->cur_ndev : nullptr; tp.ncols = ng; tp.skip = skip;
    auto thin_prof = [&](const char* what) {
      prof_begin(c, TAG_SOLVE, tp.n_dev ? 0.0 : 2.0 * c->T * c->rtot * ng);
      if (c->prof.on && c->prof.open) {
        char key[96];
        if (tp.n_dev) std::snprintf(key, sizeof key, "f64 thin %s T=%d r=%d N=live", what, c->T, c->rtot);
        else std::snprintf(key, sizeof key, "f64 thin %s T=%d r=%d N=%d", what, c->T, c->rtot, ng);
        c->prof.recs.back().shape = key;
      }
    };
    GemmP y{};
    y.skip = skip;                                               // Y = F^T (Gb R)          (rpad x nb)
    y.A = c->FTbig; y.sA = 0; y.lda = rpad; y.B = c->Xt; y.sB = 0; y.ldb = c->ld; y.C = c->Glt; y.sC = 0; y.ldc = c->ld;
    y.M = rpad; y.N = ng; y.K = c->npad; y.cols = cols; y.alpha = 1.0; y.beta = 0.0; y.slots = nullptr; y.nbatch = 1; y.mode = GEMM_FULL; y.kflags = 0;
    y.rtab = c->d_kr_ft; y.ntab = c->ntab_ft; y.k_loop_hint = c->kr_ft_len; y.flops_hint = 2.0 * c->T * c->rtot * ng;      // block-diagonal operand: only T x r_k blocks are non-zero
    if (thin) {
      tp.tab = c->d_thin_ft; tp.X = c->Xt; tp.ldx = c->ld; tp.Y = c->Glt; tp.ldy = c->ld;
      thin_prof("F^T t");
      if (vec_f32) {
        if (c->T % 4 == 0) hipLaunchKernelGGL((thin_ft_kernel<true, float>), dim3(c->nthin_ft, (ng + 15) / 16), dim3(512), 0, c->st, tp);
        else hipLaunchKernelGGL((thin_ft_kernel<false, float>), dim3(c->nthin_ft, (ng + 15) / 16), dim3(512), 0, c->st, tp);
      } else if (c->T % 4 == 0) hipLaunchKernelGGL(thin_ft_kernel<true>, dim3(c->nthin_ft, (ng + 15) / 16), dim3(512), 0, c->st, tp);
      else hipLaunchKernelGGL(thin_ft_kernel<false>, dim3(c->nthin_ft, (ng + 15) / 16), dim3(512), 0, c->st, tp);
      prof_end(c);
    } else {
      CHK(gemm(c, true, y));
    }
    GemmP z{};                                               // Zs = Sb Y
    z.skip = skip;
    z.A = c->sU; z.sA = 0; z.lda = rpad; z.B = c->Glt; z.sB = 0; z.ldb = c->ld; z.C = c->KD; z.sC = 0; z.ldc = c->ld;
    // (K = rtot, a multiple of 16: the row tiles of Y above stop at roff[p] = rtot, rows [rtot, rpad) of c->Glt are never written and may hold
    //  anything - the buffer doubles as the line search's trial gradient and is re-carved from the arena by every re-plan)
    z.M = rpad; z.N = ng; z.K = c->rtot; z.cols = cols; z.alpha = 1.0; z.beta = 0.0; z.slots = nullptr; z.nbatch = 1; z.mode = GEMM_FULL; z.kflags = 0;
    if (thin && c->thin_products >= 2) {
      // the same kernel as F^T t with Sb as the "transposed factor" of one latent with rtot rows and rtot bins (element (m, k) at k rpad + m, as the
      // GEMM reads it)
      ThinP ts = tp;
      ts.FT = c->sU; ts.ldft = rpad; ts.T = c->rtot; ts.tab = c->d_thin_s; ts.X = c->Glt; ts.ldx = c->ld; ts.Y = c->KD; ts.ldy = c->ld;
      prof_begin(c, TAG_SOLVE, tp.n_dev ? 0.0 : 2.0 * (double)c->rtot * c->rtot * ng);
      hipLaunchKernelGGL(thin_ft_kernel<true>, dim3(c->nthin_s, (ng + 15) / 16), dim3(512), 0, c->st, ts);
      prof_end(c);
    } else {
      CHK(gemm(c, true, z));
    }
    GemmP q{};                                               // Q = F Zs                (n x nb)
    q.skip = skip;
    q.A = c->Fbig; q.sA = 0; q.lda = c->ld; q.B = c->KD; q.sB = 0; q.ldb = c->ld; q.C = c->Xt; q.sC = 0; q.ldc = c->ld;
    q.M = c->n; q.N = ng; q.K = rpad; q.cols = cols; q.alpha = 1.0; q.beta = 0.0; q.slots = nullptr; q.nbatch = 1; q.mode = GEMM_FULL; q.kflags = 0;
    q.rtab = c->d_kr_f; q.ntab = c->ntab_f; q.k_loop_hint = c->kr_f_len; q.flops_hint = 2.0 * c->T * c->rtot * ng;
    if (thin) {
      tp.tab = c->d_thin_f; tp.X = c->KD; tp.ldx = c->ld; tp.Y = c->Xt; tp.ldy = c->ld;
      thin_prof("F v");
      if (vec_f32) {
        if (c->T % 4 == 0) hipLaunchKernelGGL((thin_f_kernel<true, float>), dim3(c->nthin_f, (ng + 15) / 16), dim3(256), 0, c->st, tp);
        else hipLaunchKernelGGL((thin_f_kernel<false, float>), dim3(c->nthin_f, (ng + 15) / 16), dim3(256), 0, c->st, tp);
      } else if (c->T % 4 == 0) hipLaunchKernelGGL(thin_f_kernel<true>, dim3(c->nthin_f, (ng + 15) / 16), dim3(256), 0, c->st, tp);
      else hipLaunchKernelGGL(thin_f_kernel<false>, dim3(c->nthin_f, (ng + 15) / 16), dim3(256), 0, c->st, tp);
      prof_end(c);
    } else {
      CHK(gemm(c, true, q));
    }
    if (final_apply) apply_bin(R, c->Xt, c->eps, Z);
    HIPC(hipGetLastError());
    return 0;
  }
  GemmP g{};
  g.skip = skip;
  g.A = c->sU; g.sA = 0; g.lda = c->ld;                      // P^-1, symmetric
  g.B = R; g.sB = 0; g.ldb = c->ld;                          // K x N column-major: slot vectors
  g.C = Z; g.sC = 0; g.ldc = c->ld;
  g.M = c->npad; g.N = ng; g.K = c->npad; g.alpha = 1.0; g.beta = 0.0; g.cols = cols;
  g.slots = nullptr; g.nbatch = 1; g.mode = GEMM_FULL; g.kflags = 0;
  return gemm(c, true, g);
}

// per (slot, bin) blocks G = (I + eps W)^-1, Wt = W G [, log det]: register kernel up to 10 latents, LDS kernel beyond
int bin_blocks(pgpfa_ctx* c, const double* W, long long sW, double* G, double* Wt, long long sO, int nslots, double* ldet) {
  const int T = c->T, p = c->p, pp = p * p;
  const long long items = (long long)nslots * T;
  bool done = false;
  if (p <= 10) {
    dispatch_pw(p, [&](auto pw) {
      constexpr int PW = decltype(pw)::value;
      if constexpr (PW <= 10) {
        hipLaunchKernelGGL(bin_blocks_reg_kernel<PW>, dim3((unsigned)((items + BBR_MPB - 1) / BBR_MPB)), dim3(BBR_TPB), 0, c->st, W, sW, G, Wt, sO, T, p, c->eps,
                           c->ident, nslots, ldet);
        done = true;
      }
    });
  }
  if (!done && p <= 32) {
    // 32 lanes per matrix; as many pairs of matrices (waves) per block as 64 KB of LDS hold, four at most
    const size_t per = bin_blocks_coop_doubles(p) * sizeof(double);
    const int nw = (int)std::max<size_t>(1, std::min<size_t>(4, (64 * 1024) / (2 * per)));
    const int per_block = 2 * nw;
    dispatch_pmax(p, [&](auto pm) {
      constexpr int PM = decltype(pm)::value;
      if constexpr (PM >= 16)
        hipLaunchKernelGGL(bin_blocks_coop_kernel<PM>, dim3((unsigned)((items + per_block - 1) / per_block)), dim3(64 * nw), per_block * per, c->st, W,
                           sW, G, Wt, sO, T, p, c->eps, c->ident, nslots, ldet);
    });
    done = true;
  }
  if (!done) {
    int th = (int)(48 * 1024 / ((2 * pp + 1) * sizeof(double)));
    th = std::max(1, std::min(64, th));
    hipLaunchKernelGGL(bin_blocks_kernel, dim3((unsigned)((items + th - 1) / th)), dim3(th), (size_t)th * (2 * pp + 1) * sizeof(double), c->st, W, sW, G,
                       Wt, sO, T, p, c->eps, c->ident, nslots, ldet);
  }
  HIPC(hipGetLastError());
  return 0;
}

// low-rank form of the shared preconditioner: Gb, Wtb from the mean curvature, Sb = (I + F^T Wtb F)^-1 (r x r, one slot)
static int shared_factor_lowrank(pgpfa_ctx* c, int nb) {
  const int T = c->T, p = c->p, pp = p * p, len = T * pp, rpad = c->rpad;
  hipLaunchKernelGGL(mean_w_kernel, dim3((len + 255) / 256), dim3(256), 0, c->st, c->W, (long long)len, c->ident, nb, len, c->Wbar);
  CHK(bin_blocks(c, c->Wbar, 0LL, c->Gbar, c->Wtbar, 0LL, 1, nullptr));
  {
    const int npk = T * (p * (p + 1) / 2);
    hipLaunchKernelGGL(pack_sym_t_kernel<double>, dim3((npk + 255) / 256), dim3(256), 0, c->st, (const double*)c->Gbar, c->GbT, T, p);
    hipLaunchKernelGGL(pack_sym_t_kernel<float>, dim3((npk + 255) / 256), dim3(256), 0, c->st, (const double*)c->Wbar, reinterpret_cast<float*>(c->WbT), T, p);
  }
  CholWS lw = c->sws;
  lw.ld = rpad; lw.npad = rpad; lw.nact = round_up(c->rtot, 64);
  const int nblk64 = c->rank_compact ? round_up(c->rtot16, 64) / 64 : rpad / 64, npairs = nblk64 * (nblk64 + 1) / 2;
  const int* cmap = c->rank_compact ? c->d_cmap : nullptr;
  hipLaunchKernelGGL(assemble_b_kernel_t<double>, dim3(npairs, 1), dim3(256), 0, c->st, lw.H, (long long)lw.sH, rpad, nblk64, (const double*)c->Flr, c->Tp, T, p, c->d_blk_lat,
                     c->d_blk_col, c->Wtbar, 0LL, c->ident, 1, cmap);
  if (cmap && rpad > c->rtot)
    hipLaunchKernelGGL(pad_identity_kernel<double>, dim3((rpad + 3) / 4, 1), dim3(256), 0, c->st, lw.H, (long long)lw.sH, rpad, c->rtot, rpad, lw.nact, (int)NB, c->ident);
  HIPC(hipGetLastError());
  HIPC(hipMemsetAsync(c->sws.info, 0, sizeof(int), c->st));
  CHK(factor(c, lw, nullptr, 1));
  HIPC(hipMemsetAsync(lw.Mt, 0, (size_t)rpad * rpad * sizeof(double), c->st));
  CHK(inverse_t(c, lw, nullptr, 1));
  GemmP g{};
  g.A = lw.Mt; g.sA = 0; g.lda = rpad; g.B = lw.Mt; g.sB = 0; g.ldb = rpad;
  g.C = c->sU; g.sC = 0; g.ldc = rpad;
  g.M = rpad; g.N = rpad; g.K = rpad; g.alpha = 1.0; g.beta = 0.0;
  g.slots = nullptr; g.nbatch = 1; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
  CHK(gemm(c, false, g));
  int info = 0;
  CHK(dl_enqueue(c, &info, c->sws.info, sizeof(int)));
  CHK(dl_flush(c));
  if (info != 0) return fail("shared low-rank preconditioner not positive definite (pivot %d)", info);
  return 0;
}

// explicit inverse of the mean-trial Hessian  P = Kinv + scatter(mean_r W_r[t])  of the slots [0,nb)
static int shared_factor(pgpfa_ctx* c, int nb) {
  if (c->plan_lowrank) return shared_factor_lowrank(c, nb);
  const int len = c->T * c->p * c->p;
  hipLaunchKernelGGL(mean_w_kernel, dim3((len + 255) / 256), dim3(256), 0, c->st, c->W, (long long)len, c->ident, nb, len, c->Wbar);
  hipLaunchKernelGGL(assemble_h_kernel, dim3(c->npad, 1), dim3(256), 0, c->st, c->sws.H, c->sws.sH, c->ld, c->npad, c->n, c->T, c->Tp, c->p,
                     c->Kinv, c->Wbar, 0LL, c->ident, 1.0);
  HIPC(hipMemsetAsync(c->sws.info, 0, sizeof(int), c->st));
  CHK(factor(c, c->sws, nullptr, 1));
  CHK(inverse_t(c, c->sws, nullptr, 1));
  GemmP g{};
  g.A = c->sws.Mt; g.sA = 0; g.lda = c->ld; g.B = c->sws.Mt; g.sB = 0; g.ldb = c->ld;
  g.C = c->sU; g.sC = 0; g.ldc = c->ld;
  g.M = c->npad; g.N = c->npad; g.K = c->npad; g.alpha = 1.0; g.beta = 0.0;
  g.slots = nullptr; g.nbatch = 1; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
  CHK(gemm(c, false, g));
  int info = 0;
  CHK(dl_enqueue(c, &info, c->sws.info, sizeof(int)));
  CHK(dl_flush(c));
  HIPC(hipGetLastError());
  if (info != 0) return fail("shared preconditioner not positive definite (pivot %d)", info);
  return 0;
}

// H (from the W blocks of slots [0,nb), diagonal scaled by diag_scale) -> factor -> L^-T -> post_vsmGP and
// post_vsm of the trials bound to the slots.  Shared by the Laplace and the dual-variational E-step.

// The reference's negative log-posterior (inference.py:12-32) of a slot's trial AT x = 0: f0 = sum_n (T exp(d_n) - d_n sum_t y_nt), the objective of
// the cold start - what a warm start has to beat (estep_impl: a start point that does worse is replaced by zero).  One pass over the slot's count
// rows (q T bytes).  grid = (slots), block = 256 (a wave per neuron row); mask (may be null): the neuron a leave-one-out item leaves out.
static __global__ __launch_bounds__(256) void cold_objective_kernel(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ Yhi, const double* __restrict__ d,
                                                                    const int* __restrict__ trial_of_slot, const int* __restrict__ mask, int q, int T,
                                                                    double* __restrict__ f0) {
  __shared__ double red[4];
  const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = (size_t)trial_of_slot[s] * q * T;
  const int skip = mask ? mask[s] : -1;
  double acc = 0.0;
  for (int n = wave; n < q; n += 4) {
    if (n == skip) continue;
    unsigned cnt = 0;
    const uint8_t* row = Y + base + (size_t)n * T;
    for (int t = lane; t < T; t += 64) cnt += row[t];
    if (Yhi) {
      const uint8_t* rh = Yhi + base + (size_t)n * T;
      for (int t = lane; t < T; t += 64) cnt += (unsigned)rh[t] << 8;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if (lane == 0) acc += (double)T * exp(d[n]) - d[n] * (double)cnt;
  }
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) f0[s] = (red[0] + red[1]) + (red[2] + red[3]);
}

// X[slot][0 .. n) = 0 for the listed slots.  grid = (ceil(n / 256), listed slots)
static __global__ void zero_rows_kernel(double* __restrict__ X, long long ld, int n, const int* __restrict__ list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) X[(size_t)list[blockIdx.y] * ld + i] = 0.0;
}

int estep_impl(pgpfa_ctx* c, const Trials& tr, int warm_start, bool allow_lr, double* obj_sum, int32_t* iters, int32_t* status,
                      const LooJob* loo, const VarJob* var) {
  PhaseRange range_estep(var ? "pgpfa.dual_fixed_point" : loo ? "pgpfa.loo_mode_search" : "pgpfa.estep_laplace");
  c->want_slots = std::max(c->want_slots, std::min((int)tr.v.size(), c->R));
  CHK(ready_estep(c, allow_lr));
  struct MaskGuard { pgpfa_ctx* c; ~MaskGuard() { c->mask_active = false; c->var_active = false; c->lam_out_active = false; } } mask_guard{c};
  if (var) CHK(ensure_lambda(c));
  const int N = (int)tr.v.size();
  const auto t_begin = std::chrono::steady_clock::now();
  const int nvec = c->n, p = c->p, T = c->T;
  const long long ld = c->ld;
  double total = 0.0;
  double n_fact = 0.0, n_solve = 0.0, n_pcg = 0.0, n_shared = 0.0;
  double newton_bytes = 0.0;                          // mandatory HBM bytes of the inner PCG iterations run (see below)
  double newton_bytes_moved = 0.0;                    // what the kernels of the form in use really move per slot-iteration (single-precision vectors counted as such)
  double newton_bytes_survey = 0.0;                   // the same slot-iterations priced by SURVEY 8(d)'s B_E = q T s_y + 8 (2 p T + T p^2) per pass per trial
  std::vector<std::pair<hipEvent_t, hipEvent_t>> newton_ev;   // events around every inner solve (the Newton-solve kernels)
  int max_it_seen = 0;
  double n_cold = 0.0;                                          // warm starts replaced by zero (start_guard)
  double n_fb_dir = 0.0, n_fb_search = 0.0, n_fb_cap = 0.0;   // slots the shared-preconditioner phase gave up on: no descent direction, line search exhausted, outer cap
  std::vector<double> f(c->B), qxx(c->B), qdx(c->B), qdd(c->B), dec(c->B), smax(c->B), alpha(c->B), ftry(c->B);
  std::vector<int> its(c->B), stat(c->B), info(c->B);

  for (int c0 = 0; c0 < N; c0 += c->B) {
    const int nb = std::min(c->B, N - c0);
    std::vector<int> tos(tr.v.begin() + c0, tr.v.begin() + c0 + nb);
    CHK(upload_list(c, c->trial_of_slot, tos));
    if (loo) {
      std::vector<int> mk(loo->mask->begin() + c0, loo->mask->begin() + c0 + nb);
      CHK(upload_list(c, c->mask_of_slot, mk));
      c->mask_active = true;
    }
    HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int) * nb, c->st));
    const size_t mlam = (size_t)c->q * T;
    std::vector<double> vdelta(nb, 0.0), vdelta_prev(nb, -1.0), vdamp(nb, 1.0);
    std::vector<int> vstat(nb, 1), vouter(nb, 0), vslow(nb, 0);
    std::vector<int> how(nb, 0);                  // start point of a slot: 0 cold (zero), 1 the resident mode, 2 its extrapolation
    bool any_warm = false;
    // (error paths of the fixed point: the passes have overwritten the per-bin blocks of the chunk's trials - whatever posterior they had is gone)
    auto var_superseded = [&]() {
      for (int t : tos) { c->trial_dual[t] = 0; c->trial_snap[t] = -1; c->vsmgp_ok[t] = 0; c->lam_resident[t] = 0; c->lam_valid[t] = 0; }
    };
    if (var) {
      // lambda of the chunk -> W = C^T diag(lambda) C (+ the reference's jitter), the covariance blocks and from them the first offsets;
      // start point of the mode search: the variational mean of that lambda, -K C_big (lambda - y) (inference.py:194)
      // (exp / log of the q T entries of every trial run on the device: on the host they were 1.3e8 libm calls per 256 config-5 trials - half a
      // second each way, more than the whole fixed point)
      if (var->start == 3) {
        for (int s = 0; s < nb; ++s) CHK(copy_dev(c, c->lamd + (size_t)s * mlam, c->lam_keep + (size_t)tos[s] * mlam, mlam * sizeof(double)));
      } else if (var->start != 0) {
        CHK(upload(c, c->lamd, var->rho + (size_t)c0 * mlam, (size_t)nb * mlam));
      }
      int* bad_dev = reinterpret_cast<int*>(c->pcg_ratio);              // (scratch word: no inner solve is running)
      HIPC(hipMemsetAsync(bad_dev, 0, sizeof(int), c->st));
      if (var->start != 3) hipLaunchKernelGGL(var_exp_kernel, dim3(2048), dim3(256), 0, c->st, c->lamd, (size_t)nb * mlam, var->start == 0 ? 1 : 0, 0.5, bad_dev);
      {
        int bad = 0;
        CHK(dl_enqueue(c, &bad, bad_dev, sizeof(int)));
        CHK(dl_flush(c));
        if (bad) return fail("rho must be finite with a positive finite exp (trials %d..%d)", tos.front(), tos.back());
      }
      std::vector<double> sB_, sD_, vKv_;
      CHK(dual_common(c, nb, &sB_, &sD_, &vKv_));
      // (only when lambda is a previous optimum: from a cold lambda that mean is far out - hundreds in the log rate - and zero is the safe start)
      if (var->start >= 2) hipLaunchKernelGGL(negate_rows_kernel, dim3((nvec + 255) / 256, nb), dim3(256), 0, c->st, c->KD, ld, c->Xc, ld, nvec, c->ident);
      else HIPC(hipMemsetAsync(c->Xc, 0, (size_t)ld * nb * sizeof(double), c->st));
      if (c->plan_lowrank) { CHK(dual_jitter(c, nb)); CHK(posterior_blocks(c, nb, 1.0, false, false)); }
      else CHK(posterior_blocks(c, nb, 1.0 + 1e-6, false));
      CHK(var_offsets(c, nb, c->voff));
      c->var_active = true;
    } else {
      // start points: cold (zero), the resident mode, or its extrapolation; warm_start = 2 takes the resident mode only
      // for trials some earlier E-step has produced one for (minibatches revisiting trials) and starts the others cold
      bool any = false;
      // (the difference of the last two modes is only a prediction while the parameters keep their pace: ctx.h, extrapolate_guard)
      const bool trend_ok = c->extrapolate && (c->extrapolate_guard <= 0.0 || (c->par_step > 0.0 && c->par_step_prev >= 0.0 &&
                                                                                 c->par_step_prev <= c->extrapolate_guard * c->par_step));
      for (int s = 0; s < nb && warm_start; ++s) {
        const int tr_ = tos[s];
        if (warm_start == 2 && c->mode_serial[tr_] < 0) continue;
        how[s] = 1;
        if (trend_ok && c->mode_serial[tr_] == c->estep_serial - 1 && c->prev_serial[tr_] == c->estep_serial - 2) how[s] = 2;
        any = true;
      }
      any_warm = any;
      if (any) {
        CHK(upload_list(c, c->list_a, how));
        hipLaunchKernelGGL(gather_start_kernel, dim3((nvec + 255) / 256, nb), dim3(256), 0, c->st, c->Xmode, c->Xprev, nvec, c->Xc, ld,
                           c->trial_of_slot, c->list_a, c->extrapolate_beta);
      } else {
        hipLaunchKernelGGL(gather_rows_kernel, dim3((nvec + 255) / 256, nb), dim3(256), 0, c->st, c->Xmode, nvec, c->Xc, ld, c->trial_of_slot, 1);
      }
    }
    std::vector<int> active;
    for (int vo = 0;; ++vo) {                     // (one pass for the Laplace E-step; the variational fixed point comes back here with new offsets)
    // objective, gradient pieces and curvature blocks at the start point
    auto eval_start = [&]() -> int {
      CHK(prior_mv_all(c, nb, c->Xc, c->KX));
      hipLaunchKernelGGL(dots3_kernel, dim3(nb), dim3(256), 0, c->st, c->Xc, ld, c->KX, ld, (const double*)nullptr, 0LL,
                         (const double*)nullptr, 0LL, nvec, c->ident, c->sc_qxx, c->sc_qdx, c->sc_qdd);
      CHK(poisson(c, c->ident, nb, c->Xc, c->Gl, c->W, c->sc_f, 1));
      CHK(dl_enqueue(c, f.data(), c->sc_f, nb * sizeof(double)));
      CHK(download(c, qxx.data(), c->sc_qxx, nb));
      for (int s = 0; s < nb; ++s) f[s] += 0.5 * qxx[s];
      return 0;
    };
    if (vo == 0 && !var && any_warm && c->start_guard) {
      // (enqueued ahead of the evaluation: its result comes back with that one's read-back)
      hipLaunchKernelGGL(cold_objective_kernel, dim3(nb), dim3(256), 0, c->st, c->Y, c->Yhi, c->d, c->trial_of_slot,
                         c->mask_active ? c->mask_of_slot : (const int*)nullptr, c->q, T, c->sc_alpha);
      CHK(dl_enqueue(c, ftry.data(), c->sc_alpha, nb * sizeof(double)));
    }
    CHK(eval_start());
    if (vo == 0 && !var && any_warm && c->start_guard) {
      // A warm start that does worse than x = 0 is no start: after a jump of the parameters (another fold's fit, the generating parameters, a second
      // fit in one process) the resident modes belong to other loadings - log rates of +-100, a curvature the single-precision copies cannot hold -
      // and 931 of 1024 trials went through the dense per-trial retry (1.8 s where a cold E-step takes 0.12: tools/jump_probe.py).  Those slots
      // restart at zero; the objective is strictly convex, the start point never changes the mode.
      std::vector<int> cold;
      for (int s = 0; s < nb; ++s)
        if (how[s] != 0 && !(f[s] <= ftry[s])) cold.push_back(s);
      if (!cold.empty()) {
        CHK(upload_list(c, c->list_b, cold));
        hipLaunchKernelGGL(zero_rows_kernel, dim3((nvec + 255) / 256, (unsigned)cold.size()), dim3(256), 0, c->st, c->Xc, ld, nvec, c->list_b);
        CHK(eval_start());
        n_cold += (double)cold.size();
      }
    }
    active.clear();
    for (int s = 0; s < nb; ++s) {
      if (vo == 0) its[s] = 0;
      if (var && vstat[s] != 1) continue;         // (this slot's fixed point is settled)
      active.push_back(s);
      stat[s] = 1;
    }
    std::vector<int> leftovers;
    // (small chunks are launch-latency bound: there the extra packing / check launches of the host-free form cost more than
    // the round trips they remove - measured at config 2: 8.6 vs 8.0 ms per E-step)
    const bool thin_ok = c->thin_products && c->mfma && T >= 4 && (size_t)c->rpad <= (size_t)c->ld;
    // (11 .. 20 latents: only the round-5 form of the host-free step exists - pcgw_*_kernel - and it needs the thin products)
    const bool wide_ok = p <= 20 && c->pcg_form >= 2 && thin_ok && c->pcg_w32 && c->pcg_retire && c->pcg_blk != nullptr;
    const bool fused = c->pcg_fused && c->plan_lowrank && (p <= 16 || wide_ok) && c->h_pcg != nullptr && (c->pcg_fused == 2 || (double)nb * c->n >= 1.0e6);
    // form of the host-free iteration (pcg.h): the two-kernel step without the prior mat-vec needs the packed FP32 curvature and per-slot
    // retirement; otherwise the split kernels of round 3
    const bool onek = fused && c->pcg_w32 && c->pcg_retire && c->pcg_form != 0 && (p <= 10 || wide_ok);
    // round 5 (pcg_form = 2): the solve's private vectors on line-aligned latent rows, ONE start kernel (gradient, residual, zero step, first
    // per-bin application), the step's closing inside kernel A, ONE upload (control block, list, forcing terms); needs the thin products
    const bool f2 = onek && c->pcg_form >= 2 && thin_ok && c->pcg_blk != nullptr;
    const int TB = p <= 10 ? 64 : PCGW_TB;               // bins per workgroup tile of the step's per-bin kernels
    // ... and (second half of round 5) z, s, p, q, t / y of the solve stored in single precision (PcgCgP::vec32)
    const bool v32 = f2 && c->pcg_vec32;
    const int Tl = !f2 ? T : (v32 && (long long)p * round_up(T, 32) <= ld) ? round_up(T, 32) : ((long long)p * round_up(T, 16) <= ld) ? round_up(T, 16) : T;
    // ... and (round 6) the residual and the step as well, up to 10 latents (PcgCgP::X32): x in the free second half of Z's buffer
    const bool rx32 = v32 && c->pcg_rx32 && p <= 10;
    auto with_tv = [&](auto&& fn) { if (rx32) fn(float{}, float{}); else if (v32) fn(float{}, double{}); else fn(double{}, double{}); };
    // packed single-precision curvature of the two-kernel step (pcg.h): component-major [c][Tw], rows on 128-byte lines.  It is written where W is:
    // by pack_w32t_kernel at the start of a solve for slots whose W came from the Poisson pass at the E-step's start point, and by the commit of an
    // accepted step (commit_w_pack_kernel: the copy W <- Wt and the packing in one pass over Wt - a solve after the first finds every active slot packed)
    const int npk = p * (p + 1) / 2;
    const int Tw = round_up(T, 32);
    const long long sW32 = (long long)Tw * npk;
    std::vector<char> w32_ok(nb, 0);

    // backtracking line search along Dl for the slots in `cand` (objective with rounding-noise slack as in
    // the oracle); needs dec/qxx/qdx/qdd of those slots on the host.  Accepted slots are committed
    // (X, K^-1 x, likelihood gradient, W); returns the slots whose search was exhausted.
    auto line_search = [&](const std::vector<int>& cand, std::vector<int>* failed) -> int {
      std::vector<int> pending;
      for (int s : cand) { alpha[s] = 1.0; pending.push_back(s); }
      for (int ls = 0; ls < 40 && !pending.empty(); ++ls) {
        const int np_ = (int)pending.size();
        CHK(upload_nosync(c, c->list_b, pending.data(), sizeof(int) * pending.size()));
        CHK(upload_nosync(c, c->sc_alpha, alpha.data(), sizeof(double) * nb));
        hipLaunchKernelGGL(make_try_kernel, dim3((nvec + 255) / 256, np_), dim3(256), 0, c->st, c->Xc, ld, c->Dl, ld, c->sc_alpha, c->Xt, ld, nvec,
                           c->list_b);
        CHK(poisson(c, c->list_b, np_, c->Xt, c->Glt, c->Wt, c->sc_f, 1));
        CHK(download(c, ftry.data(), c->sc_f, nb));
        std::vector<int> acc, rej;
        for (int s : pending) {
          const double a = alpha[s];
          const double ft = ftry[s] + 0.5 * (qxx[s] + 2.0 * a * qdx[s] + a * a * qdd[s]);
          const double slack = 1e-12 * (1.0 + std::fabs(f[s]));
          if (std::isfinite(ft) && ft <= f[s] - 1e-4 * a * dec[s] + slack) {
            f[s] = ft;
            acc.push_back(s);
          } else {
            alpha[s] = 0.5 * a;
            rej.push_back(s);
          }
        }
        if (!acc.empty()) {
          const int nacc = (int)acc.size();
          CHK(upload_nosync(c, c->list_b, acc.data(), sizeof(int) * acc.size()));
          const int nw = T * p * p;
          hipLaunchKernelGGL(commit_kernel, dim3((nvec + 255) / 256, nacc), dim3(256), 0, c->st, c->Xc, c->Xt, c->KX, c->KD, c->Gl, c->Glt, ld,
                             c->W, c->Wt, (long long)nw, c->sc_alpha, nvec, onek ? 0 : nw, c->list_b);
          if (onek) {
            hipLaunchKernelGGL(commit_w_pack_kernel, dim3((T + 63) / 64, nacc), dim3(256), (size_t)npk * 65 * sizeof(float), c->st, (const double*)c->Wt, c->W,
                               (long long)nw, c->W32, sW32, Tw, T, p, c->list_b);
            for (int s : acc) w32_ok[s] = 1;
          }
          HIPC(hipGetLastError());
        }
        pending.swap(rej);
      }
      *failed = pending;
      return 0;
    };

    // ---- phase 1: inexact Newton, all slots in lockstep, PCG on H_r delta = -g preconditioned by ONE shared factor
    // (the mean-trial Hessian: cond(P^-1 H_r) stays below ~4, measured).  Every preconditioner application is two
    // multi-RHS triangular sweeps run as GEMMs over the slots; no per-trial factorisation in this phase.
    if (c->shared_pcg && (nb >= c->shared_min || c->plan_lowrank)) {
      PhaseRange range_newton("pgpfa.newton_pcg");
      CHK(shared_factor(c, nb));
      n_shared += 1;
      std::vector<double> rr(nb), rr0(nb), err_pred(nb, -1.0);
      for (int outer = 0; outer < c->pcg_outer_max && !active.empty(); ++outer) {
        // forcing term of this outer iteration (relative residual the inner solve is run to).  With e the predicted
        // error of a slot's current iterate, solving beyond eta ~ e buys nothing (the Newton step itself leaves ~e^2),
        // and when a looser solve already lands below the stopping tolerance that looser value is enough.
        double eta_target = c->pcg_eta0;
        for (int s : active) {
          if (err_pred[s] < 0.0) continue;                       // first outer iteration of this slot
          const double e = std::max(err_pred[s], 1e-300);
          const double want = std::max(e, c->chord_xtol / (20.0 * e));
          eta_target = std::min(eta_target, std::max(1e-9, std::min(c->pcg_eta0, want)));
        }
        const int na = (int)active.size();
        int done_inner = 0;
        PcgCtl& fused_ctl = c->fused_ctl_host;             // (context member: a queued read-back must not point into this frame)
        fused_ctl = PcgCtl{};
        if (!f2) CHK(upload_nosync(c, c->list_a, active.data(), sizeof(int) * active.size()));
        if (c->time_newton) {
          newton_ev.emplace_back(prof_event(c->prof), prof_event(c->prof));
          hipEventRecord(newton_ev.back().first, c->st);
        }
        if (!f2) {
          hipLaunchKernelGGL(grad_total_kernel, dim3((nvec + 255) / 256, na), dim3(256), 0, c->st, c->Gl, ld, c->KX, ld, c->Gt, ld, nvec, c->list_a);
          hipLaunchKernelGGL(pcg_init_kernel, dim3((c->npad + 255) / 256, na), dim3(256), 0, c->st, c->Gt, c->Rv, c->Dl, ld, nvec, c->npad, c->list_a);
        }
        if (onek) {
          // ---- inner solve: per step pcg_cg_a_kernel, pcg_cg_b_kernel, the closing kernel and the three preconditioner products (pcg.h)
          const int* skip = &c->pcgctl->stop;
          {
            std::vector<float> eta_s(nb, (float)eta_target);
            for (int s : active) {
              double es = c->pcg_eta0;
              if (err_pred[s] >= 0.0) {
                const double e = std::max(err_pred[s], 1e-300);
                es = std::max(1e-9, std::min(c->pcg_eta0, std::max(e, c->chord_xtol / (20.0 * e))));
              }
              eta_s[s] = (float)es;
            }
            PcgCtl h0{};
            h0.nlive = na; h0.nl[0] = na;
            if (f2) {
              // one upload: [control block (16 words) | list_a | first live list | forcing terms] - consecutive pieces of c->pcg_blk
              const size_t nB = (size_t)c->B;
              std::vector<int> img(16 + 3 * nB, 0);
              std::memcpy(img.data(), &h0, sizeof(PcgCtl));
              std::memcpy(img.data() + 16, active.data(), sizeof(int) * na);
              std::memcpy(img.data() + 16 + nB, active.data(), sizeof(int) * na);
              std::memcpy(img.data() + 16 + 2 * nB, eta_s.data(), sizeof(float) * nb);
              CHK(upload_nosync(c, c->pcg_blk, img.data(), img.size() * sizeof(int)));
            } else {
              CHK(upload_nosync(c, c->pcg_eta, eta_s.data(), sizeof(float) * nb));
              CHK(copy_dev(c, c->live, c->list_a, sizeof(int) * na));
              CHK(upload_nosync(c, c->pcgctl, &h0, sizeof(PcgCtl)));
            }
          }
          c->h_pcg[0] = 0; c->h_pcg[1] = 0; c->h_pcg[2] = na;
          PcgCgP cp{};
          cp.GbT = c->GbT; cp.WbT = reinterpret_cast<const float*>(c->WbT); cp.W32T = c->W32; cp.sW32 = sW32; cp.Tw = Tw;
          cp.X = c->Dl; cp.R = c->Rv; cp.P = c->Pv; cp.Q = c->Qv; cp.Z = c->Zv; cp.S = c->Sv; cp.Y = c->Xt; cp.sV = ld;
          cp.part = c->sc_part2; cp.gam = c->cg_scal; cp.alp = c->cg_scal + 2 * (size_t)c->B; cp.rr = c->sc_rr; cp.rr0 = c->sc_rr0; cp.eta = c->pcg_eta;
          cp.ctl = c->pcgctl; cp.live0 = c->live; cp.live1 = c->live1;
          cp.eps = c->eps; cp.T = T; cp.p = p; cp.inner_min = c->pcg_inner_min; cp.ntile = (T + TB - 1) / TB; cp.B = c->B; cp.xcd_map = c->pcg_xcd;
          cp.X32 = reinterpret_cast<float*>(c->Zv) + (size_t)ld * ((size_t)c->B + 128);     // (the single-precision z fills the first half of Zv)
          cp.Tl = Tl; cp.Tx = T; cp.vec32 = v32 ? 1 : 0; cp.fold_close = f2 ? 1 : 0; cp.host = (volatile int*)c->d_hpcg; cp.Gl = c->Gl; cp.KX = c->KX; cp.Gt = c->Gt;
          auto cg_grid = [&](int bound) {
            cp.spw = !c->pcg_adapt ? PCG_SLOTS : bound > 640 ? 16 : bound > 320 ? 8 : 4;
            if (TB != 64) cp.spw = std::max(cp.spw, PCGW_SL);                                     // (a 32-bin workgroup has 8 slots in flight)
            return dim3((T + TB - 1) / TB, round_up((bound + cp.spw - 1) / cp.spw, 8));           // (slot groups in blocks of 8: pcg_cg_wg)
          };
          if (f2) {
            const dim3 g0 = cg_grid(na);
            dispatch_pw(p, [&](auto pw) {
              constexpr int PW = decltype(pw)::value;
              if constexpr (PW <= 10) {
                const size_t lb = pcg_cg_b_lds(PW);
                with_tv([&](auto tv, auto tx) {
                  using TV = decltype(tv); using TX = decltype(tx); (void)sizeof(TX);
                  if (lb > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_cg_start_kernel<PW, TV, TX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                  hipLaunchKernelGGL((pcg_cg_start_kernel<PW, TV, TX>), g0, dim3(256), lb, c->st, cp);
                });
              } else if constexpr (PW <= 20) {
                const size_t lb = pcgw_b_lds(PW);
                with_tv([&](auto tv, auto tx) {
                  using TV = decltype(tv); using TX = decltype(tx); (void)sizeof(TX);
                  if (lb > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcgw_start_kernel<PW, TV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                  hipLaunchKernelGGL((pcgw_start_kernel<PW, TV>), g0, dim3(256), lb, c->st, cp);
                });
              }
            });
          }
          {
            std::vector<int> need;
            for (int s : active) if (!w32_ok[s]) { need.push_back(s); w32_ok[s] = 1; }
            if (!need.empty()) {
              CHK(upload_nosync(c, c->list_b, need.data(), sizeof(int) * need.size()));
              hipLaunchKernelGGL(pack_w32t_kernel, dim3((T + 63) / 64, (unsigned)need.size()), dim3(256), (size_t)npk * 65 * sizeof(float), c->st, c->W,
                                 (long long)T * p * p, c->W32, sW32, Tw, T, p, c->list_b);
            }
          }
          // t = Gb r0 (the start kernel has it already), then y = F Sb F^T t over the listed columns (left in c->Xt)
          CHK(shared_solve(c, nb, c->Rv, c->Zv, nullptr, !f2, false, c->list_a, na, f2 ? Tl : 0, v32));
          struct NdevGuard { pgpfa_ctx* c; ~NdevGuard() { c->cur_ndev = nullptr; } } ndev_guard{c};
          c->live_gemms.clear();
          int last_step = -1;
          for (int it = 0; it < c->pcg_inner_max; ++it) {
            cp.par = it & 1; cp.first = (it == 0) ? 1 : 0; cp.step = it; last_step = it;
            // The launches of a step are sized by the live count the closing kernel last mirrored to the host (it only falls during a solve, so a
            // value that is a step or two old is an upper bound; the kernels read the true count on the device).  Few live slots: fewer slots per
            // workgroup, so that the per-bin kernels still offer every CU a workgroup and a wave walks one slot instead of four in a row.
            const int seen = *(volatile int*)&c->h_pcg[2];
            const int bound = c->pcg_adapt ? std::max(1, std::min(na, seen)) : na;
            const dim3 gcg = cg_grid(bound);
            dispatch_pw(p, [&](auto pw) {
              constexpr int PW = decltype(pw)::value;
              if constexpr (PW <= 10) {
                const size_t la = pcg_cg_a_lds(PW), lb = pcg_cg_b_lds(PW);
                // (per launch, not once per process: contexts of one process may sit on different devices)
                with_tv([&](auto tv, auto tx) {
                  using TV = decltype(tv); using TX = decltype(tx); (void)sizeof(TX);
                  if (la > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_cg_a_kernel<PW, TV, TX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)la);
                  if (lb > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_cg_b_kernel<PW, TV, TX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                  hipLaunchKernelGGL((pcg_cg_a_kernel<PW, TV, TX>), gcg, dim3(256), la, c->st, cp);
                  hipLaunchKernelGGL((pcg_cg_b_kernel<PW, TV, TX>), gcg, dim3(256), lb, c->st, cp);
                });
              } else if constexpr (PW <= 20) {
                const size_t la = pcgw_a_lds(PW), lb = pcgw_b_lds(PW);
                with_tv([&](auto tv, auto tx) {
                  using TV = decltype(tv); using TX = decltype(tx); (void)sizeof(TX);
                  if (la > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcgw_a_kernel<PW, TV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)la);
                  if (lb > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcgw_b_kernel<PW, TV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                  hipLaunchKernelGGL((pcgw_a_kernel<PW, TV>), gcg, dim3(256), la, c->st, cp);
                  hipLaunchKernelGGL((pcgw_b_kernel<PW, TV>), gcg, dim3(256), lb, c->st, cp);
                });
              }
            });
            if (!f2) hipLaunchKernelGGL(pcg_iter_close_kernel, dim3(1), dim3(64), 0, c->st, c->pcgctl, it & 1, (volatile int*)c->d_hpcg, -1);
            // the preconditioner products for the NEXT iteration run over the list this launch has just written
            c->live_gemm_collect = (it == 0);
            c->cur_ndev = &c->pcgctl->nl[(it & 1) ^ 1];
            CHK(shared_solve(c, nb, c->Rv, c->Zv, skip, false, false, (it & 1) ? c->live : c->live1, bound, f2 ? Tl : 0, v32));
            if (*(volatile int*)&c->h_pcg[0]) break;           // the device has already stopped: whatever is enqueued is a no-op
            if (it + 1 < c->pcg_inner_max) {
              const auto t_spin = std::chrono::steady_clock::now();
              while (!*(volatile int*)&c->h_pcg[0] && (it + 1) - *(volatile int*)&c->h_pcg[1] > 2) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 5.0) break;   // (never hang on a lost flag)
              }
              if (*(volatile int*)&c->h_pcg[0]) break;
            }
          }
          // (pcg_form 2: kernel A of step i + 1 closes step i; the last step enqueued is closed here unless the solve had stopped before it)
          if (f2 && last_step >= 0) hipLaunchKernelGGL(pcg_iter_close_kernel, dim3(1), dim3(64), 0, c->st, c->pcgctl, last_step & 1, (volatile int*)c->d_hpcg, last_step);
          c->cur_ndev = nullptr;
          c->live_gemm_collect = false;
          CHK(dl_enqueue(c, &fused_ctl, c->pcgctl, sizeof(PcgCtl)));
          HIPC(hipGetLastError());
          done_inner = -1;
        } else if (fused) {
          // ---- inner solve without host round trips (pcg.h): the stopping test runs on the device, iterations are enqueued
          // ahead, kernels of iterations past the stop return at once
          const int ntile = (T + 63) / 64;
          const int* skip = &c->pcgctl->stop;
          const long long sW32 = (long long)T * (p * (p + 1) / 2);
          // the live list starts as the active list; every slot carries its own forcing term (with pcg_retire = 0: the common one)
          {
            std::vector<float> eta_s(nb, (float)eta_target);
            if (c->pcg_retire)
              for (int s : active) {
                double es = c->pcg_eta0;
                if (err_pred[s] >= 0.0) {
                  const double e = std::max(err_pred[s], 1e-300);
                  es = std::max(1e-9, std::min(c->pcg_eta0, std::max(e, c->chord_xtol / (20.0 * e))));
                }
                eta_s[s] = (float)es;
              }
            CHK(upload_nosync(c, c->pcg_eta, eta_s.data(), sizeof(float) * nb));
            CHK(copy_dev(c, c->live, c->list_a, sizeof(int) * na));
            PcgCtl h0{};
            h0.nlive = na;
            CHK(upload_nosync(c, c->pcgctl, &h0, sizeof(PcgCtl)));
          }
          c->h_pcg[0] = 0; c->h_pcg[1] = 0;
          if (c->pcg_w32)
            hipLaunchKernelGGL(pack_w32_kernel, dim3((unsigned)((sW32 + 255) / 256), na), dim3(256), 0, c->st, c->W, (long long)T * p * p, c->W32,
                               sW32, T, p, c->list_a);
          const dim3 gbin(ntile, (na + PCG_SLOTS - 1) / PCG_SLOTS);
          // z0 = P^-1 r0, p0 = z0
          CHK(shared_solve(c, nb, c->Rv, c->Zv, nullptr, true, false, c->list_a, na));
          dispatch_pw(p, [&](auto pw) {
            constexpr int PW = decltype(pw)::value;
            if constexpr (PW <= 16)
              hipLaunchKernelGGL(pcg_apply2_dots_kernel<PW>, gbin, dim3(256), 0, c->st, c->Gbar, c->Rv, c->Xt, c->eps, c->Zv, ld, T, p, c->list_a, na,
                                 c->sc_part2, (const int*)nullptr, (const PcgCtl*)nullptr);
          });
          hipLaunchKernelGGL(pcg_update_p2_kernel, dim3(na), dim3(256), 0, c->st, c->Zv, c->Pv, ld, nvec, c->list_a, c->sc_part2, ntile, c->sc_rz,
                             c->sc_rr, c->sc_rr0, 1, (PcgCtl*)nullptr, (float*)nullptr);
          c->cur_ndev = &c->pcgctl->nlive;
          struct NdevGuard { pgpfa_ctx* c; ~NdevGuard() { c->cur_ndev = nullptr; } } ndev_guard{c};
          c->live_gemms.clear();
          for (int it = 0; it < c->pcg_inner_max; ++it) {
            c->live_gemm_collect = (it == 0);
            CHK(prior_mv_all(c, nb, c->Pv, c->Qv, nullptr, skip, c->live, na));
            dispatch_pw(p, [&](auto pw) {
              constexpr int PW = decltype(pw)::value;
              if constexpr (PW <= 16) {
                if (c->pcg_w32)
                  hipLaunchKernelGGL(pcg_hessvec32_dot_kernel<PW>, dim3(ntile, na), dim3(256), 0, c->st, c->W32, sW32, c->Pv, c->Qv, ld, T, p,
                                     c->live, c->sc_pq, skip, (const PcgCtl*)c->pcgctl);
                else
                  hipLaunchKernelGGL(pcg_hessvec_dot_kernel<PW>, dim3(ntile, na), dim3(256), 0, c->st, c->W, (long long)T * p * p, c->Pv, c->Qv,
                                     ld, T, p, c->live, c->sc_pq, (const int*)&c->pcgctl->nlive);
                hipLaunchKernelGGL(pcg_xr_apply_kernel<PW>, gbin, dim3(256), 0, c->st, c->Gbar, c->Dl, c->Rv, c->Pv, c->Qv, c->Xt, ld, T, p,
                                   c->live, na, c->sc_rz, c->sc_pq, ntile, skip, (const PcgCtl*)c->pcgctl);
              }
            });
            CHK(shared_solve(c, nb, c->Rv, c->Zv, skip, false, false, c->live, na));
            dispatch_pw(p, [&](auto pw) {
              constexpr int PW = decltype(pw)::value;
              if constexpr (PW <= 16)
                hipLaunchKernelGGL(pcg_apply2_dots_kernel<PW>, gbin, dim3(256), 0, c->st, c->Gbar, c->Rv, c->Xt, c->eps, c->Zv, ld, T, p, c->live,
                                   na, c->sc_part2, skip, (const PcgCtl*)c->pcgctl);
            });
            hipLaunchKernelGGL(pcg_update_p2_kernel, dim3(na), dim3(256), 0, c->st, c->Zv, c->Pv, ld, nvec, c->live, c->sc_part2, ntile,
                               c->sc_rz, c->sc_rr, c->sc_rr0, 0, c->pcgctl, c->pcg_ratio);
            hipLaunchKernelGGL(pcg_check_kernel, dim3(1), dim3(256), 0, c->st, c->pcgctl, (volatile int*)c->d_hpcg, c->live,
                               (const float*)c->pcg_ratio, (const float*)c->pcg_eta, c->pcg_inner_min);
            if (*(volatile int*)&c->h_pcg[0]) break;           // the device has already stopped: whatever is enqueued is a no-op
            // stay at most 3 iterations ahead of the device (an iteration enqueued past the stop costs ~13 empty launches: that
            // matters when the kernels themselves take microseconds); the wait spins on the host-mapped counter, no API call
            if (it + 1 < c->pcg_inner_max) {
              const auto t_spin = std::chrono::steady_clock::now();
              while (!*(volatile int*)&c->h_pcg[0] && (it + 1) - *(volatile int*)&c->h_pcg[1] > 3) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 5.0) break;   // (never hang on a lost flag)
              }
              if (*(volatile int*)&c->h_pcg[0]) break;
            }
          }
          c->cur_ndev = nullptr;
          c->live_gemm_collect = false;
          CHK(dl_enqueue(c, &fused_ctl, c->pcgctl, sizeof(PcgCtl)));
          HIPC(hipGetLastError());
          done_inner = -1;                                     // read from the control block with the scalars below
        } else {
        CHK(shared_solve(c, nb, c->Rv, c->Zv, nullptr, true, true, c->list_a, na));
        hipLaunchKernelGGL(pcg_update_p_kernel, dim3(na), dim3(256), 0, c->st, c->Rv, c->Zv, c->Pv, ld, nvec, c->list_a, c->sc_rz, c->sc_rr0, 1);
        for (int it = 0; it < c->pcg_inner_max; ++it) {
          CHK(prior_mv_all(c, nb, c->Pv, c->Qv, nullptr, nullptr, c->list_a, na));
          int pq_tiles = 1;
          dispatch_pw(p, [&](auto pw) {
            constexpr int PW = decltype(pw)::value;
            if constexpr (PW <= 16) {
              pq_tiles = (T + 63) / 64;
              hipLaunchKernelGGL(pcg_hessvec_dot_kernel<PW>, dim3(pq_tiles, na), dim3(256), 0, c->st, c->W, (long long)T * p * p, c->Pv,
                                 c->Qv, ld, T, p, c->list_a, c->sc_pq);
            } else {
              hipLaunchKernelGGL(pcg_hessvec_dot_wide_kernel<PW>, dim3(na), dim3(256), 0, c->st, c->W, (long long)T * p * p, c->Pv, c->Qv, ld,
                                 T, p, c->list_a, c->sc_pq);
            }
          });
          hipLaunchKernelGGL(pcg_update_xr_kernel, dim3(na), dim3(256), 0, c->st, c->Dl, c->Rv, c->Pv, c->Qv, ld, nvec, c->list_a, c->sc_rz, c->sc_pq,
                             pq_tiles);
          CHK(shared_solve(c, nb, c->Rv, c->Zv, nullptr, true, true, c->list_a, na));
          hipLaunchKernelGGL(pcg_update_p_kernel, dim3(na), dim3(256), 0, c->st, c->Rv, c->Zv, c->Pv, ld, nvec, c->list_a, c->sc_rz, c->sc_rr, 0);
          done_inner = it + 1;
          if (done_inner >= c->pcg_inner_min) {
            CHK(download(c, rr.data(), c->sc_rr, nb));
            if (it == c->pcg_inner_min - 1) CHK(download(c, rr0.data(), c->sc_rr0, nb));
            double worst = 0.0;
            for (int s : active) worst = std::max(worst, rr0[s] > 0.0 ? std::sqrt(rr[s] / rr0[s]) : 0.0);
            if (worst <= eta_target) break;
          }
        }
        }
        if (onek && rx32) {
          // the step comes back in single precision on the solve's private rows: widened into Dl by the kernel that reads it first (inside the timed
          // region of the solve: it is part of what the single-precision step costs)
          hipLaunchKernelGGL(step_stats_x32_kernel, dim3(na), dim3(256), 0, c->st, c->Gt, reinterpret_cast<const float*>(c->Zv) + (size_t)ld * ((size_t)c->B + 128),
                             c->Dl, ld, T, Tl, p, c->list_a, c->sc_dec, c->sc_smax);
          if (c->time_newton) hipEventRecord(newton_ev.back().second, c->st);
        } else {
          if (c->time_newton) hipEventRecord(newton_ev.back().second, c->st);
          hipLaunchKernelGGL(step_stats_kernel, dim3(na), dim3(256), 0, c->st, c->Gt, c->Dl, ld, nvec, c->list_a, c->sc_dec, c->sc_smax);
        }
        CHK(prior_mv_all(c, nb, c->Dl, c->KD));
        hipLaunchKernelGGL(dots3_kernel, dim3(na), dim3(256), 0, c->st, c->Xc, ld, c->KX, ld, c->Dl, ld, c->KD, ld, nvec, c->list_a, c->sc_qxx,
                           c->sc_qdx, c->sc_qdd);
        HIPC(hipGetLastError());
        {
          const size_t nB = (size_t)c->B;
          std::vector<double> pack(7 * nB);
          CHK(download(c, pack.data(), c->sc_pack, 7 * nB));
          std::copy(pack.begin(), pack.begin() + nb, dec.begin());
          std::copy(pack.begin() + nB, pack.begin() + nB + nb, smax.begin());
          std::copy(pack.begin() + 2 * nB, pack.begin() + 2 * nB + nb, qxx.begin());
          std::copy(pack.begin() + 3 * nB, pack.begin() + 3 * nB + nb, qdx.begin());
          std::copy(pack.begin() + 4 * nB, pack.begin() + 4 * nB + nb, qdd.begin());
          std::copy(pack.begin() + 5 * nB, pack.begin() + 5 * nB + nb, rr.begin());
          std::copy(pack.begin() + 6 * nB, pack.begin() + 6 * nB + nb, rr0.begin());
        }
        double slot_iters = (double)na * done_inner;
        if (done_inner < 0) {                                   // (the download above synchronised the stream)
          done_inner = fused_ctl.iters;
          slot_iters = (double)fused_ctl.slot_iters;
          if (c->prof.on)                                       // algorithmic flops of the live-list products: per column x slot-iterations
            for (const auto& lg : c->live_gemms) {
              c->prof.flops[TAG_GEMM] += lg.second * slot_iters;
              c->prof.shapes[lg.first].flops += lg.second * slot_iters;
            }
        }
        n_pcg += slot_iters;
        if (c->pcg_trace) {
          // achieved residual ratios of the live slots: worst, median, and how many already met the target
          std::vector<double> ratio;
          for (int s : active) ratio.push_back(rr0[s] > 0.0 ? std::sqrt(rr[s] / rr0[s]) : 0.0);
          std::sort(ratio.begin(), ratio.end());
          int met = 0;
          for (double v : ratio) met += (v <= eta_target) ? 1 : 0;
          std::fprintf(stderr, "pcg_trace: outer %d live %d inner %d eta_target %.2e achieved worst %.2e median %.2e best %.2e met %d\n", outer, na, done_inner,
                       eta_target, ratio.back(), ratio[ratio.size() / 2], ratio.front(), met);
        }
        {
          // mandatory HBM traffic of one PCG iteration (the bytes a perfect implementation still moves): per live slot
          // 20 passes over an n-vector (H p = K^-1 p + W p: 5; x, r updates: 6; preconditioner G(eps r + F S F^T G r): 6; p = z + beta p: 3),
          // 4 over an r-vector, the curvature blocks (packed FP32 lower triangles, or FP64 full blocks); once per iteration the operators
          // K^-1 (p T^2), F and F^T (T r each) and S (r^2).  Dense plan: P^-1 (n^2) instead of F / S.
          const double npk = (double)(p * (p + 1) / 2);
          // Two-kernel step (pcg_cg_a/b_kernel): 17 passes (A reads r, y and writes z, s; B reads z, s, p, q, x, r and writes p, q, x, r, t; the
          // products read t and write y), 4 over an r-vector, the packed FP32 curvature; once per step F, F^T, S and the packed triangles of Gb
          // (FP64) and Wb (FP32).  No K^-1 in the loop.
          const double vecs = (onek ? 17.0 : 20.0) * nvec * 8.0 + (c->plan_lowrank ? 4.0 * c->rtot * 8.0 : 0.0);
          const double curv = (fused && c->pcg_w32) ? (double)T * npk * 4.0 : (double)T * p * p * 8.0;
          const double ops = onek ? (2.0 * T * c->rtot + (double)c->rtot * c->rtot + 1.5 * T * npk) * 8.0
                                  : (double)p * T * T * 8.0 + (c->plan_lowrank ? (2.0 * T * c->rtot + (double)c->rtot * c->rtot) * 8.0 : (double)nvec * nvec * 8.0);
          newton_bytes += slot_iters * (vecs + curv) + (double)done_inner * ops;
          // (bytes per entry of an n-vector and slot-step: FP64 form 136 = 17 passes; z, s, p, q, t / y in single precision 88; r and x too 68 -
          //  A reads r 4, y 4, writes z 4, s 4; B reads z 4, s 4, x 4, r 4, p 4, q 4, writes p 4, q 4, x 4, r 4, t 4; the products read t 4, write y 4 -
          //  plus, once per solve and slot, the widening of the step: 4 read, 8 written)
          const double vec_b = rx32 ? 68.0 : (v32 ? 88.0 : 0.0);
          newton_bytes_moved += slot_iters * ((v32 ? vec_b * nvec + (c->plan_lowrank ? 4.0 * c->rtot * 8.0 : 0.0) : vecs) + curv) + (double)done_inner * ops
                                + ((onek && rx32) ? 12.0 * nvec * (double)na : 0.0);
          newton_bytes_survey += slot_iters * ((double)c->q * T + 8.0 * (2.0 * p * T + (double)T * p * p));
        }
        std::vector<int> cand, next, failed;
        for (int s : active) {
          if (!(dec[s] > 0.0) || !std::isfinite(dec[s]) || !std::isfinite(smax[s])) continue;   // leave to the fallback
          cand.push_back(s);
        }
        CHK(line_search(cand, &failed));
        std::vector<char> bad(nb, 0);
        for (int s : failed) bad[s] = 1;
        std::vector<double> gnew;
        if (onek) {
          // The inner solve ran on H~ = Kt^-1 + fl32(W): its residual says how well H~ delta = -g was solved, not how far the step took the
          // TRUE gradient down.  Measure that: |g(x + delta)| / |g(x)| of the accepted full steps (the committed Gl + KX; rr0 = |g(x)|^2) enters
          // the error prediction next to the inner ratio, so the stopping rule never rests on the model matrix.
          std::vector<int> okl;
          for (int s : cand) if (!bad[s]) okl.push_back(s);
          gnew.assign(nb, 0.0);
          if (!okl.empty()) {
            CHK(upload_nosync(c, c->list_b, okl.data(), sizeof(int) * okl.size()));
            hipLaunchKernelGGL(grad_norm2_kernel, dim3((unsigned)okl.size()), dim3(256), 0, c->st, c->Gl, c->KX, ld, nvec, c->list_b, c->sc_f);
            CHK(download(c, gnew.data(), c->sc_f, nb));
          }
        }
        for (int s : cand) {
          if (bad[s]) continue;
          // inexact Newton: the error after the step is ~ max(eta, |step|) * |step|, eta = achieved relative residual
          double eta = rr0[s] > 0.0 ? std::sqrt(rr[s] / rr0[s]) : 0.0;
          if (onek && alpha[s] == 1.0 && rr0[s] > 0.0) eta = std::max(eta, std::sqrt(gnew[s] / rr0[s]));
          const double step = alpha[s] * smax[s];
          if (alpha[s] == 1.0 && 10.0 * step * std::max(eta, step) < c->chord_xtol) { stat[s] = 0; continue; }
          err_pred[s] = (alpha[s] == 1.0) ? step * std::max(eta, step) : step;
          next.push_back(s);
        }
        // slots with a non-descent direction or an exhausted search drop to the per-trial fallback below
        std::vector<int> fallback;
        {
          std::vector<char> in_cand(nb, 0);
          for (int s : cand) in_cand[s] = 1;
          for (int s : active) if (!in_cand[s] || bad[s]) { fallback.push_back(s); if (bad[s]) n_fb_search += 1; else n_fb_dir += 1; }
        }
        active.swap(next);
        leftovers.insert(leftovers.end(), fallback.begin(), fallback.end());
        max_it_seen = std::max(max_it_seen, outer + 1);
      }
      // anything still active after the outer cap also goes to the fallback
      n_fb_cap += (double)active.size();
      leftovers.insert(leftovers.end(), active.begin(), active.end());
      active = leftovers;
      std::sort(active.begin(), active.end());
    }

    // ---- phase 1b (fallback, and the only path when shared_pcg is off or the chunk is tiny): per-trial Newton with
    // factor reuse.  A slot factors H at its current point only when it has no factor yet or its chord steps (steps
    // with the stale factor, still descent directions since that factor is SPD) contract too slowly; otherwise the
    // resident factor is reused: one HBM-bound solve instead of n^3/3 flops.
    if (c->plan_lowrank && !active.empty()) {
      // the per-trial fallback needs full-size factor slabs: leave these trials to the dense retry pass of the caller
      for (int s : active) stat[s] = 4;
      active.clear();
    }
    std::vector<char> has_factor(nb, 0), fresh(nb, 0), refactor(nb, 0);
    std::vector<double> prev_step(nb, 0.0);
    std::vector<int> n_chord(nb, 0);
    for (int iter = 0; iter < c->max_iter && !active.empty(); ++iter) {
      const int na = (int)active.size();
      std::vector<int> need;
      for (int s : active) {
        if (!has_factor[s] || refactor[s] || !c->chord) need.push_back(s);
        fresh[s] = 0;
      }
      if (!need.empty()) {
        CHK(upload_list(c, c->list_a, need));
        CHK(assemble(c, c->list_a, (int)need.size()));
        CHK(factor(c, c->ws, c->list_a, (int)need.size()));
        n_fact += (double)need.size();
        for (int s : need) { has_factor[s] = 1; fresh[s] = 1; refactor[s] = 0; n_chord[s] = 0; its[s] += 1; }
      }
      n_solve += na;
      CHK(upload_list(c, c->list_a, active));
      hipLaunchKernelGGL(grad_total_kernel, dim3((nvec + 255) / 256, na), dim3(256), 0, c->st, c->Gl, ld, c->KX, ld, c->Gt, ld, nvec, c->list_a);
      prof_begin(c, TAG_SOLVE, 2.0 * na * (double)c->npad * c->npad);
      hipLaunchKernelGGL(chol_solve_kernel, dim3(na), dim3(256), 0, c->st, c->ws.H, c->ws.sH, c->ld, c->npad, c->ws.Dinv, c->ws.sD, c->Gt, c->Dl,
                         ld, c->list_a, c->sc_dec, c->sc_smax, nvec);
      prof_end(c);
      CHK(prior_mv(c, c->list_a, na, c->Dl, c->KD));
      hipLaunchKernelGGL(dots3_kernel, dim3(na), dim3(256), 0, c->st, c->Xc, ld, c->KX, ld, c->Dl, ld, c->KD, ld, nvec, c->list_a, c->sc_qxx,
                         c->sc_qdx, c->sc_qdd);
      HIPC(hipGetLastError());
      CHK(download(c, dec.data(), c->sc_dec, nb));
      CHK(download(c, smax.data(), c->sc_smax, nb));
      CHK(download(c, qxx.data(), c->sc_qxx, nb));
      CHK(download(c, qdx.data(), c->sc_qdx, nb));
      CHK(download(c, qdd.data(), c->sc_qdd, nb));
      CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
      CHK(dl_flush(c));

      std::vector<int> cand, failed;
      for (int s : active) {
        if (info[s] != 0 || !std::isfinite(dec[s])) { stat[s] = 3; continue; }
        cand.push_back(s);
      }
      CHK(line_search(cand, &failed));
      for (int s : failed) stat[s] = 2;   // line search exhausted
      std::vector<int> next;
      for (int s : active) {
        if (stat[s] == 2 || stat[s] == 3) continue;
        const double step = alpha[s] * smax[s];
        if (fresh[s]) {
          // true Newton step: quadratic convergence, the error after the step is ~step^2
          if (step < c->xtol) { stat[s] = 0; continue; }
          if (step > c->chord_max_step) refactor[s] = 1;      // still far from the mode: keep factoring
        } else {
          // chord step: linear convergence with ratio rho, the error after the step is ~rho/(1-rho)*step
          const double rho = prev_step[s] > 0.0 ? step / prev_step[s] : 1.0;
          n_chord[s] += 1;
          if (step < c->chord_xtol && rho < 0.5) { stat[s] = 0; continue; }
          if (rho > c->chord_rho || n_chord[s] >= c->chord_max) refactor[s] = 1;
        }
        prev_step[s] = step;
        next.push_back(s);
      }
      active.swap(next);
      max_it_seen = std::max(max_it_seen, iter + 1);
    }

    if (!var) break;
    // ---- variational fixed point: rates at the modes, their covariance blocks, new offsets
    // A mode search that did not settle (iteration cap 1, line search exhausted 2, factor failure 3 / 4) hands ITS trial back - status 2: the
    // caller finishes it with L-BFGS from the lambda of the point the search reached - and the other slots go on.  Only a non-finite state
    // (checked below on the offsets of every slot still open or handed back in this pass) fails the call.
    std::vector<char> handed(nb, 0);
    for (int s = 0; s < nb; ++s)
      if (vstat[s] == 1 && stat[s] != 0) { vstat[s] = 2; handed[s] = 1; vouter[s] = vo + 1; }
    c->lam_out_active = true;
    CHK(poisson(c, c->ident, nb, c->Xc, c->Glt, c->Wt, c->sc_f, 0));        // lambda = exp(C m + d + offset) -> c->lamd
    c->lam_out_active = false;
    if (c->plan_lowrank) { CHK(dual_jitter(c, nb)); CHK(posterior_blocks(c, nb, 1.0, false, false)); }   // (c->W: curvature at the modes = C^T diag(lambda) C)
    else CHK(posterior_blocks(c, nb, 1.0 + 1e-6, false));
    CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
    CHK(dl_flush(c));
    for (int s = 0; s < nb; ++s)
      if (info[s] != 0) { var_superseded(); return fail("variational fixed point: posterior precision of trial %d not positive definite", tos[s]); }
    CHK(var_offsets(c, nb, c->dgrad));
    {
      // change of the offsets first (step 0: nothing moves), the update afterwards and only for the slots that go on - a slot that settles
      // keeps the offsets its lambda was computed with, so that (lambda, mode, offsets) stay one consistent triple
      std::vector<double> zero(nb, 0.0);
      CHK(upload_nosync(c, c->sc_alpha, zero.data(), sizeof(double) * nb));
      hipLaunchKernelGGL(var_update_kernel, dim3(nb), dim3(256), 0, c->st, c->voff, (const double*)c->dgrad, mlam, (const double*)c->sc_alpha, c->sc_f);
      CHK(download(c, vdelta.data(), c->sc_f, nb));
    }
    bool any_open = false;
    for (int s = 0; s < nb; ++s) {
      if ((vstat[s] == 1 || handed[s]) && !std::isfinite(vdelta[s])) { var_superseded(); return fail("variational fixed point: non-finite offsets for trial %d", tos[s]); }
      if (vstat[s] != 1) continue;
      vouter[s] = vo + 1;
      if (vdelta[s] <= var->tol) { vstat[s] = 0; continue; }
      // the map contracts by about half the largest posterior variance of a log rate per pass; a pass that does not shrink the change
      // halves the step, three such passes give the trial back to the caller (status 2: the L-BFGS driver takes it from this lambda)
      if (vdelta_prev[s] >= 0.0 && vdelta[s] > 0.7 * vdelta_prev[s]) { vdamp[s] *= 0.5; if (++vslow[s] >= 3) { vstat[s] = 2; continue; } }
      vdelta_prev[s] = vdelta[s];
      if (vo + 1 >= var->max_outer) continue;       // (stays 1: iteration cap)
      any_open = true;
    }
    if (!any_open) break;
    {
      std::vector<double> step(nb, 0.0);
      for (int s = 0; s < nb; ++s) step[s] = (vstat[s] == 1) ? vdamp[s] : 0.0;
      CHK(upload_nosync(c, c->sc_alpha, step.data(), sizeof(double) * nb));
      hipLaunchKernelGGL(var_update_kernel, dim3(nb), dim3(256), 0, c->st, c->voff, (const double*)c->dgrad, mlam, (const double*)c->sc_alpha, c->sc_f);
      HIPC(hipGetLastError());
    }
    }
    if (var) {
      // optimum out: rho = log lambda, the dual cost there (inference.py:196-213), statuses
      c->var_active = false;
      CHK(dual_eval_slots(c, nb, tos, false, var->fopt + c0, false));
      // the optimum stays on the device for pgpfa_dual_finalize(lam = NULL) and for blocks rebuilt on demand
      if (!c->lam_keep) {
        const size_t bytes = (size_t)c->R * mlam * sizeof(double);
        if (hipMalloc((void**)&c->lam_keep, bytes) != hipSuccess) { (void)hipGetLastError(); c->lam_keep = nullptr; return fail("hipMalloc(%zu bytes) for the resident dual variables failed", bytes); }
        c->bytes += bytes;
      }
      for (int s = 0; s < nb; ++s) {
        CHK(copy_dev(c, c->lam_keep + (size_t)tos[s] * mlam, c->lamd + (size_t)s * mlam, mlam * sizeof(double)));
        c->lam_resident[tos[s]] = 1; c->lam_valid[tos[s]] = 1;
        // (lam_keep also feeds the blocks rebuilt on demand of a dual posterior: whatever posterior the trial had is superseded until
        // pgpfa_dual_finalize has run on the new optimum)
        c->trial_dual[tos[s]] = 0; c->trial_snap[tos[s]] = -1; c->vsmgp_ok[tos[s]] = 0;
      }
      if (var->lam_out) CHK(download(c, var->lam_out + (size_t)c0 * mlam, c->lamd, (size_t)nb * mlam));
      if (var->rho) {
        hipLaunchKernelGGL(var_log_kernel, dim3(2048), dim3(256), 0, c->st, (const double*)c->lamd, c->dgrad, (size_t)nb * mlam);
        CHK(download(c, var->rho + (size_t)c0 * mlam, c->dgrad, (size_t)nb * mlam));
      }
      for (int s = 0; s < nb; ++s) {
        if (var->outer) var->outer[c0 + s] = vouter[s];
        var->vstatus[c0 + s] = vstat[s];
        if (iters) iters[c0 + s] = its[s];
        if (status) status[c0 + s] = stat[s];
      }
      n_fact += nb;
      continue;
    }
    if (loo) {
      // prediction of the held-out neurons from the modes in Xc (Xt and sc_f are free scratch here)
      hipLaunchKernelGGL(loo_predict_kernel, dim3(nb), dim3(256), 0, c->st, c->Xc, ld, c->C, c->d, c->Y, c->Yhi, c->trial_of_slot, c->mask_of_slot,
                         c->q, p, T, c->Xt, ld, c->sc_f);
      HIPC(hipGetLastError());
      HIPC(hipMemcpy2DAsync(loo->y_pred + (size_t)c0 * T, (size_t)T * sizeof(double), c->Xt, (size_t)ld * sizeof(double), (size_t)T * sizeof(double),
                            nb, hipMemcpyDeviceToHost, c->st));
      CHK(download(c, loo->err + c0, c->sc_f, nb));
      HIPC(hipStreamSynchronize(c->st));
      for (int s = 0; s < nb; ++s) {
        if (iters) iters[c0 + s] = its[s];
        if (status) status[c0 + s] = stat[s];
      }
      continue;
    }
    // posterior covariance blocks at the mode
    {
      const bool sum_only = c->plan_lowrank && !c->keep_trial_vsmgp;
      PhaseRange range_cov("pgpfa.covariance_blocks");
      CHK(posterior_blocks(c, nb, 1.0, true, sum_only));
      for (int t : tos) c->vsmgp_ok[t] = sum_only ? 0 : 1;
    }
    n_fact += nb;
    for (int s = 0; s < nb; ++s) its[s] += 1;
    {
      // the mode a trial had before this E-step becomes its extrapolation base (once per E-step: a dense retry pass
      // of the same E-step must not overwrite it with its own unfinished start point)
      std::vector<int> rot(nb, 0);
      for (int s = 0; s < nb; ++s) {
        const int tr_ = tos[s];
        if (c->mode_serial[tr_] != c->estep_serial) {
          rot[s] = 1;
          c->prev_serial[tr_] = c->mode_serial[tr_];
          c->mode_serial[tr_] = c->estep_serial;
        }
      }
      CHK(upload_list(c, c->list_a, rot));
      hipLaunchKernelGGL(scatter_rotate_kernel, dim3((nvec + 255) / 256, nb), dim3(256), 0, c->st, c->Xc, ld, nvec, c->Xmode, c->Xprev,
                         c->trial_of_slot, c->list_a);
    }
    CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
    CHK(dl_flush(c));
    HIPC(hipGetLastError());
    for (int s = 0; s < nb; ++s) {
      if (info[s] != 0 && stat[s] == 0) stat[s] = 3;
      total += f[s];
      if (iters) iters[c0 + s] = its[s];
      if (status) status[c0 + s] = stat[s];
    }
  }
  if (obj_sum) *obj_sum = total;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  c->info["last_estep_ms"] = ms;
  c->info["last_newton_factorizations"] = n_fact;
  c->info["last_newton_solves"] = n_solve;
  c->info["last_pcg_iterations"] = n_pcg;
  c->info["last_shared_factorizations"] = n_shared;
  c->info["last_cov_lowrank"] = c->last_cov_lowrank ? 1.0 : 0.0;
  c->info["last_newton_max_iter"] = max_it_seen;
  c->info["last_cold_restarts"] = n_cold;
  c->info["last_fallback_no_descent"] += n_fb_dir;            // (summed over the passes of one E-step: pgpfa_estep_laplace resets them)
  c->info["last_fallback_line_search"] += n_fb_search;
  c->info["last_fallback_outer_cap"] += n_fb_cap;
  if (c->time_newton) {
    // (every chunk ended on a stream synchronisation: the events are complete)
    double nms = 0.0;
    for (auto& ev : newton_ev) {
      float e_ms = 0.f;
      if (hipEventElapsedTime(&e_ms, ev.first, ev.second) == hipSuccess) nms += e_ms;
      c->prof.idle.push_back(ev.first); c->prof.idle.push_back(ev.second);
    }
    (void)hipGetLastError();
    c->info["last_newton_solve_ms"] = nms;
    c->info["last_newton_solve_bytes"] = newton_bytes;
    c->info["last_newton_solve_bytes_survey"] = newton_bytes_survey;
    c->info["last_newton_solve_bytes_moved"] = newton_bytes_moved;
  }
  return 0;
}

int pgpfa_estep_laplace(pgpfa_ctx* c, int n, const int32_t* idx, int warm_start, double* obj_sum, int32_t* iters, int32_t* status) {
  if (!c) return fail("null context");
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  const int N = (int)tr.v.size();
  std::vector<int32_t> it1(N), st1(N);
  double obj = 0.0;
  HIPC(hipSetDevice(c->device));
  c->pacc_used = false; c->pacc_valid = false;
  c->estep_serial += 1;
  c->info["last_eps_wt_norm"] = 0.0; c->info["last_eps_wt_rms"] = 0.0;      // maxima over the chunks of THIS call
  HIPC(hipMemsetAsync(c->Pacc, 0, (size_t)c->Tp * c->Tp * c->p * sizeof(double), c->st));
  snapshot_params(c, tr.v);
  for (int t : tr.v) c->trial_dual[t] = 0;
  {
    // how far the parameters have moved since the last Laplace E-step (relative: the largest of |dC| / |C|, |dd| / |d|, |d log tau|) - the start
    // points are only extrapolated while that pace holds (ctx.h, extrapolate_guard)
    double step = -1.0;
    if (c->estepC.size() == c->hC.size() && c->estepd.size() == c->hd.size() && c->esteptau.size() == c->htau.size() && !c->hC.empty()) {
      auto rel = [](const std::vector<double>& a, const std::vector<double>& b) {
        double num = 0.0, den = 0.0;
        for (size_t i = 0; i < a.size(); ++i) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
        return std::sqrt(num) / std::max(std::sqrt(den), 1e-300);
      };
      step = std::max(rel(c->hC, c->estepC), rel(c->hd, c->estepd));
      for (size_t k = 0; k < c->htau.size(); ++k) step = std::max(step, std::fabs(std::log(c->htau[k] / c->esteptau[k])));
      if (!std::isfinite(step)) step = -1.0;
    }
    c->par_step_prev = c->par_step; c->par_step = step;
    c->estepC = c->hC; c->estepd = c->hd; c->esteptau = c->htau;
    c->info["last_param_step"] = c->par_step; c->info["last_param_step_prev"] = c->par_step_prev;
    c->info["last_fallback_no_descent"] = 0.0; c->info["last_fallback_line_search"] = 0.0; c->info["last_fallback_outer_cap"] = 0.0;
  }
  CHK(estep_impl(c, tr, warm_start, true, &obj, it1.data(), st1.data()));
  c->info["last_retry_ms"] = 0.0;
  // trials the low-rank plan could not finish (its shared-preconditioner Newton gave up on them and the per-trial
  // fallback needs full-size slabs) are redone under the dense plan, warm-started from where they stopped
  Trials retry;
  std::vector<int> pos;
  for (int i = 0; i < N; ++i)
    if (st1[i] == 4) { retry.v.push_back(tr.v[i]); pos.push_back(i); }
  if (!retry.v.empty()) {
    // their partial objective is replaced: recompute the total from scratch for them
    std::vector<int32_t> it2(retry.v.size()), st2(retry.v.size());
    double obj_bad = 0.0, obj_redo = 0.0;
    {
      // objective of the unfinished trials as counted in the first pass
      std::vector<double> X((size_t)retry.v.size() * c->n), fv(retry.v.size());
      std::vector<int32_t> ridx(retry.v.begin(), retry.v.end());
      CHK(pgpfa_get_post_mean(c, (int)ridx.size(), ridx.data(), X.data()));
      CHK(pgpfa_laplace_eval(c, (int)ridx.size(), ridx.data(), X.data(), fv.data(), nullptr));
      for (double v : fv) obj_bad += v;
    }
    {
      // the dense plan of this pass is sized for the retry list, not for the largest list the context has seen (a plan for 1024 dense slabs maps and
      // clears > 200 GB: seconds); the next E-step re-partitions the arena for its own plan
      const auto t_retry = std::chrono::steady_clock::now();
      struct WantGuard { pgpfa_ctx* c; int keep; ~WantGuard() { c->want_slots = keep; } } want_guard{c, c->want_slots};
      c->want_slots = 0;
      CHK(estep_impl(c, retry, 1, false, &obj_redo, it2.data(), st2.data()));
      c->info["last_retry_ms"] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_retry).count();
    }
    obj += obj_redo - obj_bad;
    for (size_t j = 0; j < pos.size(); ++j) { it1[pos[j]] += it2[j]; st1[pos[j]] = st2[j]; }
    c->info["last_dense_retries"] = (double)retry.v.size();
  } else {
    c->info["last_dense_retries"] = 0.0;
  }
  CHK(remember_trials(c, tr.v));
  c->pacc_valid = c->pacc_used && retry.v.empty();
  if (obj_sum) *obj_sum = obj;
  for (int i = 0; i < N; ++i) {
    if (iters) iters[i] = it1[i];
    if (status) status[i] = st1[i];
  }
  return 0;
}

// Exact integer moments of the resident counts over the listed trials: sum[q], cross[q][q] (symmetric), n_samples.

int pgpfa_loo_predict(pgpfa_ctx* c, int n, const int32_t* idx, double* y_pred, double* err_sum) {
  if (!c || !y_pred || !err_sum) return fail("null argument");
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr));
  const int q = c->q, T = c->T;
  Trials items;
  std::vector<int> mask;
  for (int t : tr.v)
    for (int nn = 0; nn < q; ++nn) { items.v.push_back(t); mask.push_back(nn); }
  const int N = (int)items.v.size();
  std::vector<double> err(N);
  std::vector<int32_t> st(N), it(N);
  LooJob job{&mask, y_pred, err.data()};
  CHK(estep_impl(c, items, 0, true, nullptr, it.data(), st.data(), &job));
  // items the low-rank plan could not finish are redone under the dense plan (as in pgpfa_estep_laplace)
  Trials redo;
  std::vector<int> redo_mask, pos;
  for (int i = 0; i < N; ++i)
    if (st[i] == 4) { redo.v.push_back(items.v[i]); redo_mask.push_back(mask[i]); pos.push_back(i); }
  if (!redo.v.empty()) {
    const int M = (int)redo.v.size();
    std::vector<double> yp2((size_t)M * T), err2(M);
    std::vector<int32_t> st2(M), it2(M);
    LooJob job2{&redo_mask, yp2.data(), err2.data()};
    {
      struct WantGuard { pgpfa_ctx* c; int keep; ~WantGuard() { c->want_slots = keep; } } want_guard{c, c->want_slots};
      c->want_slots = 0;                    // (dense plan for the redo list only: see pgpfa_estep_laplace)
      CHK(estep_impl(c, redo, 0, false, nullptr, it2.data(), st2.data(), &job2));
    }
    for (int j = 0; j < M; ++j) {
      std::copy(yp2.begin() + (size_t)j * T, yp2.begin() + (size_t)(j + 1) * T, y_pred + (size_t)pos[j] * T);
      err[pos[j]] = err2[j];
      st[pos[j]] = st2[j];
    }
  }
  double total = 0.0;
  int bad = 0;
  for (int i = 0; i < N; ++i) { total += err[i]; if (st[i] != 0) ++bad; }
  c->info["last_loo_unconverged"] = (double)bad;
  *err_sum = total;
  return 0;
}


