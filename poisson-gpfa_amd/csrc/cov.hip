// libpgpfa_hip.so - cov.hip (one translation unit of the C-ABI library; shared declarations: ctx.h)
#include "ctx.h"
#include "model.h"
#include "split.h"
#include "ytmix.h"

using namespace pgpfa;

// post_vsm[t] = Gram of the rows (., t) of the panel Mt (ncol columns, column stride ld) for ns slots: matrix-core kernel beyond 10
// latents, vector kernel up to 10.  (f32: the panel is single precision)
template <typename TIN>
void launch_post_vsm(pgpfa_ctx* c, const TIN* Mt, long long sM, int ncol, int ns, int full_range, const int* roff = nullptr, int ts = 0) {
  const int T = c->T, p = c->p;
  if (ts <= 0) ts = T;                                       // row stride between latents in the panel
  if (p > 10 && c->vsm_mfma) {
    const int CB = post_vsm_mfma_cb(p, sizeof(TIN) == 4);
    if (p <= 20 && c->vsm_b4) {                              // four latents x four bins per 4 x 4 x 4 block product
      constexpr int VW = 16 / (int)sizeof(TIN);
      const bool vec = c->vsm_b4 >= 2 && T % VW == 0 && ts % VW == 0 && c->ld % VW == 0 && sM % VW == 0 && (((size_t)Mt) & 15) == 0;
      auto go = [&](auto nbk, auto vv) {
        constexpr int NBK = decltype(nbk)::value;
        constexpr bool VEC = decltype(vv)::value;
        const size_t lds4 = (size_t)CB * post_vsm_b4_cs(p, VEC) * sizeof(TIN);
        constexpr int BINS = VEC ? 64 : 32;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&post_vsm_b4_kernel<NBK, TIN, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        hipLaunchKernelGGL((post_vsm_b4_kernel<NBK, TIN, VEC>), dim3((T + BINS - 1) / BINS, ns), dim3(512), lds4, c->st, Mt, sM, c->ld, ncol, T, p, c->vsm,
                           c->ident, c->trial_of_slot, full_range, CB, roff, (int)GBN, ts);
      };
      auto gov = [&](auto nbk) { if (vec) go(nbk, std::true_type{}); else go(nbk, std::false_type{}); };
      if (p <= 12) gov(std::integral_constant<int, 3>{}); else if (p <= 16) gov(std::integral_constant<int, 4>{}); else gov(std::integral_constant<int, 5>{});
      return;
    }
    const size_t lds = (size_t)CB * p * 33 * sizeof(TIN);
    if (p <= 16)
      hipLaunchKernelGGL((post_vsm_mfma_kernel<1, TIN>), dim3((T + 31) / 32, ns), dim3(512), lds, c->st, Mt, sM, c->ld, ncol, T, p, c->vsm, c->ident,
                         c->trial_of_slot, full_range, CB, roff, (int)GBN, ts);
    else
      hipLaunchKernelGGL((post_vsm_mfma_kernel<2, TIN>), dim3((T + 31) / 32, ns), dim3(512), lds, c->st, Mt, sM, c->ld, ncol, T, p, c->vsm, c->ident,
                         c->trial_of_slot, full_range, CB, roff, (int)GBN, ts);
    return;
  }
  const int KY = post_vsm_rows(p);
  dispatch_pmax(p, [&](auto pm) {
    hipLaunchKernelGGL((post_vsm_kernel<decltype(pm)::value, TIN>), dim3((T + 63) / 64, ns), dim3(64, KY), 0, c->st, Mt, sM, c->ld, ncol, T, p, c->vsm,
                       c->ident, c->trial_of_slot, full_range, ts);
  });
}


int post_vsm_from_mt(pgpfa_ctx* c, int nslots) {
  launch_post_vsm(c, (const double*)c->ws.Mt, (long long)c->ws.sM, c->npad, nslots, 0);
  HIPC(hipGetLastError());
  return 0;
}

// The GEMM addresses A, B and C with one slot index; post_vsmGP is indexed by trial, so the slot
// result goes through a slot-indexed staging slab and is scattered afterwards.
__global__ void scatter_vsmgp_kernel(const double* __restrict__ src, long long sSrc, double* __restrict__ dst, int T, int p, int k,
                                     const int* __restrict__ trial_of_slot) {
  const int slot = blockIdx.y;
  const size_t r = trial_of_slot[slot];
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (size_t)T * T) {
    const size_t a = e % T, b = e / T;                 // column-major (a,b); only a >= b was computed
    dst[(r * p + k) * T * T + e] = (a >= b) ? src[(size_t)slot * sSrc + e] : src[(size_t)slot * sSrc + a * T + b];
  }
}


int ensure_mt_clean(pgpfa_ctx* c) {
  if (!c->mt_dirty) return 0;
  HIPC(hipMemsetAsync(c->ws.Mt, 0, (size_t)c->ws.sM * c->B * sizeof(double), c->st));
  c->mt_dirty = false;
  return 0;
}

// Per-trial T x T blocks live in an allocation of their own, freed only with the context: allocated lazily (often in the
// middle of an E-step, i.e. after the workspace mark), it must not be swept up by free_workspace on a re-plan.
int ensure_vsmgp_buffer(pgpfa_ctx* c) {
  if (c->vsmgp) return 0;
  const size_t bytes = (size_t)c->R * c->p * c->T * c->T * sizeof(double);
  void* ptr = nullptr;
  hipError_t e = hipMalloc(&ptr, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail("hipMalloc(%zu bytes) for the per-trial post_vsmGP blocks failed: %s", bytes, hipGetErrorString(e)); }
  e = hipMemsetAsync(ptr, 0, bytes, c->st);
  if (e != hipSuccess) { hipFree(ptr); return fail("hipMemset failed: %s", hipGetErrorString(e)); }
  c->vsmgp = reinterpret_cast<double*>(ptr);
  c->bytes += bytes;
  return 0;
}

static int posterior_blocks_dense(pgpfa_ctx* c, int nb, double diag_scale, bool want_vsmgp) {
  const int T = c->T, p = c->p;
  CHK(ensure_mt_clean(c));
  c->last_cov_lowrank = false;
  CHK(assemble(c, c->ident, nb, diag_scale));
  CHK(factor(c, c->ws, c->ident, nb));
  CHK(inverse_t(c, c->ws, c->ident, nb));
  if (want_vsmgp) {
    CHK(ensure_vsmgp_buffer(c));
    for (int k = 0; k < p; ++k) {
      const int kal = (k * T) / 16 * 16;
      GemmP g{};
      g.A = c->ws.Mt + (size_t)kal * c->ld + (size_t)k * T; g.sA = c->ws.sM; g.lda = c->ld;
      g.B = g.A; g.sB = c->ws.sM; g.ldb = c->ld;
      g.C = c->ws.H; g.sC = c->ws.sH; g.ldc = T;        // slot-indexed staging: the factor slab is free now
      g.M = T; g.N = T; g.K = c->npad - kal; g.alpha = 1.0; g.beta = 0.0;
      // lower tiles only (the scatter mirrors them); Mt is upper triangular, so tile (ti,tj) starts its k range
      // at max(ti,tj)*128 relative to the first column kept (kal <= k*T)
      g.slots = c->ident; g.nbatch = nb; g.mode = GEMM_LOWER; g.kflags = KF_BEGIN_MAXRC;
      CHK(gemm(c, false, g));
      hipLaunchKernelGGL(scatter_vsmgp_kernel, dim3((unsigned)(((size_t)T * T + 255) / 256), nb), dim3(256), 0, c->st, c->ws.H, c->ws.sH, c->vsmgp,
                         T, p, k, c->trial_of_slot);
    }
  }
  prof_begin(c, TAG_VSM, (double)nb * c->npad * c->npad * p);
  launch_post_vsm(c, (const double*)c->ws.Mt, (long long)c->ws.sM, c->npad, nb, 0);
  prof_end(c);
  HIPC(hipGetLastError());
  return 0;
}


// Sum-only covariance output by the exact split form (split.h): Pacc[k] += sum over the chunk's slots of Y~_k Y~_k^T + eps diag(G_t[k][k])
// from L^-T (lw.Mt), Yt (lw.H) and the per-bin blocks G (c->Gbin), without the full-width FP64 product.  Also writes post_vsm.
static int accumulate_split(pgpfa_ctx* c, const CholWS& lw, int nb, int ract, int Ts, bool skip_zero_cols, int ctile, bool fused) {
  const int T = c->T, p = c->p, Tp = c->Tp, rpad = c->rpad;
  const long long sW = (long long)T * p * p;
  const size_t tt = (size_t)T * T;
  (void)skip_zero_cols;
  // scratch: S parts [NS][r_k^2] | X parts [NG][r_k T] | Ssum | Xsum | Z | T1 [p][T^2] | Xfull [p][T^2]; a latent's padded rank r_k can reach
  // T rounded up to 16, so the first five are laid out in units of tq = round_up(T, 16)^2
  constexpr int NS = 256, NG = PACC_SPLITS + 1;
  const size_t tq = (size_t)round_up(T, 16) * round_up(T, 16);
  if (!c->split_buf) {
    const size_t len = ((size_t)NS + NG + 3) * tq + 2 * (size_t)p * tt + 1024;
    if (hipMalloc((void**)&c->split_buf, len * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); c->split_buf = nullptr; return fail("out of device memory for the split accumulation (%zu bytes)", len * sizeof(double)); }
    c->bytes += len * sizeof(double);
  }
  double* Spart = c->split_buf;
  double* Xpart = Spart + (size_t)NS * tq;
  double* Ssum = Xpart + (size_t)NG * tq;
  double* Xsum = Ssum + tq;
  double* Zb = Xsum + tq;
  double* T1 = Zb + tq;
  double* Xfull = T1 + (size_t)p * tt;
  float* D = reinterpret_cast<float*>(lw.H + (size_t)c->ld * rpad);          // behind Yt in every slot's slab
  const long long sD = 2 * (long long)lw.sH;                                   // slab stride in floats
  const int ldd = c->ld;
  // 1. mixing pass: post_vsm and the correction D = eps Wt Yt (single precision); Yt itself stays
  double mix_cols = 0.0;                                   // columns of Yt a latent's rows really hold: left of its first column tile nothing was written
  for (int k = 0; k < p; ++k) mix_cols += std::max(0, ract - (ctile > 0 ? (c->roff[k] / ctile) * ctile : 0));
  // (bytes: the columns of Yt that hold something read once in FP64; D - dense, the mixing couples the latents - written once in FP32)
  if (fused) {
    // Yt was not formed: product and mixing in one kernel.  Work recorded: its FLOPs - products 2 T (ract^2 / 2 + 8 ract) over the
    // triangular panels in 16-column blocks, mixing 2 p^2 + p (p + 1) per (bin, column); info key "last_yt_mix_fused" tells the reader of
    // prof_mix_flops that these are flops, not the bytes of the stand-alone pass.
    prof_begin(c, TAG_MIX, (double)nb * T * ((double)ract * ract + 16.0 * ract + (double)ract * (2.0 * p * p + p * (p + 1.0))));
    dispatch_pw(p, [&](auto pw) {
      constexpr int PW = decltype(pw)::value;
      if constexpr (PW <= 10) {
        YtMixArgs a{};
        a.F = c->Flr; a.Tp = Tp; a.Mts = lw.Mt; a.sM = (long long)lw.sM; a.rpad = rpad;
        a.D = D; a.sD = sD; a.ldd = ldd; a.G = c->Gbin; a.sG = sW;
        a.T = T; a.p = p; a.ract = ract; a.nbx = (T + YTM_BINS - 1) / YTM_BINS; a.nslots = nb; a.eps = c->eps;
        a.vsm = c->vsm; a.slots = c->ident; a.trial_of_slot = c->trial_of_slot; a.ts = Ts; a.dbg = c->yt_mix_dbg;
        a.roff = c->rank_compact ? c->d_roff16 : c->d_roff; a.cmap = c->rank_compact ? c->d_cmap : nullptr; a.nrtab = c->rank_compact ? c->d_nrtab : nullptr;
        a.ncmap = c->rank_compact ? round_up(c->rtot16, (int)NB) : 0;
        if (ytmix_lds(PW) + (size_t)a.ncmap * sizeof(int) > (size_t)160 * 1024) a.ncmap = 0;     // (the kernel then reads the map from memory)
        const size_t lds = ytmix_lds(PW) + (size_t)a.ncmap * sizeof(int);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&yt_mix_kernel<PW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(yt_mix_kernel<PW>, dim3((unsigned)(a.nbx * nb)), dim3(YTM_THREADS), lds, c->st, a);
      }
    });
    prof_end(c);
  } else {
  prof_begin(c, TAG_MIX, (double)nb * T * (mix_cols * 8.0 + (double)p * ract * 4.0));
  if (p > 16)                                               // (17..20 latents: split_candidate admits no others beyond 16)
    hipLaunchKernelGGL((mix_vsm_wide2_kernel<20, true>), dim3((T + 63) / 64, nb), dim3(256), 0, c->st, lw.H, (long long)lw.sH, c->ld, c->Gbin, sW, T, p, ract, c->eps,
                       c->vsm, c->ident, c->trial_of_slot, Ts, c->sink, D, sD, ldd);
  else
  dispatch_pw(p, [&](auto pw) {
    constexpr int PW = decltype(pw)::value;
    if constexpr (PW <= 10) {
      if (c->mix_slot >= 3 && p == PW && ract % 4 == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_slot3_kernel<PW, 128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_slot_lds(PW, 128));
        hipLaunchKernelGGL((mix_slot3_kernel<PW, 128, 2>), dim3((T + 127) / 128, nb), dim3(256), mix_slot_lds(PW, 128), c->st, (const double*)lw.H, (long long)lw.sH, c->ld, D, sD, ldd,
                           c->Gbin, sW, T, ract, c->eps, c->vsm, c->ident, c->trial_of_slot, c->d_roff, ctile, Ts);
        return;
      }
      if (c->mix_slot >= 2 && p == PW && ract % 4 == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_slot2_kernel<PW, 256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_slot_lds(PW, 256));
        hipLaunchKernelGGL((mix_slot2_kernel<PW, 256, 2>), dim3((T + 255) / 256, nb), dim3(256), mix_slot_lds(PW, 256), c->st, (const double*)lw.H, (long long)lw.sH, c->ld, D, sD, ldd,
                           c->Gbin, sW, T, ract, c->eps, c->vsm, c->ident, c->trial_of_slot, c->d_roff, ctile, Ts);
        return;
      }
      if (c->mix_slot) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_slot_kernel<PW, 256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_slot_lds(PW, 256));
        hipLaunchKernelGGL((mix_slot_kernel<PW, 256, 2>), dim3((T + 255) / 256, nb), dim3(256), mix_slot_lds(PW, 256), c->st, (const double*)lw.H, (long long)lw.sH, c->ld, D, sD, ldd,
                           c->Gbin, sW, T, p, ract, c->eps, c->vsm, c->ident, c->trial_of_slot, c->d_roff, ctile, Ts);
        return;
      }
    }
    if constexpr (PW <= 16)
      hipLaunchKernelGGL(mix_vsm_split_kernel<PW>, dim3((T + 63) / 64, nb), dim3(256), 0, c->st, (const double*)lw.H, (long long)lw.sH, c->ld, D, sD, ldd,
                         c->Gbin, sW, T, p, ract, c->eps, c->vsm, c->ident, c->trial_of_slot, c->d_roff, ctile, Ts);
  });
  prof_end(c);
  }
  // 2. the full-width term on the FP16 matrix cores: partial sums per (latent, group of slots) into c->ppart
  const int sps = std::max(1, (nb + PACC_SPLITS - 1) / PACC_SPLITS);
  const int ngroups = (nb + sps - 1) / sps;
  {
    SyrkF16Args a{};
    a.D = D; a.sD = sD; a.ldd = ldd; a.ts = Ts; a.part = c->ppart;
    a.T = T; a.p = p; a.ract = ract; a.nslots = nb; a.sps = sps; a.ngroups = ngroups;
    a.dbg = c->syrk_dbg;
    // 256 x 256 tiles (half the reads of D per output) where T spans more than one of them and the 16-byte loads of the fast path are allowed
    const int t256 = (T + 255) / 256;
    const bool big = c->syrk_tile >= 256 && T > 256 && (Ts & 3) == 0 && (ldd & 3) == 0 && t256 * 256 <= Ts && (((size_t)D) & 15) == 0 && (sD & 3) == 0;
    a.tiles = big ? t256 : (T + 127) / 128; a.ntiles = a.tiles * (a.tiles + 1) / 2;
    const long long blocks = (long long)a.ntiles * ngroups * p;
    prof_begin(c, TAG_VSM, 3.0 * (double)nb * ract * T * T * p);
    if (big) {
      const size_t lds = (size_t)4 * 2 * 256 * 32 * sizeof(_Float16);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&syrk256_f16x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(syrk256_f16x2_kernel, dim3((unsigned)blocks), dim3(512), lds, c->st, a);
    } else {
      hipLaunchKernelGGL(syrk_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, c->st, a);
    }
    prof_end(c);
  }
  // 3. per latent: S_k = sum_r A A^T (r_k x r_k), X_k = sum_r A D_k^T (r_k x T) with A = rows of latent k of L^-T right of column
  //    roff_k (it is upper triangular), both as segmented-K products over groups of slots; then T1 = F S F^T and Xfull = F X
  const int spsS = std::max(1, (nb + NS - 1) / NS);
  for (int k = 0; k < p; ++k) {
    // (compact rank offsets: the rk rows taken from r0 on end in the next latent's first rows - they meet zero columns of F_k in T1 and Xfull
    //  below - and the columns start at r0 rounded down to 16, so that their count stays a multiple of 16: the entries left of r0 are below the
    //  diagonal of L^-T, i.e. zeros)
    const int rk = c->rk[k], r0 = c->roff[k] & ~15;
    const int rrow = c->roff[k];
    const int kw = ract - r0;                                                   // columns of A that are not identically zero
    if (kw <= 0 || rk <= 0) {
      HIPC(hipMemsetAsync(T1 + (size_t)k * tt, 0, tt * sizeof(double), c->st));
      HIPC(hipMemsetAsync(Xfull + (size_t)k * tt, 0, tt * sizeof(double), c->st));
      continue;
    }
    const double* A0 = lw.Mt + rrow + (size_t)r0 * rpad;
    auto seg = [&](int sper, bool is_x, double* out) -> int {                   // groups of `sper` slots (the last one may be short)
      const int nfull = nb / sper, rem = nb - nfull * sper;
      for (int part = 0; part < 2; ++part) {
        const int first = part ? nfull * sper : 0, per = part ? rem : sper, ng = part ? (rem ? 1 : 0) : nfull;
        if (ng == 0 || per == 0) continue;
        GemmP g{};
        g.A = A0 + (size_t)first * lw.sM; g.sA = (long long)per * lw.sM; g.lda = rpad;
        g.kseg = kw; g.sAseg = lw.sM;
        g.K = per * kw; g.alpha = 1.0; g.beta = 0.0; g.slots = nullptr; g.nbatch = ng; g.kflags = 0; g.bm = 64;
        g.M = rk;
        if (is_x) {
          g.B = reinterpret_cast<const double*>(D + (size_t)first * sD + (size_t)k * Ts + (size_t)r0 * ldd);
          g.sB = (long long)per * sD; g.ldb = ldd; g.sBseg = sD; g.b_f32 = 1;
          g.N = T; g.mode = GEMM_FULL;
          g.C = out + (size_t)(part ? nfull : 0) * rk * T; g.sC = (long long)rk * T; g.ldc = rk;
        } else {
          g.B = g.A; g.sB = g.sA; g.ldb = rpad; g.sBseg = lw.sM;
          g.N = rk; g.mode = GEMM_LOWER;
          g.C = out + (size_t)(part ? nfull : 0) * rk * rk; g.sC = (long long)rk * rk; g.ldc = rk;
        }
        CHK(gemm(c, false, g));
      }
      return 0;
    };
    CHK(seg(spsS, false, Spart));
    // (the S sums stay with the general kernel: through cross_term_kernel<.., double, true> - lower 16 x 16 tiles only - they ran no faster)
    if (c->cross_kernel && c->mfma && rk % 16 == 0 && kw % 16 == 0) {
      // the cross term with (up to 128) rows of the latent in one workgroup (split.h): no padded row tiles on the matrix cores
      for (int row0 = 0; row0 < rk; row0 += 128) {
        CrossArgs ca{};
        ca.A = A0; ca.sM = lw.sM; ca.lda = rpad;
        ca.D = D + (size_t)k * Ts + (size_t)r0 * ldd; ca.sD = sD; ca.ldd = ldd;
        ca.C = Xpart; ca.sC = (long long)rk * T;
        ca.rk = std::min(128, rk - row0); ca.T = T; ca.kw = kw; ca.nslots = nb; ca.sps = sps;
        ca.row0 = row0; ca.ldc = rk;
        prof_begin(c, TAG_GEMM, 2.0 * (double)nb * ca.rk * (double)kw * T);
        cross_term_launch<float, false>(ca, dim3((T + 63) / 64, ngroups), c->st);
        prof_end(c);
      }
      HIPC(hipGetLastError());
    } else {
      CHK(seg(sps, true, Xpart));
    }
    const int ngS = (nb + spsS - 1) / spsS;
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)(((size_t)rk * rk + 63) / 64)), dim3(256), 0, c->st, Spart, ngS, rk, rk, 32, Ssum);
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)(((size_t)rk * T + 63) / 64)), dim3(256), 0, c->st, Xpart, ngroups, rk, T, 0, Xsum);
    const double* Fk = c->Flr + (size_t)k * Tp * Tp;
    GemmP z{};                                                                  // Z = F_k S_k   (T x r_k)
    z.A = Fk; z.lda = Tp; z.B = Ssum; z.ldb = rk; z.C = Zb; z.ldc = T;
    z.M = T; z.N = rk; z.K = rk; z.alpha = 1.0; z.beta = 0.0; z.nbatch = 1; z.mode = GEMM_FULL;
    CHK(gemm(c, true, z));
    GemmP t1 = z;                                                               // T1 = Z F_k^T  (T x T)
    t1.A = Zb; t1.lda = T; t1.B = Fk; t1.ldb = Tp; t1.C = T1 + (size_t)k * tt; t1.ldc = T; t1.N = T;
    CHK(gemm(c, false, t1));
    GemmP xf = z;                                                               // Xfull = F_k X_k  (T x T)
    xf.B = Xsum; xf.ldb = rk; xf.C = Xfull + (size_t)k * tt; xf.ldc = T; xf.N = T;
    CHK(gemm(c, true, xf));
  }
  // 4. Pacc += eps diag + T1 - Xfull - Xfull^T + sum of the FP16 partial sums
  {
    const int nt64 = (T + PACC_TS - 1) / PACC_TS;
    hipLaunchKernelGGL(pacc_split_reduce_kernel, dim3(nt64 * (nt64 + 1) / 2, p), dim3(256), 0, c->st, T1, Xfull, c->ppart, ngroups, c->Gbin, sW, nb, c->eps, T, Tp,
                       p, c->Pacc);
  }
  HIPC(hipGetLastError());
  c->pacc_used = true;
  return 0;
}

// Covariance blocks through the low-rank form of the prior (see model.h): per slot an r x r SPD system
// B = I + F^T Wt F instead of the n x n Hessian.  Uses the dense engine's slabs as scratch (ld = rpad views).
// logdet_out (optional, host, nb entries): log det of the posterior precision K^-1 + scatter(W) of every slot,
//   = -sum_k log det K_k + sum_t log det(I + eps W_t) + log det(I + F^T Wt F)   (Sylvester; K_k = eps I + F_k F_k^T)
int posterior_blocks_lowrank(pgpfa_ctx* c, int nb, bool want_vsmgp, bool accumulate, double* logdet_out) {
  const int T = c->T, p = c->p, Tp = c->Tp, pp = p * p;
  const int rpad = c->rpad;
  const int ract = round_up(c->rtot, 16);          // columns of Yt that are not identically zero (rpad rounds to 128 for the factor)
  const long long sW = (long long)T * pp;
  c->last_cov_lowrank = true;
  // a. per-bin blocks G = (I + eps W)^-1, Wt = W G
  CHK(bin_blocks(c, c->W, sW, c->Gbin, c->Wt, sW, nb, logdet_out ? c->ldet_buf : nullptr));
  // sum-only accumulation by the split form (split.h)?  Decided by the relative size of the mixing correction of this chunk,
  // max_t eps ||Wt_t||_inf, measured here and read back just before the mixing pass (info key "last_eps_wt_norm")
  // (want_vsmgp passes run the FP64 engine whatever dual_f32 says - it only concerns the dual's evaluations - so the split form does not ask)
  const bool split_candidate = want_vsmgp && accumulate && c->split_cov && c->mfma && (p <= 16 || (p <= 20 && c->mix_wide));
  unsigned* norm_bits = reinterpret_cast<unsigned*>(c->pcg_ratio);         // (scratch word: the inner solves are over)
  if (split_candidate || c->measure_mix) {
    HIPC(hipMemsetAsync(norm_bits, 0, 4 * sizeof(unsigned), c->st));
    const long long nblk = (long long)nb * T;
    hipLaunchKernelGGL(block_norm_max_kernel, dim3((unsigned)std::min<long long>((nblk * p + 255) / 256, 2048)), dim3(256), 0, c->st, c->Wt, nblk, p, c->eps, norm_bits,
                       reinterpret_cast<double*>(norm_bits + 2));
  }
  // b. B = I + F^T Wt F into the factor slabs viewed with ld = rpad; factor; L^-T
  CholWS lw = c->ws;
  lw.ld = rpad; lw.npad = rpad; lw.nact = round_up(c->rtot, 64);
  // (compact rank offsets: the tiles of B are walked in the padded index space - rtot16 rows - and stored through cmap; build_lowrank)
  const int nblk64 = c->rank_compact ? round_up(c->rtot16, 64) / 64 : rpad / 64, npairs = nblk64 * (nblk64 + 1) / 2;
  const int* cmap = c->rank_compact ? c->d_cmap : nullptr;
  // Mixed precision (option dual_f32, dual-variational evaluations only): B, its Cholesky factor, L^-T and Yt = F L^-T - the O(T r^2) and
  // O(r^3) parts - run on the FP32 matrix cores (twice the FP64 rate, half the bytes); log det and the per-bin covariance blocks are
  // accumulated in FP64 from the single-precision factors.  (dual_f32 = 2: B is still assembled in FP64 and rounded once.)
  const bool f32 = c->dual_f32 && !want_vsmgp;
  // Yt = F L^-T: L^-T is upper triangular, so the rows of Yt that belong to latent k vanish left of column roff[k].  When the consumer knows
  // the same offsets and takes those entries as zeros without reading them (the mixing pass up to 16 latents, the matrix-core post_vsm
  // beyond 10) the product skips the whole 128-column tiles left of it: ~45 % of the flops and stores.
  const bool skip_zero_cols = want_vsmgp ? p <= 16 : (p > 10 && c->vsm_mfma);
  // granularity of that skipping: the mixing passes take any multiple of 16 (the ranks are padded to 16, so exactly the zero columns are left
  // out: 128-column tiles kept 64 of them per latent on average - 18 % of the product and of the pass at config 3), the matrix-core post_vsm whole
  // 128-column tiles
  const int ctile = (want_vsmgp && p <= 16) ? 16 : (int)GBN;
  // rows (k, t) of the Yt slab sit at k * Ts + t with Ts = T rounded up to 16 when the slab is tall enough: the 64-bin runs of the mixing pass and
  // the product's stores then start on 128-byte lines (at T = 500 every run straddled one: 1.4x the bytes fetched, PMC)
  const int Ts = (c->slab_row_align && p * round_up(T, 16) <= c->ld) ? round_up(T, 16) : T;
  CholWS lwf = lw;                                         // single-precision views: B / L in the Mt slabs, L^-T and Yt in the H slabs
  float* Ytf = nullptr;
  if (f32) {
    lwf.H = lw.Mt; lwf.sH = 2 * lw.sM;
    lwf.Mt = lw.H; lwf.sM = 2 * lw.sH;
    lwf.sD = 2 * lw.sD; lwf.sP = 2 * lw.sP;
    Ytf = reinterpret_cast<float*>(lw.H) + (size_t)rpad * rpad;
    if (!c->Flr32) {
      if (hipMalloc((void**)&c->Flr32, ((size_t)Tp * Tp * p + 256 * (size_t)Tp) * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return fail("out of device memory for the single-precision factors"); }
      c->flr32_valid = false;
    }
    if (!c->flr32_valid) {
      const size_t nf = (size_t)Tp * Tp * p;
      hipLaunchKernelGGL(cvt_f32_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, c->st, c->Flr, c->Flr32, nf);
      c->flr32_valid = true;
    }
  }
  HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int) * nb, c->st));
  if (f32 && c->dual_f32 == 1) {
    hipLaunchKernelGGL(assemble_b_kernel_t<float>, dim3(npairs, (nb + AB_SLOTS - 1) / AB_SLOTS), dim3(256), 0, c->st, reinterpret_cast<float*>(lwf.H),
                       (long long)lwf.sH, rpad, nblk64, (const float*)c->Flr32, Tp, T, p, c->d_blk_lat, c->d_blk_col, c->Wt, sW, c->ident, nb, cmap);
    if (cmap && rpad > c->rtot)
      hipLaunchKernelGGL(pad_identity_kernel<float>, dim3((rpad + 3) / 4, nb), dim3(256), 0, c->st, reinterpret_cast<float*>(lwf.H), (long long)lwf.sH, rpad, c->rtot, rpad, lw.nact, (int)NB, c->ident);
  } else {
    hipLaunchKernelGGL(assemble_b_kernel_t<double>, dim3(npairs, (nb + AB_SLOTS - 1) / AB_SLOTS), dim3(256), 0, c->st, lw.H, (long long)lw.sH, rpad,
                       nblk64, (const double*)c->Flr, Tp, T, p, c->d_blk_lat, c->d_blk_col, c->Wt, sW, c->ident, nb, cmap);
    if (cmap && rpad > c->rtot)
      hipLaunchKernelGGL(pad_identity_kernel<double>, dim3((rpad + 3) / 4, nb), dim3(256), 0, c->st, lw.H, (long long)lw.sH, rpad, c->rtot, rpad, lw.nact, (int)NB, c->ident);
  }
  HIPC(hipGetLastError());
  if (f32) {
    if (c->dual_f32 != 1)
      hipLaunchKernelGGL(cvt_lower_f32_kernel, dim3((unsigned)(((size_t)rpad * rpad + 1023) / 1024), nb), dim3(256), 0, c->st, lw.H, (long long)lw.sH,
                         reinterpret_cast<float*>(lwf.H), (long long)lwf.sH, rpad);
    c->mt_dirty = true;
    CHK(factor(c, lwf, c->ident, nb, true));
  } else {
    CHK(factor(c, lw, c->ident, nb));
  }
  if (logdet_out) {
    std::vector<double> a(nb), b2(nb);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(nb), dim3(256), 0, c->st, c->ldet_buf, T, c->sc_f);
    CHK(download(c, a.data(), c->sc_f, nb));
    if (f32)
      hipLaunchKernelGGL(logdet_batch_f32_kernel, dim3(nb), dim3(256), 0, c->st, reinterpret_cast<const float*>(lwf.H), (long long)lwf.sH, rpad, rpad,
                         c->sc_f);
    else
      hipLaunchKernelGGL(logdet_batch_kernel, dim3(nb), dim3(256), 0, c->st, lw.H, (long long)lw.sH, rpad, rpad, c->sc_f);
    CHK(download(c, b2.data(), c->sc_f, nb));
    double ldk = 0.0;
    for (double v : c->logdetK) ldk += v;
    for (int s2 = 0; s2 < nb; ++s2) logdet_out[s2] = -ldk + a[s2] + b2[s2];
  }
  c->mt_dirty = true;
  if (f32) {
    hipLaunchKernelGGL(fill_slabs_f32_kernel, dim3((unsigned)(((size_t)rpad * rpad + 1023) / 1024), nb), dim3(256), 0, c->st,
                       reinterpret_cast<float*>(lwf.Mt), (long long)lwf.sM, (size_t)rpad * rpad, 0.0f);
    CHK(inverse_t(c, lwf, c->ident, nb, true));
    // Yt (float, n x ract, ld = c->ld) behind L^-T in the same slab
    for (int k = 0; k < p; ++k) {
      const int c0 = skip_zero_cols ? (c->roff[k] / ctile) * ctile : 0;      // (see the FP64 product below)
      if (c0 >= ract) continue;
      GemmP g{};
      g.A = reinterpret_cast<const double*>(c->Flr32 + (size_t)k * Tp * Tp); g.sA = 0; g.lda = Tp;
      g.B = reinterpret_cast<const double*>(reinterpret_cast<float*>(lwf.Mt) + c->roff[k] + (size_t)c0 * rpad); g.sB = lwf.sM; g.ldb = rpad;
      g.C = reinterpret_cast<double*>(Ytf + (size_t)k * Ts + (size_t)c0 * c->ld); g.sC = lwf.sM; g.ldc = c->ld;
      g.M = T; g.N = ract - c0; g.K = c->rk[k]; g.alpha = 1.0; g.beta = 0.0;
      g.slots = c->ident; g.nbatch = nb; g.mode = GEMM_FULL; g.kflags = 0;
      CHK(gemm(c, true, g, true));
    }
    if (p > WIDE_MAX) return fail("low-rank covariance engine supports up to %d latents (p=%d)", WIDE_MAX, p);
    prof_begin(c, TAG_VSM, (double)nb * c->n * rpad * p);
    launch_post_vsm(c, (const float*)Ytf, (long long)lwf.sM, ract, nb, 1, skip_zero_cols ? c->d_roff : nullptr, Ts);
    prof_end(c);
    dispatch_pw(p, [&](auto pw) {
      constexpr int PW = decltype(pw)::value;
      if constexpr (PW <= 16) {
        constexpr int BT = 256 / PW;
        hipLaunchKernelGGL(vsm_finish_kernel<PW>, dim3((T + BT - 1) / BT, nb), dim3(256), 0, c->st, c->vsm, c->Gbin, sW, T, p, c->eps, c->ident,
                           c->trial_of_slot);
      } else {
        const int bins = wide_bins(p) / 2;
        hipLaunchKernelGGL(vsm_finish_wide_kernel, dim3((T + bins - 1) / bins, nb), dim3(bins * 32), wide_lds_bytes(p, bins, 2), c->st, c->vsm,
                           c->Gbin, sW, T, p, c->eps, c->ident, c->trial_of_slot, bins);
      }
    });
    HIPC(hipGetLastError());
    return 0;
  }
  // L^-T's slab holds whatever the last use left (another rank layout, the factor of a dense pass): clear what will be read and not written - all of it,
  // or, when every consumer starts at the latent's own columns (skip_zero_cols), the strictly lower entries of the p rectangles they read
  if (skip_zero_cols && c->mt_fill)
    hipLaunchKernelGGL(clear_lower_reads_kernel, dim3(p, nb), dim3(256), 0, c->st, lw.Mt, (long long)lw.sM, rpad, c->d_roff, ctile, c->ident);
  else
    hipLaunchKernelGGL(fill_slabs_kernel, dim3((unsigned)(((size_t)rpad * rpad + 1023) / 1024), nb), dim3(256), 0, c->st, lw.Mt, lw.sM,
                       (size_t)rpad * rpad, 0.0);
  CHK(inverse_t(c, lw, c->ident, nb));
  if (p > WIDE_MAX) return fail("low-rank covariance engine supports up to %d latents (p=%d)", WIDE_MAX, p);
  // the split form's verdict: the relative size of the mixing correction, measured when the chunk's blocks were formed
  bool split = false, decided = false;
  auto decide_split = [&]() -> int {
    decided = true;
    if (!(split_candidate || c->measure_mix)) return 0;
    unsigned hw[4] = {0u, 0u, 0u, 0u};
    CHK(dl_enqueue(c, hw, norm_bits, 4 * sizeof(unsigned)));
    CHK(dl_flush(c));
    float hv;
    double hsq;
    std::memcpy(&hv, &hw[0], sizeof(float));
    std::memcpy(&hsq, &hw[2], sizeof(double));
    const double rms = std::sqrt(hsq / std::max(1.0, (double)nb * T));
    c->info["last_eps_wt_norm"] = std::max(c->info["last_eps_wt_norm"], (double)hv);
    c->info["last_eps_wt_rms"] = std::max(c->info["last_eps_wt_rms"], rms);
    // (the precision of the split form follows the root mean square of the correction; the maximum only has to stay a contraction)
    split = split_candidate && std::isfinite(hv) && std::isfinite(rms) && rms <= c->split_max_norm && (double)hv <= 0.9;
    return 0;
  };
  // c. Yt = F Mts  (n x rpad, ld = c->ld) into the factor slab (the factor itself is dead now)
  // (column tiles left of roff[k] skipped under skip_zero_cols, see above)
  // Under the split form Yt has one consumer, the mixing pass: up to 10 latents the two run as one kernel (ytmix.h) and Yt is never written.  That
  // needs the verdict before the product is queued instead of behind it (the host then waits for the factorisation: one launch gap per chunk).
  // (compact offsets: the last column block of Yt is partly padding - identity columns of L^-T meeting no row of F: zeros)
  const bool fuse_candidate = split_candidate && c->yt_mix && c->mfma && p <= 10 && (ract == c->rtot || (c->rank_compact && ract <= lw.nact));
  if (fuse_candidate) CHK(decide_split());
  const bool fused = fuse_candidate && split;
  c->info["last_yt_mix_fused"] = fused ? 1.0 : 0.0;
  if (!fused) {
    for (int k = 0; k < p; ++k) {
      const int c0 = skip_zero_cols ? (c->roff[k] / ctile) * ctile : 0;
      if (c0 >= ract) continue;
      GemmP g{};
      g.A = c->Flr + (size_t)k * Tp * Tp; g.sA = 0; g.lda = Tp;
      g.B = lw.Mt + c->roff[k] + (size_t)c0 * rpad; g.sB = lw.sM; g.ldb = rpad;     // rows roff[k].. of Mts, K x N column-major
      g.C = lw.H + (size_t)k * Ts + (size_t)c0 * c->ld; g.sC = lw.sH; g.ldc = c->ld;
      g.M = T; g.N = ract - c0; g.K = c->rk[k]; g.alpha = 1.0; g.beta = 0.0;
      g.slots = c->ident; g.nbatch = nb; g.mode = GEMM_FULL; g.kflags = 0;
      CHK(gemm(c, true, g));
    }
  }
  if (!decided) CHK(decide_split());
  c->info["last_split_cov"] = split ? 1.0 : 0.0;
  if (want_vsmgp && split) {
    CHK(accumulate_split(c, lw, nb, ract, Ts, skip_zero_cols, ctile, fused));
  } else if (want_vsmgp) {
    // d+e. one pass over Yt: post_vsm[t] = eps G_t + sum_b (G_t y_b)(G_t y_b)^T, and Yt is mixed in place (y <- G_t y) so that
    //      rows (k,.) of the slab become Ymix_k, the GEMM operand of post_vsmGP_k = eps diag(G_t[k][k]) + Ymix_k Ymix_k^T
    prof_begin(c, TAG_MIX, (double)nb * c->n * ract * 16.0);           // (bytes: Yt read and written in place)
    dispatch_pw(p, [&](auto pw) {
      constexpr int PW = decltype(pw)::value;
      if constexpr (PW <= 16) {
        hipLaunchKernelGGL(mix_vsm_kernel<PW>, dim3((T + 63) / 64, nb), dim3(256), 0, c->st, lw.H, lw.sH, c->ld, c->Gbin, sW, T, p, ract, c->eps,
                           c->vsm, c->ident, c->trial_of_slot, c->d_roff, ctile, Ts);
      } else if (c->mix_wide && p <= 20) {
        hipLaunchKernelGGL((mix_vsm_wide2_kernel<20, false>), dim3((T + 63) / 64, nb), dim3(256), 0, c->st, lw.H, (long long)lw.sH, c->ld, c->Gbin, sW, T, p, ract, c->eps,
                           c->vsm, c->ident, c->trial_of_slot, Ts, c->sink, (float*)nullptr, 0LL, 0);
      } else {
        const int bins = wide_bins(p);
        hipLaunchKernelGGL(mix_vsm_wide_kernel, dim3((T + bins - 1) / bins, nb), dim3(bins * 32), wide_lds_bytes(p, bins, 1), c->st, lw.H, lw.sH,
                           c->ld, c->Gbin, sW, T, p, ract, c->eps, c->vsm, c->ident, c->trial_of_slot, bins, Ts);
      }
    });
    prof_end(c);
  } else {
    // d. post_vsm[t] = eps G_t + G_t (Y_t^T Y_t) G_t
    prof_begin(c, TAG_VSM, (double)nb * c->n * rpad * p);
    launch_post_vsm(c, (const double*)lw.H, (long long)lw.sH, ract, nb, 1, skip_zero_cols ? c->d_roff : nullptr, Ts);
    prof_end(c);
    dispatch_pw(p, [&](auto pw) {
      constexpr int PW = decltype(pw)::value;
      if constexpr (PW <= 16) {
        constexpr int BT = 256 / PW;
        hipLaunchKernelGGL(vsm_finish_kernel<PW>, dim3((T + BT - 1) / BT, nb), dim3(256), 0, c->st, c->vsm, c->Gbin, sW, T, p, c->eps, c->ident,
                           c->trial_of_slot);
      } else {
        const int bins = wide_bins(p) / 2;                 // two staged blocks per bin
        hipLaunchKernelGGL(vsm_finish_wide_kernel, dim3((T + bins - 1) / bins, nb), dim3(bins * 32), wide_lds_bytes(p, bins, 2), c->st, c->vsm,
                           c->Gbin, sW, T, p, c->eps, c->ident, c->trial_of_slot, bins);
      }
    });
  }
  if (want_vsmgp && !split) {
    if (accumulate) {
      // sum-only output: Pacc[k] += sum over the chunk's slots of Ymix_k Ymix_k^T as ONE split-K product per launch -
      // batch = (latent, group of `sps` consecutive slots), the K dimension walks the rpad-wide panels of the
      // group's slabs - followed by a reduction of the partial products (+ the eps G_t[k][k] diagonals)
      const int sps = std::max(1, (nb + PACC_SPLITS - 1) / PACC_SPLITS);
      const int nfull = nb / sps, rem = nb - nfull * sps, nsplit = nfull + (rem ? 1 : 0);
      auto launch = [&](int first_slot, int slots_per, int ngroups, int part_first) -> int {
        GemmP g{};
        g.A = lw.H + (size_t)first_slot * lw.sH; g.sA = (long long)slots_per * lw.sH; g.lda = c->ld;
        g.B = g.A; g.sB = g.sA; g.ldb = c->ld;
        g.C = c->ppart + (size_t)part_first * T * T; g.sC = (long long)T * T; g.ldc = T;
        g.M = T; g.N = T; g.K = slots_per * ract; g.alpha = 1.0; g.beta = 0.0;
        g.slots = nullptr; g.nb_lo = ngroups; g.nbatch = ngroups * p;
        g.sA_hi = Ts; g.sB_hi = Ts; g.sC_hi = (long long)nsplit * T * T;
        g.kseg = ract; g.sAseg = lw.sH; g.sBseg = lw.sH;    // ract columns of each slab, slabs sH apart
        g.mode = GEMM_LOWER; g.kflags = 0;
        return gemm(c, false, g);
      };
      if (nfull) CHK(launch(0, sps, nfull, 0));
      if (rem) CHK(launch(nfull * sps, rem, 1, nfull));
      hipLaunchKernelGGL(pacc_reduce_kernel, dim3(T, p), dim3(128), 0, c->st, c->ppart, nsplit, c->Gbin, sW, nb, c->eps, T, Tp, p, c->Pacc);
      c->pacc_used = true;
    } else {
      CHK(ensure_vsmgp_buffer(c));
      const size_t off_stage = (size_t)c->ld * rpad + (size_t)Tp * rpad;
      for (int k = 0; k < p; ++k) {
        GemmP g{};
        g.A = lw.H + (size_t)k * Ts; g.sA = lw.sH; g.lda = c->ld;
        g.B = g.A; g.sB = lw.sH; g.ldb = c->ld;
        g.C = lw.H + off_stage; g.sC = lw.sH; g.ldc = T;
        g.M = T; g.N = T; g.K = ract; g.alpha = 1.0; g.beta = 0.0;
        g.slots = c->ident; g.nbatch = nb; g.mode = GEMM_LOWER; g.kflags = 0;
        CHK(gemm(c, false, g));
        hipLaunchKernelGGL(scatter_vsmgp_lr_kernel, dim3((unsigned)(((size_t)T * T + 255) / 256), nb), dim3(256), 0, c->st, lw.H + off_stage, lw.sH, T,
                           c->vsmgp, T, p, k, c->Gbin, sW, c->eps, c->trial_of_slot);
      }
    }
  }
  HIPC(hipGetLastError());
  return 0;
}

int posterior_blocks(pgpfa_ctx* c, int nb, double diag_scale, bool want_vsmgp, bool accumulate) {
  if (c->plan_lowrank) {
    if (diag_scale != 1.0) return fail("internal: jittered covariance requested under the low-rank workspace plan");
    return posterior_blocks_lowrank(c, nb, want_vsmgp, accumulate);
  }
  return posterior_blocks_dense(c, nb, diag_scale, want_vsmgp);
}


static int get_rows(pgpfa_ctx* c, int n, const int32_t* idx, const double* src, size_t len, double* out) {
  if (!c || !out) return fail("null argument");
  HIPC(hipSetDevice(c->device));
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr));
  for (size_t i = 0; i < tr.v.size(); ++i)
    CHK(dl_enqueue(c, out + i * len, src + (size_t)tr.v[i] * len, len * sizeof(double)));
  CHK(dl_flush(c));
  return 0;
}
int pgpfa_get_post_mean(pgpfa_ctx* c, int n, const int32_t* idx, double* out) { return get_rows(c, n, idx, c ? c->Xmode : nullptr, c ? (size_t)c->n : 0, out); }
int pgpfa_get_post_vsm(pgpfa_ctx* c, int n, const int32_t* idx, double* out) {
  return get_rows(c, n, idx, c ? c->vsm : nullptr, c ? (size_t)c->T * c->p * c->p : 0, out);
}

// Rebuild the per-trial T x T blocks of trials whose last E-step ran sum-only: covariance blocks at the resident
// modes, under the parameters of that E-step (restored around the call when an M-step has moved on since).

// (dual: the trials' posterior is the dual-variational one - curvature blocks W_t = C^T diag(lambda_t) C from the kept lambda, with the
// reference's jitter, instead of the Laplace curvature at the mode)
static int materialize_impl(pgpfa_ctx* c, const std::vector<int>& need, bool dual) {
  CHK(ready_estep(c, dual ? c->dual_lowrank : true));
  if (dual) CHK(ensure_lambda(c));
  const int N = (int)need.size();
  std::vector<int> info(c->B);
  const size_t m = (size_t)c->q * c->T;
  for (int c0 = 0; c0 < N; c0 += c->B) {
    const int nb = std::min(c->B, N - c0);
    std::vector<int> tos(need.begin() + c0, need.begin() + c0 + nb);
    CHK(upload_list(c, c->trial_of_slot, tos));
    HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int) * nb, c->st));
    if (dual) {
      for (int s = 0; s < nb; ++s)
        HIPC(hipMemcpyAsync(c->lamd + (size_t)s * m, c->lam_keep + (size_t)tos[s] * m, m * sizeof(double), hipMemcpyDeviceToDevice, c->st));
      std::vector<double> sB, sD, vKv;
      CHK(dual_common(c, nb, &sB, &sD, &vKv));
      if (c->plan_lowrank) { CHK(dual_jitter(c, nb)); CHK(posterior_blocks(c, nb, 1.0, true, false)); }
      else CHK(posterior_blocks(c, nb, 1.0 + 1e-6, true, false));
    } else {
      hipLaunchKernelGGL(gather_rows_kernel, dim3((c->n + 255) / 256, nb), dim3(256), 0, c->st, c->Xmode, c->n, c->Xc, (long long)c->ld,
                         c->trial_of_slot, 0);
      CHK(poisson(c, c->ident, nb, c->Xc, c->Gl, c->W, c->sc_f, 1));
      CHK(posterior_blocks(c, nb, 1.0, true, false));
    }
    CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
    CHK(dl_flush(c));
    for (int s = 0; s < nb; ++s) {
      if (info[s] != 0) return fail("posterior precision of trial %d is not positive definite at the resident mode", tos[s]);
      c->vsmgp_ok[tos[s]] = 1;
    }
  }
  return 0;
}

int ensure_trial_vsmgp(pgpfa_ctx* c, const std::vector<int>& trials) {
  // trials whose blocks are not resident, grouped by the E-step (parameter snapshot) that produced their posterior
  std::map<std::pair<int, int>, std::vector<int>> need;
  for (int t : trials) {
    if (c->vsmgp_ok[t]) continue;
    if (c->trial_snap[t] < 0) return fail("post_vsmGP of trial %d is not resident: no E-step or pgpfa_set_posterior produced it", t);
    std::vector<int>& v = need[std::make_pair(c->trial_snap[t], (int)c->trial_dual[t])];
    if (std::find(v.begin(), v.end(), t) == v.end()) v.push_back(t);
  }
  for (auto& kv : need) {
    const std::vector<int>& v = kv.second;
    const bool dual = kv.first.second != 0;
    CHK(with_snapshot(c, kv.first.first, [&]() { return materialize_impl(c, v, dual); }));
  }
  return 0;
}

int pgpfa_get_post_vsmgp(pgpfa_ctx* c, int n, const int32_t* idx, double* out) {
  if (!c || !out) return fail("null argument");
  HIPC(hipSetDevice(c->device));
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr));
  CHK(ensure_vsmgp_buffer(c));
  CHK(ensure_trial_vsmgp(c, tr.v));
  const size_t len = (size_t)c->T * c->T * c->p;
  double* tmp = nullptr;
  HIPC(hipMalloc((void**)&tmp, len * sizeof(double)));
  for (size_t i = 0; i < tr.v.size(); ++i) {
    hipLaunchKernelGGL(vsmgp_to_ref_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->st, c->vsmgp + (size_t)tr.v[i] * len, tmp, c->T, c->p);
    hipMemcpyAsync(out + i * len, tmp, len * sizeof(double), hipMemcpyDeviceToHost, c->st);
    hipStreamSynchronize(c->st);
  }
  hipFree(tmp);
  HIPC(hipGetLastError());
  return 0;
}

static int post_cov_impl(pgpfa_ctx* c, int trial, double* out) {
  CHK(ready(c));
  std::vector<int> tr{trial};
  CHK(upload_list(c, c->trial_of_slot, tr));
  HIPC(hipMemsetAsync(c->ws.info, 0, sizeof(int), c->st));
  hipLaunchKernelGGL(gather_rows_kernel, dim3((c->n + 255) / 256, 1), dim3(256), 0, c->st, c->Xmode, c->n, c->Xc, (long long)c->ld, c->trial_of_slot, 0);
  CHK(poisson(c, c->ident, 1, c->Xc, c->Gl, c->W, c->sc_f, 1));
  CHK(ensure_mt_clean(c));
  CHK(assemble(c, c->ident, 1));
  CHK(factor(c, c->ws, c->ident, 1));
  CHK(inverse_t(c, c->ws, c->ident, 1));
  GemmP g{};
  g.A = c->ws.Mt; g.sA = c->ws.sM; g.lda = c->ld;
  g.B = c->ws.Mt; g.sB = c->ws.sM; g.ldb = c->ld;
  g.C = c->ws.H; g.sC = c->ws.sH; g.ldc = c->ld;
  g.M = c->npad; g.N = c->npad; g.K = c->npad; g.alpha = 1.0; g.beta = 0.0;
  g.slots = c->ident; g.nbatch = 1; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
  CHK(gemm(c, false, g));
  HIPC(hipMemcpy2DAsync(out, (size_t)c->n * sizeof(double), c->ws.H, (size_t)c->ld * sizeof(double), (size_t)c->n * sizeof(double), c->n,
                        hipMemcpyDeviceToHost, c->st));
  HIPC(hipStreamSynchronize(c->st));
  return 0;
}

// Dense posterior covariance of one trial at its resident mode, under the parameters of the E-step that produced that mode (they are
// restored around the call when later calls have moved the context on).
int pgpfa_get_post_cov(pgpfa_ctx* c, int trial, double* out) {
  if (!c || !out) return fail("null argument");
  if (trial < 0 || trial >= c->R) return fail("trial %d out of range", trial);
  if (!c->have_params) return fail("set_params has not been called");
  return with_snapshot(c, c->trial_snap[trial], [&]() { return (c->trial_dual[trial] && c->lam_keep) ? post_cov_dual_impl(c, trial, out) : post_cov_impl(c, trial, out); });
}

int pgpfa_set_posterior(pgpfa_ctx* c, int n, const int32_t* idx, const double* post_mean, const double* post_vsm, const double* post_vsmgp) {
  if (!c || !post_mean || !post_vsm) return fail("null argument");
  HIPC(hipSetDevice(c->device));
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  const size_t lm = c->n, lv = (size_t)c->T * c->p * c->p, lg = (size_t)c->T * c->T * c->p;
  double* tmp = nullptr;
  if (post_vsmgp) CHK(ensure_vsmgp_buffer(c));
  if (post_vsmgp) HIPC(hipMalloc((void**)&tmp, lg * sizeof(double)));
  for (size_t i = 0; i < tr.v.size(); ++i) {
    const size_t r = tr.v[i];
    hipMemcpyAsync(c->Xmode + r * lm, post_mean + i * lm, lm * sizeof(double), hipMemcpyHostToDevice, c->st);
    hipMemcpyAsync(c->vsm + r * lv, post_vsm + i * lv, lv * sizeof(double), hipMemcpyHostToDevice, c->st);
    if (post_vsmgp) {
      hipMemcpyAsync(tmp, post_vsmgp + i * lg, lg * sizeof(double), hipMemcpyHostToDevice, c->st);
      hipLaunchKernelGGL(vsmgp_from_ref_kernel, dim3((unsigned)((lg + 255) / 256)), dim3(256), 0, c->st, tmp, c->vsmgp + r * lg, c->T, c->p);
      hipStreamSynchronize(c->st);
    }
  }
  hipStreamSynchronize(c->st);
  if (tmp) hipFree(tmp);
  HIPC(hipGetLastError());
  for (int t : tr.v) { c->vsmgp_ok[t] = 1; c->mode_serial[t] = -10; c->trial_snap[t] = -1; c->trial_dual[t] = 0; }   // whatever the caller provided (or left) is the resident value
  return remember_trials(c, tr.v);
}


