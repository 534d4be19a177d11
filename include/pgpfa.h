/*
 * pgpfa.h - C-ABI of libpgpfa_hip.so: the MI355X (gfx950) implementation of the
 * Poisson-GPFA EM hot path.
 *
 * The reference (mackelab/poisson-gpfa) is pure Python and has no FFI; the boundary it
 * exposes is the call surface of funs/inference.py, funs/learning.py and funs/engine.py.
 * Every entry point below names the reference function (file:line under the reference
 * tree) whose arithmetic it replaces.  The host mirror of that call surface lives in
 * poisson-gpfa_amd/funs/ and binds these symbols with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; all host arrays are caller-owned, C-contiguous.
 *   - every function returns 0 on success, non-zero on failure; pgpfa_last_error()
 *     returns the message of the last failure on the calling thread.
 *   - doubles everywhere (the reference computes in float64); spike counts are packed
 *     to uint8 on upload (counts above 255 are rejected).
 *   - one context = one GPU = one host thread at a time; no callbacks into the caller.
 *   - trial index lists are int32, relative to the count tensor uploaded into the
 *     context; a NULL list means "all trials".
 *   - sizes: q neurons (ydim), p latents (xdim), T bins, R trials, n = p*T.
 *   - vector layouts are the reference's: xbar[k*T+t] = X[k][t] (inference.py:129),
 *     vecCd = [C[:,0],...,C[:,p-1], d] (util.py:560-574).
 */
#ifndef PGPFA_H
#define PGPFA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgpfa_ctx pgpfa_ctx;

/* ---- library ------------------------------------------------------------------ */
const char* pgpfa_last_error(void);
int pgpfa_version(void);
int pgpfa_device_count(int* count);

/* ---- context ------------------------------------------------------------------ */
/* One fit = one context: sizes of experiment.data (engine.py:131-135), binSize in ms. */
int pgpfa_create(pgpfa_ctx** out, int device, int q, int p, int T, int R, double bin_ms);
int pgpfa_destroy(pgpfa_ctx* ctx);
/* Options: "newton_xtol" (1e-5), "newton_max_iter" (50), "use_mfma" (1),
 * "chunk_trials" (0 = auto), "eps_noise" (1e-3, util.py:599), "chord" (1: reuse the first factor for
 * chord steps), "chord_xtol" (1e-9), "chord_rho" (0.6), "chord_max_step" (1.0), "profile" (0; 1 = HIP events around
 * every tagged launch, 2 = around GEMM launches and the one mixing (tag "mix") launch of an E-step only; sums are read with pgpfa_get_info "prof_<family>_ms|_flops|_launches", the longest
 * single launch with "_max_ms|_max_flops"), "profile_pause" (1: stop recording without touching the sums, 0: go on - an event pair costs
 * ~10 us of device time per launch, so a caller that wants rates over a long region samples it),
 * "shared_pcg" (1: phase-1 Newton with the shared preconditioner), "shared_min" (16), "pcg_inner" (16: cap on
 * the inner PCG iterations of one outer Newton iteration), "pcg_eta0" (1e-2: relative residual of the first inner solve;
 * later ones adapt to the predicted error), "splitk_target" (1280: thin GEMMs are cut along k until about this many
 * workgroups are in flight),
 * "pcg_outer_max" (12), "cov_mode" (0 auto, 1 dense, 2 low-rank covariance engine), "lowrank_tol" (1e-10: stopping residual of the pivoted Cholesky behind the low-rank form K = eps I + F F^T; the covariance blocks
 * come out accurate to ~2x this value - measured against the reference at config 3 - the modes and the objective do not depend on it;
 * 1e-13 makes the form exact to rounding at ~10 % more rank),
 * "keep_trial_vsmgp" (0: the low-rank engine accumulates sum_r post_vsmGP_r for the tau M-step and rebuilds
 * per-trial T x T blocks only when pgpfa_get_post_vsmgp asks for them; 1: store them in every E-step),
 * "dual_lowrank" (1: the dual-variational entry points use the low-rank engine when it pays - log det through the r x r system; the
 * reference's 1e-6 diagonal jitter, inference.py:190, is carried by the per-bin blocks, so both engines evaluate the reference's
 * function; 0: always the dense engine),
 * "dual_f32" (0; 1: with dual_lowrank, the r x r system B = I + F^T Wt F, its factorisation, its inverse and the Yt product run in single
 * precision on the FP32 matrix cores, log det / covariance blocks / gradient accumulated in FP64: the mixed-precision form BASELINE
 * config 5 asks for; 2: the same with B assembled in FP64 and rounded once),
 * "pcg_fused" (1: inner PCG iterations without host round trips, pcg.h), "pcg_w32" (1: packed FP32 curvature blocks in the PCG
 * Hessian-vector product), "cd_mfma" (1: (C,d) sweep on the matrix cores, mstep.h), "cd_hess_mfma" (1: the Newton pass
 * of the (C,d) M-step - cost, gradient, per-neuron Hessians - on the matrix cores up to 10 latents; 0: the vector kernel), "vsm_mfma" (1: beyond 10 latents the per-bin
 * covariance blocks are Gram products on the matrix cores - post_vsm_mfma_kernel, model.h; 0: vector kernel),
 * "extrapolate_start" (1: a warm-started E-step begins at m + beta (m - m_prev) for trials whose two previous
 * E-steps are resident), "extrapolate_beta" (1.0), "extrapolate_guard" (3.0, round 6: that extrapolation is only used while the parameter step
 * between the last two Laplace E-steps - info "last_param_step_prev" - was at most this many times the step being taken - "last_param_step";
 * behind a JUMP of the parameters the difference of the two previous modes is the jump's effect, not a trend; 0: no test),
 * "start_guard" (1, round 6: a warm start whose objective is above that of x = 0 - the resident mode belongs to other loadings - restarts at
 * zero; info "last_cold_restarts"),
 * "pcg_form" (2, round 5: as 1 with the solve's private vectors on line-aligned latent rows (T rounded up to 16 doubles), one start kernel for gradient / residual /
 * first per-bin application, the closing of a step inside kernel A of the next one and one upload per solve - needs "thin_products";
 * 1: a step of the host-free inner iteration is two tile-parallel kernels - the one-reduction (Chronopoulos-Gear) form of
 * PCG - plus the three preconditioner products, the prior mat-vec gone from the loop through Kt^-1 z = r - Wb z, pcg.h; up to 10 latents, needs "pcg_w32" and
 * "pcg_retire"; 0: the split kernels of round 3 with K^-1 p as a product),
 * "pcg_vec32" (1, with "pcg_form" 2: the vectors of an inner solve that carry no accumulated state - z, s, p, q and the preconditioner's t / y -
 * are stored in single precision, x, r, every product and dot stay FP64: a slot-step moves 13.8 n-vector equivalents instead of 19.8; 0: all FP64),
 * "pcg_rx32" (1, round 6, with "pcg_vec32", up to 10 latents: the residual r and the step x of an inner solve are stored in single precision as
 * well - every product, dot and update still in FP64 from the widened values; the step is widened into the caller's FP64 vector once per solve),
 * "pcg_adapt" (1: the launches of a step of that iteration are sized by the live count the device last mirrored to the host - it only falls during a
 * solve - and the per-bin kernels take 16 / 8 / 4 slots per workgroup above 640 / 320 / below; 0: sized by the solve's first count, 16 slots),
 * "pcg_xcd" (1: the two per-bin kernels of that step map workgroup ids so that the bin tiles of a slot group run on one XCD - they share the
 * boundary lines of rows that are not line-aligned and the slot's scalars in one L2; 0: bin tile = fast grid index),
 * "pcg_retire" (1: every slot of the inner PCG has its own forcing term and leaves the iteration when it reaches it - device-side
 * live list, pcg.h; 0: one common forcing term), "pcg_trace" (0; 1: one stderr line per inner solve),
 * "time_newton" (0; 1: HIP events around the inner solves -> info "last_newton_solve_ms" / "last_newton_solve_bytes"),
 * "small_tile_below" (2^30: products with fewer 128 x 128 tiles than this run on 64 x 64 workgroup tiles - i.e. all; 0: never),
 * "f32_tile64" (1: the single-precision products of the mixed-precision dual evaluation run on 64 x 64 workgroup tiles like the FP64 ones; 0: 128 x 128 only),
 * "splitk_below64" (400: products on 64 x 64 tiles are cut along k only below this many tiles),
 * "copy_kernels" (1: read-backs and uploads up to 256 KB move through host-mapped staging memory as one-workgroup kernels, and a flush is a
 * kernel that raises a sequence number the host spins on - no hipMemcpyAsync / hipStreamSynchronize on those paths; 0: the runtime's copies),
 * "poisson_tiles" (2, round 5: 16-bin tiles per wave of the matrix-core Poisson pass up to 10 latents - the table fragments of a neuron tile, read from
 * L2, serve that many tiles; 1: one tile per wave),
 * "vsm_b4" (2, round 5; 1: the same with 32 bins per workgroup and scalar loads: the per-bin covariance blocks post_vsm for 11..20 latents on the 4 x 4 x 4 block shape of the FP64 matrix cores - four bins per
 * instruction, instructions over the lower pairs of four-latent blocks: 15 x 19 cycles per four bins and four columns at 20 latents against 12 x 68 on
 * the 16 x 16 x 4 shape padded to 32 rows; 0: that form),
 * "pivchol_pairs" (1, round 5: beyond 256 bins the pivoted Cholesky of the Gram matrices runs with two bins per row thread and four column groups -
 * half the dependent memory round trips per step, the pivot search on the diagonal while it is in registers; same pivots; 0: rbf_pivchol_kernel),
 * "yt_mix" (1, round 5: under the split covariance form and up to 10 latents the product Yt = F L^-T and the mixing pass run as ONE kernel - tiles of all
 * latents on the FP64 matrix cores, mixed in registers, only the correction D and post_vsm leave the chip; Yt is never written; 0: product, then mixing pass),
 * "yt_mix_dbg" (0; bit mask for timing experiments on that kernel - 1 no loads of F, 2 no products, 4 no mixing, 8 no stores of D, 16 no panel staging, 32 no
 * barriers, 64 no panel loads: the results are WRONG when it is set; tools/ab_opts.sh only),
 * "syrk_tile" (256, round 5: 256 x 256 workgroup tiles in the FP16 term of the split covariance sum where there are more than 256 bins and the strides
 * allow 16-byte loads - every output then costs half the reads of the correction D; 128: the 128 x 128 kernel),
 * "syrk_dbg" (0; bit mask for timing experiments on syrk_f16x2_kernel - 1 no conversions, 2 no products, 8 no barriers: the results are WRONG when it is
 * set; tools/ab_opts.sh only),
 * "mix_slot" (3, round 5: as 2 with a workgroup of 128 bins x 2 column halves - both halves read one G_t image of 56 KB, two workgroups = two waves per SIMD on a CU -
 * the halves' pair sums meet through LDS at the end; 2: as 1 with the next pair of columns requested before this pair's arithmetic - two register sets, no branch in the loop - where
 * p is a template width and the rank a multiple of 4, else 1; 1: the mixing pass of that split form with a thread per bin and a workgroup per (slot, 256 bins) that walks whole columns of
 * the slab - contiguous 2-KB runs instead of 512-byte pieces, no LDS; up to 10 latents; 0: mix_vsm_split_kernel, 64 bins x 4 columns),
 * "thin_products" (2: the three products of the low-rank preconditioner application - the block-diagonal F^T t and F v, and Sb u - as kernels of
 * their own that feed the matrix cores straight from global memory, csrc/thin.h; 1: the two block-diagonal ones only; 0: products of the general
 * GEMM kernel, block-sparse and with split-K),
 * "overlap_factors" (1: in pgpfa_set_params the pivoted Cholesky of the Gram matrices runs on a side stream next to the Gram inverses; 0: one stream),
 * "mt_fill" (1: before a slot's L^-T is formed in the low-rank covariance engine only the entries its consumers read below the diagonal are cleared -
 * the strictly lower part of p rectangles of r_k rows - where every consumer starts at the latent's own columns; 0: the whole rpad x rpad slab),
 * "mix_wide" (1: the in-place mixing pass y <- G_t y of the full-width covariance product at 17..20 latents with the lanes along the bins - 64 bins x 4 waves
 * per workgroup, G_t's rows in registers, y and G_t y exchanged through LDS, csrc/model.h mix_vsm_wide2_kernel; 0: mix_vsm_wide_kernel, lanes along the latents),
 * "cross_kernel" (1: the cross term of that split form in a kernel with all rows of a latent in one workgroup, 128 rows per
 * launch; 0: through the general GEMM kernel),
 * "split_cov" (1: the sum over trials of the T x T covariance blocks by the exact split form of csrc/split.h (up to 20 latents; 17..20 need "mix_wide") - FP64 cross term, FP16
 * two-half product for the second-order term - while the root mean square of eps ||Wt_t|| stays below "split_max_norm" (0.07);
 * 0: always the full-width FP64 product), "measure_mix" (0; 1: measure eps ||Wt_t|| in every covariance pass, also where the split form is not a candidate -> info
 * "last_eps_wt_norm" / "last_eps_wt_rms": the maximum over the chunks of the LAST pgpfa_estep_laplace / pgpfa_dual_finalize call of the
 * largest and of the root-mean-square value over (trial, bin); reset at the start of those calls),
 * "workspace_headroom" (2.0: a low-rank workspace plan leaves room for the ranks to grow by this factor before it is re-made),
 * "workspace_grow_budget_ms" (200, round 6: a RE-plan stops mapping memory into the arena after this long once the chunk has what
 * "workspace_grow_floor_slots" (128) slots need, and runs the E-step in balanced chunks of the slots that fit - mapping pages another process has
 * just released costs up to 40 ms per GB on this stack; the first plan of a context is not bounded; 0: no limit),
 * "rank_gran" (4 - the default -, 8 or 16, round 6; also the environment variable PGPFA_RANK_GRAN at pgpfa_create: the ranks of the latents' low-rank factors are
 * rounded up to this many columns in the r x r system of the low-rank engine, in L^-T and in the columns of Yt / D - 4 and 8: COMPACT offsets, the
 * products with F_k still take the rank rounded up to 16 rows and meet zero columns of F_k; needs "thin_products" 2, falls back to 16 otherwise),
 * "workspace_pool" (1: a new context attaches the arena - address range and mapped memory - a closed context of the process left behind; 0: it
 * reserves and maps its own; before the first E-step),
 * "workspace_vmm" (1: the chunk workspace is a reserved address range that grows by mapping memory; 0: plain allocations; before
 * the first E-step), "workspace_granule_mb" (1024: size of the mapped chunks),
 * "chord_max" (most chord steps of the per-trial fallback Newton on one factor), "slab_row_align" (1: rows of latent k of the low-rank
 * slab start on 128-byte lines), "dual_gemm" (1: the neuron contractions of the dual evaluation as GEMMs against a pair / loading table),
 * "cd_debug" (0; measurement only: bit switches that drop the exp / the products / the staging of the (C,d) kernels, tools/cd_probe.py). */
int pgpfa_set_option(pgpfa_ctx* ctx, const char* key, double value);
/* Info: "chunk_trials", "plan_lowrank", "n_pad", "lowrank_rtot", "last_estep_ms", "last_newton_factorizations",
 * "last_newton_solves", "last_pcg_iterations", "last_shared_factorizations", "last_cov_lowrank",
 * "last_dense_retries", "hbm_bytes_allocated", "hbm_bytes_free" / "hbm_bytes_total" (hipMemGetInfo of the context's device, now), "n_trials_global", "prof_<tag>_{ms,flops,launches}" (tags gemm, potrf, solve, poisson, assemble, vsm, cd, mix; "prof_mix_flops" counts BYTES for the stand-alone mixing
 * passes and FLOPs - products + mixing - when "last_yt_mix_fused" is 1), "counts_two_bytes",
 * "arena_bytes", "last_split_cov", "last_yt_mix_fused" (1 when the last covariance pass ran product and mixing as one kernel), "last_eps_wt_norm", "last_eps_wt_rms", "last_newton_solve_ms", "last_newton_solve_bytes",
 * "last_newton_solve_bytes_moved" (what the step's kernels really move: with "pcg_vec32" five of its vectors are single precision),
 * "last_newton_solve_bytes_survey" (the same slot-iterations priced at q T + 8 (2 p T + T p^2) bytes each: SURVEY 8(d)'s B_E per pass per trial),
 * "arena_vmm_failed" (1 once the virtual-memory arena fell back to plain allocations), "last_newton_max_iter", "last_dual_evaluations",
 * "last_loo_unconverged"; round 6, diagnostics of the last pgpfa_estep_laplace call: "last_param_step" / "last_param_step_prev" (relative
 * displacement of the parameters between consecutive Laplace E-steps: the largest of |dC| / |C|, |dd| / |d|, |d log tau|; -1: unknown),
 * "last_cold_restarts", "last_retry_ms" (time of the dense retry pass), "last_fallback_no_descent" / "last_fallback_line_search" /
 * "last_fallback_outer_cap" (slots the shared-preconditioner Newton phase gave up on, by reason); of the context: "plans" and "plan_ms_total"
 * (workspace plans made and their time), "arena_grow_ms_total" (of it: mapping memory), "set_params_calls". */
int pgpfa_get_info(pgpfa_ctx* ctx, const char* key, double* value);

/* ---- data ---------------------------------------------------------------------- */
/* experiment.data[r]['Y'] for all r, as one [R][q][T] tensor (inference.py:96-97).  The reference keeps counts as float64 / int64 of
 * any size (util.py:741,750); here any non-negative integer up to 65535 is accepted (one byte per entry resident, a second plane of
 * high bytes only while some count exceeds 255 - info key "counts_two_bytes"). */
int pgpfa_upload_counts_f64(pgpfa_ctx* ctx, const double* Y);
int pgpfa_upload_counts_u8(pgpfa_ctx* ctx, const uint8_t* Y);
int pgpfa_upload_counts_u16(pgpfa_ctx* ctx, const uint16_t* Y);
/* resident counts of the listed trials (idx NULL: all) as uint16 [n][q][T] */
int pgpfa_get_counts_u16(pgpfa_ctx* ctx, int n, const int32_t* idx, uint16_t* out);
/* params = {'C': [q][p], 'd': [q], 'tau': [p] seconds} (engine.py:40-44).  Builds the p Gram
 * matrices (util.makeK_big, util.py:599-619) and their inverses (inference.py:82) on device. */
int pgpfa_set_params(pgpfa_ctx* ctx, const double* C, const double* d, const double* tau_s);
int pgpfa_get_gram(pgpfa_ctx* ctx, double* K /* [p][T][T] */);
int pgpfa_get_gram_inverse(pgpfa_ctx* ctx, double* Kinv /* [p][T][T] */);

/* ---- Laplace callbacks (inference.py:12-65) -------------------------------------- */
/* negLogPosteriorUnNorm and _grad at caller-supplied points X[n][p][T] for trials idx[n]. */
int pgpfa_laplace_eval(pgpfa_ctx* ctx, int n, const int32_t* idx, const double* X,
                       double* f /* [n] */, double* grad /* [n][p][T] or NULL */);
/* negLogPosteriorUnNorm_hess for one trial, dense [pT][pT] latent-major. */
int pgpfa_laplace_hessian(pgpfa_ctx* ctx, int trial, const double* X, double* H);

/* ---- Laplace E-step (inference.laplace, inference.py:67-185) ---------------------- */
/* Mode finding + posterior covariance blocks for the listed trials.  warm_start = 1
 * starts from the modes resident in the context (prevOptimRes, inference.py:99-102), 0 from
 * zeros, 2 per trial from its resident mode if an earlier E-step produced one and from zeros
 * otherwise (minibatches that revisit trials).  obj_sum = sum over the listed trials of the objective at the
 * mode (the reference returns -obj_sum/numTrials, inference.py:175,183).
 * iters/status (may be NULL): Cholesky factorizations per trial (Newton + the final one at the
 * mode), status 0 = converged. */
int pgpfa_estep_laplace(pgpfa_ctx* ctx, int n, const int32_t* idx, int warm_start,
                        double* obj_sum, int32_t* iters, int32_t* status);
int pgpfa_set_modes(pgpfa_ctx* ctx, int n, const int32_t* idx, const double* X /* [n][p][T] */);
int pgpfa_get_post_mean(pgpfa_ctx* ctx, int n, const int32_t* idx, double* out /* [n][p][T] */);
int pgpfa_get_post_vsm(pgpfa_ctx* ctx, int n, const int32_t* idx, double* out /* [n][T][p][p] */);
/* post_vsmGP in the reference's (T,T,p) layout (inference.py:164-167). */
int pgpfa_get_post_vsmgp(pgpfa_ctx* ctx, int n, const int32_t* idx, double* out /* [n][T][T][p] */);
/* post_cov of one trial, recomputed on demand (inference.py:130-131,161-162). */
int pgpfa_get_post_cov(pgpfa_ctx* ctx, int trial, double* out /* [pT][pT] */);
/* Upload E-step results produced elsewhere (e.g. a reference infRes dict) for the M-step. */
int pgpfa_set_posterior(pgpfa_ctx* ctx, int n, const int32_t* idx, const double* post_mean,
                        const double* post_vsm, const double* post_vsmgp /* [n][T][T][p] */);

/* ---- M-step ----------------------------------------------------------------------- */
/* MStepObservationCost(_grad) (learning.py:20-91) over the trials of the last E-step /
 * pgpfa_set_posterior call; with prior_center != NULL adds |v-center|^2 * inv_s2 / 2 and its
 * gradient ('useDiag' prior of learning.py:445-534).  Multi-GPU: all-reduced over ranks. */
int pgpfa_mstep_cd_costgrad(pgpfa_ctx* ctx, const double* vecCd, const double* prior_center,
                            double inv_s2, double* cost, double* grad /* [q*(p+1)] */);
/* Device Newton solver for the same cost (separable over neurons: q convex problems of dimension p+1):
 * one pass returns the per-neuron costs at vecCd, the Newton steps delta[(p+1)*q] (vecCd layout) and the
 * Newton decrements dec[q]; pgpfa_mstep_cd_cost_per_neuron evaluates trial points for the line search. */
int pgpfa_mstep_cd_newton_pass(pgpfa_ctx* ctx, const double* vecCd, const double* prior_center, double inv_s2,
                               double* cost_n /* [q] */, double* delta /* [q*(p+1)] */, double* dec /* [q] */);
/* Chord pass: cost and gradient at vecCd (one cheap sweep), step from the per-neuron Hessians of the last
 * pgpfa_mstep_cd_newton_pass (also across EM iterations: they stay SPD, the step stays a descent direction). */
int pgpfa_mstep_cd_chord_pass(pgpfa_ctx* ctx, const double* vecCd, const double* prior_center, double inv_s2,
                              double* cost_n /* [q] */, double* delta /* [q*(p+1)] */, double* dec /* [q] */);
int pgpfa_mstep_cd_cost_per_neuron(pgpfa_ctx* ctx, const double* vecCd, const double* prior_center, double inv_s2,
                                   double* cost_n /* [q] */);
/* makePrecomp (learning.py:145-173): PautoSum[p][T][T] over the same trials (all-reduced). */
int pgpfa_mstep_precomp(pgpfa_ctx* ctx, double* num_trials);
int pgpfa_get_pautosum(pgpfa_ctx* ctx, double* out /* [p][T][T] */);
/* MStepGPtimescaleCost(_grad) (learning.py:175-255) for latent k at log-gamma = logp. */
int pgpfa_mstep_tau_costgrad(pgpfa_ctx* ctx, int k, double logp, double* cost, double* grad);
/* The same for all p latents at once (one batched factorisation): logp[p] -> cost[p], grad[p]. */
int pgpfa_mstep_tau_costgrad_batch(pgpfa_ctx* ctx, const double* logp, double* cost, double* grad);
/* m (1..4) candidate points per latent in one batched pass, candidate-major: logp[m][p] -> cost[m][p], grad[m][p].
 * The pass is a latency-bound chain of small launches, so 4 candidates cost about as much as 1; the host-side
 * root finder of learnGPparams brackets and interpolates with them instead of stepping serially. */
int pgpfa_mstep_tau_costgrad_multi(pgpfa_ctx* ctx, int m, const double* logp, double* cost, double* grad);
/* The same pass in two halves (round 6): _begin enqueues it on the context's side stream and returns at once, _end waits for it and returns
 * cost[m][p], grad[m][p] - the same bits as the call above.  Between the two the caller may run the (C,d) passes of the same M-step (learning.py:
 * 93-141 and 257-293 are independent problems; the reference solves them one after the other); one pass in flight per context, every other
 * timescale / precomp / E-step entry point fails while one is. */
int pgpfa_mstep_tau_costgrad_multi_begin(pgpfa_ctx* ctx, int m, const double* logp);
int pgpfa_mstep_tau_costgrad_multi_end(pgpfa_ctx* ctx, double* cost, double* grad);

/* ---- count moments (util.py:523-533 Poisson-PCA initialiser; engine.py:487-492 diagnostics) ---- */
/* Exact integer moments of the resident counts over all (trial, bin) samples of the listed trials:
 * sum[i] = sum y_i, cross[i][j] = sum y_i y_j, n_samples = trials * T (np.mean / np.cov of the raster follow). */
int pgpfa_count_moments(pgpfa_ctx* ctx, int n, const int32_t* idx, int64_t* sum /* [q] */, int64_t* cross /* [q][q] */,
                        int64_t* n_samples);

/* ---- synthetic population (util.dataset, util.py:705-750) ------------------------------------------ */
/* Draws x_k ~ N(0, K(tau_k)) and y_nt ~ Poisson(exp(c_n . x_t + d_n)) for the listed trials (NULL: all) under the parameters of
 * pgpfa_set_params, from a counter-based generator keyed by `seed` (NOT NumPy's stream).  The counts replace those trials in
 * the resident tensor; X_out[n][p][T] / Y_out[n][q][T] (either may be NULL) receive copies. */
int pgpfa_generate(pgpfa_ctx* ctx, unsigned long long seed, int n, const int32_t* idx, double* X_out, uint8_t* Y_out);

/* ---- leave-one-neuron-out prediction (util.py:289-334, engine.py:599-644) ----------- */
/* For each listed trial (idx NULL: all R) and each neuron nn: the Laplace mode of the latents given the other
 * q-1 neurons (cold start), then y_pred[(trial, nn)][t] = exp(C[nn] . x_t + d[nn]); err_sum = sum of squared
 * differences to the held-out counts.  The R*q mode searches run batched on the E-step machinery. */
int pgpfa_loo_predict(pgpfa_ctx* ctx, int n, const int32_t* idx, double* y_pred /* [n][q][T] */, double* err_sum);

/* ---- dual variational E-step (inference.py:188-432) -------------------------------- */
/* dualProblem and dualProblem_grad for one trial at lambda[q*T] (structured: never forms
 * C_big or diag(lambda)). */
int pgpfa_dual_costgrad(pgpfa_ctx* ctx, int trial, const double* lam, double* cost,
                        double* grad /* [q*T] or NULL */);
/* The same for a list of distinct trials at once, each at its own lambda: lam[n][q*T] -> cost[n], grad[n][q*T]
 * (or NULL).  The dense factorisations of a chunk of trials are batched; inference.dualVariational drives the
 * reference's per-trial L-BFGS-B runs concurrently so that one round of their requests is one call. */
int pgpfa_dual_costgrad_batch(pgpfa_ctx* ctx, int n, const int32_t* idx, const double* lam, double* cost, double* grad);
/* The whole dual optimisation of the listed distinct trials on the device: one L-BFGS run (10 correction pairs, Armijo
 * backtracking) per trial in rho = log(lambda) - the unconstrained form of the reference's optimizeLogLambda path
 * (inference.py:222-256, 391-396) - all runs of a chunk in lockstep, every iteration one batched dual evaluation.
 * Stops per trial on scipy's L-BFGS-B criteria (relative decrease <= factr * 2.2e-16, or max |gradient| <= pgtol).
 * rho[n][q*T]: start in, optimum out; fopt[n]: dual optimum (inference.py:325/397); iters[n] may be NULL. */
int pgpfa_dual_lbfgs(pgpfa_ctx* ctx, int n, const int32_t* idx, double* rho, int max_iter, double factr, double pgtol,
                     double* fopt, int32_t* iters);
/* The optimum of the same dual problem by a fixed point instead of a quasi-Newton run in lambda.  At the optimum of the reference's dual
 * (zero of dualProblem_grad, inference.py:215-219)  log lambda = d + C m + v  with  m = -K C_big (lambda - y)  (VIPostMean, :193) and
 * v = 1/2 diag(C Sigma C^T)  (VIPostCov with its 1e-6 jitter, :188-191).  Given v, m is the mode of the Laplace objective with the log
 * rates shifted by v (the batched Newton-PCG of pgpfa_estep_laplace, warm-started) and lambda = exp(C m + d + v); given lambda, v follows
 * from the per-bin covariance blocks.  The map v -> v contracts by about half the largest posterior variance of a log rate per pass
 * (a digit or more), so a trial needs ~5-10 covariance passes where L-BFGS in rho = log lambda needs thousands of evaluations.
 * Stops per trial when max |v_new - v| <= tol - exactly the max-norm of dualProblem_grad at the returned lambda.
 * rho[n][q*T] (may be NULL with start = 0 / 3: nothing read, nothing written): log lambda - read as the start (start = 1, 2), written with the
 * optimum; start: 0 cold, lambda = 0.5 everywhere (the reference's start, inference.py:302), 1 rho is the start and the mode search begins at
 * zero, 2 rho is a previous optimum and the mode search begins at its variational mean, 3 the same from the optimum resident on the device; fopt[n]: dual cost at the optimum (inference.py:196-213); outer[n] (may be NULL):
 * passes used; vstatus[n]: 0 converged, 1 pass cap reached, 2 not contracting (hand the trial to pgpfa_dual_lbfgs from the rho returned);
 * lam_out[n][q*T] (may be NULL): the optimal lambda itself (exp and log of the q T entries run on the device).  The optimum also stays on the
 * device: pgpfa_dual_finalize with lam = NULL takes it from there. */
int pgpfa_dual_fixed_point(pgpfa_ctx* ctx, int n, const int32_t* idx, double* rho, int start, int max_outer, double tol, double* fopt,
                           int32_t* outer, int32_t* vstatus, double* lam_out);
/* The dual variables resident for the listed trials (the optimum of the last pgpfa_dual_fixed_point, or what pgpfa_dual_finalize was given):
 * out[n][q*T]. */
int pgpfa_get_dual_lambda(pgpfa_ctx* ctx, int n, const int32_t* idx, double* out);
/* VIPostMean (inference.py:193-194): mean[p*T] = -K_big C_big (lambda - ybar) for one trial, latent-major. */
int pgpfa_dual_post_mean(pgpfa_ctx* ctx, int trial, const double* lam, double* mean);
/* VIPostCov (inference.py:188-191) for one trial, dense [pT][pT] latent-major: prec = K_big^-1 + C_big diag(lambda) C_big^T
 * (may be NULL) and cov = (prec + 1e-6 diag(diag(prec)))^-1. */
int pgpfa_dual_post_cov(pgpfa_ctx* ctx, int trial, const double* lam, double* cov, double* prec);
/* VIPostMean / VIPostCov blocks at lambda for the listed trials; fills the same posterior
 * slots as the Laplace E-step and returns sum of negLogPosteriorUnNorm at the VI mean. */
int pgpfa_dual_finalize(pgpfa_ctx* ctx, int n, const int32_t* idx, const double* lam /* [n][q*T]; NULL: the optimum pgpfa_dual_fixed_point left
                        on the device for these trials */, double* nlp_sum);

/* ---- multi-GPU (trial sharding; RCCL over xGMI) -------------------------------------- */
int pgpfa_comm_unique_id(char* id128 /* 128 bytes */);
int pgpfa_comm_init(pgpfa_ctx* ctx, const char* id128, int rank, int nranks);
int pgpfa_comm_allreduce_host(pgpfa_ctx* ctx, double* buf, int count);
/* One line about this rank's communicator for start-up logs: "rank r/n device d pci <bus id> comm_ranks m comm_device d" (comm_ranks = ncclCommCount:
 * the number of ranks RCCL itself saw); without a communicator "rank 0/1 device d pci <bus id> comm none".  buf is NUL-terminated. */
int pgpfa_comm_describe(pgpfa_ctx* ctx, char* buf, int len);

/* ---- test / bench hooks ---------------------------------------------------------------- */
/* Batched SPD factor (+ optional inverse) of caller matrices through the production kernels:
 * A[batch][n][n] symmetric in, Cholesky factor L (lower, row-major) out; inv (may be NULL)
 * receives A^-1.  Used by the parity tests and by bench.py's roofline leg. */
int pgpfa_test_potrf(pgpfa_ctx* ctx, int batch, int n, const double* A, double* L, double* inv);
/* C = alpha*A*B^T + beta*C, column-major, through the production MFMA GEMM (use_mfma=1) or
 * the scalar check kernel (use_mfma=0). */
int pgpfa_test_gemm_nt(pgpfa_ctx* ctx, int M, int N, int K, double alpha, const double* A,
                       const double* B, double beta, double* C);
/* The same with B given as K x N column-major (C = alpha*A*B + beta*C): the multi-RHS sweep form. */
int pgpfa_test_gemm_nn(pgpfa_ctx* ctx, int M, int N, int K, double alpha, const double* A,
                       const double* B, double beta, double* C);
/* The same two products through the single-precision instantiation of the MFMA kernel (v_mfma_f32_16x16x4_f32; operands are
 * rounded to float on the way in, C comes back widened): the GEMM of the mixed-precision dual evaluation ("dual_f32"). */
int pgpfa_test_gemm_nt_f32(pgpfa_ctx* ctx, int M, int N, int K, double alpha, const double* A,
                           const double* B, double beta, double* C);
int pgpfa_test_gemm_nn_f32(pgpfa_ctx* ctx, int M, int N, int K, double alpha, const double* A,
                           const double* B, double beta, double* C);
/* Times `reps` launches of the dominant kernel (batched trailing SYRK update, K=512) with HIP
 * events on the context stream; returns average ms per launch and the flops of one launch. */
int pgpfa_bench_syrk(pgpfa_ctx* ctx, int batch, int n, int k, int reps, double* ms_per_launch,
                     double* flops_per_launch);
/* GEMM launches since option "profile" was switched on, grouped by operand shape (text, one line per shape: launches, total ms, algorithmic
 * GFLOP, TFLOP/s; longest total first).  Returns the bytes the full report needs incl. the terminator; writes at most len. */
int pgpfa_gemm_shape_report(pgpfa_ctx* ctx, char* buf, int len);
/* Phase timings of the 128 x 128 diagonal-block kernel (chol.h): phases 0 = load/store only, 1 = + Cholesky steps, 3 = + inverse; + 4 (5, 7): the
 * round-1 form of the Cholesky steps (a row and 32 columns per thread) instead of the 4 x 8 register blocks. */
int pgpfa_bench_potrf_diag(pgpfa_ctx* ctx, int batch, int reps, int phases, double* us_per_launch);
/* Sustained v_mfma_f64_16x16x4_f64 rate of the device (register-only loop): the practical MFMA
 * ceiling under the clock the chip holds, reported next to the datasheet peak. */
int pgpfa_bench_mfma_peak(pgpfa_ctx* ctx, int iters, double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* PGPFA_H */
