// Shared declarations of the translation units of libpgpfa_hip.so: the context (one GPU, one stream, all resident state), error / check
// macros, and the host-side helpers that cross translation units.  Kernels live in the kernel headers; a translation unit includes only
// the ones it launches (core.hip: context, workspace, copies; linalg.hip: GEMM / factor; estep.hip: Newton-PCG E-step; cov.hip:
// covariance engines; mstep.hip; dual.hip; misc.hip: comm, generator, count moments).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <limits>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include <mutex>
#include <rocprofiler-sdk-roctx/roctx.h>
#include "../../include/pgpfa.h"
#include "types.h"

using namespace pgpfa;


extern thread_local std::string g_err;
extern thread_local unsigned long long g_fail_count;   // failures reported on this thread (queued read-backs of a failed call are void: dl_enqueue / dl_flush)

int fail(const char* fmt, ...);

#define HIPC(expr)                                                                         \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)
#define CHK(expr)            \
  do {                       \
    int _r = (expr);         \
    if (_r != 0) return _r;  \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// roctx range around a phase of the path (tracing): rocprofv3 --marker-trace shows E-step / Newton solves / covariance blocks /
// M-step passes as named ranges above the kernel rows; a push / pop pair costs nothing measurable when no tool is attached.
struct PhaseRange {
  explicit PhaseRange(const char* name) { roctxRangePushA(name); }
  ~PhaseRange() { roctxRangePop(); }
  PhaseRange(const PhaseRange&) = delete;
  PhaseRange& operator=(const PhaseRange&) = delete;
};

struct Prof {
  bool on = false;
  bool configured = false;           // option "profile" is set (option "profile_pause" toggles `on` under it)
  int only_tag = -1;                 // >= 0: time launches of this tag only (option "profile" = 2: the GEMM kernel)
  std::vector<hipEvent_t> pool;      // every event ever created (destroyed with the context)
  std::vector<hipEvent_t> idle;      // events free for reuse
  struct Rec { int tag; hipEvent_t e0, e1; double flops; std::string shape; };
  std::deque<Rec> recs;              // launches whose events have not been read yet, oldest first
  bool open = false;                 // prof_begin recorded, prof_end pending
  std::map<int, double> ms, flops, count;
  std::map<int, double> max_ms, max_flops;   // the longest single launch of each family and its algorithmic flops
  struct Shape { double ms = 0.0, flops = 0.0, count = 0.0; };
  std::map<std::string, Shape> shapes;       // GEMM launches by operand shape (pgpfa_gemm_shape_report)
};
constexpr int TAU_MULTI_MAX = 4;  // candidate points per latent in one batched timescale cost/gradient pass
constexpr int PACC_SPLITS = 64;   // split-K groups of the sum-only vsmGP product
// (TAG_MIX: the mixing passes over the Yt slab - HBM-bound, so the number recorded with a launch is its algorithmic BYTES, not flops)
enum { TAG_GEMM = 0, TAG_POTRF = 1, TAG_SOLVE = 2, TAG_POISSON = 3, TAG_ASSEMBLE = 4, TAG_VSM = 5, TAG_CD = 6, TAG_MIX = 7, TAG_N };


struct pgpfa_ctx {
  int device = 0, q = 0, p = 0, T = 0, R = 0, n = 0, npad = 0, ld = 0, Tp = 0;
  double bin = 10.0, eps = 1e-3;
  hipStream_t st = nullptr;
  // asynchronous timescale pass (pgpfa_mstep_tau_costgrad_multi_begin / _end): runs on st2, pinned block for its inputs and results
  double* tau_pin = nullptr; hipEvent_t ev_tau_fork = nullptr, ev_tau_done = nullptr; int tau_inflight = 0;
  hipStream_t st2 = nullptr;                     // side stream: the pivoted Cholesky of the Gram matrices (10 workgroups) next to the Gram inverses in pgpfa_set_params
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int overlap_factors = 1;                       // 1: use it; 0: one stream
  // options
  double xtol = 1e-5;
  int max_iter = 60;
  bool chord = true;
  double chord_xtol = 1e-9, chord_rho = 0.6, chord_max_step = 1.0;
  int chord_max = 40;
  bool mfma = true;
  int chunk_opt = 0;
  // persistent device state
  uint8_t* Y = nullptr;
  uint8_t* Yhi = nullptr;                        // high bytes of the counts: allocated only while the tensor holds a count above 255
  double *C = nullptr, *d = nullptr, *tau = nullptr;
  double *Kpad = nullptr, *Kinv = nullptr;      // [p][Tp][Tp]
  double* Xmode = nullptr;                       // [R][p][T]   post_mean / warm start
  double* Xprev = nullptr;                       // [R][p][T]   modes of the E-step before (warm-start extrapolation)
  std::vector<int> mode_serial, prev_serial;     // E-step serial that produced Xmode / Xprev of a trial (-10: unknown)
  int estep_serial = 0;
  bool extrapolate = true;
  double extrapolate_beta = 1.0;
  // (round 6) The extrapolated start m_k-1 + beta (m_k-1 - m_k-2) predicts the new mode when the parameters keep moving the way they moved between
  // the last two E-steps.  After a JUMP of the parameters (cross-validation folds, a second fit in one process, a fit evaluated at other parameters)
  // the difference m_k-1 - m_k-2 is the jump's effect and adding it again throws every start point far out: the E-step behind a jump took 2.2 s where
  // a cold one takes 0.12 (profiles/r05_bench_c3_driver_protocol.json, plateau.estep_ms_of_the_four).  The displacement of the parameters between
  // consecutive Laplace E-steps is tracked (par_step: relative, the largest of C, d, tau) and the extrapolation is only used while the step
  // that produced m_k-1 - m_k-2 was not much longer than the one being taken (extrapolate_guard; 0 switches the test off).
  std::vector<double> estepC, estepd, esteptau;  // parameters of the last Laplace E-step
  double par_step = -1.0, par_step_prev = -1.0;  // |theta_k - theta_k-1|, |theta_k-1 - theta_k-2| (relative; < 0: unknown)
  double extrapolate_guard = 3.0;
  bool start_guard = true;                       // a warm start whose objective is above the cold start's (x = 0) is replaced by zero (estep.hip)
  double* vsm = nullptr;                         // [R][T][p][p]
  double* vsmgp = nullptr;                       // [R][p][T][T]
  double* Pauto = nullptr;                       // [p][Tp][Tp]
  // sum-only covariance output of the low-rank engine (option keep_trial_vsmgp = 0): the E-step accumulates
  // sum_r Sigma_r^{kk} here instead of storing R x p blocks of T x T; per-trial blocks are rebuilt on request
  double* Pacc = nullptr;                        // [p][Tp][Tp]
  double* gemm_part = nullptr; size_t gemm_part_len = 0;   // split-K partial products of the thin GEMMs
  double *CCu = nullptr, *C16 = nullptr;         // zero-padded pair-product / loading tables of the MFMA Poisson pass
  int qpad = 0, ccu_cols = 0;
  double* ppart = nullptr;                       // [p][PACC_SPLITS + 1][T x T] split-K partial products
  double* split_buf = nullptr;                   // scratch of the split accumulation (split.h), allocated on first use
  bool split_cov = true;                         // option split_cov: sum_r Y~Y~^T by the exact split form when eps ||Wt|| allows
  // ... i.e. up to this value of the root mean square over (trial, bin) of eps ||Wt_t||_inf (option split_max_norm).  Measured against the
  // FP64 product at config-3 dimensions (tools/split_probe.py, 512 trials): 3e-12 of PautoSum at the generating parameters (rms 0.02,
  // max 0.22), 3.8e-11 / 2.1e-10 for populations firing 3 / 8 times faster (rms = max = 0.018 / 0.045): the error grows like the
  // square of the rms, 0.07 keeps 1e-9 with a factor two to spare
  double split_max_norm = 0.07;
  double *cdym = nullptr, *cdym_part = nullptr;   // count terms of the (C,d) cost: sum_t y m_t, sum_t y per neuron (per E-step)
  bool cdym_valid = false, cd_mfma = true, cd_hess_mfma = true; int cd_debug = 0;
  bool cd_hess_valid = false; int cd_hess_ntr = 0;   // per-neuron Hessian sums of the last Newton pass are resident
  std::vector<double> logdetK;                  // log det of the p Gram matrices (from the factor in build_kinv)
  bool dual_lowrank = true;                     // dual-variational entry points use the low-rank engine when it pays (want_lowrank)
  double* dual_tbl = nullptr; int dual_ncol = 0, dual_npd = 0; bool dual_gemm = true;   // pair / loading table of the GEMM form (dual.h)
  bool vsm_mfma = true;                         // per-bin Gram blocks (post_vsm) on the matrix cores beyond 10 latents
  bool slab_row_align = true;                   // latent row stride of the Yt slab rounded up to 16 rows (128-byte lines)
  int dual_f32 = 0;                             // ... with the r x r factorisation, its inverse and Yt in single precision (mixed)
  float* Flr32 = nullptr; bool flr32_valid = false;   // single-precision copy of the low-rank factors
  bool keep_trial_vsmgp = false;
  bool pacc_used = false, pacc_valid = false;
  std::vector<char> vsmgp_ok;                    // per trial: c->vsmgp holds the blocks of the resident posterior
  std::vector<double> hC, hd, htau;              // parameters as last set
  // parameters every resident posterior was computed under: one snapshot per E-step (or dual finalize), referenced per trial, so that
  // blocks rebuilt on demand (post_vsmGP under the sum-only plan, post_cov) are those of the trial's OWN E-step even when other trials
  // have been through later E-steps at other parameters (minibatch EM)
  struct ParamSnap { std::vector<double> C, d, tau; };
  std::map<int, ParamSnap> snaps;
  std::vector<int> trial_snap;                   // per trial: key into snaps (-1: posterior not produced by an E-step of this context)
  // trials whose resident posterior is a dual-variational one (pgpfa_dual_finalize): their blocks follow from lambda, not from the mode,
  // so the optimal lambda of those trials stays on the device for rebuilds on demand (allocated by the first finalize)
  std::vector<char> trial_dual;
  std::vector<char> lam_resident;                // per trial: lam_keep holds the optimum of the last pgpfa_dual_fixed_point (pgpfa_dual_finalize with lam = NULL)
  std::vector<char> lam_valid;                   // per trial: lam_keep holds the dual variables of the trial's last variational E-step (set by the fixed point and by
                                                 // pgpfa_dual_finalize, cleared only when the counts change: a Laplace E-step or an uploaded posterior in between
                                                 // supersede the trial's POSTERIOR, not the dual variables a caller still holds as varOptimRes)
  double* lam_keep = nullptr;                    // [R][q][T]
  int snap_serial = 0;
  double *vec = nullptr, *cdpart = nullptr, *cdout = nullptr;
  double *cdhpart = nullptr, *cdhout = nullptr, *cdcenter = nullptr, *cdpack = nullptr;   // Newton M-step (cdpack: [cost sums | delta | dec | R], read back in one copy)
  int* last_trials = nullptr;                    // device list of the trials of the last E-step
  std::vector<int> last_trials_h;
  bool have_counts = false, have_params = false, have_post = false, have_precomp = false;
  double n_trials_global = 0.0;
  // chunk workspace
  int B = 0;
  double grow_budget_ms = 200.0;                  // time a plan may spend mapping memory beyond what grow_floor_slots slots need (0: no limit)
  int grow_floor_slots = 128;
  bool use_pool = true;                           // take the arena a closed context of this process left in the pool (option workspace_pool)
  int plan_target = 0;                            // trial-list length the last workspace plan was made for
  int want_slots = 0;                             // largest trial list an E-step-like call has asked for
  bool B_capped = false;
  CholWS ws{};
  double *Xc = nullptr, *Xt = nullptr, *KX = nullptr, *KD = nullptr, *Gl = nullptr, *Glt = nullptr, *Gt = nullptr, *Dl = nullptr;
  double *W = nullptr, *Wt = nullptr, *fpart = nullptr;
  double *lamd = nullptr, *dgrad = nullptr, *dpart = nullptr, *ldet_buf = nullptr;   // dual variational scratch
  double* voff = nullptr;                         // [B][q][T] variance offsets 1/2 c_n^T Sigma_t c_n of the variational fixed point
  bool var_active = false, lam_out_active = false;   // Poisson passes add voff to the log rate / write the rates into lamd
  double* dual_scr = nullptr; long long dual_sscr = 0;   // [B][T x max(pairs padded, p^2)] packed pair tables of the GEMM form
  // shared-preconditioner Newton-PCG: one factor per chunk (mean-trial Hessian), PCG vectors per slot
  CholWS sws{};
  double *sU = nullptr, *sDinvT = nullptr, *Wbar = nullptr;
  double *Rv = nullptr, *Zv = nullptr, *Pv = nullptr, *Qv = nullptr;
  PcgCtl* pcgctl = nullptr;                      // device-side control block of the inner PCG loop (pcg.h)
  int* live = nullptr; float *pcg_ratio = nullptr, *pcg_eta = nullptr;   // device live list of the inner solve, per-slot residual ratio / target
  int* live1 = nullptr;                          // second live list of the two-kernel step
  // (pcgctl, list_a, live, pcg_eta, live1 are consecutive pieces of ONE block - 16 + 4 B words - so that a solve's control block, first live list
  //  and forcing terms go up in one copy: pcg_blk_host is its pinned image)
  int* pcg_blk = nullptr;
  const int* cur_ndev = nullptr;                 // while set: products with a column list take their column count from this device word
  bool live_gemm_collect = false;                // profiling: the first iteration of an inner solve lists its live-list products here
  std::vector<std::pair<std::string, double>> live_gemms;   // (shape key, algorithmic flops per column)
  bool pcg_retire = true;                        // slots leave the inner solve as they reach their own targets (option pcg_retire)
  int* h_pcg = nullptr; int* d_hpcg = nullptr;   // host-mapped copy {stop, iterations}: the host peeks, never waits
  float* W32 = nullptr;                          // packed single-precision curvature triangles of the chunk's slots (PCG matvec)
  double* sc_part2 = nullptr;                    // per (slot, tile) partial sums r.z, r.r
  int pcg_fused = 1; bool pcg_w32 = true;        // pcg_fused: 0 off, 1 when the chunk is large enough, 2 always (tests)
  int mt_fill = 1;                               // 1: before the inverse only the entries of the L^-T slabs that are read and not written are cleared; 0: the whole slab
  int pcg_xcd = 1;                               // 1: the per-bin kernels of the inner step place the bin tiles of a slot group on one XCD (pcg_cg_wg)
  int pcg_adapt = 1;                             // 1: launches of the host-free inner step sized by the mirrored live count, 16 / 8 / 4 slots per workgroup; 0: by the solve's first count
  int pcg_form = 2;                              // host-free inner iteration (pcg.h): 2 (round 5) = 1 with the solve's private vectors on line-aligned rows, one start kernel,
                                                 // the step's closing folded into kernel A and one upload per solve; 1 two tile-parallel kernels per step, no prior mat-vec (pcg_cg_a/b_kernel);
                                                 // 0 the split kernels of round 3 with K^-1 p as a product
  int pcg_rx32 = 1;                              // with pcg_vec32, up to 10 latents: the residual and the step of a solve stored in single precision too (pcg.h: PcgCgP::X32)
  int pcg_vec32 = 1;                             // with pcg_form 2: z, s, p, q and the preconditioner's t / y stored in single precision (pcg.h: PcgCgP::vec32)
  double *Sv = nullptr, *cg_scal = nullptr;      // s = H~ z of the two-kernel form; its per-slot scalars [gamma | alpha] x step parity
  double *GbT = nullptr, *WbT = nullptr;         // [NP][T] packed triangles of the shared preconditioner's Gb and of the mean curvature (pcg_cg_a/b_kernel)
  double *sc_rz = nullptr, *sc_pq = nullptr, *sc_rr = nullptr, *sc_rr0 = nullptr, *sc_pack = nullptr;
  // low-rank covariance engine
  double* Flr = nullptr;                          // [p][Tp x Tp] pivoted-Cholesky factors of the RBF part
  double* Gbin = nullptr;                         // [B][T][p][p]
  int *d_rank = nullptr, *d_blk_lat = nullptr, *d_blk_col = nullptr, *d_roff = nullptr;
  int *d_kr_ft = nullptr, *d_kr_f = nullptr;     // per-row-tile k ranges of the block-diagonal F^T / F GEMMs
  int kr_ft_len = 0, kr_f_len = 0;               // longest of those ranges
  int ntab_ft = 0, ntab_f = 0; size_t tab_cap = 0;   // entries of the two row-tile tables, capacity (ints) of their device buffers
  int *d_thin_ft = nullptr, *d_thin_f = nullptr, *d_thin_s = nullptr;  // work tables of the same two products - and of Sb u - as kernels of their own (thin.h)
  int nthin_ft = 0, nthin_f = 0, nthin_s = 0;
  int thin_products = 2;                         // 1: F^T t and F v of the preconditioner application by thin.h's kernels, 2: Sb u too; 0: GEMMs
  double *Fbig = nullptr, *FTbig = nullptr, *Gbar = nullptr, *Wtbar = nullptr;   // low-rank shared preconditioner
  std::vector<int> rk, roff;                      // ranks padded to 16; offsets of the latents in the r x r system (core.hip: build_lowrank)
  std::vector<int> rr, roff16;                    // ranks rounded to rank_gran; offsets in the padded-16 index space B is assembled in
  int rtot = 0, rpad = 0, rtot16 = 0;
  int rank_gran = 4;                              // option: 4 (default since round 6) / 8 = compact offsets, or 16 (build_lowrank)
  bool rank_compact = false;
  int *d_roff16 = nullptr, *d_cmap = nullptr, *d_nrtab = nullptr;
  int cov_mode = 0;                               // 0 auto, 1 dense, 2 low-rank
  double lr_tol = 1e-10;
  bool plan_lowrank = false;                      // current workspace plan
  size_t slab_elems = 0, mt_elems = 0, ws_mark = 0;     // doubles per slot of the factor slabs (H / Yt) and of the L^-T slabs
  // the chunk workspace lives in ONE device allocation that re-plans re-partition (hipFree + hipMalloc of ~10^11 bytes
  // costs seconds); arena_mode: 0 = dmalloc is a plain hipMalloc, 1 = only measure, 2 = carve from the arena
  char* arena = nullptr; size_t arena_cap = 0, arena_off = 0; int arena_mode = 0;
  // The arena is a reserved virtual address range into which physical memory is mapped as the need grows (HIP virtual memory
  // management): growing never moves it and only the NEW bytes pay the driver's page clearing (~25 ms per GB).  vmm: 0 untried,
  // 1 in use, -1 unavailable (plain hipMalloc of the size needed, re-allocated on growth).
  int vmm = 0; size_t va_size = 0, vmm_gran = 0, vmm_granule = (size_t)1 << 30;
  std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> vmm_chunks;
  double arena_headroom = 2.0;                  // rank head-room of a low-rank plan (option workspace_headroom)
  bool mt_dirty = false;                          // low-rank use scribbled over the Mt slabs' zero triangle
  bool last_cov_lowrank = false;
  bool shared_pcg = true;
  bool pcg_trace = false;
  double* sink = nullptr;                        // 128 doubles nobody reads: where the rows past p of mix_vsm_wide2_kernel store
  int mix_wide = 1;                              // 1: the mixing pass of 17..20 latents with lanes along the bins (mix_vsm_wide2_kernel); 0: mix_vsm_wide_kernel
  int poisson_tiles = 2;                      // option poisson_tiles: 16-bin tiles per wave of the matrix-core Poisson pass up to 10 latents (2: the table fragments of a neuron tile serve two tiles; 1)
  int vsm_b4 = 2;                             // option vsm_b4: post_vsm for 11..20 latents on the 4 x 4 x 4 block shape of the FP64 matrix cores (post_vsm_b4_kernel; 2: 64 bins per workgroup staged with 16-byte loads where the strides allow, 1: 32 bins, scalar loads), 0: the 16 x 16 x 4 form
  int pivchol_pairs = 1;                      // option pivchol_pairs: rbf_pivchol2_kernel (two bins per row thread, four column groups) beyond 256 bins
  int yt_mix = 1;                             // option yt_mix: Yt = F L^-T and the mixing pass of the split form as one kernel up to 10 latents - Yt is never written (ytmix.h)
  int syrk_tile = 256;                        // option syrk_tile: workgroup tile of the FP16 term of the split sum (256 where T > 256 and the strides allow; 128)
  int syrk_dbg = 0;                           // option syrk_dbg: timing experiments on syrk_f16x2_kernel (parts switched off, results wrong)
  int yt_mix_dbg = 0;                         // option yt_mix_dbg: timing experiments on that kernel (parts switched off, results wrong); never set outside tools/
  int mix_slot = 3;                           // option mix_slot (3: two column halves per bin, two workgroups per CU - mix_slot3_kernel; 2: mix_slot2_kernel): the mixing pass of the split form with a thread per bin and whole columns per workgroup (split.h)
  bool cross_kernel = true;                       // option cross_kernel = 0: the cross term of the split form through the general GEMM kernel
  bool measure_mix = false;                       // option measure_mix: record max_t eps ||Wt_t|| of every covariance pass
  bool time_newton = false;                       // option time_newton: HIP events around the inner PCG solves (last_newton_solve_ms / _bytes)
  int shared_min = 16, pcg_inner_min = 2, pcg_inner_max = 16, pcg_outer_max = 12;
  double pcg_eta0 = 1e-2;
  int splitk_target = 1280;                      // thin GEMMs are cut along k until about this many workgroups are in flight
  int f32_tile64 = 1;                            // 1: single-precision products on 64 x 64 tiles too (as FP64); 0: 128 x 128 only
  int small_tile_below = 1 << 30;                // products with fewer 128 x 128 tiles than this run on 64 x 64 tiles (0: never); measured: the
                                                 // small tile wins at every shape of the E-step (44.5 -> 50 TFLOP/s on the largest launch too)
  int splitk_below64 = 400;                      // ... and are cut along k only below this many 64 x 64 tiles (round 4: 160 -> 400 - with the prior mat-vec out of the PCG step
                                                 // its thin products are what is left: Newton solves 12.4 -> 11.9 ms per EM iteration at config 3)
  double *sc_f = nullptr, *sc_qxx = nullptr, *sc_qdx = nullptr, *sc_qdd = nullptr, *sc_dec = nullptr, *sc_smax = nullptr, *sc_alpha = nullptr;
  int *trial_of_slot = nullptr, *list_a = nullptr, *list_b = nullptr, *ident = nullptr;
  int* mask_of_slot = nullptr;                    // leave-one-neuron-out passes: neuron excluded from the likelihood of a slot
  bool mask_active = false;
  // small workspace for the T x T systems (Kinv, tau M-step): p slots of Tp
  CholWS kws{};
  double *tK = nullptr, *tM = nullptr, *tA1 = nullptr, *tA2 = nullptr, *tscal = nullptr, *tpart = nullptr;
  // pinned host staging
  double* hbuf = nullptr; size_t hbuf_len = 0;
  struct DlEntry { void* host; size_t off, bytes; };
  PcgCtl fused_ctl_host{};
  char* dl_stage = nullptr; size_t dl_used = 0; std::vector<DlEntry> dl_pending;   // pinned staging of small read-backs (dl_enqueue / dl_flush)
  // Small copies as kernels (option copy_kernels): the staging areas are host-mapped, a one-block kernel moves the bytes and the flush is a kernel
  // that raises a sequence number in mapped memory the host spins on - a hipMemcpyAsync of a few KB costs 130-570 us of device idle time on
  // this stack (the runtime's blit path: kernel trace, tools/trace_gaps.py), a kernel launch 5-10
  char* dl_stage_dev = nullptr; char* ring_dev = nullptr;
  unsigned* h_seq = nullptr; unsigned* d_seq = nullptr; unsigned seq_next = 0;
  bool copy_kernels = true;
  unsigned long long dl_fail_mark = 0;           // g_fail_count when the oldest pending read-back was queued
  int* hibuf = nullptr; size_t hibuf_len = 0;
  // ring of pinned staging slots for small host -> device uploads that must not cost a stream synchronisation each (Newton driver)
  char* ring = nullptr; size_t ring_slot = 0; int ring_cur = 0, ring_pending = 0;
  // stats
  std::map<std::string, double> info;
  std::vector<void*> allocs;
  size_t bytes = 0;
  Prof prof;
  // comm
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  double* commbuf = nullptr; size_t commbuf_len = 0;
};

template <typename T>
int dmalloc(pgpfa_ctx* c, T** out, size_t count, bool zero = false) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  if (c->arena_mode) {
    const size_t aligned = (bytes + 255) & ~(size_t)255;
    *out = nullptr;
    if (c->arena_mode == 2) {
      if (c->arena_off + aligned > c->arena_cap) return fail("workspace arena overflow (%zu + %zu > %zu bytes)", c->arena_off, aligned, c->arena_cap);
      p = c->arena + c->arena_off;
      if (zero && hipMemsetAsync(p, 0, bytes, c->st) != hipSuccess) return fail("hipMemset failed");
      *out = reinterpret_cast<T*>(p);
    }
    c->arena_off += aligned;
    return 0;
  }
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  if (zero) {
    e = hipMemsetAsync(p, 0, bytes, c->st);
    if (e != hipSuccess) return fail("hipMemset failed: %s", hipGetErrorString(e));
  }
  c->allocs.push_back(p);
  c->bytes += bytes;
  *out = reinterpret_cast<T*>(p);
  return 0;
}

// exact (or nearest) unrolled latent width for the kernels that pad with zeros instead of guarding
template <typename F>
void dispatch_pw(int p, F&& f) {
  switch (p) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 5: f(std::integral_constant<int, 5>{}); break;
    case 6: f(std::integral_constant<int, 6>{}); break;
    case 7: case 8: f(std::integral_constant<int, 8>{}); break;
    case 9: case 10: f(std::integral_constant<int, 10>{}); break;
    case 11: case 12: f(std::integral_constant<int, 12>{}); break;
    case 13: case 14: case 15: case 16: f(std::integral_constant<int, 16>{}); break;
    case 17: case 18: case 19: case 20: f(std::integral_constant<int, 20>{}); break;
    default: f(std::integral_constant<int, 32>{}); break;
  }
}

template <typename F>
void dispatch_pmax(int p, F&& f) {
  if (p <= 4) f(std::integral_constant<int, 4>{});
  else if (p <= 8) f(std::integral_constant<int, 8>{});
  else if (p <= 16) f(std::integral_constant<int, 16>{});
  else if (p <= 24) f(std::integral_constant<int, 24>{});      // (config 5 has 20: the 32-wide instantiations spill)
  else f(std::integral_constant<int, 32>{});
}

// Rows of latents (blockDim.y) of the (bins x latents) blocks of post_vsm_kernel / poisson_pass_kernel.  Beyond 16 latents a thread
// owns two rows; 17..24 (post_vsm) and 17..20 (poisson_pass) run 12 / 10 rows so that the block stays under 1024 threads and
// keeps more than 128 registers per lane (their launch bounds in model.h say the same).
inline int post_vsm_rows(int p) { return p <= 16 ? p : (p <= 24 ? 12 : 16); }
inline int poisson_rows(int p) { return p <= 16 ? p : (p <= 20 ? 10 : 16); }

struct Trials {
  std::vector<int> v;
};

// Leave-one-neuron-out job riding on the E-step machinery: item i is (trial tr.v[i], held-out neuron mask[i]); only the
// mode is found (no covariance blocks, nothing written to the per-trial state), then the held-out neuron is predicted.
struct LooJob {
  const std::vector<int>* mask;
  double* y_pred;        // host [N][T]
  double* err;           // host [N]
};

// Variational fixed point (pgpfa_dual_fixed_point): the mode search below runs with the variance offsets in the log rate, in a loop with
// the covariance blocks that produce them.
struct VarJob {
  double* rho;            // host [N][q*T]: log lambda, start in / optimum out
  int max_outer;
  double tol;             // stop: max |1/2 c_n^T Sigma_t c_n - offset| <= tol (the max-norm of the reference's dual gradient, inference.py:218)
  double* fopt;           // host [N]: dual cost at the optimum
  int32_t* outer;         // host [N] (may be NULL): outer iterations
  int32_t* vstatus;       // host [N]: 0 converged, 1 iteration cap, 2 not contracting
  int start;              // 0: cold - lambda = 0.5 everywhere (the reference's, inference.py:302), rho is not read; 1: rho is the start, the mode
                          // search begins at zero; 2: rho is a previous optimum, the mode search begins at its variational mean -K C_big (lambda - y);
                          // 3: like 2 with the previous optimum taken from the device (lam_keep), rho is not read
  double* lam_out;        // host [N][q*T] (may be NULL): the optimal lambda itself
};


// ---- shared host-side helpers (definitions: the translation unit named in the comment) ---------------------------------------
// core.hip: staging, copies, workspace plans, parameter-derived operators, trial lists, snapshots
int ensure_hbuf(pgpfa_ctx* c, size_t len);
int ensure_hibuf(pgpfa_ctx* c, size_t len);
void prof_harvest(Prof& P, bool all);
hipEvent_t prof_event(Prof& P);
void prof_begin(pgpfa_ctx* c, int tag, double flops);
void prof_end(pgpfa_ctx* c);
void prof_collect(pgpfa_ctx* c);
int alloc_cholws(pgpfa_ctx* c, pgpfa::CholWS* w, int nslots, int npad, bool with_mt, size_t slab_elems = 0, bool zero_mt = true, size_t mt_elems = 0);
size_t lowrank_slab_elems(const pgpfa_ctx* c);
bool lowrank_pays(const pgpfa_ctx* c);
bool want_lowrank(const pgpfa_ctx* c);
size_t ld_bytes(const pgpfa_ctx* c);
size_t per_slot_bytes(const pgpfa_ctx* c, size_t slab_elems, size_t mt_elems);
int free_workspace(pgpfa_ctx* c);
int arena_grow(pgpfa_ctx* c, size_t need, double budget_ms = 0.0, size_t must = 0);
void arena_release(pgpfa_ctx* c, bool unmap = false);
int ensure_workspace(pgpfa_ctx* c, bool plan_lr);
int upload_list(pgpfa_ctx* c, int* dst, const std::vector<int>& v);
int copy_dev(pgpfa_ctx* c, void* dst, const void* src, size_t bytes);
int dl_flush(pgpfa_ctx* c);
int dl_enqueue(pgpfa_ctx* c, void* host, const void* dev, size_t bytes);
int download(pgpfa_ctx* c, double* host, const double* dev, size_t n);
int upload(pgpfa_ctx* c, double* dev, const double* host, size_t n);
int upload_nosync(pgpfa_ctx* c, void* dev, const void* host, size_t bytes);
int build_kinv(pgpfa_ctx* c);
int launch_pivchol(pgpfa_ctx* c, hipStream_t st);
int build_lowrank(pgpfa_ctx* c, bool pivchol_launched = false);
int resolve_trials(pgpfa_ctx* c, int n, const int32_t* idx, Trials* out, bool distinct = false);
void counts_changed(pgpfa_ctx* c, const std::vector<int>* trials);
int ensure_high_plane(pgpfa_ctx* c);
int ready(pgpfa_ctx* c);
int ready_estep(pgpfa_ctx* c, bool allow_lowrank);
void snapshot_params(pgpfa_ctx* c, const std::vector<int>& trials);
int with_snapshot(pgpfa_ctx* c, int id, const std::function<int()>& fn);
int remember_trials(pgpfa_ctx* c, const std::vector<int>& v);
int get_slabs(pgpfa_ctx* c, const double* src, double* out);
// linalg.hip: the GEMM kernel's launcher with profiling / tile choice / split-K, blocked factorisation and triangular inverse
double gemm_flops(const pgpfa::GemmP& g);
int gemm(pgpfa_ctx* c, bool transb, pgpfa::GemmP g, bool f32 = false);
int factor(pgpfa_ctx* c, const pgpfa::CholWS& w, const int* slots, int nb, bool f32 = false);
int inverse_t(pgpfa_ctx* c, const pgpfa::CholWS& w, const int* slots, int nb, bool f32 = false);
// estep.hip: Poisson pass, prior products, shared preconditioner, the E-step driver
int poisson(pgpfa_ctx* c, const int* d_list, int nl, const double* X, double* G, double* W, double* flik, int full);
int prior_mv(pgpfa_ctx* c, const int* d_list, int nl, const double* in, double* out, const double* mat = nullptr);
int prior_mv_all(pgpfa_ctx* c, int nb, const double* in, double* out, const double* mat = nullptr, const int* skip = nullptr,
                 const int* cols = nullptr, int ncols = 0);
int assemble(pgpfa_ctx* c, const int* d_list, int nl, double diag_scale = 1.0);
int bin_blocks(pgpfa_ctx* c, const double* W, long long sW, double* G, double* Wt, long long sO, int nslots, double* ldet);
int estep_impl(pgpfa_ctx* c, const Trials& tr, int warm_start, bool allow_lr, double* obj_sum, int32_t* iters, int32_t* status,
               const LooJob* loo = nullptr, const VarJob* var = nullptr);
// cov.hip: posterior covariance blocks (dense and low-rank engines), blocks rebuilt on demand
int ensure_mt_clean(pgpfa_ctx* c);
int ensure_vsmgp_buffer(pgpfa_ctx* c);
int posterior_blocks_lowrank(pgpfa_ctx* c, int nb, bool want_vsmgp, bool accumulate, double* logdet_out = nullptr);
int posterior_blocks(pgpfa_ctx* c, int nb, double diag_scale, bool want_vsmgp, bool accumulate = false);
int post_vsm_from_mt(pgpfa_ctx* c, int nslots);        // post_vsm[t] of the first nslots slots from their dense L^-T slabs
int ensure_trial_vsmgp(pgpfa_ctx* c, const std::vector<int>& trials);
// dual.hip
int ensure_lambda(pgpfa_ctx* c);
int dual_common(pgpfa_ctx* c, int nb, std::vector<double>* sB, std::vector<double>* sD, std::vector<double>* vKv);
int dual_jitter(pgpfa_ctx* c, int nb);
int dual_eval_slots(pgpfa_ctx* c, int nb, const std::vector<int>& tos, bool want_grad, double* cost, bool tolerate = false);
int var_offsets(pgpfa_ctx* c, int nb, double* out);
int check_distinct(const std::vector<int>& v);
int post_cov_dual_impl(pgpfa_ctx* c, int trial, double* out);
// misc.hip
int allreduce_dev(pgpfa_ctx* c, double* buf, size_t count);

