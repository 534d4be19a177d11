// libpgpfa_hip.so - C-ABI (include/pgpfa.h) + host-side orchestration of the HIP kernels.
// One context = one GPU = one stream.  All heavy state (counts, modes, posterior blocks,
// factor slabs) stays resident in HBM between calls.
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
#include "../../include/pgpfa.h"
#include "chol.h"
#include "gemm.h"
#include "model.h"
#include "mstep.h"
#include "pcg.h"
#include "dual.h"
#include "sample.h"
#include "split.h"
#include "thin.h"

using namespace pgpfa;

namespace {

thread_local std::string g_err;
thread_local unsigned long long g_fail_count = 0;   // failures reported on this thread (queued read-backs of a failed call are void: dl_enqueue / dl_flush)

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  ++g_fail_count;
  return 1;
}

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
enum { TAG_GEMM = 0, TAG_POTRF = 1, TAG_SOLVE = 2, TAG_POISSON = 3, TAG_ASSEMBLE = 4, TAG_VSM = 5, TAG_CD = 6, TAG_N };

}  // namespace

struct pgpfa_ctx {
  int device = 0, q = 0, p = 0, T = 0, R = 0, n = 0, npad = 0, ld = 0, Tp = 0;
  double bin = 10.0, eps = 1e-3;
  hipStream_t st = nullptr;
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
  int pcg_form = 1;                              // host-free inner iteration (pcg.h): 1 two tile-parallel kernels per step, no prior mat-vec (pcg_cg_a/b_kernel);
                                                 // 0 the split kernels of round 3 with K^-1 p as a product
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
  std::vector<int> rk, roff;                      // ranks padded to 16, offsets
  int rtot = 0, rpad = 0;
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
  int mix_slot = 2;                           // option mix_slot: the mixing pass of the split form with a thread per bin and whole columns per workgroup (split.h)
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

namespace {

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

int ensure_hbuf(pgpfa_ctx* c, size_t len) {
  if (len <= c->hbuf_len) return 0;
  if (c->hbuf) hipHostFree(c->hbuf);
  c->hbuf = nullptr;
  HIPC(hipHostMalloc((void**)&c->hbuf, len * sizeof(double)));
  c->hbuf_len = len;
  return 0;
}
int ensure_hibuf(pgpfa_ctx* c, size_t len) {
  if (len <= c->hibuf_len) return 0;
  if (c->hibuf) hipHostFree(c->hibuf);
  c->hibuf = nullptr;
  HIPC(hipHostMalloc((void**)&c->hibuf, len * sizeof(int)));
  c->hibuf_len = len;
  return 0;
}

// ---- profiling (HIP events on the context stream; summed on demand) -------------------------------
// Finished launches are read back (hipEventQuery, no synchronisation) and their events recycled while the run goes on,
// so the number of outstanding events stays bounded however long profiling stays switched on.
void prof_harvest(Prof& P, bool all) {
  while (!P.recs.empty()) {
    Prof::Rec& r = P.recs.front();
    if (!all && hipEventQuery(r.e1) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      P.ms[r.tag] += ms;
      P.flops[r.tag] += r.flops;
      P.count[r.tag] += 1;
      if (ms > P.max_ms[r.tag]) { P.max_ms[r.tag] = ms; P.max_flops[r.tag] = r.flops; }
      if (!r.shape.empty()) { Prof::Shape& sh = P.shapes[r.shape]; sh.ms += ms; sh.flops += r.flops; sh.count += 1; }
    }
    P.idle.push_back(r.e0);
    P.idle.push_back(r.e1);
    P.recs.pop_front();
  }
  (void)hipGetLastError();           // hipEventQuery reports hipErrorNotReady through the sticky error as well
}
hipEvent_t prof_event(Prof& P) {
  if (P.idle.empty()) {
    hipEvent_t e;
    hipEventCreate(&e);
    P.pool.push_back(e);
    return e;
  }
  hipEvent_t e = P.idle.back();
  P.idle.pop_back();
  return e;
}
void prof_begin(pgpfa_ctx* c, int tag, double flops) {
  Prof& P = c->prof;
  if (!P.on || (P.only_tag >= 0 && tag != P.only_tag)) return;
  if (P.recs.size() >= 256 && (P.recs.size() & 63) == 0) prof_harvest(P, false);
  Prof::Rec r{tag, prof_event(P), prof_event(P), flops, std::string()};
  hipEventRecord(r.e0, c->st);
  P.recs.push_back(r);
  P.open = true;
}
void prof_end(pgpfa_ctx* c) {
  Prof& P = c->prof;
  if (!P.on || !P.open) return;
  hipEventRecord(P.recs.back().e1, c->st);
  P.open = false;
}
void prof_collect(pgpfa_ctx* c) {
  Prof& P = c->prof;
  if (P.recs.empty()) return;
  hipStreamSynchronize(c->st);
  prof_harvest(P, true);
}

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
int gemm(pgpfa_ctx* c, bool transb, GemmP g, bool f32 = false) {
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
int factor(pgpfa_ctx* c, const CholWS& w, const int* slots, int nb, bool f32 = false) {
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

int inverse_t(pgpfa_ctx* c, const CholWS& w, const int* slots, int nb, bool f32 = false) {
  auto at = [f32](double* base, size_t off) { return f32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(base) + off) : base + off; };
  const int np = w.npad, ld = w.ld;
  const int na = (w.nact > 0 && w.nact <= np) ? w.nact : np;   // rows >= na of Mt are identity padding
  for (int j0 = 0; j0 < np; j0 += NB) {
    if (f32)
      hipLaunchKernelGGL(diag_transpose_kernel_t<float>, dim3(nb), dim3(256), 0, c->st, reinterpret_cast<float*>(w.Mt), w.sM, ld, j0,
                         reinterpret_cast<const float*>(w.Dinv), w.sD, slots);
    else
      hipLaunchKernelGGL(diag_transpose_kernel_t<double>, dim3(nb), dim3(256), 0, c->st, w.Mt, w.sM, ld, j0, w.Dinv, w.sD, slots);
    if (j0 == 0) continue;
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

// allocate a factor workspace: nslots slabs of ld x ld (+ Mt), diagonal inverses, scratch panel
int alloc_cholws(pgpfa_ctx* c, CholWS* w, int nslots, int npad, bool with_mt, size_t slab_elems = 0, bool zero_mt = true, size_t mt_elems = 0) {
  w->npad = npad;
  w->ld = npad;
  const size_t slab = slab_elems ? slab_elems : (size_t)npad * npad;
  const size_t slab_mt = mt_elems ? mt_elems : slab;
  const size_t slack = (size_t)256 * npad;
  w->sH = slab; w->sM = slab_mt; w->sD = (size_t)npad * NB; w->sP = (size_t)npad * NB;
  CHK(dmalloc(c, &w->H, slab * nslots + slack));
  if (with_mt) CHK(dmalloc(c, &w->Mt, slab_mt * nslots + slack, zero_mt)); else w->Mt = nullptr;
  CHK(dmalloc(c, &w->Dinv, w->sD * nslots + slack));
  CHK(dmalloc(c, &w->P, w->sP * nslots + slack));
  CHK(dmalloc(c, &w->info, nslots, true));
  return 0;
}

// per-slot scratch (doubles) the low-rank covariance engine needs inside a factor slab
size_t lowrank_slab_elems(const pgpfa_ctx* c) {
  // Yt (ld x rpad), then either the staging of per-trial blocks (Tp x rpad + T^2) or the single-precision correction D of the split
  // accumulation (ld x rpad floats)
  const size_t yt = (size_t)c->ld * c->rpad;
  const size_t need = yt + std::max((size_t)c->Tp * c->rpad + (size_t)c->T * c->T, yt / 2 + 64);
  return std::max(need, (size_t)c->rpad * c->rpad);
}

// engine choice: the low-rank form pays when r << n (long timescales); the dense form is the general one
bool lowrank_pays(const pgpfa_ctx* c) {
  const double n = c->n, r = c->rpad, T = c->T, p = c->p;
  if (c->p > WIDE_MAX || c->rpad < NB || c->rpad * 2 > c->npad) return false;
  if (lowrank_slab_elems(c) > (size_t)c->ld * c->ld) return false;
  const double dense = 0.72 * n * n * n;
  const double lr = 6.0 * T * r * r + 0.7 * r * r * r + p * T * T * r;
  return lr < 0.5 * dense;
}

// (cov_mode 2 forces the low-rank engine at any size it supports - its slabs are sized for what it needs, not by the dense ld x ld; the
// padded rank must fit the n-sized buffers of the shared preconditioner, which near-full-rank priors - timescales of a bin or two,
// every latent's rank rounded up to 16 - can exceed: those run the dense engine)
bool want_lowrank(const pgpfa_ctx* c) { return c->cov_mode == 2 ? (c->p <= WIDE_MAX && c->rpad >= NB && c->rpad <= c->ld) : (c->cov_mode == 0 && lowrank_pays(c)); }

size_t ld_bytes(const pgpfa_ctx* c) { return (size_t)c->ld * c->ld * sizeof(double); }

size_t per_slot_bytes(const pgpfa_ctx* c, size_t slab_elems, size_t mt_elems) {
  const size_t ld = c->ld;
  size_t dbl = slab_elems + mt_elems + 2 * ld * NB + 12 * ld + 3 * (size_t)c->T * c->p * c->p + (size_t)c->T * (c->p * (c->p + 1) / 2) / 2 + 3 * ((c->T + 63) / 64) + 32;
  return dbl * sizeof(double);
}

// free every workspace allocation (everything allocated after the persistent state)
int free_workspace(pgpfa_ctx* c) {
  if (c->B == 0) return 0;
  HIPC(hipStreamSynchronize(c->st));
  while (c->allocs.size() > c->ws_mark) { hipFree(c->allocs.back()); c->allocs.pop_back(); }
  c->B = 0;
  c->lamd = c->dgrad = c->dpart = c->ldet_buf = c->voff = nullptr;
  c->dual_scr = nullptr;
  c->commbuf = nullptr; c->commbuf_len = 0;
  c->mt_dirty = false;
  return 0;
}

// Grow the arena to at least `need` bytes.  Preferred: map more physical memory behind the reserved address range (the arena does not
// move, the bytes already mapped are not cleared again).  Fallback when the virtual-memory calls are not available: free and
// re-allocate at the size needed.
void arena_release(pgpfa_ctx* c);

// Reserved address ranges of closed contexts, kept for the next context of this process instead of being handed back:
// hipMemAddressFree crashed inside the runtime about once in ten runs of a test sequence that opens and closes a few dozen contexts
// (native backtrace: arena_release -> hipMemAddressFree -> libamdhip64; never under a debugger).  A range is address space only - its
// physical chunks are unmapped and released when the context closes - and there are never more ranges than contexts alive at once.
std::mutex g_va_mu;
std::vector<std::pair<void*, size_t>> g_va_free;

int arena_grow(pgpfa_ctx* c, size_t need) {
  g_err.clear();
  if (c->vmm == 0) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t gran = 0, free_b = 0, total_b = 0;
    void* va = nullptr;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0 &&
        hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t va_size = (total_b + gran - 1) / gran * gran;
      {
        std::lock_guard<std::mutex> lk(g_va_mu);
        for (size_t i = 0; i < g_va_free.size(); ++i)
          if (g_va_free[i].second == va_size) { va = g_va_free[i].first; g_va_free.erase(g_va_free.begin() + i); break; }
      }
      if (va || (hipMemAddressReserve(&va, va_size, gran, nullptr, 0) == hipSuccess && va)) {
        c->arena = reinterpret_cast<char*>(va); c->va_size = va_size; c->vmm_gran = gran; c->vmm = 1; c->arena_cap = 0;
      }
    }
    if (c->vmm != 1) { (void)hipGetLastError(); c->vmm = -1; }
  }
  if (c->vmm == 1) {
    const size_t gran = c->vmm_gran;
    // physical chunks of ONE granule each (one hipMemCreate / hipMemMap / hipMemSetAccess per chunk; 1 GiB by default).  Measured on this
    // stack: hipMemSetAccess returns "invalid argument" for a chunk whose size differs from the first one mapped into the range (13 chunks
    // of 4 GiB, then a 1-GiB remainder: fails; 1 GiB then 4 GiB: fails), so every chunk has the same size.
    const size_t G = std::max(gran, c->vmm_granule) / gran * gran;
    size_t want = (need - c->arena_cap + G - 1) / G * G;
    const char* what = "address range exhausted";
    hipError_t err = hipSuccess;
    bool ok = c->arena_cap + want <= c->va_size;
    const size_t piece_max = G;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    hipMemAccessDesc desc{};
    desc.location = prop.location;
    desc.flags = hipMemAccessFlagsProtReadWrite;
    while (ok && want > 0) {
      const size_t add = std::min(want, std::max(piece_max, gran));
      hipMemGenericAllocationHandle_t h;
      err = hipMemCreate(&h, add, &prop, 0);
      if (err != hipSuccess) { what = "hipMemCreate"; ok = false; break; }
      err = hipMemMap(c->arena + c->arena_cap, add, 0, h, 0);
      if (err != hipSuccess) { what = "hipMemMap"; hipMemRelease(h); ok = false; break; }
      err = hipMemSetAccess(c->arena + c->arena_cap, add, &desc, 1);
      if (err != hipSuccess) { what = "hipMemSetAccess"; hipMemUnmap(c->arena + c->arena_cap, add); hipMemRelease(h); ok = false; break; }
      c->vmm_chunks.emplace_back(h, add);
      c->arena_cap += add;
      c->bytes += add;
      want -= add;
    }
    c->info["arena_bytes"] = (double)c->arena_cap;
    if (ok) return 0;
    (void)hipGetLastError();
    // (the arena only grows between plans, when nothing in it is live: give the range back and carry on with one plain allocation)
    c->info["arena_vmm_failed"] = 1.0;
    std::fprintf(stderr, "pgpfa: workspace arena: %s failed (%s) growing from %zu to %zu bytes; falling back to hipMalloc\n", what,
                 hipGetErrorString(err), c->arena_cap, need);
    c->bytes -= c->arena_cap;
    arena_release(c);
    (void)hipGetLastError();
    c->va_size = 0; c->vmm = -1;
  }
  if (c->arena) { hipFree(c->arena); c->bytes -= c->arena_cap; }
  c->arena = nullptr; c->arena_cap = 0;
  if (hipMalloc((void**)&c->arena, need) != hipSuccess) { (void)hipGetLastError(); c->arena = nullptr; return 1; }
  c->arena_cap = need;
  c->bytes += need;
  c->info["arena_bytes"] = (double)c->arena_cap;
  return 0;
}

void arena_release(pgpfa_ctx* c) {
  if (c->vmm == 1) {
    size_t off = 0;
    for (auto& ch : c->vmm_chunks) { hipMemUnmap(c->arena + off, ch.second); hipMemRelease(ch.first); off += ch.second; }
    c->vmm_chunks.clear();
    if (c->arena) {
      std::lock_guard<std::mutex> lk(g_va_mu);
      g_va_free.emplace_back(c->arena, c->va_size);
    }
  } else if (c->arena) {
    hipFree(c->arena);
  }
  c->arena = nullptr; c->arena_cap = 0;
}

// Workspace plan.  "dense": factor slabs of ld x ld per slot (the general engine, per-trial fallback Newton, post_cov,
// dual variational).  "low-rank": slabs only as large as the r x r systems and their products need, so that ~7x more
// trials fit in one chunk.  Switching plans reallocates the workspace (persistent state is untouched).
int ensure_workspace(pgpfa_ctx* c, bool plan_lr) {
  const size_t dense = (size_t)c->ld * c->ld;
  const size_t slab = plan_lr ? (lowrank_slab_elems(c) + 1023) / 1024 * 1024 : dense;
  // the L^-T slabs only ever hold r x r under the low-rank plan (the big slab is the one that carries Yt)
  const size_t mt = plan_lr ? ((size_t)c->rpad * c->rpad + 1023) / 1024 * 1024 : dense;
  // the chunk is sized for the largest trial list seen so far, not for all R resident trials: minibatch EM over a large
  // resident set then keeps one chunk with generous rank head-room instead of re-planning as the ranks grow
  const int target = (c->want_slots > 0) ? std::min(c->want_slots, c->R) : c->R;
  if (c->B > 0 && c->plan_lowrank == plan_lr && slab <= c->slab_elems && mt <= c->mt_elems && (c->B >= target || c->B_capped)) return 0;
  CHK(free_workspace(c));
  c->ws_mark = c->allocs.size();
  c->plan_lowrank = plan_lr;
  size_t free_b = 0, total_b = 0;
  HIPC(hipMemGetInfo(&free_b, &total_b));
  const size_t avail = (size_t)(0.85 * (double)(free_b + c->arena_cap));     // the arena's bytes are ours to re-partition
  size_t budget = avail;
  {
    const size_t shared = (3 * ld_bytes(c) + 1024 * (size_t)c->ld * sizeof(double) * 4);
    budget = budget > shared ? budget - shared : 0;
  }
  // Low-rank plan: the learnt timescales of a fit move, and the ranks with them.  The slabs get head-room for the ranks to grow by
  // `arena_headroom` (2: Yt slab x 2, r x r slab x 4) before the plan has to be re-made, when that fits next to the whole
  // trial list; a re-plan re-partitions the arena and maps more physical memory into it if it must (only the new bytes cost).
  c->slab_elems = slab;
  c->mt_elems = mt;
  if (plan_lr) {
    const double h = std::max(1.0, c->arena_headroom);
    const size_t want_slab = std::min(dense, (size_t)((double)slab * h) / 1024 * 1024), want_mt = std::min(dense, (size_t)((double)mt * h * h) / 1024 * 1024);
    if (per_slot_bytes(c, want_slab, want_mt) * (size_t)std::max(target, 1) <= budget) { c->slab_elems = std::max(slab, want_slab); c->mt_elems = std::max(mt, want_mt); }
  }
  const size_t per = per_slot_bytes(c, c->slab_elems, c->mt_elems);
  long long B = (long long)(budget / per);
  if (c->chunk_opt > 0) B = std::min<long long>(B, c->chunk_opt);
  c->B_capped = B < target;                          // memory (or chunk_trials) bound: asking again would not give more
  if (B >= target) {
    B = target;                                      // everything in one chunk
  } else if (B >= 16) {
    B = B / 8 * 8;                                   // groups of 8 slots map onto the 8 XCDs
    const long long nchunks = (target + B - 1) / B;  // balance the chunks
    const long long Bb = ((target + nchunks - 1) / nchunks + 7) / 8 * 8;
    B = std::min(B, Bb);
  }
  if (B < 1) return fail("not enough device memory for one trial slab (%zu bytes needed, %zu free)", per, free_b);
  c->B = (int)B;
  // pass 1 measures the plan, then the arena is grown if it has to be, pass 2 hands out the pointers
  auto carve = [&]() -> int {
  // (the dense engine needs the strictly lower part of its L^-T slabs zero; the low-rank engine fills its r x r views itself, so under
  // that plan the ~10^11-byte clear is left out and the slabs are marked dirty for a later dense use)
  CHK(alloc_cholws(c, &c->ws, c->B, c->npad, true, c->slab_elems, !plan_lr, c->mt_elems));
  c->ws.nact = round_up(c->n, 64);
  const size_t ld = c->ld, nB = c->B;
  const size_t nBs = nB + 128;                    // slack: multi-RHS GEMM tiles read up to 127 slots past the end
  // all slot vectors: zero-initialised with slack (rows >= n stay zero; GEMM tiles over-read into finite data)
  CHK(dmalloc(c, &c->Xc, ld * nBs, true)); CHK(dmalloc(c, &c->Xt, ld * nBs, true));
  CHK(dmalloc(c, &c->KX, ld * nBs, true)); CHK(dmalloc(c, &c->KD, ld * nBs, true));
  CHK(dmalloc(c, &c->Gl, ld * nBs, true)); CHK(dmalloc(c, &c->Glt, ld * nBs, true));
  CHK(dmalloc(c, &c->Gt, ld * nBs, true)); CHK(dmalloc(c, &c->Dl, ld * nBs, true));
  CHK(dmalloc(c, &c->Rv, ld * nBs, true)); CHK(dmalloc(c, &c->Zv, ld * nBs, true));
  CHK(dmalloc(c, &c->Pv, ld * nBs, true)); CHK(dmalloc(c, &c->Qv, ld * nBs, true));
  CHK(dmalloc(c, &c->Sv, ld * nBs, true)); CHK(dmalloc(c, &c->cg_scal, 4 * nB, true));
  CHK(alloc_cholws(c, &c->sws, 1, c->npad, true));
  c->sws.nact = round_up(c->n, 64);
  CHK(dmalloc(c, &c->sU, ld * ld + 256 * ld, true));
  CHK(dmalloc(c, &c->sDinvT, ld * NB + 256 * ld));
  CHK(dmalloc(c, &c->Wbar, (size_t)c->T * c->p * c->p));
  CHK(dmalloc(c, &c->Gbin, (size_t)c->T * c->p * c->p * nB));
  CHK(dmalloc(c, &c->sc_rz, nB)); CHK(dmalloc(c, &c->sc_pq, nB * (size_t)((c->T + 63) / 64)));
  // per-slot scalars the Newton drivers read back together: one contiguous block, one download
  CHK(dmalloc(c, &c->sc_pack, 7 * nB));
  c->sc_dec = c->sc_pack; c->sc_smax = c->sc_pack + nB; c->sc_qxx = c->sc_pack + 2 * nB; c->sc_qdx = c->sc_pack + 3 * nB;
  c->sc_qdd = c->sc_pack + 4 * nB; c->sc_rr = c->sc_pack + 5 * nB; c->sc_rr0 = c->sc_pack + 6 * nB;
  const size_t wlen = (size_t)c->T * c->p * c->p;
  CHK(dmalloc(c, &c->W, wlen * nB)); CHK(dmalloc(c, &c->Wt, wlen * nB));
  CHK(dmalloc(c, &c->fpart, (size_t)((c->T + 63) / 64) * nB));
  CHK(dmalloc(c, &c->sc_part2, 3 * (size_t)((c->T + 63) / 64) * nB));
  CHK(dmalloc(c, &c->W32, (size_t)round_up(c->T, 32) * (c->p * (c->p + 1) / 2) * nB + 64));   // (rows of the component-major form start on 128-byte lines)
  CHK(dmalloc(c, &c->pcgctl, 1, true));
  CHK(dmalloc(c, &c->live, 2 * nB)); CHK(dmalloc(c, &c->pcg_ratio, nB, true)); CHK(dmalloc(c, &c->pcg_eta, nB, true));
  CHK(dmalloc(c, &c->GbT, (size_t)c->T * (c->p * (c->p + 1) / 2) + 64)); CHK(dmalloc(c, &c->WbT, (size_t)c->T * (c->p * (c->p + 1) / 2) + 64));
  CHK(dmalloc(c, &c->sc_f, nB));
  CHK(dmalloc(c, &c->sc_alpha, nB));
  CHK(dmalloc(c, &c->trial_of_slot, nB)); CHK(dmalloc(c, &c->list_a, nB)); CHK(dmalloc(c, &c->list_b, nB));
  CHK(dmalloc(c, &c->mask_of_slot, nB));
  CHK(dmalloc(c, &c->ident, nB));
  return 0;
  };
  c->arena_mode = 1; c->arena_off = 0;
  int rc_carve = carve();
  c->arena_mode = 0;
  if (rc_carve) return rc_carve;
  const size_t need = c->arena_off + ((size_t)1 << 20);
  if (need > c->arena_cap) {
    HIPC(hipStreamSynchronize(c->st));
    if (arena_grow(c, need)) {
      c->B = 0;
      const std::string why = g_err;
      return fail("not enough device memory for the chunk workspace (%zu bytes needed, %zu free)%s%s", need, free_b, why.empty() ? "" : ": ", why.c_str());
    }
  }
  c->arena_mode = 2; c->arena_off = 0;
  rc_carve = carve();
  c->arena_mode = 0;
  if (rc_carve) { c->B = 0; return rc_carve; }
  std::vector<int> id(c->B);
  for (int i = 0; i < c->B; ++i) id[i] = i;
  HIPC(hipMemcpyAsync(c->ident, id.data(), sizeof(int) * c->B, hipMemcpyHostToDevice, c->st));
  HIPC(hipStreamSynchronize(c->st));
  if (plan_lr) c->mt_dirty = true;
  c->info["chunk_trials"] = c->B;
  c->info["plan_lowrank"] = c->plan_lowrank ? 1.0 : 0.0;
  return 0;
}

int upload_nosync(pgpfa_ctx* c, void* dev, const void* host, size_t bytes);
int upload_list(pgpfa_ctx* c, int* dst, const std::vector<int>& v) {
  if (v.empty()) return 0;
  if (v.size() * sizeof(int) <= (size_t)65536) return upload_nosync(c, dst, v.data(), v.size() * sizeof(int));   // through the pinned ring, no synchronisation
  CHK(ensure_hibuf(c, v.size()));
  // staged through pinned memory; the stream sync in callers orders reuse of the staging buffer
  std::memcpy(c->hibuf, v.data(), v.size() * sizeof(int));
  HIPC(hipMemcpyAsync(dst, c->hibuf, v.size() * sizeof(int), hipMemcpyHostToDevice, c->st));
  HIPC(hipStreamSynchronize(c->st));
  return 0;
}

// Read-backs.  A device-to-host copy into pageable memory makes the runtime drain the stream from the host first and then run a staging
// copy (measured in the kernel trace: 100-280 us of device idle time in front of every such copy); a copy into pinned memory is just
// another stream operation.  Small read-backs therefore land in a pinned staging area and are copied out after ONE synchronisation:
// dl_enqueue queues a copy (several may be queued back to back), dl_flush waits and hands the bytes out.
constexpr size_t DL_STAGE_BYTES = (size_t)4 << 20;
constexpr size_t COPY_KERNEL_MAX = (size_t)256 << 10;        // copies up to this size go through copy_words_kernel when the staging memory is mapped

// dst <- src, bytes a multiple of 4 (every small copy of the library is): one or a few workgroups; either side may be host-mapped memory
__global__ __launch_bounds__(256) void copy_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, size_t nwords) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy_vec16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// device-to-device copy on the context's stream as a kernel (a runtime copy between two kernels costs hundreds of microseconds of idle time:
// see the note at pgpfa_ctx::copy_kernels); any size, 16-byte vectors when both sides allow
static int copy_dev(pgpfa_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  if (c->copy_kernels && (bytes & 3) == 0 && (((size_t)dst | (size_t)src) & 3) == 0) {
    if ((bytes & 15) == 0 && (((size_t)dst | (size_t)src) & 15) == 0) {
      const size_t nv = bytes / 16;
      hipLaunchKernelGGL(copy_vec16_kernel, dim3((unsigned)std::min<size_t>((nv + 255) / 256, 4096)), dim3(256), 0, c->st, reinterpret_cast<uint4*>(dst),
                         reinterpret_cast<const uint4*>(src), nv);
    } else {
      const size_t nw = bytes / 4;
      hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<size_t>((nw + 255) / 256, 4096)), dim3(256), 0, c->st, reinterpret_cast<unsigned*>(dst),
                         reinterpret_cast<const unsigned*>(src), nw);
    }
    HIPC(hipGetLastError());
    return 0;
  }
  HIPC(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->st));
  return 0;
}
// everything enqueued before this kernel has completed (in-order stream) when the host reads `value` here
__global__ void raise_seq_kernel(volatile unsigned* __restrict__ seq, unsigned value) {
  __threadfence_system();
  *seq = value;
  __threadfence_system();
}

// wait until the stream has drained: by the sequence number in mapped memory when the copies run as kernels, else hipStreamSynchronize
static int stream_drain(pgpfa_ctx* c) {
  if (c->copy_kernels && c->h_seq) {
    const unsigned want = ++c->seq_next;
    hipLaunchKernelGGL(raise_seq_kernel, dim3(1), dim3(1), 0, c->st, (volatile unsigned*)c->d_seq, want);
    if (hipGetLastError() == hipSuccess) {
      const auto t0 = std::chrono::steady_clock::now();
      long spins = 0;
      while (*(volatile unsigned*)c->h_seq != want) {
        if ((++spins & 0xfff) == 0) {
          const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          // a faulted kernel never raises the number: past 50 ms ask the runtime now and then, which also reports the fault
          if (el > 0.05 && hipStreamQuery(c->st) != hipErrorNotReady) break;
        }
      }
      if (*(volatile unsigned*)c->h_seq == want) return 0;
    }
  }
  const hipError_t e_sync = hipStreamSynchronize(c->st);
  if (e_sync != hipSuccess) return fail("%s:%d hipStreamSynchronize -> %s", __FILE__, __LINE__, hipGetErrorString(e_sync));
  return 0;
}
// The queue holds raw host pointers (stack locals, vector buffers, caller arrays) that are only good inside the call that queued them: a
// failure between dl_enqueue and dl_flush - a CHK / HIPC that returned, or the synchronisation below - voids the whole queue, so that no later
// flush copies into memory that call has given back (dl_drop_stale: anything queued before the last fail() of this thread is dropped).
static void dl_drop_stale(pgpfa_ctx* c) {
  if (!c->dl_pending.empty() && c->dl_fail_mark != g_fail_count) { c->dl_pending.clear(); c->dl_used = 0; }
}
int dl_flush(pgpfa_ctx* c) {
  dl_drop_stale(c);
  const int rc_sync = stream_drain(c);
  c->ring_pending = 0;
  if (rc_sync) {
    c->dl_pending.clear();
    c->dl_used = 0;
    return rc_sync;
  }
  for (const auto& e : c->dl_pending) std::memcpy(e.host, c->dl_stage + e.off, e.bytes);
  c->dl_pending.clear();
  c->dl_used = 0;
  return 0;
}
int dl_enqueue(pgpfa_ctx* c, void* host, const void* dev, size_t bytes) {
  if (bytes == 0) return 0;
  if (!c->dl_stage) {
    if (hipHostMalloc((void**)&c->dl_stage, DL_STAGE_BYTES, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); c->dl_stage = nullptr; }
    if (c->dl_stage && hipHostGetDevicePointer((void**)&c->dl_stage_dev, c->dl_stage, 0) != hipSuccess) { (void)hipGetLastError(); c->dl_stage_dev = nullptr; }
    if (!c->h_seq) {
      if (hipHostMalloc((void**)&c->h_seq, 64, hipHostMallocMapped) != hipSuccess ||
          hipHostGetDevicePointer((void**)&c->d_seq, c->h_seq, 0) != hipSuccess) { (void)hipGetLastError(); c->h_seq = nullptr; c->d_seq = nullptr; }
      if (c->h_seq) *c->h_seq = 0u;
    }
  }
  const size_t need = (bytes + 63) / 64 * 64;
  if (!c->dl_stage || need > DL_STAGE_BYTES) {
    // large (or no staging area): straight into the caller's memory; complete when this returns
    CHK(dl_flush(c));
    const hipError_t e_copy = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->st);
    if (e_copy != hipSuccess) return fail("device-to-host copy of %zu bytes: %s", bytes, hipGetErrorString(e_copy));
    return dl_flush(c);
  }
  dl_drop_stale(c);
  if (c->dl_used + need > DL_STAGE_BYTES) CHK(dl_flush(c));
  if (c->copy_kernels && c->dl_stage_dev && c->h_seq && bytes <= COPY_KERNEL_MAX && (bytes & 3) == 0 && (((size_t)dev) & 3) == 0) {
    const size_t nw = bytes / 4;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<size_t>((nw + 255) / 256, 64)), dim3(256), 0, c->st,
                       reinterpret_cast<unsigned*>(c->dl_stage_dev + c->dl_used), reinterpret_cast<const unsigned*>(dev), nw);
    HIPC(hipGetLastError());
  } else {
    HIPC(hipMemcpyAsync(c->dl_stage + c->dl_used, dev, bytes, hipMemcpyDeviceToHost, c->st));
  }
  if (c->dl_pending.empty()) c->dl_fail_mark = g_fail_count;
  c->dl_pending.push_back({host, c->dl_used, bytes});
  c->dl_used += need;
  return 0;
}
int download(pgpfa_ctx* c, double* host, const double* dev, size_t n) {
  if (c->hbuf && host >= c->hbuf && host < c->hbuf + c->hbuf_len) {      // already pinned
    // (small: through the staging area like any other read-back - the runtime's copy costs more than the extra memcpy)
    if (c->copy_kernels && n * sizeof(double) <= COPY_KERNEL_MAX) {
      CHK(dl_enqueue(c, host, dev, n * sizeof(double)));
      return dl_flush(c);
    }
    HIPC(hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, c->st));
    return dl_flush(c);
  }
  CHK(dl_enqueue(c, host, dev, n * sizeof(double)));
  return dl_flush(c);
}
int upload(pgpfa_ctx* c, double* dev, const double* host, size_t n) {
  // (small: through the pinned ring - the bytes are out of the caller's buffer when this returns, and nothing waits for the device)
  if (n * sizeof(double) <= (size_t)65536) return upload_nosync(c, dev, host, n * sizeof(double));
  HIPC(hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPC(hipStreamSynchronize(c->st));
  c->ring_pending = 0;
  return 0;
}
// Small upload without a synchronisation: the bytes are copied into the next slot of a ring of pinned buffers and sent asynchronously; a
// slot comes round again after RING_N uploads, and every download / synchronising upload in between (there is at least one per Newton
// outer iteration and per line-search round) has drained the stream by then - enforced by the pending count.
constexpr int RING_N = 16;
int upload_nosync(pgpfa_ctx* c, void* dev, const void* host, size_t bytes) {
  if (bytes == 0) return 0;
  if (bytes > c->ring_slot) {
    HIPC(hipStreamSynchronize(c->st));
    if (c->ring) hipHostFree(c->ring);
    c->ring = nullptr; c->ring_dev = nullptr;
    const size_t slot = (bytes + 4095) / 4096 * 4096;
    HIPC(hipHostMalloc((void**)&c->ring, slot * RING_N, hipHostMallocMapped));
    if (hipHostGetDevicePointer((void**)&c->ring_dev, c->ring, 0) != hipSuccess) { (void)hipGetLastError(); c->ring_dev = nullptr; }
    c->ring_slot = slot; c->ring_cur = 0; c->ring_pending = 0;
  }
  if (c->ring_pending >= RING_N - 1) { HIPC(hipStreamSynchronize(c->st)); c->ring_pending = 0; }
  char* slot = c->ring + (size_t)c->ring_cur * c->ring_slot;
  std::memcpy(slot, host, bytes);
  if (c->copy_kernels && c->ring_dev && bytes <= COPY_KERNEL_MAX && (bytes & 3) == 0 && (((size_t)dev) & 3) == 0) {
    const size_t nw = bytes / 4;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<size_t>((nw + 255) / 256, 64)), dim3(256), 0, c->st, reinterpret_cast<unsigned*>(dev),
                       reinterpret_cast<const unsigned*>(c->ring_dev + (size_t)c->ring_cur * c->ring_slot), nw);
    HIPC(hipGetLastError());
  } else {
    HIPC(hipMemcpyAsync(dev, slot, bytes, hipMemcpyHostToDevice, c->st));
  }
  c->ring_cur = (c->ring_cur + 1) % RING_N;
  c->ring_pending += 1;
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

// post_vsm[t] = Gram of the rows (., t) of the panel Mt (ncol columns, column stride ld) for ns slots: matrix-core kernel beyond 10
// latents, vector kernel up to 10.  (f32: the panel is single precision)
template <typename TIN>
void launch_post_vsm(pgpfa_ctx* c, const TIN* Mt, long long sM, int ncol, int ns, int full_range, const int* roff = nullptr, int ts = 0) {
  const int T = c->T, p = c->p;
  if (ts <= 0) ts = T;                                       // row stride between latents in the panel
  if (p > 10 && c->vsm_mfma) {
    const int CB = post_vsm_mfma_cb(p, sizeof(TIN) == 4);
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
  const bool gemm_form = !(c->CCu && c->mfma) && c->mfma && c->dual_gemm && c->dual_tbl && c->lamd && c->dual_scr && !c->mask_active && !c->lam_out_active;
  if (gemm_form) {
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
      hipLaunchKernelGGL(dual_unpack_w_kernel, dim3((unsigned)(((size_t)c->T * np + 255) / 256), nl), dim3(256), 0, c->st, c->dual_scr, c->dual_sscr, W,
                         (long long)c->T * c->p * c->p, c->T, c->p, d_list);
    }
    hipLaunchKernelGGL(sum_tiles_kernel, dim3((nl + 255) / 256), dim3(256), 0, c->st, c->fpart, a.ntile, d_list, nl, flik);
    HIPC(hipGetLastError());
    return 0;
  }
  prof_begin(c, TAG_POISSON, fl);
  if (c->CCu && c->mfma) {
    dispatch_pw(c->p, [&](auto pm) {
      constexpr int PW = decltype(pm)::value;
      if constexpr (PW <= 16)
        hipLaunchKernelGGL(poisson_mfma_kernel<PW>, grid, dim3(256), 0, c->st, a, c->CCu, c->C16, c->qpad);
    });
  } else {
    dispatch_pw(c->p, [&](auto pm) { hipLaunchKernelGGL(poisson_pass_kernel<decltype(pm)::value>, grid, block, 0, c->st, a); });
  }
  prof_end(c);
  hipLaunchKernelGGL(sum_tiles_kernel, dim3((nl + 255) / 256), dim3(256), 0, c->st, c->fpart, a.ntile, d_list, nl, flik);
  HIPC(hipGetLastError());
  return 0;
}

int prior_mv(pgpfa_ctx* c, const int* d_list, int nl, const double* in, double* out, const double* mat = nullptr) {
  hipLaunchKernelGGL(prior_matvec_kernel, dim3(c->p, nl), dim3(256), c->T * sizeof(double), c->st, mat ? mat : c->Kinv, c->Tp, c->T, c->p,
                     in, (long long)c->ld, out, (long long)c->ld, d_list);
  HIPC(hipGetLastError());
  return 0;
}

// out[slot][k] = mat_k * in[slot][k] for ALL slots [0,nb) as one batched MFMA GEMM (batch = latents, N = slots)
// (cols / ncols: only the listed slots - the live ones of a Newton-PCG solve; the product then has ncols columns)
int prior_mv_all(pgpfa_ctx* c, int nb, const double* in, double* out, const double* mat = nullptr, const int* skip = nullptr,
                 const int* cols = nullptr, int ncols = 0) {
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

int assemble(pgpfa_ctx* c, const int* d_list, int nl, double diag_scale = 1.0) {
  prof_begin(c, TAG_ASSEMBLE, 0.0);
  hipLaunchKernelGGL(assemble_h_kernel, dim3(c->npad, nl), dim3(256), 0, c->st, c->ws.H, c->ws.sH, c->ld, c->npad, c->n, c->T, c->Tp,
                     c->p, c->Kinv, c->W, (long long)c->T * c->p * c->p, d_list, diag_scale);
  prof_end(c);
  HIPC(hipGetLastError());
  return 0;
}

// Kinv (and logdet) of the p Gram slabs currently in Kpad, through the production factor kernels
int build_kinv(pgpfa_ctx* c) {
  const size_t slab = (size_t)c->Tp * c->Tp;
  CHK(copy_dev(c, c->kws.H, c->Kpad, slab * c->p * sizeof(double)));
  HIPC(hipMemsetAsync(c->kws.info, 0, sizeof(int) * c->p, c->st));
  CHK(factor(c, c->kws, nullptr, c->p));
  c->logdetK.assign(c->p, 0.0);
  hipLaunchKernelGGL(logdet_batch_kernel, dim3(c->p), dim3(256), 0, c->st, c->kws.H, (long long)c->kws.sH, c->Tp, c->Tp, c->tscal);
  CHK(download(c, c->logdetK.data(), c->tscal, c->p));
  CHK(inverse_t(c, c->kws, nullptr, c->p));
  GemmP g{};
  g.A = c->kws.Mt; g.sA = c->kws.sM; g.lda = c->Tp;
  g.B = c->kws.Mt; g.sB = c->kws.sM; g.ldb = c->Tp;
  g.C = c->Kinv; g.sC = slab; g.ldc = c->Tp;
  g.M = c->Tp; g.N = c->Tp; g.K = c->Tp; g.alpha = 1.0; g.beta = 0.0;
  g.slots = nullptr; g.nbatch = c->p; g.mode = GEMM_FULL; g.kflags = KF_BEGIN_MAXRC;
  CHK(gemm(c, false, g));
  std::vector<int> info(c->p);
  CHK(dl_enqueue(c, info.data(), c->kws.info, sizeof(int) * c->p));
  CHK(dl_flush(c));
  for (int k = 0; k < c->p; ++k)
    if (info[k] != 0) return fail("GP Gram matrix of latent %d is not positive definite (pivot %d)", k, info[k]);
  return 0;
}


// pivoted-Cholesky factors of the RBF part of every Gram matrix and the block tables of the r x r system
// (the pivoted Cholesky itself - p workgroups, a chain of r_k dependent steps each: 1-2 ms with the chip empty - on stream `st`)
int launch_pivchol(pgpfa_ctx* c, hipStream_t st) {
  const int p = c->p, T = c->T, Tp = c->Tp;
  const int rmax = std::min(T, Tp);
  const size_t shm = ((size_t)2 * T + rmax + 16) * sizeof(double) + 16 * sizeof(int);
  if (T > 256)
    hipLaunchKernelGGL((rbf_pivchol_kernel<512, 2>), dim3(p), dim3(1024), shm, st, c->Flr, Tp, T, c->tau, c->bin, c->eps, c->lr_tol, rmax, c->d_rank);
  else
    hipLaunchKernelGGL((rbf_pivchol_kernel<256, 1>), dim3(p), dim3(256), shm, st, c->Flr, Tp, T, c->tau, c->bin, c->eps, c->lr_tol, rmax, c->d_rank);
  HIPC(hipGetLastError());
  return 0;
}

// (pivchol_launched: the kernel is already running on the side stream and c->ev_join marks its end)
int build_lowrank(pgpfa_ctx* c, bool pivchol_launched = false) {
  const int p = c->p, T = c->T, Tp = c->Tp;
  if (pivchol_launched) HIPC(hipStreamWaitEvent(c->st, c->ev_join, 0));
  else CHK(launch_pivchol(c, c->st));
  std::vector<int> r(p);
  CHK(dl_enqueue(c, r.data(), c->d_rank, sizeof(int) * p));
  CHK(dl_flush(c));
  c->rk.assign(p, 0);
  c->roff.assign(p + 1, 0);
  for (int k = 0; k < p; ++k) {
    c->rk[k] = round_up(std::max(r[k], 1), 16);
    c->roff[k + 1] = c->roff[k] + c->rk[k];
  }
  c->rtot = c->roff[p];
  c->rpad = round_up(c->rtot, NB);
  const int nblk = c->rpad / 16;
  std::vector<int> lat(nblk, -1), col(nblk, 0);
  for (int k = 0; k < p; ++k)
    for (int b = c->roff[k] / 16; b < c->roff[k + 1] / 16; ++b) { lat[b] = k; col[b] = b * 16 - c->roff[k]; }
  CHK(upload_list(c, c->d_blk_lat, lat));
  CHK(upload_list(c, c->d_blk_col, col));
  CHK(upload_list(c, c->d_roff, c->roff));
  {
    // Row-tile tables of the two block-diagonal products of the low-rank preconditioner, 64 rows per tile, one latent per tile:
    // F^T (rpad x n): rank rows [roff[k], roff[k+1]) x the latent's bins [kT, (k+1)T) (rounded out to multiples of 16: the
    //   neighbours' columns in these rows are zero);  F (n x rpad): rows [kT, (k+1)T) x the latent's rank columns.
    std::vector<int> tft, tf;
    c->kr_ft_len = 0; c->kr_f_len = 0;
    for (int k = 0; k < p; ++k) {
      const int kb = (k * T) / 16 * 16, ke = std::min(c->npad, round_up((k + 1) * T, 16));
      for (int r0 = c->roff[k]; r0 < c->roff[k + 1]; r0 += 64) { tft.push_back(r0); tft.push_back(c->roff[k + 1]); tft.push_back(kb); tft.push_back(ke); }
      c->kr_ft_len = std::max(c->kr_ft_len, ke - kb);
      for (int i0 = k * T; i0 < (k + 1) * T; i0 += 64) { tf.push_back(i0); tf.push_back((k + 1) * T); tf.push_back(c->roff[k]); tf.push_back(c->roff[k + 1]); }
      c->kr_f_len = std::max(c->kr_f_len, c->rk[k]);
    }
    c->ntab_ft = (int)tft.size() / 4; c->ntab_f = (int)tf.size() / 4;
    if (tft.size() > c->tab_cap || tf.size() > c->tab_cap) return fail("internal: row-tile table overflow");
    CHK(upload_list(c, c->d_kr_ft, tft));
    CHK(upload_list(c, c->d_kr_f, tf));
    // thin.h: F^T t by (latent, 64 rank rows), F v by (latent, 512 bins)
    std::vector<int> hft, hf;
    for (int k = 0; k < p; ++k) {
      // (row groups of a latent of equal size, a multiple of 4 up to 64: 80 rank rows are 40 + 40, not 64 + 16)
      const int ngr = (c->rk[k] + 63) / 64, per = round_up((c->rk[k] + ngr - 1) / ngr, 4);
      for (int m0 = 0; m0 < c->rk[k]; m0 += per) { hft.push_back(k); hft.push_back(m0); hft.push_back(std::min(per, c->rk[k] - m0)); hft.push_back(c->roff[k]); }
      for (int t0 = 0; t0 < T; t0 += 256) { hf.push_back(k); hf.push_back(t0); hf.push_back(c->rk[k]); hf.push_back(c->roff[k]); }
    }
    // Sb u with the same kernel as F^T t: one "latent" of rtot rows and rtot "bins", 64 rows per workgroup
    std::vector<int> hs;
    for (int m0 = 0; m0 < c->rtot; m0 += 64) { hs.push_back(0); hs.push_back(m0); hs.push_back(std::min(64, c->rtot - m0)); hs.push_back(0); }
    c->nthin_s = (int)hs.size() / 4;
    if (hs.size() > c->tab_cap) return fail("internal: thin-product table overflow");
    CHK(upload_list(c, c->d_thin_s, hs));
    c->nthin_ft = (int)hft.size() / 4; c->nthin_f = (int)hf.size() / 4;
    if (hft.size() > c->tab_cap || hf.size() > c->tab_cap) return fail("internal: thin-product table overflow");
    CHK(upload_list(c, c->d_thin_ft, hft));
    CHK(upload_list(c, c->d_thin_f, hf));
  }
  if ((size_t)c->rpad <= (size_t)c->ld) {
    HIPC(hipMemsetAsync(c->Fbig, 0, (size_t)c->ld * c->rpad * sizeof(double), c->st));
    HIPC(hipMemsetAsync(c->FTbig, 0, (size_t)c->rpad * c->ld * sizeof(double), c->st));
    int rkmax = 0;
    for (int k = 0; k < p; ++k) rkmax = std::max(rkmax, c->rk[k]);
    hipLaunchKernelGGL(build_fbig_kernel, dim3(rkmax, p), dim3(128), 0, c->st, c->Flr, Tp, T, c->d_roff, c->Fbig, c->ld, c->FTbig, c->rpad);
    HIPC(hipGetLastError());
  }
  c->info["lowrank_rtot"] = c->rtot;
  c->flr32_valid = false;
  return 0;
}

int allreduce_dev(pgpfa_ctx* c, double* buf, size_t count) {
  if (!c->comm) return 0;
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, c->st);
  if (r != ncclSuccess) return fail("ncclAllReduce failed: %s", ncclGetErrorString(r));
  return 0;
}

struct Trials {
  std::vector<int> v;
};
// distinct = true for every entry point that writes per-trial state (two slots of one chunk scattering to the same
// trial row would race, and the device list of the last E-step holds R entries)
int resolve_trials(pgpfa_ctx* c, int n, const int32_t* idx, Trials* out, bool distinct = false) {
  if (idx == nullptr) {
    out->v.resize(c->R);
    for (int i = 0; i < c->R; ++i) out->v[i] = i;
    return 0;
  }
  if (n < 1) return fail("empty trial list");
  for (int i = 0; i < n; ++i)
    if (idx[i] < 0 || idx[i] >= c->R) return fail("trial index %d out of range [0,%d)", idx[i], c->R);
  if (distinct) {
    if (n > c->R) return fail("trial list of %d entries for %d resident trials", n, c->R);
    std::vector<char> seen(c->R, 0);
    for (int i = 0; i < n; ++i) {
      if (seen[idx[i]]) return fail("trial %d listed twice", idx[i]);
      seen[idx[i]] = 1;
    }
  }
  out->v.assign(idx, idx + n);
  return 0;
}

}  // namespace

// The GEMM addresses A, B and C with one slot index; post_vsmGP is indexed by trial, so the slot
// result goes through a slot-indexed staging slab and is scattered afterwards.
namespace {
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
}  // namespace

extern "C" {

const char* pgpfa_last_error(void) { return g_err.c_str(); }
int pgpfa_version(void) { return 100; }

int pgpfa_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail("hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return 0;
}

int pgpfa_create(pgpfa_ctx** out, int device, int q, int p, int T, int R, double bin_ms) {
  if (!out) return fail("null out pointer");
  *out = nullptr;
  if (q < 1 || p < 1 || T < 1 || R < 1) return fail("invalid sizes q=%d p=%d T=%d R=%d", q, p, T, R);
  if (p > 32) return fail("p=%d latents not supported (max 32)", p);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail("no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail("device %d out of range (%d devices)", device, ndev);
  HIPC(hipSetDevice(device));
  pgpfa_ctx* c = new pgpfa_ctx();
  c->device = device; c->q = q; c->p = p; c->T = T; c->R = R; c->bin = bin_ms;
  c->n = p * T;
  c->npad = round_up(c->n, NB);
  c->ld = c->npad;
  c->Tp = round_up(T, NB);
  hipError_t se = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
  if (se != hipSuccess) { delete c; return fail("hipStreamCreate: %s", hipGetErrorString(se)); }
  if (hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();                                // (no side stream: everything stays on the one stream)
    if (c->st2) { hipStreamDestroy(c->st2); c->st2 = nullptr; }
  }
  int rc = 0;
  const size_t slab = (size_t)c->Tp * c->Tp;
  rc |= dmalloc(c, &c->Y, (size_t)R * q * T);
  rc |= dmalloc(c, &c->C, (size_t)q * p); rc |= dmalloc(c, &c->d, q); rc |= dmalloc(c, &c->tau, p);
  rc |= dmalloc(c, &c->Kpad, slab * p); rc |= dmalloc(c, &c->Kinv, slab * p);
  rc |= dmalloc(c, &c->Xmode, (size_t)R * c->n + 64, true);
  rc |= dmalloc(c, &c->Xprev, (size_t)R * c->n + 64, true);
  c->mode_serial.assign(R, -10); c->prev_serial.assign(R, -10);
  rc |= dmalloc(c, &c->vsm, (size_t)R * T * p * p + 2048, true);
  // c->vsmgp (R*p blocks of T x T: 20 GB at config 3) is allocated on first use: the low-rank engine's default
  // sum-only output never touches it
  rc |= dmalloc(c, &c->Pauto, slab * p, true);
  rc |= dmalloc(c, &c->Pacc, slab * p, true);
  c->gemm_part_len = (size_t)16 << 20;
  rc |= dmalloc(c, &c->gemm_part, c->gemm_part_len);
  c->qpad = round_up(q, 16);
  c->ccu_cols = round_up(p * (p + 1) / 2, 16);
  if (p <= 16) {
    // widths must match the kernel instantiation dispatch_pw picks for p
    dispatch_pw(p, [&](auto pw) { constexpr int PW = decltype(pw)::value; c->ccu_cols = round_up(PW * (PW + 1) / 2, 16); });
    rc |= dmalloc(c, &c->CCu, (size_t)c->qpad * c->ccu_cols + 64, true);
    rc |= dmalloc(c, &c->C16, (size_t)c->qpad * 16 + 64, true);
  }
  c->dual_npd = round_up(p * (p + 1) / 2, 16);
  c->dual_ncol = c->dual_npd + round_up(p, 16);
  rc |= dmalloc(c, &c->dual_tbl, (size_t)round_up(q, 128) * c->dual_ncol + 4096, true);   // (GEMM tiles read whole 128-row blocks of it)
  rc |= dmalloc(c, &c->ppart, (size_t)p * (PACC_SPLITS + 1) * T * T + 256);
  c->vsmgp_ok.assign(R, 0);
  c->trial_snap.assign(R, -1);
  c->trial_dual.assign(R, 0);
  c->lam_resident.assign(R, 0);
  rc |= dmalloc(c, &c->Flr, slab * p + 256 * (size_t)c->Tp, true);
  rc |= dmalloc(c, &c->d_rank, p); rc |= dmalloc(c, &c->d_roff, p + 1);
  c->tab_cap = 4 * ((size_t)c->ld / 64 + 2 * (size_t)p + 4);
  rc |= dmalloc(c, &c->d_kr_ft, c->tab_cap); rc |= dmalloc(c, &c->d_kr_f, c->tab_cap);
  rc |= dmalloc(c, &c->sink, 128, true);
  rc |= dmalloc(c, &c->d_thin_ft, c->tab_cap); rc |= dmalloc(c, &c->d_thin_f, c->tab_cap); rc |= dmalloc(c, &c->d_thin_s, c->tab_cap);
  rc |= dmalloc(c, &c->Fbig, (size_t)c->ld * c->ld + 256 * (size_t)c->ld, true); rc |= dmalloc(c, &c->FTbig, (size_t)c->ld * c->ld + 256 * (size_t)c->ld, true);
  rc |= dmalloc(c, &c->Gbar, (size_t)T * p * p + 64); rc |= dmalloc(c, &c->Wtbar, (size_t)T * p * p + 64); rc |= dmalloc(c, &c->d_blk_lat, (size_t)p * c->Tp / 16 + 64); rc |= dmalloc(c, &c->d_blk_col, (size_t)p * c->Tp / 16 + 64);
  rc |= dmalloc(c, &c->vec, (size_t)q * (p + 1));
  rc |= dmalloc(c, &c->cdpart, (size_t)1024 * (p + 2) * q);
  rc |= dmalloc(c, &c->cdout, (size_t)(p + 2) * q + 8);
  rc |= dmalloc(c, &c->cdym, (size_t)(p + 1) * q + 8);
  rc |= dmalloc(c, &c->cdym_part, (size_t)1024 * (p + 1) * q);
  rc |= dmalloc(c, &c->last_trials, R);
  {
    const size_t NH = 1 + (size_t)(p + 1) + (size_t)(p + 1) * (p + 2) / 2;
    rc |= dmalloc(c, &c->cdhpart, (size_t)128 * NH * q);
    rc |= dmalloc(c, &c->cdhout, NH * q + 8);
    rc |= dmalloc(c, &c->cdcenter, (size_t)q * (p + 1));
    rc |= dmalloc(c, &c->cdpack, (size_t)q * (p + 3) + 8);
  }
  if (hipHostMalloc((void**)&c->h_pcg, 4 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->d_hpcg, c->h_pcg, 0) != hipSuccess) { (void)hipGetLastError(); c->h_pcg = nullptr; c->d_hpcg = nullptr; }
  rc |= alloc_cholws(c, &c->kws, p * TAU_MULTI_MAX, c->Tp, true);
  c->kws.nact = round_up(T, 64);
  {
    const size_t nqmax = (size_t)p * TAU_MULTI_MAX;
    rc |= dmalloc(c, &c->tK, slab * nqmax); rc |= dmalloc(c, &c->tM, slab * nqmax); rc |= dmalloc(c, &c->tA1, slab * nqmax);
    rc |= dmalloc(c, &c->tA2, slab * nqmax);
    rc |= dmalloc(c, &c->tscal, 16 + 8 * nqmax); rc |= dmalloc(c, &c->tpart, 1024 + 64 * nqmax);
  }
  if (rc) { pgpfa_destroy(c); return 1; }
  e = hipStreamSynchronize(c->st);
  if (e != hipSuccess) { pgpfa_destroy(c); return fail("sync: %s", hipGetErrorString(e)); }
  c->info["n_pad"] = c->npad;
  c->info["counts_two_bytes"] = 0.0;
  c->info["arena_bytes"] = 0.0;
  c->info["last_eps_wt_norm"] = 0.0;
  c->info["last_eps_wt_rms"] = 0.0;
  c->info["last_split_cov"] = 0.0;
  *out = c;
  return 0;
}

int pgpfa_destroy(pgpfa_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  if (c->st) hipStreamSynchronize(c->st);
  if (c->comm) ncclCommDestroy(c->comm);
  for (void* p : c->allocs) hipFree(p);
  if (c->vsmgp) hipFree(c->vsmgp);
  if (c->Flr32) hipFree(c->Flr32);
  if (c->lam_keep) hipFree(c->lam_keep);
  if (c->split_buf) hipFree(c->split_buf);
  if (c->Yhi) hipFree(c->Yhi);
  arena_release(c);
  if (c->hbuf) hipHostFree(c->hbuf);
  if (c->dl_stage) hipHostFree(c->dl_stage);
  if (c->hibuf) hipHostFree(c->hibuf);
  if (c->ring) hipHostFree(c->ring);
  if (c->h_pcg) hipHostFree(c->h_pcg);
  if (c->h_seq) hipHostFree(c->h_seq);
  for (auto e : c->prof.pool) hipEventDestroy(e);
  if (c->st2) { hipStreamSynchronize(c->st2); hipStreamDestroy(c->st2); }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->st) hipStreamDestroy(c->st);
  delete c;
  return 0;
}

int pgpfa_set_option(pgpfa_ctx* c, const char* key, double v) {
  if (!c || !key) return fail("null argument");
  const std::string k(key);
  if (k == "newton_xtol") c->xtol = v;
  else if (k == "newton_max_iter") c->max_iter = (int)v;
  else if (k == "use_mfma") c->mfma = (v != 0.0);
  else if (k == "cd_mfma") c->cd_mfma = (v != 0.0);
  else if (k == "cd_hess_mfma") c->cd_hess_mfma = (v != 0.0);
  else if (k == "cross_kernel") c->cross_kernel = (v != 0.0);
  else if (k == "cd_debug") c->cd_debug = (int)v;
  else if (k == "pcg_fused") c->pcg_fused = (int)v;
  else if (k == "pcg_w32") c->pcg_w32 = (v != 0.0);
  else if (k == "pcg_form") c->pcg_form = (int)v;
  else if (k == "pcg_adapt") c->pcg_adapt = (int)v;
  else if (k == "pcg_xcd") c->pcg_xcd = (int)v;
  else if (k == "mt_fill") c->mt_fill = (int)v;
  else if (k == "overlap_factors") c->overlap_factors = (int)v;
  else if (k == "mix_slot") c->mix_slot = (int)v;
  else if (k == "mix_wide") c->mix_wide = (int)v;
  else if (k == "thin_products") c->thin_products = (int)v;
  else if (k == "copy_kernels") c->copy_kernels = (v != 0.0);
  else if (k == "chord") c->chord = (v != 0.0);
  else if (k == "shared_pcg") c->shared_pcg = (v != 0.0);
  else if (k == "time_newton") c->time_newton = (v != 0.0);
  else if (k == "pcg_trace") c->pcg_trace = (v != 0.0);
  else if (k == "measure_mix") { c->measure_mix = (v != 0.0); c->info["last_eps_wt_norm"] = 0.0; c->info["last_eps_wt_rms"] = 0.0; }
  else if (k == "pcg_retire") c->pcg_retire = (v != 0.0);
  else if (k == "split_cov") c->split_cov = (v != 0.0);
  else if (k == "split_max_norm") c->split_max_norm = v;
  else if (k == "cov_mode") c->cov_mode = (int)v;
  else if (k == "lowrank_tol") c->lr_tol = v;
  else if (k == "keep_trial_vsmgp") c->keep_trial_vsmgp = (v != 0.0);
  else if (k == "dual_lowrank") c->dual_lowrank = (v != 0.0);
  else if (k == "dual_f32") c->dual_f32 = (int)v;
  else if (k == "slab_row_align") c->slab_row_align = (v != 0.0);
  else if (k == "vsm_mfma") c->vsm_mfma = (v != 0.0);
  else if (k == "dual_gemm") c->dual_gemm = (v != 0.0);
  else if (k == "extrapolate_start") c->extrapolate = (v != 0.0);
  else if (k == "extrapolate_beta") c->extrapolate_beta = v;
  else if (k == "shared_min") c->shared_min = (int)v;
  else if (k == "pcg_inner") c->pcg_inner_max = std::max(1, (int)v);
  else if (k == "pcg_eta0") c->pcg_eta0 = v;
  else if (k == "splitk_target") c->splitk_target = std::max(1, (int)v);
  else if (k == "small_tile_below") c->small_tile_below = (int)v;
  else if (k == "f32_tile64") c->f32_tile64 = (int)v;
  else if (k == "splitk_below64") c->splitk_below64 = (int)v;
  else if (k == "pcg_outer_max") c->pcg_outer_max = (int)v;
  else if (k == "chord_xtol") c->chord_xtol = v;
  else if (k == "chord_rho") c->chord_rho = v;
  else if (k == "chord_max_step") c->chord_max_step = v;
  else if (k == "chord_max") c->chord_max = (int)v;
  else if (k == "chunk_trials") { if (c->B > 0) return fail("chunk_trials must be set before the first E-step"); c->chunk_opt = (int)v; }
  else if (k == "workspace_headroom") c->arena_headroom = std::max(1.0, v);
  else if (k == "workspace_granule_mb") c->vmm_granule = (size_t)std::max(2.0, v) << 20;
  else if (k == "workspace_vmm") { if (c->arena_cap > 0) return fail("workspace_vmm must be set before the first E-step"); c->vmm = (v != 0.0) ? 0 : -1; }
  else if (k == "eps_noise") c->eps = v;
  else if (k == "profile") {
    // 0: off (the accumulated sums stay readable), 1: time every tagged launch, 2: GEMM launches only
    if (v != 0.0) {
      prof_collect(c);
      c->prof.ms.clear(); c->prof.flops.clear(); c->prof.count.clear(); c->prof.max_ms.clear(); c->prof.max_flops.clear();
      // the events the run will cycle through are created and recorded once NOW: the runtime sets up its signal pools
      // on first use (a one-off ~30 ms that would otherwise land somewhere inside the region being timed)
      HIPC(hipSetDevice(c->device));
      const size_t want = 2 * (256 + 64) + 2;
      while (c->prof.pool.size() < want) {
        hipEvent_t e;
        HIPC(hipEventCreate(&e));
        c->prof.pool.push_back(e);
        c->prof.idle.push_back(e);
      }
      for (int rep = 0; rep < 8; ++rep)            // (the one-off was seen after ~2000 recordings, not at creation)
        for (hipEvent_t e : c->prof.idle) HIPC(hipEventRecord(e, c->st));
      HIPC(hipStreamSynchronize(c->st));
    }
    c->prof.on = (v != 0.0);
    c->prof.configured = c->prof.on;
    c->prof.only_tag = (v == 2.0) ? TAG_GEMM : -1;
  } else if (k == "profile_pause") {
    // 1: stop recording events without touching the sums or waiting for anything; 0: go on (only while "profile" is set).  An event pair
    // around a launch costs ~10 us of device time (two barrier packets): a caller that wants rates over a long region samples it.
    c->prof.on = (v == 0.0) && c->prof.configured;
  } else return fail("unknown option '%s'", key);
  return 0;
}

int pgpfa_get_info(pgpfa_ctx* c, const char* key, double* value) {
  if (!c || !key || !value) return fail("null argument");
  const std::string k(key);
  static const char* tags[TAG_N] = {"gemm", "potrf", "solve", "poisson", "assemble", "vsm", "cd"};
  if (k.rfind("prof_", 0) == 0) {
    prof_collect(c);
    for (int t = 0; t < TAG_N; ++t) {
      const std::string base = std::string("prof_") + tags[t];
      if (k == base + "_ms") { *value = c->prof.ms[t]; return 0; }
      if (k == base + "_flops") { *value = c->prof.flops[t]; return 0; }
      if (k == base + "_launches") { *value = c->prof.count[t]; return 0; }
      if (k == base + "_max_ms") { *value = c->prof.max_ms[t]; return 0; }
      if (k == base + "_max_flops") { *value = c->prof.max_flops[t]; return 0; }
    }
    return fail("unknown info key '%s'", key);
  }
  if (k == "hbm_bytes_allocated") { *value = (double)c->bytes; return 0; }
  if (k == "n_trials_global") { *value = c->n_trials_global; return 0; }
  auto it = c->info.find(k);
  if (it == c->info.end()) return fail("unknown info key '%s'", key);
  *value = it->second;
  return 0;
}

// The resident counts of the listed trials (NULL: all) have been replaced: everything derived from them is stale - the hoisted count
// terms and Hessian sums of the (C,d) M-step, the accumulated covariance sum, and the posterior of those trials.
static void counts_changed(pgpfa_ctx* c, const std::vector<int>* trials) {
  c->cdym_valid = false;
  c->cd_hess_valid = false;
  c->pacc_valid = false;
  c->have_precomp = false;
  c->have_post = false;
  if (trials) { for (int t : *trials) { c->vsmgp_ok[t] = 0; c->mode_serial[t] = -10; c->trial_snap[t] = -1; c->trial_dual[t] = 0; c->lam_resident[t] = 0; } }
  else {
    std::fill(c->vsmgp_ok.begin(), c->vsmgp_ok.end(), 0); std::fill(c->mode_serial.begin(), c->mode_serial.end(), -10);
    std::fill(c->trial_snap.begin(), c->trial_snap.end(), -1); std::fill(c->trial_dual.begin(), c->trial_dual.end(), 0);
    std::fill(c->lam_resident.begin(), c->lam_resident.end(), 0);
  }
}

static void drop_high_plane(pgpfa_ctx* c) {
  if (!c->Yhi) return;
  hipStreamSynchronize(c->st);
  hipFree(c->Yhi);
  c->bytes -= (size_t)c->R * c->q * c->T;
  c->Yhi = nullptr;
}
static int ensure_high_plane(pgpfa_ctx* c) {
  if (c->Yhi) return 0;
  const size_t n = (size_t)c->R * c->q * c->T;
  if (hipMalloc((void**)&c->Yhi, n) != hipSuccess) { (void)hipGetLastError(); c->Yhi = nullptr; return fail("hipMalloc(%zu bytes) for the high bytes of the counts failed", n); }
  HIPC(hipMemsetAsync(c->Yhi, 0, n, c->st));
  c->bytes += n;
  return 0;
}

int pgpfa_upload_counts_u8(pgpfa_ctx* c, const uint8_t* Y) {
  if (!c || !Y) return fail("null argument");
  HIPC(hipSetDevice(c->device));
  drop_high_plane(c);
  HIPC(hipMemcpyAsync(c->Y, Y, (size_t)c->R * c->q * c->T, hipMemcpyHostToDevice, c->st));
  HIPC(hipStreamSynchronize(c->st));
  c->have_counts = true;
  counts_changed(c, nullptr);
  c->info["counts_two_bytes"] = 0.0;
  return 0;
}

// counts from a host array of TS (double or uint16): validated and split into byte planes on the device, staged in pieces
extern "C++" {
template <typename TS>
static int upload_counts_wide(pgpfa_ctx* c, const TS* Y) {
  HIPC(hipSetDevice(c->device));
  const size_t n = (size_t)c->R * c->q * c->T;
  const size_t piece = std::min<size_t>(n, (size_t)1 << 26);
  TS* tmp = nullptr;
  int* flags = nullptr;
  HIPC(hipMalloc((void**)&tmp, piece * sizeof(TS)));
  hipError_t e = hipMalloc((void**)&flags, 2 * sizeof(int));
  if (e != hipSuccess) { hipFree(tmp); return fail("hipMalloc: %s", hipGetErrorString(e)); }
  drop_high_plane(c);
  int hf[2] = {0, 0};
  int rc = 0;
  for (int pass = 0; pass < 2 && !rc; ++pass) {
    // pass 0 writes the low bytes and finds out whether any count needs a second byte; only then is the plane of high bytes
    // allocated and the split repeated (pass 1)
    hipMemsetAsync(flags, 0, 2 * sizeof(int), c->st);
    for (size_t off = 0; off < n; off += piece) {
      const size_t m = std::min(piece, n - off);
      hipMemcpyAsync(tmp, Y + off, m * sizeof(TS), hipMemcpyHostToDevice, c->st);
      hipLaunchKernelGGL(pack_counts_kernel<TS>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->st, tmp, c->Y + off, c->Yhi ? c->Yhi + off : nullptr, m, flags);
      hipStreamSynchronize(c->st);
    }
    hipMemcpy(hf, flags, 2 * sizeof(int), hipMemcpyDeviceToHost);
    if (hf[0] || !hf[1] || pass == 1) break;
    rc = ensure_high_plane(c);
  }
  hipFree(tmp);
  hipFree(flags);
  // from the first piece on, c->Y holds a mixture of old and new (or clamped) counts and the old high plane is gone: on ANY failure the
  // context has no counts and nothing derived from the old ones survives (a caller that catches the error must upload again)
  const hipError_t e_last = hipGetLastError();
  if (rc || e_last != hipSuccess || hf[0]) {
    c->have_counts = false;
    counts_changed(c, nullptr);
    c->info["counts_two_bytes"] = 0.0;
    if (rc) return rc;
    if (e_last != hipSuccess) return fail("count upload: %s", hipGetErrorString(e_last));
    return fail("spike counts must be integers in [0, 65535]");
  }
  c->have_counts = true;
  counts_changed(c, nullptr);
  c->info["counts_two_bytes"] = c->Yhi ? 1.0 : 0.0;
  return 0;
}
}  // extern "C++"

int pgpfa_upload_counts_f64(pgpfa_ctx* c, const double* Y) {
  if (!c || !Y) return fail("null argument");
  return upload_counts_wide<double>(c, Y);
}

int pgpfa_upload_counts_u16(pgpfa_ctx* c, const uint16_t* Y) {
  if (!c || !Y) return fail("null argument");
  return upload_counts_wide<uint16_t>(c, Y);
}

// resident counts of the listed trials as uint16 [n][q][T]
int pgpfa_get_counts_u16(pgpfa_ctx* c, int n, const int32_t* idx, uint16_t* out) {
  if (!c || !out) return fail("null argument");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  HIPC(hipSetDevice(c->device));
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr));
  const size_t m = (size_t)c->q * c->T;
  std::vector<uint8_t> lo(m), hi(m, 0);
  for (size_t i = 0; i < tr.v.size(); ++i) {
    HIPC(hipMemcpyAsync(lo.data(), c->Y + (size_t)tr.v[i] * m, m, hipMemcpyDeviceToHost, c->st));
    if (c->Yhi) CHK(dl_enqueue(c, hi.data(), c->Yhi + (size_t)tr.v[i] * m, m));
    CHK(dl_flush(c));
    for (size_t e = 0; e < m; ++e) out[i * m + e] = (uint16_t)(lo[e] | (hi[e] << 8));
  }
  return 0;
}

int pgpfa_set_params(pgpfa_ctx* c, const double* C, const double* d, const double* tau_s) {
  if (!c || !C || !d || !tau_s) return fail("null argument");
  HIPC(hipSetDevice(c->device));
  for (int k = 0; k < c->p; ++k)
    if (!(tau_s[k] > 0.0) || !std::isfinite(tau_s[k])) return fail("tau[%d] = %g must be positive and finite", k, tau_s[k]);
  CHK(upload(c, c->C, C, (size_t)c->q * c->p));
  CHK(upload(c, c->d, d, c->q));
  CHK(upload(c, c->tau, tau_s, c->p));
  {
    // (the arguments may alias the stored copies: pgpfa_set_params(c, c->eC.data(), ...) restores a snapshot)
    std::vector<double> nC(C, C + (size_t)c->q * c->p), nd(d, d + c->q), nt(tau_s, tau_s + c->p);
    c->hC.swap(nC); c->hd.swap(nd); c->htau.swap(nt);
  }
  hipLaunchKernelGGL(gram_tau_kernel, dim3(c->Tp, c->p), dim3(256), 0, c->st, c->Kpad, c->Tp, c->T, c->tau, c->bin, c->eps);
  if (c->CCu)
    hipLaunchKernelGGL(poisson_tables_kernel, dim3(c->qpad), dim3(64), 0, c->st, c->C, c->q, c->p, c->qpad, c->ccu_cols, c->CCu, c->C16);
  hipLaunchKernelGGL(dual_table_kernel, dim3(c->qpad), dim3(64), 0, c->st, c->C, c->q, c->p, c->dual_ncol, c->dual_npd, c->dual_tbl);
  HIPC(hipGetLastError());
  // The two things built from the timescales do not depend on each other: the Gram inverses (p slots through the batched factor kernels: ~40 small
  // launches, two host read-backs) and the pivoted Cholesky factors (one kernel of p workgroups).  The latter goes to the side stream first.
  bool side = false;
  if (c->overlap_factors && c->st2) {
    if (hipEventRecord(c->ev_fork, c->st) == hipSuccess && hipStreamWaitEvent(c->st2, c->ev_fork, 0) == hipSuccess) {
      CHK(launch_pivchol(c, c->st2));
      HIPC(hipEventRecord(c->ev_join, c->st2));
      side = true;
    } else {
      (void)hipGetLastError();
    }
  }
  const int rc_kinv = build_kinv(c);
  if (rc_kinv) {                                            // (never leave the side stream running into buffers a failed call may free)
    if (side) (void)hipStreamSynchronize(c->st2);
    return rc_kinv;
  }
  CHK(build_lowrank(c, side));
  c->have_params = true;
  return 0;
}

static int get_slabs(pgpfa_ctx* c, const double* src, double* out) {
  for (int k = 0; k < c->p; ++k) {
    HIPC(hipMemcpy2DAsync(out + (size_t)k * c->T * c->T, (size_t)c->T * sizeof(double), src + (size_t)k * c->Tp * c->Tp,
                          (size_t)c->Tp * sizeof(double), (size_t)c->T * sizeof(double), c->T, hipMemcpyDeviceToHost, c->st));
  }
  HIPC(hipStreamSynchronize(c->st));
  return 0;
}
int pgpfa_get_gram(pgpfa_ctx* c, double* K) {
  if (!c || !K) return fail("null argument");
  if (!c->have_params) return fail("set_params has not been called");
  HIPC(hipSetDevice(c->device));
  return get_slabs(c, c->Kpad, K);
}
int pgpfa_get_gram_inverse(pgpfa_ctx* c, double* Kinv) {
  if (!c || !Kinv) return fail("null argument");
  if (!c->have_params) return fail("set_params has not been called");
  HIPC(hipSetDevice(c->device));
  return get_slabs(c, c->Kinv, Kinv);
}

static int ready(pgpfa_ctx* c) {
  if (!c) return fail("null context");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  if (!c->have_params) return fail("set_params has not been called");
  HIPC(hipSetDevice(c->device));
  return ensure_workspace(c, false);
}

static int ready_estep(pgpfa_ctx* c, bool allow_lowrank) {
  if (!c) return fail("null context");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  if (!c->have_params) return fail("set_params has not been called");
  HIPC(hipSetDevice(c->device));
  return ensure_workspace(c, allow_lowrank && want_lowrank(c));
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

int pgpfa_set_modes(pgpfa_ctx* c, int n, const int32_t* idx, const double* X) {
  if (!c || !X) return fail("null argument");
  HIPC(hipSetDevice(c->device));
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  for (size_t i = 0; i < tr.v.size(); ++i)
    HIPC(hipMemcpyAsync(c->Xmode + (size_t)tr.v[i] * c->n, X + i * c->n, c->n * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPC(hipStreamSynchronize(c->st));
  c->pacc_valid = false;          // the accumulated covariance sum belonged to the modes just overwritten
  c->cdym_valid = false;          // ... and so do the hoisted count terms sum_t y m_t and the per-neuron Hessian sums of the (C,d) M-step
  c->cd_hess_valid = false;
  for (int t_ : tr.v) c->mode_serial[t_] = -10;
  return 0;
}

// The posterior of the listed trials has just been computed under the current parameters: snapshot them once, point the trials at
// the snapshot, drop snapshots no trial refers to any more.
static void snapshot_params(pgpfa_ctx* c, const std::vector<int>& trials) {
  const int id = ++c->snap_serial;
  c->snaps[id] = pgpfa_ctx::ParamSnap{c->hC, c->hd, c->htau};
  for (int t : trials) c->trial_snap[t] = id;
  std::vector<char> used(c->snaps.size() + 1, 0);
  std::map<int, int> pos;
  int i = 0;
  for (auto& kv : c->snaps) pos[kv.first] = i++;
  for (int s : c->trial_snap) if (s >= 0) used[pos[s]] = 1;
  for (auto it = c->snaps.begin(); it != c->snaps.end();) {
    if (!used[pos[it->first]]) it = c->snaps.erase(it); else ++it;
  }
}

// Run fn under the parameters of snapshot `id` (no-op switch when they are the current ones), then put the current ones back.
static int with_snapshot(pgpfa_ctx* c, int id, const std::function<int()>& fn) {
  auto it = c->snaps.find(id);
  if (it == c->snaps.end()) return fn();
  const pgpfa_ctx::ParamSnap snap = it->second;              // (copy: pgpfa_set_params rewrites hC..., never the snapshots, but keep it simple)
  const bool moved = (c->hC != snap.C || c->hd != snap.d || c->htau != snap.tau);
  if (!moved) return fn();
  const std::vector<double> curC = c->hC, curd = c->hd, curtau = c->htau;
  CHK(pgpfa_set_params(c, snap.C.data(), snap.d.data(), snap.tau.data()));
  int rc = fn();
  const std::string err = g_err;
  const int rc2 = pgpfa_set_params(c, curC.data(), curd.data(), curtau.data());
  if (rc) { g_err = err; return rc; }
  return rc2;
}

static int remember_trials(pgpfa_ctx* c, const std::vector<int>& v) {
  c->last_trials_h = v;
  CHK(upload_list(c, c->last_trials, v));
  c->have_post = true;
  c->have_precomp = false;
  c->pacc_valid = false;
  c->cdym_valid = false;
  return 0;
}


// Z <- P^-1 R for the nb slot vectors at once.  Dense plan: the shared preconditioner is ONE matrix, its explicit
// inverse is formed once per chunk and applied with a single multi-RHS GEMM.  Low-rank plan: the same matrix in
// Woodbury form, P^-1 v = Gb (eps v + F Sb F^T Gb v), with Gb the per-bin blocks of the mean curvature and Sb the
// inverse of the r x r system: two per-bin kernels and three thin GEMMs, no n x n matrix anywhere.
// (skip: device stop flag of the inner PCG loop; final_apply = false leaves the last per-bin application to the caller, with
// y = F Sb F^T Gb R in c->Xt; first_apply = false: the caller has already put Gb R into c->Xt)
static int shared_solve(pgpfa_ctx* c, int nb, const double* R, double* Z, const int* skip = nullptr, bool first_apply = true,
                        bool final_apply = true, const int* cols = nullptr, int ncols = 0) {
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
    ThinP tp{};
    tp.F = c->Flr; tp.Tf = c->Tp; tp.T = c->T; tp.FT = c->FTbig; tp.ldft = c->rpad;
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
      if (c->T % 4 == 0) hipLaunchKernelGGL(thin_ft_kernel<true>, dim3(c->nthin_ft, (ng + 15) / 16), dim3(512), 0, c->st, tp);
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
      if (c->T % 4 == 0) hipLaunchKernelGGL(thin_f_kernel<true>, dim3(c->nthin_f, (ng + 15) / 16), dim3(256), 0, c->st, tp);
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
static int bin_blocks(pgpfa_ctx* c, const double* W, long long sW, double* G, double* Wt, long long sO, int nslots, double* ldet) {
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
  const int nblk64 = rpad / 64, npairs = nblk64 * (nblk64 + 1) / 2;
  hipLaunchKernelGGL(assemble_b_kernel_t<double>, dim3(npairs, 1), dim3(256), 0, c->st, lw.H, (long long)lw.sH, rpad, nblk64, (const double*)c->Flr, c->Tp, T, p, c->d_blk_lat,
                     c->d_blk_col, c->Wtbar, 0LL, c->ident, 1);
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
static int ensure_mt_clean(pgpfa_ctx* c) {
  if (!c->mt_dirty) return 0;
  HIPC(hipMemsetAsync(c->ws.Mt, 0, (size_t)c->ws.sM * c->B * sizeof(double), c->st));
  c->mt_dirty = false;
  return 0;
}

// Per-trial T x T blocks live in an allocation of their own, freed only with the context: allocated lazily (often in the
// middle of an E-step, i.e. after the workspace mark), it must not be swept up by free_workspace on a re-plan.
static int ensure_vsmgp_buffer(pgpfa_ctx* c) {
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
static int accumulate_split(pgpfa_ctx* c, const CholWS& lw, int nb, int ract, int Ts, bool skip_zero_cols, int ctile) {
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
  prof_begin(c, TAG_VSM, (double)nb * c->n * rpad * (3.0 * p + 1.0));
  if (p > 16)                                               // (17..20 latents: split_candidate admits no others beyond 16)
    hipLaunchKernelGGL((mix_vsm_wide2_kernel<20, true>), dim3((T + 63) / 64, nb), dim3(256), 0, c->st, lw.H, (long long)lw.sH, c->ld, c->Gbin, sW, T, p, ract, c->eps,
                       c->vsm, c->ident, c->trial_of_slot, Ts, c->sink, D, sD, ldd);
  else
  dispatch_pw(p, [&](auto pw) {
    constexpr int PW = decltype(pw)::value;
    if constexpr (PW <= 10) {
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
  // 2. the full-width term on the FP16 matrix cores: partial sums per (latent, group of slots) into c->ppart
  const int sps = std::max(1, (nb + PACC_SPLITS - 1) / PACC_SPLITS);
  const int ngroups = (nb + sps - 1) / sps;
  {
    SyrkF16Args a{};
    a.D = D; a.sD = sD; a.ldd = ldd; a.ts = Ts; a.part = c->ppart;
    a.T = T; a.p = p; a.ract = ract; a.nslots = nb; a.sps = sps; a.ngroups = ngroups;
    a.tiles = (T + 127) / 128; a.ntiles = a.tiles * (a.tiles + 1) / 2;
    const long long blocks = (long long)a.ntiles * ngroups * p;
    prof_begin(c, TAG_VSM, 3.0 * (double)nb * ract * T * T * p);
    hipLaunchKernelGGL(syrk_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, c->st, a);
    prof_end(c);
  }
  // 3. per latent: S_k = sum_r A A^T (r_k x r_k), X_k = sum_r A D_k^T (r_k x T) with A = rows of latent k of L^-T right of column
  //    roff_k (it is upper triangular), both as segmented-K products over groups of slots; then T1 = F S F^T and Xfull = F X
  const int spsS = std::max(1, (nb + NS - 1) / NS);
  for (int k = 0; k < p; ++k) {
    const int rk = c->rk[k], r0 = c->roff[k];
    const int kw = ract - r0;                                                   // columns of A that are not identically zero
    if (kw <= 0 || rk <= 0) {
      HIPC(hipMemsetAsync(T1 + (size_t)k * tt, 0, tt * sizeof(double), c->st));
      HIPC(hipMemsetAsync(Xfull + (size_t)k * tt, 0, tt * sizeof(double), c->st));
      continue;
    }
    const double* A0 = lw.Mt + r0 + (size_t)r0 * rpad;
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
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)(((size_t)rk * rk + 255) / 256)), dim3(256), 0, c->st, Spart, ngS, rk, rk, 32, Ssum);
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)(((size_t)rk * T + 255) / 256)), dim3(256), 0, c->st, Xpart, ngroups, rk, T, 0, Xsum);
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
static int posterior_blocks_lowrank(pgpfa_ctx* c, int nb, bool want_vsmgp, bool accumulate, double* logdet_out = nullptr) {
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
  const int nblk64 = rpad / 64, npairs = nblk64 * (nblk64 + 1) / 2;
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
                       (long long)lwf.sH, rpad, nblk64, (const float*)c->Flr32, Tp, T, p, c->d_blk_lat, c->d_blk_col, c->Wt, sW, c->ident, nb);
  } else {
    hipLaunchKernelGGL(assemble_b_kernel_t<double>, dim3(npairs, (nb + AB_SLOTS - 1) / AB_SLOTS), dim3(256), 0, c->st, lw.H, (long long)lw.sH, rpad,
                       nblk64, (const double*)c->Flr, Tp, T, p, c->d_blk_lat, c->d_blk_col, c->Wt, sW, c->ident, nb);
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
  // c. Yt = F Mts  (n x rpad, ld = c->ld) into the factor slab (the factor itself is dead now)
  // (column tiles left of roff[k] skipped under skip_zero_cols, see above)
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
  if (p > WIDE_MAX) return fail("low-rank covariance engine supports up to %d latents (p=%d)", WIDE_MAX, p);
  bool split = false;
  if (split_candidate || c->measure_mix) {
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
  }
  c->info["last_split_cov"] = split ? 1.0 : 0.0;
  if (want_vsmgp && split) {
    CHK(accumulate_split(c, lw, nb, ract, Ts, skip_zero_cols, ctile));
  } else if (want_vsmgp) {
    // d+e. one pass over Yt: post_vsm[t] = eps G_t + sum_b (G_t y_b)(G_t y_b)^T, and Yt is mixed in place (y <- G_t y) so that
    //      rows (k,.) of the slab become Ymix_k, the GEMM operand of post_vsmGP_k = eps diag(G_t[k][k]) + Ymix_k Ymix_k^T
    prof_begin(c, TAG_VSM, (double)nb * c->n * rpad * (3.0 * p + 1.0));
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

static int posterior_blocks(pgpfa_ctx* c, int nb, double diag_scale, bool want_vsmgp, bool accumulate = false) {
  if (c->plan_lowrank) {
    if (diag_scale != 1.0) return fail("internal: jittered covariance requested under the low-rank workspace plan");
    return posterior_blocks_lowrank(c, nb, want_vsmgp, accumulate);
  }
  return posterior_blocks_dense(c, nb, diag_scale, want_vsmgp);
}

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
static int ensure_lambda(pgpfa_ctx* c);
static int dual_common(pgpfa_ctx* c, int nb, std::vector<double>* sB, std::vector<double>* sD, std::vector<double>* vKv);
static int dual_jitter(pgpfa_ctx* c, int nb);
static int dual_eval_slots(pgpfa_ctx* c, int nb, const std::vector<int>& tos, bool want_grad, double* cost, bool tolerate);
static int var_offsets(pgpfa_ctx* c, int nb, double* out);
static int check_distinct(const std::vector<int>& v);

static int estep_impl(pgpfa_ctx* c, const Trials& tr, int warm_start, bool allow_lr, double* obj_sum, int32_t* iters, int32_t* status,
                      const LooJob* loo = nullptr, const VarJob* var = nullptr) {
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
  std::vector<std::pair<hipEvent_t, hipEvent_t>> newton_ev;   // events around every inner solve (the Newton-solve kernels)
  int max_it_seen = 0;
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
      std::vector<int> how(nb, 0);
      bool any = false;
      for (int s = 0; s < nb && warm_start; ++s) {
        const int tr_ = tos[s];
        if (warm_start == 2 && c->mode_serial[tr_] < 0) continue;
        how[s] = 1;
        if (c->extrapolate && c->mode_serial[tr_] == c->estep_serial - 1 && c->prev_serial[tr_] == c->estep_serial - 2) how[s] = 2;
        any = true;
      }
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
    CHK(prior_mv_all(c, nb, c->Xc, c->KX));
    hipLaunchKernelGGL(dots3_kernel, dim3(nb), dim3(256), 0, c->st, c->Xc, ld, c->KX, ld, (const double*)nullptr, 0LL,
                       (const double*)nullptr, 0LL, nvec, c->ident, c->sc_qxx, c->sc_qdx, c->sc_qdd);
    CHK(poisson(c, c->ident, nb, c->Xc, c->Gl, c->W, c->sc_f, 1));
    CHK(dl_enqueue(c, f.data(), c->sc_f, nb * sizeof(double)));
    CHK(download(c, qxx.data(), c->sc_qxx, nb));
    active.clear();
    for (int s = 0; s < nb; ++s) {
      f[s] += 0.5 * qxx[s];
      if (vo == 0) its[s] = 0;
      if (var && vstat[s] != 1) continue;         // (this slot's fixed point is settled)
      active.push_back(s);
      stat[s] = 1;
    }
    std::vector<int> leftovers;

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
                             c->W, c->Wt, (long long)nw, c->sc_alpha, nvec, nw, c->list_b);
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
        CHK(upload_nosync(c, c->list_a, active.data(), sizeof(int) * active.size()));
        if (c->time_newton) {
          newton_ev.emplace_back(prof_event(c->prof), prof_event(c->prof));
          hipEventRecord(newton_ev.back().first, c->st);
        }
        hipLaunchKernelGGL(grad_total_kernel, dim3((nvec + 255) / 256, na), dim3(256), 0, c->st, c->Gl, ld, c->KX, ld, c->Gt, ld, nvec, c->list_a);
        hipLaunchKernelGGL(pcg_init_kernel, dim3((c->npad + 255) / 256, na), dim3(256), 0, c->st, c->Gt, c->Rv, c->Dl, ld, nvec, c->npad, c->list_a);
        // (small chunks are launch-latency bound: there the extra packing / check launches of the host-free form cost more than
        // the round trips they remove - measured at config 2: 8.6 vs 8.0 ms per E-step)
        const bool fused = c->pcg_fused && c->plan_lowrank && p <= 16 && c->h_pcg != nullptr && (c->pcg_fused == 2 || (double)nb * c->n >= 1.0e6);
        int done_inner = 0;
        PcgCtl& fused_ctl = c->fused_ctl_host;             // (context member: a queued read-back must not point into this frame)
        fused_ctl = PcgCtl{};
        // form of the host-free iteration (pcg.h): the two-kernel step without the prior mat-vec needs the packed FP32 curvature and per-slot
        // retirement; otherwise the split kernels of round 3
        const bool onek = fused && c->pcg_w32 && c->pcg_retire && c->pcg_form != 0 && p <= 10;
        if (onek) {
          // ---- inner solve: per step pcg_cg_a_kernel, pcg_cg_b_kernel, the closing kernel and the three preconditioner products (pcg.h)
          const int* skip = &c->pcgctl->stop;
          const int npk = p * (p + 1) / 2;
          const int Tw = round_up(T, 32);                        // row stride of the component-major packed curvature: rows start on 128-byte lines
          const long long sW32 = (long long)Tw * npk;
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
            CHK(upload_nosync(c, c->pcg_eta, eta_s.data(), sizeof(float) * nb));
            CHK(copy_dev(c, c->live, c->list_a, sizeof(int) * na));
            PcgCtl h0{};
            h0.nlive = na; h0.nl[0] = na;
            CHK(upload_nosync(c, c->pcgctl, &h0, sizeof(PcgCtl)));
          }
          c->h_pcg[0] = 0; c->h_pcg[1] = 0; c->h_pcg[2] = na;
          hipLaunchKernelGGL(pack_w32t_kernel, dim3((T + 63) / 64, na), dim3(256), (size_t)npk * 65 * sizeof(float), c->st, c->W, (long long)T * p * p,
                             c->W32, sW32, Tw, T, p, c->list_a);
          // t = Gb r0, then y = F Sb F^T t over the listed columns (left in c->Xt)
          CHK(shared_solve(c, nb, c->Rv, c->Zv, nullptr, true, false, c->list_a, na));
          struct NdevGuard { pgpfa_ctx* c; ~NdevGuard() { c->cur_ndev = nullptr; } } ndev_guard{c};
          c->live_gemms.clear();
          PcgCgP cp{};
          cp.GbT = c->GbT; cp.WbT = reinterpret_cast<const float*>(c->WbT); cp.W32T = c->W32; cp.sW32 = sW32; cp.Tw = Tw;
          cp.X = c->Dl; cp.R = c->Rv; cp.P = c->Pv; cp.Q = c->Qv; cp.Z = c->Zv; cp.S = c->Sv; cp.Y = c->Xt; cp.sV = ld;
          cp.part = c->sc_part2; cp.gam = c->cg_scal; cp.alp = c->cg_scal + 2 * (size_t)c->B; cp.rr = c->sc_rr; cp.rr0 = c->sc_rr0; cp.eta = c->pcg_eta;
          cp.ctl = c->pcgctl; cp.live0 = c->live; cp.live1 = c->live + c->B;
          cp.eps = c->eps; cp.T = T; cp.p = p; cp.inner_min = c->pcg_inner_min; cp.ntile = (T + 63) / 64; cp.B = c->B; cp.xcd_map = c->pcg_xcd;
          for (int it = 0; it < c->pcg_inner_max; ++it) {
            cp.par = it & 1; cp.first = (it == 0) ? 1 : 0;
            // The launches of a step are sized by the live count the closing kernel last mirrored to the host (it only falls during a solve, so a
            // value that is a step or two old is an upper bound; the kernels read the true count on the device).  Few live slots: fewer slots per
            // workgroup, so that the per-bin kernels still offer every CU a workgroup and a wave walks one slot instead of four in a row.
            const int seen = *(volatile int*)&c->h_pcg[2];
            const int bound = c->pcg_adapt ? std::max(1, std::min(na, seen)) : na;
            cp.spw = !c->pcg_adapt ? PCG_SLOTS : bound > 640 ? 16 : bound > 320 ? 8 : 4;
            const dim3 gcg((T + 63) / 64, round_up((bound + cp.spw - 1) / cp.spw, 8));       // (slot groups in blocks of 8: pcg_cg_wg)
            dispatch_pw(p, [&](auto pw) {
              constexpr int PW = decltype(pw)::value;
              if constexpr (PW <= 10) {
                const size_t la = pcg_cg_a_lds(PW), lb = pcg_cg_b_lds(PW);
                // (per launch, not once per process: contexts of one process may sit on different devices)
                if (la > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_cg_a_kernel<PW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)la);
                if (lb > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_cg_b_kernel<PW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL(pcg_cg_a_kernel<PW>, gcg, dim3(256), la, c->st, cp);
                hipLaunchKernelGGL(pcg_cg_b_kernel<PW>, gcg, dim3(256), lb, c->st, cp);
              }
            });
            hipLaunchKernelGGL(pcg_iter_close_kernel, dim3(1), dim3(64), 0, c->st, c->pcgctl, it & 1, (volatile int*)c->d_hpcg);
            // the preconditioner products for the NEXT iteration run over the list this launch has just written
            c->live_gemm_collect = (it == 0);
            c->cur_ndev = &c->pcgctl->nl[(it & 1) ^ 1];
            CHK(shared_solve(c, nb, c->Rv, c->Zv, skip, false, false, (it & 1) ? c->live : c->live + c->B, bound));
            if (*(volatile int*)&c->h_pcg[0]) break;           // the device has already stopped: whatever is enqueued is a no-op
            if (it + 1 < c->pcg_inner_max) {
              const auto t_spin = std::chrono::steady_clock::now();
              while (!*(volatile int*)&c->h_pcg[0] && (it + 1) - *(volatile int*)&c->h_pcg[1] > 2) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 5.0) break;   // (never hang on a lost flag)
              }
              if (*(volatile int*)&c->h_pcg[0]) break;
            }
          }
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
        if (c->time_newton) hipEventRecord(newton_ev.back().second, c->st);
        hipLaunchKernelGGL(step_stats_kernel, dim3(na), dim3(256), 0, c->st, c->Gt, c->Dl, ld, nvec, c->list_a, c->sc_dec, c->sc_smax);
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
          for (int s : active) if (!in_cand[s] || bad[s]) fallback.push_back(s);
        }
        active.swap(next);
        leftovers.insert(leftovers.end(), fallback.begin(), fallback.end());
        max_it_seen = std::max(max_it_seen, outer + 1);
      }
      // anything still active after the outer cap also goes to the fallback
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
    for (int s = 0; s < nb; ++s)
      if (vstat[s] == 1 && stat[s] != 0) return fail("variational fixed point: the mode search of trial %d failed (status %d)", tos[s], stat[s]);
    c->lam_out_active = true;
    CHK(poisson(c, c->ident, nb, c->Xc, c->Glt, c->Wt, c->sc_f, 0));        // lambda = exp(C m + d + offset) -> c->lamd
    c->lam_out_active = false;
    if (c->plan_lowrank) { CHK(dual_jitter(c, nb)); CHK(posterior_blocks(c, nb, 1.0, false, false)); }   // (c->W: curvature at the modes = C^T diag(lambda) C)
    else CHK(posterior_blocks(c, nb, 1.0 + 1e-6, false));
    CHK(dl_enqueue(c, info.data(), c->ws.info, sizeof(int) * nb));
    CHK(dl_flush(c));
    for (int s = 0; s < nb; ++s)
      if (info[s] != 0) return fail("variational fixed point: posterior precision of trial %d not positive definite", tos[s]);
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
      if (vstat[s] != 1) continue;
      vouter[s] = vo + 1;
      if (!std::isfinite(vdelta[s])) return fail("variational fixed point: non-finite offsets for trial %d", tos[s]);
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
        c->lam_resident[tos[s]] = 1;
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
  CHK(estep_impl(c, tr, warm_start, true, &obj, it1.data(), st1.data()));
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
    CHK(estep_impl(c, retry, 1, false, &obj_redo, it2.data(), st2.data()));
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
int pgpfa_count_moments(pgpfa_ctx* c, int n, const int32_t* idx, int64_t* sum, int64_t* cross, int64_t* n_samples) {
  if (!c || !sum || !cross || !n_samples) return fail("null argument");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  HIPC(hipSetDevice(c->device));
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr));
  const int q = c->q, T = c->T, N = (int)tr.v.size();
  unsigned long long* dev = nullptr;
  int* dtr = nullptr;
  const size_t len = (size_t)q * q + q;
  HIPC(hipMalloc((void**)&dev, len * sizeof(unsigned long long)));
  HIPC(hipMalloc((void**)&dtr, (size_t)std::max(N, 1) * sizeof(int)));
  HIPC(hipMemsetAsync(dev, 0, len * sizeof(unsigned long long), c->st));
  HIPC(hipMemcpyAsync(dtr, tr.v.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->st));
  const int nt = (q + CM_TILE - 1) / CM_TILE, npairs = nt * (nt + 1) / 2;
  if (N > 0) {
    unsigned long long* dsum = dev + (size_t)q * q;
    hipLaunchKernelGGL(count_moments_kernel, dim3(npairs, N), dim3(256), 0, c->st, c->Y, c->Y, dtr, q, T, dsum, dev, 1ull, 1ull);
    if (c->Yhi) {            // y = lo + 256 hi: the mixed and high-high byte products (exact: integer arithmetic)
      hipLaunchKernelGGL(count_moments_kernel, dim3(npairs, N), dim3(256), 0, c->st, c->Y, c->Yhi, dtr, q, T, dsum, dev, 256ull, 0ull);
      hipLaunchKernelGGL(count_moments_kernel, dim3(npairs, N), dim3(256), 0, c->st, c->Yhi, c->Y, dtr, q, T, dsum, dev, 256ull, 0ull);
      hipLaunchKernelGGL(count_moments_kernel, dim3(npairs, N), dim3(256), 0, c->st, c->Yhi, c->Yhi, dtr, q, T, dsum, dev, 65536ull, 256ull);
    }
  }
  std::vector<unsigned long long> hostv(len);
  CHK(dl_enqueue(c, hostv.data(), dev, len * sizeof(unsigned long long)));
  CHK(dl_flush(c));
  hipFree(dev); hipFree(dtr);
  HIPC(hipGetLastError());
  for (int i = 0; i < q; ++i) {
    sum[i] = (int64_t)hostv[(size_t)q * q + i];
    for (int j = 0; j <= i; ++j) {            // tiles with ti > tj hold only the lower part; diagonal tiles both
      const int64_t v = (int64_t)hostv[(size_t)i * q + j];
      cross[(size_t)i * q + j] = v;
      cross[(size_t)j * q + i] = v;
    }
  }
  *n_samples = (int64_t)N * T;
  return 0;
}

// util.dataset (util.py:705-750) on the device: latent trajectories and counts of the listed trials drawn under the parameters
// of the context (pgpfa_set_params), counts written into the resident tensor (and copied out on request).
int pgpfa_generate(pgpfa_ctx* c, unsigned long long seed, int n, const int32_t* idx, double* X_out, uint8_t* Y_out) {
  if (!c) return fail("null context");
  if (!c->have_params) return fail("set_params has not been called");
  if (c->T > 65536 || c->q > 65535) return fail("generator supports up to 65535 neurons and 65536 bins");
  HIPC(hipSetDevice(c->device));
  Trials tr;
  CHK(resolve_trials(c, n, idx, &tr, true));
  const int N = (int)tr.v.size(), q = c->q, p = c->p, T = c->T;
  double* X = nullptr;
  int* dtr = nullptr;
  int* flag = nullptr;
  HIPC(hipMalloc((void**)&X, (size_t)c->R * p * T * sizeof(double)));
  hipError_t e1 = hipMalloc((void**)&dtr, (size_t)N * sizeof(int)), e2 = hipMalloc((void**)&flag, 2 * sizeof(int));
  int rc = 0, over[2] = {0, 0};
  if (e1 != hipSuccess || e2 != hipSuccess) rc = fail("hipMalloc failed");
  if (!rc) {
    hipMemcpyAsync(dtr, tr.v.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->st);
    int rmax = 0;
    for (int k = 0; k < p; ++k) rmax = std::max(rmax, c->rk[k]);
    hipLaunchKernelGGL(sample_latents_kernel, dim3(p, N), dim3(256), (size_t)(rmax + 2) * sizeof(double), c->st, c->Flr, c->Tp, T, p, c->d_rank, c->eps,
                       seed, dtr, X);
    for (int pass = 0; pass < 2 && !rc; ++pass) {
      // a count above 255 needs the plane of high bytes: allocate it and draw again (a draw is a pure function of its counters)
      hipMemsetAsync(flag, 0, 2 * sizeof(int), c->st);
      hipLaunchKernelGGL(sample_counts_kernel, dim3((T + 63) / 64, q, N), dim3(64), 0, c->st, X, c->C, c->d, q, p, T, seed, dtr, c->Y, c->Yhi, flag);
      hipMemcpyAsync(over, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, c->st);
      if (hipStreamSynchronize(c->st) != hipSuccess || hipGetLastError() != hipSuccess) { rc = fail("generator launch failed"); break; }
      if (!over[0] || over[1]) break;
      rc = ensure_high_plane(c);
    }
    if (!rc && !over[1]) {
      for (int i = 0; i < N; ++i) {
        if (X_out) hipMemcpyAsync(X_out + (size_t)i * p * T, X + (size_t)tr.v[i] * p * T, (size_t)p * T * sizeof(double), hipMemcpyDeviceToHost, c->st);
        if (Y_out && !c->Yhi) hipMemcpyAsync(Y_out + (size_t)i * q * T, c->Y + (size_t)tr.v[i] * q * T, (size_t)q * T, hipMemcpyDeviceToHost, c->st);
      }
      if (hipStreamSynchronize(c->st) != hipSuccess) rc = fail("generator copy-out failed");
    }
  }
  hipFree(X); if (dtr) hipFree(dtr); if (flag) hipFree(flag);
  if (rc) return rc;
  if (over[1]) return fail("%d sampled counts exceed 65535 (rates too high for the count tensor)", over[1]);
  c->have_counts = true;
  counts_changed(c, &tr.v);
  c->info["counts_two_bytes"] = c->Yhi ? 1.0 : 0.0;
  if (Y_out && c->Yhi) return fail("sampled counts exceed 255: the uint8 output cannot hold them, read them back with pgpfa_get_counts_u16");
  return 0;
}

// util.leaveOneOutPrediction (util.py:289-334): for every listed trial and every neuron, the Laplace mode of the latents
// given all other neurons (cold start, same Newton machinery with that neuron's likelihood term dropped) and the
// held-out neuron's predicted rate per bin; R*q mode searches, batched like trials.
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
    CHK(estep_impl(c, redo, 0, false, nullptr, it2.data(), st2.data(), &job2));
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
static int ensure_lambda(pgpfa_ctx* c);
static int dual_common(pgpfa_ctx* c, int nb, std::vector<double>* sB, std::vector<double>* sD, std::vector<double>* vKv);
static int dual_jitter(pgpfa_ctx* c, int nb);
static int post_cov_dual_impl(pgpfa_ctx* c, int trial, double* out);

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

static int ensure_trial_vsmgp(pgpfa_ctx* c, const std::vector<int>& trials) {
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

// ---- M-step ------------------------------------------------------------------------------------------
// One (C,d) cost / gradient sweep over the trials of the last E-step at the parameters in c->vec: c->cdout <- per-neuron sums
// [(p+2)][q] (rows 0..p-1: sum (y - yhat) m - yhat V c, row p: sum (y - yhat), row p+1: sum (y hh - yhat)).  Matrix-core kernel
// up to 20 latents (mstep.h), the vector kernel beyond that or with option cd_mfma = 0.
// count terms of the (C,d) cost, linear in (c_n, d_n): c->cdym[(p+1)][q] = sum_t y m_t | sum_t y over the trials of the last E-step
static int ensure_cdym(pgpfa_ctx* c) {
  if (c->cdym_valid) return 0;
  const int q = c->q, p = c->p, ntr = (int)c->last_trials_h.size();
  const int nbk = std::max(1, std::min(1024, ntr));
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
  if (!c) return fail("null context");
  if (!c->have_post) return fail("no E-step result resident");
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
int pgpfa_mstep_tau_costgrad_multi(pgpfa_ctx* c, int m, const double* logp, double* cost, double* grad) {
  if (!c || !logp || !cost || !grad) return fail("null argument");
  if (m < 1 || m > TAU_MULTI_MAX) return fail("between 1 and %d candidates per latent (m=%d)", TAU_MULTI_MAX, m);
  if (!c->have_precomp) return fail("pgpfa_mstep_precomp has not been called");
  HIPC(hipSetDevice(c->device));
  const int Tp = c->Tp, p = c->p, nq = m * p;
  const size_t slab = (size_t)Tp * Tp;
  for (int k = 0; k < nq; ++k)
    if (!std::isfinite(logp[k])) return fail("log-gamma[%d] is not finite", k % p);
  double* dlogp = c->tscal + 16;                     // [nq]
  double* dres = c->tscal + 16 + nq;                 // [4][nq]: logdet, tr(KinvP), tr(KinvM), tr(KinvMKinvP)
  CHK(upload(c, dlogp, logp, nq));
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
  std::vector<double> h(4 * (size_t)nq);
  std::vector<int> info(nq);
  CHK(dl_enqueue(c, h.data(), dres, 4 * nq * sizeof(double)));
  CHK(dl_enqueue(c, info.data(), c->kws.info, sizeof(int) * nq));
  CHK(dl_flush(c));
  const double R = c->n_trials_global;
  for (int k = 0; k < nq; ++k) {
    if (info[k] != 0) return fail("timescale Gram matrix of latent %d not positive definite at log-gamma=%g", k % p, logp[k]);
    cost[k] = 0.5 * R * h[k] + 0.5 * h[nq + k];
    const double dE = -0.5 * R * h[2 * nq + k] + 0.5 * h[3 * nq + k];
    grad[k] = -dE * std::exp(logp[k]);
  }
  return 0;
}

int pgpfa_mstep_tau_costgrad_batch(pgpfa_ctx* c, const double* logp, double* cost, double* grad) {
  return pgpfa_mstep_tau_costgrad_multi(c, 1, logp, cost, grad);
}

// ---- dual variational E-step (inference.py:188-432) ----------------------------------------------------
static int ensure_lambda(pgpfa_ctx* c) {
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
static int dual_common(pgpfa_ctx* c, int nb, std::vector<double>* sB, std::vector<double>* sD, std::vector<double>* vKv) {
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
    hipLaunchKernelGGL(dual_unpack_w_kernel, dim3((unsigned)(((size_t)T * np + 255) / 256), nb), dim3(256), 0, c->st, c->dual_scr, c->dual_sscr, c->W, sW,
                       T, p);
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
    launch_post_vsm(c, (const double*)c->ws.Mt, (long long)c->ws.sM, c->npad, 1, 0);
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
static int post_cov_dual_impl(pgpfa_ctx* c, int trial, double* out) {
  CHK(ready(c));
  CHK(ensure_lambda(c));
  std::vector<int> tr{trial};
  CHK(upload_list(c, c->trial_of_slot, tr));
  const size_t m = (size_t)c->q * c->T;
  HIPC(hipMemcpyAsync(c->lamd, c->lam_keep + (size_t)trial * m, m * sizeof(double), hipMemcpyDeviceToDevice, c->st));
  return dual_post_cov_dev(c, out, nullptr);
}

// the reference's 1e-6 relative jitter on the diagonal of the posterior precision, applied to the W blocks of the slots [0, nb) in place
static int dual_jitter(pgpfa_ctx* c, int nb) {
  hipLaunchKernelGGL(dual_jitter_kernel, dim3((unsigned)((c->T * c->p + 255) / 256), nb), dim3(256), 0, c->st, c->W, (long long)c->T * c->p * c->p,
                     c->Kinv, c->Tp, c->T, c->p, 1e-6);
  HIPC(hipGetLastError());
  return 0;
}

// dualProblem_grad (inference.py:218) of the slots [0, nb) into c->dgrad from K v (c->KD) and the per-bin covariance blocks in c->vsm
static int dual_gradient(pgpfa_ctx* c, int nb) {
  const int q = c->q, p = c->p, T = c->T;
  if (c->dual_gemm && c->mfma) {
    hipLaunchKernelGGL(dual_pack_sigma_kernel, dim3((unsigned)(((size_t)T * c->dual_npd + 255) / 256), nb), dim3(256), 0, c->st, c->vsm,
                       c->trial_of_slot, c->dual_scr, c->dual_sscr, T, p, c->dual_npd);
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
static int dual_eval_slots(pgpfa_ctx* c, int nb, const std::vector<int>& tos, bool want_grad, double* cost, bool tolerate = false) {
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
    launch_post_vsm(c, (const double*)c->ws.Mt, (long long)c->ws.sM, c->npad, nb, 0);
    CHK(dual_gradient(c, nb));
  }
  return 0;
}

// out[slot][n][t] = 1/2 c_n^T Sigma_t c_n of the slots [0, nb) from the per-bin covariance blocks in c->vsm (the variance term of the
// reference's dual gradient, inference.py:218)
static int var_offsets(pgpfa_ctx* c, int nb, double* out) {
  const int q = c->q, p = c->p, T = c->T;
  if (c->dual_gemm && c->mfma && c->dual_tbl) {
    hipLaunchKernelGGL(dual_pack_sigma_kernel, dim3((unsigned)(((size_t)T * c->dual_npd + 255) / 256), nb), dim3(256), 0, c->st, c->vsm,
                       c->trial_of_slot, c->dual_scr, c->dual_sscr, T, p, c->dual_npd);
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
      if (!c->lam_resident[t] && !c->trial_dual[t]) return fail("trial %d has no resident dual optimum (start = 3)", t);
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

static int check_distinct(const std::vector<int>& v) {
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
    if (!c->lam_resident[t] && !c->trial_dual[t]) return fail("trial %d has no resident dual variables", t);
  for (size_t i = 0; i < tr.v.size(); ++i) {
    CHK(dl_enqueue(c, out + i * m, c->lam_keep + (size_t)tr.v[i] * m, m * sizeof(double)));
  }
  return dl_flush(c);
}

int pgpfa_dual_finalize(pgpfa_ctx* c, int n, const int32_t* idx, const double* lam, double* nlp_sum) {
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
      if (lam) { CHK(copy_dev(c, c->lam_keep + (size_t)tos[s] * mq, c->lamd + (size_t)s * mq, mq * sizeof(double))); c->lam_resident[tos[s]] = 0; }
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

// ---- multi-GPU ---------------------------------------------------------------------------------------------
int pgpfa_comm_unique_id(char* id128) {
  if (!id128) return fail("null argument");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId failed: %s", ncclGetErrorString(r));
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  std::memcpy(id128, &id, 128);
  return 0;
}

int pgpfa_comm_init(pgpfa_ctx* c, const char* id128, int rank, int nranks) {
  if (!c || !id128) return fail("null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail("invalid rank %d of %d", rank, nranks);
  HIPC(hipSetDevice(c->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) { c->comm = nullptr; return fail("ncclCommInitRank failed: %s", ncclGetErrorString(r)); }
  c->rank = rank;
  c->nranks = nranks;
  return 0;
}

int pgpfa_comm_allreduce_host(pgpfa_ctx* c, double* buf, int count) {
  if (!c || !buf || count < 0) return fail("invalid argument");
  if (!c->comm) return 0;
  HIPC(hipSetDevice(c->device));
  if ((size_t)count > c->commbuf_len) {
    CHK(dmalloc(c, &c->commbuf, (size_t)count));
    c->commbuf_len = count;
  }
  CHK(upload(c, c->commbuf, buf, count));
  CHK(allreduce_dev(c, c->commbuf, count));
  return download(c, buf, c->commbuf, count);
}

// ---- crash diagnostics (opt-in: PGPFA_BACKTRACE=1 in the environment when the library is loaded) ------------------------------
// SIGSEGV / SIGABRT print the native call stack of the faulting thread to stderr (module + offset: resolve with addr2line) and
// then take the default action.  The GPU boxes write no core files and a debugger changes the timing: this is what is left.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
struct sigaction g_prev_action[65];
void pgpfa_crash_handler(int sig, siginfo_t* info, void* uctx) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char head[] = "\npgpfa: fatal signal, native backtrace:\n";
  (void)!write(2, head, sizeof head - 1);
  backtrace_symbols_fd(frames, n, 2);
  // hand over to whoever was installed before (Python's faulthandler prints the interpreter's stack), else the default action
  const struct sigaction& prev = g_prev_action[sig];
  if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction) { prev.sa_sigaction(sig, info, uctx); return; }
  if (!(prev.sa_flags & SA_SIGINFO) && prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN && prev.sa_handler) { prev.sa_handler(sig); return; }
  signal(sig, SIG_DFL);
  raise(sig);
}
struct PgpfaCrashInit {
  PgpfaCrashInit() {
    const char* e = std::getenv("PGPFA_BACKTRACE");
    if (e && e[0] == '1') {
      void* warm[2];
      (void)backtrace(warm, 2);                       // (loads libgcc now, not inside the handler)
      static char altstack[1 << 16];
      stack_t cur{};
      if (sigaltstack(nullptr, &cur) == 0 && (cur.ss_flags & SS_DISABLE)) {
        stack_t ss{};
        ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
      }
      for (int sig : {SIGSEGV, SIGABRT, SIGBUS}) {
        struct sigaction sa{};
        sa.sa_sigaction = pgpfa_crash_handler;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        sigaction(sig, &sa, &g_prev_action[sig]);
      }
    }
  }
} g_pgpfa_crash_init;
}  // namespace

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
  if (batch < 1 || reps < 1 || (phases != 0 && phases != 1 && phases != 3)) return fail("batch, reps >= 1; phases 0, 1 or 3");
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
    if (phases == 3)
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

}  // extern "C"
