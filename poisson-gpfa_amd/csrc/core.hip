// libpgpfa_hip.so - core.hip (one translation unit of the C-ABI library; shared declarations: ctx.h)
#include "ctx.h"
#include "model.h"
#include "dual.h"

using namespace pgpfa;

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
  if (!P.on || (P.only_tag >= 0 && tag != P.only_tag && tag != TAG_MIX)) return;     // (the one mixing / product-and-mixing launch per E-step rides along)
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


// allocate a factor workspace: nslots slabs of ld x ld (+ Mt), diagonal inverses, scratch panel
int alloc_cholws(pgpfa_ctx* c, CholWS* w, int nslots, int npad, bool with_mt, size_t slab_elems, bool zero_mt, size_t mt_elems) {
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
// (round 5: measured at both ends - bench.py --workload floor, tests/test_gpu_round5.py - the low-rank engine is still 9 % faster at a flop
// ratio of 1.6 (100 x 5 x 400, every timescale 3 bins: rank 1520 of 2000) and 23 % at 1.1 (200 x 10 x 500, rank 3856 of 5000: 218 against 284 ms
// per EM iteration of 128 trials): its cubic term runs on r x r systems, its T^2 term mostly on the FP16 / FP32 matrix cores, and the dense engine
// reaches 41 TFLOP/s on 0.72 n^3.  The old rule - ratio below 0.5 and rank at most half of n - sent all of that to the dense engine.)
bool lowrank_pays(const pgpfa_ctx* c) {
  const double n = c->n, r = c->rpad, T = c->T, p = c->p;
  if (c->p > WIDE_MAX || c->rpad < NB || (size_t)c->rpad > (size_t)c->ld) return false;
  const double dense = 0.72 * n * n * n;
  const double lr = 6.0 * T * r * r + 0.7 * r * r * r + p * T * T * r;
  // (small systems - configs 1 and 2 - are launch-bound under either engine and the dense one has fewer launches: there the old rule stands)
  if (c->npad < 1536) return c->rpad * 2 <= c->npad && lr < 0.5 * dense;
  return lr < 1.7 * dense;
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
//
// Arenas of closed contexts stay in a process-wide pool, address range AND mapped memory (round 5).  Two facts of this stack decide that:
// hipMemAddressFree crashed inside the runtime about once in ten runs of a test sequence that opens and closes a few dozen contexts (native
// backtrace: arena_release -> hipMemAddressFree -> libamdhip64; never under a debugger), so ranges were already kept; and memory given back with
// hipMemUnmap + hipMemRelease is NOT returned to the device's free memory while its range lives - tools/leak_probe.py: 308 GB free, a context
// with a 91-GB arena, closed: 217 GB free; the next context's hipMemCreate calls are served from those 91 GB, hipMalloc and hipMemGetInfo never
// see them again.  A plan sized by hipMemGetInfo then shrank with every large context the process had closed before (the config-5 test got 32
// slots per chunk behind the rest of the suite, 128 alone).  Kept mapped, the memory is at least accounted for: the next context starts with
// the pooled arena attached (nothing to map, nothing to clear) and the budget counts it.  When growth fails the other pooled arenas are released.
struct PooledArena { void* va; size_t va_size; std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> chunks; size_t cap; };
std::mutex g_va_mu;
std::vector<PooledArena> g_va_pool;

static void pool_release_chunks(PooledArena& pa) {
  size_t off = 0;
  for (auto& ch : pa.chunks) { hipMemUnmap(reinterpret_cast<char*>(pa.va) + off, ch.second); hipMemRelease(ch.first); off += ch.second; }
  pa.chunks.clear();
  pa.cap = 0;
  (void)hipGetLastError();
}

// bytes of pooled arenas no context holds (device `dev`: ranges are per process, memory per device - contexts of one process on several
// devices are rare enough to ignore the distinction: a range is only re-attached on the device it was mapped for)
static size_t pool_bytes() {
  std::lock_guard<std::mutex> lk(g_va_mu);
  size_t n = 0;
  for (auto& pa : g_va_pool) n += pa.cap;
  return n;
}

// first use of the arena by this context: virtual-memory management available?  reserve a range, or take a pooled one (with its memory)
static void arena_init(pgpfa_ctx* c) {
  if (c->vmm != 0) return;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c->device;
  size_t gran = 0, free_b = 0, total_b = 0;
  void* va = nullptr;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0 &&
      hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    const size_t va_size = (total_b + gran - 1) / gran * gran;
    const size_t G = std::max(gran, c->vmm_granule) / gran * gran;
    {
      std::lock_guard<std::mutex> lk(g_va_mu);
      // the pooled arena with the most memory whose chunks have this context's chunk size and device
      int best = -1;
      for (size_t i = 0; c->use_pool && i < g_va_pool.size(); ++i) {
        const PooledArena& pa = g_va_pool[i];
        if (pa.va_size != va_size) continue;
        const bool fits = pa.chunks.empty() || pa.chunks.front().second == G;
        if (fits && (best < 0 || pa.cap > g_va_pool[best].cap)) best = (int)i;
      }
      if (best >= 0) {
        PooledArena pa = std::move(g_va_pool[best]);
        g_va_pool.erase(g_va_pool.begin() + best);
        va = pa.va;
        c->vmm_chunks = std::move(pa.chunks);
        c->arena_cap = pa.cap;
        c->bytes += pa.cap;
      }
    }
    if (va || (hipMemAddressReserve(&va, va_size, gran, nullptr, 0) == hipSuccess && va)) {
      c->arena = reinterpret_cast<char*>(va); c->va_size = va_size; c->vmm_gran = gran; c->vmm = 1;
      c->info["arena_bytes"] = (double)c->arena_cap;
    }
  }
  if (c->vmm != 1) { (void)hipGetLastError(); c->vmm = -1; }
}

// budget_ms > 0: stop mapping once `must` bytes are there and the call has taken that long (returns 0 with arena_cap < need: the caller plans with what
// it has).  Mapping is not a fixed price on this stack: 116 GB in 2 ms in the first process on a box, 4.3 s for the same growth in the next one
// (tools/jump_probe.py, round 6) - pages another process has just given back are cleared before they are handed out again.
int arena_grow(pgpfa_ctx* c, size_t need, double budget_ms, size_t must) {
  g_err.clear();
  arena_init(c);
  const auto t_start = std::chrono::steady_clock::now();
  if (c->vmm == 1) {
    const size_t gran = c->vmm_gran;
    // physical chunks of ONE granule each (one hipMemCreate / hipMemMap / hipMemSetAccess per chunk; 1 GiB by default).  Measured on this
    // stack: hipMemSetAccess returns "invalid argument" for a chunk whose size differs from the first one mapped into the range (13 chunks
    // of 4 GiB, then a 1-GiB remainder: fails; 1 GiB then 4 GiB: fails), so every chunk has the same size.
    const size_t G = std::max(gran, c->vmm_granule) / gran * gran;
    size_t want = need > c->arena_cap ? (need - c->arena_cap + G - 1) / G * G : 0;
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
    bool pool_trimmed = false;
    while (ok && want > 0) {
      const size_t add = std::min(want, std::max(piece_max, gran));
      hipMemGenericAllocationHandle_t h;
      err = hipMemCreate(&h, add, &prop, 0);
      if (err != hipSuccess && !pool_trimmed) {
        // out of memory: the arenas other closed contexts left in the pool give theirs up (the runtime serves later hipMemCreate calls from it)
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(g_va_mu);
        for (auto& pa : g_va_pool) pool_release_chunks(pa);
        pool_trimmed = true;
        continue;
      }
      if (err != hipSuccess) { what = "hipMemCreate"; ok = false; break; }
      err = hipMemMap(c->arena + c->arena_cap, add, 0, h, 0);
      if (err != hipSuccess) { what = "hipMemMap"; hipMemRelease(h); ok = false; break; }
      err = hipMemSetAccess(c->arena + c->arena_cap, add, &desc, 1);
      if (err != hipSuccess) { what = "hipMemSetAccess"; hipMemUnmap(c->arena + c->arena_cap, add); hipMemRelease(h); ok = false; break; }
      c->vmm_chunks.emplace_back(h, add);
      c->arena_cap += add;
      c->bytes += add;
      want -= add;
      if (budget_ms > 0.0 && c->arena_cap >= must &&
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() > budget_ms) break;
    }
    c->info["arena_bytes"] = (double)c->arena_cap;
    if (ok) return 0;
    (void)hipGetLastError();
    // (the arena only grows between plans, when nothing in it is live: give the range back and carry on with one plain allocation)
    c->info["arena_vmm_failed"] = 1.0;
    std::fprintf(stderr, "pgpfa: workspace arena: %s failed (%s) growing from %zu to %zu bytes; falling back to hipMalloc\n", what,
                 hipGetErrorString(err), c->arena_cap, need);
    c->bytes -= c->arena_cap;
    arena_release(c, true);
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

// (unmap: also release the physical chunks - the fallback path, which is about to try plain allocations; a closing context keeps them mapped
//  in the pool)
void arena_release(pgpfa_ctx* c, bool unmap) {
  if (c->vmm == 1) {
    if (c->arena) {
      PooledArena pa{c->arena, c->va_size, std::move(c->vmm_chunks), c->arena_cap};
      if (unmap) pool_release_chunks(pa);
      std::lock_guard<std::mutex> lk(g_va_mu);
      g_va_pool.push_back(std::move(pa));
    }
    c->vmm_chunks.clear();
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
  const auto t_plan = std::chrono::steady_clock::now();
  struct PlanTimer { pgpfa_ctx* c; std::chrono::steady_clock::time_point t0; ~PlanTimer() {
    c->info["plan_ms_total"] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); c->info["plans"] += 1.0; } } plan_timer{c, t_plan};
  CHK(free_workspace(c));
  c->ws_mark = c->allocs.size();
  c->plan_lowrank = plan_lr;
  size_t free_b = 0, total_b = 0;
  arena_init(c);                                                              // (a pooled arena of a closed context comes with its memory)
  HIPC(hipMemGetInfo(&free_b, &total_b));
  // the arena's bytes are ours to re-partition; so are those of pooled arenas nobody holds (arena_grow releases them when it must)
  const size_t avail = (size_t)(0.85 * (double)(free_b + c->arena_cap + (c->vmm == 1 ? pool_bytes() : 0)));
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
    // (round 6) a RE-plan prefers the largest head-room that fits into the memory the arena already holds - re-partitioning costs nothing, mapping may
    // cost seconds (arena_grow) - and otherwise grows for a quarter of head-room only; the first plan of a context takes the full factor
    const double h = std::max(1.0, c->arena_headroom);
    const size_t shared_b = (3 * ld_bytes(c) + 1024 * (size_t)c->ld * sizeof(double) * 4) + ((size_t)64 << 20);
    const size_t mapped = c->arena_cap > shared_b ? c->arena_cap - shared_b : 0;
    auto sized = [&](double hh, size_t* ws, size_t* wm) {
      *ws = std::max(slab, std::min(dense, (size_t)((double)slab * hh) / 1024 * 1024));
      *wm = std::max(mt, std::min(dense, (size_t)((double)mt * hh * hh) / 1024 * 1024));
      return per_slot_bytes(c, *ws, *wm) * (size_t)std::max(target, 1);
    };
    double pick = c->arena_cap > 0 ? std::min(h, 1.25) : h;
    if (c->arena_cap > 0)
      for (double hh : {h, 1.75, 1.5, 1.25, 1.1}) {
        size_t ws, wm;
        if (hh <= h && sized(hh, &ws, &wm) <= mapped) { pick = hh; break; }
      }
    size_t want_slab, want_mt;
    if (sized(pick, &want_slab, &want_mt) <= budget) { c->slab_elems = want_slab; c->mt_elems = want_mt; }
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
  CHK(dmalloc(c, &c->sc_part2, 3 * (size_t)((c->T + 31) / 32) * nB));       // (per (slot, bin tile) partial sums; 32-bin tiles beyond 10 latents)
  // (rows of the component-major form start on 128-byte lines; beyond 10 latents the step's kernels read whole row chunks of the packed triangle at the
  //  template width - up to 210 rows - past a slot's own: one slot's worth of slack behind the last)
  CHK(dmalloc(c, &c->W32, (size_t)round_up(c->T, 32) * (c->p * (c->p + 1) / 2) * nB + (size_t)round_up(c->T, 32) * 210 + 64, true));
  static_assert(sizeof(PcgCtl) <= 64, "the control block is the first 16 words of the solve's upload block");
  CHK(dmalloc(c, &c->pcg_blk, 16 + 4 * nB, true));
  if (c->pcg_blk) {
    c->pcgctl = reinterpret_cast<PcgCtl*>(c->pcg_blk); c->list_a = c->pcg_blk + 16; c->live = c->list_a + nB;
    c->pcg_eta = reinterpret_cast<float*>(c->live + nB); c->live1 = c->live + 2 * nB;
  }
  CHK(dmalloc(c, &c->pcg_ratio, nB, true));
  CHK(dmalloc(c, &c->GbT, (size_t)c->T * (c->p * (c->p + 1) / 2) + 64)); CHK(dmalloc(c, &c->WbT, (size_t)c->T * (c->p * (c->p + 1) / 2) + 64));
  CHK(dmalloc(c, &c->sc_f, nB));
  CHK(dmalloc(c, &c->sc_alpha, nB));
  CHK(dmalloc(c, &c->trial_of_slot, nB)); CHK(dmalloc(c, &c->list_b, nB));
  CHK(dmalloc(c, &c->mask_of_slot, nB));
  CHK(dmalloc(c, &c->ident, nB));
  return 0;
  };
  int rc_carve = 0;
  auto measure = [&](int nslots, size_t* bytes) -> int {
    c->B = nslots;
    c->arena_mode = 1; c->arena_off = 0;
    const int rc = carve();
    c->arena_mode = 0;
    *bytes = c->arena_off + ((size_t)1 << 20);
    return rc;
  };
  size_t need = 0;
  CHK(measure(c->B, &need));
  if (need > c->arena_cap) {
    HIPC(hipStreamSynchronize(c->st));
    // (round 6) growth under a time budget (option workspace_grow_budget_ms): whatever it costs the arena gets what `floor_slots` slots need; beyond that
    // mapping stops when the budget is spent and the chunk takes the slots that fit - two chunks of 512 trials cost a few per cent, mapping 100 GB of
    // pages another process has just released costs seconds
    const int B_full = c->B;
    const int floor_slots = std::min(B_full, std::max(8, c->grow_floor_slots));
    size_t must = need;
    if (floor_slots < B_full) CHK(measure(floor_slots, &must));
    const auto t_grow = std::chrono::steady_clock::now();
    const size_t cap_before = c->arena_cap;
    // (not bounded: the first plan of a context - a fit that starts is better off with its whole arena than with a few slow first iterations -
    //  and a plan for a LONGER trial list than the last one: that memory is asked for by the caller, not by ranks that drifted)
    const bool bounded = cap_before > 0 && target <= c->plan_target;
    const int rc_grow = arena_grow(c, need, bounded ? c->grow_budget_ms : 0.0, must);
    const double grow_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_grow).count();
    c->info["arena_grow_ms_total"] += grow_ms;
    if (std::getenv("PGPFA_PLAN_TRACE"))
      std::fprintf(stderr, "pgpfa: plan %s, %d slots: arena %.1f -> %.1f GB (asked %.1f) in %.0f ms\n", plan_lr ? "low-rank" : "dense", B_full, (double)cap_before / 1e9,
                   (double)c->arena_cap / 1e9, (double)need / 1e9, grow_ms);
    if (rc_grow) {
      c->B = 0;
      const std::string why = g_err;
      return fail("not enough device memory for the chunk workspace (%zu bytes needed, %zu free)%s%s", need, free_b, why.empty() ? "" : ": ", why.c_str());
    }
    int B_fit = B_full;
    if (c->arena_cap < need) {
      // the budget ran out: the largest balanced chunk that fits what is mapped
      const double per_b = (double)(need - must) / (double)std::max(1, B_full - floor_slots);
      B_fit = floor_slots + (int)((double)(c->arena_cap - must) / std::max(per_b, 1.0));
      B_fit = std::max(floor_slots, std::min(B_full, B_fit / 8 * 8));
      const int nchunks = (target + B_fit - 1) / B_fit;
      B_fit = std::max(floor_slots, std::min(B_fit, ((target + nchunks - 1) / nchunks + 7) / 8 * 8));
      for (;;) {
        CHK(measure(B_fit, &need));
        if (need <= c->arena_cap || B_fit <= floor_slots) break;
        B_fit = std::max(floor_slots, B_fit - 8);
      }
      if (need > c->arena_cap) { c->B = 0; return fail("workspace arena short of its floor (%zu bytes needed, %zu mapped)", need, c->arena_cap); }
      // half the list per chunk is good enough to stay with until the ranks ask for a new plan; less than that keeps growing at every call
      c->B_capped = 2 * B_fit >= target;
    }
    c->B = B_fit;
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
  c->plan_target = target;
  if (std::getenv("PGPFA_PLAN_TRACE"))
    std::fprintf(stderr, "pgpfa: plan %s, %d slots (target %d), slab %zu + %zu elements, %.1f GB carved, %.0f ms\n", plan_lr ? "low-rank" : "dense", c->B, target,
                 c->slab_elems, c->mt_elems, (double)need / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan).count());
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
int copy_dev(pgpfa_ctx* c, void* dst, const void* src, size_t bytes) {
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
  if (T > 256 && c->pivchol_pairs && Tp % 2 == 0)
    hipLaunchKernelGGL((rbf_pivchol2_kernel<256, 4>), dim3(p), dim3(1024), pivchol2_lds(T, rmax, 1024, 4), st, c->Flr, Tp, T, c->tau, c->bin, c->eps, c->lr_tol, rmax,
                       c->d_rank);
  else if (T > 256)
    hipLaunchKernelGGL((rbf_pivchol_kernel<512, 2>), dim3(p), dim3(1024), shm, st, c->Flr, Tp, T, c->tau, c->bin, c->eps, c->lr_tol, rmax, c->d_rank);
  else
    hipLaunchKernelGGL((rbf_pivchol_kernel<256, 1>), dim3(p), dim3(256), shm, st, c->Flr, Tp, T, c->tau, c->bin, c->eps, c->lr_tol, rmax, c->d_rank);
  HIPC(hipGetLastError());
  return 0;
}

// (pivchol_launched: the kernel is already running on the side stream and c->ev_join marks its end)
int build_lowrank(pgpfa_ctx* c, bool pivchol_launched) {
  const int p = c->p, T = c->T, Tp = c->Tp;
  if (pivchol_launched) HIPC(hipStreamWaitEvent(c->st, c->ev_join, 0));
  else CHK(launch_pivchol(c, c->st));
  std::vector<int> r(p);
  CHK(dl_enqueue(c, r.data(), c->d_rank, sizeof(int) * p));
  CHK(dl_flush(c));
  // Rank tables.  rk[k]: the latent's rank rounded up to 16 - the size every product with F_k is issued with (columns of F_k past the rank are zero:
  // the pivoted Cholesky clears its slab first).  roff[k]: where the latent's rows / columns START in the r x r system and in L^-T.
  // rank_gran = 16 (rounds 1-5): roff[k + 1] = roff[k] + rk[k] - every latent owns whole 16-blocks, 7.5 % (late) to 18 % (early iterations of the
  // bench) of the rank total is padding, identity rows that the r x r factorisation, its inverse, the columns of Yt and of the split sums pay for.
  // rank_gran = 4 (round 6): COMPACT offsets roff[k + 1] = roff[k] + round_up(rank, 4).  A product with F_k still takes rk[k] rows from roff[k]
  // on - the last few belong to the next latent and meet zero columns of F_k - so only two kernels need to know: B = I + F^T Wt F is still
  // computed on 16-blocks that never straddle a latent (assemble_b, in the PADDED index space roff16 / blk_lat / blk_col) and stored at its
  // compact place (cmap), and the panel of L^-T that yt_mix stages keeps padded rows (gathered through the same map).
  // (compact offsets need the preconditioner's three products as thin.h's kernels: the general GEMM's row-tile tables start tiles on multiples of 16)
  const bool thin_all = c->thin_products >= 2 && c->mfma && T >= 4;
  const int G = ((c->rank_gran == 4 || c->rank_gran == 8) && thin_all) ? c->rank_gran : 16;
  c->rk.assign(p, 0);
  c->roff.assign(p + 1, 0);
  c->rr.assign(p, 0);
  c->roff16.assign(p + 1, 0);
  int extent = 0;
  for (int k = 0; k < p; ++k) {
    c->rr[k] = round_up(std::max(r[k], 1), G);
    c->rk[k] = round_up(c->rr[k], 16);
    c->roff[k + 1] = c->roff[k] + c->rr[k];
    c->roff16[k + 1] = c->roff16[k] + c->rk[k];
    extent = std::max(extent, c->roff[k] + c->rk[k]);
  }
  c->rtot = c->roff[p];
  c->rtot16 = c->roff16[p];
  c->rank_compact = (G != 16);
  c->rpad = round_up(std::max(c->rtot, extent), NB);
  const int nblk = round_up(c->rtot16, NB) / 16;
  std::vector<int> lat(nblk, -1), col(nblk, 0);
  for (int k = 0; k < p; ++k)
    for (int b = c->roff16[k] / 16; b < c->roff16[k + 1] / 16; ++b) { lat[b] = k; col[b] = b * 16 - c->roff16[k]; }
  CHK(upload_list(c, c->d_blk_lat, lat));
  CHK(upload_list(c, c->d_blk_col, col));
  CHK(upload_list(c, c->d_roff, c->roff));
  CHK(upload_list(c, c->d_roff16, c->roff16));
  if (c->rank_compact) {
    // padded index -> compact index (-1: a padding row); nrtab[b]: padded rows that can hold something in the 16-column block b of L^-T (it is upper
    // triangular in the compact order: rows up to compact index 16 b + 15), rounded up to a whole 16-row group of the latent that holds the last one
    const int n16 = round_up(c->rtot16, NB);
    std::vector<int> cmap(n16, -1), pmap(round_up(c->rtot, 16) + 16, 0);
    for (int k = 0; k < p; ++k)
      for (int j = 0; j < c->rr[k]; ++j) { cmap[c->roff16[k] + j] = c->roff[k] + j; pmap[c->roff[k] + j] = c->roff16[k] + j; }
    std::vector<int> nrtab((round_up(c->rtot, 16)) / 16, 0);
    for (size_t b = 0; b < nrtab.size(); ++b) {
      const int last = std::min((int)b * 16 + 15, c->rtot - 1);
      nrtab[b] = std::min(c->rtot16, round_up(pmap[last] + 1, 16));
    }
    CHK(upload_list(c, c->d_cmap, cmap));
    CHK(upload_list(c, c->d_nrtab, nrtab));
  }
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
      // (rr, not rk: these rows are WRITTEN - past the latent's own they are the next latent's)
      const int ngr = (c->rr[k] + 63) / 64, per = round_up((c->rr[k] + ngr - 1) / ngr, 4);
      for (int m0 = 0; m0 < c->rr[k]; m0 += per) { hft.push_back(k); hft.push_back(m0); hft.push_back(std::min(per, c->rr[k] - m0)); hft.push_back(c->roff[k]); }
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


// distinct = true for every entry point that writes per-trial state (two slots of one chunk scattering to the same
// trial row would race, and the device list of the last E-step holds R entries)
int resolve_trials(pgpfa_ctx* c, int n, const int32_t* idx, Trials* out, bool distinct) {
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
  c->lam_valid.assign(R, 0);
  rc |= dmalloc(c, &c->Flr, slab * p + 256 * (size_t)c->Tp, true);
  rc |= dmalloc(c, &c->d_rank, p); rc |= dmalloc(c, &c->d_roff, p + 1); rc |= dmalloc(c, &c->d_roff16, p + 1);
  rc |= dmalloc(c, &c->d_cmap, (size_t)p * c->Tp + 2 * NB + 64); rc |= dmalloc(c, &c->d_nrtab, (size_t)p * c->Tp / 16 + 64);
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
  if (const char* g = std::getenv("PGPFA_RANK_GRAN")) { const int v = std::atoi(g); if (v == 4 || v == 8 || v == 16) c->rank_gran = v; }
  c->info["n_pad"] = c->npad;
  c->info["counts_two_bytes"] = 0.0;
  c->info["arena_bytes"] = 0.0;
  c->info["last_eps_wt_norm"] = 0.0;
  c->info["last_eps_wt_rms"] = 0.0;
  c->info["last_split_cov"] = 0.0;
  c->info["last_yt_mix_fused"] = 0.0;
  for (const char* k : {"plan_ms_total", "plans", "set_params_calls", "last_retry_ms", "last_dense_retries", "last_param_step", "last_param_step_prev",
                        "last_fallback_no_descent", "last_fallback_line_search", "last_fallback_outer_cap", "arena_grow_ms_total", "last_cold_restarts"}) c->info[k] = 0.0;
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
  if (c->tau_pin) hipHostFree(c->tau_pin);
  if (c->ev_tau_fork) hipEventDestroy(c->ev_tau_fork);
  if (c->ev_tau_done) hipEventDestroy(c->ev_tau_done);
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
  else if (k == "pcg_vec32") c->pcg_vec32 = (int)v;
  else if (k == "pcg_rx32") c->pcg_rx32 = (int)v;
  else if (k == "pcg_adapt") c->pcg_adapt = (int)v;
  else if (k == "pcg_xcd") c->pcg_xcd = (int)v;
  else if (k == "mt_fill") c->mt_fill = (int)v;
  else if (k == "overlap_factors") c->overlap_factors = (int)v;
  else if (k == "mix_slot") c->mix_slot = (int)v;
  else if (k == "yt_mix") c->yt_mix = (v != 0.0);
  else if (k == "pivchol_pairs") c->pivchol_pairs = (v != 0.0);
  else if (k == "vsm_b4") c->vsm_b4 = (int)v;
  else if (k == "poisson_tiles") c->poisson_tiles = (int)v;
  else if (k == "yt_mix_dbg") c->yt_mix_dbg = (int)v;
  else if (k == "syrk_dbg") c->syrk_dbg = (int)v;
  else if (k == "syrk_tile") c->syrk_tile = (int)v;
  else if (k == "mix_wide") c->mix_wide = (int)v;
  else if (k == "thin_products") {
    const bool was = c->thin_products >= 2;
    c->thin_products = (int)v;
    // (the rank tables depend on it under compact offsets: rebuild them)
    if (c->have_params && c->rank_gran != 16 && was != (c->thin_products >= 2))
      CHK(pgpfa_set_params(c, std::vector<double>(c->hC).data(), std::vector<double>(c->hd).data(), std::vector<double>(c->htau).data()));
  }
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
  else if (k == "extrapolate_guard") c->extrapolate_guard = v;
  else if (k == "start_guard") c->start_guard = (v != 0.0);
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
  else if (k == "rank_gran") { if (v != 4.0 && v != 8.0 && v != 16.0) return fail("rank_gran is 4, 8 or 16"); c->rank_gran = (int)v; if (c->have_params) CHK(pgpfa_set_params(c, std::vector<double>(c->hC).data(), std::vector<double>(c->hd).data(), std::vector<double>(c->htau).data())); }
  else if (k == "workspace_pool") { if (c->arena_cap > 0) return fail("workspace_pool must be set before the first E-step"); c->use_pool = (v != 0.0); }
  else if (k == "workspace_grow_budget_ms") c->grow_budget_ms = v;
  else if (k == "workspace_grow_floor_slots") c->grow_floor_slots = std::max(1, (int)v);
  else if (k == "workspace_granule_mb") c->vmm_granule = (size_t)std::max(2.0, v) << 20;
  else if (k == "workspace_vmm") { if (c->arena_cap > 0) return fail("workspace_vmm must be set before the first E-step"); c->vmm = (v != 0.0) ? 0 : -1; }
  else if (k == "eps_noise") c->eps = v;
  else if (k == "profile") {
    // 0: off (the accumulated sums stay readable), 1: time every tagged launch, 2: GEMM launches only
    if (v != 0.0) {
      prof_collect(c);
      c->prof.ms.clear(); c->prof.flops.clear(); c->prof.count.clear(); c->prof.max_ms.clear(); c->prof.max_flops.clear(); c->prof.shapes.clear();
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
    c->prof.only_tag = (v == 2.0) ? TAG_GEMM : (v == 3.0) ? TAG_MIX : -1;     // (3: only the product-and-mixing launch of the covariance phase)
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
  static const char* tags[TAG_N] = {"gemm", "potrf", "solve", "poisson", "assemble", "vsm", "cd", "mix"};
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
  if (k == "hbm_bytes_free" || k == "hbm_bytes_total") {
    size_t free_b = 0, total_b = 0;
    HIPC(hipSetDevice(c->device));
    HIPC(hipMemGetInfo(&free_b, &total_b));
    *value = (double)(k == "hbm_bytes_free" ? free_b : total_b);
    return 0;
  }
  if (k == "n_trials_global") { *value = c->n_trials_global; return 0; }
  auto it = c->info.find(k);
  if (it == c->info.end()) return fail("unknown info key '%s'", key);
  *value = it->second;
  return 0;
}

// The resident counts of the listed trials (NULL: all) have been replaced: everything derived from them is stale - the hoisted count
// terms and Hessian sums of the (C,d) M-step, the accumulated covariance sum, and the posterior of those trials.
void counts_changed(pgpfa_ctx* c, const std::vector<int>* trials) {
  c->cdym_valid = false;
  c->cd_hess_valid = false;
  c->pacc_valid = false;
  c->have_precomp = false;
  c->have_post = false;
  if (trials) { for (int t : *trials) { c->vsmgp_ok[t] = 0; c->mode_serial[t] = -10; c->trial_snap[t] = -1; c->trial_dual[t] = 0; c->lam_resident[t] = 0; c->lam_valid[t] = 0; } }
  else {
    std::fill(c->vsmgp_ok.begin(), c->vsmgp_ok.end(), 0); std::fill(c->mode_serial.begin(), c->mode_serial.end(), -10);
    std::fill(c->trial_snap.begin(), c->trial_snap.end(), -1); std::fill(c->trial_dual.begin(), c->trial_dual.end(), 0);
    std::fill(c->lam_resident.begin(), c->lam_resident.end(), 0); std::fill(c->lam_valid.begin(), c->lam_valid.end(), 0);
  }
}

static void drop_high_plane(pgpfa_ctx* c) {
  if (!c->Yhi) return;
  hipStreamSynchronize(c->st);
  hipFree(c->Yhi);
  c->bytes -= (size_t)c->R * c->q * c->T;
  c->Yhi = nullptr;
}
int ensure_high_plane(pgpfa_ctx* c) {
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
  PhaseRange range_phase("pgpfa.set_params");
  if (!c || !C || !d || !tau_s) return fail("null argument");
  if (c->tau_inflight) return fail("a timescale pass is in flight (pgpfa_mstep_tau_costgrad_multi_begin): collect it first");
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
  c->info["set_params_calls"] += 1.0;
  return 0;
}

int get_slabs(pgpfa_ctx* c, const double* src, double* out) {
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

int ready(pgpfa_ctx* c) {
  if (!c) return fail("null context");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  if (!c->have_params) return fail("set_params has not been called");
  if (c->tau_inflight) return fail("a timescale pass is in flight (pgpfa_mstep_tau_costgrad_multi_begin): collect it first");
  HIPC(hipSetDevice(c->device));
  return ensure_workspace(c, false);
}

int ready_estep(pgpfa_ctx* c, bool allow_lowrank) {
  if (!c) return fail("null context");
  if (!c->have_counts) return fail("spike counts have not been uploaded");
  if (!c->have_params) return fail("set_params has not been called");
  if (c->tau_inflight) return fail("a timescale pass is in flight (pgpfa_mstep_tau_costgrad_multi_begin): collect it first");
  HIPC(hipSetDevice(c->device));
  return ensure_workspace(c, allow_lowrank && want_lowrank(c));
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
void snapshot_params(pgpfa_ctx* c, const std::vector<int>& trials) {
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
int with_snapshot(pgpfa_ctx* c, int id, const std::function<int()>& fn) {
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

int remember_trials(pgpfa_ctx* c, const std::vector<int>& v) {
  c->last_trials_h = v;
  CHK(upload_list(c, c->last_trials, v));
  c->have_post = true;
  c->have_precomp = false;
  c->pacc_valid = false;
  c->cdym_valid = false;
  return 0;
}



