// libpgpfa_hip.so - misc.hip (one translation unit of the C-ABI library; shared declarations: ctx.h)
#include "ctx.h"
#include "model.h"
#include "sample.h"

using namespace pgpfa;

int allreduce_dev(pgpfa_ctx* c, double* buf, size_t count) {
  if (!c->comm) return 0;
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, c->st);
  if (r != ncclSuccess) return fail("ncclAllReduce failed: %s", ncclGetErrorString(r));
  return 0;
}


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

int pgpfa_comm_describe(pgpfa_ctx* c, char* buf, int len) {
  if (!c || !buf || len < 2) return fail("invalid argument");
  char bus[64] = "?";
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) != hipSuccess) { (void)hipGetLastError(); std::snprintf(bus, sizeof bus, "?"); }
  if (!c->comm) {
    std::snprintf(buf, (size_t)len, "rank %d/%d device %d pci %s comm none", c->rank, c->nranks, c->device, bus);
    return 0;
  }
  int count = -1, dev = -1;
  if (ncclCommCount(c->comm, &count) != ncclSuccess) count = -1;
  if (ncclCommCuDevice(c->comm, &dev) != ncclSuccess) dev = -1;
  std::snprintf(buf, (size_t)len, "rank %d/%d device %d pci %s comm_ranks %d comm_device %d", c->rank, c->nranks, c->device, bus, count, dev);
  return 0;
}

// ---- crash diagnostics (opt-in: PGPFA_BACKTRACE=1 in the environment when the library is loaded) ------------------------------
// SIGSEGV / SIGABRT print the native call stack of the faulting thread to stderr (module + offset: resolve with addr2line) and
// then take the default action.  The GPU boxes write no core files and a debugger changes the timing: this is what is left.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
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


