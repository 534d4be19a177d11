// thin.h's kernels on synthetic operands of config 3's shape, with per-workgroup clock stamps: where does a launch spend its time?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTHIN_STAMPS -Ipoisson-gpfa_amd/csrc -o tools/probes/thin_probe tools/probes/thin_probe.hip
// run:   tools/probes/thin_probe [live slots, default 400] [rank per latent, default 48]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "thin.h"
using namespace pgpfa;
static int round_up(int a, int b) { return (a + b - 1) / b * b; }
int main(int argc, char** argv) {
  const int live = argc > 1 ? atoi(argv[1]) : 400, rk = argc > 2 ? atoi(argv[2]) : 48;
  const int T = 500, p = 10, S = 1024, Tf = 512, n = p * T, ld = round_up(n, 128), rtot = p * rk, rpad = round_up(rtot, 128);
  std::vector<double> hF((size_t)p * Tf * Tf, 0.0), hFT((size_t)ld * rpad, 0.0), hX((size_t)S * ld), hV((size_t)S * ld);
  for (int l = 0; l < p; ++l)
    for (int j = 0; j < rk; ++j)
      for (int t = 0; t < T; ++t) {
        const double v = ((l * 131 + j * 17 + t * 7) % 97) / 97.0 - 0.5;
        hF[(size_t)l * Tf * Tf + (size_t)j * Tf + t] = v;
        hFT[((size_t)l * T + t) * rpad + l * rk + j] = v;
      }
  for (size_t i = 0; i < hX.size(); ++i) { hX[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5; hV[i] = ((i * 40503u) % 1000) / 1000.0 - 0.5; }
  std::vector<int> tft, tf, cols(S);
  for (int l = 0; l < p; ++l) {
    const int ngr = (rk + 63) / 64, per = round_up((rk + ngr - 1) / ngr, 4);
    for (int m0 = 0; m0 < rk; m0 += per) { tft.push_back(l); tft.push_back(m0); tft.push_back(std::min(per, rk - m0)); tft.push_back(l * rk); }
    for (int t0 = 0; t0 < T; t0 += 256) { tf.push_back(l); tf.push_back(t0); tf.push_back(rk); tf.push_back(l * rk); }
  }
  for (int s = 0; s < S; ++s) cols[s] = (s * 7) % S;       // a shuffled live list
  double *F, *FT, *X, *V, *U, *Y; int *dft, *df, *dcols, *dn; unsigned long long* stamps;
  hipMalloc(&F, hF.size() * 8); hipMalloc(&FT, hFT.size() * 8 + (1 << 20)); hipMalloc(&X, hX.size() * 8); hipMalloc(&V, hV.size() * 8);
  hipMalloc(&U, hX.size() * 8); hipMalloc(&Y, hX.size() * 8);
  hipMalloc(&dft, tft.size() * 4); hipMalloc(&df, tf.size() * 4); hipMalloc(&dcols, S * 4); hipMalloc(&dn, 4);
  hipMalloc(&stamps, 8 * 4096 * 8); hipMemset(stamps, 0, 8 * 4096 * 8);
  hipMemcpy(F, hF.data(), hF.size() * 8, hipMemcpyHostToDevice); hipMemcpy(FT, hFT.data(), hFT.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice); hipMemcpy(V, hV.data(), hV.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dft, tft.data(), tft.size() * 4, hipMemcpyHostToDevice); hipMemcpy(df, tf.data(), tf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dcols, cols.data(), S * 4, hipMemcpyHostToDevice); hipMemcpy(dn, &live, 4, hipMemcpyHostToDevice);
  ThinP a{};
  a.F = F; a.Tf = Tf; a.T = T; a.Tx = T; a.FT = FT; a.ldft = rpad; a.cols = dcols; a.n_dev = dn; a.ncols = S; a.skip = nullptr;
  {
    int nft = 0, nf = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nft, thin_ft_kernel<true>, 512, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, thin_f_kernel<true>, 256, 0);
    printf("occupancy (workgroups per CU the runtime computes): thin_ft %d, thin_f %d\n", nft, nf);
    hipFuncAttributes fa{};
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&thin_ft_kernel<true>));
    printf("thin_ft_kernel<true>: numRegs %d, sharedSizeBytes %zu, maxThreadsPerBlock %d, localSizeBytes %zu\n", fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock, fa.localSizeBytes);
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&thin_f_kernel<true>));
    printf("thin_f_kernel<true>: numRegs %d, sharedSizeBytes %zu, maxThreadsPerBlock %d, localSizeBytes %zu\n", fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock, fa.localSizeBytes);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    ThinP q = a;
    if (which == 0) { q.tab = dft; q.X = X; q.ldx = ld; q.Y = U; q.ldy = ld; }
    else { q.tab = df; q.X = V; q.ldx = ld; q.Y = Y; q.ldy = ld; }
    q.stamps = stamps;
    const dim3 grid(which == 0 ? (unsigned)tft.size() / 4 : (unsigned)tf.size() / 4, (S + 15) / 16);
    auto go = [&]() { if (which == 0) thin_ft_kernel<true><<<grid, 512>>>(q); else thin_f_kernel<true><<<grid, 256>>>(q); };
    go(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) go();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(8 * 4096);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    // stamps of the last launch: per workgroup (first 4096) 8 values: start, after prologue, after loads issued, after first wait, after mfma, after reduce, end, xcc/cu id
    unsigned long long t0 = ~0ull, t1 = 0; int nw = 0;
    double seg[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 4096; ++w) {
      const unsigned long long* s = &hs[8 * w];
      if (!s[0]) continue;
      ++nw; t0 = std::min(t0, s[0]); t1 = std::max(t1, s[6]);
      for (int k = 0; k < 6; ++k) seg[k] += (double)(s[k + 1] - s[k]);
    }
    printf("%s: live %d rank %d: %.2f us per launch (back to back); %d working workgroups; last launch first start -> last end %.2f us (100 MHz clock)\n",
           which == 0 ? "thin_ft" : "thin_f ", live, rk, ms * 1e3 / 20, nw, (t1 - t0) / 100.0);
    printf("   mean per workgroup [us]: prologue %.2f  issue loads %.2f  first wait %.2f  multiply %.2f  reduce %.2f  store %.2f\n", seg[0] / nw / 100, seg[1] / nw / 100,
           seg[2] / nw / 100, seg[3] / nw / 100, seg[4] / nw / 100, seg[5] / nw / 100);
    // start-time spread
    std::vector<double> st;
    for (int w = 0; w < 4096; ++w) if (hs[8 * w]) st.push_back((hs[8 * w] - t0) / 100.0);
    std::sort(st.begin(), st.end());
    printf("   workgroup start offsets [us]: median %.2f  p90 %.2f  max %.2f\n", st[st.size() / 2], st[st.size() * 9 / 10], st.back());
    // placement: (xcc, se, cu) of every working workgroup with its start offset, first 24 by start time
    std::vector<std::pair<double, unsigned long long>> pl;
    for (int w = 0; w < 4096; ++w) if (hs[8 * w]) pl.push_back({(hs[8 * w] - t0) / 100.0, hs[8 * w + 7]});
    std::sort(pl.begin(), pl.end());
    std::vector<int> per_cu(8 * 64, 0);
    for (auto& e : pl) { const unsigned hw = (unsigned)e.second, xcc = (unsigned)(e.second >> 32) & 15; const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; per_cu[(xcc & 7) * 64 + se * 16 + cu * 1 + sh * 0] += 1; }
    int used = 0, mx = 0; for (int v : per_cu) { used += v > 0; mx = std::max(mx, v); }
    printf("   distinct (xcc, se, cu) used by working workgroups: %d, most on one: %d\n", used, mx);
    for (size_t i = 0; i < pl.size(); i += std::max<size_t>(1, pl.size() / 16)) { const unsigned hw = (unsigned)pl[i].second; printf("      start %.2f us xcc %u se %u sh %u cu %u wave %u simd %u\n", pl[i].first, (unsigned)(pl[i].second >> 32) & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, hw & 15, (hw >> 4) & 3); }
    hipMemset(stamps, 0, 8 * 4096 * 8);
  }
  return 0;
}
