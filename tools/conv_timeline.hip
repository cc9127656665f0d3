// Diagnostic: where does a block of the LDS-DMA conv kernel spend its time?
// Builds conv_igemm_dma.hip with -DNBC_STAMPS (thread 0 of every block stamps the 100 MHz wall clock
// at phase boundaries) and runs ONE layer shape on random bf16 data.  Not part of the library.
//   tools/_bin/conv_timeline Hi Wi Ci Co K dil res tile [streams [precision]]
// NBC_WARM=n: n untimed launches first (short layers need ~2000 for the clock to settle at what a forward sees);
// NBC_DUMP=1: one line per block (XCD, SE, CU, start, phases).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "nbc_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static unsigned short rnd_bf16(unsigned& s, float amp) {
  s = s * 1664525u + 1013904223u;
  float f = (((s >> 8) & 0xffff) / 65535.0f - 0.5f) * 2.f * amp;
  unsigned u; __builtin_memcpy(&u, &f, 4);
  return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
  if (argc < 9) { std::fprintf(stderr, "usage: %s Hi Wi Ci Co K dil res tile\n", argv[0]); return 2; }
  const int Hi = atoi(argv[1]), Wi = atoi(argv[2]), Ci = atoi(argv[3]), Co = atoi(argv[4]);
  const int K = atoi(argv[5]), dil = atoi(argv[6]), res = atoi(argv[7]), tile = atoi(argv[8]);
  if (Hi < 1 || Wi < 1 || Ci % 64 || Co % 64 || (K != 1 && K != 3) || tile < 0 || tile >= nbc::CONV_TILE_COUNT) return 2;
  const int prec = argc > 10 ? atoi(argv[10]) : 1;          // 1 = bf16 (default), 0 = f32, 2 = f16x2
  const int eb = prec == 1 ? 2 : 4;
  const int M = Hi * Wi, ksteps = K * K * Ci * eb / 128;
  const size_t xb = (size_t)M * Ci * eb, wb = (size_t)Co * ksteps * 128, yb = (size_t)M * Co * eb;
  std::vector<unsigned short> hx(xb / 2), hw(wb / 2), hr(yb / 2);
  unsigned seed = 12345u;
  if (prec == 1) {
    for (auto& v : hx) v = rnd_bf16(seed, 1.f);
    for (auto& v : hw) v = rnd_bf16(seed, 0.05f);
    for (auto& v : hr) v = rnd_bf16(seed, 1.f);
  } else {      // f32: the same values, as (bf16 bits << 16) with random low mantissa halves
    auto fill = [&](std::vector<unsigned short>& v, float amp) {
      for (size_t i = 0; i + 1 < v.size(); i += 2) { v[i] = rnd_bf16(seed, 1.f); v[i + 1] = rnd_bf16(seed, amp); }
    };
    fill(hx, 1.f); fill(hw, 0.05f); fill(hr, 1.f);
  }
  std::vector<float> hs(Co, 1.0f), hb(Co, 0.01f);
  void *dx, *dw, *dr, *dy, *dz; float *ds, *db; unsigned long long* dst;
  CK(hipMalloc(&dx, xb)); CK(hipMalloc(&dw, wb)); CK(hipMalloc(&dr, yb)); CK(hipMalloc(&dy, yb)); CK(hipMalloc(&dz, 256));
  CK(hipMalloc(&ds, Co * 4)); CK(hipMalloc(&db, Co * 4));
  CK(hipMemcpy(dx, hx.data(), xb, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), wb, hipMemcpyHostToDevice));
  CK(hipMemcpy(dr, hr.data(), yb, hipMemcpyHostToDevice)); CK(hipMemset(dz, 0, 256));
  CK(hipMemcpy(ds, hs.data(), Co * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), Co * 4, hipMemcpyHostToDevice));
  const int nblk = ((M + nbc::conv_tile_rows(tile) - 1) / nbc::conv_tile_rows(tile)) * (Co / nbc::conv_tile_cols(tile));
  CK(hipMalloc(&dst, (size_t)nblk * 512)); CK(hipMemset(dst, 0, (size_t)nblk * 512));
  nbc::ConvArgs a{};
  a.x = dx; a.w = dw; a.scale = ds; a.shift = db; a.res = res ? dr : nullptr; a.y = dy;
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.N = 1; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Hi; a.Wo = Wi; a.Co = Co;
  a.KH = a.KW = K; a.stride = 1; a.pad = dil * (K / 2); a.dil = dil; a.M = M; a.ksteps = ksteps; a.relu = 1; a.stem = 0;
  a.wo_shift = -1; a.hw_shift = -1;
  for (int s = 0; s < 30; ++s) { if ((1 << s) == Wi) a.wo_shift = s; if ((1 << s) == Hi * Wi) a.hw_shift = s; }
  a.stamps = nullptr;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < (getenv("NBC_WARM") ? atoi(getenv("NBC_WARM")) : 20); ++i) CK(nbc::launch_conv_dma(a, prec, tile, st));
  CK(hipEventRecord(e0, st));
  const int reps = 50;
  for (int i = 0; i < reps; ++i) CK(nbc::launch_conv_dma(a, prec, tile, st));
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, flops = 2.0 * M * Co * (double)K * K * Ci;
  const int nstreams = argc > 9 ? atoi(argv[9]) : 1;
  if (nstreams > 1) {     // the same launch on several streams at once: do blocks of different launches share CUs?
    std::vector<hipStream_t> ss(nstreams);
    std::vector<void*> ys(nstreams);
    for (int k = 0; k < nstreams; ++k) { CK(hipStreamCreate(&ss[k])); CK(hipMalloc(&ys[k], yb)); }
    CK(hipDeviceSynchronize());
    hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    std::vector<hipEvent_t> done(nstreams);
    CK(hipEventRecord(f0, st));
    for (int k = 0; k < nstreams; ++k) CK(hipStreamWaitEvent(ss[k], f0, 0));
    for (int i = 0; i < reps; ++i)
      for (int k = 0; k < nstreams; ++k) { nbc::ConvArgs b = a; b.y = ys[k]; CK(nbc::launch_conv_dma(b, prec, tile, ss[k])); }
    for (int k = 0; k < nstreams; ++k) { CK(hipEventCreate(&done[k])); CK(hipEventRecord(done[k], ss[k])); CK(hipStreamWaitEvent(st, done[k], 0)); }
    CK(hipEventRecord(f1, st)); CK(hipEventSynchronize(f1));
    float ms2 = 0; CK(hipEventElapsedTime(&ms2, f0, f1));
    std::printf("%d streams: %.1f us per launch aggregate (%.0f TF) vs %.1f us alone\n", nstreams, ms2 * 1e3 / (reps * nstreams),
                flops / (ms2 * 1e3 / (reps * nstreams)) * 1e-6, us);
  }
  a.stamps = dst;
  CK(nbc::launch_conv_dma(a, prec, tile, st));   // stamped launches: the last one is read
  CK(nbc::launch_conv_dma(a, prec, tile, st));
  CK(hipStreamSynchronize(st));
  std::vector<unsigned long long> h((size_t)nblk * 64);
  CK(hipMemcpy(h.data(), dst, (size_t)nblk * 512, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b * 64]); t1 = std::max(t1, h[b * 64 + 6]); }
  std::printf("shape %dx%d Ci %d Co %d k%d d%d res %d %s tile %d (%dx%d): %d blocks, %d K-steps | %.1f us/launch back-to-back, %.0f TF | stamped span %.1f us\n",
              Hi, Wi, Ci, Co, K, dil, res, prec == 1 ? "bf16" : prec == 2 ? "f16x2" : "f32", tile, nbc::conv_tile_rows(tile), nbc::conv_tile_cols(tile), nblk, ksteps, us,
              flops / us * 1e-6, (t1 - t0) * 0.01);
  auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
  const char* names[10] = {"start (since first block)", "address set-up done", "prologue issued", "first K-step landed",
                           "last K-step landed", "MFMAs done", "slab 0 in scratch", "slab 0 stores issued", "stores issued", "stores acked"};
  const int slot[10] = {0, 8, 1, 2, 3, 4, 9, 10, 5, 6};
  for (int grp = 0; grp < 2; ++grp) {      // blocks of the first round (start < 1 us) and the later ones
    size_t cnt = 0;
    for (int b = 0; b < nblk; ++b) cnt += ((h[b * 64] - t0) * 0.01 < 1.0) == (grp == 0);
    if (!cnt) continue;
    std::printf(" %s blocks (%zu):\n", grp == 0 ? "first-round" : "later-round", cnt);
    for (int i = 0; i < 10; ++i) {
      std::vector<double> v;
      for (int b = 0; b < nblk; ++b) {
        if (((h[b * 64] - t0) * 0.01 < 1.0) != (grp == 0)) continue;
        unsigned long long s = h[b * 64 + slot[i]];
        if (slot[i] == 2 && s == 0) s = h[b * 64 + 3];
        v.push_back(i == 0 ? (s - t0) * 0.01 : (s - h[b * 64]) * 0.01);
      }
      std::printf("  %-28s p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us%s\n", names[i], pct(v, 0.1), pct(v, 0.5), pct(v, 0.9),
                  pct(v, 1.0), i == 0 ? "" : "  (since block start)");
    }
  }
  {  // shader clock held inside the K loop: core cycles (s_memtime) per 100 MHz wall tick
    std::vector<double> ghz, util, fvm, fbar;
    double wvm[16] = {0}, wbar[16] = {0}; int wn = 0;
    // MAC per K-step of the tile / (4 SIMDs x MAC per clock per SIMD: 512 bf16, 32 f32)
    // (f16x2: 32 channels per K-step, three f16 products each)
    const double mfma_cyc_per_step = (double)nbc::conv_tile_rows(tile) * nbc::conv_tile_cols(tile) *
                                     (prec == 1 ? 64.0 / (4 * 512.0) : prec == 2 ? 3.0 * 32.0 / (4 * 512.0) : 32.0 / (4 * 32.0));
    for (int b = 0; b < nblk; ++b) {
      const double dt = (double)(h[b * 64 + 3] - h[b * 64 + 2]), dc = (double)(h[b * 64 + 12] - h[b * 64 + 11]);
      if (h[b * 64 + 2] == 0 || dt < 50) continue;
      ghz.push_back(dc / dt * 0.1);
      util.push_back(mfma_cyc_per_step * (ksteps - 1) / dc);
      fvm.push_back((double)h[b * 64 + 16] / dc);
      fbar.push_back((double)h[b * 64 + 32] / dc);
      for (int w = 0; w < 16; ++w) { wvm[w] += (double)h[b * 64 + 16 + w] / dc; wbar[w] += (double)h[b * 64 + 32 + w] / dc; }
      ++wn;
    }
    if (!ghz.empty())
      std::printf("  K loop: shader clock p50 %.2f GHz (p10 %.2f, p90 %.2f); MFMA-busy share of its cycles p50 %.2f; wave 0 waits: vmcnt %.2f, barrier %.2f of the loop\n",
                  pct(ghz, 0.5), pct(ghz, 0.1), pct(ghz, 0.9), pct(util, 0.5), pct(fvm, 0.5), pct(fbar, 0.5));
    if (wn) {
      std::printf("  per wave, mean share of the loop in vmcnt wait | barrier wait:");
      for (int w = 0; w < 16; ++w) if (wvm[w] + wbar[w] > 0) std::printf("  w%d %.2f|%.2f", w, wvm[w] / wn, wbar[w] / wn);
      std::printf("\n");
    }
  }
  // residency: blocks per (xcc, se, cu) and how many rounds a CU ran
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
  for (int b = 0; b < nblk; ++b) {
    const unsigned long long id = h[b * 64 + 7];
    const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    per_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back({h[b * 64], h[b * 64 + 6]});
  }
  {  // which first-round blocks share a CU: histogram of the blockIdx distance of the first two blocks of every CU
    std::map<unsigned long long, std::vector<int>> first;
    for (int b = 0; b < nblk; ++b) {
      if ((h[b * 64] - t0) * 0.01 >= 1.0) continue;
      const unsigned long long id = h[b * 64 + 7];
      const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
      first[((unsigned long long)xcc << 16) | (hw & 0xff00)].push_back(b);
    }
    std::map<int, int> hist;
    for (auto& kv : first) if (kv.second.size() >= 2) ++hist[kv.second[1] - kv.second[0]];
    std::printf("  first-round co-residents, blockIdx distance: ");
    for (auto& kv : hist) std::printf(" %d x%d", kv.first, kv.second);
    std::printf("\n");
  }
  size_t maxb = 0, maxc = 0;
  for (auto& kv : per_cu) {
    maxb = std::max(maxb, kv.second.size());
    size_t conc = 0;
    for (auto& x : kv.second) { size_t c = 0; for (auto& y : kv.second) if (y.first <= x.first && y.second > x.first) ++c; conc = std::max(conc, c); }
    maxc = std::max(maxc, conc);
  }
  std::printf("  CUs used %zu, blocks per CU max %zu, co-resident per CU max %zu\n", per_cu.size(), maxb, maxc);
  if (getenv("NBC_DUMP")) {      // one line per block: where it ran and how long
    for (int b = 0; b < nblk; ++b) {
      const unsigned long long id = h[b * 64 + 7];
      const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
      std::printf("blk %4d xcc %u se %u sh %u cu %2u simd0 %u start %7.2f first %6.2f kloop %7.2f total %7.2f\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf,
                  (hw >> 4) & 3, (h[b * 64] - t0) * 0.01, (h[b * 64 + 2] - h[b * 64]) * 0.01, (h[b * 64 + 3] - h[b * 64 + 2]) * 0.01, (h[b * 64 + 6] - h[b * 64]) * 0.01);
    }
  }
  return 0;
}
