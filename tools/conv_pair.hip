// Diagnostic: do a compute-bound and a memory-bound convolution overlap when they run on two streams?
//   tools/_bin/conv_pair  "Hi Wi Ci Co K dil res tile"  "Hi Wi Ci Co K dil res tile"  [reps]
// bf16, random data.  Prints each layer's time alone (back to back on one stream) and the time of `reps` launches of
// each on two streams at once.  Blocks of the two kernels can share a CU only if both tiles leave room (LDS <= 80 KiB).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <vector>

#include "nbc_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Layer { nbc::ConvArgs a; int tile; double flops; size_t yb; };

static int make(const char* spec, Layer& L) {
  int Hi, Wi, Ci, Co, K, dil, res, tile;
  std::istringstream is(spec);
  if (!(is >> Hi >> Wi >> Ci >> Co >> K >> dil >> res >> tile)) return 1;
  const int M = Hi * Wi, ksteps = K * K * Ci * 2 / 128;
  const size_t xb = (size_t)M * Ci * 2, wb = (size_t)Co * ksteps * 128, yb = (size_t)M * Co * 2;
  std::vector<unsigned short> hx(xb / 2), hw(wb / 2), hr(yb / 2);
  unsigned s = 12345u;
  auto rnd = [&](float amp) { s = s * 1664525u + 1013904223u; float f = (((s >> 8) & 0xffff) / 65535.0f - 0.5f) * 2.f * amp; unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
  for (auto& v : hx) v = rnd(1.f);
  for (auto& v : hw) v = rnd(0.05f);
  for (auto& v : hr) v = rnd(1.f);
  std::vector<float> hs(Co, 1.0f), hb(Co, 0.01f);
  void *dx, *dw, *dr, *dy; float *ds, *db;
  CK(hipMalloc(&dx, xb)); CK(hipMalloc(&dw, wb)); CK(hipMalloc(&dr, yb)); CK(hipMalloc(&dy, yb)); CK(hipMalloc(&ds, Co * 4)); CK(hipMalloc(&db, Co * 4));
  CK(hipMemcpy(dx, hx.data(), xb, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), wb, hipMemcpyHostToDevice));
  CK(hipMemcpy(dr, hr.data(), yb, hipMemcpyHostToDevice));
  CK(hipMemcpy(ds, hs.data(), Co * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), Co * 4, hipMemcpyHostToDevice));
  nbc::ConvArgs a{};
  a.x = dx; a.w = dw; a.scale = ds; a.shift = db; a.res = res ? dr : nullptr; a.y = dy;
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.N = 1; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Hi; a.Wo = Wi; a.Co = Co;
  a.KH = a.KW = K; a.stride = 1; a.pad = dil * (K / 2); a.dil = dil; a.M = M; a.ksteps = ksteps; a.relu = 1; a.stem = 0;
  a.wo_shift = -1; a.hw_shift = -1;
  for (int q = 0; q < 30; ++q) { if ((1 << q) == Wi) a.wo_shift = q; if ((1 << q) == Hi * Wi) a.hw_shift = q; }
  L.a = a; L.tile = tile; L.flops = 2.0 * M * Co * (double)K * K * Ci; L.yb = yb;
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s \"layer A\" \"layer B\" [reps]\n", argv[0]); return 2; }
  const int reps = argc > 3 ? atoi(argv[3]) : 40;
  Layer A, B;
  if (make(argv[1], A) || make(argv[2], B)) return 2;
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  auto alone = [&](Layer& L, float* us) -> int {
    for (int i = 0; i < 300; ++i) CK(nbc::launch_conv_dma(L.a, 1, L.tile, s1));
    CK(hipEventRecord(e0, s1));
    for (int i = 0; i < reps; ++i) CK(nbc::launch_conv_dma(L.a, 1, L.tile, s1));
    CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); *us = ms * 1e3f / reps; return 0;
  };
  float ua = 0, ub = 0;
  if (alone(A, &ua) || alone(B, &ub)) return 1;
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0));
  for (int i = 0; i < reps; ++i) { CK(nbc::launch_conv_dma(A.a, 1, A.tile, s1)); CK(nbc::launch_conv_dma(B.a, 1, B.tile, s2)); }
  CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s1, e2, 0));
  CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const float both = ms * 1e3f / reps;
  std::printf("A tile %d: %.1f us alone (%.0f TF) | B tile %d: %.1f us alone (%.0f TF) | one after the other %.1f us | on two streams %.1f us per pair (%.2f of the sum)\n",
              A.tile, ua, A.flops / ua * 1e-6, B.tile, ub, B.flops / ub * 1e-6, ua + ub, both, both / (ua + ub));
  return 0;
}
