// Probe: what keeps a SIMD's matrix pipe from running v_mfma_f32_32x32x2_f32 back to back?
// One block per CU, 4 or 8 waves (1 or 2 per SIMD); a loop of "K-steps" of 64 MFMAs on 4 accumulator tiles
// (the f32 conv kernel's wave tile), optionally with the things the conv kernel has around them:
//   bit 0  one s_barrier per K-step            bit 1  16 ds_read_b128 per K-step feeding the MFMAs
//   bit 2  64 v_add_f32 per K-step (two-level sum)   bit 3  6 LDS-DMAs (1 KiB each) per K-step
//   bit 5  accumulators in AGPRs (inline-asm MFMA) instead of arch VGPRs
//   bit 6  accumulators in AGPRs, compiler-scheduled (builtin MFMA; an inline-asm AGPR operand elsewhere in the
//          kernel makes hipcc select the AGPR form)
// Prints cycles per MFMA per SIMD (64 = the pipe's rate) for every combination asked for.
//   tools/_bin/mfma_f32_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MASK>
__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lds = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = src[i];
  __syncthreads();
  f32x16 acc[4], accI[4];
  if (MASK & 64) { float z = 0.f; asm volatile("; an AGPR operand: hipcc then selects the AGPR form of every MFMA %0" ::"a"(z)); }
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) { acc[t][e] = 0.f; accI[t][e] = 0.f; }
  float4 a[2][2], b[2][2];
  for (int q = 0; q < 2; ++q)
    for (int k = 0; k < 2; ++k) { a[q][k] = make_float4(lane * 0.001f, 1.f, 0.5f, 0.25f); b[q][k] = make_float4(1.f, lane * 0.002f, 0.5f, 2.f); }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned lds_base = (unsigned)(size_t)(lds_u8*)smem + 65536u + (unsigned)wave * 8192u;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if ((MASK & 1) && ks == 2) __builtin_amdgcn_s_barrier();
      const float4* p = reinterpret_cast<const float4*>(smem + ((wave * 64 + lane) * 16 + ((s + ks) & 7) * 8192) % 65536);
      if ((MASK & 2) && !(MASK & 16)) {
        a[(ks + 1) & 1][0] = p[0]; a[(ks + 1) & 1][1] = p[64]; b[(ks + 1) & 1][0] = p[128]; b[(ks + 1) & 1][1] = p[192];
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float4 wv = a[ks & 1][n >> 1], pv = b[ks & 1][n & 1];
        const float we[4] = {wv.x, wv.y, wv.z, wv.w}, pe[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (MASK & 32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(we[e]), "v"(pe[e]));      // accumulators in AGPRs
          else if (MASK & 4) accI[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(we[e], pe[e], (ks == 0 && e == 0) ? zero : accI[n], 0, 0, 0);
          else acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(we[e], pe[e], acc[n], 0, 0, 0);
          if ((MASK & 2) && (MASK & 16) && e == 0) {      // spread: one fragment read behind the first MFMA of each tile
            if (n == 0) a[(ks + 1) & 1][0] = p[0];
            if (n == 1) a[(ks + 1) & 1][1] = p[64];
            if (n == 2) b[(ks + 1) & 1][0] = p[128];
            if (n == 3) b[(ks + 1) & 1][1] = p[192];
          }
          if ((MASK & 4) && ((ks == 0 && n == 0) || (ks == 3 && n >= 1))) {
            const int tl = ks == 0 ? 3 : n - 1;
            const int lo = (MASK & 16) ? (e == 0 ? 0 : e == 1 ? 0 : e == 2 ? 6 : 11) : 4 * e;
            const int hi = (MASK & 16) ? (e == 0 ? 0 : e == 1 ? 6 : e == 2 ? 11 : 16) : 4 * e + 4;
#pragma unroll
            for (int q = lo; q < hi; ++q) {
              float v = acc[tl][q] + accI[tl][q];
              asm volatile("" : "+v"(v));
              acc[tl][q] = v;
            }
          }
          if (MASK & 16) __builtin_amdgcn_sched_barrier(0);
        }
        if ((MASK & 8) && ks >= 2 && (ks == 3) == (wave >= 4)) {
          const int nd = n == 0 || n == 2 ? 2 : 1;
          for (int d = 0; d < nd; ++d) {
            unsigned keep;
            const float* g = src + ((s * 6 + n * 2 + d) % 16) * 1024 + lane * 4;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(lds_base + (unsigned)((n * 2 + d) % 8) * 1024u) : "memory");
          }
          if (MASK & 16) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (MASK & 8) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) sum += acc[t][e] + ((MASK & 4) ? accI[t][e] : 0.f);
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int MASK>
int run(int waves, const float* src, float* out, unsigned long long* cyc, int steps) {
  const int blocks = 256;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MASK>, dim3(blocks), dim3(waves * 64), 140 * 1024, 0, src, out, cyc, steps);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= blocks;
  const double per = mean / ((double)steps * 64.0 * (waves / 4));     // cycles per MFMA per SIMD
  std::printf("  %d waves/SIMD  barrier %d  ds_read %d  adds %d  dma %d  spread %d  agpr %d : %.2f cycles per MFMA per SIMD  (pipe busy %.3f)\n", waves / 4, MASK & 1,
              (MASK >> 1) & 1, (MASK >> 2) & 1, (MASK >> 3) & 1, (MASK >> 4) & 1, (MASK >> 5) & 3, per, 64.0 / per);
  return 0;
}

int main() {
  float *src, *out;
  unsigned long long* cyc;
  CK(hipMalloc(&src, 1 << 20));
  CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMalloc(&cyc, 256 * 8));
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice));
  const int steps = 400;
  for (int waves = 4; waves <= 8; waves += 4) {
    if (run<1>(waves, src, out, cyc, steps)) return 1;
    if (run<3>(waves, src, out, cyc, steps)) return 1;
    if (run<3 + 16>(waves, src, out, cyc, steps)) return 1;
    if (run<7>(waves, src, out, cyc, steps)) return 1;
    if (run<7 + 16>(waves, src, out, cyc, steps)) return 1;
    if (run<11>(waves, src, out, cyc, steps)) return 1;
    if (run<11 + 16>(waves, src, out, cyc, steps)) return 1;
    if (run<15>(waves, src, out, cyc, steps)) return 1;
    if (run<15 + 16>(waves, src, out, cyc, steps)) return 1;
    if (run<9>(waves, src, out, cyc, steps)) return 1;
    if (run<1 + 32>(waves, src, out, cyc, steps)) return 1;
    if (run<3 + 32>(waves, src, out, cyc, steps)) return 1;
    if (run<3 + 16 + 32>(waves, src, out, cyc, steps)) return 1;
    if (run<9 + 32>(waves, src, out, cyc, steps)) return 1;
    if (run<11 + 32>(waves, src, out, cyc, steps)) return 1;
    if (run<1 + 64>(waves, src, out, cyc, steps)) return 1;
    if (run<3 + 64>(waves, src, out, cyc, steps)) return 1;
    if (run<9 + 64>(waves, src, out, cyc, steps)) return 1;
    if (run<11 + 64>(waves, src, out, cyc, steps)) return 1;
    if (run<7 + 64>(waves, src, out, cyc, steps)) return 1;
    if (run<15 + 64>(waves, src, out, cyc, steps)) return 1;
  }
  return 0;
}
