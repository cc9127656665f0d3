// Probe: f32-grade dot products from the 16-bit matrix pipe.
//
// An f32 operand splits into 16-bit pieces; the products of the pieces that matter are exact in f32 and are summed
// by the f16 / bf16 MFMA, which runs at 16x the rate of v_mfma_f32_32x32x2_f32:
//   bf16x3   x = b0 + b1 + b2 (8 significant bits each, exact to 24 bits); 6 products b0b0 b0b1 b1b0 b1b1 b0b2 b2b0
//            (dropped: 2^-24 relative and below)                                       -> 16/6 = 2.7x the f32 rate
//   f16x2    x ~ h0 + h1 * 2^-11 (11 significant bits each: 2^-24 relative, like one f32 rounding), h1 kept scaled by
//            2^11 so that it stays a normal f16; 3 products h0h0 (h0h1 + h1h0) * 2^-11 (dropped: h1h1, 2^-24 relative)
//                                                                                      -> 16/3 = 5.3x the f32 rate
// Part 1 (accuracy): 32x32 output tiles of X[M][K] . W[N][K]^T at the K of the network's layers (18 432 = the head
// conv), every method against a float64 sum, next to the library's two-level f32 chain (32 products per K-step
// through v_mfma_f32_32x32x2_f32, flushed to a second accumulator every 8 K-steps: conv_igemm_dma.hip).
// Part 2 (rate): a K loop of one wave tile (64 pixels x 64 channels, fragments read from LDS with ds_read_b128, the
// LDS-DMAs of a 256x128 tile's refill beside them) on two waves per SIMD of every CU: cycles per K-step, shader
// clock, and the f32-equivalent TFLOP/s that would be (2 x 64 x 64 x 64 FLOPs per wave and 64-channel K-step).
//   tools/_bin/split_probe            (built by tools/build_tools.sh)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum Method { M_F32_2LEVEL = 0, M_F32_CHAIN, M_BF16X3, M_BF16X3_ONEACC, M_F16X2, M_F16X2_FLUSH, M_F16X2_4PROD, M_BF16X2, M_F16X1, M_COUNT };
static const char* kMethodName[M_COUNT] = {
    "f32 MFMA, two-level sum (the library today)", "f32 MFMA, one chain", "bf16x3, 6 products, big/small accumulators",
    "bf16x3, 6 products, one accumulator", "f16x2, 3 products, big/small accumulators", "f16x2, 3 products, big sum flushed every 256",
    "f16x2, 4 products (with h1h1)", "bf16x2, 3 products (16-bit operands: not f32 grade)", "f16 alone (1 product: the half-precision floor)"};

__device__ __forceinline__ void split_bf16x3(float x, __bf16& b0, __bf16& b1, __bf16& b2) {
  b0 = (__bf16)x;
  const float r1 = x - (float)b0;      // exact
  b1 = (__bf16)r1;
  const float r2 = r1 - (float)b1;     // exact
  b2 = (__bf16)r2;
}
constexpr float kH1Scale = 2048.f;     // 2^11: keeps the low piece a normal f16 wherever the value itself is one
__device__ __forceinline__ void split_f16x2(float x, _Float16& h0, _Float16& h1) {
  h0 = (_Float16)x;
  h1 = (_Float16)((x - (float)h0) * kH1Scale);
}

// One wave = one 32x32 tile of D = X . W^T; lane (r = lane & 31, h = lane >> 5).
template <int METHOD>
__global__ __launch_bounds__(64) void dot_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ D,
                                                 int K, int N) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int tm = blockIdx.x, tn = blockIdx.y;
  const float* xrow = X + (size_t)(tm * 32 + r) * K;
  const float* wrow = W + (size_t)(tn * 32 + r) * K;
  f32x16 acc, accS;
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accS[e] = 0.f; }
  f32x16 accI = acc;
  if constexpr (METHOD == M_F32_2LEVEL || METHOD == M_F32_CHAIN) {
    // K-steps of 32: ks 0..3, chunk 2ks+h of four consecutive k, four MFMAs (one per element): conv_igemm_dma.hip
    for (int t = 0; t < K / 32; ++t) {
      if (METHOD == M_F32_2LEVEL && t > 0 && (t & 7) == 0) { acc += accI; for (int e = 0; e < 16; ++e) accI[e] = 0.f; }
      for (int ks = 0; ks < 4; ++ks) {
        const float4 xv = *reinterpret_cast<const float4*>(xrow + 32 * t + 4 * (2 * ks + h));
        const float4 wv = *reinterpret_cast<const float4*>(wrow + 32 * t + 4 * (2 * ks + h));
        accI = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, accI, 0, 0, 0);
        accI = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, accI, 0, 0, 0);
        accI = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, accI, 0, 0, 0);
        accI = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, accI, 0, 0, 0);
      }
    }
    acc += accI;
  } else {
    // chunks of 16 k: lane (r, h) holds k = 16c + 8h .. + 7 of its row (the 32x32x16 operand layout)
    for (int c = 0; c < K / 16; ++c) {
      float xv[8], wv[8];
      for (int q = 0; q < 2; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(xrow + 16 * c + 8 * h + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(wrow + 16 * c + 8 * h + 4 * q);
        xv[4 * q] = a.x; xv[4 * q + 1] = a.y; xv[4 * q + 2] = a.z; xv[4 * q + 3] = a.w;
        wv[4 * q] = b.x; wv[4 * q + 1] = b.y; wv[4 * q + 2] = b.z; wv[4 * q + 3] = b.w;
      }
      if constexpr (METHOD == M_BF16X3 || METHOD == M_BF16X3_ONEACC || METHOD == M_BF16X2) {
        bf16x8 x0, x1, x2, w0, w1, w2;
        for (int e = 0; e < 8; ++e) {
          __bf16 a, b, c2;
          split_bf16x3(xv[e], a, b, c2); x0[e] = a; x1[e] = b; x2[e] = c2;
          split_bf16x3(wv[e], a, b, c2); w0[e] = a; w1[e] = b; w2[e] = c2;
        }
        if constexpr (METHOD == M_BF16X3) {
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x0, accS, 0, 0, 0);
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x2, accS, 0, 0, 0);
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, accS, 0, 0, 0);
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, accS, 0, 0, 0);
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, accS, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc, 0, 0, 0);
        } else if constexpr (METHOD == M_BF16X3_ONEACC) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc, 0, 0, 0);
        } else {
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, accS, 0, 0, 0);
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, accS, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc, 0, 0, 0);
        }
      } else {
        f16x8 x0, x1, w0, w1;
        for (int e = 0; e < 8; ++e) {
          _Float16 a, b;
          split_f16x2(xv[e], a, b); x0[e] = a; x1[e] = b;
          split_f16x2(wv[e], a, b); w0[e] = a; w1[e] = b;
        }
        if constexpr (METHOD == M_F16X2_FLUSH) {
          if (c > 0 && (c & 15) == 0) { acc += accI; for (int e = 0; e < 16; ++e) accI[e] = 0.f; }
          accI = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, accI, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, acc, 0, 0, 0);
        }
        if constexpr (METHOD != M_F16X1) {
          accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x0, accS, 0, 0, 0);     // carries 2^11
          accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x1, accS, 0, 0, 0);
        }
        if constexpr (METHOD == M_F16X2_4PROD) accI = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, accI, 0, 0, 0);   // carries 2^22
      }
    }
    if constexpr (METHOD == M_F16X2_FLUSH) acc += accI;
    if constexpr (METHOD == M_F16X2_4PROD) accS += accI * (1.f / kH1Scale);
    if constexpr (METHOD == M_F16X2 || METHOD == M_F16X2_FLUSH || METHOD == M_F16X2_4PROD) acc += accS * (1.f / kH1Scale);
    if constexpr (METHOD == M_BF16X3 || METHOD == M_BF16X2) acc += accS;
  }
  // 32x32 D layout: lane (r, h) holds pixel column r and channel rows 8g + 4h + e (register 4g + e)
  for (int g = 0; g < 4; ++g)
    for (int e = 0; e < 4; ++e)
      D[(size_t)(tm * 32 + r) * N + tn * 32 + 8 * g + 4 * h + e] = acc[4 * g + e];
}

template <int METHOD>
void launch_dot(const float* X, const float* W, float* D, int M, int N, int K) {
  hipLaunchKernelGGL(dot_kernel<METHOD>, dim3(M / 32, N / 32), dim3(64), 0, 0, X, W, D, K, N);
}

// ---- part 2: the K loop's rate -------------------------------------------------------------------------------
// MODE 0: f16x2 (12 MFMA 32x32x16 per 32x32 tile and K-step: h0h0 into the big sums, h0h1 + h1h0 into the small ones)
// MODE 1: bf16x3 (24 per tile)          MODE 2: bf16 alone (4 per tile: the throughput mode's K-step, for the clock)
// MODE 3: f16x2 on v_mfma_f32_16x16x32_f16 (same products; the chip holds a higher clock on the 16x16 form)
// WITH bit 0: fragments from LDS (ds_read_b128), bit 1: NDMA LDS-DMAs per wave and K-step, bit 2: one barrier per K-step
template <int MODE, int WITH, int NDMA>
__global__ __launch_bounds__(512, 2) void rate_kernel(const float* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lds = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 24576; i += blockDim.x) lds[i] = src[i % 16384] * 1e-3f;
  __syncthreads();
  constexpr int PLANES = MODE == 1 ? 3 : (MODE == 2 ? 1 : 2);
  constexpr int NACC = MODE == 2 ? 1 : 2;
  f32x16 acc[NACC][4];
  f32x4 acc16[MODE == 3 ? 2 : 1][MODE == 3 ? 16 : 1];
  for (int a = 0; a < NACC; ++a)
    for (int t = 0; t < 4; ++t)
      for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.f;
  if (MODE == 3)
    for (int a = 0; a < 2; ++a)
      for (int t = 0; t < 16; ++t)
        for (int e = 0; e < 4; ++e) acc16[a][t][e] = 0.f;
  uint4 xa[2][PLANES][2], wa[2][PLANES][2];       // [buffer][plane][row block]
  for (int b = 0; b < 2; ++b)
    for (int p = 0; p < PLANES; ++p)
      for (int q = 0; q < 2; ++q) { xa[b][p][q] = make_uint4(0x3c003c00u + lane, 0x3c003c00u, 0x38003800u, 0x3c003c00u); wa[b][p][q] = make_uint4(0x3c003c00u, 0x38003800u + lane, 0x3c003c00u, 0x3c003c00u); }
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned lds_base = (unsigned)(size_t)(lds_u8*)smem + 98304u + (unsigned)wave * 4096u;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {              // 16 channels of the 64-channel K-step
      if ((WITH & 4) && ks == 0) __builtin_amdgcn_s_barrier();
      if (WITH & 1) {
        const uint4* p = reinterpret_cast<const uint4*>(smem + ((wave * 64 + lane) * 16 + ((s * 4 + ks) & 15) * 4096) % 65536);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
          for (int q = 0; q < 2; ++q) { xa[(ks + 1) & 1][pl][q] = p[(pl * 4 + q) * 64]; wa[(ks + 1) & 1][pl][q] = p[(pl * 4 + q + 2) * 64 + 1024]; }
      }
      const int cb = ks & 1;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int i = n & 1, j = n >> 1;
        if constexpr (MODE == 0) {
          const f16x8 x0 = __builtin_bit_cast(f16x8, xa[cb][0][i]), x1 = __builtin_bit_cast(f16x8, xa[cb][1][i]);
          const f16x8 w0 = __builtin_bit_cast(f16x8, wa[cb][0][j]), w1 = __builtin_bit_cast(f16x8, wa[cb][1][j]);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x0, acc[1][n], 0, 0, 0);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x1, acc[1][n], 0, 0, 0);
          acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, acc[0][n], 0, 0, 0);
        } else if constexpr (MODE == 1) {
          const bf16x8 x0 = __builtin_bit_cast(bf16x8, xa[cb][0][i]), x1 = __builtin_bit_cast(bf16x8, xa[cb][1][i]), x2 = __builtin_bit_cast(bf16x8, xa[cb][2][i]);
          const bf16x8 w0 = __builtin_bit_cast(bf16x8, wa[cb][0][j]), w1 = __builtin_bit_cast(bf16x8, wa[cb][1][j]), w2 = __builtin_bit_cast(bf16x8, wa[cb][2][j]);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x0, acc[1][n], 0, 0, 0);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x2, acc[1][n], 0, 0, 0);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[1][n], 0, 0, 0);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc[1][n], 0, 0, 0);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc[1][n], 0, 0, 0);
          acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc[0][n], 0, 0, 0);
        } else if constexpr (MODE == 2) {
          const bf16x8 x0 = __builtin_bit_cast(bf16x8, xa[cb][0][i]);
          const bf16x8 w0 = __builtin_bit_cast(bf16x8, wa[cb][0][j]);
          acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc[0][n], 0, 0, 0);
        } else {
          // 16x16x32: a 32x32 tile and 16 channels = two 16-deep halves ... the same MFMA count per FLOP: 4 tiles of
          // 16x16, K 32 per instruction -> per 32x32 tile and 32 channels 4 instructions per product; issued here per
          // 16-channel quarter as 2 per product (the fragment registers stand in for the two halves' data)
          const f16x8 x0 = __builtin_bit_cast(f16x8, xa[cb][0][i]), x1 = __builtin_bit_cast(f16x8, xa[cb][1][i]);
          const f16x8 w0 = __builtin_bit_cast(f16x8, wa[cb][0][j]), w1 = __builtin_bit_cast(f16x8, wa[cb][1][j]);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            acc16[1][n * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x0, acc16[1][n * 4 + q], 0, 0, 0);
            acc16[1][n * 4 + q + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x1, acc16[1][n * 4 + q + 2], 0, 0, 0);
            acc16[0][n * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0, acc16[0][n * 4 + q], 0, 0, 0);
          }
        }
        if ((WITH & 2) && (ks >> 1) == (wave >= 4 ? 1 : 0)) {     // the SIMD's two waves issue their DMAs in different halves
          constexpr int PER = (NDMA + 7) / 8;                      // 8 (ks, n) slots per half
          const int slot = (ks & 1) * 4 + n;
#pragma unroll
          for (int d = 0; d < PER; ++d) {
            if (slot * PER + d < NDMA) {
              const float* g = src + ((s * NDMA + slot * PER + d) % 16) * 1024 + lane * 4;
              asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base + (unsigned)((slot * PER + d) % 4) * 1024u) : "memory");
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int t = 0; t < 4; ++t)
      for (int e = 0; e < 16; ++e) sum += acc[a][t][e];
  if (MODE == 3)
    for (int a = 0; a < 2; ++a)
      for (int t = 0; t < 16; ++t)
        for (int e = 0; e < 4; ++e) sum += acc16[a][t][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = c1 - c0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

template <int MODE, int WITH, int NDMA>
int run_rate(const char* what, const float* src, float* out, unsigned long long* cyc, int mfma_per_step, int cycles_each) {
  const int steps = 4000, blocks = 256;
  auto k = rate_kernel<MODE, WITH, NDMA>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 160 * 1024 - 512, 0, src, out, cyc, steps);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(blocks * 16);
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double c = 0, rt = 0;
  for (int i = 0; i < blocks * 8; ++i) { c += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  c /= blocks * 8; rt /= blocks * 8;
  const double us = rt / 100.0;                       // s_memrealtime: 100 MHz
  // s_memtime counts at a fixed 100 MHz too on this chip family's probes?  report both; GHz from the MFMA count below
  const double per_step_us = us / steps;
  // two waves per SIMD share the pipe: per SIMD and K-step 2 * mfma_per_step * cycles_each pipe cycles
  const double pipe_cycles = 2.0 * mfma_per_step * cycles_each;
  const double flops_equiv = 256.0 * 8 * 2.0 * 64 * 64 * 64 / (per_step_us * 1e-6);      // f32-equivalent FLOP/s, whole chip
  std::printf("%-64s %8.3f us/K-step  pipe cycles/K-step/SIMD %6.0f -> %5.2f GHz if the pipe never waited | %7.1f TF f32-equivalent (s_memtime %.0f ticks/step)\n",
              what, per_step_us, pipe_cycles, pipe_cycles / (per_step_us * 1e3), flops_equiv / 1e12, c / steps);
  return 0;
}

int main() {
  // ---- part 1
  const int M = 128, N = 128;
  std::mt19937_64 rng(12345);
  std::normal_distribution<double> gauss(0.0, 1.0);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  const int Ks[3] = {18432, 2304, 512};
  const char* data_names[3] = {"relu activations x gaussian weights", "wide dynamic range (x 10^U(-3,1), w 10^U(-2,0))", "small magnitudes (x ~ 1e-4: f16 subnormal territory)"};
  for (int data = 0; data < 3; ++data) {
    for (int ki = 0; ki < 3; ++ki) {
      const int K = Ks[ki];
      std::vector<float> X((size_t)M * K), W((size_t)N * K);
      for (auto& v : X) {
        double g = std::max(0.0, gauss(rng));
        if (data == 1) g *= std::pow(10.0, -3.0 + 4.0 * uni(rng));
        if (data == 2) g *= 1e-4;
        v = (float)g;
      }
      for (auto& v : W) {
        double g = gauss(rng) * std::sqrt(2.0 / K);
        if (data == 1) g *= std::pow(10.0, -2.0 + 2.0 * uni(rng));
        v = (float)g;
      }
      std::vector<double> ref((size_t)M * N);
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
          double s = 0;
          const float* x = &X[(size_t)m * K];
          const float* w = &W[(size_t)n * K];
          for (int k = 0; k < K; ++k) s += (double)x[k] * (double)w[k];
          ref[(size_t)m * N + n] = s;
        }
      double lo = 1e300, hi = -1e300, rms = 0;
      for (double v : ref) { lo = std::min(lo, v); hi = std::max(hi, v); rms += v * v; }
      rms = std::sqrt(rms / ref.size());
      float *dX, *dW, *dD;
      CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dD, (size_t)M * N * 4));
      CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
      std::printf("\n%s, K = %d: outputs in [%.3g, %.3g], rms %.3g\n", data_names[data], K, lo, hi, rms);
      for (int meth = 0; meth < M_COUNT; ++meth) {
        switch (meth) {
          case 0: launch_dot<0>(dX, dW, dD, M, N, K); break;
          case 1: launch_dot<1>(dX, dW, dD, M, N, K); break;
          case 2: launch_dot<2>(dX, dW, dD, M, N, K); break;
          case 3: launch_dot<3>(dX, dW, dD, M, N, K); break;
          case 4: launch_dot<4>(dX, dW, dD, M, N, K); break;
          case 5: launch_dot<5>(dX, dW, dD, M, N, K); break;
          case 6: launch_dot<6>(dX, dW, dD, M, N, K); break;
          case 7: launch_dot<7>(dX, dW, dD, M, N, K); break;
          default: launch_dot<8>(dX, dW, dD, M, N, K); break;
        }
        CK(hipDeviceSynchronize());
        std::vector<float> D((size_t)M * N);
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, e2 = 0;
        for (size_t i = 0; i < D.size(); ++i) { const double e = (double)D[i] - ref[i]; emax = std::max(emax, std::fabs(e)); e2 += e * e; }
        // the floor: the float64 sum rounded to f32 once
        std::printf("  %-58s max |err| %.3e (%.2e of the range)  rms err %.3e (%.2e of the outputs' rms)\n", kMethodName[meth], emax,
                    emax / (hi - lo), std::sqrt(e2 / D.size()), std::sqrt(e2 / D.size()) / rms);
      }
      double emax = 0, e2 = 0;
      for (size_t i = 0; i < ref.size(); ++i) { const double e = (double)(float)ref[i] - ref[i]; emax = std::max(emax, std::fabs(e)); e2 += e * e; }
      std::printf("  %-58s max |err| %.3e (%.2e of the range)  rms err %.3e\n", "(floor: the float64 sum rounded to f32 once)", emax, emax / (hi - lo), std::sqrt(e2 / ref.size()));
      (void)hipFree(dX); (void)hipFree(dW); (void)hipFree(dD);
    }
  }

  // ---- part 2
  std::printf("\nK loop of a 64x64 wave tile, two waves per SIMD, 256 CUs (per K-step of 64 channels; f32 MFMA today: 128 x 64 = 8192 pipe cycles per wave)\n");
  float *src, *out;
  unsigned long long* cyc;
  CK(hipMalloc(&src, 65536 * 4)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 16 * 8));
  std::vector<float> hs(65536);
  for (auto& v : hs) v = (float)gauss(rng);
  CK(hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  if (run_rate<2, 0, 0>("bf16 alone, MFMAs only (16 per K-step)", src, out, cyc, 16, 32)) return 1;
  if (run_rate<2, 1, 0>("bf16 alone + fragment reads", src, out, cyc, 16, 32)) return 1;
  if (run_rate<0, 0, 0>("f16x2, MFMAs only (48 x 32x32x16 per K-step)", src, out, cyc, 48, 32)) return 1;
  if (run_rate<0, 1, 0>("f16x2 + 32 ds_read_b128", src, out, cyc, 48, 32)) return 1;
  if (run_rate<0, 3, 12>("f16x2 + reads + 12 LDS-DMAs", src, out, cyc, 48, 32)) return 1;
  if (run_rate<0, 7, 12>("f16x2 + reads + DMAs + barrier", src, out, cyc, 48, 32)) return 1;
  if (run_rate<3, 0, 0>("f16x2 on 16x16x32, MFMAs only (96 per K-step)", src, out, cyc, 96, 16)) return 1;
  if (run_rate<3, 7, 12>("f16x2 on 16x16x32 + reads + DMAs + barrier", src, out, cyc, 96, 16)) return 1;
  if (run_rate<1, 0, 0>("bf16x3, MFMAs only (96 per K-step)", src, out, cyc, 96, 32)) return 1;
  if (run_rate<1, 7, 18>("bf16x3 + 48 reads + 18 DMAs + barrier", src, out, cyc, 96, 32)) return 1;
  return 0;
}
