// NBC_PREC_F16X2 on the device: an f32 value as two f16 pieces (include/nbc.h).
//   split:  h0 = f16(x) (round to nearest even), h1 = f16((x - h0) * 2^11)   -- the difference is exact
//   join:   x~ = h0 + h1 * 2^-11  (one fma; exact whenever the sum needs no more than 24 bits)
// Storage: a pixel's channels in groups of 32 = 128 bytes, [h0 x 32][h1 x 32]; the stem's input pixel is 16 bytes,
// [h0 x 4][h1 x 4] (3 channels + a zero).
#pragma once
#include <hip/hip_runtime.h>

namespace nbc {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
constexpr float kH1Scale = 2048.0f, kH1Unscale = 1.0f / 2048.0f;
constexpr _Float16 kH1UnscaleH = (_Float16)(1.0f / 2048.0f);      // 2^-11 is a normal f16

__device__ __forceinline__ void split16(float x, _Float16& h0, _Float16& h1) {
  h0 = (_Float16)x;
  h1 = (_Float16)((x - (float)h0) * kH1Scale);
}
__device__ __forceinline__ float join16(_Float16 h0, _Float16 h1) { return __builtin_fmaf((float)h1, kH1Unscale, (float)h0); }

// eight values <-> one 16-byte chunk of h0 pieces and one of h1 pieces
__device__ __forceinline__ void split16x8(const float (&v)[8], uint4& c0, uint4& c1) {
  f16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 h0, h1;
    split16(v[e], h0, h1);
    a[e] = h0; b[e] = h1;
  }
  c0 = __builtin_bit_cast(uint4, a);
  c1 = __builtin_bit_cast(uint4, b);
}
__device__ __forceinline__ void join16x8(const uint4& c0, const uint4& c1, float (&v)[8]) {
  const f16x8 a = __builtin_bit_cast(f16x8, c0), b = __builtin_bit_cast(f16x8, c1);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = join16(a[e], b[e]);
}

}  // namespace nbc
