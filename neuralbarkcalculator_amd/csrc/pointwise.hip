// Memory-bound kernels of the path: image ingest, max-pool, the 1x1 classifier and the fused
// bicubic-upsample + argmax + class-count kernel.  All are HBM/L2-bound byte movers: 16-byte
// lane accesses, no MFMA.
#include <cstdlib>

#include "nbc_kernels.hpp"
#include "split16.hpp"

namespace nbc {
namespace {

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// One pixel's three normalised channels -> one 16-byte NHWC pixel (f32x4 or bf16x8, zero padded).
template <int PREC>
__device__ __forceinline__ void store_pixel(void* y, size_t pix, float c0, float c1, float c2) {
  if constexpr (PREC == 0) {
    reinterpret_cast<float4*>(y)[pix] = make_float4(c0, c1, c2, 0.f);
  } else if constexpr (PREC == 2) {               // [h0 x 4][h1 x 4]
    _Float16 a0, a1, b0, b1, d0, d1;
    split16(c0, a0, a1); split16(c1, b0, b1); split16(c2, d0, d1);
    const auto bits = [](_Float16 h) { return (unsigned)__builtin_bit_cast(unsigned short, h); };
    uint4 o;
    o.x = bits(a0) | (bits(b0) << 16); o.y = bits(d0);
    o.z = bits(a1) | (bits(b1) << 16); o.w = bits(d1);
    reinterpret_cast<uint4*>(y)[pix] = o;
  } else {
    uint4 o;
    o.x = (unsigned)f32_to_bf16_bits(c0) | ((unsigned)f32_to_bf16_bits(c1) << 16);
    o.y = (unsigned)f32_to_bf16_bits(c2);
    o.z = 0; o.w = 0;
    reinterpret_cast<uint4*>(y)[pix] = o;
  }
}

// `batch[0].to(device)` of models.py:269: float32 NCHW in, NHWC out.
template <int PREC>
__global__ __launch_bounds__(256) void ingest_f32_kernel(const float* __restrict__ x, void* __restrict__ y, int HW) {
  // grid = (ceil(HW / 512), N): two pixels per thread, no index division
  const int img = blockIdx.y;
  const float* xp = x + (size_t)img * 3 * HW;
  const int p0 = blockIdx.x * 512 + threadIdx.x, p1 = p0 + 256;
  float a[3], b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    a[c] = p0 < HW ? xp[(size_t)c * HW + p0] : 0.f;
    b[c] = p1 < HW ? xp[(size_t)c * HW + p1] : 0.f;
  }
  if (p0 < HW) store_pixel<PREC>(y, (size_t)img * HW + p0, a[0], a[1], a[2]);
  if (p1 < HW) store_pixel<PREC>(y, (size_t)img * HW + p1, b[0], b[1], b[2]);
}

// ToTensor (u8 / 255) then Normalize ((x - mean) / std), dataset.py:175-186, in IEEE f32.
template <int PREC>
__global__ void ingest_u8_kernel(const uint8_t* __restrict__ x, void* __restrict__ y, size_t total,
                                 float m0, float m1, float m2, float s0, float s1, float s2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const uint8_t* xp = x + 3 * i;
    const float a = __fdiv_rn((float)xp[0], 255.0f);
    const float b = __fdiv_rn((float)xp[1], 255.0f);
    const float c = __fdiv_rn((float)xp[2], 255.0f);
    store_pixel<PREC>(y, i, __fdiv_rn(__fsub_rn(a, m0), s0), __fdiv_rn(__fsub_rn(b, m1), s1),
                      __fdiv_rn(__fsub_rn(c, m2), s2));
  }
}

// MaxPool2d(kernel 3, stride 2, padding 1): padding is -inf, i.e. out-of-range taps do not count;
// NaN propagates like ATen's max_pool2d (a NaN tap wins): v_maximum3_f32.
// grid = (ceil(Wo*chunks / 256), Ho, N); a thread owns one 16-byte channel chunk of one output pixel,
// consecutive threads consecutive chunks (whole 128/256-byte pixel rows per 8/16 lanes).
template <int PREC>
__global__ __launch_bounds__(256) void maxpool_kernel(const void* __restrict__ x, void* __restrict__ y, int Hi, int Wi,
                                                      int chunk_shift, int Ho, int Wo) {
  // f16x2: a thread owns eight channels = the 16-byte h0 chunk j of a 128-byte group and its h1 chunk (j + 4);
  // `chunks` counts those units (C / 8), a pixel is 2 * chunks 16-byte chunks long
  constexpr int EPC = PREC == 0 ? 4 : 8;          // elements per thread
  constexpr int CPU = PREC == 2 ? 2 : 1;          // 16-byte chunks per unit
  const int chunks = 1 << chunk_shift;
  const int e = blockIdx.x * 256 + threadIdx.x;   // (ox, chunk)
  const int ox = e >> chunk_shift, ch = e & (chunks - 1);
  const int oy = blockIdx.y, img = blockIdx.z;
  if (ox >= Wo) return;
  const int ch16 = PREC == 2 ? (ch >> 2) * 8 + (ch & 3) : ch;     // the unit's first 16-byte chunk inside the pixel
  const uint4* xv = static_cast<const uint4*>(x) + (size_t)img * Hi * Wi * chunks * CPU + ch16;
  float best[EPC];
#pragma unroll
  for (int k = 0; k < EPC; ++k) best[k] = -__builtin_inff();
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = oy * 2 - 1 + dy;
    const bool oky = (unsigned)iy < (unsigned)Hi;
    const int cy = oky ? iy : oy * 2;             // a valid row to read when the tap is padding
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox * 2 - 1 + dx;
      const bool ok = oky && (unsigned)ix < (unsigned)Wi;
      const int cx = (unsigned)ix < (unsigned)Wi ? ix : ox * 2;
      const uint4 v = xv[((size_t)cy * Wi + cx) * chunks * CPU];
      float f[EPC];
      if constexpr (PREC == 2) {
        join16x8(v, xv[((size_t)cy * Wi + cx) * chunks * CPU + 4], f);
      } else if constexpr (PREC == 0) {
        f[0] = __builtin_bit_cast(float, v.x); f[1] = __builtin_bit_cast(float, v.y);
        f[2] = __builtin_bit_cast(float, v.z); f[3] = __builtin_bit_cast(float, v.w);
      } else {
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f[2 * q] = __builtin_bit_cast(float, u[q] << 16);
          f[2 * q + 1] = __builtin_bit_cast(float, u[q] & 0xffff0000u);
        }
      }
#pragma unroll
      for (int k = 0; k < EPC; ++k) best[k] = __builtin_elementwise_maximum(best[k], ok ? f[k] : -__builtin_inff());
    }
  }
  uint4 o;
  if constexpr (PREC == 2) {   // the winner's pieces again: split(join(pair)) gives the pair back (an equal value at a tie)
    uint4 o1;
    split16x8(best, o, o1);
    uint4* yp = static_cast<uint4*>(y) + (((size_t)img * Ho + oy) * Wo + ox) * chunks * CPU + ch16;
    yp[0] = o;
    yp[4] = o1;
    return;
  } else if constexpr (PREC == 0) {
    o.x = __builtin_bit_cast(unsigned, best[0]); o.y = __builtin_bit_cast(unsigned, best[1]);
    o.z = __builtin_bit_cast(unsigned, best[2]); o.w = __builtin_bit_cast(unsigned, best[3]);
  } else {          // inputs were bf16, so the max is exactly representable: truncate
    unsigned u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      u[q] = (__builtin_bit_cast(unsigned, best[2 * q]) >> 16) |
             (__builtin_bit_cast(unsigned, best[2 * q + 1]) & 0xffff0000u);
    o = make_uint4(u[0], u[1], u[2], u[3]);
  }
  static_cast<uint4*>(y)[(((size_t)img * Ho + oy) * Wo + ox) * chunks + ch] = o;
}

// classifier.4 (models.py:121): 1x1 conv 512 -> 3 with bias.  One wave per pixel: lane l owns
// channels 8l..8l+7 (an f32 fma chain in channel order), then a 64-lane xor-tree.
template <int PREC>
__global__ __launch_bounds__(256) void head1x1_kernel(const void* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ y, int M, int hw,
                                                      unsigned long long* __restrict__ counts_zero, int ncounts,
                                                      unsigned* __restrict__ nonfinite) {
  constexpr int CIN = 512;
  constexpr int PIX_PER_WAVE = 8;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float wr[3][8];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) wr[c][e] = w[c * CIN + lane * 8 + e];
  const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
  const int first = (blockIdx.x * 4 + wave) * PIX_PER_WAVE;
  if (counts_zero && blockIdx.x == 0 && threadIdx.x < ncounts) counts_zero[threadIdx.x] = 0ull;   // for the next launch
  // the wave's 8 pixel rows are requested together (one memory round trip), then reduced one by one
  constexpr int VPP = PREC == 1 ? 1 : 2;                  // 16-byte loads per lane and pixel
  uint4 raw[PIX_PER_WAVE][VPP];
#pragma unroll
  for (int q = 0; q < PIX_PER_WAVE; ++q) {
    const int m = min(first + q, M - 1);
    if constexpr (PREC == 2) {      // channels 8l .. 8l+7: h0 chunk l % 4 of group l / 4, and its h1 chunk 64 bytes on
      const uint4* xp = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(x) + (size_t)m * CIN * 4 +
                                                       (lane >> 2) * 128 + (lane & 3) * 16);
      raw[q][0] = xp[0];
      raw[q][VPP - 1] = xp[4];
    } else {
      const uint4* xp = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(x) +
                                                       ((size_t)m * CIN + lane * 8) * (PREC == 0 ? 4 : 2));
#pragma unroll
      for (int k = 0; k < VPP; ++k) raw[q][k] = xp[k];
    }
  }
#pragma unroll
  for (int q = 0; q < PIX_PER_WAVE; ++q) {
    const int m = first + q;
    float f[8];
    if constexpr (PREC == 2) {
      join16x8(raw[q][0], raw[q][VPP - 1], f);
    } else if constexpr (PREC == 0) {
      const uint4 a = raw[q][0], b = raw[q][VPP - 1];
      f[0] = __builtin_bit_cast(float, a.x); f[1] = __builtin_bit_cast(float, a.y);
      f[2] = __builtin_bit_cast(float, a.z); f[3] = __builtin_bit_cast(float, a.w);
      f[4] = __builtin_bit_cast(float, b.x); f[5] = __builtin_bit_cast(float, b.y);
      f[6] = __builtin_bit_cast(float, b.z); f[7] = __builtin_bit_cast(float, b.w);
    } else {
      const unsigned u[4] = {raw[q][0].x, raw[q][0].y, raw[q][0].z, raw[q][0].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[2 * k] = __builtin_bit_cast(float, u[k] << 16);
        f[2 * k + 1] = __builtin_bit_cast(float, u[k] & 0xffff0000u);
      }
    }
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int c = 0; c < 3; ++c) s[c] = __builtin_fmaf(f[e], wr[c][e], s[c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
      for (int c = 0; c < 3; ++c) s[c] += __shfl_xor(s[c], off, 64);
    if (lane == 0 && m < M) {
      const int img = m / hw, pix = m - img * hw;
      float* yp = y + (size_t)img * 3 * hw + pix;
      const float l0 = s[0] + b0, l1 = s[1] + b1, l2 = s[2] + b2;
      yp[0] = l0;
      yp[(size_t)hw] = l1;
      yp[2 * (size_t)hw] = l2;
      // a logit that is not finite (NaN / inf in the input or the weights; in f16x2 mode an activation beyond f16's
      // range) raises the context's sticky flag: nbc_nonfinite_seen
      if (nonfinite && !(__builtin_isfinite(l0) && __builtin_isfinite(l1) && __builtin_isfinite(l2))) atomicOr(nonfinite, 1u);
    }
  }
}

// Cubic-convolution taps, A = -0.75, exactly the expressions of ATen's
// get_cubic_upsample_coefficients (f32).
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  const float x1 = t;
  c[0] = ((A * (x1 + 1.0f) - 5.0f * A) * (x1 + 1.0f) + 8.0f * A) * (x1 + 1.0f) - 4.0f * A;
  c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  const float x2 = 1.0f - t;
  c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  c[3] = ((A * (x2 + 1.0f) - 5.0f * A) * (x2 + 1.0f) + 8.0f * A) * (x2 + 1.0f) - 4.0f * A;
}

// Source position for output index o: s = scale*(o+0.5)-0.5 (not clamped below 0 for cubic);
// i0 = floor(s) capped at in-1; t = clamp(s - i0, 0, 1); taps i0-1..i0+2 clamped to [0,in-1]
// (ATen area_pixel_compute_source_index + guard_index_and_lambda + upsample_get_value_bounded).
__device__ __forceinline__ void cubic_setup(int o, float scale, int in, int idx[4], float c[4]) {
  const float s = scale * ((float)o + 0.5f) - 0.5f;
  int i0 = (int)floorf(s);
  if (i0 > in - 1) i0 = in - 1;
  float t = s - (float)i0;
  t = fminf(fmaxf(t, 0.f), 1.f);
  cubic_coeffs(t, c);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = i0 - 1 + k;
    idx[k] = i < 0 ? 0 : (i > in - 1 ? in - 1 : i);
  }
}

// models.py:38-41 (bicubic to the input size) + models.py:270 (argmax over the 3 classes) +
// models.py:273-276 (optional 2 -> 1) + the counting of models.py:324-331.
// grid = (ceil(W/256), H, N), one output pixel per thread.
__global__ __launch_bounds__(256) void upsample_argmax_kernel(
    const float* __restrict__ lowres, int h, int w, int H, int W, float scale_y, float scale_x,
    float* __restrict__ logits_full, void* __restrict__ labels, int labels_i64,
    unsigned long long* __restrict__ counts, int exclude_nodes) {
  __shared__ unsigned int blk_counts[3];
  if (threadIdx.x < 3) blk_counts[threadIdx.x] = 0;
  __syncthreads();
  const int ox = blockIdx.x * 256 + threadIdx.x;
  const int oy = blockIdx.y;
  const int img = blockIdx.z;
  const bool live = ox < W;
  int label = -1;
  if (live) {
    int iy[4], ix[4];
    float cy[4], cx[4];
    cubic_setup(oy, scale_y, h, iy, cy);
    cubic_setup(ox, scale_x, w, ix, cx);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* src = lowres + ((size_t)img * 3 + c) * h * w;
      float out = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* row = src + (size_t)iy[j] * w;
        float t = row[ix[0]] * cx[0];
        t = __builtin_fmaf(row[ix[1]], cx[1], t);
        t = __builtin_fmaf(row[ix[2]], cx[2], t);
        t = __builtin_fmaf(row[ix[3]], cx[3], t);
        out = (j == 0) ? t * cy[0] : __builtin_fmaf(t, cy[j], out);
      }
      v[c] = out;
      if (logits_full) logits_full[(((size_t)img * 3 + c) * H + oy) * W + ox] = out;
    }
    // torch.argmax: first maximum wins, NaN counts as the maximum.
    int best = 0;
    float bv = v[0];
#pragma unroll
    for (int c = 1; c < 3; ++c) {
      const bool take = (v[c] > bv) || (v[c] != v[c] && bv == bv);
      if (take) { best = c; bv = v[c]; }
    }
    if (exclude_nodes && best == 2) best = 1;
    label = best;
    if (labels) {
      const size_t o = ((size_t)img * H + oy) * W + ox;
      if (labels_i64) static_cast<long long*>(labels)[o] = best;
      else static_cast<unsigned char*>(labels)[o] = (unsigned char)best;
    }
  }
  if (counts) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned long long mask = __ballot(label == c);
      if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&blk_counts[c], (unsigned)__popcll(mask));
    }
    __syncthreads();
    if (threadIdx.x < 3 && blk_counts[threadIdx.x])
      atomicAdd(&counts[(size_t)img * 3 + threadIdx.x], (unsigned long long)blk_counts[threadIdx.x]);
  }
}

// Tiled form of the same computation: a block owns 256 columns x UP_ROWS rows of the output and
// first copies the low-resolution window those pixels read (at most UP_WIN_R x UP_WIN_C taps per
// class, edge-clamped) into LDS, so the 48 taps per pixel are LDS reads instead of 48 gather loads
// (the one-pixel-per-thread kernel above was texture-address bound: 62 us per 1024^2 image).
// Arithmetic and its order are identical to upsample_argmax_kernel.
constexpr int UP_WIN_R = 9, UP_WIN_C = 72;

template <int UP_ROWS>
__global__ __launch_bounds__(256) void upsample_argmax_tiled_kernel(
    const float* __restrict__ lowres, int h, int w, int H, int W, float scale_y, float scale_x,
    float* __restrict__ logits_full, void* __restrict__ labels, int labels_i64,
    unsigned long long* __restrict__ counts, int exclude_nodes) {
  __shared__ float win[3][UP_WIN_R][UP_WIN_C];
  __shared__ float hsum[3][UP_WIN_R][256];
  __shared__ unsigned int blk_counts[3];
  const int tid = threadIdx.x;
  const int ox0 = blockIdx.x * 256, oy0 = blockIdx.y * UP_ROWS, img = blockIdx.z;
  if (tid < 3) blk_counts[tid] = 0;
  int ti[4];
  float tc[4];
  cubic_setup(oy0, scale_y, h, ti, tc);
  const int wy0 = ti[0];                       // tap indices are monotone in the output index
  cubic_setup(ox0, scale_x, w, ti, tc);
  const int wx0 = ti[0];
  // all of a thread's window loads are issued before the first LDS store (one memory round trip
  // per block instead of one per loop iteration)
  constexpr int WIN_ELEMS = 3 * UP_WIN_R * UP_WIN_C, WIN_ITERS = (WIN_ELEMS + 255) / 256;
  float wv[WIN_ITERS];
#pragma unroll
  for (int it = 0; it < WIN_ITERS; ++it) {
    const int e = min(tid + it * 256, WIN_ELEMS - 1);
    const int c = e / (UP_WIN_R * UP_WIN_C);
    const int rem = e - c * (UP_WIN_R * UP_WIN_C);
    const int rr = rem / UP_WIN_C, cc = rem - rr * UP_WIN_C;
    const int sy = min(wy0 + rr, h - 1), sx = min(wx0 + cc, w - 1);
    wv[it] = lowres[(((size_t)img * 3 + c) * h + sy) * w + sx];
  }
#pragma unroll
  for (int it = 0; it < WIN_ITERS; ++it) {
    const int e = tid + it * 256;
    if (e < WIN_ELEMS) (&win[0][0][0])[e] = wv[it];
  }
  __syncthreads();
  const int ox = ox0 + tid;
  const bool live_x = ox < W;
  int ix[4];
  float cx[4];
  cubic_setup(live_x ? ox : W - 1, scale_x, w, ix, cx);
#pragma unroll
  for (int k = 0; k < 4; ++k) ix[k] -= wx0;
  // Horizontal pass once per (class, window row): the four output rows that share a source row would
  // otherwise each redo its four-tap sum (48 LDS reads per output pixel; now 12 + the shared pass).
  // The per-row sums and their order are exactly those of the one-pass form, so results are identical.
  cubic_setup(min(oy0 + UP_ROWS - 1, H - 1), scale_y, h, ti, tc);
  const int nrows = ti[3] - wy0 + 1;               // window rows this block reads (block-uniform, <= UP_WIN_R)
  for (int rr = 0; rr < nrows; ++rr) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* rowp = win[c][rr];
      float t = rowp[ix[0]] * cx[0];
      t = __builtin_fmaf(rowp[ix[1]], cx[1], t);
      t = __builtin_fmaf(rowp[ix[2]], cx[2], t);
      t = __builtin_fmaf(rowp[ix[3]], cx[3], t);
      hsum[c][rr][tid] = t;                        // read back by this thread only
    }
  }
  unsigned cnt0 = 0, cnt1 = 0, cnt2 = 0;
  for (int row = 0; row < UP_ROWS; ++row) {
    const int oy = oy0 + row;
    if (oy >= H) break;                        // block-uniform
    int iy[4];
    float cy[4];
    cubic_setup(oy, scale_y, h, iy, cy);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float out = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = hsum[c][iy[j] - wy0][tid];
        out = (j == 0) ? t * cy[0] : __builtin_fmaf(t, cy[j], out);
      }
      v[c] = out;
      if (logits_full && live_x) logits_full[(((size_t)img * 3 + c) * H + oy) * W + ox] = out;
    }
    int best = 0;
    float bv = v[0];
#pragma unroll
    for (int c = 1; c < 3; ++c) {
      const bool take = (v[c] > bv) || (v[c] != v[c] && bv == bv);
      if (take) { best = c; bv = v[c]; }
    }
    if (exclude_nodes && best == 2) best = 1;
    if (live_x) {
      cnt0 += best == 0; cnt1 += best == 1; cnt2 += best == 2;
      if (labels) {
        const size_t o = ((size_t)img * H + oy) * W + ox;
        if (labels_i64) static_cast<long long*>(labels)[o] = best;
        else static_cast<unsigned char*>(labels)[o] = (unsigned char)best;
      }
    }
  }
  if (counts) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      cnt0 += __shfl_xor(cnt0, off, 64);
      cnt1 += __shfl_xor(cnt1, off, 64);
      cnt2 += __shfl_xor(cnt2, off, 64);
    }
    if ((tid & 63) == 0) {
      if (cnt0) atomicAdd(&blk_counts[0], cnt0);
      if (cnt1) atomicAdd(&blk_counts[1], cnt1);
      if (cnt2) atomicAdd(&blk_counts[2], cnt2);
    }
    __syncthreads();
    if (tid < 3 && blk_counts[tid])
      atomicAdd(&counts[(size_t)img * 3 + tid], (unsigned long long)blk_counts[tid]);
  }
}

template <int PREC>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
  const size_t total = (size_t)N * HW * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t pix = i / C;                 // img*HW + p
    const size_t img = pix / HW, p = pix - img * HW;
    float v;
    if constexpr (PREC == 0) v = static_cast<const float*>(x)[i];
    else if constexpr (PREC == 2) {
      const _Float16* hp = static_cast<const _Float16*>(x) + pix * (size_t)C * 2 + (c >> 5) * 64 + (c & 31);
      v = join16(hp[0], hp[32]);
    } else v = bf16_bits_to_f32(static_cast<const unsigned short*>(x)[i]);
    y[(img * C + c) * (size_t)HW + p] = v;
  }
}

// Largest finite |value| of an activation tensor as it is STORED (f16x2: pieces joined; the calibration guard of
// nbc_activation_peaks): one wave-reduced atomicMax on the float's bit pattern (non-negative floats order like integers).
template <int PREC>
__global__ void absmax_kernel(const void* __restrict__ x, size_t elems, int C, unsigned* __restrict__ out) {
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x) {
    float v;
    if constexpr (PREC == 0) v = static_cast<const float*>(x)[i];
    else if constexpr (PREC == 2) {
      const size_t pix = i / C;
      const int c = (int)(i - pix * C);
      const _Float16* hp = static_cast<const _Float16*>(x) + pix * (size_t)C * 2 + (c >> 5) * 64 + (c & 31);
      v = join16(hp[0], hp[32]);
    } else v = bf16_bits_to_f32(static_cast<const unsigned short*>(x)[i]);
    v = __builtin_fabsf(v);
    if (v < __builtin_inff() && v > m) m = v;          // NaN and infinity do not count (the non-finite flag reports those)
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __builtin_bit_cast(unsigned, m));
}

// ---- preprocessor (next-row N4): models.py:191-198, skimage.transform.resize(order=3, mode='reflect',
// anti_aliasing=False) of the ToTensor'd image, as scikit-image 0.18.3's compiled _warp_fast evaluates it for a
// float32 image (restated and pinned value for value in neuralbarkcalculator_amd/predict.py,
// resize_bicubic_reflect): EVERYTHING in f32, one rounding per operation, no contraction -- the sample
// position f*i + (f/2 - 1/2) with f and the offset rounded from double first, its fractional part, the
// 4 x 4 taps (reflected borders) through the Catmull-Rom polynomial in its source order row-wise then
// column-wise, the clip to the input range.  Bit-identical to the numpy form, hence to scikit-image.
__device__ __forceinline__ int reflect_index(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i %= period;
  if (i < 0) i += period;
  return i >= n ? period - i : i;
}
// (plain operators under `fp contract(off)`: every product and sum is rounded on its own; the __f*_rn
// intrinsics inline to operations that carry the library's own contraction flags and were fused)
__device__ __forceinline__ float cubic_f32(float x, float f0, float f1, float f2, float f3) {
#pragma clang fp contract(off)
  // f1 + 0.5*x*(f2 - f0 + x*(2*f0 - 5*f1 + 4*f2 - f3 + x*(3*(f1 - f2) + f3 - f0))), left to right as written
  const float d12 = f1 - f2;
  const float t3 = 3.0f * d12;
  const float inner = (t3 + f3) - f0;
  const float p2 = 2.0f * f0, p5 = 5.0f * f1, p4 = 4.0f * f2;
  const float xi = x * inner;
  const float mid = (((p2 - p5) + p4) - f3) + xi;
  const float xm = x * mid;
  const float outer = (f2 - f0) + xm;
  const float hx = 0.5f * x;
  const float ho = hx * outer;
  return f1 + ho;
}

__global__ void minmax_u8_kernel(const uint8_t* __restrict__ x, size_t n, unsigned* __restrict__ mm) {
  unsigned lo = 255u, hi = 0u;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned v = x[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}

// grid = (ceil(out_w / 256), out_h); a thread owns one output pixel (three channels).
// Outputs, each optional: the float32 image (what skimage's resize returns), its uint8 form as
// skimage.io.imsave writes it through imageio (uint8(float64(x) * 255 + 0.499999999), models.py:203), and per
// output row the number of pixels trim_black calls lit (float32 channel sum > 1e-3, models.py:158-159).
__global__ __launch_bounds__(256) void resize_cubic_kernel(const uint8_t* __restrict__ src, int H, int W, float* __restrict__ dst,
                                                           uint8_t* __restrict__ dst_u8, int* __restrict__ row_lit,
                                                           int out_h, int out_w, const unsigned* __restrict__ mm) {
#pragma clang fp contract(off)
  const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  bool lit = false;
  if (ox < out_w) {
    const double fy = __ddiv_rn((double)H, (double)out_h), fx = __ddiv_rn((double)W, (double)out_w);
    const double hy = fy * 0.5, hx = fx * 0.5;
    const float by = (float)(hy - 0.5), bx = (float)(hx - 0.5);
    const float my = (float)fy * (float)oy, mx = (float)fx * (float)ox;
    const float ry = my + by, rx = mx + bx;
    const int y0 = (int)floorf(ry), x0 = (int)floorf(rx);
    const float ty = ry - (float)y0, tx = rx - (float)x0;
    int cols[4], rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { cols[k] = reflect_index(x0 + k - 1, W); rows[k] = reflect_index(y0 + k - 1, H); }
    const float lo = __fdiv_rn((float)mm[0], 255.0f), hi = __fdiv_rn((float)mm[1], 255.0f);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float fr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint8_t* row = src + (size_t)rows[k] * W * 3 + c;
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __fdiv_rn((float)row[(size_t)cols[j] * 3], 255.0f);      // ToTensor
        fr[k] = cubic_f32(tx, f[0], f[1], f[2], f[3]);
      }
      const float r = cubic_f32(ty, fr[0], fr[1], fr[2], fr[3]);
      v[c] = r < lo ? lo : (r > hi ? hi : r);                // np.clip(out, img.min(), img.max())
    }
    const size_t o = ((size_t)oy * out_w + ox) * 3;
    if (dst) { dst[o] = v[0]; dst[o + 1] = v[1]; dst[o + 2] = v[2]; }
    if (dst_u8) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double q255 = (double)v[c] * 255.0;
        double q = q255 + 0.499999999;
        q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q);
        dst_u8[o + c] = (uint8_t)(int)q;                       // astype(uint8): truncation
      }
    }
    const float s01 = v[0] + v[1];
    lit = (s01 + v[2]) > 1e-3f;                              // np.sum(image, axis=-1) > 1e-3 in float32
  }
  if (row_lit) {
    const unsigned long long m = __ballot(lit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&row_lit[oy], (int)__popcll(m));
  }
}

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  const size_t cap = 256 * 8;                 // 8 blocks per CU, grid-stride the rest
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

hipError_t launch_ingest_f32(const float* x, void* y, int N, int H, int W, int precision, hipStream_t s) {
  if (N > 65535 || (long long)H * W > 0x7fffffffLL) return hipErrorInvalidValue;
  const int HW = H * W;
  dim3 grid((HW + 511) / 512, N);
  if (precision == 0) hipLaunchKernelGGL(ingest_f32_kernel<0>, grid, dim3(256), 0, s, x, y, HW);
  else if (precision == 2) hipLaunchKernelGGL(ingest_f32_kernel<2>, grid, dim3(256), 0, s, x, y, HW);
  else hipLaunchKernelGGL(ingest_f32_kernel<1>, grid, dim3(256), 0, s, x, y, HW);
  return hipGetLastError();
}

hipError_t launch_ingest_u8(const uint8_t* x, void* y, int N, int H, int W, const float mean[3],
                            const float stdv[3], int precision, hipStream_t s) {
  const size_t total = (size_t)N * H * W;
  const int g = grid_for(total, 256);
  if (precision == 0)
    hipLaunchKernelGGL(ingest_u8_kernel<0>, dim3(g), dim3(256), 0, s, x, y, total, mean[0], mean[1],
                       mean[2], stdv[0], stdv[1], stdv[2]);
  else if (precision == 2)
    hipLaunchKernelGGL(ingest_u8_kernel<2>, dim3(g), dim3(256), 0, s, x, y, total, mean[0], mean[1],
                       mean[2], stdv[0], stdv[1], stdv[2]);
  else
    hipLaunchKernelGGL(ingest_u8_kernel<1>, dim3(g), dim3(256), 0, s, x, y, total, mean[0], mean[1],
                       mean[2], stdv[0], stdv[1], stdv[2]);
  return hipGetLastError();
}

hipError_t launch_maxpool3x3s2(const void* x, void* y, int N, int Hi, int Wi, int C, int Ho, int Wo,
                               int precision, hipStream_t s) {
  const int epc = precision == 0 ? 4 : 8;
  if (C % epc != 0 || (precision == 2 && C % 32 != 0)) return hipErrorInvalidValue;
  const int chunks = C / epc;
  int shift = 0;
  while ((1 << shift) < chunks) ++shift;
  if ((1 << shift) != chunks || chunks > 256 || Ho > 65535 || N > 65535) return hipErrorInvalidValue;   // C = 64 on this path
  dim3 grid((Wo * chunks + 255) / 256, Ho, N);
  if (precision == 0) hipLaunchKernelGGL(maxpool_kernel<0>, grid, dim3(256), 0, s, x, y, Hi, Wi, shift, Ho, Wo);
  else if (precision == 2) hipLaunchKernelGGL(maxpool_kernel<2>, grid, dim3(256), 0, s, x, y, Hi, Wi, shift, Ho, Wo);
  else hipLaunchKernelGGL(maxpool_kernel<1>, grid, dim3(256), 0, s, x, y, Hi, Wi, shift, Ho, Wo);
  return hipGetLastError();
}

hipError_t launch_head1x1(const void* x, const float* w, const float* bias, float* y, int N, int hw,
                          int precision, unsigned long long* counts_zero, unsigned* nonfinite, hipStream_t s) {
  if (3 * N > 256) counts_zero = nullptr;      // the caller then clears the counters itself
  const int M = N * hw;
  const int blocks = (M + 31) / 32;           // 4 waves x 8 pixels
  if (precision == 0) hipLaunchKernelGGL(head1x1_kernel<0>, dim3(blocks), dim3(256), 0, s, x, w, bias, y, M, hw, counts_zero, 3 * N, nonfinite);
  else if (precision == 2) hipLaunchKernelGGL(head1x1_kernel<2>, dim3(blocks), dim3(256), 0, s, x, w, bias, y, M, hw, counts_zero, 3 * N, nonfinite);
  else hipLaunchKernelGGL(head1x1_kernel<1>, dim3(blocks), dim3(256), 0, s, x, w, bias, y, M, hw, counts_zero, 3 * N, nonfinite);
  return hipGetLastError();
}

hipError_t launch_upsample_argmax(const float* lowres, int N, int h, int w, int H, int W,
                                  float* logits_full, void* labels, int labels_i64,
                                  unsigned long long* counts, int exclude_nodes, hipStream_t s) {
  if (H > 65535 || N > 65535) return hipErrorInvalidValue;
  // ATen area_pixel_compute_scale<float>(in, out, align_corners=false, scales=nullopt)
  const float scale_y = (float)h / (float)H;
  const float scale_x = (float)w / (float)W;
  // the tiled kernel needs the block's tap window to fit its LDS image (always true for the
  // path's own x8 geometry: 36 x 5 taps); anything else takes the one-pixel-per-thread kernel
  constexpr int up_rows = 8;                       // output rows per block (4 and 16 measured slower)
  const bool fits = (int)(scale_x * 255.0f) + 6 <= UP_WIN_C && (int)(scale_y * (up_rows - 1)) + 6 <= UP_WIN_R;
  if (fits) {
    dim3 grid((W + 255) / 256, (H + up_rows - 1) / up_rows, N);
    hipLaunchKernelGGL(upsample_argmax_tiled_kernel<up_rows>, grid, dim3(256), 0, s, lowres, h, w, H, W, scale_y,
                       scale_x, logits_full, labels, labels_i64, counts, exclude_nodes);
  } else {
    dim3 grid((W + 255) / 256, H, N);
    hipLaunchKernelGGL(upsample_argmax_kernel, grid, dim3(256), 0, s, lowres, h, w, H, W, scale_y, scale_x,
                       logits_full, labels, labels_i64, counts, exclude_nodes);
  }
  return hipGetLastError();
}

hipError_t launch_resize_cubic_u8(const uint8_t* src, int H, int W, float* dst, uint8_t* dst_u8, int* row_lit, int out_h, int out_w,
                                  unsigned* minmax, hipStream_t s) {
  if (H < 1 || W < 1 || out_h < 1 || out_w < 1 || out_h > 65535) return hipErrorInvalidValue;
  // min starts at 255, max at 0: two device-side fills, no host source buffer that would have to outlive the call
  hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(minmax), 255, 1, s);
  if (e != hipSuccess) return e;
  e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(minmax + 1), 0, 1, s);
  if (e != hipSuccess) return e;
  if (row_lit) {
    e = hipMemsetAsync(row_lit, 0, sizeof(int) * (size_t)out_h, s);
    if (e != hipSuccess) return e;
  }
  const size_t n = (size_t)H * W * 3;
  hipLaunchKernelGGL(minmax_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, n, minmax);
  hipLaunchKernelGGL(resize_cubic_kernel, dim3((out_w + 255) / 256, out_h), dim3(256), 0, s, src, H, W, dst, dst_u8, row_lit, out_h,
                     out_w, minmax);
  return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw_f32(const void* x, float* y, int N, int H, int W, int C, int precision,
                                   hipStream_t s) {
  const size_t total = (size_t)N * H * W * C;
  const int g = grid_for(total, 256);
  if (precision == 0) hipLaunchKernelGGL(nhwc_to_nchw_kernel<0>, dim3(g), dim3(256), 0, s, x, y, N, H * W, C);
  else if (precision == 2) hipLaunchKernelGGL(nhwc_to_nchw_kernel<2>, dim3(g), dim3(256), 0, s, x, y, N, H * W, C);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<1>, dim3(g), dim3(256), 0, s, x, y, N, H * W, C);
  return hipGetLastError();
}

hipError_t launch_absmax(const void* x, size_t elems, int C, int precision, unsigned* out, hipStream_t s) {
  const int g = grid_for(elems, 256);
  if (precision == 0) hipLaunchKernelGGL(absmax_kernel<0>, dim3(g), dim3(256), 0, s, x, elems, C, out);
  else if (precision == 2) hipLaunchKernelGGL(absmax_kernel<2>, dim3(g), dim3(256), 0, s, x, elems, C, out);
  else hipLaunchKernelGGL(absmax_kernel<1>, dim3(g), dim3(256), 0, s, x, elems, C, out);
  return hipGetLastError();
}

}  // namespace nbc
