// Implicit-GEMM convolution for gfx950 (MI355X), im2col-free, fused BN(+residual)(+ReLU) epilogue.
//
// Replaces the ATen conv2d + batch_norm + relu (+ add) sequences that the reference issues from
// torchvision's Bottleneck / the FCNHead (/root/reference/src/bark_calculator/models.py:36-37,
// 117-119).
//
// GEMM view   D[n][m] = sum_k W[n][k] * X[m][k]
//   m = output pixel (image, oy, ox) raster order      (MFMA "B" operand, ends up on the lane)
//   n = output channel                                 (MFMA "A" operand, ends up in the registers)
//   k = (kh, kw, ci): the K axis is walked in K-steps of 128 BYTES per row (64 bf16 / 32 f32);
//       with NHWC activations the 128 bytes of one K-step of one pixel are contiguous in HBM, so
//       the activation tile is a row gather with zero fill for the padding halo: no im2col
//       buffer ever exists.
// Both precisions share the data movement byte for byte; only the MFMA differs:
//   bf16: v_mfma_f32_32x32x16_bf16, one per 16-byte chunk pair
//   f32 : v_mfma_f32_32x32x2_f32, four per 16-byte chunk pair (exact f32 fma chain; lane half h
//         of MFMA q of chunk-pair ks consumes k = 8*ks + 4*h + q of the K-step)
//
// Block = 256 threads = 4 waves (2 along m x 2 along n), tile 128 (m) x BN (n), wave tile
// 64 x BN/2 as 2 x (BN/64) MFMA tiles of 32x32.  LDS: two stages of (128 + BN) rows x 128 B,
// XOR-swizzled in 16-byte chunks so that the ds_read_b128 fragment reads are conflict free;
// global -> register -> LDS staging with the next K-step's loads in flight during the MFMAs.
#include <atomic>

#include "nbc_kernels.hpp"

namespace nbc {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BM = 128;
constexpr int THREADS = 256;

// Byte offset of 16-byte chunk `chunk` (0..7) of row `row` in a [rows][128 B] tile.  The XOR
// makes 16 rows that differ in (row>>1)&7 or in parity hit 16 different 16-byte slots of the
// 256-byte bank row.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  // plain cast keeps NaN a NaN and lowers to v_cvt_pk_bf16_f32 (MI355X_MICROARCH.md, hazards)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

template <int PREC, int BN, bool STEM>
__global__ __launch_bounds__(THREADS, 2) void conv_igemm_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EB = PREC == 0 ? 4 : 2;
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int B_ROWS = BN / 32;   // weight rows staged per thread
  constexpr int NT = BN / 64;       // 32-wide n tiles per wave

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- tile coordinates; blocks that share an XCD (bid % 8) get a contiguous range of tiles,
  // n fastest, so the n-tiles of one pixel tile and the halo rows of neighbouring pixel tiles
  // meet in the same L2.
  const int tiles_n = p.Co / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- loader geometry: thread -> (chunk lc of the K-step, rows lr + 32 i)
  const int lc = tid & 7;
  const int lr = tid >> 3;
  const int pix_bytes = p.Ci * EB;
  const unsigned wrow_bytes = (unsigned)p.ksteps * 128u;

  int a_iy0[4], a_ix0[4], a_img[4];
  {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lr + 32 * i;
      if (m < p.M) {
        const int img = m / hw;
        const int rem = m - img * hw;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_img[i] = img * p.Hi * p.Wi;
      } else {            // rows past M: every tap fails the bounds test -> zeros
        a_iy0[i] = -(1 << 24);
        a_ix0[i] = 0;
        a_img[i] = 0;
      }
    }
  }
  unsigned wrow[B_ROWS];          // byte offset of this thread's chunk in weight row n, K-step 0
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) wrow[i] = (unsigned)(n0 + lr + 32 * i) * wrow_bytes + lc * 16;
  // buffer descriptors (wave-uniform: kernel arguments only); 32-bit byte offsets, range-checked
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;   // every buffer is < 2 GiB (checked on the host)

  const int cblocks = STEM ? 1 : pix_bytes / 128;   // K-steps per tap
  int ld_kh = 0, ld_kw = 0, ld_cb = 0;              // position of the next K-step to load
  u32x4 ra[4], rb[B_ROWS];

  auto load_step = [&](int t) {
    if constexpr (!STEM) {
      const int dy = ld_kh * p.dil, dx = ld_kw * p.dil;
      const int coff = ld_cb * 128 + lc * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
        const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        // halo / tail lanes get an out-of-range buffer offset: the hardware range check returns
        // zeros, the load itself stays unconditional (a select around a plain load makes hipcc
        // branch around it and drain vmcnt per element)
        const unsigned off = (unsigned)(a_img[i] + iy * p.Wi + ix) * (unsigned)pix_bytes + (unsigned)coff;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? off : kOOB, 0, 0);
      }
      if (++ld_cb == cblocks) {
        ld_cb = 0;
        if (++ld_kw == p.KW) { ld_kw = 0; ++ld_kh; }
      }
    } else {
      const int kh = t, kw = lc;                   // one kernel row per K-step, eight slots (kw < KW used)
      const bool tap_ok = lc < p.KW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = a_iy0[i] + kh * p.dil, ix = a_ix0[i] + kw * p.dil;
        const bool ok = tap_ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const unsigned off = (unsigned)(a_img[i] + iy * p.Wi + ix) * 16u;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? off : kOOB, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i)
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wrow[i] + (unsigned)t * 128u, 0, 0);
  };

  auto store_step = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE_BYTES;
    unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4*>(sa + lds_off(lr + 32 * i, lc)) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i)
      *reinterpret_cast<u32x4*>(sb + lds_off(lr + 32 * i, lc)) = rb[i];
  };

  // ---- MFMA geometry
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  f32x16 acc[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

  auto compute = [&](int stage) {
    const unsigned char* sa = smem + stage * STAGE_BYTES;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = 2 * ks + h;
      uint4 pf[2], wf[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        pf[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * 64 + i * 32 + r, chunk));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wf[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * (BN / 2) + j * 32 + r, chunk));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (PREC == 1) {
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, pf[i]), acc[j][i], 0, 0, 0);
          } else {
            const float4 wv = __builtin_bit_cast(float4, wf[j]);
            const float4 pv = __builtin_bit_cast(float4, pf[i]);
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, pv.x, acc[j][i], 0, 0, 0);
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, pv.y, acc[j][i], 0, 0, 0);
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, pv.z, acc[j][i], 0, 0, 0);
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, pv.w, acc[j][i], 0, 0, 0);
          }
        }
    }
  };

  // ---- main loop: one barrier per K-step, next step's global loads in flight under the MFMAs
  const int T = p.ksteps;
  load_step(0);
  store_step(0);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const bool more = (t + 1 < T);
    if (more) load_step(t + 1);
    compute(t & 1);
    if (more) store_step((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds pixel m (column of D) and, per accumulator group g, the four
  // consecutive channels n..n+3 (rows e + 8g + 4h of D).
  unsigned char* yb = static_cast<unsigned char*>(p.y);
  const unsigned char* resb = static_cast<const unsigned char*>(p.res);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 64 + i * 32 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * (BN / 2) + j * 32 + 8 * g + 4 * h;
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + n);
        const float4 sh = *reinterpret_cast<const float4*>(p.shift + n);
        float v[4];
        v[0] = __builtin_fmaf(acc[j][i][4 * g + 0], sc.x, sh.x);
        v[1] = __builtin_fmaf(acc[j][i][4 * g + 1], sc.y, sh.y);
        v[2] = __builtin_fmaf(acc[j][i][4 * g + 2], sc.z, sh.z);
        v[3] = __builtin_fmaf(acc[j][i][4 * g + 3], sc.w, sh.w);
        const size_t eoff = ((size_t)m * p.Co + n) * EB;
        if (resb) {
          if constexpr (PREC == 0) {
            const float4 rv = *reinterpret_cast<const float4*>(resb + eoff);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
          } else {
            const ushort4 rv = *reinterpret_cast<const ushort4*>(resb + eoff);
            v[0] += bf16_bits_to_f32(rv.x); v[1] += bf16_bits_to_f32(rv.y);
            v[2] += bf16_bits_to_f32(rv.z); v[3] += bf16_bits_to_f32(rv.w);
          }
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : (v[e] != v[e] ? v[e] : 0.f);
        }
        if constexpr (PREC == 0) {
          *reinterpret_cast<float4*>(yb + eoff) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          ushort4 o;
          o.x = f32_to_bf16_bits(v[0]); o.y = f32_to_bf16_bits(v[1]);
          o.z = f32_to_bf16_bits(v[2]); o.w = f32_to_bf16_bits(v[3]);
          *reinterpret_cast<ushort4*>(yb + eoff) = o;
        }
      }
    }
}

template <int PREC, int BN, bool STEM>
hipError_t launch_one(const ConvArgs& a, hipStream_t s) {
  constexpr int smem = 2 * (BM + BN) * 128;
  static std::atomic<unsigned long long> attr_done{0};     // bit d: attribute set on device d
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1ull)) {   // setting it twice from two threads is harmless
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<PREC, BN, STEM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << dev, std::memory_order_release);
  }
  const int tiles = ((a.M + BM - 1) / BM) * (a.Co / BN);
  hipLaunchKernelGGL((conv_igemm_kernel<PREC, BN, STEM>), dim3(tiles), dim3(THREADS), smem, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_conv_igemm(const ConvArgs& a, int precision, hipStream_t s) {
  // Shapes the kernel assumes (checked on the host so that a bad plan can never reach the GPU).
  const int eb = precision == 0 ? 4 : 2;
  if (a.M <= 0 || a.Co % 64 != 0 || a.ksteps <= 0) return hipErrorInvalidValue;
  if (a.x_bytes == 0 || a.x_bytes >= 0x80000000u || a.w_bytes == 0 || a.w_bytes >= 0x80000000u)
    return hipErrorInvalidValue;     // 32-bit range-checked buffer offsets
  if (a.stem) {
    if (a.Ci * eb != 16 || a.ksteps != a.KH || a.KW > 8) return hipErrorInvalidValue;
  } else {
    if ((a.Ci * eb) % 128 != 0 || a.ksteps != a.KH * a.KW * (a.Ci * eb / 128)) return hipErrorInvalidValue;
  }
  const bool wide = (a.Co % 128 == 0);
  if (precision == 0) {
    if (a.stem) return wide ? launch_one<0, 128, true>(a, s) : launch_one<0, 64, true>(a, s);
    return wide ? launch_one<0, 128, false>(a, s) : launch_one<0, 64, false>(a, s);
  } else {
    if (a.stem) return wide ? launch_one<1, 128, true>(a, s) : launch_one<1, 64, true>(a, s);
    return wide ? launch_one<1, 128, false>(a, s) : launch_one<1, 64, false>(a, s);
  }
}

}  // namespace nbc
