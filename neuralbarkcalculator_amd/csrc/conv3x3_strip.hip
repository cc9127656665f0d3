// 3x3 stride-1 convolution with an LDS-resident input row strip (gfx950).
//
// The generic implicit-GEMM kernel (conv_igemm_dma.hip) re-gathers the activation tile for every
// tap: nine 128-byte K-steps of one channel block move 9 x (BM + BN) rows through L2 -> LDS, and the
// ablation in DESIGN.md section 6 shows that this delivery path, not the MFMAs, bounds the 3x3 layers.
// For a tile of R whole output rows the three horizontal taps (kw = 0, 1, 2) of one (kh, channel
// block) read the SAME input pixels shifted by `dil` columns, so this kernel
//   * walks K in the order (kh, channel block, kw) -- the packer keeps a second copy of the 3x3
//     weights in that order (nbc_net.cpp, w_strip_off), so weight rows are still read sequentially;
//   * DMAs, once per (kh, channel block), a STRIP of R x (Wo + 2*dil) input pixels (128 bytes of
//     channels each) into one of two LDS strip slots, zero page for the padding halo;
//   * DMAs one BN x 128-byte weight tile per K-step into a four-slot ring;
//   * reads the pixel (MFMA B) fragments of tap kw from the strip at a row offset of kw*dil.
// Per three K-steps a 256 x 128 tile then moves 2 x (Wo+2d) x 128 B + 3 x 16 KiB (82 KiB at Wo = 128,
// d = 1) instead of 3 x 48 KiB: 43 % fewer bytes through the L2 -> LDS path.
//
// Tile 256 pixels (R = 256 / Wo whole rows of one image) x 128 channels, 8 waves (4 along pixels x
// 2 along channels, 64 x 64 each), bf16 on v_mfma_f32_16x16x32_bf16, f32 on v_mfma_f32_32x32x2_f32.
// Eligibility (checked on the host, strip_eligible()): 3x3, stride 1, pad == dil, Wo a power of two
// in [16, 256], R x (Wo + 2*dil) <= 320 strip rows, Cin a multiple of the 128-byte K-step, Cout a
// multiple of 128.  The choice depends on the layer shape only, never on timing, so a plan is
// deterministic; the K order differs from the generic kernel's, so the two agree to rounding only.
//
// Pipeline: one s_barrier per K-step; vector-memory ops are issued in the order
//   step (g, kw=0): W(t+3) x2, strip(g+1) x5      step (g, 1): W(t+3) x2      step (g, 2): W(t+3) x2
// and retire in order, so the counted waits at the top of a step are vmcnt(4) for kw = 0 (the strip of
// this group and W(t) are older than the four W ops of the two previous steps) and vmcnt(9) for
// kw = 1, 2; the last group drains with vmcnt(0).
#include <cstdlib>

#include "nbc_kernels.hpp"

namespace nbc {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 256, BN = 128, THREADS = 512;
constexpr int STRIP_ROWS = 320;                    // 5 DMA passes of 64 rows
constexpr int STRIP_BYTES = STRIP_ROWS * 128;      // 40 KiB
constexpr int W_BYTES = BN * 128;                  // 16 KiB
constexpr int W_SLOTS = 4;
constexpr int TABLE_OFF = 2 * STRIP_BYTES + W_SLOTS * W_BYTES;     // scale/shift table (2 KiB)
constexpr int SMEM_BYTES = TABLE_OFF + 2048;                       // 146 KiB

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int PREC>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_strip_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EB = PREC == 0 ? 4 : 2;
  constexpr bool M16 = (PREC == 1);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- geometry: a tile is R whole output rows of one image x BN channels
  const int wsh = p.wo_shift;                       // log2(Wo), 4..8
  const int Wo = 1 << wsh;
  const int R = BM >> wsh;
  const int d = p.dil;
  const int SW = Wo + 2 * d;                        // strip pixels per row
  const int srows = R * SW;                         // <= STRIP_ROWS (host)
  const int tiles_per_img = (p.Ho + R - 1) / R;
  const int tiles_n = p.Co / BN;
  const int nblk = p.N * tiles_per_img * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int img = tile_m / tiles_per_img;
  const int row0 = (tile_m - img * tiles_per_img) * R;   // first output row of the tile
  const int n0 = tile_n * BN;

  const unsigned char* xb = static_cast<const unsigned char*>(p.x);
  const unsigned char* zpage = static_cast<const unsigned char*>(p.zero);
  const int pix_bytes = p.Ci * EB;
  const int cblocks = pix_bytes / 128;
  const long long row_pitch = (long long)p.Wi * pix_bytes;
  const size_t wrow_bytes = (size_t)p.ksteps * 128;

  // ---- loader: thread -> physical slot ps, rows lr + 64 i (strip: 5 passes, weights: 2 passes)
  const int ps = tid & 7;
  const int lr = tid >> 3;
  const unsigned char* s_base[5];    // centre-row (kh = 1) source of this thread's strip row, channel block 0
  unsigned s_ok[5];                  // bit kh: input row inside the image (and the strip row real, column inside)
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int s = lr + 64 * i;
    const int r = s / SW, xs = s - r * SW;
    const int oy = row0 + r, ix = xs - d;
    const bool colok = s < srows && oy < p.Ho && (unsigned)ix < (unsigned)p.Wi;
    unsigned bits = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy + (kh - 1) * d;
      if (colok && (unsigned)iy < (unsigned)p.Hi) bits |= 1u << kh;
    }
    s_ok[i] = bits;
    s_base[i] = xb + ((long long)(img * p.Hi + oy) * p.Wi + ix) * pix_bytes + (ps ^ ((s >> 1) & 7)) * 16;
  }
  const unsigned char* wsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = lr + 64 * i;
    wsrc[i] = static_cast<const unsigned char*>(p.w_strip) + (size_t)(n0 + row) * wrow_bytes +
              (ps ^ ((row >> 1) & 7)) * 16;
  }
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned smem_base = (unsigned)(size_t)(lds_u8*)smem;
  const unsigned wave_off = (unsigned)wave * 1024u;

  auto issue_w = [&](int t) {        // weight tile of K-step t -> ring slot t % 4
    const unsigned dst = smem_base + 2 * STRIP_BYTES + (unsigned)(t & (W_SLOTS - 1)) * W_BYTES + wave_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma16(wsrc[i], dst + (unsigned)(64 * 128 * i));
      wsrc[i] += 128;
    }
  };
  auto issue_strip = [&](int g) {    // strip of group g = (kh, cb) -> strip slot g & 1
    const int kh = g / cblocks, cb = g - kh * cblocks;
    const long long delta = (long long)(kh - 1) * d * row_pitch + (long long)cb * 128;
    const unsigned dst = smem_base + (unsigned)(g & 1) * STRIP_BYTES + wave_off;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const bool ok = ((s_ok[i] >> kh) & 1u) != 0;
      const unsigned char* src = ok ? s_base[i] + delta : zpage;
      dma16(src, dst + (unsigned)(64 * 128 * i));
    }
  };

  // ---- MFMA geometry
  const int wm = wave & 3, wn = wave >> 2;          // 4 x 2 waves, 64 pixels x 64 channels each
  const int r16 = lane & 15, q16 = lane >> 4;
  const int r32 = lane & 31, h32 = lane >> 5;
  // strip row (before the kw shift) of the pixels this lane feeds to the MFMA B operand
  constexpr int PT = M16 ? 4 : 2;                   // pixel tiles per wave (16- or 32-wide)
  int prow[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int pix = wm * 64 + (M16 ? i * 16 + r16 : i * 32 + r32);
    prow[i] = (pix >> wsh) * SW + (pix & (Wo - 1));
  }
  f32x4 acc16[M16 ? 4 : 1][M16 ? 4 : 1];
  f32x16 acc32[M16 ? 1 : 2][M16 ? 1 : 2];
  if constexpr (M16) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[j][i][e] = 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[j][i][e] = 0.f;
  }

  auto compute = [&](int strip_slot, int w_slot, int shift) {
    const unsigned char* sa = smem + strip_slot * STRIP_BYTES;
    const unsigned char* sb = smem + 2 * STRIP_BYTES + w_slot * W_BYTES;
    if constexpr (M16) {
      uint4 pf[2][4], wf[2][4];
      auto load_half = [&](int half, uint4 (&pfr)[4], uint4 (&wfr)[4]) {
        const int chunk = 4 * half + q16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          pfr[i] = *reinterpret_cast<const uint4*>(sa + lds_off(prow[i] + shift, chunk));
#pragma unroll
        for (int j = 0; j < 4; ++j)
          wfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * 64 + j * 16 + r16, chunk));
      };
      load_half(0, pf[0], wf[0]);
      load_half(1, pf[1], wf[1]);
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc16[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, wf[half][j]), __builtin_bit_cast(bf16x8, pf[half][i]), acc16[j][i], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int chunk = 2 * ks + h32;
        uint4 pf[2], wf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          pf[i] = *reinterpret_cast<const uint4*>(sa + lds_off(prow[i] + shift, chunk));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * 64 + j * 32 + r32, chunk));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float4 wv = __builtin_bit_cast(float4, wf[j]);
            const float4 pv = __builtin_bit_cast(float4, pf[i]);
            acc32[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, pv.x, acc32[j][i], 0, 0, 0);
            acc32[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, pv.y, acc32[j][i], 0, 0, 0);
            acc32[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, pv.z, acc32[j][i], 0, 0, 0);
            acc32[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, pv.w, acc32[j][i], 0, 0, 0);
          }
      }
    }
  };

  // ---- pipeline
  const int G = 3 * cblocks;                        // (kh, cb) groups; T = 3 G K-steps
  const int T = 3 * G;
  if (wave == 0 && lane < BN / 4) {                 // scale/shift table, older than every ring DMA
    dma16(p.scale + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF);
    dma16(p.shift + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF + 1024u);
  }
  issue_strip(0);
  issue_w(0);
  issue_w(1);
  issue_w(2);
  for (int g = 0; g < G; ++g) {
    const bool last = (g == G - 1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int t = 3 * g + kw;
      if (kw == 0) wait_vmcnt<4>();
      else if (last) wait_vmcnt<0>();
      else wait_vmcnt<9>();
      __builtin_amdgcn_s_barrier();
      if (t + 3 < T) issue_w(t + 3);
      if (kw == 0 && !last) issue_strip(g + 1);
      compute(g & 1, t & (W_SLOTS - 1), kw * d);
    }
  }

  // ---- epilogue: same LDS transpose as the generic kernel, 32 pixels x 64 channels per wave at a time
  constexpr int PITCH = 64 * 4 + 16;
  constexpr int OUT_CH = 16 / EB;
  constexpr int CPR = 64 / OUT_CH;                  // 16-byte output chunks per pixel row of the slab
  constexpr int PIX_PER_PASS = 64 / CPR;
  constexpr int PASSES = 32 / PIX_PER_PASS;
  __syncthreads();
  unsigned char* scr = smem + wave * (32 * PITCH);
  unsigned char* yb = static_cast<unsigned char*>(p.y);
  const int o_pix = lane / CPR, o_chunk = lane % CPR;
#pragma unroll
  for (int i = 0; i < 2; ++i) {                     // two 32-pixel slabs per wave
    if constexpr (M16) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nl = j * 16 + 4 * q16;
          const float4 sc = *reinterpret_cast<const float4*>(smem + TABLE_OFF + (wn * 64 + nl) * 4);
          const float4 sh = *reinterpret_cast<const float4*>(smem + TABLE_OFF + 1024 + (wn * 64 + nl) * 4);
          const f32x4 a = acc16[j][2 * i + i2];
          float4 v;
          v.x = __builtin_fmaf(a[0], sc.x, sh.x);
          v.y = __builtin_fmaf(a[1], sc.y, sh.y);
          v.z = __builtin_fmaf(a[2], sc.z, sh.z);
          v.w = __builtin_fmaf(a[3], sc.w, sh.w);
          *reinterpret_cast<float4*>(scr + (i2 * 16 + r16) * PITCH + nl * 4) = v;
        }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int nl = j * 32 + 8 * g4 + 4 * h32;
          const float4 sc = *reinterpret_cast<const float4*>(smem + TABLE_OFF + (wn * 64 + nl) * 4);
          const float4 sh = *reinterpret_cast<const float4*>(smem + TABLE_OFF + 1024 + (wn * 64 + nl) * 4);
          float4 v;
          v.x = __builtin_fmaf(acc32[j][i][4 * g4 + 0], sc.x, sh.x);
          v.y = __builtin_fmaf(acc32[j][i][4 * g4 + 1], sc.y, sh.y);
          v.z = __builtin_fmaf(acc32[j][i][4 * g4 + 2], sc.z, sh.z);
          v.w = __builtin_fmaf(acc32[j][i][4 * g4 + 3], sc.w, sh.w);
          *reinterpret_cast<float4*>(scr + r32 * PITCH + nl * 4) = v;
        }
    }
#pragma unroll
    for (int ps2 = 0; ps2 < PASSES; ++ps2) {
      const int pl = ps2 * PIX_PER_PASS + o_pix;                 // pixel inside the slab
      const int pix = wm * 64 + i * 32 + pl;                     // pixel inside the tile
      const int oy = row0 + (pix >> wsh), ox = pix & (Wo - 1);
      float v[OUT_CH];
      const float4* sp = reinterpret_cast<const float4*>(scr + pl * PITCH + o_chunk * OUT_CH * 4);
#pragma unroll
      for (int q = 0; q < OUT_CH / 4; ++q) {
        const float4 t4 = sp[q];
        v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
      }
      if (oy < p.Ho) {
        const size_t m = ((size_t)img * p.Ho + oy) * Wo + ox;
        const size_t eoff = (m * p.Co + n0 + wn * 64 + o_chunk * OUT_CH) * EB;
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < OUT_CH; ++e) v[e] = v[e] > 0.f ? v[e] : (v[e] != v[e] ? v[e] : 0.f);
        }
        uint4 o;
        if constexpr (PREC == 0) {
          o.x = __builtin_bit_cast(unsigned, v[0]); o.y = __builtin_bit_cast(unsigned, v[1]);
          o.z = __builtin_bit_cast(unsigned, v[2]); o.w = __builtin_bit_cast(unsigned, v[3]);
        } else {
          o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
          o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
          o.z = (unsigned)f32_to_bf16_bits(v[4 % OUT_CH]) | ((unsigned)f32_to_bf16_bits(v[5 % OUT_CH]) << 16);
          o.w = (unsigned)f32_to_bf16_bits(v[6 % OUT_CH]) | ((unsigned)f32_to_bf16_bits(v[7 % OUT_CH]) << 16);
        }
        *reinterpret_cast<uint4*>(yb + eoff) = o;
      }
    }
  }
}

}  // namespace

// Shape test for the row-strip kernel (see the header comment).  Deterministic: shape only.
bool strip_eligible(const ConvArgs& a, int precision) {
  const int eb = precision == 0 ? 4 : 2;
  if (a.stem || a.res != nullptr || a.w_strip == nullptr) return false;
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != a.dil) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi || a.wo_shift < 4 || a.wo_shift > 8) return false;
  if ((a.Ci * eb) % 128 != 0 || a.Co % 128 != 0 || a.Ci < 256) return false;
  const int R = 256 >> a.wo_shift;
  if (R * (a.Wo + 2 * a.dil) > 320) return false;
  return true;
}

hipError_t launch_conv3x3_strip(const ConvArgs& a, int precision, hipStream_t s) {
  if (!strip_eligible(a, precision) || a.zero == nullptr) return hipErrorInvalidValue;
  static unsigned long long attr_done[2] = {0, 0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  const void* kern = precision == 0 ? reinterpret_cast<const void*>(&conv3x3_strip_kernel<0>)
                                    : reinterpret_cast<const void*>(&conv3x3_strip_kernel<1>);
  if (!((attr_done[precision] >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return e;
    attr_done[precision] |= 1ull << dev;
  }
  const int R = 256 >> a.wo_shift;
  const int tiles = a.N * ((a.Ho + R - 1) / R) * (a.Co / 128);
  if (precision == 0) hipLaunchKernelGGL(conv3x3_strip_kernel<0>, dim3(tiles), dim3(THREADS), SMEM_BYTES, s, a);
  else hipLaunchKernelGGL(conv3x3_strip_kernel<1>, dim3(tiles), dim3(THREADS), SMEM_BYTES, s, a);
  return hipGetLastError();
}

}  // namespace nbc
