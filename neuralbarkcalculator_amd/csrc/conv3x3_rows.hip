// 3x3 stride-1 convolution on 128-pixel-wide feature maps in f16x2 (NBC_PREC_F16X2) whose pixel rows STAY IN LDS for the three
// taps of a kernel row, one barrier per ROW-STEP: layer3 / layer4 conv2 and classifier.0 (tiles 18 and 20) and layer2.1-3 conv2
// (tile 19) of a 1024-pixel-wide image -- half of the forward's time.
//
// Why.  The generic kernel (conv_igemm_dma.hip) fetches, for every K-step (tap, 32-channel block), the 128 pixel rows of its
// tile again: 16 KiB of pixels + 16-32 KiB of weights per step from L2 into LDS, behind one barrier per K-step.  Measured in the
// network (profiles/r05_f16x2_kloop_ablations_in_network.log): the head conv's K loop WITHOUT a single MFMA or fragment read --
// LDS-DMA and barriers only -- takes 423 of its 738 us (7.25 GB at 17 TB/s, the L2 -> LDS gather rate of this chip), the same
// loop without its DMAs 496: the L2 -> LDS stream is a bottleneck of its own beside the matrix pipe, and the chip's power
// limit couples the two.  The three taps (kh, kw = 0, 1, 2) of a kernel row read the SAME input row shifted by the dilation.
// Here a pipeline step is a ROW-STEP (channel block, kh): one input row (128 + 2 dil pixels x 128 bytes of the channel block,
// zero halo from the buffer resource's range check) fetched ONCE into a row slot and read by its three K-steps (kw) at a
// shifted pixel index, the three taps' weight panels in flight with it, ONE barrier and one LDS-DMA round trip for three
// K-steps of MFMAs.  Pixel traffic falls to a third (22 KiB per K-step where the generic 128 x 128 tile fetches 32, the
// 128 x 256 tile 48 for twice the outputs), barriers and round trips to a third.
// Measured (profiles/r05_rows_kernel_*.log, r05_rows_tile18_one_barrier_per_rowstep.log): the head conv 738 -> 680 us,
// layer4's conv2 187 -> 174, layer3's 57 -> 51, layer2.1-3's 23.5 -> 20.5; the forward +2.9 %.
// Tile 20 then takes TWO output rows a dilation apart x 64 channels per block: four input rows per channel block serve six (output
// row, kernel row) pairs and a weight panel is half as wide -- 72 KiB into LDS per output row and channel block where tile 18
// moves 99; same K order per output, same bits: head conv -2.3 ... -3.1 %, layer4's conv2 -1.4 ... -2.4 %, layer3's -1 %, the forward
// +1.4 % (profiles/r05_rowstep_two_output_rows.log).
// (Built, measured and removed on the way, same logs: the same rows with one barrier per K-step -- 128 x 128 with loader waves
// and a 256 x 128 tile of 64 x 64 wave tiles that halves the weight traffic as well: +2.0 % on the forward, every layer
// slower than on this kernel; 64 x 64 wave tiles here: +2-4 % time.)
//
// K order.  (channel block, kh, kw) -- the generic kernel walks (kh, kw, channel block).  The order is a property of the
// LAYER AND SHAPE, never of the tile: a convolution these kernels take (conv_rows_kind) runs on its one tile whatever the
// caller forces or the autotuner measures, so logits stay bit-identical across tiles.  Everything else is the generic f16x2
// arithmetic, instruction for instruction: per 16x16 tile and K-step P.X0, Q.X0, (P 2^-11).X1 on v_mfma_f32_16x16x32_f16
// into one chain that joins the running f32 sum every eighth K-step; f32 BN + ReLU epilogue through a per-wave LDS
// transpose, whole 256-byte row segments stored.
//
// LDS: three row slots (slot = kh: a channel block's three kernel rows; tile 20: four, its four input rows) of 144 pixels x 128 bytes, their 16-byte chunks
// rotated by the pixel index so that a fragment block may start at ANY pixel without bank conflicts (row_off), and two or
// three stages of three weight panels (one per kw).
#include <atomic>

#include "nbc_kernels.hpp"
#include "split16.hpp"

namespace nbc {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOutOfRange = 0x80000000u;
constexpr int kRowPx = 144;                         // pixels of a row slot: 128 + 2 * dil, dil <= 8, in whole 8-pixel DMAs
constexpr int kRowBytes = kRowPx * 128;
constexpr int kRowParts = kRowPx / 8 / 3;           // LDS-DMAs (8 pixels each) of one row per K-step: a row in three parts
static_assert(kRowParts * 24 == kRowPx && kRowParts % 2 == 0, "a row slot is three even parts of whole 8-pixel DMAs");

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void dma16_buf(unsigned voff, rsrc_t rsrc, unsigned lds_base, unsigned soff) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %3 offen lds"
      :
      : "v"(voff), "s"(rsrc), "s"(lds_base), "s"(soff)
      : "memory");
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
// Tile 20: the output rows of an image in pairs (oy, oy + dil): whole groups of 2 dil rows hold dil pairs each (pair q of group g: rows
// g 2 dil + q and + dil), a last group of fewer rows one pair per row of its first half (the second row of such a pair may lie below
// the image: computed on zero rows, not stored).  Pair index -> first row: (pr / dil) 2 dil + pr % dil in both cases.
__host__ __device__ inline int rowstep_pairs(int Ho, int dil) {
  const int groups = Ho / (2 * dil), rem = Ho - groups * 2 * dil;
  return groups * dil + (rem < dil ? rem : dil);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue (conv_igemm_dma.hip's f16x2 path without identity, instruction for instruction): BN on the
// accumulators into a per-wave f32 scratch in the idle ring, read back row-wise, ReLU, split, whole row segments stored.
// acc16[j][i]: lane (r16, q16) holds pixel i*16 + r16 and channels j*16 + 4*q16 .. +3 of the wave's (MT*32) x (NT*32) tile.
template <int CW, int MT, int NT, int TABLE_OFF>
__device__ __forceinline__ void rows_epilogue(const ConvArgs& p, unsigned char* smem, f32x4 (&acc16)[2 * NT][2 * MT], int wave, int lane,
                                              int wm, int wn, int m0, int n0) {
  constexpr int NT16 = 2 * NT;
  constexpr int SLAB_CH = NT * 32;
  constexpr int PITCH = SLAB_CH * 4 + 16;
  constexpr int CPR = SLAB_CH / 8;                  // lanes per pixel row: 8 channels (an h0 chunk and an h1 chunk) each
  constexpr int PIX_PER_PASS = 64 / CPR;
  constexpr int PASSES = 32 / PIX_PER_PASS;
  static_assert(CW * 32 * PITCH <= TABLE_OFF, "epilogue scratch must fit below the scale/shift table");
  const int r16 = lane & 15, q16 = lane >> 4;
  const int o_pix = lane / CPR, o_chunk = lane % CPR;
  const int n_slab = n0 + wn * SLAB_CH;
  const unsigned row_bytes = (unsigned)p.Co * 4u;
  unsigned char* ytile = static_cast<unsigned char*>(p.y) + ((size_t)m0 * p.Co + n_slab) * 4;
  const int rows_valid = p.M - m0;
  const int row0 = wm * MT * 32 + o_pix;
  const unsigned lane_chunk = (unsigned)(o_chunk >> 2) * 128u + (unsigned)(o_chunk & 3) * 16u;
  unsigned char* scr = smem + wave * (32 * PITCH);
  const unsigned char* table = smem + TABLE_OFF + wn * SLAB_CH * 4;
  const bool relu = p.relu != 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j0 = 0; j0 < NT16; j0 += 4) {
      float4 sc[4], sh[4];
#pragma unroll
      for (int jj = 0; jj < 4 && j0 + jj < NT16; ++jj) {
        const int nl = (j0 + jj) * 16 + 4 * q16;
        sc[jj] = *reinterpret_cast<const float4*>(table + nl * 4);
        sh[jj] = *reinterpret_cast<const float4*>(table + 1024 + nl * 4);
      }
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int jj = 0; jj < 4 && j0 + jj < NT16; ++jj) {
          const int nl = (j0 + jj) * 16 + 4 * q16;
          const f32x4 a = acc16[j0 + jj][2 * i + i2];
          float4 v;
          v.x = __builtin_fmaf(a[0], sc[jj].x, sh[jj].x);
          v.y = __builtin_fmaf(a[1], sc[jj].y, sh[jj].y);
          v.z = __builtin_fmaf(a[2], sc[jj].z, sh[jj].z);
          v.w = __builtin_fmaf(a[3], sc[jj].w, sh[jj].w);
          *reinterpret_cast<float4*>(scr + (i2 * 16 + r16) * PITCH + nl * 4) = v;
        }
    }
    float v[PASSES][8];
#pragma unroll
    for (int ps2 = 0; ps2 < PASSES; ++ps2) {
      const float4* sp = reinterpret_cast<const float4*>(scr + (ps2 * PIX_PER_PASS + o_pix) * PITCH + o_chunk * 32);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 t4 = sp[q];
        v[ps2][4 * q] = t4.x; v[ps2][4 * q + 1] = t4.y; v[ps2][4 * q + 2] = t4.z; v[ps2][4 * q + 3] = t4.w;
      }
    }
#pragma unroll
    for (int ps2 = 0; ps2 < PASSES; ++ps2) {
      const int row = row0 + i * 32 + ps2 * PIX_PER_PASS;
      const unsigned loff = (unsigned)row * row_bytes + lane_chunk;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[ps2][e] = __builtin_elementwise_maximum(v[ps2][e], 0.f);
      }
      uint4 o, o1;
      split16x8(v[ps2], o, o1);
      if (row < rows_valid) {
        *reinterpret_cast<uint4*>(ytile + loff + 64) = o1;
        *reinterpret_cast<uint4*>(ytile + loff) = o;
      }
    }
  }
}

// 128 pixels (an image row; a 128-pixel segment of a wider one) x BN = WN * NT * 32 channels per block: 2 x WN MFMA waves of
// 64 x (NT * 32) and four loader waves; three row slots of 18 KiB and SB stages of 3 x BN x 128 bytes of weights: a stage is
// refilled a whole row-step (three K-steps of MFMAs) ahead.
//   tile 18: WN 4 -> 128 channels, eight MFMA waves, SB 2 (150 KiB): 256 output channels or more;
//   tile 19: WN 2 ->  64 channels, four MFMA waves,  SB 3 (126 KiB): the 64 / 128-channel layers, whose K loop ran at one
//            LDS-DMA round trip per K-step (DMA and barriers alone: 19-21 of their 21-25 us).
//   tile 20: OR 2, WN 2 -> TWO output rows (oy and oy + dilation: four input rows between them instead of six) x 64 channels,
//            eight MFMA waves (output row x pixel half x channel half), SB 2 (122 KiB): per output row and channel block 72 KiB
//            of pixels and weights where tile 18 moves 99; the same K order per output, so the same bits as tile 18.
template <int WN, int NT, int SB, int OR>
__global__ __launch_bounds__((2 * WN * OR + 4) * 64, (2 * WN * OR + 4) / 4) void conv3x3_rowstep_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(OR == 1 || (OR == 2 && SB == 2), "two output rows per block: every row-step's DMAs are waited for in full");
  constexpr int WM = 2, MT = 2, CW = WM * WN * OR, NW = 4;   // 2 x WN (x OR) MFMA waves of 64 x (NT * 32), four loader waves
  constexpr int BN = WN * NT * 32;
  // pixel-row slots.  OR 1: slot = kh.  OR 2: slot k = input row oy + (k - 1) dil; row-step kh reads slots kh (first output row) and
  // kh + 1 (second): slots 0 and 1 are requested with kh 0's weights, 2 with kh 1's, 3 with kh 2's, each a row-step ahead, into a slot
  // last read two row-steps before.  (Rows requested TWO row-steps ahead through a ring of six slots: the same times,
  // profiles/r05_rowstep_two_output_rows.log.)
  constexpr int NSLOT = OR == 2 ? 4 : 3;
  constexpr int A_SLOT = kRowBytes, A_REGION = NSLOT * A_SLOT;
  constexpr int B_TAP = BN * 128, B_STEP = 3 * B_TAP;
  constexpr int TABLE_OFF = A_REGION + SB * B_STEP;   // the ring; the epilogue scratch (18 / 36 KiB) lies inside it
  constexpr int NA = kRowPx / 8;                       // 18 pixel DMAs per row-step
  constexpr int LA_HI = (NA + NW - 1) / NW, LA_LO = NA / NW;
  constexpr int NBW = (BN / 8) / NW;                  // weight DMAs per loading wave and tap
  constexpr int LB = 3 * NBW;                         // ... and row-step
  constexpr int MT16 = 2 * MT, NT16 = 2 * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave >= CW;
  const int lw = __builtin_amdgcn_readfirstlane((wave - CW) & (NW - 1));
  const bool la_hi = lw < NA % NW;

  const int dil = p.dil;
  // OR 2: the output rows of an image in pairs (oy, oy + dil): rowstep_pairs
  const int pairs = OR == 2 ? rowstep_pairs(p.Ho, dil) : p.Ho;
  const int NH = p.N * pairs;
  const int segs = p.Wo / 128;                         // 128-pixel segments per row
  const int tiles_n = p.Co / BN;
  const int tiles_m = NH * segs;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int n0 = tile_n * BN;
  const int R = tile_m / segs, seg = tile_m - R * segs;
  const int img = R / pairs, pr = R - img * pairs;
  const int oy = OR == 2 ? (pr / dil) * 2 * dil + pr % dil : pr;    // the (first) output row

  const rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const int pix_bytes = p.Ci * 4;
  const int cblocks = pix_bytes / 128;
  const int T = p.ksteps, RS = 3 * cblocks;
  const unsigned wrow_bytes = (unsigned)T * 128u;
  const unsigned tap_stride = (unsigned)cblocks * 128u;
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned smem_base = (unsigned)(size_t)(lds_u8*)smem;

  if (wave == CW && lane < BN / 4) {
    dma16(p.scale + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF);
    dma16(p.shift + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF + 1024u);
  }
  if (is_loader) {
    const int lp = lane >> 3, ps = lane & 7;
    unsigned w_off[NBW], a_col[LA_HI];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int row = (lw + NW * i) * 8 + lp;
      w_off[i] = (unsigned)(n0 + row) * wrow_bytes + (unsigned)(ps ^ ((row >> 1) & 7)) * 16u;
    }
#pragma unroll
    for (int d = 0; d < LA_HI; ++d) {
      const int pp = (lw + NW * d) * 8 + lp;            // slot pixel
      const int ix = seg * 128 + pp - dil;
      const unsigned chunk = (unsigned)((ps - 2 * ((pp >> 1) & 3)) & 7) * 16u;
      a_col[d] = (unsigned)ix < (unsigned)p.Wi ? (unsigned)ix * (unsigned)pix_bytes + chunk : kOutOfRange;
    }
    const int la = la_hi ? LA_HI : LA_LO;
    int i_cb = 0, i_kh = 0, i_sb = 0;                   // the row-step issued next: channel block, kh (= its pixel slot), weight stage
    auto issue_row = [&](int k, int cb, int slot) __attribute__((always_inline)) {     // input row oy + (k - 1) dil of channel block cb
      const int iy = oy + (k - 1) * dil;
      const bool rowok = (unsigned)iy < (unsigned)p.Hi;
      const unsigned rowoff = (unsigned)((img * p.Hi + iy) * p.Wi) * (unsigned)pix_bytes;
#pragma unroll
      for (int d = 0; d < LA_HI; ++d)
        if (d < la)
          dma16_buf(rowok ? a_col[d] + rowoff : kOutOfRange, xrsrc, smem_base + (unsigned)(slot * A_SLOT) + (unsigned)(lw + NW * d) * 1024u,
                    (unsigned)cb * 128u);
    };
    auto issue_rowstep = [&]() __attribute__((always_inline)) {
      if constexpr (OR == 1) issue_row(i_kh, i_cb, i_kh);
      else if (i_kh == 0) { issue_row(0, i_cb, 0); issue_row(1, i_cb, 1); }
      else issue_row(i_kh + 1, i_cb, i_kh + 1);
      const unsigned soff = ((unsigned)(3 * i_kh) * (unsigned)cblocks + (unsigned)i_cb) * 128u;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int i = 0; i < NBW; ++i)
          dma16_buf(w_off[i], wrsrc, smem_base + (unsigned)(A_REGION + i_sb * B_STEP + kw * B_TAP) + (unsigned)(lw + NW * i) * 1024u,
                    soff + (unsigned)kw * tap_stride);
      if (++i_kh == 3) { i_kh = 0; ++i_cb; }
      if (++i_sb == SB) i_sb = 0;
    };
    // SB - 1 row-steps ahead (the pixel slots, three of them, never run short)
    int issued = 0;
#pragma unroll
    for (int k = 0; k < SB - 1; ++k)
      if (issued < RS) { issue_rowstep(); ++issued; }
    for (int rs = 0; rs < RS; ++rs) {
      // this wave's DMAs of row-step rs have landed when only the younger row-steps' (SB - 2 of them) are outstanding
      const int younger = issued - rs - 1;
      if (younger <= 0) wait_vmcnt<0>();
      else if (SB == 3 && younger == 1) { if (la_hi) wait_vmcnt<LB + LA_HI>(); else wait_vmcnt<LB + LA_LO>(); }
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (issued < RS) { issue_rowstep(); ++issued; }   // into the stage row-step rs-1 used: its reads retired at this barrier
    }
    __syncthreads();
    return;
  }

  // ---- MFMA waves
  const int r16 = lane & 15, q16 = lane >> 4;
  const int wm = wave % WM, orow = OR == 2 ? (wave / WM) & 1 : 0, wn = wave / (WM * OR);
  auto row_off = [](int pix, int chunk) { return pix * 128 + (((chunk + 2 * ((pix >> 1) & 3)) & 7) << 4); };
  unsigned a_rd[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int pix = wm * 64 + kw * dil + r16;
    a_rd[kw][0] = (unsigned)row_off(pix, q16);
    a_rd[kw][1] = (unsigned)row_off(pix, 4 + q16);
  }
  const unsigned b_rd0 = (unsigned)(A_REGION + lds_off(wn * NT * 32 + r16, q16));
  const unsigned b_rd1 = (unsigned)(A_REGION + lds_off(wn * NT * 32 + r16, 4 + q16));
  f32x4 acc16[NT16][MT16], accI2[NT16][MT16];
#pragma unroll
  for (int j = 0; j < NT16; ++j)
#pragma unroll
    for (int i = 0; i < MT16; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc16[j][i][e] = 0.f; accI2[j][i][e] = 0.f; }
  const f16x8 kLow = {kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH};
  unsigned a_off = (unsigned)(orow * A_SLOT), b_off = 0;   // byte offsets of the row-step's pixel slot (kh [+ 1]) and weight stage
  int t = 0;
  for (int rs = 0; rs < RS; ++rs) {
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kw = 0; kw < 3; ++kw, ++t) {
      if (t > 0 && (t & 7) == 0) {                     // the chain of the last eight K-steps joins the sum
#pragma unroll
        for (int n = 0; n < NT16 * MT16; ++n) {
          acc16[n / MT16][n % MT16] += accI2[n / MT16][n % MT16];
#pragma unroll
          for (int e = 0; e < 4; ++e) accI2[n / MT16][n % MT16][e] = 0.f;
        }
      }
      uint4 xp0[MT16], xp1[MT16], xw0[NT16], xw1[NT16];
#pragma unroll
      for (int i = 0; i < MT16; ++i) {
        xp0[i] = *reinterpret_cast<const uint4*>(smem + (a_rd[kw][0] + a_off) + i * 2048);
        xp1[i] = *reinterpret_cast<const uint4*>(smem + (a_rd[kw][1] + a_off) + i * 2048);
      }
#pragma unroll
      for (int j = 0; j < NT16; ++j) {
        xw0[j] = *reinterpret_cast<const uint4*>(smem + (b_rd0 + b_off) + (kw * B_TAP + j * 2048));
        xw1[j] = *reinterpret_cast<const uint4*>(smem + (b_rd1 + b_off) + (kw * B_TAP + j * 2048));
      }
      constexpr int NTI = NT16 * MT16;
#pragma unroll
      for (int idx = 0; idx < 3 * NTI; ++idx) {
        const int prod = idx / NTI, n = idx % NTI, j = n / MT16, i = n % MT16;
        if (prod == 0)
          accI2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xw0[j]), __builtin_bit_cast(f16x8, xp0[i]), accI2[j][i], 0, 0, 0);
        else if (prod == 1)
          accI2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xw1[j]), __builtin_bit_cast(f16x8, xp0[i]), accI2[j][i], 0, 0, 0);
        else {
          if (i == 0) xw0[j] = __builtin_bit_cast(uint4, __builtin_bit_cast(f16x8, xw0[j]) * kLow);
          accI2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xw0[j]), __builtin_bit_cast(f16x8, xp1[i]), accI2[j][i], 0, 0, 0);
        }
      }
    }
    a_off = a_off == (unsigned)((2 + orow) * A_SLOT) ? (unsigned)(orow * A_SLOT) : a_off + (unsigned)A_SLOT;
    b_off = b_off == (unsigned)(SB - 1) * B_STEP ? 0u : b_off + (unsigned)B_STEP;
  }
#pragma unroll
  for (int n = 0; n < NT16 * MT16; ++n) acc16[n / MT16][n % MT16] += accI2[n / MT16][n % MT16];
  __syncthreads();
  const int oyw = oy + orow * dil;                     // this wave's output row
  if (oyw < p.Ho)                                      // (the second row of a last, odd pair lies below the image: nothing to store)
    rows_epilogue<CW, MT, NT, TABLE_OFF>(p, smem, acc16, wave, lane, wm, wn, ((img * p.Ho + oyw) * segs + seg) * 128, n0);
}

template <int WN, int NT, int SB, int OR = 1>
hipError_t launch_rowstep_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int BN = WN * NT * 32;
  constexpr int smem = (OR == 2 ? 4 : 3) * kRowBytes + SB * 3 * BN * 128 + 2048;
  static std::atomic<unsigned long long> attr_done{0};
  auto kern = &conv3x3_rowstep_kernel<WN, NT, SB, OR>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << dev, std::memory_order_release);
  }
  if (a.Co % BN != 0 || a.Wo % 128 != 0) return hipErrorInvalidValue;
  const int pairs = OR == 2 ? rowstep_pairs(a.Ho, a.dil) : a.Ho;
  const int tiles = a.N * pairs * (a.Wo / 128) * (a.Co / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3((2 * WN * OR + 4) * 64), smem, s, a);
  return hipGetLastError();
}

}  // namespace

// Whether a convolution runs on the kernels of this file (f16x2): 3x3, stride 1, padding = dilation <= 8, no identity, and
//   kind 1: 128-pixel-wide maps, 256 output channels or more (layer3 / layer4 conv2, classifier.0 of a 1024-pixel-wide image):
//           tile 18 or tile 20 (the same K order and bits; the cost model's default is 20);
//   kind 2: 128-pixel-wide maps, 64 or 128 output channels (layer2.1-3 conv2): tile 19;
//   0: neither (the generic kernel).  A property of the layer and its shape: the K order follows from it (the head of this file).
int conv_rows_kind(int precision, int k, int stride, int pad, int dil, int Hi, int Wi, int Ho, int Wo, int Ci, int Co, bool has_res) {
  if (!(precision == 2 && k == 3 && stride == 1 && pad == dil && dil >= 1 && dil <= 8 && Ho == Hi && Wo == Wi && Ci % 32 == 0 && !has_res)) return 0;
  if (Wi == 128 && Co >= 256 && Co % 128 == 0) return 1;
  // (maps of 256 pixels -- layer1's conv2, two segments per row -- run on this kernel as well, and no faster than on the generic
  // tiles, which keep two blocks per CU there: profiles/r05_rowstep_kernel_layer1_layer2.log; left to them)
  if (Wi == 128 && Co % 64 == 0 && Co < 256) return 2;
  return 0;
}

// rows_tile (tile id - 18): 0 = one image row x 128 channels (eight 64x32 MFMA waves + four loader waves; kind 1),
// 1 = one image row x 64 channels (four MFMA + four loader waves; kind 2), 2 = two image rows x 64 channels (eight + four; kind 1)
hipError_t launch_conv3x3_rows(const ConvArgs& a, int rows_tile, hipStream_t s) {
  if (a.x_bytes == 0 || a.x_bytes >= kOutOfRange || a.w_bytes == 0 || a.w_bytes >= kOutOfRange) return hipErrorInvalidValue;
  if (a.stem || a.KH != 3 || a.KW != 3 || a.ksteps != 9 * (a.Ci * 4 / 128) ||
      a.M != a.N * a.Ho * a.Wo)
    return hipErrorInvalidValue;
  const int kind = conv_rows_kind(2, a.KH, a.stride, a.pad, a.dil, a.Hi, a.Wi, a.Ho, a.Wo, a.Ci, a.Co, a.res != nullptr);
  if (rows_tile == 0 && kind == 1) return launch_rowstep_cfg<4, 1, 2>(a, s);
  if (rows_tile == 1 && kind == 2) return launch_rowstep_cfg<2, 1, 3>(a, s);
  if (rows_tile == 2 && kind == 1) return launch_rowstep_cfg<2, 1, 2, 2>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace nbc
