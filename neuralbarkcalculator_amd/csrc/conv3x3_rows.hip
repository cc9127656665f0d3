// 3x3 stride-1 convolution on 128-pixel-wide feature maps, f16x2 (NBC_PREC_F16X2), whose pixel rows STAY IN LDS for the three
// taps of a kernel row: the long-K layers of the path (layer3 / layer4 conv2, classifier.0).
//
// Why.  The generic kernel (conv_igemm_dma.hip) fetches, for every K-step (tap, 32-channel block), the 128 pixel rows of its
// tile again: 16 KiB of pixels + 16-32 KiB of weights per step from L2 into LDS.  Measured in the network (profiles/
// r05_f16x2_kloop_ablations_in_network.log): the head conv's K loop WITHOUT a single MFMA or fragment read -- LDS-DMA and
// barriers only -- takes 423 of its 738 us (7.25 GB at 17 TB/s, the L2 -> LDS gather rate of this chip), the same loop without
// its DMAs 496: the L2 -> LDS stream is a bottleneck of its own beside the matrix pipe, and the chip's power limit couples
// the two.  The three taps (kh, kw = 0, 1, 2) of a kernel row read the SAME input row shifted by the dilation, so here a
// K-step's pixels are not a tile of their own: an input row (128 + 2 dil pixels x 128 bytes of one channel block, zero halo
// from the buffer resource's range check) is fetched ONCE per (channel block, kh) into a row slot and read by the three
// K-steps (kw) at a shifted pixel index.  Pixel traffic falls to a third, and a block of TWO image rows x 128 channels
// (256 x 128 outputs, the generic 128 x 256 tile's arithmetic) fetches 12 + 16 KiB per K-step where that tile fetches 48.
//
// K order.  (channel block, kh, kw) -- the generic kernel walks (kh, kw, channel block).  The order is a property of the
// LAYER AND SHAPE, never of the tile: a convolution this kernel is eligible for (rows_eligible) runs on one of its tiles
// whatever the caller forces or the autotuner measures, so logits stay bit-identical across tiles.  Everything else is
// the generic f16x2 arithmetic, instruction for instruction: per 16x16 tile and K-step P.X0, Q.X0, (P 2^-11).X1 on
// v_mfma_f32_16x16x32_f16 into one chain that joins the running f32 sum every eighth K-step; f32 BN + ReLU epilogue
// through a per-wave LDS transpose, whole 256-byte row segments stored.
//
// LDS: three row slots (slot = kh: a channel block's three kernel rows) of PR rows x 144 pixels x 128 bytes, their 16-byte
// chunks rotated by the pixel index so that a fragment block may start at ANY pixel without bank conflicts (row_off), and a
// three-stage ring of weight panels (stage = kw).  The
// rows of (channel block, kh) are issued in three parts, one with each of the three K-steps of the row-step BEFORE, so
// every K-step issues the same number of LDS-DMAs and the counted s_waitcnt of the generic pipeline carries over.
#include <atomic>
#include <type_traits>

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
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int rows_lds_bytes(int wm, int wn, int mt, int nt, int s) {
  const int pr = wm * mt * 32 / 128;
  const int ring = 3 * pr * kRowBytes + s * (wn * nt * 32) * 128;
  const int scratch = wm * wn * 32 * (nt * 32 * 4 + 16);
  return ring > scratch ? ring : scratch;
}

// Epilogue of both kernels (conv_igemm_dma.hip's f16x2 path without identity, instruction for instruction): BN on the
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

// Tile = PR image rows (128 pixels each) x (WN*NT*32) channels; WM*WN waves of (MT*32) x (NT*32), each of which also issues
// its share of the LDS-DMAs (64 x 64 wave tiles fill the register file: no room for loader waves beside them).  Three weight
// stages, one barrier per K-step.  (The 128 x 128 tile with loader waves ran on this kernel too; one barrier per row-step --
// conv3x3_rowstep_kernel below -- is 4-6 % faster there: profiles/r05_rows_tile18_one_barrier_per_rowstep.log.)
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(WM * WN * 64, WM * WN / 4) void conv3x3_rows_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int S = 3;
  constexpr int CW = WM * WN;
  constexpr int NW = CW;                            // waves that share the loading: all of them
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int PR = BM / 128;
  static_assert(BM % 128 == 0 && PR >= 1 && PR <= 2, "the tile is one or two image rows");
  static_assert((MT * 32) <= 128 && 128 % (MT * 32) == 0, "a wave's pixels lie in one image row");
  constexpr int A_SLOT = PR * kRowBytes;
  constexpr int A_REGION = 3 * A_SLOT;
  constexpr int B_BYTES = BN * 128;
  constexpr int TABLE_OFF = rows_lds_bytes(WM, WN, MT, NT, S);
  constexpr int NB = BN / 8;                        // weight DMAs (8 rows each) per K-step
  static_assert(NB % NW == 0, "weight rows must split evenly over the loading waves");
  constexpr int LB = NB / NW;                       // ... per loading wave
  constexpr int NA = PR * kRowParts;                // pixel-row DMAs per K-step (a third of every row of the next row-step)
  constexpr int LA_HI = (NA + NW - 1) / NW, LA_LO = NA / NW;   // ... per loading wave: waves below NA % NW take one more
  constexpr int MT16 = 2 * MT, NT16 = 2 * NT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lw = wave;                                // index among the loading waves
  const bool la_hi = (NA % NW != 0) && lw < NA % NW;   // this wave issues LA_HI pixel-row DMAs per K-step (else LA_LO)
  // the second half of the MFMA waves runs one barrier late (conv_igemm_dma.hip, STAGGER): reads of one half under the
  // MFMAs of the other; the late half at the higher priority
  constexpr bool STAGGER = CW >= 8;
  const bool late_half = STAGGER && wave >= CW / 2;

  // ---- tile coordinates: blocks that share an XCD take a contiguous range of tiles, channel tiles fastest
  const int NH = p.N * p.Ho;                          // image rows of the batch, flattened
  const int tiles_n = p.Co / BN;
  const int tiles_m = (NH + PR - 1) / PR;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int m0 = tile_m * BM;                         // first output pixel of the tile (rows are 128 pixels)
  const int n0 = tile_n * BN;
  const int R0 = tile_m * PR;                         // first flattened image row

  const rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const int pix_bytes = p.Ci * 4;
  const int cblocks = pix_bytes / 128;                // 32-channel blocks
  const int T = p.ksteps;                             // 9 * cblocks
  const unsigned wrow_bytes = (unsigned)T * 128u;
  const int dil = p.dil;

  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned smem_base = (unsigned)(size_t)(lds_u8*)smem;

  // ---- loader geometry.  One LDS-DMA = 64 lanes x 16 bytes = 8 consecutive LDS rows of 128 bytes; lane -> row lp of
  // the eight, physical chunk slot ps; the swizzles sit on the source side (slot ps holds the logical chunk the reader
  // expects there: see lds_off for the weight panels and row_off for the pixel rows)
  const int lp = lane >> 3, ps = lane & 7;
  // weights: DMA b covers panel rows 8b .. 8b+7; this wave takes b = lw + NW * i
  unsigned w_off[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int row = (lw + NW * i) * 8 + lp;
    w_off[i] = (unsigned)(n0 + row) * wrow_bytes + (unsigned)(ps ^ ((row >> 1) & 7)) * 16u;
  }
  // pixel rows: DMA (row r, group c8) covers slot pixels 8 c8 .. 8 c8 + 7 = input columns 8 c8 - dil ..: per issue slot d and
  // third of the row the lane's byte offset behind the row's first pixel (columns outside the image: out of range = zeros)
  unsigned a_col[LA_HI][3];
#pragma unroll
  for (int d = 0; d < LA_HI; ++d) {
    const int jj = lw + NW * d;                       // pixel-row DMA of a K-step this slot issues (wave-uniform)
    const int g = jj % kRowParts;
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      const int pp = (part * kRowParts + g) * 8 + lp;           // slot pixel
      const int ix = pp - dil;
      const unsigned chunk = (unsigned)((ps - 2 * ((pp >> 1) & 3)) & 7) * 16u;   // logical chunk of physical slot ps (row_off)
      a_col[d][part] = (unsigned)ix < (unsigned)p.Wi ? (unsigned)ix * (unsigned)pix_bytes + chunk : kOutOfRange;
    }
  }
  // the PR image rows of the tile: image and row inside it (wave-uniform)
  int r_oy[PR], r_base[PR];
  bool r_ok[PR];
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    const int R = R0 + r;
    r_ok[r] = R < NH;
    const int img = R / p.Ho;
    r_oy[r] = R - img * p.Ho;
    r_base[r] = img * p.Hi;                           // flattened input row of the image's row 0
  }

  // issue state (wave-uniform): the K-step whose weights go out next is (i_cb, i_kh, kw) with kw a compile-time argument of
  // the issue functions (T is a multiple of 9: K-step t has kw = t % 3 and its weights live in stage t % 3 = kw); the
  // pixel rows that go with it belong to the NEXT row-step (channel block, kh): the third `kw` of each of them
  int i_cb = 0, i_kh = 0;
  unsigned i_soff = 0;                                // weight K offset of (i_cb, i_kh, kw = 0): ((3 i_kh) cblocks + i_cb) * 128
  const unsigned tap_stride = (unsigned)cblocks * 128u;
  auto issue_a = [&](int cbn, int khn, auto PARTc, int d) __attribute__((always_inline)) {
    constexpr int PART = decltype(PARTc)::value;
    const int jj = lw + NW * d;
    const int r = jj / kRowParts;                     // (wave-uniform, not a compile-time constant: selected, never indexed)
    const int c8 = PART * kRowParts + (jj - r * kRowParts);
    const bool second = PR > 1 && r != 0;
    const int oy = second ? r_oy[PR - 1] : r_oy[0];
    const int base = second ? r_base[PR - 1] : r_base[0];
    const bool rok = second ? r_ok[PR - 1] : r_ok[0];
    const int iy = oy + (khn - 1) * dil;
    const bool rowok = rok && (unsigned)iy < (unsigned)p.Hi;
    const unsigned rowoff = (unsigned)((base + iy) * p.Wi) * (unsigned)pix_bytes;
    const unsigned off = rowok ? a_col[d][PART] + rowoff : kOutOfRange;      // (an out-of-range column stays out of range: no wrap below 4 GiB)
    dma16_buf(off, xrsrc, smem_base + (unsigned)(khn * A_SLOT + r * kRowBytes) + (unsigned)c8 * 1024u, (unsigned)cbn * 128u);
  };
  // one DMA of this wave's share of the K-step (i_cb, i_kh, KW): d < la pixel-row DMAs, then LB weight DMAs
  auto issue_one = [&](auto KWc, int d, int la) __attribute__((always_inline)) {
    constexpr int KW = decltype(KWc)::value;
    if (d < LA_HI) {
      const int khn = i_kh == 2 ? 0 : i_kh + 1;
      const int cbn = i_kh == 2 ? i_cb + 1 : i_cb;
      if (d < la && cbn < cblocks) issue_a(cbn, khn, KWc, d);
    } else {
      const int i = d - LA_HI;
      dma16_buf(w_off[i], wrsrc, smem_base + (unsigned)(A_REGION + KW * B_BYTES) + (unsigned)(lw + NW * i) * 1024u,
                i_soff + (unsigned)KW * tap_stride);
    }
  };
  constexpr int L = LA_HI + LB;                       // issue slots per K-step (slot d < LA_HI may be empty on a wave)
  const int la_mine = la_hi ? LA_HI : LA_LO;
  auto issue_part = [&](auto KWc, int part) __attribute__((always_inline)) {   // the K-step's DMAs in four parts, behind the MFMA clusters
    constexpr int KW = decltype(KWc)::value;
#pragma unroll
    for (int d = 0; d < L; ++d)
      if (d * 4 / L == part) issue_one(KWc, d, la_mine);
    if (part == 3 && KW == 2) {                       // the row-step is issued: the next one
      if (++i_kh == 3) { i_kh = 0; ++i_cb; i_soff = (unsigned)i_cb * 128u; }
      else i_soff += 3u * tap_stride;
    }
  };
  auto issue_step = [&](auto KWc) __attribute__((always_inline)) {
#pragma unroll
    for (int part = 0; part < 4; ++part) issue_part(KWc, part);
  };
  typedef std::integral_constant<int, 0> K0;
  typedef std::integral_constant<int, 1> K1;
  typedef std::integral_constant<int, 2> K2;

  // ---- MFMA geometry
  const int r16 = lane & 15, q16 = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int w_row = (wm * MT * 32) / 128;             // image row of the tile this wave's pixels lie in
  const int w_px0 = (wm * MT * 32) % 128;             // its first pixel there
  // Pixel-row swizzle: a K-step reads slot pixels (column + kw * dil), so a 16-pixel fragment block starts ANYWHERE, and the
  // generic XOR swizzle is conflict-free only for blocks that start at multiples of 16.  Here slot pixel p keeps logical
  // chunk c at physical slot (c + 2 ((p >> 1) & 3)) & 7: the two chunk classes a ds_read_b128 lane group mixes (c and c + 1)
  // never meet, and inside a class any eight consecutive pixels take eight different (pixel parity, slot) places.
  // Independent of the block index i (16 pixels on: the same slot), so block i is an immediate offset.
  auto row_off = [](int pix, int chunk) { return pix * 128 + (((chunk + 2 * ((pix >> 1) & 3)) & 7) << 4); };
  unsigned a_rd[3][2];                                // per kw: byte address (behind the row slot) of chunk q16 / 4 + q16 of block 0
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int pix = w_px0 + kw * dil + r16;
    a_rd[kw][0] = (unsigned)(w_row * kRowBytes + row_off(pix, q16));
    a_rd[kw][1] = (unsigned)(w_row * kRowBytes + row_off(pix, 4 + q16));
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

  auto x2_flush = [&](int t) __attribute__((always_inline)) {
    if (t > 0 && (t & 7) == 0) {                       // wave-uniform: the chain of the last eight K-steps joins the sum
#pragma unroll
      for (int n = 0; n < NT16 * MT16; ++n) {
        acc16[n / MT16][n % MT16] += accI2[n / MT16][n % MT16];
#pragma unroll
        for (int e = 0; e < 4; ++e) accI2[n / MT16][n % MT16][e] = 0.f;
      }
    }
  };
  // fragments of a K-step (cb, kh, KW): pixels from row slot kh (byte offset kh_off) at slot pixel (column + KW * dil),
  // weights from stage KW
  auto x2_read = [&](auto KWc, unsigned kh_off, uint4 (&xp0)[MT16], uint4 (&xp1)[MT16], uint4 (&xw0)[NT16], uint4 (&xw1)[NT16]) __attribute__((always_inline)) {
    constexpr int KW = decltype(KWc)::value;
    const unsigned char* a0 = smem + (a_rd[KW][0] + kh_off);
    const unsigned char* a1 = smem + (a_rd[KW][1] + kh_off);
#pragma unroll
    for (int i = 0; i < MT16; ++i) {
      xp0[i] = *reinterpret_cast<const uint4*>(a0 + i * 2048);
      xp1[i] = *reinterpret_cast<const uint4*>(a1 + i * 2048);
    }
#pragma unroll
    for (int j = 0; j < NT16; ++j) {
      xw0[j] = *reinterpret_cast<const uint4*>(smem + b_rd0 + (KW * B_BYTES + j * 2048));
      xw1[j] = *reinterpret_cast<const uint4*>(smem + b_rd1 + (KW * B_BYTES + j * 2048));
    }
  };
  // the MFMAs of a K-step; with do_issue the DMAs of the K-step that is issued at this point of the loop (two steps ahead
  // of the loop step: kw KI) ride behind its four quarters
  auto x2_mfma = [&](auto KIc, bool do_issue, uint4 (&xp0)[MT16], uint4 (&xp1)[MT16], uint4 (&xw0)[NT16], uint4 (&xw1)[NT16]) __attribute__((always_inline)) {
    typedef decltype(KIc) KI;
    constexpr int NTI = NT16 * MT16;
    const f16x8 kLow = {kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH};
#pragma unroll
    for (int idx = 0; idx < 3 * NTI; ++idx) {
      const int prod = idx / NTI, n = idx % NTI, j = n / MT16, i = n % MT16;
      if (prod == 0)
        accI2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xw0[j]), __builtin_bit_cast(f16x8, xp0[i]), accI2[j][i], 0, 0, 0);
      else if (prod == 1)
        accI2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xw1[j]), __builtin_bit_cast(f16x8, xp0[i]), accI2[j][i], 0, 0, 0);
      else {
        if (i == 0) xw0[j] = __builtin_bit_cast(uint4, __builtin_bit_cast(f16x8, xw0[j]) * kLow);      // P -> P 2^-11, in place
        accI2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xw0[j]), __builtin_bit_cast(f16x8, xp1[i]), accI2[j][i], 0, 0, 0);
      }
      if (do_issue && (idx + 1) % (3 * NTI / 4) == 0) issue_part(KI{}, (idx + 1) / (3 * NTI / 4) - 1);   // wave-uniform
    }
  };

  // ---- pipeline.  The BN scale/shift pairs of the block's channels go to LDS first (older than every ring DMA)
  if (wave == 0 && lane < BN / 4) {
    dma16(p.scale + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF);
    dma16(p.shift + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF + 1024u);
  }
  {
    // rows of row-step 0 (channel block 0, kh 0), whole; then K-steps 0 and 1 with their thirds of row-step 1
#pragma unroll
    for (int d = 0; d < LA_HI; ++d)
      if (d < la_mine) { issue_a(0, 0, K0{}, d); issue_a(0, 0, K1{}, d); issue_a(0, 0, K2{}, d); }
    issue_step(K0{});
    issue_step(K1{});
  }
  // top of K-step t: this wave's DMAs of step t (and every older one: the rows of its row-step among them) have landed when
  // only the younger step's are outstanding; the barrier makes every wave's share visible and retires the reads of the
  // slots about to be refilled.  The younger step t+1 carries pixel-row DMAs while a next row-step exists.
  const int RS = 3 * cblocks;                          // row-steps
  auto loop_top = [&](int t) __attribute__((always_inline)) {
    {
      if (t + 1 >= T) wait_vmcnt<0>();
      else if ((t + 1) / 3 + 1 < RS) {
        if (la_hi) wait_vmcnt<LB + LA_HI>();
        else wait_vmcnt<LB + LA_LO>();
      } else wait_vmcnt<LB>();
    }
    __builtin_amdgcn_s_barrier();
  };
  uint4 sp0[MT16], sp1[MT16], sw0[NT16], sw1[NT16];
  // One loop per role over the row-steps, the three K-steps (kw) of a row-step spelled out: kw is a compile-time constant
  // of every address.  T - 1 = 8 (mod 9): the last K-step, peeled below, is (kh 2, kw 2).
  if (!late_half) {
    unsigned kh_off = 0;
    for (int rs = 0; rs < RS; ++rs) {
      const int t = 3 * rs;
      loop_top(t);     x2_flush(t);     x2_read(K0{}, kh_off, sp0, sp1, sw0, sw1); x2_mfma(K2{}, t + 2 < T, sp0, sp1, sw0, sw1);
      loop_top(t + 1); x2_flush(t + 1); x2_read(K1{}, kh_off, sp0, sp1, sw0, sw1); x2_mfma(K0{}, t + 3 < T, sp0, sp1, sw0, sw1);
      if (rs + 1 < RS) {
        loop_top(t + 2); x2_flush(t + 2); x2_read(K2{}, kh_off, sp0, sp1, sw0, sw1); x2_mfma(K1{}, t + 4 < T, sp0, sp1, sw0, sw1);
      }
      kh_off = kh_off == 2u * A_SLOT ? 0u : kh_off + (unsigned)A_SLOT;
    }
  } else {
    // the late half: behind barrier t the MFMAs of step t-1 (fragments read in front of the barrier), then the reads of step t
    __builtin_amdgcn_s_setprio(1);
    unsigned kh_off = 0;
    loop_top(0);
    issue_step(K2{});
    x2_read(K0{}, kh_off, sp0, sp1, sw0, sw1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int rs = 0; rs < RS; ++rs) {
      const int t = 3 * rs;
      if (rs > 0) {
        loop_top(t); x2_flush(t - 1); x2_mfma(K2{}, t + 2 < T, sp0, sp1, sw0, sw1);   // (loop step t issues kw 2)
        x2_read(K0{}, kh_off, sp0, sp1, sw0, sw1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      loop_top(t + 1); x2_flush(t); x2_mfma(K0{}, t + 3 < T, sp0, sp1, sw0, sw1);   // (loop step t+1 issues kw 0)
      x2_read(K1{}, kh_off, sp0, sp1, sw0, sw1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (rs + 1 < RS) {
        loop_top(t + 2); x2_flush(t + 1); x2_mfma(K1{}, t + 4 < T, sp0, sp1, sw0, sw1);   // (loop step t+2 issues kw 1)
        x2_read(K2{}, kh_off, sp0, sp1, sw0, sw1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      kh_off = kh_off == 2u * A_SLOT ? 0u : kh_off + (unsigned)A_SLOT;
    }
    __builtin_amdgcn_s_setprio(0);
  }
  // last K-step (kh 2, kw 2), peeled: every DMA has retired
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (late_half) { x2_flush(T - 2); x2_mfma(K0{}, false, sp0, sp1, sw0, sw1); }       // the late half catches up on step T-2
  x2_flush(T - 1);
  x2_read(K2{}, 2u * A_SLOT, sp0, sp1, sw0, sw1);
  x2_mfma(K2{}, false, sp0, sp1, sw0, sw1);
#pragma unroll
  for (int n = 0; n < NT16 * MT16; ++n) acc16[n / MT16][n % MT16] += accI2[n / MT16][n % MT16];   // the last chain joins the sum

  // ---- epilogue
  __syncthreads();                                  // every wave has finished reading the ring
  rows_epilogue<CW, MT, NT, TABLE_OFF>(p, smem, acc16, wave, lane, wm, wn, m0, n0);
}

// The small-channel 3x3 layers (layer1 / layer2 conv2: 64 and 128 channels, 18 and 36 K-steps of a few hundred MFMA cycles
// each): their K loop runs at the LDS-DMA's latency, one round trip per K-step (profiles/
// r05_f16x2_kloop_ablations_in_network.log: DMA and barriers alone take 19-21 of their 21-25 us).  Here a pipeline step is a
// whole ROW-STEP (channel block, kh): ONE barrier, one input row (128 + 2 dil pixels, fetched once for its three taps) and the
// three taps' weight panels in flight together -- a third of the round trips.  128 pixels (a row, or a 128-pixel segment
// of a wider row) x 64 channels per block: four MFMA waves of 64 x 32 and four loader waves; three row-step slots of
// 18 KiB of pixels + 24 KiB of weights.  Same K order (channel block, kh, kw) and arithmetic as the kernel above.
// WN = 2: 64 channels, four MFMA waves, three weight stages (126 KiB); WN = 4: 128 channels, eight MFMA waves, two weight
// stages of 48 KiB (150 KiB): a weight stage is refilled a whole row-step (three K-steps) ahead either way.
template <int WN, int SB>
__global__ __launch_bounds__((2 * WN + 4) * 64, (2 * WN + 4) / 4) void conv3x3_rowstep_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WM = 2, MT = 2, NT = 1, CW = WM * WN, NW = 4;   // 2 x WN MFMA waves of 64 x 32, four loader waves
  constexpr int BN = WN * 32;
  constexpr int A_SLOT = kRowBytes, A_REGION = 3 * A_SLOT;
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

  const int NH = p.N * p.Ho;
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
  const int m0 = tile_m * 128, n0 = tile_n * BN;
  const int R = tile_m / segs, seg = tile_m - R * segs;
  const int img = R / p.Ho, oy = R - img * p.Ho;

  const rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const int pix_bytes = p.Ci * 4;
  const int cblocks = pix_bytes / 128;
  const int T = p.ksteps, RS = 3 * cblocks;
  const unsigned wrow_bytes = (unsigned)T * 128u;
  const unsigned tap_stride = (unsigned)cblocks * 128u;
  const int dil = p.dil;
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
    auto issue_rowstep = [&]() __attribute__((always_inline)) {
      const int iy = oy + (i_kh - 1) * dil;
      const bool rowok = (unsigned)iy < (unsigned)p.Hi;
      const unsigned rowoff = (unsigned)((img * p.Hi + iy) * p.Wi) * (unsigned)pix_bytes;
#pragma unroll
      for (int d = 0; d < LA_HI; ++d)
        if (d < la)
          dma16_buf(rowok ? a_col[d] + rowoff : kOutOfRange, xrsrc, smem_base + (unsigned)(i_kh * A_SLOT) + (unsigned)(lw + NW * d) * 1024u,
                    (unsigned)i_cb * 128u);
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
  const int wm = wave % WM, wn = wave / WM;
  auto row_off = [](int pix, int chunk) { return pix * 128 + (((chunk + 2 * ((pix >> 1) & 3)) & 7) << 4); };
  unsigned a_rd[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int pix = wm * 64 + kw * dil + r16;
    a_rd[kw][0] = (unsigned)row_off(pix, q16);
    a_rd[kw][1] = (unsigned)row_off(pix, 4 + q16);
  }
  const unsigned b_rd0 = (unsigned)(A_REGION + lds_off(wn * 32 + r16, q16));
  const unsigned b_rd1 = (unsigned)(A_REGION + lds_off(wn * 32 + r16, 4 + q16));
  f32x4 acc16[NT16][MT16], accI2[NT16][MT16];
#pragma unroll
  for (int j = 0; j < NT16; ++j)
#pragma unroll
    for (int i = 0; i < MT16; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc16[j][i][e] = 0.f; accI2[j][i][e] = 0.f; }
  const f16x8 kLow = {kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH};
  unsigned a_off = 0, b_off = 0;                       // byte offsets of the row-step's pixel slot (kh) and weight stage
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
    a_off = a_off == 2u * A_SLOT ? 0u : a_off + (unsigned)A_SLOT;
    b_off = b_off == (unsigned)(SB - 1) * B_STEP ? 0u : b_off + (unsigned)B_STEP;
  }
#pragma unroll
  for (int n = 0; n < NT16 * MT16; ++n) acc16[n / MT16][n % MT16] += accI2[n / MT16][n % MT16];
  __syncthreads();
  rows_epilogue<CW, MT, NT, TABLE_OFF>(p, smem, acc16, wave, lane, wm, wn, m0, n0);
}

template <int WM, int WN, int MT, int NT>
hipError_t launch_rows_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32, PR = BM / 128;
  constexpr int smem = rows_lds_bytes(WM, WN, MT, NT, 3) + 2048;
  static std::atomic<unsigned long long> attr_done{0};
  auto kern = &conv3x3_rows_kernel<WM, WN, MT, NT>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << dev, std::memory_order_release);
  }
  if (a.Co % BN != 0) return hipErrorInvalidValue;
  const int nh = a.N * a.Ho;
  const int tiles = ((nh + PR - 1) / PR) * (a.Co / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), smem, s, a);
  return hipGetLastError();
}

template <int WN, int SB>
hipError_t launch_rowstep_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int BN = WN * 32;
  constexpr int smem = 3 * kRowBytes + SB * 3 * BN * 128 + 2048;
  static std::atomic<unsigned long long> attr_done{0};
  auto kern = &conv3x3_rowstep_kernel<WN, SB>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << dev, std::memory_order_release);
  }
  if (a.Co % BN != 0 || a.Wo % 128 != 0) return hipErrorInvalidValue;
  const int tiles = a.N * a.Ho * (a.Wo / 128) * (a.Co / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3((2 * WN + 4) * 64), smem, s, a);
  return hipGetLastError();
}

}  // namespace

// Whether a convolution runs on the kernels of this file (f16x2): 3x3, stride 1, padding = dilation <= 8, no identity, and
//   kind 1: 128-pixel-wide maps, 256 output channels or more (layer3 / layer4 conv2, classifier.0 of a 1024-pixel-wide image):
//           tiles 18 / 19 (conv3x3_rows_kernel);
//   kind 2: 128-pixel-wide maps, 64 or 128 output channels (layer2.1-3 conv2): tile 20
//           (conv3x3_rowstep_kernel);
//   0: neither (the generic kernel).  A property of the layer and its shape: the K order follows from it (the head of this file).
int conv_rows_kind(int precision, int k, int stride, int pad, int dil, int Hi, int Wi, int Ho, int Wo, int Ci, int Co, bool has_res) {
  if (!(precision == 2 && k == 3 && stride == 1 && pad == dil && dil >= 1 && dil <= 8 && Ho == Hi && Wo == Wi && Ci % 32 == 0 && !has_res)) return 0;
  if (Wi == 128 && Co >= 256 && Co % 128 == 0) return 1;
  // (maps of 256 pixels -- layer1's conv2, two segments per row -- run on this kernel as well, and no faster than on the generic
  // tiles, which keep two blocks per CU there: profiles/r05_rowstep_kernel_layer1_layer2.log; left to them)
  if (Wi == 128 && Co % 64 == 0 && Co < 256) return 2;
  return 0;
}

// rows_tile (tile id - 18): 0 = one image row x 128 channels, one barrier per row-step (eight 64x32 MFMA waves + four loader
// waves), 1 = two image rows x 128 channels, one barrier per K-step (eight 64x64 waves), 2 = one image row x 64 channels, one
// barrier per row-step (four MFMA + four loader waves; kind 2)
hipError_t launch_conv3x3_rows(const ConvArgs& a, int rows_tile, hipStream_t s) {
  if (a.x_bytes == 0 || a.x_bytes >= kOutOfRange || a.w_bytes == 0 || a.w_bytes >= kOutOfRange) return hipErrorInvalidValue;
  if (a.stem || a.KH != 3 || a.KW != 3 || a.ksteps != 9 * (a.Ci * 4 / 128) ||
      a.M != a.N * a.Ho * a.Wo)
    return hipErrorInvalidValue;
  const int kind = conv_rows_kind(2, a.KH, a.stride, a.pad, a.dil, a.Hi, a.Wi, a.Ho, a.Wo, a.Ci, a.Co, a.res != nullptr);
  if (rows_tile == 2) return kind == 2 ? launch_rowstep_cfg<2, 3>(a, s) : hipErrorInvalidValue;
  if (kind != 1) return hipErrorInvalidValue;
  if (rows_tile == 0) return launch_rowstep_cfg<4, 2>(a, s);
  if (rows_tile == 1) return launch_rows_cfg<4, 2, 2, 2>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace nbc
