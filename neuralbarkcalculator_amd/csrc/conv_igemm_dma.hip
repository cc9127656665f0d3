// Implicit-GEMM convolution, LDS-DMA pipeline: the conv kernel of the path.
//
// GEMM view D[n][m] = sum_k W[n][k] X[m][k] with m = output pixel (image, oy, ox), n = output channel,
// k = (kh, kw, ci) walked in K-steps of 128 bytes per row; LDS image [rows][128 B] with XOR-swizzled
// 16-byte chunks; fused BN(+residual)(+ReLU) epilogue.  How tiles reach LDS:
//
//   * LDS-DMA (buffer_load_dwordx4 ... lds through a buffer resource over the activation / weight
//     buffer): each lane names a 16-byte SOURCE by a 32-bit offset (a pixel's channel chunk, a
//     weight-row chunk, or an offset outside the resource for padding-halo / tail lanes, which the
//     hardware's range check turns into zeros); the wave's 64 chunks land in 1 KiB of LDS
//     contiguously (8 rows x 128 B).  The swizzle therefore sits on the source side: physical slot
//     p of row r holds logical chunk p ^ ((r>>1)&7) (cdna_hip_programming.md rule 21).  No staging
//     VGPRs, no ds_write; the K-step's advance is the instruction's scalar offset.
//   * an S-stage LDS ring with COUNTED s_waitcnt vmcnt(N) and a raw s_barrier, one barrier per
//     K-step: at the top of step t a wave waits until only the (S-2) youngest K-steps' DMAs are
//     outstanding (its share of step t has landed), the barrier makes every wave's share visible
//     and retires the reads of step t-1, then the DMAs of step t+S-1 are issued into the slot
//     step t-1 used, then the MFMAs of step t run from LDS.
//     The DMA is issued from inline asm so that hipcc does not pair it with vmcnt(0) drains
//     (cdna_hip_programming.md section 5, "Pipelining across barriers" and 5.7).
//   * tile shape is a template parameter chosen per layer on the host (enough tiles to fill 256
//     CUs at batch 1, the largest tile otherwise: L2->LDS traffic per FLOP falls with tile size).
//
// D[n][m] orientation: weights are the MFMA A operand, pixels the B operand, so a lane
// ends up holding one pixel (column) and groups of four consecutive output channels (rows).
#include <cstdlib>

#include <atomic>

#include "nbc_kernels.hpp"
#include "split16.hpp"

namespace nbc {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// One LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to lds_base .. lds_base+1023.
// M0 carries the wave-uniform LDS base; it is compiler-reserved, so it is saved, written and
// restored inside the one statement that uses it.
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

// The same through a buffer resource: 64 lanes x 16 bytes from `rsrc` base + per-lane 32-bit offset + a
// wave-uniform scalar offset.  Three instructions per DMA instead of seven (no 64-bit pointer per row to
// advance, no M0 save/restore): every instruction a SIMD issues beside its MFMAs costs the matrix pipe
// about its own issue time (tools/mfma_f32_probe.hip).  A lane whose offset lies outside the resource
// (halo and tail lanes: kOutOfRange) gets zeros from the hardware's range check: no zero page.
// M0 is written and left: nothing else in this kernel reads it (the scale/shift DMAs above restore it).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOutOfRange = 0x80000000u;      // >= every resource size (activations and weights stay below 2 GiB)
__device__ __forceinline__ void dma16_buf(unsigned voff, rsrc_t rsrc, unsigned lds_base, unsigned soff) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %3 offen lds"
      :
      : "v"(voff), "s"(rsrc), "s"(lds_base), "s"(soff)
      : "memory");
}

// Diagnostic build (tools/conv_timeline.hip, -DNBC_STAMPS): thread 0 of every block writes the 100 MHz
// wall clock at phase boundaries into a buffer nothing else reads.  The library build has no stamps.
#ifdef NBC_STAMPS
#define NBC_STAMP(i)                                                                                   \
  do {                                                                                                 \
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define NBC_STAMP_CLK(i)                                                                               \
  do {                                                                                                 \
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define NBC_STAMP(i) do { } while (0)
#define NBC_STAMP_CLK(i) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS bytes of the ring (it also hosts the epilogue's per-wave transpose scratch, which is larger than
// the ring on the 8-wave 128x128 tile); the scale/shift table sits right behind it.
constexpr int ring_bytes(int prec, int wm, int wn, int mt, int nt, int s) {
  const int ring = s * (wm * mt * 32 + wn * nt * 32) * 128;
  const int scratch = wm * wn * 32 * (nt * 32 * 4 + 16);
  return ring > scratch ? ring : scratch;
}
// Register budget: blocks of >= 8 waves whose LDS lets two of them share a CU are compiled for four
// waves per SIMD (128 VGPRs), so that one block's epilogue overlaps the other's K loop.
constexpr int min_waves_per_simd(int prec, int wm, int wn, int mt, int nt, int s) {
  return (wm * wn >= 8 && (wm * wn >= 16 || ring_bytes(prec, wm, wn, mt, nt, s) + 2048 <= 80 * 1024)) ? 4 : 2;
}

// Tile = (WM*MT*32) pixels x (WN*NT*32) channels, WM*WN waves, S LDS stages.
// VAR 8 (f16x2): four more waves that do nothing but issue the LDS-DMAs of the ring ("loader waves"), while the WM*WN
// others only read fragments and issue MFMAs.  An LDS-DMA costs the wave that issues it 60-185 cycles
// (MI355X_MICROARCH.md, cycle constants); an f16x2 K-step is 384 MFMA cycles per wave, so four to six DMAs per wave
// and step in the MFMA waves' own instruction stream cost more than the arithmetic.
typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));   // what __builtin_nontemporal_load accepts
constexpr int loader_waves(int var) { return var == 8 ? 4 : 0; }
template <int PREC, int WM, int WN, int MT, int NT, int S, bool STEM, int VAR, bool BIGW>
__global__ __launch_bounds__((WM * WN + loader_waves(VAR)) * 64, VAR == 8 ? (WM * WN + loader_waves(VAR)) / 4 : min_waves_per_simd(PREC, WM, WN, MT, NT, S)) void conv_dma_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EB = PREC == 1 ? 2 : 4;           // f16x2: two f16 pieces per element, the f32 mode's geometry
  constexpr bool X2 = (PREC == 2);
  constexpr int CW = WM * WN;                       // waves that compute
  constexpr int LW = loader_waves(VAR);             // waves that only load (0: every wave does both)
  constexpr bool SPEC = LW > 0;
  constexpr int THREADS = (SPEC ? LW : CW) * 64;    // threads that share the loading of a K-step
  constexpr int BM = WM * MT * 32;
  constexpr int BN = WN * NT * 32;
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int TABLE_OFF = ring_bytes(PREC, WM, WN, MT, NT, S);      // scale/shift table behind ring and scratch
  constexpr int ROWS_PER_PASS = THREADS / 8;
  constexpr int A_PASSES = BM / ROWS_PER_PASS;
  constexpr int B_PASSES = BN / ROWS_PER_PASS;
  constexpr int L = A_PASSES + B_PASSES;          // DMA instructions per thread per K-step
  static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile rows must fill whole passes");
  static_assert(S >= 2 && S <= 4, "2..4 LDS stages");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = SPEC && wave >= CW;        // wave-uniform
  const bool loads = !SPEC || is_loader, computes = !SPEC || !is_loader;
  // f16x2 tiles of eight MFMA waves (two per SIMD: wave w and wave w + CW/2): the second half runs its MFMAs one barrier
  // behind the first (see the main loop).  Measured on every eight-wave tile (profiles/r04_f16x2_stagger.log): -3 ... -6 %
  // on tile 14 and -9 ... -17 % on tile 15 (loader waves), -2 ... -7 % on tile 5 (64x64 wave tiles), 0 ... -4 % on tiles
  // 6, 9, 10, 13, and +2 ... +18 % on tile 17, whose two blocks per CU already run out of phase and whose 128 registers do
  // not hold fragments that stay live across the barrier (19 spilled).  Kept where it pays: loader-wave tiles and
  // 64x64 wave tiles.
  constexpr bool STAGGER = (PREC == 2) && !STEM && CW >= 8 && (LW > 0 || MT * NT >= 4);
  const bool late_half = STAGGER && wave >= CW / 2 && wave < CW;
  NBC_STAMP(0);                                     // block start

  // ---- tile coordinates: blocks that share an XCD (blockIdx % 8) take a contiguous range of tiles, channel tiles
  // fastest, so the channel tiles of a pixel tile and the halo rows of neighbouring pixel tiles meet in one L2
  const int tiles_n = p.Co / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  // BIGW (f16x2 identity layers whose weights alone fill an XCD's 4 MiB L2 -- layer4's conv3: 2048 x 512 x 4 B; chosen by
  // launch_tile): a 4 x 2 grid of XCDs over (pixel tiles, channel tiles) halves the weights an XCD walks, and the identity
  // tile -- 64 KiB per block that nothing reads twice -- is loaded non-temporal, so that it does not push weight and input
  // panels out.  Measured (profiles/r04_f16x2_big_weight_identity_layers.log): layer4 conv3 -5 ... -10 % with both, -3 % with
  // either; every other identity layer LOSES 4-16 % with either (their panels fit beside the stream), hence the size test;
  // and as a run-time branch around the loads it keeps them from being scheduled among the last MFMAs (half the gain),
  // hence the template argument.  Which block computes which tile and how a load is hinted changes no result.
  if (BIGW && (tiles_n & 1) == 0 && (tiles_m & 3) == 0) {
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, rows = tiles_m >> 2, hc = tiles_n >> 1;
    tile_m = (xcd >> 1) * rows + li / hc;
    tile_n = (xcd & 1) * hc + li % hc;
  }
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- loader geometry: thread -> physical slot ps of the row, rows lr + ROWS_PER_PASS*i.
  // Wave w's 64 lanes cover rows 8w..8w+7 of a pass = 1 KiB of LDS, linear in the lane.
  const int ltid = SPEC ? ((tid - CW * 64) & (THREADS - 1)) : tid;    // index among the loading threads
  const int ps = ltid & 7;
  const int lr = ltid >> 3;
  const rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const int pix_bytes = p.Ci * EB;
  const unsigned wrow_bytes = (unsigned)p.ksteps * 128u;

  int a_iy0[A_PASSES], a_ix0[A_PASSES], a_img[A_PASSES], a_coff[A_PASSES];
  {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int row = lr + ROWS_PER_PASS * i;
      const int m = m0 + row;
      a_coff[i] = (ps ^ ((row >> 1) & 7)) * 16;     // logical chunk this slot holds
      if (m < p.M) {
        // integer division by a run-time value costs ~30 VALU instructions each; the common
        // shapes avoid both (one image: no image index; power-of-two width: a shift)
        const int img = p.N == 1 ? 0 : (p.hw_shift >= 0 ? (m >> p.hw_shift) : m / hw);
        const int rem = m - img * hw;
        const int oy = p.wo_shift >= 0 ? (rem >> p.wo_shift) : rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_img[i] = img * p.Hi * p.Wi;
      } else {
        a_iy0[i] = -(1 << 24);
        a_ix0[i] = 0;
        a_img[i] = 0;
      }
    }
  }
  unsigned w_off[B_PASSES];                  // byte offset of the row's chunk in K-step 0; K-step t adds the scalar t * 128
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int row = lr + ROWS_PER_PASS * i;
    w_off[i] = (unsigned)(n0 + row) * wrow_bytes + (unsigned)(ps ^ ((row >> 1) & 7)) * 16u;
  }

  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned smem_base = (unsigned)(size_t)(lds_u8*)smem;     // LDS byte offset of the ring
  const unsigned wave_off = (unsigned)(SPEC ? ((wave - CW) & (LW - 1)) : wave) * 1024u;
  const int cblocks = STEM ? 1 : pix_bytes / 128;                 // K-steps per tap
  int ld_kh = 0, ld_kw = 0, ld_cb = 0;
  int st_kh = 0, st_kw = 0;                                       // stem only: the lane's tap, see issue_one
  if constexpr (STEM) st_kw = a_coff[0] >> 4;       // the lane's slot in every kernel row

  // Byte offsets of the activation rows for the CURRENT tap (channel block 0).  Inside a tap a K-step only
  // moves 128 bytes along the channels: that is the DMA's scalar offset (ld_cb * 128), so a K-step costs no
  // vector instruction per row; the bounds test and the address arithmetic run once per tap (once per
  // kernel for a 1x1 convolution).  Halo and tail rows carry an offset outside the resource: zeros.
  unsigned a_off[A_PASSES];
  auto set_tap = [&](int kh, int kw) __attribute__((always_inline)) {
    const int dy = kh * p.dil, dx = kw * p.dil;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      const unsigned off = (unsigned)(a_img[i] + iy * p.Wi + ix) * (unsigned)pix_bytes + (unsigned)a_coff[i];
      a_off[i] = ok ? off : kOutOfRange;
    }
  };
  if constexpr (!STEM) set_tap(0, 0);

  // DMA d (0..L-1: activation passes first, then weight passes) of K-step t into ring slot `stage`.
  auto issue_one = [&](int d, int t, unsigned sa) __attribute__((always_inline)) {
    if (d < A_PASSES) {
      const int i = d;
#if defined(NBC_ABLATE) && (NBC_ABLATE == 5)
      if (X2 && !STEM && ld_kw != 0) return;           // tool builds only: pixel rows fetched for one tap column in three (timing of a
#endif                                                 // 3x3 K loop whose rows stay in LDS across the kernel's columns)
#if defined(NBC_ABLATE) && (NBC_ABLATE == 6)
      if (X2 && !STEM) return;                         // tool builds only: no pixel-row DMAs at all (weights only)
#endif
      if constexpr (!STEM) {
        dma16_buf(a_off[i], xrsrc, sa + (unsigned)(ROWS_PER_PASS * 128 * i), (unsigned)ld_cb * 128u);
      } else {
        // stem: one chunk = one tap's padded pixel, one K-step = one kernel row: this lane's tap of the
        // K-step being issued is (st_kh = the step, st_kw = the lane's chunk; slots beyond KW stay zero), so the
        // eight chunks of a row are consecutive input pixels (all passes of a lane share the chunk:
        // ROWS_PER_PASS is a multiple of 16)
        const int iy = a_iy0[i] + st_kh * p.dil, ix = a_ix0[i] + st_kw * p.dil;
        const bool ok = st_kw < p.KW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const unsigned off = (unsigned)(a_img[i] + iy * p.Wi + ix) * 16u;
        dma16_buf(ok ? off : kOutOfRange, xrsrc, sa + (unsigned)(ROWS_PER_PASS * 128 * i), 0u);
      }
    } else {
      const int i = d - A_PASSES;
      dma16_buf(w_off[i], wrsrc, sa + (unsigned)(A_BYTES + ROWS_PER_PASS * 128 * i), (unsigned)t * 128u);
    }
  };
  // The L DMAs of a K-step are issued in four parts so that the main loop can slot one part behind
  // each MFMA cluster (their SALU/VMEM issue then runs in the shadow of the matrix pipe).
  auto issue_part = [&](int part, int t, int stage) __attribute__((always_inline)) {
    if constexpr (VAR == 4 || VAR == 7) return;      // timing-only ablations: no refill DMAs in the loop
#if defined(NBC_ABLATE) && (NBC_ABLATE == 2)
    if constexpr (X2) return;                        // tool builds only (tools/build_variant.sh): f16x2 K loop without refill DMAs
#endif
    const unsigned sa = smem_base + (unsigned)stage * STAGE_BYTES + wave_off;
#pragma unroll
    for (int d = 0; d < L; ++d)
      if (d * 4 / L == part) issue_one(d, t, sa);
    if constexpr (STEM) {
      if (part == 3) ++st_kh;                      // next K-step: next kernel row
    }
    if constexpr (!STEM) {
      if (part == 3) {
        if (++ld_cb == cblocks) {                  // wave-uniform: next K-step starts a new tap
          ld_cb = 0;
          if (++ld_kw == p.KW) { ld_kw = 0; ++ld_kh; }
          if (ld_kh < p.KH) set_tap(ld_kh, ld_kw);
        }
      }
    }
  };
  auto issue_step = [&](int t, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int part = 0; part < 4; ++part) issue_part(part, t, stage);
  };

  // ---- MFMA geometry
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  // bf16 uses v_mfma_f32_16x16x32_bf16 (VAR 0): same cycles per FLOP as 32x32x16 but the chip holds a
  // higher clock on it under load (MI355X_MICROARCH.md, DVFS give-back item 7); VAR 1 keeps the
  // 32x32x16 form for A/B runs.  f32 always uses 32x32x2.
  constexpr bool M16 = (PREC == 1 && (VAR == 0 || VAR == 4)) || X2;
  constexpr int MT16 = 2 * MT, NT16 = 2 * NT;
  const int r16 = lane & 15, q16 = lane >> 4;
  // f32 (parity mode) sums in two levels: the 32 products of a K-step go through the MFMA's own fma chain
  // into accI, starting from zero, and accI is added to acc once per K-step (v_add_f32).  The longest
  // rounding chain is then T + 32 terms instead of 32*T (T up to 576): measured against a float64
  // evaluation the logit error falls about fourfold, to the level of the CPU reference's own blocked
  // summation.  Every tile shape does exactly this, so results stay independent of the tile.
  constexpr bool F32 = (PREC == 0);
  constexpr bool PREFETCH = F32 && S >= 3;         // see the f32 pipeline below
  f32x16 acc[M16 ? 1 : NT][M16 ? 1 : MT];
  f32x16 accI[F32 ? NT : 1][F32 ? MT : 1];
  f32x4 acc16[M16 ? NT16 : 1][M16 ? MT16 : 1];
  // f16x2 (NBC_PREC_F16X2, split16.hpp): a K-step is 32 channels, its LDS row [X0 x 32][X1 x 32] for a pixel (x = X0 +
  // X1 2^-11) and [P x 32][Q x 32] for a normalised weight row (w 2^k = P + Q, nbc_net.cpp).  Per 16x16 tile and K-step
  // three v_mfma_f32_16x16x32_f16, every product exact in f32, all into ONE chain accI2: P.X0, Q.X0 and (P 2^-11).X1 --
  // the last operand formed in registers (v_pk_mul_f16 by a power of two: exact for P >= 2^-3, which the row
  // normalisation gives every weight down to 2^-17 of its row's largest) -- so the three share a scale and the wave
  // keeps TWO accumulator sets instead of three: a 64x64 wave tile fits.  After FLUSH K-steps (256 channels x 3
  // products) the chain is added to the running sum acc16 and cleared: the f32 mode's two-level sum.  The dropped
  // Q.X1 is 2^-22 relative at worst.
  f32x4 accI2[X2 ? NT16 : 1][X2 ? MT16 : 1];
  if constexpr (X2) {
#pragma unroll
    for (int j = 0; j < NT16; ++j)
#pragma unroll
      for (int i = 0; i < MT16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) accI2[j][i][e] = 0.f;
  }
  if constexpr (M16) {
#pragma unroll
    for (int j = 0; j < NT16; ++j)
#pragma unroll
      for (int i = 0; i < MT16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[j][i][e] = 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
  }

#ifdef NBC_STAMPS
  unsigned long long st_vm = 0, st_bar = 0;         // diagnostic build: cycles each wave spends in the two waits
#endif
  // ---- f32 K-step (64 * MT*NT/4 MFMAs of 64 cycles each per wave; two waves share a SIMD's matrix pipe).
  // With three or more ring slots (PREFETCH) the ONE barrier of a K-step sits in its middle:
  //   * ks 0,1 run on fragments that were read before the barrier (the first ones at the end of the
  //     previous step), so no wave ever waits for an LDS round trip with the matrix pipe empty -- with the
  //     barrier at the top of the step and the first fragments read behind it the head conv kept the pipe
  //     busy for 0.91 of the K loop (2.37 GHz: 140 of 157 TF);
  //   * in front of the barrier every wave has waited for its DMAs of step t+1 (issued a whole step
  //     earlier), behind it slot t+1 is visible to all (its first fragments are read during ks 3) and slot
  //     t-1 is free: its refill (step t+S-1) is issued in parts between the MFMAs of ks 2 or ks 3;
  //   * a wave that reaches the barrier first leaves the pipe to its SIMD partner's MFMAs.
  // Two-slot tiles keep the barrier at the top of the step (`compute`'s caller) and read their first
  // fragments behind it.
  uint4 fpf[F32 ? 2 : 1][F32 ? MT : 1], fwf[F32 ? 2 : 1][F32 ? NT : 1];
  auto load_frags32 = [&](int stage, int ks, uint4 (&pfr)[F32 ? MT : 1], uint4 (&wfr)[F32 ? NT : 1]) __attribute__((always_inline)) {
    if constexpr (F32) {
      const unsigned char* sa = smem + stage * STAGE_BYTES;
      const unsigned char* sb = sa + A_BYTES;
      const int chunk = 2 * ks + h;
#pragma unroll
      for (int i = 0; i < MT; ++i)
        pfr[i] = *reinterpret_cast<const uint4*>(sa + lds_off((wm * MT + i) * 32 + r, chunk));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off((wn * NT + j) * 32 + r, chunk));
    }
  };
  // Two-level sum: accI collects FLUSH K-steps (32 * FLUSH products per output) through the MFMA's own fma
  // chain; a step whose index is a multiple of FLUSH first adds accI to acc and clears it (128 VALU
  // instructions per wave, behind a wave-uniform branch).  Every non-MFMA instruction costs the matrix pipe
  // about its own issue time (tools/mfma_f32_probe.hip: 64 adds in every K-step cost 6 %), hence not every
  // step; FLUSH = 8 also gives the smaller rounding error for the long chains (sqrt(256) + sqrt(T/8)
  // against sqrt(32) + sqrt(T)).
  constexpr int NTILES = NT * MT;
  constexpr int FLUSH = 8;
  // An LDS-DMA costs its wave 60-185 cycles of issue (MI355X_MICROARCH.md): the two waves of a SIMD (w and
  // w + half the block) issue their shares of the refill in different quarters of the step (ks 2 / ks 3).
  const bool late = WM * WN >= 8 && wave >= WM * WN / 2;
  if constexpr (F32) {
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) accI[n / MT][n % MT][e] = 0.f;
  }
  auto compute32 = [&](bool flush, int stage, bool more, bool all_issued, int next_stage, bool do_issue, int t_issue, int issue_stage)
      __attribute__((always_inline)) {
    if constexpr (F32) {
      if constexpr (!PREFETCH) load_frags32(stage, 0, fpf[0], fwf[0]);
      if (flush) {                                           // wave-uniform: every FLUSH-th step
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
          acc[n / MT][n % MT] += accI[n / MT][n % MT];
#pragma unroll
          for (int e = 0; e < 16; ++e) accI[n / MT][n % MT][e] = 0.f;
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (PREFETCH && ks == 2 && more) {                   // wave-uniform
#ifdef NBC_STAMPS
          const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
          if (S > 3 && all_issued) wait_vmcnt<(S - 3) * L>();      // step t+1 has landed (own share)
          else wait_vmcnt<0>();
#ifdef NBC_STAMPS
          const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
          __builtin_amdgcn_s_barrier();
#ifdef NBC_STAMPS
          st_vm += st1 - st0; st_bar += __builtin_amdgcn_s_memtime() - st1;
#endif
        }
        if (ks + 1 < 4) load_frags32(stage, ks + 1, fpf[(ks + 1) & 1], fwf[(ks + 1) & 1]);
        else if (PREFETCH && more) load_frags32(next_stage, 0, fpf[0], fwf[0]);
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
          const int j = n / MT, i = n % MT;
          const float4 wv = __builtin_bit_cast(float4, fwf[ks & 1][j]);
          const float4 pv = __builtin_bit_cast(float4, fpf[ks & 1][i]);
          accI[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, pv.x, accI[j][i], 0, 0, 0);
          accI[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, pv.y, accI[j][i], 0, 0, 0);
          accI[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, pv.z, accI[j][i], 0, 0, 0);
          accI[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, pv.w, accI[j][i], 0, 0, 0);
          if (PREFETCH && ks >= 2 && do_issue && (ks == 3) == late) {   // wave-uniform: refill parts behind this tile's MFMAs
#pragma unroll
            for (int part = 0; part < 4; ++part)
              if (part * NTILES / 4 == n) issue_part(part, t_issue, issue_stage);
          }
        }
      }
    }
  };
  // one K-step of the f32 kernel
  auto step32 = [&](int t, int stage, bool more, bool all_issued, int next_stage, bool do_issue, int t_issue, int issue_stage)
      __attribute__((always_inline)) {
    compute32(t > 0 && (t & (FLUSH - 1)) == 0, stage, more, all_issued, next_stage, do_issue, t_issue, issue_stage);
  };

  // f16x2 K-step in two parts, so that the two waves of a SIMD can run them out of phase (STAGGER below): the
  // fragment reads of a ring slot into registers that live across the loop's barrier, and the MFMAs on them.
  constexpr bool X2F = X2 && !STEM;
  constexpr int XM = X2F ? MT16 : 1, XN = X2F ? NT16 : 1;
  // staggered tiles: the late half's fragments live across the loop's barrier (elsewhere they are locals of a K-step)
  constexpr bool X2_STAG = STAGGER;                 // ONE predicate: which tiles stagger and whose fragments cross the barrier
  uint4 sp0[X2_STAG ? XM : 1], sp1[X2_STAG ? XM : 1], sw0[X2_STAG ? XN : 1], sw1[X2_STAG ? XN : 1];
  auto x2_flush = [&](int t) __attribute__((always_inline)) {
    if constexpr (X2) {
      if (t > 0 && (t & 7) == 0) {                     // wave-uniform: the chain of the last eight K-steps joins the sum
#pragma unroll
        for (int n = 0; n < NT16 * MT16; ++n) {
          acc16[n / MT16][n % MT16] += accI2[n / MT16][n % MT16];
#pragma unroll
          for (int e = 0; e < 4; ++e) accI2[n / MT16][n % MT16][e] = 0.f;
        }
      }
    }
  };
  auto x2_read = [&](int stage, uint4 (&xp0)[XM], uint4 (&xp1)[XM], uint4 (&xw0)[XN], uint4 (&xw1)[XN]) __attribute__((always_inline)) {
    if constexpr (X2F) {
#if defined(NBC_ABLATE) && (NBC_ABLATE == 3 || NBC_ABLATE == 4)
      // tool builds only: no fragment reads (the registers are marked written so that the MFMAs stay)
#pragma unroll
      for (int i = 0; i < XM; ++i) { asm volatile("" : "+v"(xp0[i].x), "+v"(xp0[i].y), "+v"(xp0[i].z), "+v"(xp0[i].w)); asm volatile("" : "+v"(xp1[i].x), "+v"(xp1[i].y), "+v"(xp1[i].z), "+v"(xp1[i].w)); }
#pragma unroll
      for (int j = 0; j < XN; ++j) { asm volatile("" : "+v"(xw0[j].x), "+v"(xw0[j].y), "+v"(xw0[j].z), "+v"(xw0[j].w)); asm volatile("" : "+v"(xw1[j].x), "+v"(xw1[j].y), "+v"(xw1[j].z), "+v"(xw1[j].w)); }
      return;
#endif
      // lane (r16, q16) reads, of row r16 of every 16-row block, chunk q16 (high pieces of channels 8*q16..) and chunk
      // 4 + q16 (their low pieces)
      const unsigned char* sa = smem + stage * STAGE_BYTES;
      const unsigned char* sb = sa + A_BYTES;
#pragma unroll
      for (int i = 0; i < MT16; ++i) {
        xp0[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * MT * 32 + i * 16 + r16, q16));
        xp1[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * MT * 32 + i * 16 + r16, 4 + q16));
      }
#pragma unroll
      for (int j = 0; j < NT16; ++j) {
        xw0[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * NT * 32 + j * 16 + r16, q16));
        xw1[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * NT * 32 + j * 16 + r16, 4 + q16));
      }
    }
  };
  auto x2_mfma = [&](int t, bool do_issue, int t_issue, int issue_stage, uint4 (&xp0)[XM], uint4 (&xp1)[XM], uint4 (&xw0)[XN], uint4 (&xw1)[XN])
      __attribute__((always_inline)) {
    if constexpr (X2F) {
      constexpr int NTI = NT16 * MT16;
#if defined(NBC_ABLATE) && (NBC_ABLATE == 1 || NBC_ABLATE == 4)
      // tool builds only: fragments consumed, no MFMA
#pragma unroll
      for (int i = 0; i < XM; ++i) { asm volatile("" :: "v"(xp0[i].x), "v"(xp0[i].y), "v"(xp0[i].z), "v"(xp0[i].w)); asm volatile("" :: "v"(xp1[i].x), "v"(xp1[i].y), "v"(xp1[i].z), "v"(xp1[i].w)); }
#pragma unroll
      for (int j = 0; j < XN; ++j) { asm volatile("" :: "v"(xw0[j].x), "v"(xw0[j].y), "v"(xw0[j].z), "v"(xw0[j].w)); asm volatile("" :: "v"(xw1[j].x), "v"(xw1[j].y), "v"(xw1[j].z), "v"(xw1[j].w)); }
      if (do_issue) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(part, t_issue, issue_stage);
      }
      return;
#endif
      const f16x8 kLow = {kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH};
      // product-major: two MFMAs on one accumulator are NTI instructions apart; the scaled high pieces of a weight
      // block are formed right in front of the block's third products (four v_pk_mul_f16; hoisting them cost 0-3 %,
      // profiles/r04_f16x2_kloop_schedule_variants_rejected.log)
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
        if (do_issue && (idx + 1) % (3 * NTI / 4) == 0) issue_part((idx + 1) / (3 * NTI / 4) - 1, t_issue, issue_stage);   // wave-uniform
      }
    }
  };

  // Fragments are double-buffered in registers: the ds_read_b128 of chunk pair ks+1 are issued
  // before the MFMAs of chunk pair ks, so LDS latency hides under the matrix pipe.
  auto compute = [&](int t, int stage, bool do_issue, int t_issue, int issue_stage) __attribute__((always_inline)) {
    const unsigned char* sa = smem + stage * STAGE_BYTES;
    const unsigned char* sb = sa + A_BYTES;
    if constexpr (X2) {
      constexpr int NTI = NT16 * MT16;
      if constexpr (!STEM) {
        uint4 p0[XM], p1[XM], w0[XN], w1[XN];
        x2_flush(t);
        x2_read(stage, p0, p1, w0, w1);
        x2_mfma(t, do_issue, t_issue, issue_stage, p0, p1, w0, w1);
      } else {
        x2_flush(t);
        const f16x8 kLow = {kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH};
        // stem: a chunk is one tap, pixel [X0 x 4][X1 x 4] (3 channels + a zero), weight [P x 4][Q x 4].  With the weight
        // chunk as (P, P 2^-11) the MFMA sums P.X0 + (P 2^-11).X1, as (Q, 0) it sums Q.X0: two MFMAs per tile and half
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint4 px[MT16], wa[NT16], wb[NT16];
#pragma unroll
          for (int i = 0; i < MT16; ++i) px[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * MT * 32 + i * 16 + r16, 4 * half + q16));
#pragma unroll
          for (int j = 0; j < NT16; ++j) {
            const uint4 wv = *reinterpret_cast<const uint4*>(sb + lds_off(wn * NT * 32 + j * 16 + r16, 4 * half + q16));
            const uint4 lo = __builtin_bit_cast(uint4, __builtin_bit_cast(f16x8, wv) * kLow);
            wa[j] = make_uint4(wv.x, wv.y, lo.x, lo.y);
            wb[j] = make_uint4(wv.z, wv.w, 0u, 0u);
          }
#pragma unroll
          for (int idx = 0; idx < 2 * NTI; ++idx) {
            const int prod = idx / NTI, n = idx % NTI, j = n / MT16, i = n % MT16;
            accI2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, prod == 0 ? wa[j] : wb[j]), __builtin_bit_cast(f16x8, px[i]), accI2[j][i], 0, 0, 0);
            if (do_issue && (idx + 1) % NTI == 0) issue_part(2 * half + (idx + 1) / NTI - 1, t_issue, issue_stage);
          }
        }
      }
      return;
    }
    if constexpr (VAR == 6) {                        // timing-only ablation: DMA + barriers only
      if (do_issue) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(part, t_issue, issue_stage);
      }
      return;
    }
    if constexpr (M16) {
      // two 32-deep halves per K-step; lane (r16, q16) reads row r16 of each 16-row tile, chunk 4*half+q16
      constexpr bool DBUF = (MT16 + NT16) * 8 <= 64;      // both halves' fragments in <= 64 VGPRs
      uint4 pf[DBUF ? 2 : 1][MT16], wf[DBUF ? 2 : 1][NT16];
      auto load_half = [&](int half, uint4 (&pfr)[MT16], uint4 (&wfr)[NT16]) {
        const int chunk = 4 * half + q16;
#pragma unroll
        for (int i = 0; i < MT16; ++i)
          pfr[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * MT * 32 + i * 16 + r16, chunk));
#pragma unroll
        for (int j = 0; j < NT16; ++j)
          wfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * NT * 32 + j * 16 + r16, chunk));
      };
      load_half(0, pf[0], wf[0]);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int cur = DBUF ? half : 0;
        if (DBUF && half == 0) load_half(1, pf[DBUF ? 1 : 0], wf[DBUF ? 1 : 0]);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
          for (int j = sub * NT16 / 2; j < (sub + 1) * NT16 / 2; ++j)
#pragma unroll
            for (int i = 0; i < MT16; ++i)
              acc16[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, wf[cur][j]), __builtin_bit_cast(bf16x8, pf[cur][i]), acc16[j][i], 0, 0, 0);
          if (do_issue) issue_part(2 * half + sub, t_issue, issue_stage);   // wave-uniform branch
        }
        if (!DBUF && half == 0) load_half(1, pf[0], wf[0]);
      }
      return;
    }
    uint4 pf[2][MT], wf[2][NT];
    auto load_frags = [&](int ks, uint4 (&pfr)[MT], uint4 (&wfr)[NT]) {
      const int chunk = 2 * ks + h;
#pragma unroll
      for (int i = 0; i < MT; ++i)
        pfr[i] = *reinterpret_cast<const uint4*>(sa + lds_off((wm * MT + i) * 32 + r, chunk));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off((wn * NT + j) * 32 + r, chunk));
    };
    if constexpr (VAR == 7) {                        // timing-only ablation: MFMAs on constant fragments
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int i = 0; i < MT; ++i) pf[b][i] = make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[b][j] = make_uint4(0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u);
      }
    } else {
      load_frags(0, pf[0], wf[0]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (VAR != 7 && ks + 1 < 4) load_frags(ks + 1, pf[(ks + 1) & 1], wf[(ks + 1) & 1]);
      if constexpr (VAR == 3) {                        // timing-only ablation: fragments read, no MFMA
#pragma unroll
        for (int i = 0; i < MT; ++i)
          asm volatile("" ::"v"(pf[ks & 1][i].x), "v"(pf[ks & 1][i].y), "v"(pf[ks & 1][i].z), "v"(pf[ks & 1][i].w));
#pragma unroll
        for (int j = 0; j < NT; ++j)
          asm volatile("" ::"v"(wf[ks & 1][j].x), "v"(wf[ks & 1][j].y), "v"(wf[ks & 1][j].z), "v"(wf[ks & 1][j].w));
      } else
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, wf[ks & 1][j]), __builtin_bit_cast(bf16x8, pf[ks & 1][i]),
              acc[j][i], 0, 0, 0);
        }
      if (do_issue) issue_part(ks, t_issue, issue_stage);   // wave-uniform branch
    }
  };

  // ---- epilogue geometry (needed before the last K-step: the identity tile is prefetched there)
  constexpr int SLAB_CH = NT * 32;                  // channels of the wave's slab
  constexpr int PITCH = SLAB_CH * 4 + 16;           // f32 scratch row, padded against bank conflicts
  constexpr int OUT_CH = X2 ? 8 : 16 / EB;          // channels per lane and pass: 16 output bytes (f16x2: an h0 chunk and an h1 chunk)
  constexpr int CPR = SLAB_CH / OUT_CH;             // 16-byte output chunks per pixel row
  constexpr int PIX_PER_PASS = 64 / CPR;
  constexpr int PASSES = 32 / PIX_PER_PASS;
  // identity prefetch: <= 64 VGPRs per lane, and not on the 128x64 wave tile of the 16x16 path
  // (128 accumulators + 48 fragment registers leave no room: it spilled)
  constexpr bool RES_PREFETCH = (MT * PASSES <= 16) && !(PREC == 1 && (VAR == 0 || VAR == 4) && MT * NT >= 8) && !X2;
  static_assert(WM * WN * 32 * PITCH <= TABLE_OFF, "epilogue scratch must fit below the scale/shift table");
  // Output addressing: a wave-uniform 64-bit base (first pixel of the tile, first channel of the wave's
  // slab) plus a 32-bit per-lane offset (row inside the tile x row pitch + the lane's 16-byte chunk).
  const int o_pix = lane / CPR, o_chunk = lane % CPR;
  const int n_slab = n0 + wn * SLAB_CH;
  const unsigned row_bytes = (unsigned)p.Co * EB;
  const size_t tile_off = ((size_t)m0 * p.Co + n_slab) * EB;
  unsigned char* ytile = static_cast<unsigned char*>(p.y) + tile_off;
  const unsigned char* rtile = p.res ? static_cast<const unsigned char*>(p.res) + tile_off : nullptr;
  const int rows_valid = p.M - m0;                  // rows of this tile inside the image batch (>= 1)
  const int row0 = wm * MT * 32 + o_pix;            // + i*32 + pass*PIX_PER_PASS
  // byte offset of the lane's chunk behind the slab's first channel (f16x2: the h0 chunk; its h1 chunk is 64 bytes on)
  const unsigned lane_chunk = X2 ? (unsigned)(o_chunk >> 2) * 128u + (unsigned)(o_chunk & 3) * 16u : (unsigned)o_chunk * 16u;
  uint4 rpre[RES_PREFETCH ? MT : 1][RES_PREFETCH ? PASSES : 1];
  // f16x2: the identity tile (an h0 and an h1 chunk per lane and pass) is requested right BEHIND the last K-step's
  // MFMAs -- its fragment registers are free by then -- and arrives under the accumulator sums, the block barrier and
  // the first slab's trip through the scratch
  constexpr bool RES_PREFETCH2 = X2 && MT * PASSES * 2 <= 16;
  uint4 rpre2[RES_PREFETCH2 ? MT : 1][RES_PREFETCH2 ? PASSES : 1][2];

  const int T = p.ksteps;
  auto prefetch_identity = [&]() {
    if constexpr (RES_PREFETCH) {
      if (rtile) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int ps2 = 0; ps2 < PASSES; ++ps2) {
            int row = row0 + i * 32 + ps2 * PIX_PER_PASS;
            row = row < rows_valid ? row : rows_valid - 1;   // tail rows read a valid row; never stored
            rpre[i][ps2] = *reinterpret_cast<const uint4*>(rtile + ((unsigned)row * row_bytes + lane_chunk));
          }
      }
    }
  };

  NBC_STAMP(8);                                     // address set-up done
  // ---- pipeline.  Steps beyond T issue nothing; the counted wait then over-waits, which is safe
  // (vmcnt retires in order), and the tail uses vmcnt(0).
  // First of all the block's BN scale/shift pairs go to a 2 KiB LDS table behind the ring (two
  // LDS-DMAs of wave 0, older than every ring DMA, so the first counted wait covers them): the
  // epilogue then reads them from LDS instead of paying an L2 round trip per 32-pixel slab.
  if (wave == (SPEC ? CW : 0) && lane < BN / 4) {
    dma16(p.scale + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF);
    dma16(p.shift + n0 + lane * 4, smem_base + (unsigned)TABLE_OFF + 1024u);
  }
  if (loads) {
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
      if (s < T) issue_step(s, s);
  }
  NBC_STAMP(1);                                     // prologue DMAs issued
  if constexpr (PREFETCH) {
    // f32 pipeline, S >= 3 slots (see compute32): steps 0 .. S-2 are in flight; step 0 must be visible
    // before its first fragments are read, everything later is handled by the mid-step barriers.
    if (S - 2 < T) wait_vmcnt<(S - 2) * L>();                  // step 0 has landed (own share)
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    load_frags32(0, 0, fpf[0], fwf[0]);
    NBC_STAMP(2); NBC_STAMP_CLK(11);
    for (int t = 0; t < T - 1; ++t)
      step32(t, t % S, true, t + S - 2 < T, (t + 1) % S, t + S - 1 < T, t + S - 1, (t + S - 1) % S);
    NBC_STAMP(3);
    NBC_STAMP_CLK(12);
#ifdef NBC_STAMPS
    if (p.stamps && lane == 0) { p.stamps[(size_t)blockIdx.x * 64 + 16 + wave] = st_vm; p.stamps[(size_t)blockIdx.x * 64 + 32 + wave] = st_bar; }
#endif
    prefetch_identity();                                         // no DMA is outstanding any more
    step32(T - 1, (T - 1) % S, false, false, 0, false, 0, 0);
  } else {
  // top of K-step t: own DMAs of step t have landed when at most (S-2) younger steps' DMAs are outstanding; the
  // barrier makes every wave's share visible and retires the reads of the slot about to be refilled
  auto loop_top = [&](int t) __attribute__((always_inline)) {
#ifdef NBC_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    if (t + (S - 2) < T) wait_vmcnt<(S - 2) * L>();
    else wait_vmcnt<0>();
#ifdef NBC_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef NBC_STAMPS
    if (t > 0) { st_vm += st1 - st0; st_bar += __builtin_amdgcn_s_memtime() - st1; }
#endif
    if (t == 0) { NBC_STAMP(2); NBC_STAMP_CLK(11); }   // first K-step landed
  };
  if constexpr (STAGGER) {
    // f16x2 blocks with two MFMA waves per SIMD.  In lock step both read their fragments behind the barrier -- eight
    // waves x 12 KiB through one LDS: 384 cycles in which no MFMA issues -- and then share the matrix pipe, the older
    // wave first: it waits at the next barrier for 0.3 of the loop, the pipe is busy for 0.57 of it (block timelines,
    // profiles/r04_f16x2_block_timelines.log).  The second half of the block's MFMA waves therefore runs ONE BARRIER
    // LATE: behind barrier t it issues the MFMAs of step t-1, whose fragments it read in front of the barrier, and
    // then reads the fragments of step t while the first half computes on them: the reads of one half hide under
    // the MFMAs of the other (MI355X_MICROARCH.md, two waves per SIMD, item 9).  Same products in the same order per
    // wave: bit-identical results.  The late half's reads of slot t must have RETURNED before it arrives at barrier
    // t+1 (the slot is refilled behind it): s_waitcnt lgkmcnt(0) in front of the barrier, which hipcc does not emit
    // for a raw s_barrier.  Three loops, one per role, each with T-1 barriers (a loop per role keeps the early half's
    // fragments out of the loop-carried state).
    if (is_loader) {
      for (int t = 0; t < T - 1; ++t) {
        loop_top(t);
        if (t + S - 1 < T) issue_step(t + S - 1, (t + S - 1) % S);
      }
    } else if (!late_half) {
      for (int t = 0; t < T - 1; ++t) {
        loop_top(t);
        if constexpr (!SPEC && S == 2) { if (t + S - 1 < T) issue_step(t + S - 1, (t + S - 1) % S); }
        x2_flush(t);
        x2_read(t % S, sp0, sp1, sw0, sw1);
        x2_mfma(t, !SPEC && S > 2 && t + S - 1 < T, t + S - 1, (t + S - 1) % S, sp0, sp1, sw0, sw1);
      }
    } else {
      // The older wave of a SIMD wins the arbitration for the matrix pipe; here that is the early half, which would push
      // its MFMAs of step t in front of the late half's MFMAs of step t-1 and leave the late half's reads exposed at the end
      // of the step.  With the late half at the higher priority its MFMAs run first (while the early half reads), then the
      // early half's (while the late half reads): -3.6 ... -4.6 % on the long-K layers on tile 14, nothing on the 32x32
      // wave tiles of tile 15 (profiles/r04_f16x2_stagger.log)
      if constexpr (MT * NT >= 2) __builtin_amdgcn_s_setprio(1);
      if (T > 1) {                                   // step 0: nothing to compute on yet
        loop_top(0);
        if constexpr (!SPEC) { if (S - 1 < T) issue_step(S - 1, (S - 1) % S); }
        x2_read(0, sp0, sp1, sw0, sw1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      for (int t = 1; t < T - 1; ++t) {
        loop_top(t);
        if constexpr (!SPEC && S == 2) { if (t + S - 1 < T) issue_step(t + S - 1, (t + S - 1) % S); }
        x2_flush(t - 1);
        x2_mfma(t - 1, !SPEC && S > 2 && t + S - 1 < T, t + S - 1, (t + S - 1) % S, sp0, sp1, sw0, sw1);
        x2_read(t % S, sp0, sp1, sw0, sw1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (MT * NT >= 2) __builtin_amdgcn_s_setprio(0);
    }
  } else {
  for (int t = 0; t < T - 1; ++t) {
    loop_top(t);
    // the ring slot of step t-1 is free from here on.  With three or more stages its refill (step
    // t+S-1) is issued in four parts, one behind each MFMA cluster of this step; with two stages the
    // refill is needed at the very next barrier, so it is issued at once to give it the whole step.
    if constexpr (SPEC) {                             // the loader waves refill the slot, the others compute
      if (is_loader) { if (t + S - 1 < T) issue_step(t + S - 1, (t + S - 1) % S); }
      else compute(t, t % S, false, 0, 0);
    } else if constexpr (S == 2) {
      if (t + S - 1 < T) issue_step(t + S - 1, (t + S - 1) % S);
      if constexpr (F32) step32(t, t % S, false, false, 0, false, 0, 0);
      else compute(t, t % S, false, 0, 0);
    } else {
      compute(t, t % S, t + S - 1 < T, t + S - 1, (t + S - 1) % S);
    }
  }
  }
  // last K-step, peeled: every DMA has retired, so the identity (residual) tile of the epilogue is
  // requested here and its HBM/MALL latency hides under the last MFMAs and the transposes below.
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  NBC_STAMP(3);                                     // last K-step landed (main loop done but for its MFMAs)
  NBC_STAMP_CLK(12);
#ifdef NBC_STAMPS
  if (p.stamps && lane == 0) { p.stamps[(size_t)blockIdx.x * 64 + 16 + wave] = st_vm; p.stamps[(size_t)blockIdx.x * 64 + 32 + wave] = st_bar; }
#endif
  if (computes) prefetch_identity();                 // (loader waves own no output rows: their row / channel indices lie outside the tile)
  if constexpr (F32) step32(T - 1, (T - 1) % S, false, false, 0, false, 0, 0);
  else if constexpr (STAGGER) {
    if (computes) {
      if (late_half && T > 1) { x2_flush(T - 2); x2_mfma(T - 2, false, 0, 0, sp0, sp1, sw0, sw1); }       // the late half catches up
      x2_flush(T - 1);
      x2_read((T - 1) % S, sp0, sp1, sw0, sw1);
      x2_mfma(T - 1, false, 0, 0, sp0, sp1, sw0, sw1);
    }
  } else if (computes) compute(T - 1, (T - 1) % S, false, 0, 0);
  }
  if constexpr (RES_PREFETCH2) {
    if (rtile && computes) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int ps2 = 0; ps2 < PASSES; ++ps2) {
          int row = row0 + i * 32 + ps2 * PIX_PER_PASS;
          row = row < rows_valid ? row : rows_valid - 1;     // tail rows read a valid row; never stored
          const unsigned char* rp = rtile + ((unsigned)row * row_bytes + lane_chunk);
          if constexpr (BIGW) {
            rpre2[i][ps2][0] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(rp)));
            rpre2[i][ps2][1] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(rp + 64)));
          } else {
            rpre2[i][ps2][0] = *reinterpret_cast<const uint4*>(rp);
            rpre2[i][ps2][1] = *reinterpret_cast<const uint4*>(rp + 64);
          }
        }
    }
  }

  if constexpr (F32) {                                // the last chain joins the running sum
#pragma unroll
    for (int n = 0; n < NTILES; ++n) acc[n / MT][n % MT] += accI[n / MT][n % MT];
  }
  if constexpr (X2) {                                 // likewise
#pragma unroll
    for (int n = 0; n < NT16 * MT16; ++n) acc16[n / MT16][n % MT16] += accI2[n / MT16][n % MT16];
  }

  // ---- epilogue.
  // The accumulators hold, per lane, one pixel and groups of four channels: stored as they stand,
  // a wave would touch 32 pixel rows with 16 bytes each per instruction (measured: the residual
  // 1x1 convs ran at ~2 TB/s).  Instead each wave transposes one 32-pixel x (NT*32)-channel slab at
  // a time through a private f32 scratch in the (now idle) LDS ring: BN scale/shift is applied on
  // the way in, and on the way out every lane owns 16 output bytes of one pixel, so identity loads
  // and stores are whole 128-byte (bf16) / 256-byte (f32) row segments.
  __syncthreads();                                  // every wave has finished reading the ring
  if (!computes) return;                            // loader waves: done (no barrier follows)
  NBC_STAMP(4);                                     // MFMAs done, epilogue starts
  unsigned char* scr = smem + wave * (32 * PITCH);
  const unsigned char* table = smem + TABLE_OFF + wn * SLAB_CH * 4;
  const bool relu = p.relu != 0;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    // (a) BN on the accumulators, into the scratch.  The scale/shift reads of a channel group are all
    // issued before its scratch writes (LDS operations of a wave complete in order, so a table read
    // behind a scratch write would wait for it).
    if constexpr (M16) {
      // 16x16 D layout: lane (r16, q16) holds pixel r16 and channels 4*q16..4*q16+3 of each tile
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
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float4 sc[4], sh[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * h;      // channel inside the slab
          sc[g] = *reinterpret_cast<const float4*>(table + nl * 4);
          sh[g] = *reinterpret_cast<const float4*>(table + 1024 + nl * 4);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * h;
          float4 v;
          v.x = __builtin_fmaf(acc[j][i][4 * g + 0], sc[g].x, sh[g].x);
          v.y = __builtin_fmaf(acc[j][i][4 * g + 1], sc[g].y, sh[g].y);
          v.z = __builtin_fmaf(acc[j][i][4 * g + 2], sc[g].z, sh[g].z);
          v.w = __builtin_fmaf(acc[j][i][4 * g + 3], sc[g].w, sh[g].w);
          *reinterpret_cast<float4*>(scr + r * PITCH + nl * 4) = v;
        }
      }
    }
    if (i == 0) NBC_STAMP(9);                       // first slab in scratch
    // (b) read the slab back row-wise: every lane gets 16 output bytes of one pixel per pass.  All
    // reads of the slab are issued before the first use (the scratch is wave-private).
    float v[PASSES][OUT_CH];
#pragma unroll
    for (int ps2 = 0; ps2 < PASSES; ++ps2) {
      const float4* sp = reinterpret_cast<const float4*>(scr + (ps2 * PIX_PER_PASS + o_pix) * PITCH + o_chunk * OUT_CH * 4);
#pragma unroll
      for (int q = 0; q < OUT_CH / 4; ++q) {
        const float4 t4 = sp[q];
        v[ps2][4 * q] = t4.x; v[ps2][4 * q + 1] = t4.y; v[ps2][4 * q + 2] = t4.z; v[ps2][4 * q + 3] = t4.w;
      }
    }
    // (c) + identity, ReLU (NaN-propagating: v_maximum3_f32), rounding, 16-byte stores in whole row segments
#pragma unroll
    for (int ps2 = 0; ps2 < PASSES; ++ps2) {
      const int row = row0 + i * 32 + ps2 * PIX_PER_PASS;
      const unsigned loff = (unsigned)row * row_bytes + lane_chunk;
      if constexpr (X2) {
        if (rtile) {
          float idv[8];
          if constexpr (RES_PREFETCH2) {
            join16x8(rpre2[i][ps2][0], rpre2[i][ps2][1], idv);
          } else {
            const unsigned char* rp = rtile + ((unsigned)(row < rows_valid ? row : rows_valid - 1) * row_bytes + lane_chunk);
            join16x8(*reinterpret_cast<const uint4*>(rp), *reinterpret_cast<const uint4*>(rp + 64), idv);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) v[ps2][q] += idv[q];
        }
      } else if (rtile) {
        uint4 rv;
        if constexpr (RES_PREFETCH) rv = rpre[i][ps2];
        else rv = *reinterpret_cast<const uint4*>(rtile + ((unsigned)(row < rows_valid ? row : rows_valid - 1) * row_bytes + lane_chunk));
        const unsigned u[4] = {rv.x, rv.y, rv.z, rv.w};
        if constexpr (PREC == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[ps2][q] += __builtin_bit_cast(float, u[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[ps2][2 * q] += __builtin_bit_cast(float, u[q] << 16);
            v[ps2][2 * q + 1] += __builtin_bit_cast(float, u[q] & 0xffff0000u);
          }
        }
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < OUT_CH; ++e) v[ps2][e] = __builtin_elementwise_maximum(v[ps2][e], 0.f);
      }
      uint4 o;
      if constexpr (X2) {
        uint4 o1;
        split16x8(v[ps2], o, o1);
        if (row < rows_valid) *reinterpret_cast<uint4*>(ytile + loff + 64) = o1;
      } else if constexpr (PREC == 0) {
        o.x = __builtin_bit_cast(unsigned, v[ps2][0]); o.y = __builtin_bit_cast(unsigned, v[ps2][1]);
        o.z = __builtin_bit_cast(unsigned, v[ps2][2]); o.w = __builtin_bit_cast(unsigned, v[ps2][3]);
      } else {
        unsigned pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 pr = {v[ps2][(2 * q) % OUT_CH], v[ps2][(2 * q + 1) % OUT_CH]};
          pk[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
        }
        o.x = pk[0]; o.y = pk[1]; o.z = pk[2]; o.w = pk[3];
      }
      if (row < rows_valid) *reinterpret_cast<uint4*>(ytile + loff) = o;
    }
    if (i == 0) NBC_STAMP(10);                      // first slab's stores issued
  }
#ifdef NBC_STAMPS
  NBC_STAMP(5);                                     // wave 0's stores issued
  wait_vmcnt<0>();
  NBC_STAMP(6);                                     // wave 0's stores acknowledged
  if (p.stamps && threadIdx.x == 0)                 // where the block ran: XCC_ID (reg 20) << 32 | HW_ID (reg 4)
    p.stamps[(size_t)blockIdx.x * 64 + 7] =
        ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
#endif
}

template <int PREC, int WM, int WN, int MT, int NT, int S, bool STEM, int VAR = 0, bool BIGW = false>
hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int smem = ring_bytes(PREC, WM, WN, MT, NT, S) + 2048;     // ring (or scratch) + scale/shift table
  static std::atomic<unsigned long long> attr_done{0};     // bit d: attribute set on device d (one context per device)
  auto kern = &conv_dma_kernel<PREC, WM, WN, MT, NT, S, STEM, VAR, BIGW>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1ull)) {   // setting it twice from two threads is harmless
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << dev, std::memory_order_release);
  }
  if (a.Co % BN != 0) return hipErrorInvalidValue;
  const int tiles = ((a.M + BM - 1) / BM) * (a.Co / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3((WM * WN + loader_waves(VAR)) * 64), smem, s, a);
  return hipGetLastError();
}

// The tile menu.  rows x cols = pixels x channels; "blocks/CU" is what the LDS ring (and the registers) allow.
//   id  tile      waves (m x n)  wave tile  stages  LDS      blocks/CU
//   0   128x64    2x2            64x32      3       72 KiB   2
//   1   128x128   2x2            64x64      2       64 KiB   2
//   2   256x128   4x2            64x64      3       144 KiB  1   (no f16x2 form: it spills)
//   3   256x256   2x4            128x64     2       128 KiB  1   (bf16 only)
//   4   128x128   2x2            64x64      4       128 KiB  1   (deeper prefetch; f32 / bf16)
//   5   128x256   2x4            64x64      3       144 KiB  1
//   6   256x64    4x2            64x32      3       120 KiB  1
//   7   128x64    2x2            64x32      2       48 KiB   3   (short-K layers: K fits two stages)
//   8   64x128    1x4            64x32      2       48 KiB   3
//   9   128x128   4x2            32x64      2       70 KiB   2   (8 waves: short-K layers, where the
//   10  128x64    4x2            32x32      2       48 KiB   3    serial prologue/epilogue code dominates
//   11  256x128   4x4            64x32      2       96 KiB   1    and more waves run it in parallel)
//   12  256x256   4x4            64x64      2       128 KiB  1   (bf16: short-K layers at batch >= 2; the matrix pipe is
//                                                               busier than with 8 waves, the clock lower: same TFLOP/s on
//                                                               long-K layers, 2-5 % faster epilogue-heavy 1x1 layers)
//   13  128x128   2x4            64x32      3       96 KiB   1   (8 waves)
//   f16x2 only (9, 10 there with three stages):
//   14  128x128   2x4 + 4        64x32      3       96 KiB   1   (13 with four loader waves, VAR 8: the tile of a layer whose
//                                                               128x128 tiles number 256 or fewer, one per CU: layer3 at batch 1)
//   15  128x64    4x2 + 4        32x32      3       72 KiB   1   (10 with four loader waves; layer2's 3x3 at batch 1)
//   16  128x128   2x2 + 4        64x64      3       96 KiB   1   (four MFMA waves of 64x64 + four loader waves: ties 14)
//   17  128x128   2x4            64x32      2       64 KiB   2   (13 with two stages at 128 registers: TWO blocks per CU, one's
//                                                               barrier waits, prologue and epilogue under the other's MFMAs;
//                                                               the tile of every layer with two or more 128x128 tiles per
//                                                               CU: 0.51 of the mode's peak on the head conv and layer4's
//                                                               3x3 against 0.45 for tile 14)
template <int PREC, bool STEM, int VAR>
hipError_t launch_tile(const ConvArgs& a, int tile, hipStream_t s) {
  if constexpr (PREC == 2) {             // f16x2 (two accumulator sets: 64x64 wave tiles at most)
    // identity layers with 3 MiB of weights or more (layer4's conv3) on the 128-channel tiles they run on: see BIGW in the kernel
    if constexpr (!STEM) {
      if (a.res != nullptr && a.w_bytes >= (3u << 20)) {
        if (tile == 17) return launch_cfg<PREC, 2, 4, 2, 1, 2, false, VAR, true>(a, s);
        if (tile == 1) return launch_cfg<PREC, 2, 2, 2, 2, 2, false, VAR, true>(a, s);
        if (tile == 14) return launch_cfg<PREC, 2, 4, 2, 1, 3, false, 8, true>(a, s);
      }
    }
    switch (tile) {
      case 0: return launch_cfg<PREC, 2, 2, 2, 1, 3, STEM, VAR>(a, s);
      case 1: return launch_cfg<PREC, 2, 2, 2, 2, 2, STEM, VAR>(a, s);
      case 5: return launch_cfg<PREC, 2, 4, 2, 2, 3, STEM, VAR>(a, s);
      case 16: return launch_cfg<PREC, 2, 2, 2, 2, 3, STEM, STEM ? VAR : 8>(a, s);    // 128x128 of 64x64 wave tiles + four loader waves
      case 17: return launch_cfg<PREC, 2, 4, 2, 1, 2, STEM, VAR>(a, s);               // 13 with two stages: two blocks per CU
      case 6: return launch_cfg<PREC, 4, 2, 2, 1, 3, STEM, VAR>(a, s);
      case 7: return launch_cfg<PREC, 2, 2, 2, 1, 2, STEM, VAR>(a, s);
      case 8: return launch_cfg<PREC, 1, 4, 2, 1, 2, STEM, VAR>(a, s);
      case 9: return launch_cfg<PREC, 4, 2, 1, 2, 3, STEM, VAR>(a, s);
      case 10: return launch_cfg<PREC, 4, 2, 1, 1, 3, STEM, VAR>(a, s);
      case 13: return launch_cfg<PREC, 2, 4, 2, 1, 3, STEM, VAR>(a, s);
#ifdef NBC_TILE14_S4
      case 14: return launch_cfg<PREC, 2, 4, 2, 1, 4, STEM, STEM ? VAR : 8>(a, s);     // tool builds only: four ring slots
#else
      case 14: return launch_cfg<PREC, 2, 4, 2, 1, 3, STEM, STEM ? VAR : 8>(a, s);     // 13 with four loader waves
#endif
      case 15: return launch_cfg<PREC, 4, 2, 1, 1, 3, STEM, STEM ? VAR : 8>(a, s);     // 10 with four loader waves
      default: return hipErrorInvalidValue;
    }
  } else
  switch (tile) {
    case 0: return launch_cfg<PREC, 2, 2, 2, 1, 3, STEM, VAR>(a, s);
    case 1: return launch_cfg<PREC, 2, 2, 2, 2, 2, STEM, VAR>(a, s);
    case 2: return launch_cfg<PREC, 4, 2, 2, 2, 3, STEM, VAR>(a, s);
    case 3:
      if constexpr (PREC == 0) return hipErrorInvalidValue;   // 2 x 128 accumulator registers: no f32 form
      else return launch_cfg<PREC, 2, 4, 4, 2, 2, STEM, VAR>(a, s);
    case 4: return launch_cfg<PREC, 2, 2, 2, 2, 4, STEM, VAR>(a, s);
    case 5: return launch_cfg<PREC, 2, 4, 2, 2, 3, STEM, VAR>(a, s);
    case 6: return launch_cfg<PREC, 4, 2, 2, 1, 3, STEM, VAR>(a, s);
    case 7: return launch_cfg<PREC, 2, 2, 2, 1, 2, STEM, VAR>(a, s);
    case 8: return launch_cfg<PREC, 1, 4, 2, 1, 2, STEM, VAR>(a, s);
    case 9: return launch_cfg<PREC, 4, 2, 1, 2, PREC == 0 ? 3 : 2, STEM, VAR>(a, s);    // f32: three slots (96 KiB, fragment prefetch)
    case 10: return launch_cfg<PREC, 4, 2, 1, 1, PREC == 0 ? 3 : 2, STEM, VAR>(a, s);   // f32: 72 KiB, two blocks per CU
    case 11: return launch_cfg<PREC, 4, 4, 2, 1, 2, STEM, VAR>(a, s);
    case 12:
      if constexpr (PREC == 0) return hipErrorInvalidValue;
      else return launch_cfg<PREC, 4, 4, 2, 2, 2, STEM, VAR>(a, s);   // bf16 only (f32: 128-register budget)
    case 13: return launch_cfg<PREC, 2, 4, 2, 1, 3, STEM, VAR>(a, s);
    default: return hipErrorInvalidValue;
  }
}

//   18  128x128   row-step kernel (conv3x3_rows.hip: ONE image row x 128 channels, eight 64x32 MFMA waves + four loader waves, the
//                                row in LDS for its three taps, one barrier per (channel block, kh)): the 3x3 layers of
//                                128-pixel-wide maps with 256 output channels or more run on it and on nothing else
//   19  128x64    row-step kernel (one image row x 64 channels, four MFMA + four loader waves): likewise those with 64 / 128
//                                output channels (layer2.1-3 conv2; conv_rows_kind)
//   20  256x64    row-step kernel (TWO image rows, a dilation apart, x 64 channels; eight MFMA + four loader waves): the layers of
//                                tile 18, same K order and bits, 27 % fewer bytes into LDS per product: their default
constexpr int kTileRows[CONV_TILE_COUNT] = {128, 128, 256, 256, 128, 128, 256, 128, 64, 128, 128, 256, 256, 128, 128, 128, 128, 128, 128, 128, 256};
constexpr int kTileCols[CONV_TILE_COUNT] = {64, 128, 128, 256, 128, 256, 64, 64, 128, 128, 64, 128, 256, 128, 128, 64, 128, 128, 128, 64, 64};

}  // namespace

int conv_tile_rows(int tile) { return tile >= 0 && tile < CONV_TILE_COUNT ? kTileRows[tile] : 0; }
int conv_tile_cols(int tile) { return tile >= 0 && tile < CONV_TILE_COUNT ? kTileCols[tile] : 0; }

// Whether tile id `tile` exists for this precision and divides the layer's output channels.
bool conv_tile_ok(int precision, int tile, int Co, int rows_kind) {
  if (tile < 0 || tile >= CONV_TILE_COUNT) return false;
  // the row-resident 3x3 kernel's tiles and the generic ones: never mixed (kind 1: 18; kind 2: 19; kind 0: 0 .. 17)
  const int tile_kind = tile < CONV_TILE_ROWS_FIRST ? 0 : tile == 19 ? 2 : 1;      // (kind 1: tiles 18 and 20)
  if (rows_kind != tile_kind) return false;
  if (rows_kind != 0) return precision == 2 && Co % kTileCols[tile] == 0;
  if (precision == 0 && (tile == 3 || tile == 12)) return false;   // the f32 kernel keeps two accumulator sets
  if (precision == 2 && (tile == 2 || tile == 3 || tile == 4 || tile == 11 || tile == 12)) return false;   // f16x2: no wave tiles of 128x64, no
                                                                   // 16-wave blocks; the 256x128 tile of 64x64 wave tiles spills
  if (precision != 2 && tile >= 14) return false;                  // the loader-wave tile is f16x2's (in bf16 a 256x128 tile
                                                                   // with loader waves ties the one without: section 6.4)
  return Co % kTileCols[tile] == 0;
}

// Default tile of a layer (what runs unless nbc_autotune has measured): the cheapest under a small cost model.
// A launch takes as long as the CU with the most blocks: b = ceil(blocks / 256) of them, run in groups of cap[t] -- the
// blocks of that tile a CU holds at once (LDS and registers): they share its matrix pipes, and their prologues and
// epilogues overlap -- that is g = b / cap full groups and a rest of r = b % cap blocks:
//   (g * cap / eff[t] + r / eff_r) * tile FLOPs / per-CU matrix rate  +  b * tile bytes * cb[t] / (50 GB/s)  +  ceil(b / cap) * ovh[t]
// (K = Cin*kh*kw products per output; tile bytes = the (rows + cols) x K operand panels + twice the output tile; eff_r
// lies between eff1[t], one block alone on its CU, and eff[t], cap blocks together).  What matters most is the block
// count: a 640x1024 image has 10 240 pixels at stride 8, so the head conv on 128x128 tiles is 320 blocks = two rounds of
// which the second is a quarter full, on 64x128 tiles 640 blocks = three per CU, a third faster; and whether a tile's
// blocks come in pairs: the f16x2 128x128 tile of eight waves at two blocks per CU runs the long-K layers at 0.51 of
// the mode's peak when every CU has two (or four) of them and at 0.33 when it has one, where the one-block-per-CU tiles
// (14 with loader waves, 5 with 64x64 wave tiles) reach 0.40-0.55.  Constants fitted to per-layer timings of every tile (scripts/tile_model_probe.py,
// scripts/fit_tile_model.py): f32 and bf16 on 28 (precision, batch, height) cases (profiles/r02_tile_model_fit.log:
// within 0.1-0.5 % (f32) / 0.4-4.4 % (bf16) of the per-layer best, which is where nbc_autotune lands too); f16x2 on
// eight cases (profiles/r04_tile_model_fit_f16x2.log: 0.2-1.6 % from the per-layer best, 0.9 % on average, and the same when
// every constant is perturbed by +-2 %: no choice sits on a knife edge).
namespace {
struct TileModel {
  double cu_flops_per_us;              // per-CU matrix rate the efficiencies refer to
  double eff[CONV_TILE_COUNT], eff1[CONV_TILE_COUNT], ovh_us[CONV_TILE_COUNT], cb[CONV_TILE_COUNT];
  int cap[CONV_TILE_COUNT];
};
constexpr TileModel kTileModel[3] = {
    // f32: 157.3 TF / 256 CUs
    {157.3e6 / 256.0,
     {0.85, 0.85, 0.85, 0.85, 0.85, 0.896, 0.722, 0.811, 0.894, 0.85, 0.85, 0.85, 0.85, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80},
     {0.85, 0.85, 0.85, 0.85, 0.85, 0.896, 0.722, 0.811, 0.894, 0.85, 0.85, 0.85, 0.85, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80},
     {4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 5.08, 3.14, 0.76, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0},
     {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
     {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}},
    // bf16: against the 1 400 TF/s the chip sustains on this kernel (power-limited), / 256 CUs
    {1400.0e6 / 256.0,
     {0.888, 0.85, 0.85, 0.897, 0.85, 0.911, 0.85, 0.85, 0.85, 0.754, 0.85, 0.85, 0.85, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80},
     {0.888, 0.85, 0.85, 0.897, 0.85, 0.911, 0.85, 0.85, 0.85, 0.754, 0.85, 0.85, 0.85, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80, 0.80},
     {1.19, 4.0, 4.0, 2.78, 4.0, 4.0, 4.0, 0.0, 4.0, 0.5, 4.0, 4.0, 3.61, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0},
     {0.91, 1.0, 1.0, 1.07, 1.0, 0.78, 1.0, 1.03, 1.0, 0.68, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0},
     {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}},
    // f16x2: f32-equivalent FLOPs against the 839 TF/s three f16 MFMAs per product allow (2 517 / 3), / 256 CUs
    {839.0e6 / 256.0,
     {0.421, 0.52, 0.5, 0.5, 0.5, 0.535, 0.42, 0.476, 0.42, 0.42, 0.455, 0.5, 0.5, 0.44, 0.47, 0.4, 0.448, 0.501, 0.50, 0.45, 0.515},
     {0.38, 0.36, 0.5, 0.5, 0.5, 0.535, 0.42, 0.383, 0.36, 0.42, 0.392, 0.5, 0.5, 0.44, 0.47, 0.4, 0.448, 0.36, 0.50, 0.45, 0.515},
     {3, 3, 3, 3, 3, 3.45, 3.007, 3, 2.746, 3, 2.868, 3, 3, 2.518, 3.874, 3, 3, 3.321, 4.0, 3.0, 4.0},
     {0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.309, 0.31, 0.272, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.195, 0.2, 0.3, 0.2},
     {2, 2, 1, 1, 1, 1, 1, 3, 3, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1}}};
}  // namespace

int choose_conv_tile(int M, int Co, int K, int precision, int rows_kind) {
  if (precision < 0 || precision > 2) return -1;
  const TileModel& tm = kTileModel[precision];
  const double eb = precision == 1 ? 2.0 : 4.0;
  int best = -1;
  double best_cost = 0.0;
  for (int t = 0; t < CONV_TILE_COUNT; ++t) {
    if (!conv_tile_ok(precision, t, Co, rows_kind)) continue;
    const double trows = kTileRows[t], cols = kTileCols[t];
    const long long blocks = (long long)((M + kTileRows[t] - 1) / kTileRows[t]) * (Co / kTileCols[t]);
    const long long b = (blocks + 255) / 256;
    const int cap = tm.cap[t];
    const long long g = b / cap, rest = b % cap;
    const double eff_r = cap > 1 && rest > 0 ? tm.eff1[t] + (tm.eff[t] - tm.eff1[t]) * (double)(rest - 1) / (double)(cap - 1) : tm.eff[t];
    const double flops = trows * cols * 2.0 * K;
    // operand panels per block: the row-resident kernel fetches a pixel row (144 pixels for 128) once for the three taps of a kernel
    // row, its two-row tile four rows per channel block for the six (row, kernel row) pairs
    const double prows = t == 20 ? trows / 4.0 * 1.125 : t >= CONV_TILE_ROWS_FIRST ? trows / 3.0 * 1.125 : trows;
    const double bytes = (prows + cols) * K * eb + trows * cols * eb * 2.0;
    const double cost = ((double)(g * cap) / tm.eff[t] + (double)rest / eff_r) * flops / tm.cu_flops_per_us +
                        (double)b * bytes * tm.cb[t] / 50.0e3 + (double)((b + cap - 1) / cap) * tm.ovh_us[t];
    // ties (to 1e-9 relative) go to the larger tile: fewer L2 -> LDS bytes per FLOP
    if (best < 0 || cost < best_cost * (1.0 - 1e-9) ||
        (cost <= best_cost * (1.0 + 1e-9) && trows * cols > (double)kTileRows[best] * kTileCols[best])) {
      best = t;
      best_cost = cost;
    }
  }
  return best;
}

hipError_t launch_conv_dma(const ConvArgs& a, int precision, int tile, hipStream_t s) {
  const int eb = precision == 1 ? 2 : 4;
  if (a.M <= 0 || a.Co % 64 != 0 || a.ksteps <= 0 || a.x_bytes == 0 || a.x_bytes >= kOutOfRange || a.w_bytes == 0 ||
      a.w_bytes >= kOutOfRange)
    return hipErrorInvalidValue;
  if (a.stem) {
    if (a.Ci * eb != 16 || a.ksteps != a.KH || a.KW > 8) return hipErrorInvalidValue;
  } else {
    if ((a.Ci * eb) % 128 != 0 || a.ksteps != a.KH * a.KW * (a.Ci * eb / 128)) return hipErrorInvalidValue;
  }
  const int rows = (!a.stem && a.KW == a.KH && a.M == a.N * a.Ho * a.Wo)
                       ? conv_rows_kind(precision, a.KH, a.stride, a.pad, a.dil, a.Hi, a.Wi, a.Ho, a.Wo, a.Ci, a.Co, a.res != nullptr) : 0;
  if (tile < 0) tile = choose_conv_tile(a.M, a.Co, a.ksteps * (128 / eb), precision, rows);
  if (!conv_tile_ok(precision, tile, a.Co, rows)) return hipErrorInvalidValue;
  if (rows != 0) return launch_conv3x3_rows(a, tile - CONV_TILE_ROWS_FIRST, s);
#ifdef NBC_DIAG
  // measurement builds only (tools/build_tools.sh): the library never reads these variables
  // NBC_CONV_ABLATE=1 (no MFMA) / 2 (no refill DMA): timing-only builds of the bf16 256x256 and
  // 128x256 tiles, results are garbage.  Never set outside an experiment.
  static const int ablate = [] { const char* e = getenv("NBC_CONV_ABLATE"); return e ? atoi(e) : 0; }();
  if (ablate && precision == 1 && !a.stem && (tile == 3 || tile == 5)) {
    // 1: no MFMA (DMA + fragment reads)  2: no refill DMA (MFMA + fragment reads)
    // 3: DMA + barriers only             4: MFMA on constant fragments, no DMA, no fragment reads
    switch (ablate) {
      case 1: return tile == 3 ? launch_cfg<1, 2, 4, 4, 2, 2, false, 3>(a, s) : launch_cfg<1, 2, 4, 2, 2, 3, false, 3>(a, s);
      case 2: return tile == 3 ? launch_cfg<1, 2, 4, 4, 2, 2, false, 4>(a, s) : launch_cfg<1, 2, 4, 2, 2, 3, false, 4>(a, s);
      case 3: return tile == 3 ? launch_cfg<1, 2, 4, 4, 2, 2, false, 6>(a, s) : launch_cfg<1, 2, 4, 2, 2, 3, false, 6>(a, s);
      default: return tile == 3 ? launch_cfg<1, 2, 4, 4, 2, 2, false, 7>(a, s) : launch_cfg<1, 2, 4, 2, 2, 3, false, 7>(a, s);
    }
  }
#endif
  if (precision == 0) return a.stem ? launch_tile<0, true, 0>(a, tile, s) : launch_tile<0, false, 0>(a, tile, s);
  if (precision == 2) return a.stem ? launch_tile<2, true, 0>(a, tile, s) : launch_tile<2, false, 0>(a, tile, s);
  if (a.stem) return launch_tile<1, true, 0>(a, tile, s);
  // bf16: the MFMA-heavy layers run on v_mfma_f32_16x16x32_bf16 (VAR 0: +4-5 % measured on the
  // head and layer4 3x3 convs, the chip holds a higher clock on it); the residual 1x1 layers keep
  // v_mfma_f32_32x32x16_bf16 (VAR 1), whose fragment registers leave room for the identity prefetch
  // on the 256x256 tile.  The choice depends on the layer only, never on the tile, so the
  // tile-invariance of the results holds.
#ifdef NBC_DIAG
  static const int mfma32 = [] { const char* e = getenv("NBC_CONV_MFMA32"); return e ? atoi(e) : 0; }();   // A/B runs
  if (mfma32) return launch_tile<1, false, 1>(a, tile, s);
#endif
  if (a.res != nullptr) return launch_tile<1, false, 1>(a, tile, s);
  return launch_tile<1, false, 0>(a, tile, s);
}

}  // namespace nbc
