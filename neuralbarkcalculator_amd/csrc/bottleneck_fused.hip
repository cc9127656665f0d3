// Whole-bottleneck fusion for the f16x2 mode: conv1 (1x1) -> BN -> ReLU -> conv2 (3x3, stride 1, pad 1) -> BN -> ReLU ->
// conv3 (1x1) -> BN -> + identity -> ReLU of a ResNet bottleneck WITHOUT downsample (torchvision Bottleneck behind
// /root/reference/src/bark_calculator/models.py:128-134) in ONE launch: layer1.1, layer1.2 (64 mid channels) and
// layer2.1 .. layer2.3 (128).  These stages are short-K and move f32-sized maps: unfused, a layer1 bottleneck reads its
// 67-MB input twice (conv1's operand, conv3's identity), writes and re-reads the two mid-channel maps and starts three
// grids for 9 GFLOP; fused it reads x once (plus a one-pixel halo), keeps t1 and t2 in LDS and writes the output.
//
// A block owns a TH x 16 patch of output pixels.  Three GEMM phases on v_mfma_f32_16x16x32_f16 with the arithmetic of
// conv_igemm_dma.hip's f16x2 path, product for product and sum for sum (same K order: channel blocks; (kh, kw, channel
// block); same chain of eight K-steps joined to a running sum, cross terms apart, same final fma, BN fma, split), so the
// result is BIT-IDENTICAL to the three launches it replaces (tests/test_gpu_fusion.py compares whole forwards):
//   1. t1 = relu(bn1(conv1(x))) on the (TH+2) x 18 halo patch (rows padded to a multiple of 16): x rows and w1 rows come
//      through a two-slot LDS-DMA ring; halo pixels outside the image give t1 = 0 (conv2 pads t1, not x); t1 goes to an
//      LDS image [channel block][halo row][128 B] in the ring's row format (f16x2 pieces, XOR-swizzled chunks);
//   2. t2 = relu(bn2(conv2(t1))): the pixel fragments are read straight from the t1 image at (row + kh, col + kw), w2's
//      K-steps come through a three-slot ring; t2 replaces t1 in LDS;
//   3. out = relu(bn3(conv3(t2)) + x): w3 arrives in channel chunks (double buffer), each chunk's accumulators go through
//      a per-wave transpose scratch so that identity loads and stores are 128-byte row segments.
// `stop_after` 1 / 2 (tests only) ends the launch after a phase and writes t1 / t2 to `dbg` as an ordinary activation
// tensor, so that each phase is checked against the unfused kernels on its own.
#include "nbc_kernels.hpp"
#include "split16.hpp"

namespace nbc {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOutside = 0x80000000u;        // an offset beyond every resource: the hardware's range check returns zeros

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
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int CM>
struct Geo {
  static constexpr int TH = CM == 64 ? 8 : 4;           // output rows of a patch (16 columns)
  static constexpr int PX = TH * 16;                    // output pixels
  static constexpr int HRV = (TH + 2) * 18;             // halo pixels
  static constexpr int HR = (HRV + 15) / 16 * 16;       // halo rows of the LDS image
  static constexpr int P = CM / 32;                     // 128-byte channel blocks of t1 / t2
  static constexpr int C4 = 4 * CM;
  static constexpr int KS1 = C4 / 32, KS2 = 9 * P, KS3 = P;
  // waves: (rows x cols) of 16x16 tiles per wave and waves along the rows, per phase
  static constexpr int WR1 = CM == 64 ? 4 : 1, RB1 = (HR / 16) / WR1, CB1 = (CM / 16) / (8 / WR1);
  static constexpr int WR2 = (PX / 16) / 2, RB2 = 2, CB2 = (CM / 16) / (8 / WR2);
  static constexpr int NC = CM == 64 ? 128 : 64;        // output channels per chunk of phase 3
  static constexpr int NCH = C4 / NC;
  static constexpr int WR3 = (PX / 16) / 2, RB3 = 2, CB3 = (NC / 16) / (8 / WR3);
  static constexpr int STG1 = (HR + CM) * 128;          // ring slot of phase 1 (x rows, then w1 rows)
  static constexpr int STG2 = CM * 128;                 // ring slot of phase 2 (w2 rows)
  static constexpr int CHUNK3 = KS3 * NC * 128;         // one chunk of w3: [K-step][NC rows][128 B]
  // LDS regions.  T: the t1 / t2 image.  B: w3's two chunk buffers.  T and B together: the three slots of phase 1's
  // ring (the image is written only when that ring is done).  C: w2's ring (S2 slots; started at kernel entry) and,
  // once phase 2 is over, the per-wave transpose scratch of phase 3.
  static constexpr int R_T = 0;
  static constexpr int T_BYTES = HR * P * 128;
  static constexpr int R_B = T_BYTES;
  static constexpr int B_BYTES = 65536;
  static constexpr int R_C = R_B + B_BYTES;
  static constexpr int S2 = CM == 64 ? 3 : 2;           // slots of w2's ring
  static constexpr int PITCH3 = CB3 * 16 * 4 + 16;
  static constexpr int C_BYTES = (8 * 16 * PITCH3 > S2 * STG2) ? 8 * 16 * PITCH3 : S2 * STG2;
  static constexpr int LDS = R_C + C_BYTES;
  static_assert(3 * STG1 <= T_BYTES + B_BYTES && 2 * CHUNK3 <= B_BYTES && LDS <= 160 * 1024, "regions fit");
  static_assert(RB1 * WR1 * 16 == HR && CB1 * (8 / WR1) * 16 == CM, "phase 1 tiling");
  static_assert(RB2 * WR2 * 16 == PX && CB2 * (8 / WR2) * 16 == CM, "phase 2 tiling");
  static_assert(RB3 * WR3 * 16 == PX && CB3 * (8 / WR3) * 16 == NC, "phase 3 tiling");
};

// Three accumulator sets of a wave's R x C tiles, and the f16x2 K-step on them (conv_igemm_dma.hip, X2 path).
template <int R, int C>
struct Acc {
  f32x4 run[R][C], chain[R][C], cross[R][C];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { run[i][j][e] = 0.f; chain[i][j][e] = 0.f; cross[i][j][e] = 0.f; }
  }
  __device__ __forceinline__ void flush() {            // the chain of the last eight K-steps joins the running sum
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j) {
        run[i][j] += chain[i][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) chain[i][j][e] = 0.f;
      }
  }
  // product-major, like the conv kernel: all h0.h0', then h1'.h0, then h0'.h1
  __device__ __forceinline__ void step(const uint4 (&p0)[R], const uint4 (&p1)[R], const uint4 (&w0)[C], const uint4 (&w1)[C]) {
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
      for (int i = 0; i < R; ++i) chain[i][j] = mfma16(w0[j], p0[i], chain[i][j]);
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
      for (int i = 0; i < R; ++i) cross[i][j] = mfma16(w1[j], p0[i], cross[i][j]);
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
      for (int i = 0; i < R; ++i) cross[i][j] = mfma16(w0[j], p1[i], cross[i][j]);
  }
  __device__ __forceinline__ f32x4 total(int i, int j) const {       // (run + chain) + cross * 2^-11, as the conv kernel ends
    const f32x4 big = run[i][j] + chain[i][j];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(cross[i][j][e], kH1Unscale, big[e]);
    return r;
  }
};

// BN + ReLU of four channels, split, into the LDS image row of the lane's pixel: an 8-byte half of the h0 chunk and of
// the h1 chunk (channels ch .. ch+3 of the 32-channel block).
__device__ __forceinline__ void store_t_pieces(unsigned char* plane, int row, int ch_in_block, const f32x4& v, const float4& sc,
                                               const float4& sh, bool zero) {
  float o[4];
  o[0] = __builtin_fmaf(v[0], sc.x, sh.x); o[1] = __builtin_fmaf(v[1], sc.y, sh.y);
  o[2] = __builtin_fmaf(v[2], sc.z, sh.z); o[3] = __builtin_fmaf(v[3], sc.w, sh.w);
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
  f16x4 h0, h1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r = __builtin_elementwise_maximum(o[e], 0.f);
    r = zero ? 0.f : r;
    _Float16 a, b;
    split16(r, a, b);
    h0[e] = a; h1[e] = b;
  }
  const int chunk = ch_in_block >> 3, half = (ch_in_block >> 2) & 1;
  *reinterpret_cast<f16x4*>(plane + lds_off(row, chunk) + half * 8) = h0;
  *reinterpret_cast<f16x4*>(plane + lds_off(row, 4 + chunk) + half * 8) = h1;
}

template <int CM>
__global__ __launch_bounds__(512, 2) void bottleneck_x2_kernel(const BottleneckArgs p) {
  typedef Geo<CM> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q16 = lane >> 4;
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned smem_base = (unsigned)(size_t)(lds_u8*)smem;

  // ---- the patch
  const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + G::TH - 1) / G::TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int img = bid / tiles_y;
  const int y0 = ty * G::TH, x0 = tx * 16;
  const rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const unsigned pix_bytes = (unsigned)G::C4 * 4u;
  const unsigned img_base = (unsigned)img * (unsigned)p.H * (unsigned)p.W;

  // w2's ring (region C) is loaded by waves 0-3, w3's chunks (region B) by waves 4-7: each wave's vmcnt then counts
  // one stream of LDS-DMAs at a time.  The first w2 K-steps start here, under phase 1.
  const rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, p.w2_bytes, 0x00020000);
  constexpr int L2 = (CM / 8) / 4;                       // w2 pieces per K-step and loading wave: 2 or 4
  unsigned off2[L2];
#pragma unroll
  for (int i = 0; i < L2; ++i) {
    const int row = 8 * ((wave & 3) + 4 * i) + (lane >> 3);
    off2[i] = (unsigned)row * (unsigned)(G::KS2 * 128) + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
  auto issue2 = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < L2; ++i)
      dma16_buf(off2[i], w2rsrc, smem_base + (unsigned)(G::R_C + (t % G::S2) * G::STG2 + ((wave & 3) + 4 * i) * 1024), (unsigned)t * 128u);
  };
  if (wave < 4) {
#pragma unroll
    for (int t = 0; t < G::S2 - 1; ++t) issue2(t);
  }
  const rsrc_t w3rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w3), 0, p.w3_bytes, 0x00020000);
  constexpr int RPK = G::NC / 8;                         // pieces per K-step plane of a w3 chunk (32 pieces per chunk)
  unsigned off3[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int slot = (wave & 3) + 4 * i;
    const int ks = slot / RPK, row = 8 * (slot - ks * RPK) + (lane >> 3);
    off3[i] = (unsigned)row * (unsigned)(G::KS3 * 128) + (unsigned)ks * 128u + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
  auto issue3 = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      dma16_buf(off3[i], w3rsrc, smem_base + (unsigned)(G::R_B + (c & 1) * G::CHUNK3 + ((wave & 3) + 4 * i) * 1024),
                (unsigned)c * (unsigned)(G::NC * G::KS3 * 128));
  };

  // ================= phase 1: t1 = relu(bn1(conv1(x))) on the halo patch =================
  {
    const rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, p.w1_bytes, 0x00020000);
    constexpr int NS = (G::HR + CM) / 8;                 // 1-KiB pieces (8 rows) of a ring slot: x rows first, then w1 rows
    constexpr int NA = G::HR / 8;
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int slot = wave + 8 * i;
      const int row = 8 * slot + (lane >> 3);
      const unsigned lchunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
      if (slot < NA) {
        const int hy = row / 18, hx = row - hy * 18;
        const int y = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = row < G::HRV && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        off[i] = ok ? (img_base + (unsigned)y * (unsigned)p.W + (unsigned)xx) * pix_bytes + lchunk : kOutside;
      } else {
        off[i] = slot < NS ? (unsigned)(row - G::HR) * (unsigned)(G::KS1 * 128) + lchunk : kOutside;
      }
    }
    auto issue = [&](int t, int stage) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int slot = wave + 8 * i;                   // wave-uniform
        if (slot < NS) {
          const unsigned dst = smem_base + (unsigned)(stage * G::STG1 + slot * 1024);      // slots span regions T and B
          if (slot < NA) dma16_buf(off[i], xrsrc, dst, (unsigned)t * 128u);
          else dma16_buf(off[i], w1rsrc, dst, (unsigned)t * 128u);
        }
      }
    };
    const int wr = wave % G::WR1, wc = wave / G::WR1;
    Acc<G::RB1, G::CB1> acc;
    acc.clear();
    const bool four = wave + 24 < NS;                    // this wave issues four pieces per K-step (else three)
    issue(0, 0);
    issue(1, 1);
    for (int t = 0; t < G::KS1; ++t) {
      // own pieces of step t have landed when only step t+1's are outstanding (older LDS-DMAs -- w2's first K-steps --
      // retire first: vmcnt is in order)
      if (t + 1 < G::KS1) { if (four) wait_vmcnt<4>(); else wait_vmcnt<3>(); }
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                      // slot t % 3 is complete and visible; slot (t - 1) % 3 has been read by all
      if (t + 2 < G::KS1) issue(t + 2, (t + 2) % 3);
      if (t > 0 && (t & 7) == 0) acc.flush();
      const unsigned char* sa = smem + (t % 3) * G::STG1;
      const unsigned char* sb = sa + G::HR * 128;
      uint4 p0[G::RB1], p1[G::RB1], w0[G::CB1], w1[G::CB1];
#pragma unroll
      for (int i = 0; i < G::RB1; ++i) {
        const int row = (wr * G::RB1 + i) * 16 + r16;
        p0[i] = *reinterpret_cast<const uint4*>(sa + lds_off(row, q16));
        p1[i] = *reinterpret_cast<const uint4*>(sa + lds_off(row, 4 + q16));
      }
#pragma unroll
      for (int j = 0; j < G::CB1; ++j) {
        const int row = (wc * G::CB1 + j) * 16 + r16;
        w0[j] = *reinterpret_cast<const uint4*>(sb + lds_off(row, q16));
        w1[j] = *reinterpret_cast<const uint4*>(sb + lds_off(row, 4 + q16));
      }
      acc.step(p0, p1, w0, w1);
    }
    __syncthreads();                                     // the ring (regions T and B) has been read by all
    if (wave >= 4) { issue3(0); if (G::NCH > 1) issue3(1); }       // w3's first chunks arrive under phase 2
    // t1 -> LDS image
#pragma unroll
    for (int j = 0; j < G::CB1; ++j) {
      const int ch = (wc * G::CB1 + j) * 16 + 4 * q16;
      const float4 sc = *reinterpret_cast<const float4*>(p.s1 + ch), sh = *reinterpret_cast<const float4*>(p.b1 + ch);
#pragma unroll
      for (int i = 0; i < G::RB1; ++i) {
        const int row = (wr * G::RB1 + i) * 16 + r16;
        const int hy = row / 18, hx = row - hy * 18;
        const int y = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool outside = !((unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W);     // conv2 pads t1 with zeros
        store_t_pieces(smem + G::R_T + (ch >> 5) * (G::HR * 128), row, ch & 31, acc.total(i, j), sc, sh, outside);
      }
    }
  }
  __syncthreads();                                       // t1 complete; every wave has left the phase-1 ring
  if (p.stop_after == 1) {                               // tests: t1 of the patch's own pixels as an activation tensor
    for (int e = tid; e < G::PX * G::P * 8; e += 512) {
      const int chunk = e & 7, pl = (e >> 3) % G::P, px = (e >> 3) / G::P;
      const int py = px >> 4, pxx = px & 15, y = y0 + py, xx = x0 + pxx;
      if (y < p.H && xx < p.W) {
        const int row = (py + 1) * 18 + pxx + 1;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + G::R_T + pl * (G::HR * 128) + lds_off(row, chunk));
        *reinterpret_cast<uint4*>(static_cast<unsigned char*>(p.dbg) + ((size_t)(img_base + (unsigned)y * p.W + xx) * CM * 4 + pl * 128 + chunk * 16)) = v;
      }
    }
    return;
  }

  // ================= phase 2: t2 = relu(bn2(conv2(t1))) =================
  {
    const int wr = wave % G::WR2, wc = wave / G::WR2;
    Acc<G::RB2, G::CB2> acc;
    acc.clear();
    for (int t = 0; t < G::KS2; ++t) {
      if (wave < 4) {                                    // the loading waves: own pieces of step t have landed
        if (t + G::S2 - 2 < G::KS2) wait_vmcnt<(G::S2 - 2) * L2>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();                      // slot t % S2 visible; slot (t-1) % S2 has been read by all
      if (wave < 4 && t + G::S2 - 1 < G::KS2) issue2(t + G::S2 - 1);
      if (t > 0 && (t & 7) == 0) acc.flush();
      const int tap = t / G::P, cb = t - tap * G::P;
      const int kh = tap / 3, kw = tap - kh * 3;
      const unsigned char* sa = smem + G::R_T + cb * (G::HR * 128);
      const unsigned char* sb = smem + G::R_C + (t % G::S2) * G::STG2;
      uint4 p0[G::RB2], p1[G::RB2], w0[G::CB2], w1[G::CB2];
#pragma unroll
      for (int i = 0; i < G::RB2; ++i) {
        const int row = (wr * G::RB2 + i + kh) * 18 + r16 + kw;      // output row wr*RB2+i, column r16, shifted by the tap
        p0[i] = *reinterpret_cast<const uint4*>(sa + lds_off(row, q16));
        p1[i] = *reinterpret_cast<const uint4*>(sa + lds_off(row, 4 + q16));
      }
#pragma unroll
      for (int j = 0; j < G::CB2; ++j) {
        const int row = (wc * G::CB2 + j) * 16 + r16;
        w0[j] = *reinterpret_cast<const uint4*>(sb + lds_off(row, q16));
        w1[j] = *reinterpret_cast<const uint4*>(sb + lds_off(row, 4 + q16));
      }
      acc.step(p0, p1, w0, w1);
    }
    __syncthreads();                                     // every wave has finished reading t1 and the w2 ring
#pragma unroll
    for (int j = 0; j < G::CB2; ++j) {
      const int ch = (wc * G::CB2 + j) * 16 + 4 * q16;
      const float4 sc = *reinterpret_cast<const float4*>(p.s2 + ch), sh = *reinterpret_cast<const float4*>(p.b2 + ch);
#pragma unroll
      for (int i = 0; i < G::RB2; ++i) {
        const int row = (wr * G::RB2 + i) * 16 + r16;    // t2 image: row = pixel of the patch
        store_t_pieces(smem + G::R_T + (ch >> 5) * (G::PX * 128), row, ch & 31, acc.total(i, j), sc, sh, false);
      }
    }
  }
  __syncthreads();                                       // t2 complete
  if (p.stop_after == 2) {
    for (int e = tid; e < G::PX * G::P * 8; e += 512) {
      const int chunk = e & 7, pl = (e >> 3) % G::P, px = (e >> 3) / G::P;
      const int y = y0 + (px >> 4), xx = x0 + (px & 15);
      if (y < p.H && xx < p.W) {
        const uint4 v = *reinterpret_cast<const uint4*>(smem + G::R_T + pl * (G::PX * 128) + lds_off(px, chunk));
        *reinterpret_cast<uint4*>(static_cast<unsigned char*>(p.dbg) + ((size_t)(img_base + (unsigned)y * p.W + xx) * CM * 4 + pl * 128 + chunk * 16)) = v;
      }
    }
    return;
  }

  // ================= phase 3: out = relu(bn3(conv3(t2)) + x) =================
  {
    const int wr = wave % G::WR3, wc = wave / G::WR3;
    // pixel fragments of t2 do not change from chunk to chunk: read once
    uint4 p0[G::KS3][G::RB3], p1[G::KS3][G::RB3];
#pragma unroll
    for (int ks = 0; ks < G::KS3; ++ks)
#pragma unroll
      for (int i = 0; i < G::RB3; ++i) {
        const int row = (wr * G::RB3 + i) * 16 + r16;
        p0[ks][i] = *reinterpret_cast<const uint4*>(smem + G::R_T + ks * (G::PX * 128) + lds_off(row, q16));
        p1[ks][i] = *reinterpret_cast<const uint4*>(smem + G::R_T + ks * (G::PX * 128) + lds_off(row, 4 + q16));
      }
    // epilogue geometry: a lane owns eight channels of one pixel per pass
    constexpr int LPP = G::CB3 * 2;                      // lanes per pixel
    constexpr int PPP = 64 / LPP;                        // pixels per pass
    constexpr int PASSES = PPP >= 16 ? 1 : 16 / PPP;
    const int o_pix = lane / LPP, o_c8 = lane % LPP;
    unsigned char* scr = smem + G::R_C + wave * (16 * G::PITCH3);
    for (int c = 0; c < G::NCH; ++c) {
      if (wave >= 4) {                                   // the loading waves: own pieces of chunk c have landed
        if (c + 1 < G::NCH) wait_vmcnt<8>();             // (chunk c+1's may be in flight; stores of the last epilogue are
        else wait_vmcnt<0>();                            //  younger and make this wait longer than needed, never shorter)
      }
      __builtin_amdgcn_s_barrier();
      // the identity rows of this chunk are requested now and arrive under the chunk's MFMAs and the first transposes
      uint4 idp[G::RB3][PASSES][2];
#pragma unroll
      for (int i = 0; i < G::RB3; ++i)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int prow = ps * PPP + o_pix;
          const int y = y0 + wr * G::RB3 + i, xx = x0 + prow;
          const int yc = y < p.H ? y : p.H - 1, xc = (prow < 16 && xx < p.W) ? xx : x0;     // a valid address; unused when outside
          const int ch = c * G::NC + wc * (G::CB3 * 16) + o_c8 * 8;
          const unsigned char* xp = static_cast<const unsigned char*>(p.x) + (size_t)(img_base + (unsigned)yc * p.W + xc) * pix_bytes +
                                    (unsigned)(ch >> 5) * 128u + (unsigned)((ch & 31) >> 3) * 16u;
          idp[i][ps][0] = *reinterpret_cast<const uint4*>(xp);
          idp[i][ps][1] = *reinterpret_cast<const uint4*>(xp + 64);
        }
      Acc<G::RB3, G::CB3> acc;
      acc.clear();
      const unsigned char* cb_base = smem + G::R_B + (c & 1) * G::CHUNK3;
#pragma unroll
      for (int ks = 0; ks < G::KS3; ++ks) {
        uint4 w0[G::CB3], w1[G::CB3];
#pragma unroll
        for (int j = 0; j < G::CB3; ++j) {
          const int row = (wc * G::CB3 + j) * 16 + r16;
          w0[j] = *reinterpret_cast<const uint4*>(cb_base + ks * (G::NC * 128) + lds_off(row, q16));
          w1[j] = *reinterpret_cast<const uint4*>(cb_base + ks * (G::NC * 128) + lds_off(row, 4 + q16));
        }
        acc.step(p0[ks], p1[ks], w0, w1);
      }
      __builtin_amdgcn_s_barrier();                      // every wave has read chunk c: its buffer takes chunk c + 2
      if (wave >= 4 && c + 2 < G::NCH) issue3(c + 2);
      // epilogue of the chunk, one 16-pixel row block at a time through the wave's scratch
#pragma unroll
      for (int i = 0; i < G::RB3; ++i) {
#pragma unroll
        for (int j = 0; j < G::CB3; ++j) {
          const int chl = j * 16 + 4 * q16;              // channel inside the wave's span of the chunk
          const int ch = c * G::NC + wc * (G::CB3 * 16) + chl;
          const float4 sc = *reinterpret_cast<const float4*>(p.s3 + ch), sh = *reinterpret_cast<const float4*>(p.b3 + ch);
          const f32x4 a = acc.total(i, j);
          float4 v;
          v.x = __builtin_fmaf(a[0], sc.x, sh.x); v.y = __builtin_fmaf(a[1], sc.y, sh.y);
          v.z = __builtin_fmaf(a[2], sc.z, sh.z); v.w = __builtin_fmaf(a[3], sc.w, sh.w);
          *reinterpret_cast<float4*>(scr + r16 * G::PITCH3 + chl * 4) = v;
        }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int prow = ps * PPP + o_pix;             // pixel (column) inside the row block
          if (prow < 16) {
            const float4* sp = reinterpret_cast<const float4*>(scr + prow * G::PITCH3 + o_c8 * 32);
            const float4 a = sp[0], b = sp[1];
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const int y = y0 + wr * G::RB3 + i, xx = x0 + prow;
            if (y < p.H && xx < p.W) {
              const int ch = c * G::NC + wc * (G::CB3 * 16) + o_c8 * 8;
              const size_t at = (size_t)(img_base + (unsigned)y * p.W + xx) * pix_bytes + (unsigned)(ch >> 5) * 128u + (unsigned)((ch & 31) >> 3) * 16u;
              float idv[8];
              join16x8(idp[i][ps][0], idp[i][ps][1], idv);
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = __builtin_elementwise_maximum(v[q] + idv[q], 0.f);
              uint4 o0, o1;
              split16x8(v, o0, o1);
              unsigned char* op = static_cast<unsigned char*>(p.out) + at;
              *reinterpret_cast<uint4*>(op) = o0;
              *reinterpret_cast<uint4*>(op + 64) = o1;
            }
          }
        }
      }
    }
  }
}

template <int CM>
hipError_t launch_cm(const BottleneckArgs& a, hipStream_t s) {
  typedef Geo<CM> G;
  static unsigned long long attr_done = 0;               // bit d: attribute set on device d (racing setters are harmless)
  auto kern = &bottleneck_x2_kernel<CM>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  if (!((__atomic_load_n(&attr_done, __ATOMIC_ACQUIRE) >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return e;
    __atomic_fetch_or(&attr_done, 1ull << dev, __ATOMIC_RELEASE);
  }
  const long long tiles = (long long)a.N * ((a.H + G::TH - 1) / G::TH) * ((a.W + 15) / 16);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), G::LDS, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_bottleneck_x2(const BottleneckArgs& a, int cmid, hipStream_t s) {
  if (a.N < 1 || a.H < 1 || a.W < 1 || !a.x || a.x_bytes == 0 || a.x_bytes >= kOutside) return hipErrorInvalidValue;
  if (a.stop_after != 0 && a.stop_after != 1 && a.stop_after != 2) return hipErrorInvalidValue;
  if (a.stop_after == 0 ? !a.out : !a.dbg) return hipErrorInvalidValue;
  if (cmid == 64) return launch_cm<64>(a, s);
  if (cmid == 128) return launch_cm<128>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace nbc
