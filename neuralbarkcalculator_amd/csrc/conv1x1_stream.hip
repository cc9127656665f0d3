// 1x1 convolution (+ identity + ReLU) in f16x2 as ONE persistent workgroup per CU that streams its tiles: while the MFMA
// waves run the K loop of tile i+1, dedicated epilogue waves drain tile i (BN, identity add, ReLU, split, stores) and the
// loader waves never stop filling the ring across the tile boundary.
//
// Why.  The short-K layers with large outputs (conv3 + identity of layer3 / layer4: 8 and 16 K-steps per 128 x 128 tile,
// 0.68 ms of the forward at 0.30 of the mode's peak) spend a third of a block's life in the epilogue with the matrix
// pipe idle and another tenth waiting for the first K-step to land (profiles/r04_f16x2_block_timelines_final.log: last
// K-step landed at 19.0 us, stores acknowledged at 27.9 us); two co-resident blocks of the generic kernel run those
// phases together.  Here the phases of consecutive tiles overlap by construction:
//   * waves 0-7   (M): fragment reads + MFMAs, the generic f16x2 K-step instruction for instruction (P.X0, Q.X0, (P 2^-11).X1
//                      into one chain that joins the running f32 sum every eighth K-step); after a tile's last K-step the RAW
//                      accumulators go to a 64-KiB staging tile in LDS and the next tile's K loop starts at once;
//   * waves 8-11  (L): the LDS-DMAs of the three-stage ring, two K-steps ahead, straight across tile boundaries;
//   * waves 12-15 (E): the staging tile of the PREVIOUS tile -> fma(acc, scale, shift) + identity -> ReLU -> split -> whole
//                      512-byte pixel rows stored, cut into pieces that ride the K-step barriers of the tile in flight.
// gfx950 has one barrier per workgroup, so every wave takes part in every K-step's barrier; the epilogue waves' work is cut
// accordingly (as the loader waves of the generic tiles 14-16 live).  16 waves at 128 registers, 96 KiB ring + 64 KiB staging =
// the CU's whole LDS.  Same K order and the same epilogue arithmetic as the generic kernel: bit-identical results (it is one
// more tile of the menu, id 20).
#include <atomic>

#include "nbc_kernels.hpp"
#include "split16.hpp"

namespace nbc {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOutOfRange = 0x80000000u;

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

// An identity load the compiler does not track (it would drain vmcnt to zero in front of every use inside the pass loop,
// i.e. wait for the stores of the pass before): issued here, waited for with the COUNTED s_waitcnt of e_wait_for below.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// (written IN PLACE into the caller's variable: nothing between here and the wait may copy a register whose data is still
// on its way)
__device__ __forceinline__ void buffer_load16_untracked(u32x4& dst, rsrc_t rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
// Vector-memory operations an epilogue wave issues BEHIND the identity loads of pass q before it uses them (loads, then
// stores, retire in issue order): with `ahead` passes in flight, pass q's two loads go out with the tile's first `ahead`
// passes (q < ahead) or inside pass q - ahead, in front of that pass's two stores; a pass issues two loads (for pass + ahead,
// while there is one) and two stores.
constexpr int e_younger_ops(int q, int ahead, int passes) {
  int n = 0;
  if (q < ahead) {
    n += 2 * (ahead - 1 - q);
    for (int r = 0; r < q; ++r) n += 2 + (r + ahead < passes ? 2 : 0);
  } else {
    n += 2;
    for (int r = q - ahead + 1; r < q; ++r) n += 2 + (r + ahead < passes ? 2 : 0);
  }
  return n;
}

// ... the same for every pass q = u, u + ahead, u + 2 ahead, ... behind the first: the smallest count (waiting for fewer
// outstanding operations than necessary is safe, for more is not)
constexpr int e_younger_ops_later(int u, int ahead, int passes) {
  int n = 1 << 20;
  for (int q = u + ahead; q < passes; q += ahead) n = e_younger_ops(q, ahead, passes) < n ? e_younger_ops(q, ahead, passes) : n;
  return n;
}

constexpr int kS = 3;                                // ring stages
constexpr int kStage = 2 * 128 * 128;                // 128 pixel rows + 128 weight rows of 128 bytes
constexpr int kRing = kS * kStage;                   // 96 KiB
constexpr int kStaging = 128 * 128 * 4;              // 64 KiB: the raw f32 accumulators of one tile, [pixel][channel], 16-byte chunks
                                                     // XOR-ed with the pixel index (both sides conflict-free)
constexpr int kLds = kRing + kStaging;               // 160 KiB: all of it
constexpr int kMWaves = 8, kLWaves = 4;            // + 4 epilogue waves
constexpr int kPasses = 8;                           // epilogue passes per E wave and tile: 4 pixels x 16 lanes of 8 channels each
constexpr int kAhead = 2;                            // passes whose identity chunks an E wave keeps in flight

// The tiles of one block: blocks that share an XCD (blockIdx % 8) take a contiguous range of tiles, channel tiles fastest
// (the generic kernel's order); inside the range the XCD's blocks take tiles round-robin, so at any time they work on
// neighbouring tiles that share pixel rows and weight panels in the XCD's L2.
struct TileWalk {
  int first, step, count;                            // tile indices first, first + step, ... (count of them)
  int tiles_n;
  __device__ __forceinline__ void tile(int k, int& m0, int& n0) const {
    const int idx = first + k * step;
    m0 = (idx / tiles_n) * 128;
    n0 = (idx % tiles_n) * 128;
  }
};

__global__ __launch_bounds__(1024, 4) void conv1x1_stream_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.ksteps;                            // K-steps per tile (1x1: one per 32-channel block)

  TileWalk tw;
  {
    tw.tiles_n = p.Co / 128;
    const int tiles_m = (p.M + 127) / 128;
    const int nblk = tiles_m * tw.tiles_n;
    const int q = nblk >> 3, rr = nblk & 7, xcd = blockIdx.x & 7;
    const int start = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    const int cnt = xcd < rr ? q + 1 : q;
    const int li = blockIdx.x >> 3, per = gridDim.x >> 3;
    tw.first = start + li;
    tw.step = per;
    tw.count = li < cnt ? (cnt - li + per - 1) / per : 0;
  }
  const int n_tiles = tw.count;
  const int steps_total = n_tiles * T;               // K-steps of this block; one more tile's worth of barriers drains the last tile
  if (n_tiles == 0) return;                          // (whole workgroup: no barrier is pending)

  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned smem_base = (unsigned)(size_t)(lds_u8*)smem;

  if (wave >= kMWaves && wave < kMWaves + kLWaves) {
    // ================================================================ loader waves
    const int ltid = tid - kMWaves * 64;              // 0 .. 255
    const int ps = ltid & 7, lr = ltid >> 3;          // physical chunk slot, row of a 32-row pass
    const unsigned wave_off = (unsigned)(wave - kMWaves) * 1024u;
    const rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const int pix_bytes = p.Ci * 4;
    const unsigned wrow_bytes = (unsigned)T * 128u;
    unsigned a_off[4], w_off[4];
    auto set_tile = [&](int k) __attribute__((always_inline)) {
      int m0, n0;
      tw.tile(k, m0, n0);
      const int hw = p.Ho * p.Wo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = lr + 32 * i;
        const unsigned chunk = (unsigned)(ps ^ ((row >> 1) & 7)) * 16u;
        const int m = m0 + row;
        unsigned off = kOutOfRange;
        if (m < p.M) {
          const int img = p.N == 1 ? 0 : (p.hw_shift >= 0 ? (m >> p.hw_shift) : m / hw);
          const int rem = m - img * hw;
          const int oy = p.wo_shift >= 0 ? (rem >> p.wo_shift) : rem / p.Wo;
          const int ox = rem - oy * p.Wo;
          off = (unsigned)((img * p.Hi + oy * p.stride) * p.Wi + ox * p.stride) * (unsigned)pix_bytes + chunk;
        }
        a_off[i] = off;
        w_off[i] = (unsigned)(n0 + row) * wrow_bytes + chunk;
      }
    };
    int l_tile = 0, l_t = 0, l_g = 0;                  // the K-step issued next: tile, step inside it, global index
    auto issue_step = [&]() __attribute__((always_inline)) {
      const unsigned sa = smem_base + (unsigned)(l_g % kS) * kStage + wave_off;
      const unsigned soff = (unsigned)l_t * 128u;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma16_buf(a_off[i], xrsrc, sa + (unsigned)(4096 * i), soff);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma16_buf(w_off[i], wrsrc, sa + (unsigned)(128 * 128 + 4096 * i), soff);
      ++l_g;
      if (++l_t == T) { l_t = 0; if (++l_tile < n_tiles) set_tile(l_tile); }
    };
    set_tile(0);
#pragma unroll
    for (int s = 0; s < kS - 1; ++s)
      if (l_g < steps_total) issue_step();
    for (int g = 0; g < steps_total + T; ++g) {
      // this wave's DMAs of step g have landed when only the younger step's (8 of them) are outstanding
      if (g + 1 < steps_total) wait_vmcnt<8>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (l_g < steps_total) issue_step();             // into the slot step g-1 used: its reads retired at this barrier
    }
    return;
  }

  if (wave < kMWaves) {
    // ================================================================ MFMA waves: 2 x 4 of 64 pixels x 32 channels
    const int r16 = lane & 15, q16 = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    f32x4 acc16[2][4], accI2[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc16[j][i][e] = 0.f; accI2[j][i][e] = 0.f; }
    const f16x8 kLow = {kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH, kH1UnscaleH};
    int g = 0;
    for (int k = 0; k < n_tiles; ++k) {
      for (int t = 0; t < T; ++t, ++g) {
        __builtin_amdgcn_s_barrier();
        if (t > 0 && (t & 7) == 0) {                   // the chain of the last eight K-steps joins the sum
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            acc16[n / 4][n % 4] += accI2[n / 4][n % 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) accI2[n / 4][n % 4][e] = 0.f;
          }
        }
        const unsigned char* sa = smem + (g % kS) * kStage;
        const unsigned char* sb = sa + 128 * 128;
        uint4 xp0[4], xp1[4], xw0[2], xw1[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xp0[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * 64 + i * 16 + r16, q16));
          xp1[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * 64 + i * 16 + r16, 4 + q16));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          xw0[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * 32 + j * 16 + r16, q16));
          xw1[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * 32 + j * 16 + r16, 4 + q16));
        }
#pragma unroll
        for (int idx = 0; idx < 24; ++idx) {           // product-major, like the generic kernel
          const int prod = idx / 8, n = idx % 8, j = n / 4, i = n % 4;
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
      // the tile's accumulators, raw, to the staging tile (the epilogue waves finished with the previous tile's before the
      // barrier of this tile's last K-step), and a clean slate for the next tile
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = acc16[j][i] + accI2[j][i];
          const int px = wm * 64 + i * 16 + r16;
          const int c = wn * 8 + j * 4 + q16;
          *reinterpret_cast<f32x4*>(smem + kRing + px * 512 + ((c ^ (px & 31)) << 4)) = v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc16[j][i][e] = 0.f; accI2[j][i][e] = 0.f; }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the staging tile is written before the next (raw) barrier
    }
    for (int t = 0; t < T; ++t) __builtin_amdgcn_s_barrier();     // the barriers the last tile's epilogue rides on
    return;
  }

  // ==================================================================== epilogue waves
  {
    const int e_wave = wave - (kMWaves + kLWaves);
    const int pl = lane >> 4, kch = lane & 15;         // pixel of the pass's four, 8-channel chunk
    const unsigned row_bytes = (unsigned)p.Co * 4u;
    // output and identity through buffer resources over exactly M rows: 32-bit offsets, and a row beyond the last pixel
    // (the tail of the last tile) is dropped / reads zeros by the range check -- no clamping, no predicate
    const unsigned out_bytes = (unsigned)p.M * row_bytes;
    const rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), 0, out_bytes, 0x00020000);
    // the lane's h0 chunk behind the first pixel of a pass; its h1 chunk 64 bytes on
    const unsigned lane_off = (unsigned)(4 * e_wave + pl) * row_bytes + (unsigned)(kch >> 2) * 128u + (unsigned)(kch & 3) * 16u;
    const unsigned stg_lane = (unsigned)(kRing + (4 * e_wave + pl) * 512);
    const bool relu = p.relu != 0;
    const bool has_res = p.res != nullptr;
    // passes per barrier interval: the eight passes of a tile within the first T-1 intervals of the next tile
    const int ppi = (kPasses + (T - 1) - 1) / (T - 1);
    for (int k = 0; k <= n_tiles; ++k) {               // during tile k's K-steps (the last round: the drain) tile k-1 is drained
      const bool work = k > 0;
      int m0 = 0, n0 = 0;
      if (work) tw.tile(k - 1, m0, n0);
      const unsigned tile_off = (unsigned)m0 * row_bytes + (unsigned)n0 * 4u;     // (wave-uniform)
      float4 sc[2], sh[2];
      u32x4 idt[kAhead][2] = {};                       // identity chunks of the passes in flight: pass q's live in idt[q % kAhead]
      int done = 0;
      for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_barrier();
        if (!work) continue;
        if (t == 0) {
          // this tile's scale / shift for the lane's eight channels, then the identity chunks of the first kAhead passes; each
          // pass requests those of the pass kAhead further on (untracked loads, counted waits: e_younger_ops)
          const float* scp = p.scale + n0 + kch * 8;
          const float* shp = p.shift + n0 + kch * 8;
          sc[0] = *reinterpret_cast<const float4*>(scp); sc[1] = *reinterpret_cast<const float4*>(scp + 4);
          sh[0] = *reinterpret_cast<const float4*>(shp); sh[1] = *reinterpret_cast<const float4*>(shp + 4);
          if (has_res) {
#pragma unroll
            for (int q = 0; q < kAhead; ++q) {
              const unsigned off = lane_off + (tile_off + (unsigned)(16 * q) * row_bytes);
              buffer_load16_untracked(idt[q][0], rrsrc, off);
              buffer_load16_untracked(idt[q][1], rrsrc, off + 64u);
            }
          }
        }
        const int upto = (t + 1) * ppi < kPasses ? (t + 1) * ppi : kPasses;
        if (t == T - 1 || done >= upto) continue;      // the last interval belongs to the MFMA waves' staging writes
#pragma unroll 1
        for (int qq = 0; qq < kPasses / kAhead; ++qq)
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
          const int q = qq * kAhead + u;                // (u, hence the identity buffer idt[u], is a compile-time constant)
          if (q < done || q >= upto) continue;
          const int row = 4 * (4 * q + e_wave) + pl;    // pixel of the tile
          const unsigned char* st = smem + stg_lane + q * (16 * 512);
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + (((2 * kch) ^ (row & 31)) << 4));
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + (((2 * kch + 1) ^ (row & 31)) << 4));
          float v[8];
          v[0] = __builtin_fmaf(a0[0], sc[0].x, sh[0].x); v[1] = __builtin_fmaf(a0[1], sc[0].y, sh[0].y);
          v[2] = __builtin_fmaf(a0[2], sc[0].z, sh[0].z); v[3] = __builtin_fmaf(a0[3], sc[0].w, sh[0].w);
          v[4] = __builtin_fmaf(a1[0], sc[1].x, sh[1].x); v[5] = __builtin_fmaf(a1[1], sc[1].y, sh[1].y);
          v[6] = __builtin_fmaf(a1[2], sc[1].z, sh[1].z); v[7] = __builtin_fmaf(a1[3], sc[1].w, sh[1].w);
          const unsigned off = lane_off + (tile_off + (unsigned)(16 * q) * row_bytes);
          if (has_res) {
            // counted wait for this pass's identity chunks (qq is the only run-time part of q)
            if (qq == 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(idt[u][0]), "+v"(idt[u][1]) : "n"(e_younger_ops(u, kAhead, kPasses)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(idt[u][0]), "+v"(idt[u][1]) : "n"(e_younger_ops_later(u, kAhead, kPasses)) : "memory");
            float idv[8];
            join16x8(__builtin_bit_cast(uint4, idt[u][0]), __builtin_bit_cast(uint4, idt[u][1]), idv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += idv[e];
            if (q + kAhead < kPasses) {
              const unsigned offn = off + (unsigned)(16 * kAhead) * row_bytes;
              buffer_load16_untracked(idt[u][0], rrsrc, offn);
              buffer_load16_untracked(idt[u][1], rrsrc, offn + 64u);
            }
          }
          if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_elementwise_maximum(v[e], 0.f);
          }
          uint4 o, o1;
          split16x8(v, o, o1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), yrsrc, off + 64u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, off, 0, 0);
          __builtin_amdgcn_sched_barrier(0);            // one pass after the other: interleaved they do not fit 128 registers
        }
        done = upto;
      }
    }
  }
}

}  // namespace

// tile 20 of the menu: the streaming 1x1 kernel (f16x2).  Worth it where a CU gets several tiles (>= 512 tiles of 128 x 128).
bool conv_stream_eligible(int precision, int k, int pad, int Ci, int Co, int M, int ksteps, bool stem) {
  return precision == 2 && !stem && k == 1 && pad == 0 && Ci % 32 == 0 && Co % 128 == 0 && ksteps >= 2 &&
         (long long)((M + 127) / 128) * (Co / 128) >= 512;
}

hipError_t launch_conv1x1_stream(const ConvArgs& a, hipStream_t s) {
  if (a.x_bytes == 0 || a.x_bytes >= kOutOfRange || a.w_bytes == 0 || a.w_bytes >= kOutOfRange) return hipErrorInvalidValue;
  if (!conv_stream_eligible(2, a.KH, a.pad, a.Ci, a.Co, a.M, a.ksteps, a.stem != 0) || a.KW != 1 || a.ksteps != a.Ci * 4 / 128)
    return hipErrorInvalidValue;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << dev, std::memory_order_release);
  }
  hipLaunchKernelGGL(conv1x1_stream_kernel, dim3(256), dim3(1024), kLds, s, a);
  return hipGetLastError();
}

}  // namespace nbc
