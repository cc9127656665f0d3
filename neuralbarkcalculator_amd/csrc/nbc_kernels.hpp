// Launchers of the gfx950 kernels (definitions in conv_igemm_dma.hip, pointwise.hip and small_zones.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace nbc {

// Implicit-GEMM convolution on NHWC activations, fused  y = relu?(acc*scale + shift (+ res)).
// GEMM view: rows m = (image, oy, ox), columns n = cout, K = (kh, kw, ci) in 128-byte K-steps.
struct ConvArgs {
  const void* x;        // [N][Hi][Wi][Ci] elements
  const void* w;        // [Co][ksteps*128 bytes]
  const float* scale;   // [Co]
  const float* shift;   // [Co]
  const void* res;      // nullable, [M][Co] elements (the identity of a bottleneck)
  void* y;              // [M][Co] elements
  unsigned x_bytes;     // size of x in bytes (< 2 GiB): the buffer resource's range check zero-fills halo and tail lanes
  unsigned w_bytes;     // size of w in bytes
  int N, Hi, Wi, Ci;
  int Ho, Wo, Co;
  int KH, KW, stride, pad, dil;
  int M;                // N*Ho*Wo
  int ksteps;
  int relu;
  int stem;             // one 16-byte chunk per tap (Ci*elem == 16 bytes)
  int wo_shift;         // log2(Wo) when Wo is a power of two, else -1
  int hw_shift;         // log2(Ho*Wo) when it is a power of two, else -1 (batches: image index without a division)
#ifdef NBC_STAMPS
  unsigned long long* stamps;   // diagnostic build only (tools/conv_timeline.hip): 8 stamps per block
#endif
};

// precision: 0 = f32 (v_mfma_f32_32x32x2_f32), 1 = bf16 (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16),
// 2 = f16x2 (two f16 pieces per f32 value, three v_mfma_f32_16x16x32_f16 per product: split16.hpp)
// LDS-DMA ring (conv_igemm_dma.hip).  tile < 0 = choose_conv_tile(M, Co, K, precision).
constexpr int CONV_TILE_COUNT = 21;   // tile menu: see launch_tile() in conv_igemm_dma.hip; 18, 19, 20 = the row-resident 3x3 kernel
constexpr int CONV_TILE_ROWS_FIRST = 18;   // (conv3x3_rows.hip: one image row x 128 / 64 channels; 20: two rows x 64 channels)
int conv_tile_rows(int tile);
int conv_tile_cols(int tile);
// Whether a convolution runs on the row-resident 3x3 kernels (conv3x3_rows.hip; f16x2, 3x3, stride 1, no identity): 0 no;
// 1: 128-pixel-wide maps, >= 256 output channels: tile 18 ONLY; 2: 128-pixel-wide maps, 64 / 128 output channels: tile 19 ONLY.  A property of the layer and its shape that fixes its K order; every other convolution runs on
// tiles 0 .. 17 only.
int conv_rows_kind(int precision, int k, int stride, int pad, int dil, int Hi, int Wi, int Ho, int Wo, int Ci, int Co, bool has_res);
bool conv_tile_ok(int precision, int tile, int Co, int rows_kind);   // the tile exists for the precision and the kind of convolution and divides Co
// K = Cin*kh*kw; the default tile of a layer (cost model); rows_kind: conv_rows_kind
int choose_conv_tile(int M, int Co, int K, int precision, int rows_kind);
hipError_t launch_conv_dma(const ConvArgs& a, int precision, int tile, hipStream_t s);
hipError_t launch_conv3x3_rows(const ConvArgs& a, int rows_tile, hipStream_t s);   // rows_tile: tile id - CONV_TILE_ROWS_FIRST

// float32 NCHW [N,3,H,W] -> NHWC elements padded to 16 bytes per pixel.
hipError_t launch_ingest_f32(const float* x, void* y, int N, int H, int W, int precision, hipStream_t s);
// uint8 NHWC [N,H,W,3] -> same, applying (u8/255 - mean)/std in f32 (dataset.py:175-186).
hipError_t launch_ingest_u8(const uint8_t* x, void* y, int N, int H, int W, const float mean[3],
                            const float stdv[3], int precision, hipStream_t s);
// MaxPool2d(3, stride 2, padding 1) on NHWC, C a multiple of the 16-byte chunk.
hipError_t launch_maxpool3x3s2(const void* x, void* y, int N, int Hi, int Wi, int C, int Ho, int Wo,
                               int precision, hipStream_t s);
// classifier.4: 1x1 conv 512 -> 3 with bias; f32 weights [3][512]; output f32 NCHW [N,3,h,w].
// counts_zero (nullable, 3*N <= 256 counters): cleared by this launch for the upsample/argmax launch
// that follows it, which accumulates the per-class pixel counts there (saves a memset node).
// nonfinite (nullable): one device word that gets bit 0 set when a logit is NaN or infinite.
hipError_t launch_head1x1(const void* x, const float* w, const float* bias, float* y, int N, int hw,
                          int precision, unsigned long long* counts_zero, unsigned* nonfinite, hipStream_t s);
// Bicubic (A=-0.75, align_corners=False) upsample of f32 NCHW [N,3,h,w] to HxW, fused with the
// per-pixel argmax, the optional 2->1 remap and the per-class pixel counts.
hipError_t launch_upsample_argmax(const float* lowres, int N, int h, int w, int H, int W,
                                  float* logits_full, void* labels, int labels_i64,
                                  unsigned long long* counts, int exclude_nodes, hipStream_t s);
// remove_small_zones (utils.py:135-148) in place on device labels (u8 or i64, [N,H,W]): 8-connected
// components below min_pixels of the non-background, then of the filled background, flip; optional
// 2 -> 1 remap and per-class counts afterwards.  Workspaces: bg N*H*W bytes, parent and size N*H*W ints.
hipError_t launch_remove_small_zones(void* labels, int labels_i64, int N, int H, int W, int min_pixels, int exclude_nodes,
                                     unsigned char* bg, int* parent, int* size, unsigned long long* counts, hipStream_t s);
// Preprocessor (models.py:191-203): uint8 HWC [H,W,3] -> ToTensor -> skimage cubic resize with reflected borders,
// clipped to the input range -> any of: float32 HWC [out_h,out_w,3]; its uint8 form as imsave writes it; per output
// row the number of pixels trim_black counts as lit.  minmax: two uints of scratch.
hipError_t launch_resize_cubic_u8(const uint8_t* src, int H, int W, float* dst, uint8_t* dst_u8, int* row_lit, int out_h, int out_w,
                                  unsigned* minmax, hipStream_t s);
// NHWC elements -> float32 NCHW (debug read-back of activations).
hipError_t launch_nhwc_to_nchw_f32(const void* x, float* y, int N, int H, int W, int C, int precision,
                                   hipStream_t s);

// Largest finite |value| of `elems` stored activation elements (C channels per pixel) as float bits, atomicMax-ed into *out.
hipError_t launch_absmax(const void* x, size_t elems, int C, int precision, unsigned* out, hipStream_t s);

}  // namespace nbc
