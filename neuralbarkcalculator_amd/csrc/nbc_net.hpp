// Network description and packed-weight layout shared by the packer (host only) and the
// launcher.  The layer list restates /root/reference/src/bark_calculator/models.py:127-139
// (torchvision resnet50 with replace_stride_with_dilation=[False,True,True], cut at layer4)
// and models.py:113-124 (FCNHead); see neuralbarkcalculator_amd/topology.py for the same
// table in Python.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace nbc {

constexpr int kNumClasses = 3;
constexpr float kBnEps = 1e-5f;
constexpr int kKStepBytes = 128;   // one K-step of the implicit GEMM = 128 bytes of K per row
constexpr int kChunkBytes = 16;    // one lane-load

struct ConvUnit {
  std::string name;      // "backbone.layer1.0.conv1"
  std::string bn;        // "" when there is no BatchNorm (classifier.4)
  int cin, cout, k, stride, pad, dil;
  bool relu, bias, residual;
  int block_first;       // 1 when this is conv1 of a bottleneck (plan building)
};

struct StateKey {
  std::string name;
  int64_t shape[4];
  int ndim;
  int dtype;             // 0 f32, 1 i64
};

const std::vector<ConvUnit>& conv_units();
const std::vector<StateKey>& state_keys();

// bytes per activation / weight element: f32 4, bf16 2, f16x2 4 (two f16 pieces; a 128-byte group holds 32 channels:
// [h0 x 32][h1 x 32], so tensors, K-steps and LDS rows have the f32 mode's geometry)
inline int elem_bytes(int precision) { return precision == 1 ? 2 : 4; }
inline bool known_precision(int precision) { return precision >= 0 && precision <= 2; }

// Packed layout of one conv unit inside the blob.
struct PackedConv {
  size_t w_off;          // weights: [cout][ksteps*128 bytes]
  size_t scale_off;      // float[cout]
  size_t shift_off;      // float[cout]
  int cin_pad;           // channels per input pixel as the kernel sees them
  int ksteps;            // K-steps of 128 bytes
  bool stem;             // one 16-byte chunk per tap (cin_pad*elem = 16 bytes)
  bool head;             // classifier.4: weights kept f32 [3][512], shift = bias
};

// Trailer of the blob: what a rank that receives the blob by broadcast must know besides the panels.
//   int32 meta[kMetaWords]: [0] kMetaMagic, [1] NBC_PACK_* flags, [2] number of conv units,
//   [kMetaExpBase + u] the power of two the OUTPUT tensor of conv unit u is stored with (f16x2; 0 elsewhere): see
//   activation_exponents in nbc_net.cpp
constexpr int kMetaWords = 256;
constexpr int kMetaExpBase = 8;
constexpr int32_t kMetaMagic = 0x4e424335;   // "NBC5"

struct PackedLayout {
  std::vector<PackedConv> convs;
  size_t meta_off;       // int32[kMetaWords]
  size_t total_bytes;
};

PackedLayout packed_layout(int precision);

}  // namespace nbc
