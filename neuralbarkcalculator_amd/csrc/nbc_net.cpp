// Host-only part of libnbc_hip.so: topology tables, state_dict checking and weight packing.
// No HIP calls in this file (it is also what the CPU-only tests exercise).
#include "nbc_net.hpp"

#include <algorithm>
#include <cmath>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstdio>
#include <cstring>
#include <map>
#include <set>

#include "../../include/nbc.h"
#include "nbc_internal.hpp"

namespace nbc {

static std::vector<ConvUnit> build_units() {
  std::vector<ConvUnit> u;
  u.push_back({"backbone.conv1", "backbone.bn1", 3, 64, 7, 2, 3, 1, true, false, false, 0});
  int inplanes = 64, dilation = 1;
  const int planes_[4] = {64, 128, 256, 512};
  const int blocks_[4] = {3, 4, 6, 3};
  const int stride_[4] = {1, 2, 2, 2};
  const bool dilate_[4] = {false, false, true, true};
  for (int li = 0; li < 4; ++li) {
    int planes = planes_[li], stride = stride_[li];
    int prev_dil = dilation;
    if (dilate_[li]) { dilation *= stride; stride = 1; }
    for (int bi = 0; bi < blocks_[li]; ++bi) {
      std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      int s = bi == 0 ? stride : 1;
      int d = bi == 0 ? prev_dil : dilation;
      u.push_back({p + ".conv1", p + ".bn1", inplanes, planes, 1, 1, 0, 1, true, false, false, 1});
      u.push_back({p + ".conv2", p + ".bn2", planes, planes, 3, s, d, d, true, false, false, 0});
      if (bi == 0)
        u.push_back({p + ".downsample.0", p + ".downsample.1", inplanes, planes * 4, 1, s, 0, 1,
                     false, false, false, 0});
      u.push_back({p + ".conv3", p + ".bn3", planes, planes * 4, 1, 1, 0, 1, true, false, true, 0});
      inplanes = planes * 4;
    }
  }
  u.push_back({"classifier.0", "classifier.1", 2048, 512, 3, 1, 1, 1, true, false, false, 0});
  u.push_back({"classifier.4", "", 512, kNumClasses, 1, 1, 0, 1, false, true, false, 0});
  return u;
}

const std::vector<ConvUnit>& conv_units() {
  static const std::vector<ConvUnit> u = build_units();
  return u;
}

static std::vector<StateKey> build_keys() {
  // nn.Module.state_dict() order: inside a Bottleneck conv1,bn1,conv2,bn2,conv3,bn3,downsample.
  std::vector<StateKey> keys;
  const auto& units = conv_units();
  std::vector<const ConvUnit*> ordered;
  for (size_t i = 0; i < units.size(); ++i) {
    const ConvUnit& c = units[i];
    if (c.name.size() > 13 && c.name.compare(c.name.size() - 13, 13, ".downsample.0") == 0) continue;
    ordered.push_back(&c);
    if (c.residual && i >= 1) {
      const ConvUnit& prev = units[i - 1];
      if (prev.name.size() > 13 && prev.name.compare(prev.name.size() - 13, 13, ".downsample.0") == 0)
        ordered.push_back(&prev);
    }
  }
  auto add = [&](const std::string& n, std::initializer_list<int64_t> shp, int dtype) {
    StateKey k;
    k.name = n;
    k.ndim = (int)shp.size();
    k.dtype = dtype;
    int i = 0;
    for (int j = 0; j < 4; ++j) k.shape[j] = 1;
    for (int64_t s : shp) k.shape[i++] = s;
    keys.push_back(k);
  };
  for (const ConvUnit* c : ordered) {
    add(c->name + ".weight", {c->cout, c->cin, c->k, c->k}, 0);
    if (c->bias) add(c->name + ".bias", {c->cout}, 0);
    if (!c->bn.empty()) {
      add(c->bn + ".weight", {c->cout}, 0);
      add(c->bn + ".bias", {c->cout}, 0);
      add(c->bn + ".running_mean", {c->cout}, 0);
      add(c->bn + ".running_var", {c->cout}, 0);
      add(c->bn + ".num_batches_tracked", {}, 1);
    }
  }
  return keys;
}

const std::vector<StateKey>& state_keys() {
  static const std::vector<StateKey> k = build_keys();
  return k;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

PackedLayout packed_layout(int precision) {
  PackedLayout L;
  const int eb = elem_bytes(precision);
  size_t off = 0;
  for (const ConvUnit& c : conv_units()) {
    PackedConv p{};
    p.stem = (c.cin == 3);
    p.head = c.bn.empty();
    if (p.head) {                      // f32 [3][512] + bias
      p.cin_pad = c.cin;
      p.ksteps = 0;
      p.w_off = off;
      off = align_up(off + (size_t)c.cout * c.cin * 4, 256);
      p.scale_off = off;               // unused
      p.shift_off = off;               // bias
      off = align_up(off + (size_t)c.cout * 4, 256);
    } else {
      if (p.stem) {
        // one 16-byte chunk per tap; a K-step is one kernel ROW: its kw taps + zero slots up to eight (k <= 8),
        // so a K-step of an output pixel reads consecutive input pixels (one 128-byte segment)
        p.cin_pad = kChunkBytes / eb;
        p.ksteps = c.k;
      } else {
        p.cin_pad = c.cin;
        p.ksteps = c.k * c.k * c.cin * eb / kKStepBytes;
      }
      p.w_off = off;
      off = align_up(off + (size_t)c.cout * p.ksteps * kKStepBytes, 256);
      p.scale_off = off;
      off = align_up(off + (size_t)c.cout * 4, 256);
      p.shift_off = off;
      off = align_up(off + (size_t)c.cout * 4, 256);
    }
    L.convs.push_back(p);
  }
  L.meta_off = off;
  off = align_up(off + (size_t)kMetaWords * 4, 256);
  L.total_bytes = off;
  return L;
}

uint16_t f32_to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // keep NaN a NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                               // RNE
  return (uint16_t)(u >> 16);
}

uint16_t f32_to_f16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // >= 65520 rounds to infinity (and infinity itself)
  if (a < 0x33000001u) return sign;                                 // <= 2^-25: rounds to zero (2^-25 is the tie to even 0)
  const int e = (int)(a >> 23) - 127;                               // unbiased exponent, -24 .. 15 here
  uint32_t m = (a & 0x007fffffu) | 0x00800000u;                     // 24-bit significand
  int shift = e >= -14 ? 13 : 13 + (-14 - e);                       // bits dropped (subnormal results drop more)
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;                 // round to nearest even (a carry runs into the exponent)
  if (e >= -14) return (uint16_t)(sign | (uint16_t)(((uint32_t)(e + 15) << 10) + (q - 0x400u)));
  return (uint16_t)(sign | (uint16_t)q);                            // subnormal (q == 0x400 is the smallest normal)
}

float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const int e = (h >> 10) & 0x1f;
  const uint32_t m = h & 0x3ffu;
  float out;
  uint32_t u;
  if (e == 0x1f) u = sign | 0x7f800000u | (m << 13);
  else if (e != 0) u = sign | ((uint32_t)(e - 15 + 127) << 23) | (m << 13);
  else {
    out = (float)m * 5.9604644775390625e-08f;                       // m * 2^-24, exact
    std::memcpy(&u, &out, 4);
    u |= sign;
  }
  std::memcpy(&out, &u, 4);
  return out;
}

void split_f16x2(float x, uint16_t* h0, uint16_t* h1) {
  *h0 = f32_to_f16(x);
  *h1 = f32_to_f16((x - f16_to_f32(*h0)) * 2048.0f);                // the difference is exact, the scaling a power of two
}

// In-place split of packed f32 weight rows into the f16x2 group layout: every `group` consecutive floats (32 = one
// 128-byte K-step row segment; 4 = the stem's 16-byte tap) become [h0 x group][h1 x group].  33 M weights: the x86 F16C
// conversion (round to nearest even, subnormals: IEEE, like split_f16x2) where the host has it, else the portable one.
#if defined(__x86_64__)
__attribute__((target("avx,f16c"))) static void split_groups_f16c(float* data, size_t nfloats, int group, float low_scale) {
  alignas(32) uint16_t h0[32], h1[32];
  const __m256 scale = _mm256_set1_ps(low_scale);
  for (size_t g = 0; g + group <= nfloats; g += group) {
    float* x = data + g;
    if (group == 32) {
      for (int i = 0; i < 32; i += 8) {
        const __m256 v = _mm256_loadu_ps(x + i);
        const __m128i a = _mm256_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const __m256 r = _mm256_mul_ps(_mm256_sub_ps(v, _mm256_cvtph_ps(a)), scale);
        _mm_store_si128(reinterpret_cast<__m128i*>(h0 + i), a);
        _mm_store_si128(reinterpret_cast<__m128i*>(h1 + i), _mm256_cvtps_ph(r, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
      }
      std::memcpy(x, h0, 64);
      std::memcpy(reinterpret_cast<unsigned char*>(x) + 64, h1, 64);
    } else {                                           // group 4
      const __m128 v = _mm_loadu_ps(x);
      const __m128i a = _mm_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
      const __m128 r = _mm_mul_ps(_mm_sub_ps(v, _mm_cvtph_ps(a)), _mm_set1_ps(low_scale));
      const __m128i b = _mm_cvtps_ph(r, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
      const uint64_t lo = (uint64_t)_mm_cvtsi128_si64(a), hi = (uint64_t)_mm_cvtsi128_si64(b);
      std::memcpy(x, &lo, 8);
      std::memcpy(reinterpret_cast<unsigned char*>(x) + 8, &hi, 8);
    }
  }
}
#endif

// low_scale: what the low piece is multiplied by before it is rounded (2^11 for activations, 1 for the normalised weights)
static void split_groups(float* data, size_t nfloats, int group, float low_scale) {
#if defined(__x86_64__)
  if (__builtin_cpu_supports("f16c") && __builtin_cpu_supports("avx")) { split_groups_f16c(data, nfloats, group, low_scale); return; }
#endif
  uint16_t h0[32], h1[32];
  for (size_t g = 0; g + group <= nfloats; g += group) {
    for (int i = 0; i < group; ++i) {
      h0[i] = f32_to_f16(data[g + i]);
      h1[i] = f32_to_f16((data[g + i] - f16_to_f32(h0[i])) * low_scale);
    }
    std::memcpy(data + g, h0, (size_t)group * 2);
    std::memcpy(reinterpret_cast<unsigned char*>(data + g) + group * 2, h1, (size_t)group * 2);
  }
}

// The power of two k that nbc_pack_weights multiplies a weight row by in f16x2 mode: largest finite |w| * 2^k in
// [2^14, 2^15), the top of f16's range (65504), which leaves the most room below: the high piece P = f16(w 2^k) and the
// low piece Q = f16(w 2^k - P) stay normal f16 numbers for weights down to 2^-15 of the row's largest, and so does
// P * 2^-11, which the kernel forms for the third product.  0 for a row of zeros (or of nothing finite).  Clamped to
// [-66, 80] so that the BatchNorm scale the inverse is folded into stays a normal f32 for any gamma / sqrt(var + eps)
// between 2^-46 and 2^60; a row beyond the clamp keeps the rest of its exponent.
int f16x2_row_exponent(const float* row, size_t n, bool* clamped) {
  if (clamped) *clamped = false;
  float m = 0.f;
  for (size_t i = 0; i < n; ++i) {
    const float a = std::fabs(row[i]);
    if (std::isfinite(a) && a > m) m = a;
  }
  if (m == 0.f) return 0;
  int e = 0;
  (void)std::frexp(m, &e);                            // m = f * 2^e, f in [0.5, 1)
  const int k = 15 - e;                               // m * 2^k in [2^14, 2^15)
  if (clamped) *clamped = k > 80 || k < -66;
  return k > 80 ? 80 : (k < -66 ? -66 : k);
}

// f16x2: the power of two every activation tensor is STORED with.  An activation between a BatchNorm (+ ReLU, + max-pool:
// positively homogeneous) and the next convolution is scale-free, so a checkpoint (models.py:222 takes any) may hold tensors
// of any magnitude, and f16 pieces cannot: below 2^-12 the low piece of a value is an f16 subnormal (an ABSOLUTE error of
// 2^-36, which is everything once a whole tensor sits down there), beyond 65 504 the high piece overflows.  The pieces of an
// activation are formed on the fly, so the cure sits in the f32 epilogue that produces the tensor: its magnitude is
// estimated from the producing BatchNorm, est = max over channels of |beta| + 3 |gamma| sqrt(var / (var + eps)) (what a channel
// reaches at three standard deviations when the running statistics describe the data: they do for a trained checkpoint and for every
// rescaling of one), a tensor whose estimate lies outside [2^-5, 2^7] gets the power of two 2^a that brings it to [2, 4),
// and the launch that PRODUCES the tensor applies scale 2^a, shift 2^a (its f32 BatchNorm pair), every launch that READS it
// scale 2^-a: fma(2^a_in acc, 2^(a_out - a_in) scale, 2^a_out shift) = 2^a_out fma(acc, scale, shift), exact.  A residual
// stream is ONE tensor as far as this goes (bn3 of every block of a stage and the downsample BatchNorm of its first add into
// it): one power per stage.  Tensors of ordinary size keep a = 0, so an ordinary checkpoint computes what it computed before
// this existed, bit for bit.  classifier.4 reads the last tensor with f32 weights: they take the 2^-a.
// out_exp[u] / in_exp[u]: power of conv unit u's output tensor / of the tensor it reads.
static int tensor_exponent(float est) {
#ifdef NBC_NO_ACT_EXP
  return 0;                                           // tool builds only (scripts/act_floor_probe.py): the library before this existed
#endif
  if (!(est > 0.f) || !std::isfinite(est)) return 0;
  if (est >= 0.03125f && est <= 128.0f) return 0;
  int e = 0;
  (void)std::frexp(est, &e);                          // est = f * 2^e, f in [0.5, 1)
  const int a = 2 - e;                                // est * 2^a in [2, 4)
  return a > 100 ? 100 : (a < -100 ? -100 : a);
}

// gamma counts with the share of the normalised value that is data: a channel whose running variance lies below eps comes
// out of the BatchNorm with a standard deviation of gamma sqrt(var / (var + eps)), not gamma
static float bn_estimate(const float* gamma, const float* beta, const float* var, int n) {
  float m = 0.f;
  for (int i = 0; i < n; ++i) {
    const float vr = var[i] > 0.f ? var[i] : 0.f;
    const float v = std::fabs(beta[i]) + 3.0f * std::fabs(gamma[i]) * std::sqrt(vr / (vr + kBnEps));
    if (std::isfinite(v) && v > m) m = v;
  }
  return m;
}

static void activation_exponents(const std::vector<const float*>& gamma, const std::vector<const float*>& beta,
                          const std::vector<const float*>& var, std::vector<int>& out_exp, std::vector<int>& in_exp) {
  const auto& units = conv_units();
  const size_t n = units.size();
  out_exp.assign(n, 0);
  in_exp.assign(n, 0);
  auto est = [&](size_t u) { return bn_estimate(gamma[u], beta[u], var[u], units[u].cout); };
  int cur = tensor_exponent(est(0));                  // the stem's output (the max-pool keeps it)
  out_exp[0] = cur;
  size_t ui = 1;
  while (ui < n && units[ui].block_first) {
    // the units of one stage: bottlenecks up to (not including) the next one that has a downsample branch
    size_t end = ui;
    std::vector<size_t> c1, c2, c3, ds;
    bool first = true;
    while (end < n && units[end].block_first) {
      const bool has_ds = !units[end + 2].residual;
      if (has_ds && !first) break;
      c1.push_back(end); c2.push_back(end + 1);
      if (has_ds) { ds.push_back(end + 2); c3.push_back(end + 3); end += 4; }
      else { c3.push_back(end + 2); end += 3; }
      first = false;
    }
    float stream = 0.f;
    for (size_t u : c3) stream = std::max(stream, est(u));
    for (size_t u : ds) stream = std::max(stream, est(u));
    const int a_stream = tensor_exponent(stream);
    for (size_t b = 0; b < c1.size(); ++b) {
      in_exp[c1[b]] = b == 0 ? cur : a_stream;
      out_exp[c1[b]] = tensor_exponent(est(c1[b]));
      in_exp[c2[b]] = out_exp[c1[b]];
      out_exp[c2[b]] = tensor_exponent(est(c2[b]));
      in_exp[c3[b]] = out_exp[c2[b]];
      out_exp[c3[b]] = a_stream;
    }
    for (size_t u : ds) { in_exp[u] = cur; out_exp[u] = a_stream; }
    cur = a_stream;
    ui = end;
  }
  in_exp[ui] = cur;                                   // classifier.0
  out_exp[ui] = tensor_exponent(est(ui));
  in_exp[ui + 1] = out_exp[ui];                       // classifier.4: f32 weights, no BatchNorm
  out_exp[ui + 1] = 0;
}

// v * 2^e for a BatchNorm scale / shift; NBC_PACK_SCALE_RANGE when that leaves f32's normal range (the product is then no
// longer the exact power-of-two multiple the normalisations rest on)
static float ldexp_flagged(float v, int e, int* flags) {
  if (e == 0) return v;
  const float r = std::ldexp(v, e);
  if (std::isfinite(v) && v != 0.f && (!std::isfinite(r) || std::fabs(r) < 1.17549435e-38f)) *flags |= NBC_PACK_SCALE_RANGE;
  return r;
}

thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

}  // namespace nbc

using namespace nbc;

extern "C" {

const char* nbc_last_error(void) { return g_last_error.c_str(); }
const char* nbc_version(void) { return "nbc-hip 0.1 (gfx950)"; }

int nbc_num_convs(void) { return (int)conv_units().size(); }

int nbc_conv_info(int index, nbc_conv_desc* out) {
  const auto& u = conv_units();
  if (!out || index < 0 || index >= (int)u.size()) return set_error(NBC_ERR_INVALID, "nbc_conv_info: bad index");
  const ConvUnit& c = u[index];
  std::memset(out, 0, sizeof(*out));
  std::snprintf(out->name, sizeof(out->name), "%s", c.name.c_str());
  std::snprintf(out->bn, sizeof(out->bn), "%s", c.bn.c_str());
  out->cin = c.cin; out->cout = c.cout; out->k = c.k; out->stride = c.stride;
  out->pad = c.pad; out->dil = c.dil;
  out->relu = c.relu; out->bias = c.bias; out->residual = c.residual;
  return NBC_OK;
}

int nbc_num_state_keys(void) { return (int)state_keys().size(); }

int nbc_state_key(int index, const char** name, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  const auto& k = state_keys();
  if (index < 0 || index >= (int)k.size()) return set_error(NBC_ERR_INVALID, "nbc_state_key: bad index");
  if (name) *name = k[index].name.c_str();
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = k[index].shape[i];
  if (ndim) *ndim = k[index].ndim;
  if (dtype) *dtype = k[index].dtype;
  return NBC_OK;
}

int nbc_lowres_size(int H, int W, int* h, int* w) {
  if (H < 1 || W < 1) return set_error(NBC_ERR_INVALID, "nbc_lowres_size: H,W must be >= 1");
  for (int i = 0; i < 3; ++i) { H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1; }
  if (h) *h = H;
  if (w) *w = W;
  return NBC_OK;
}

int nbc_split_f16x2(const float* x, size_t n, uint16_t* h0, uint16_t* h1) {
  if (!x || !h0 || !h1) return set_error(NBC_ERR_INVALID, "nbc_split_f16x2: null argument");
  for (size_t i = 0; i < n; ++i) split_f16x2(x[i], &h0[i], &h1[i]);
  return NBC_OK;
}

size_t nbc_packed_weights_bytes(int precision) {
  if (!known_precision(precision)) return 0;
  return packed_layout(precision).total_bytes;
}

int nbc_pack_weights(const nbc_tensor* tensors, int n, int precision, void* blob, size_t blob_bytes) {
  if (!known_precision(precision)) return set_error(NBC_ERR_INVALID, "nbc_pack_weights: unknown precision");
  if (!tensors || n < 0 || !blob) return set_error(NBC_ERR_INVALID, "nbc_pack_weights: null argument");
  const PackedLayout L = packed_layout(precision);
  if (blob_bytes < L.total_bytes) return set_error(NBC_ERR_INVALID, "nbc_pack_weights: blob too small");

  // --- strict key / shape check, in the manner of nn.Module.load_state_dict (models.py:222)
  std::map<std::string, const nbc_tensor*> given;
  std::string unexpected, missing, badshape;
  std::set<std::string> expected;
  for (const StateKey& k : state_keys()) expected.insert(k.name);
  for (int i = 0; i < n; ++i) {
    if (!tensors[i].name) return set_error(NBC_ERR_INVALID, "nbc_pack_weights: tensor without a name");
    if (!expected.count(tensors[i].name)) unexpected += std::string(" \"") + tensors[i].name + "\"";
    given[tensors[i].name] = &tensors[i];
  }
  for (const StateKey& k : state_keys()) {
    auto it = given.find(k.name);
    if (it == given.end()) { missing += " \"" + k.name + "\""; continue; }
    const nbc_tensor* t = it->second;
    bool ok = (t->ndim == k.ndim) && t->dtype == k.dtype && (t->data != nullptr);
    for (int j = 0; ok && j < k.ndim; ++j) ok = (t->shape[j] == k.shape[j]);
    if (!ok) badshape += " \"" + k.name + "\"";
  }
  if (!missing.empty() || !unexpected.empty() || !badshape.empty()) {
    std::string msg = "Error(s) in loading state_dict for fcn_resnet50:";
    if (!missing.empty()) msg += " Missing key(s) in state_dict:" + missing + ".";
    if (!unexpected.empty()) msg += " Unexpected key(s) in state_dict:" + unexpected + ".";
    if (!badshape.empty()) msg += " size or dtype mismatch for:" + badshape + ".";
    return set_error(NBC_ERR_KEYS, msg);
  }

  std::memset(blob, 0, L.total_bytes);
  unsigned char* base = static_cast<unsigned char*>(blob);
  const int eb = elem_bytes(precision);
  const auto& units = conv_units();
  int flags = 0;
  std::vector<int> out_exp(units.size(), 0), in_exp(units.size(), 0);
  if (precision == NBC_PREC_F16X2) {
    std::vector<const float*> gam(units.size(), nullptr), bet(units.size(), nullptr), var(units.size(), nullptr);
    for (size_t ui = 0; ui < units.size(); ++ui) {
      const ConvUnit& c = units[ui];
      if (c.bn.empty()) continue;                                           // classifier.4: never asked for an estimate
      gam[ui] = static_cast<const float*>(given[c.bn + ".weight"]->data);
      bet[ui] = static_cast<const float*>(given[c.bn + ".bias"]->data);
      var[ui] = static_cast<const float*>(given[c.bn + ".running_var"]->data);
    }
    activation_exponents(gam, bet, var, out_exp, in_exp);
  }
  for (size_t ui = 0; ui < units.size(); ++ui) {
    const ConvUnit& c = units[ui];
    const PackedConv& p = L.convs[ui];
    const float* w = static_cast<const float*>(given[c.name + ".weight"]->data);
    if (p.head) {
      float* hw = reinterpret_cast<float*>(base + p.w_off);
      for (size_t e = 0; e < (size_t)c.cout * c.cin; ++e) hw[e] = ldexp_flagged(w[e], -in_exp[ui], &flags);   // f16x2: its input's power off
      std::memcpy(base + p.shift_off, given[c.name + ".bias"]->data, (size_t)c.cout * 4);
      continue;
    }
    const size_t row_bytes = (size_t)p.ksteps * kKStepBytes;
    for (int o = 0; o < c.cout; ++o) {
      unsigned char* row = base + p.w_off + (size_t)o * row_bytes;
      for (int kh = 0; kh < c.k; ++kh)
        for (int kw = 0; kw < c.k; ++kw)
          for (int ci = 0; ci < c.cin; ++ci) {
            const float v = w[(((size_t)o * c.cin + ci) * c.k + kh) * c.k + kw];
            const size_t kidx = p.stem ? (size_t)(kh * 8 + kw) * p.cin_pad + ci      // stem: eight slots per kernel row
                                       : (size_t)(kh * c.k + kw) * p.cin_pad + ci;
            if (eb == 4) reinterpret_cast<float*>(row)[kidx] = v;        // f16x2: split in place below
            else reinterpret_cast<uint16_t*>(row)[kidx] = f32_to_bf16(v);
          }
    }
    // f16x2 weights.  A convolution in front of a BatchNorm is scale-free, so a checkpoint may hold weight rows of any
    // magnitude, and f16 pieces cannot (subnormal below 6e-5, infinite beyond 65504).  Every output channel's row is
    // therefore multiplied by the power of two 2^k that puts its largest |w| into [2^14, 2^15) -- exact -- and split
    // into P = f16(w 2^k) and Q = f16(w 2^k - P) (the difference is exact; Q is NOT scaled, unlike the low piece of
    // an activation): w 2^k = P + Q to 2^-23.  The kernel sums P.X0 + Q.X0 + (P 2^-11).X1 in ONE f32 chain (X0, X1 =
    // the activation's pieces, X1 carrying 2^11), and the channel's f32 BatchNorm scale takes the 2^-k below -- exact:
    // the epilogue's fma(acc, scale, shift) sees 2^k acc * 2^-k scale.  What remains is relative to the row: a weight
    // below 2^-15 of its row's largest loses low bits (an absolute error of 2^-39 of that largest weight).
    std::vector<int> row_exp(precision == NBC_PREC_F16X2 ? c.cout : 0, 0);
    if (precision == NBC_PREC_F16X2) {
      const size_t row_floats = row_bytes / 4;
      for (int o = 0; o < c.cout; ++o) {
        float* row = reinterpret_cast<float*>(base + p.w_off + (size_t)o * row_bytes);
        bool clamped = false;
        row_exp[o] = f16x2_row_exponent(row, row_floats, &clamped);
        if (clamped) flags |= NBC_PACK_ROW_CLAMPED;
        if (row_exp[o] != 0)
          for (size_t e = 0; e < row_floats; ++e) row[e] = std::ldexp(row[e], row_exp[o]);
      }
      // element e of a row in its f32-sized slot -> 32-element groups [P x 32][Q x 32] (stem: taps of 4)
      split_groups(reinterpret_cast<float*>(base + p.w_off), (size_t)c.cout * row_bytes / 4, p.stem ? 4 : 32, 1.0f);
    }
    // eval-mode BatchNorm as ATen applies it: alpha = gamma * invstd, beta = bias - mean * alpha
    const float* g = static_cast<const float*>(given[c.bn + ".weight"]->data);
    const float* b = static_cast<const float*>(given[c.bn + ".bias"]->data);
    const float* mu = static_cast<const float*>(given[c.bn + ".running_mean"]->data);
    const float* var = static_cast<const float*>(given[c.bn + ".running_var"]->data);
    float* scale = reinterpret_cast<float*>(base + p.scale_off);
    float* shift = reinterpret_cast<float*>(base + p.shift_off);
    for (int o = 0; o < c.cout; ++o) {
      const float invstd = 1.0f / std::sqrt(var[o] + kBnEps);
      const float alpha = g[o] * invstd;
      const float t = mu[o] * alpha;
      // f16x2: the row's power of two comes off again here (exact unless alpha * 2^-k leaves f32's normal range:
      // f16x2_row_exponent's clamp)
      // ... and the powers of the tensors this launch reads and writes (activation_exponents; all zero outside f16x2)
      scale[o] = ldexp_flagged(alpha, (row_exp.empty() ? 0 : -row_exp[o]) + out_exp[ui] - in_exp[ui], &flags);
      shift[o] = ldexp_flagged(b[o] - t, out_exp[ui], &flags);
    }
  }
  int32_t* meta = reinterpret_cast<int32_t*>(base + L.meta_off);
  meta[0] = kMetaMagic;
  meta[1] = flags;
  meta[2] = (int32_t)units.size();
  for (size_t ui = 0; ui < units.size(); ++ui) meta[kMetaExpBase + ui] = out_exp[ui];
  return NBC_OK;
}

int nbc_packed_weights_flags(const void* blob, size_t blob_bytes, int precision) {
  if (!known_precision(precision) || !blob) return set_error(NBC_ERR_INVALID, "nbc_packed_weights_flags: bad argument");
  const PackedLayout L = packed_layout(precision);
  if (blob_bytes < L.total_bytes) return set_error(NBC_ERR_INVALID, "nbc_packed_weights_flags: blob too small");
  const int32_t* meta = reinterpret_cast<const int32_t*>(static_cast<const unsigned char*>(blob) + L.meta_off);
  if (meta[0] != kMetaMagic) return set_error(NBC_ERR_INVALID, "nbc_packed_weights_flags: not a blob of nbc_pack_weights (this version)");
  return meta[1];
}

}  // extern "C"
