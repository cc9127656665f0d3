// Context, launch plan and the forward entry point of libnbc_hip.so.
//
// nbc_forward replaces, at the C ABI, `outputs = self.model(batch[0].to(self.device))` followed
// by `torch.argmax(outputs, dim=1)` (/root/reference/src/bark_calculator/models.py:269-270).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/nbc.h"
#include "nbc_internal.hpp"
#include "nbc_kernels.hpp"
#include "nbc_net.hpp"

using namespace nbc;

namespace {

#define NBC_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return set_error(NBC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

enum OpKind { OP_INGEST, OP_CONV, OP_MAXPOOL, OP_HEAD1X1, OP_UPSAMPLE };

struct Op {
  OpKind kind;
  int unit;            // conv unit index (OP_CONV / OP_HEAD1X1), -1 otherwise
  int in_buf, out_buf, res_buf;
  int Hi, Wi, Ci, Ho, Wo, Co;
  std::string name;
  double flops, bytes;
  int tile;            // OP_CONV: tile id of the LDS-DMA kernel (default choice or autotuned)
  int rows;            // OP_CONV: conv_rows_kind: 0 generic tiles, 1 / 2 the row-resident 3x3 kernel (tile 18 / 19 only)
};

struct Plan {
  int N = 0, H = 0, W = 0, precision = -1;
  bool keep = false;
  int h = 0, w = 0;                    // low-res logits size
  std::vector<Op> ops;
  std::vector<size_t> buf_bytes;       // per activation buffer
};

}  // namespace

struct nbc_ctx {
  int device = 0;
  int precision = -1;
  const unsigned char* weights = nullptr;   // device blob
  void* owned_weights = nullptr;
  PackedLayout layout;
  float mean[3] = {0.7399f, 0.6139f, 0.4401f};   // models.py:208
  float stdv[3] = {0.1068f, 0.1272f, 0.1271f};   // models.py:209
  Plan plan;                                // plan of the current (N,H,W)
  std::vector<Plan> plan_cache;             // plans (with their tuned tiles) of the other shapes seen, oldest first
  std::vector<void*> bufs;
  std::vector<size_t> buf_cap;
  float* lowres = nullptr;
  size_t lowres_cap = 0;
  int conv_tile = -1;                       // tile override, -1 = per-layer choice
  bool keep = false;
  bool profiling = false;
  // profiling: one event set (nops+1 events) per profiled forward, read back lazily so that the
  // timed loop never synchronises; nbc_num_op_records() averages over the sets and resets.
  std::vector<std::vector<hipEvent_t>> prof_sets;
  size_t prof_used = 0;
  std::vector<Op> prof_ops;
  std::vector<nbc_op_record> records;
  std::map<std::string, int> act_of;        // conv unit name -> op index (keep mode)
  void* scratch256 = nullptr;               // 256 bytes of device scratch (min/max of the preprocessor resize)
  int pack_flags = 0;                       // NBC_PACK_* of the attached blob (its trailer)
  std::vector<int> act_exp;                 // per conv unit: power of two its output tensor is stored with (trailer)
  unsigned* nonfinite = nullptr;            // one device word: bit 0 = a forward produced a NaN / infinite logit (sticky)
  void* zones_ws = nullptr;                 // remove_small_zones workspace: bg bytes, parent ints, size ints
  size_t zones_px = 0;                      // pixels it is sized for
};

namespace {

constexpr size_t kPlanCacheEntries = 64;

bool same_shape(const Plan& p, int N, int H, int W, int precision, bool keep) {
  return p.N == N && p.H == H && p.W == W && p.precision == precision && p.keep == keep;
}
bool same_plan(const Plan& p, const nbc_ctx* c, int N, int H, int W) {
  return same_shape(p, N, H, W, c->precision, c->keep);
}

// Park the current plan (folders of height-trimmed images alternate between a few shapes: each keeps
// its launch list and its measured tile choice instead of being rebuilt on every change).
void stash_plan(nbc_ctx* c) {
  Plan& cur = c->plan;
  if (cur.N == 0) return;
  for (Plan& p : c->plan_cache)
    if (same_shape(p, cur.N, cur.H, cur.W, cur.precision, cur.keep)) {
      p = cur; cur = Plan(); return;
    }
  if (c->plan_cache.size() >= kPlanCacheEntries) c->plan_cache.erase(c->plan_cache.begin());
  c->plan_cache.push_back(cur);
  cur = Plan();
}

// Build the launch list for an (N,H,W).  Activation buffers are recycled through a small pool
// unless `keep` asks for one buffer per op (layer-by-layer parity tests).
int build_plan(nbc_ctx* c, int N, int H, int W) {
  Plan P;
  P.N = N; P.H = H; P.W = W; P.precision = c->precision; P.keep = c->keep;
  const int eb = elem_bytes(c->precision);
  const auto& units = conv_units();
  const auto& L = c->layout;

  std::vector<bool> in_use;
  auto acquire = [&](size_t bytes) {
    if (!P.keep)
      for (size_t i = 0; i < in_use.size(); ++i)
        if (!in_use[i]) { in_use[i] = true; P.buf_bytes[i] = std::max(P.buf_bytes[i], bytes); return (int)i; }
    in_use.push_back(true);
    P.buf_bytes.push_back(bytes);
    return (int)in_use.size() - 1;
  };
  auto release = [&](int b) { if (b >= 0 && !P.keep) in_use[b] = false; };

  auto conv_out = [](int x, int k, int s, int p, int d) { return (x + 2 * p - d * (k - 1) - 1) / s + 1; };

  // ingest: image -> NHWC with one 16-byte pixel
  const int cin_img = kChunkBytes / eb;
  int cur = acquire((size_t)N * H * W * kChunkBytes);
  {
    Op o{};
    o.kind = OP_INGEST; o.unit = -1; o.in_buf = -1; o.out_buf = cur; o.res_buf = -1;
    o.Hi = H; o.Wi = W; o.Ci = 3; o.Ho = H; o.Wo = W; o.Co = cin_img; o.name = "ingest";
    P.ops.push_back(o);
  }
  int curH = H, curW = W, curC = cin_img;

  auto add_conv = [&](int ui, int in_buf, int inH, int inW, int inC, int res_buf, int* oH, int* oW) {
    const ConvUnit& u = units[ui];
    const int Ho = conv_out(inH, u.k, u.stride, u.pad, u.dil);
    const int Wo = conv_out(inW, u.k, u.stride, u.pad, u.dil);
    Op o{};
    o.kind = OP_CONV; o.unit = ui; o.in_buf = in_buf; o.res_buf = res_buf;
    o.Hi = inH; o.Wi = inW; o.Ci = inC; o.Ho = Ho; o.Wo = Wo; o.Co = u.cout; o.name = u.name;
    o.out_buf = acquire((size_t)N * Ho * Wo * u.cout * eb);
    o.rows = conv_rows_kind(c->precision, u.k, u.stride, u.pad, u.dil, inH, inW, Ho, Wo, inC, u.cout, res_buf >= 0);
    o.tile = choose_conv_tile(N * Ho * Wo, u.cout, u.cin * u.k * u.k, c->precision, o.rows);
    const double M = (double)N * Ho * Wo;
    o.flops = 2.0 * M * u.cout * u.cin * u.k * u.k;
    o.bytes = ((double)N * inH * inW * u.cin + (double)u.cout * u.cin * u.k * u.k + M * u.cout +
               (res_buf >= 0 ? M * u.cout : 0.0)) * eb;
    P.ops.push_back(o);
    *oH = Ho; *oW = Wo;
    return o.out_buf;
  };

  size_t ui = 0;
  // stem
  {
    int oH, oW;
    const int b = add_conv(0, cur, curH, curW, curC, -1, &oH, &oW);
    release(cur);
    cur = b; curH = oH; curW = oW; curC = units[0].cout;
    const int pH = (curH - 1) / 2 + 1, pW = (curW - 1) / 2 + 1;
    Op o{};
    o.kind = OP_MAXPOOL; o.unit = -1; o.in_buf = cur; o.res_buf = -1;
    o.Hi = curH; o.Wi = curW; o.Ci = curC; o.Ho = pH; o.Wo = pW; o.Co = curC; o.name = "backbone.maxpool";
    o.out_buf = acquire((size_t)N * pH * pW * curC * eb);
    o.bytes = ((double)N * curH * curW * curC + (double)N * pH * pW * curC) * eb;
    P.ops.push_back(o);
    release(cur);
    cur = o.out_buf; curH = pH; curW = pW;
    ui = 1;
  }
  // bottlenecks
  while (ui < units.size() && units[ui].block_first) {
    int h1, w1, h2, w2, h3, w3;
    const int t1 = add_conv((int)ui, cur, curH, curW, curC, -1, &h1, &w1);
    const int t2 = add_conv((int)ui + 1, t1, h1, w1, units[ui].cout, -1, &h2, &w2);
    release(t1);
    int idt = cur;
    size_t c3 = ui + 2;
    if (!units[ui + 2].residual) {      // downsample present
      int hd, wd;
      idt = add_conv((int)ui + 2, cur, curH, curW, curC, -1, &hd, &wd);
      release(cur);
      c3 = ui + 3;
    }
    const int out = add_conv((int)c3, t2, h2, w2, units[ui + 1].cout, idt, &h3, &w3);
    release(t2);
    release(idt);
    cur = out; curH = h3; curW = w3; curC = units[c3].cout;
    ui = c3 + 1;
  }
  // head
  {
    int oH, oW;
    const int t = add_conv((int)ui, cur, curH, curW, curC, -1, &oH, &oW);   // classifier.0
    release(cur);
    Op o{};
    o.kind = OP_HEAD1X1; o.unit = (int)ui + 1; o.in_buf = t; o.out_buf = -1; o.res_buf = -1;
    o.Hi = oH; o.Wi = oW; o.Ci = units[ui].cout; o.Ho = oH; o.Wo = oW; o.Co = kNumClasses;
    o.name = units[ui + 1].name;
    o.flops = 2.0 * N * oH * oW * o.Ci * kNumClasses;
    o.bytes = (double)N * oH * oW * o.Ci * eb + (double)N * oH * oW * kNumClasses * 4;
    P.ops.push_back(o);
    release(t);
    P.h = oH; P.w = oW;
    Op up{};
    up.kind = OP_UPSAMPLE; up.unit = -1; up.in_buf = -1; up.out_buf = -1; up.res_buf = -1;
    up.Hi = oH; up.Wi = oW; up.Ci = kNumClasses; up.Ho = H; up.Wo = W; up.Co = kNumClasses;
    up.name = "upsample_argmax";
    up.bytes = (double)N * oH * oW * kNumClasses * 4 + (double)N * H * W;
    P.ops.push_back(up);
  }
  (void)L;

  c->plan = P;
  return NBC_OK;
}

// Workspace of the current plan: buffers only ever grow, so a plan taken back from the cache finds
// them large enough unless a later, smaller-indexed plan never needed that slot.  A buffer that has to grow
// grows by at least half (hipFree synchronises the device: shapes that rise one after the other then reallocate
// O(log) times, not once per shape); a caller that knows its largest shape reserves it first (nbc_reserve).
int ensure_buffers(nbc_ctx* c) {
  const Plan& P = c->plan;
  if (c->bufs.size() < P.buf_bytes.size()) { c->bufs.resize(P.buf_bytes.size(), nullptr); c->buf_cap.resize(P.buf_bytes.size(), 0); }
  for (size_t i = 0; i < P.buf_bytes.size(); ++i) {
    if (c->buf_cap[i] < P.buf_bytes[i]) {
      const size_t grown = c->buf_cap[i] + c->buf_cap[i] / 2;
      if (c->bufs[i]) NBC_HIP(hipFree(c->bufs[i]));
      c->bufs[i] = nullptr; c->buf_cap[i] = 0;
      size_t want = std::max(P.buf_bytes[i], grown);
      hipError_t e = hipMalloc(&c->bufs[i], want);
      if (e != hipSuccess && want > P.buf_bytes[i]) {          // no room for the margin: the exact size
        (void)hipGetLastError();
        want = P.buf_bytes[i];
        e = hipMalloc(&c->bufs[i], want);
      }
      if (e != hipSuccess) return set_error(NBC_ERR_NOMEM, std::string("hipMalloc(workspace): ") + hipGetErrorString(e));
      c->buf_cap[i] = want;
    }
  }
  const size_t lr = (size_t)P.N * kNumClasses * P.h * P.w * sizeof(float);
  if (c->lowres_cap < lr) {
    if (c->lowres) NBC_HIP(hipFree(c->lowres));
    c->lowres = nullptr; c->lowres_cap = 0;
    NBC_HIP(hipMalloc((void**)&c->lowres, lr));
    c->lowres_cap = lr;
  }
  c->act_of.clear();
  for (size_t i = 0; i < P.ops.size(); ++i) c->act_of[P.ops[i].name] = (int)i;
  return NBC_OK;
}

// One convolution launch of the plan (shared by nbc_forward and nbc_autotune).
int launch_conv_op(nbc_ctx* c, const Op& o, int N, int tile, hipStream_t s, hipError_t* err) {
  const auto& units = conv_units();
  const ConvUnit& u = units[o.unit];
  const PackedConv& pc = c->layout.convs[o.unit];
  const int prec = c->precision;
  const size_t eb = elem_bytes(prec);
  ConvArgs a{};
  a.x = c->bufs[o.in_buf];
  a.w = c->weights + pc.w_off;
  a.scale = reinterpret_cast<const float*>(c->weights + pc.scale_off);
  a.shift = reinterpret_cast<const float*>(c->weights + pc.shift_off);
  a.res = o.res_buf >= 0 ? c->bufs[o.res_buf] : nullptr;
  a.y = c->bufs[o.out_buf];
  a.N = N; a.Hi = o.Hi; a.Wi = o.Wi; a.Ci = o.Ci;
  a.Ho = o.Ho; a.Wo = o.Wo; a.Co = o.Co;
  a.KH = u.k; a.KW = u.k; a.stride = u.stride; a.pad = u.pad; a.dil = u.dil;
  a.M = N * o.Ho * o.Wo;
  a.ksteps = pc.ksteps;
  a.relu = u.relu ? 1 : 0;
  a.stem = pc.stem ? 1 : 0;
  a.wo_shift = -1;
  a.hw_shift = -1;
  for (int sft = 0; sft < 31; ++sft) {
    if ((1 << sft) == o.Wo) a.wo_shift = sft;
    if ((1 << sft) == o.Ho * o.Wo) a.hw_shift = sft;
  }
  if (o.Ci != pc.cin_pad) return set_error(NBC_ERR_STATE, "plan/channel mismatch at " + o.name);
  const size_t xb = (size_t)N * o.Hi * o.Wi * o.Ci * eb;
  const size_t wbts = (size_t)o.Co * pc.ksteps * kKStepBytes;
  if (xb >= 0x80000000ull || wbts >= 0x80000000ull)
    return set_error(NBC_ERR_INVALID, "activation of " + o.name + " exceeds 2 GiB: lower the batch size");
  a.x_bytes = (unsigned)xb;
  a.w_bytes = (unsigned)wbts;
  *err = launch_conv_dma(a, prec, tile, s);
  return NBC_OK;
}

const char* kernel_name(OpKind k) {
  switch (k) {
    case OP_INGEST: return "ingest";
    case OP_CONV: return "conv_dma";
    case OP_MAXPOOL: return "maxpool";
    case OP_HEAD1X1: return "head1x1";
    default: return "upsample_argmax";
  }
}

}  // namespace

extern "C" {

int nbc_create(nbc_ctx** out, int hip_device) {
  if (!out) return set_error(NBC_ERR_INVALID, "nbc_create: null out");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return set_error(NBC_ERR_HIP, std::string("nbc_create: no HIP device (") + hipGetErrorString(e) + ")");
  if (hip_device < 0 || hip_device >= ndev) return set_error(NBC_ERR_INVALID, "nbc_create: bad device index");
  NBC_HIP(hipSetDevice(hip_device));
  hipDeviceProp_t prop;
  NBC_HIP(hipGetDeviceProperties(&prop, hip_device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(NBC_ERR_HIP, std::string("nbc_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  nbc_ctx* c = new nbc_ctx();
  c->device = hip_device;
  *out = c;
  return NBC_OK;
}

int nbc_destroy(nbc_ctx* c) {
  if (!c) return NBC_OK;
  (void)hipSetDevice(c->device);
  for (void* b : c->bufs) if (b) (void)hipFree(b);
  if (c->lowres) (void)hipFree(c->lowres);
  if (c->zones_ws) (void)hipFree(c->zones_ws);
  if (c->scratch256) (void)hipFree(c->scratch256);
  if (c->nonfinite) (void)hipFree(c->nonfinite);
  if (c->owned_weights) (void)hipFree(c->owned_weights);
  for (auto& set : c->prof_sets) for (hipEvent_t ev : set) (void)hipEventDestroy(ev);
  delete c;
  return NBC_OK;
}

int nbc_attach_weights(nbc_ctx* c, const void* dev_blob, size_t bytes, int precision) {
  if (!c || !dev_blob) return set_error(NBC_ERR_INVALID, "nbc_attach_weights: null argument");
  if (!known_precision(precision)) return set_error(NBC_ERR_INVALID, "nbc_attach_weights: unknown precision");
  PackedLayout L = packed_layout(precision);
  if (bytes < L.total_bytes) return set_error(NBC_ERR_INVALID, "nbc_attach_weights: blob smaller than the packed layout");
  if (reinterpret_cast<uintptr_t>(dev_blob) % 256 != 0)
    return set_error(NBC_ERR_INVALID, "nbc_attach_weights: blob must be 256-byte aligned");
  // the trailer: pack flags and the powers of two the activation tensors are stored with (nbc_net.hpp)
  int32_t meta[kMetaWords];
  NBC_HIP(hipSetDevice(c->device));
  NBC_HIP(hipMemcpy(meta, static_cast<const unsigned char*>(dev_blob) + L.meta_off, sizeof(meta), hipMemcpyDeviceToHost));
  const int nunits = (int)conv_units().size();
  if (meta[0] != kMetaMagic || meta[2] != nunits)
    return set_error(NBC_ERR_INVALID, "nbc_attach_weights: not a blob of this library's nbc_pack_weights (trailer mismatch)");
  if (c->owned_weights && c->owned_weights != dev_blob) { (void)hipFree(c->owned_weights); c->owned_weights = nullptr; }
  c->pack_flags = meta[1];
  c->act_exp.assign(meta + kMetaExpBase, meta + kMetaExpBase + nunits);
  c->weights = static_cast<const unsigned char*>(dev_blob);
  c->layout = L;
  if (c->precision != precision) stash_plan(c);      // element size changed: another plan
  c->precision = precision;
  return NBC_OK;
}

int nbc_load_weights(nbc_ctx* c, const nbc_tensor* tensors, int n, int precision) {
  if (!c) return set_error(NBC_ERR_INVALID, "nbc_load_weights: null context");
  const size_t bytes = nbc_packed_weights_bytes(precision);
  if (bytes == 0) return set_error(NBC_ERR_INVALID, "nbc_load_weights: unknown precision");
  std::vector<unsigned char> host(bytes);
  int rc = nbc_pack_weights(tensors, n, precision, host.data(), bytes);
  if (rc != NBC_OK) return rc;
  NBC_HIP(hipSetDevice(c->device));
  void* dev = nullptr;
  NBC_HIP(hipMalloc(&dev, bytes));
  hipError_t e = hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(dev); return set_error(NBC_ERR_HIP, std::string("hipMemcpy(weights): ") + hipGetErrorString(e)); }
  if (c->owned_weights) (void)hipFree(c->owned_weights);
  c->owned_weights = nullptr;
  rc = nbc_attach_weights(c, dev, bytes, precision);
  if (rc != NBC_OK) { (void)hipFree(dev); return rc; }
  c->owned_weights = dev;
  return NBC_OK;
}

// RCCL is resolved at call time from the host process (the library itself links libamdhip64 only):
// whatever librccl the host has loaded -- torch's own, or /opt/rocm's -- is the one whose communicator
// the caller hands in.
namespace {
typedef int (*nccl_bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_rank_fn)(void*, int*);
void* rccl_symbol(const char* name) {
  void* f = dlsym(RTLD_DEFAULT, name);
  if (f) return f;
  static void* handle = [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return h ? h : dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  }();
  return handle ? dlsym(handle, name) : nullptr;
}
}  // namespace

int nbc_bcast_weights(nbc_ctx* c, void* rccl_comm, int root, int precision, void* hip_stream) {
  if (!c || !rccl_comm) return set_error(NBC_ERR_INVALID, "nbc_bcast_weights: null argument");
  const size_t bytes = nbc_packed_weights_bytes(precision);
  if (bytes == 0) return set_error(NBC_ERR_INVALID, "nbc_bcast_weights: unknown precision");
  auto bcast = reinterpret_cast<nccl_bcast_fn>(rccl_symbol("ncclBroadcast"));
  auto user_rank = reinterpret_cast<nccl_rank_fn>(rccl_symbol("ncclCommUserRank"));
  if (!bcast || !user_rank) return set_error(NBC_ERR_STATE, "nbc_bcast_weights: no RCCL (librccl.so) in this process");
  int rank = -1;
  if (user_rank(rccl_comm, &rank) != 0) return set_error(NBC_ERR_INVALID, "nbc_bcast_weights: ncclCommUserRank failed (bad communicator?)");
  NBC_HIP(hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  void* blob = nullptr;
  bool fresh = false;
  if (rank == root) {
    if (!c->weights || c->precision != precision)
      return set_error(NBC_ERR_STATE, "nbc_bcast_weights: the root rank has no weights of this precision attached");
    blob = const_cast<unsigned char*>(c->weights);
  } else {
    NBC_HIP(hipMalloc(&blob, bytes));
    fresh = true;
  }
  const int rc = bcast(blob, blob, bytes, /*ncclUint8*/ 1, root, rccl_comm, s);
  if (rc != 0) {
    if (fresh) (void)hipFree(blob);
    return set_error(NBC_ERR_HIP, "nbc_bcast_weights: ncclBroadcast returned " + std::to_string(rc));
  }
  if (fresh) {
    if (c->owned_weights) (void)hipFree(c->owned_weights);     // synchronises: nothing of the old blob is in flight
    c->owned_weights = nullptr;
    const int arc = nbc_attach_weights(c, blob, bytes, precision);
    if (arc != NBC_OK) { (void)hipFree(blob); return arc; }
    c->owned_weights = blob;
  }
  return NBC_OK;
}

int nbc_weights_flags(nbc_ctx* c) {
  if (!c) return set_error(NBC_ERR_INVALID, "null context");
  if (!c->weights) return set_error(NBC_ERR_STATE, "nbc_weights_flags: no weights attached");
  return c->pack_flags;
}

namespace {
// power of two the output of plan op `name` is stored with: a conv unit's own, the max-pool keeps the stem's, the image has none
bool stored_exponent(const nbc_ctx* c, const std::string& name, int* e) {
  const auto& units = conv_units();
  if (name == "ingest") { *e = 0; return true; }
  if (name == "backbone.maxpool") { *e = c->act_exp.empty() ? 0 : c->act_exp[0]; return true; }
  for (size_t u = 0; u < units.size(); ++u)
    if (units[u].name == name) { *e = u < c->act_exp.size() ? c->act_exp[u] : 0; return true; }
  return false;
}
}  // namespace

int nbc_activation_exponent(nbc_ctx* c, const char* name, int32_t* exponent) {
  if (!c || !name || !exponent) return set_error(NBC_ERR_INVALID, "nbc_activation_exponent: null argument");
  if (!c->weights) return set_error(NBC_ERR_STATE, "nbc_activation_exponent: no weights attached");
  int e = 0;
  if (!stored_exponent(c, name, &e)) return set_error(NBC_ERR_INVALID, std::string("nbc_activation_exponent: unknown op ") + name);
  *exponent = e;
  return NBC_OK;
}

int nbc_set_normalization(nbc_ctx* c, const float mean[3], const float stdv[3]) {
  if (!c || !mean || !stdv) return set_error(NBC_ERR_INVALID, "nbc_set_normalization: null argument");
  for (int i = 0; i < 3; ++i) { c->mean[i] = mean[i]; c->stdv[i] = stdv[i]; }
  return NBC_OK;
}

int nbc_set_conv_tile(nbc_ctx* c, int tile) {
  if (!c) return set_error(NBC_ERR_INVALID, "null context");
  if (tile < -1 || tile >= CONV_TILE_COUNT) return set_error(NBC_ERR_INVALID, "nbc_set_conv_tile: bad tile id");
  c->conv_tile = tile;
  return NBC_OK;
}

int nbc_set_keep_activations(nbc_ctx* c, int on) {
  if (!c) return set_error(NBC_ERR_INVALID, "null context");
  if (c->keep != (on != 0)) stash_plan(c);
  c->keep = on != 0;
  return NBC_OK;
}

int nbc_nonfinite_seen(nbc_ctx* c, int reset) {
  if (!c) return set_error(NBC_ERR_INVALID, "null context");
  if (!c->nonfinite) return 0;                       // no forward yet
  NBC_HIP(hipSetDevice(c->device));
  unsigned v = 0;
  NBC_HIP(hipDeviceSynchronize());                   // every forward on every stream (non-blocking ones too) has finished
  NBC_HIP(hipMemcpy(&v, c->nonfinite, sizeof(v), hipMemcpyDeviceToHost));
  if (reset && v) NBC_HIP(hipMemset(c->nonfinite, 0, sizeof(v)));
  return v ? 1 : 0;
}

int nbc_nonfinite_peek_async(nbc_ctx* c, uint32_t* host_dst, void* hip_stream) {
  if (!c || !host_dst) return set_error(NBC_ERR_INVALID, "nbc_nonfinite_peek_async: null argument");
  NBC_HIP(hipSetDevice(c->device));
  {
    // pinned memory only: a copy to pageable memory is staged and may block (the contract is "no synchronisation")
    hipPointerAttribute_t at{};
    const hipError_t pe = hipPointerGetAttributes(&at, host_dst);
    if (pe != hipSuccess || at.type != hipMemoryTypeHost) {
      (void)hipGetLastError();
      return set_error(NBC_ERR_INVALID, "nbc_nonfinite_peek_async: host_dst must be pinned host memory (hipHostMalloc / hipHostRegister)");
    }
  }
  if (!c->nonfinite) { *host_dst = 0; return NBC_OK; }   // no forward yet
  NBC_HIP(hipMemcpyAsync(host_dst, c->nonfinite, sizeof(uint32_t), hipMemcpyDeviceToHost, static_cast<hipStream_t>(hip_stream)));
  return NBC_OK;
}

int nbc_set_profiling(nbc_ctx* c, int on) {
  if (!c) return set_error(NBC_ERR_INVALID, "null context");
  c->profiling = on != 0;
  return NBC_OK;
}

int nbc_reserve(nbc_ctx* c, int N, int H, int W) {
  if (!c) return set_error(NBC_ERR_INVALID, "null context");
  if (c->precision < 0) return set_error(NBC_ERR_STATE, "nbc_reserve: no weights attached");
  if (N < 1 || H < 8 || W < 8) return set_error(NBC_ERR_INVALID, "nbc_reserve: need N>=1, H>=8, W>=8");
  NBC_HIP(hipSetDevice(c->device));
  if (same_plan(c->plan, c, N, H, W)) return NBC_OK;
  stash_plan(c);
  bool found = false;
  for (const Plan& p : c->plan_cache)
    if (same_plan(p, c, N, H, W)) { c->plan = p; found = true; break; }
  if (!found) {
    int rc = build_plan(c, N, H, W);
    if (rc != NBC_OK) return rc;
  }
  int rc = ensure_buffers(c);
  if (rc != NBC_OK) c->plan = Plan();
  return rc;
}

int nbc_forward(nbc_ctx* c, const void* x_dev, int x_dtype, int N, int H, int W,
                float* logits_lowres_dev, float* logits_full_dev, void* labels_dev, int labels_dtype,
                int64_t* counts_dev, int exclude_nodes, void* hip_stream) {
  if (!c || !x_dev) return set_error(NBC_ERR_INVALID, "nbc_forward: null argument");
  if (!c->weights) return set_error(NBC_ERR_STATE, "nbc_forward: no weights attached (load_state_dict first)");
  if (x_dtype != NBC_IN_F32_NCHW && x_dtype != NBC_IN_U8_NHWC) return set_error(NBC_ERR_INVALID, "nbc_forward: bad x_dtype");
  if (labels_dtype != NBC_LABEL_U8 && labels_dtype != NBC_LABEL_I64) return set_error(NBC_ERR_INVALID, "nbc_forward: bad labels_dtype");
  int rc = nbc_reserve(c, N, H, W);
  if (rc != NBC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const Plan& P = c->plan;
  const int prec = c->precision;
  if (!c->nonfinite) {
    NBC_HIP(hipMalloc((void**)&c->nonfinite, 256));
    NBC_HIP(hipMemset(c->nonfinite, 0, 256));
  }

  const size_t nops = P.ops.size();
  const size_t nl = nops;                              // one launch per op
  constexpr size_t kMaxProfSets = 4096;
  std::vector<hipEvent_t>* evs = nullptr;
  if (c->profiling && c->prof_used < kMaxProfSets) {
    if (c->prof_used > 0 && c->prof_ops.size() != nops) c->prof_used = 0;   // plan changed: restart
    if (c->prof_sets.size() <= c->prof_used) c->prof_sets.emplace_back();
    evs = &c->prof_sets[c->prof_used];
    while (evs->size() < nl + 1) {
      hipEvent_t ev;
      NBC_HIP(hipEventCreate(&ev));
      evs->push_back(ev);
    }
    c->prof_ops = P.ops;
  }
  float* lowres = logits_lowres_dev ? logits_lowres_dev : c->lowres;

  if (evs) NBC_HIP(hipEventRecord((*evs)[0], s));
  for (size_t l = 0; l < nl; ++l) {
    const Op& o = P.ops[l];
    hipError_t e = hipSuccess;
    int rc = NBC_OK;
    switch (o.kind) {
      case OP_INGEST:
        if (x_dtype == NBC_IN_F32_NCHW)
          e = launch_ingest_f32(static_cast<const float*>(x_dev), c->bufs[o.out_buf], N, H, W, prec, s);
        else
          e = launch_ingest_u8(static_cast<const uint8_t*>(x_dev), c->bufs[o.out_buf], N, H, W, c->mean, c->stdv, prec, s);
        break;
      case OP_CONV: {
        int tile = c->conv_tile;
        if (!conv_tile_ok(prec, tile, o.Co, o.rows)) tile = o.tile;   // no override, or it does not fit (a generic tile on a
                                                                      // layer of the row-resident kernel, or the reverse): planned tile
        rc = launch_conv_op(c, o, N, tile, s, &e);
        if (rc != NBC_OK) return rc;
        break;
      }
      case OP_MAXPOOL:
        e = launch_maxpool3x3s2(c->bufs[o.in_buf], c->bufs[o.out_buf], N, o.Hi, o.Wi, o.Ci, o.Ho, o.Wo, prec, s);
        break;
      case OP_HEAD1X1: {
        const PackedConv& pc = c->layout.convs[o.unit];
        if (o.Ci != 512) return set_error(NBC_ERR_STATE, "classifier.4 expects 512 input channels");
        // also clears this launch's share of the counters (3 per image) when the batch has at most 256 of them
        unsigned long long* cz = counts_dev && 3 * N <= 256 ? reinterpret_cast<unsigned long long*>(counts_dev) : nullptr;
        e = launch_head1x1(c->bufs[o.in_buf], reinterpret_cast<const float*>(c->weights + pc.w_off),
                           reinterpret_cast<const float*>(c->weights + pc.shift_off),
                           lowres, N, o.Ho * o.Wo, prec, cz, c->nonfinite, s);
        break;
      }
      case OP_UPSAMPLE:
        if (counts_dev && 3 * N > 256) {             // more counters than classifier.4's launch clears
          e = hipMemsetAsync(counts_dev, 0, sizeof(int64_t) * 3 * N, s);
          if (e != hipSuccess) break;
        }
        if (logits_full_dev || labels_dev || counts_dev)
          e = launch_upsample_argmax(lowres, N, P.h, P.w, H, W, logits_full_dev, labels_dev,
                                     labels_dtype == NBC_LABEL_I64 ? 1 : 0,
                                     reinterpret_cast<unsigned long long*>(counts_dev), exclude_nodes, s);
        break;
    }
    if (e != hipSuccess)
      return set_error(NBC_ERR_HIP, "launch of " + o.name + " failed: " + hipGetErrorString(e));
    if (evs) NBC_HIP(hipEventRecord((*evs)[l + 1], s));
  }
  if (evs) ++c->prof_used;
  return NBC_OK;
}

}  // extern "C"

// Average the event sets recorded since the last call (synchronises on the last one), then reset.
static int collect_profile(nbc_ctx* c) {
  if (c->prof_used == 0) return NBC_OK;
  const auto& units = conv_units();
  const size_t nops = c->prof_ops.size();
  NBC_HIP(hipEventSynchronize(c->prof_sets[c->prof_used - 1][nops]));
  c->records.assign(nops, nbc_op_record{});
  std::vector<double> sum(nops, 0.0);
  for (size_t i = 0; i < nops; ++i)
    for (size_t k = 0; k < c->prof_used; ++k) {
      float ms = 0.f;
      NBC_HIP(hipEventElapsedTime(&ms, c->prof_sets[k][i], c->prof_sets[k][i + 1]));
      sum[i] += ms;
    }
  for (size_t i = 0; i < nops; ++i) {
    const Op& o = c->prof_ops[i];
    nbc_op_record& r = c->records[i];
    std::snprintf(r.name, sizeof(r.name), "%s", o.name.c_str());
    std::snprintf(r.kernel, sizeof(r.kernel), "%s", kernel_name(o.kind));
    r.ms = (float)(sum[i] / (double)c->prof_used);
    r.calls = (int32_t)c->prof_used;
    r.launches = 1;
    r.flops = o.flops;
    r.bytes = o.bytes;
    r.kh = r.kw = (o.kind == OP_CONV || o.kind == OP_HEAD1X1) ? units[o.unit].k : 0;
    r.cout = o.Co;
  }
  c->prof_used = 0;
  return NBC_OK;
}

extern "C" {

int nbc_autotune(nbc_ctx* c, const void* x_dev, int x_dtype, int N, int H, int W, int reps, int objective,
                 void* hip_stream) {
  if (!c || !x_dev) return set_error(NBC_ERR_INVALID, "nbc_autotune: null argument");
  if (reps < 1) reps = 3;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  // one real forward first: every activation buffer then holds realistic data for the timed launches
  int rc = nbc_forward(c, x_dev, x_dtype, N, H, W, nullptr, nullptr, nullptr, NBC_LABEL_U8, nullptr, 0, hip_stream);
  if (rc != NBC_OK) return rc;
  NBC_HIP(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  NBC_HIP(hipEventCreate(&e0));
  NBC_HIP(hipEventCreate(&e1));
  Plan& P = c->plan;
  for (size_t oi = 0; oi < P.ops.size(); ++oi) {
    Op& o = P.ops[oi];
    if (o.kind != OP_CONV) continue;
    const int NB = N;
    float best_ms = 1e30f;
    int best = o.tile;
    for (int tile = 0; tile < CONV_TILE_COUNT; ++tile) {
      if (!conv_tile_ok(c->precision, tile, o.Co, o.rows)) continue;
      hipError_t e = hipSuccess;
      rc = launch_conv_op(c, o, NB, tile, s, &e);                         // warm-up (and attribute set-up)
      if (rc != NBC_OK || e != hipSuccess) continue;
      (void)hipEventRecord(e0, s);
      for (int k = 0; k < reps; ++k) (void)launch_conv_op(c, o, NB, tile, s, &e);
      (void)hipEventRecord(e1, s);
      if (hipEventSynchronize(e1) != hipSuccess) continue;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
      if (objective == 1) {
        // throughput objective: several forwards run concurrently on other streams, so a launch that
        // fills only part of the chip costs only that part: weigh the time by the fraction of CUs used
        const int tiles = ((NB * o.Ho * o.Wo + conv_tile_rows(tile) - 1) / conv_tile_rows(tile)) * (o.Co / conv_tile_cols(tile));
        if (tiles < 256) ms *= (float)tiles / 256.0f;
      }
      if (ms < best_ms) { best_ms = ms; best = tile; }
    }
    o.tile = best;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return NBC_OK;
}

int nbc_get_plan_tiles(nbc_ctx* c, int32_t* tiles, int capacity) {
  if (!c || !tiles) return set_error(NBC_ERR_INVALID, "nbc_get_plan_tiles: null argument");
  int n = 0;
  for (const Op& o : c->plan.ops)
    if (o.kind == OP_CONV) { if (n < capacity) tiles[n] = o.tile; ++n; }
  return n;
}

int nbc_default_conv_tile(int M, int Cout, int K, int precision) {
  if (M < 1 || Cout < 1 || K < 1) return -1;
  return choose_conv_tile(M, Cout, K, precision, 0);
}

int nbc_set_plan_tiles(nbc_ctx* c, const int32_t* tiles, int n) {
  if (!c || !tiles) return set_error(NBC_ERR_INVALID, "nbc_set_plan_tiles: null argument");
  int convs = 0;
  for (const Op& o : c->plan.ops) convs += o.kind == OP_CONV;
  if (convs == 0) return set_error(NBC_ERR_STATE, "nbc_set_plan_tiles: no plan (nbc_reserve first)");
  if (n != convs) return set_error(NBC_ERR_INVALID, "nbc_set_plan_tiles: the plan has " + std::to_string(convs) + " convolutions");
  int k = 0;
  for (const Op& o : c->plan.ops) {
    if (o.kind != OP_CONV) continue;
    const int t = tiles[k++];
    if (!conv_tile_ok(c->precision, t, o.Co, o.rows))
      return set_error(NBC_ERR_INVALID, "nbc_set_plan_tiles: tile " + std::to_string(t) + " does not fit " + o.name);
  }
  k = 0;
  for (Op& o : c->plan.ops)
    if (o.kind == OP_CONV) o.tile = tiles[k++];
  return NBC_OK;
}

int nbc_upsample_argmax(nbc_ctx* c, const float* lowres, int N, int h, int w, int H, int W,
                        float* logits_full_dev, void* labels_dev, int labels_dtype,
                        int64_t* counts_dev, int exclude_nodes, void* hip_stream) {
  if (!c || !lowres) return set_error(NBC_ERR_INVALID, "nbc_upsample_argmax: null argument");
  if (N < 1 || h < 1 || w < 1 || H < 1 || W < 1) return set_error(NBC_ERR_INVALID, "nbc_upsample_argmax: bad shape");
  if (labels_dtype != NBC_LABEL_U8 && labels_dtype != NBC_LABEL_I64) return set_error(NBC_ERR_INVALID, "nbc_upsample_argmax: bad labels_dtype");
  NBC_HIP(hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  if (counts_dev) NBC_HIP(hipMemsetAsync(counts_dev, 0, sizeof(int64_t) * 3 * N, s));
  NBC_HIP(launch_upsample_argmax(lowres, N, h, w, H, W, logits_full_dev, labels_dev,
                                 labels_dtype == NBC_LABEL_I64 ? 1 : 0,
                                 reinterpret_cast<unsigned long long*>(counts_dev), exclude_nodes, s));
  return NBC_OK;
}

int nbc_remove_small_zones(nbc_ctx* c, void* labels_dev, int labels_dtype, int N, int H, int W, int min_pixels,
                           int exclude_nodes, int64_t* counts_dev, void* hip_stream) {
  if (!c || !labels_dev) return set_error(NBC_ERR_INVALID, "nbc_remove_small_zones: null argument");
  if (N < 1 || N > 65535 || H < 1 || W < 1 || min_pixels < 0) return set_error(NBC_ERR_INVALID, "nbc_remove_small_zones: bad shape");
  if (labels_dtype != NBC_LABEL_U8 && labels_dtype != NBC_LABEL_I64) return set_error(NBC_ERR_INVALID, "nbc_remove_small_zones: bad labels_dtype");
  NBC_HIP(hipSetDevice(c->device));
  const size_t px = (size_t)N * H * W;
  if (px > c->zones_px) {                             // grows by at least half, like the activation buffers
    size_t cap = std::max(px, c->zones_px + c->zones_px / 2);
    if (c->zones_ws) { NBC_HIP(hipFree(c->zones_ws)); c->zones_ws = nullptr; c->zones_px = 0; }
    if (hipMalloc(&c->zones_ws, cap * 9 + 256) != hipSuccess) {
      (void)hipGetLastError();
      cap = px;
      NBC_HIP(hipMalloc(&c->zones_ws, cap * 9 + 256));
    }
    c->zones_px = cap;
  }
  int* parent = static_cast<int*>(c->zones_ws);
  int* size = parent + px;
  unsigned char* bg = reinterpret_cast<unsigned char*>(size + px);
  NBC_HIP(launch_remove_small_zones(labels_dev, labels_dtype == NBC_LABEL_I64 ? 1 : 0, N, H, W, min_pixels, exclude_nodes, bg,
                                    parent, size, reinterpret_cast<unsigned long long*>(counts_dev),
                                    static_cast<hipStream_t>(hip_stream)));
  return NBC_OK;
}

int nbc_resize_cubic_u8(nbc_ctx* c, const uint8_t* src_dev, int H, int W, float* dst_dev, int out_h, int out_w,
                        void* hip_stream) {
  if (!c || !src_dev || !dst_dev) return set_error(NBC_ERR_INVALID, "nbc_resize_cubic_u8: null argument");
  if (H < 1 || W < 1 || out_h < 1 || out_w < 1) return set_error(NBC_ERR_INVALID, "nbc_resize_cubic_u8: bad shape");
  NBC_HIP(hipSetDevice(c->device));
  if (!c->scratch256) NBC_HIP(hipMalloc(&c->scratch256, 256));
  unsigned* minmax = static_cast<unsigned*>(c->scratch256);
  NBC_HIP(launch_resize_cubic_u8(src_dev, H, W, dst_dev, nullptr, nullptr, out_h, out_w, minmax, static_cast<hipStream_t>(hip_stream)));
  return NBC_OK;
}

int nbc_preprocess_u8(nbc_ctx* c, const uint8_t* src_dev, int H, int W, uint8_t* dst_u8_dev, int32_t* row_lit_dev, int out_h,
                      int out_w, void* hip_stream) {
  if (!c || !src_dev || !dst_u8_dev) return set_error(NBC_ERR_INVALID, "nbc_preprocess_u8: null argument");
  if (H < 1 || W < 1 || out_h < 1 || out_w < 1) return set_error(NBC_ERR_INVALID, "nbc_preprocess_u8: bad shape");
  NBC_HIP(hipSetDevice(c->device));
  if (!c->scratch256) NBC_HIP(hipMalloc(&c->scratch256, 256));
  unsigned* minmax = static_cast<unsigned*>(c->scratch256);
  NBC_HIP(launch_resize_cubic_u8(src_dev, H, W, nullptr, dst_u8_dev, row_lit_dev, out_h, out_w, minmax,
                                 static_cast<hipStream_t>(hip_stream)));
  return NBC_OK;
}

int nbc_num_op_records(nbc_ctx* c) {
  if (!c) return 0;
  if (collect_profile(c) != NBC_OK) return NBC_ERR_HIP;
  return (int)c->records.size();
}

int nbc_get_op_record(nbc_ctx* c, int index, nbc_op_record* out) {
  if (!c || !out || index < 0 || index >= (int)c->records.size())
    return set_error(NBC_ERR_INVALID, "nbc_get_op_record: bad index");
  *out = c->records[index];
  return NBC_OK;
}

int nbc_activation_peaks(nbc_ctx* c, float* peaks_host, int capacity) {
  if (!c || !peaks_host) return set_error(NBC_ERR_INVALID, "nbc_activation_peaks: null argument");
  if (!c->plan.keep) return set_error(NBC_ERR_STATE, "nbc_activation_peaks: keep-activations is off (nbc_set_keep_activations, then a forward)");
  const int nunits = (int)conv_units().size();
  if (capacity < nunits) return set_error(NBC_ERR_INVALID, "nbc_activation_peaks: need room for nbc_num_convs() values");
  NBC_HIP(hipSetDevice(c->device));
  unsigned* dev = nullptr;
  NBC_HIP(hipMalloc((void**)&dev, sizeof(unsigned) * nunits));
  hipError_t e = hipMemset(dev, 0, sizeof(unsigned) * nunits);
  const int N = c->plan.N;
  for (const Op& o : c->plan.ops) {
    if (e != hipSuccess) break;
    if (o.kind != OP_CONV || o.out_buf < 0) continue;
    e = launch_absmax(c->bufs[o.out_buf], (size_t)N * o.Ho * o.Wo * o.Co, o.Co, c->precision, dev + o.unit, nullptr);
  }
  std::vector<unsigned> bits(nunits, 0u);
  if (e == hipSuccess) e = hipMemcpy(bits.data(), dev, sizeof(unsigned) * nunits, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) return set_error(NBC_ERR_HIP, std::string("nbc_activation_peaks: ") + hipGetErrorString(e));
  for (int u = 0; u < nunits; ++u) std::memcpy(&peaks_host[u], &bits[u], 4);
  return nunits;
}

int nbc_read_activation(nbc_ctx* c, const char* name, float* dst_host, size_t capacity, int64_t shape[4]) {
  if (!c || !name || !dst_host) return set_error(NBC_ERR_INVALID, "nbc_read_activation: null argument");
  if (!c->plan.keep) return set_error(NBC_ERR_STATE, "nbc_read_activation: keep-activations is off");
  auto it = c->act_of.find(name);
  if (it == c->act_of.end()) return set_error(NBC_ERR_INVALID, std::string("nbc_read_activation: unknown op ") + name);
  const Op& o = c->plan.ops[it->second];
  if (o.out_buf < 0) return set_error(NBC_ERR_INVALID, "nbc_read_activation: op has no activation buffer");
  const int N = c->plan.N;
  const size_t elems = (size_t)N * o.Ho * o.Wo * o.Co;
  if (capacity < elems) return set_error(NBC_ERR_INVALID, "nbc_read_activation: destination too small");
  NBC_HIP(hipSetDevice(c->device));
  float* tmp = nullptr;
  NBC_HIP(hipMalloc((void**)&tmp, elems * sizeof(float)));
  hipError_t e = launch_nhwc_to_nchw_f32(c->bufs[o.out_buf], tmp, N, o.Ho, o.Wo, o.Co, c->precision, nullptr);
  if (e == hipSuccess) e = hipMemcpy(dst_host, tmp, elems * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (e != hipSuccess) return set_error(NBC_ERR_HIP, std::string("nbc_read_activation: ") + hipGetErrorString(e));
  int a = 0;
  if (stored_exponent(c, o.name, &a) && a != 0)        // f16x2: the tensor as the network defines it (power of two taken off, exact)
    for (size_t i = 0; i < elems; ++i) dst_host[i] = std::ldexp(dst_host[i], -a);
  if (shape) { shape[0] = N; shape[1] = o.Co; shape[2] = o.Ho; shape[3] = o.Wo; }
  return NBC_OK;
}

}  // extern "C"
