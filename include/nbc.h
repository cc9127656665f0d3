/*
 * nbc.h -- C ABI of the MI355X-native FCN-ResNet-50 segmentation path ("nbc" = neural bark
 * calculator).  One shared library, libnbc_hip.so, plain pointers and sizes only; no torch
 * types.  All file:line citations are into /root/reference/src/bark_calculator/.
 *
 * What this boundary replaces in the reference:
 *   - the object bound to `self.model`            models.py:221   fcn_resnet50(pretrained=False)
 *   - `self.model.load_state_dict(...)`           models.py:222   -> nbc_pack_weights / nbc_load_weights
 *   - `self.model.to(device)`                     models.py:223   -> nbc_create(device) + nbc_attach_weights
 *   - `outputs = self.model(batch[0].to(device))` models.py:269   -> nbc_forward (logits_full_dev)
 *   - `outputs = torch.argmax(outputs, dim=1)`    models.py:270   -> nbc_forward (labels_dev)
 *   - the `--exclude_nodes` remap 2 -> 1          models.py:273-276 -> nbc_forward (exclude_nodes)
 *   - the per-class pixel counting                models.py:324-331 -> nbc_forward (counts_dev)
 *   - ToTensor + Normalize of the loaded image    dataset.py:175-186, models.py:233-237
 *                                                 -> nbc_forward with NBC_IN_U8_NHWC
 *
 * Conventions: every function returns 0 (NBC_OK) or a negative error code; the message of the
 * last error on the calling thread is nbc_last_error().  Nothing throws across the ABI.  A
 * context is bound to one HIP device and is not thread-safe (the reference loop is
 * single-threaded, models.py:257-270).  The caller owns every input/output buffer; the context
 * owns its activation workspace (sized on the first call for an (N,H,W), then reused: no
 * allocation in steady state).  Work is enqueued on the caller's stream; outputs are valid
 * once that stream has been synchronised.
 *
 * Eval-mode only: BatchNorm uses running statistics, Dropout is the identity.
 */
#ifndef NBC_H
#define NBC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbc_ctx nbc_ctx;

enum {
  NBC_OK = 0,
  NBC_ERR_INVALID = -1,   /* bad argument / shape */
  NBC_ERR_KEYS = -2,      /* state_dict keys or shapes do not match (models.py:222 would raise) */
  NBC_ERR_HIP = -3,       /* HIP runtime error */
  NBC_ERR_STATE = -4,     /* call order (e.g. forward before weights) */
  NBC_ERR_NOMEM = -5
};

/* Arithmetic of the convolution stack. */
enum {
  NBC_PREC_FP32 = 0,  /* f32 activations/weights, v_mfma_f32_32x32x2_f32: the parity mode */
  NBC_PREC_BF16 = 1,  /* bf16 activations/weights, f32 accumulate + f32 BN epilogue: throughput mode */
  NBC_PREC_F16X2 = 2  /* f32-grade on the 16-bit matrix pipe: every f32 value is kept as two f16 pieces, 4 bytes per element
                         like f32.
                         Activations: X0 = f16(x), X1 = f16((x - X0) * 2^11); x = X0 + X1 * 2^-11 to 2^-23 relative at
                         worst (rms 4e-8: the f32 level) for |x| >= 2^-12 = 2.4e-4; below that X1 is an f16 subnormal
                         and the error is ABSOLUTE, at most 2^-36 (1.5e-11); beyond +-65504 (f16's range) a value turns
                         into NaN, never into a silently wrong number (nbc_nonfinite_seen).  A tensor between a
                         BatchNorm and the next convolution is scale-free, so a checkpoint may hold activations of any
                         magnitude: nbc_pack_weights therefore estimates every tensor's size from its producing
                         BatchNorm (max over channels of |beta| + 3 |gamma| sqrt(var / (var + eps))) and stores a tensor whose estimate lies
                         outside [2^-5, 2^7] times the power of two that brings it to [2, 4) -- folded into the
                         producing launch's f32 (scale, shift) and, inverted, into the scale of every launch that reads
                         it (one power per residual stream); exact, ReLU and max-pool being positively homogeneous, and
                         0 for every tensor of an ordinary checkpoint.  The 2^-36 floor is then 2^-37 of the tensor's
                         expected size or less, whatever that size was in the checkpoint; what remains is a tensor
                         whose running statistics do not describe its data (an untrained BatchNorm behind weights
                         of another scale), where the floor can reach the values again silently and overflow raises the
                         flag.  nbc_read_activation returns the tensor as the network defines it (power taken off).
                         Weights: nbc_pack_weights multiplies every output channel's row by the power of two that puts
                         its largest |w| into [2^14, 2^15) -- exact -- splits it into P = f16(w 2^k), Q = f16(w 2^k - P)
                         (to 2^-23 of the row's largest weight; a weight below 2^-15 of it loses low bits, an absolute
                         error of 2^-39 of the largest) and folds 2^-k into the channel's f32 BatchNorm scale -- exact.
                         The magnitude of a checkpoint's weights therefore does not matter (a convolution in front of a
                         BatchNorm is scale-free): 1e-6 or 1e6 pack to the same pieces.  Limits, reported and not
                         silent (nbc_packed_weights_flags / nbc_weights_flags): a row whose largest |w| lies below
                         2^-51 or above 2^81 is normalised only as far as k in [-66, 80] reaches (NBC_PACK_ROW_CLAMPED),
                         and a (scale, shift) that leaves f32's normal range under its powers of two is no longer exact
                         (NBC_PACK_SCALE_RANGE); a caller that sees either runs NBC_PREC_FP32 (the folder driver does).
                         A product is P*X0 + Q*X0 + (P*2^-11)*X1, three EXACT f16 products on v_mfma_f32_16x16x32_f16,
                         summed in ONE f32 chain per 256 channels that joins a running f32 sum (two levels, like the f32
                         mode); the dropped Q*X1 is 2^-22 relative at worst.  Error against float64 at the level of
                         NBC_PREC_FP32 (profiles/r04_fp64_adjudication_*: 4.2-5.2e-6 on logits of range 2-3.5, the f32
                         MFMA 4.0-6.2e-6, the CPU reference 3.6-4.5e-6), same tolerances in the tests. */
};

/* What nbc_pack_weights had to give up (bits of nbc_packed_weights_flags / nbc_weights_flags; 0 = nothing): NBC_PREC_F16X2 only. */
enum {
  NBC_PACK_ROW_CLAMPED = 1,  /* a weight row beyond the reach of the row normalisation (largest |w| < 2^-51 or > 2^81): its
                                pieces keep fewer bits than f32 */
  NBC_PACK_SCALE_RANGE = 2   /* a BatchNorm scale / shift left f32's normal range under the powers of two folded into it */
};

/* Layout of the image handed to nbc_forward. */
enum {
  NBC_IN_F32_NCHW = 0,  /* float32 [N,3,H,W], already normalised: exactly `batch[0]` of models.py:269 */
  NBC_IN_U8_NHWC = 1    /* uint8 [N,H,W,3] RGB as decoded by pil_loader (dataset.py:82-90); the
                           library applies ToTensor (/255) and Normalize((x-mean)/std) in f32 */
};

enum {
  NBC_LABEL_U8 = 0,   /* uint8 [N,H,W]  */
  NBC_LABEL_I64 = 1   /* int64 [N,H,W]: dtype of torch.argmax at models.py:270 */
};

/* One entry of a state_dict (models.py:222).  `data` is HOST memory, C-contiguous. */
typedef struct {
  const char* name;   /* e.g. "backbone.layer1.0.conv1.weight" */
  const void* data;
  int64_t shape[4];
  int32_t ndim;
  int32_t dtype;      /* 0 = float32, 1 = int64 (num_batches_tracked: checked for presence, unused) */
} nbc_tensor;

/* One convolution unit of the network, in execution order (introspection / tests). */
typedef struct {
  char name[64];      /* state_dict prefix of the conv */
  char bn[64];        /* state_dict prefix of its BatchNorm, "" for classifier.4 */
  int32_t cin, cout, k, stride, pad, dil;
  int32_t relu, bias, residual;
} nbc_conv_desc;

/* Per-op record of the last profiled forwards (nbc_set_profiling). */
typedef struct {
  char name[64];      /* conv unit name, or "ingest" / "maxpool" / "upsample_argmax" */
  char kernel[32];    /* kernel family: "conv_dma", "head1x1", ... */
  float ms;           /* mean HIP-event time of the op per forward on the forward's stream (all its launches together) */
  int32_t calls;      /* forwards averaged over */
  double flops;       /* algorithmic: 2*MAC of the convolution (0 for non-conv ops) */
  double bytes;       /* algorithmic: input read once + weights once + output once (+ identity) */
  int32_t kh, kw;     /* kernel extent (0 for non-conv) */
  int32_t cout;       /* output channels (conv ops): Co % 128 == 0 selects the wide-tile kernel */
  int32_t launches;   /* launches of the op per forward (1) */
} nbc_op_record;

const char* nbc_last_error(void);
const char* nbc_version(void);

/* ---- topology introspection (no GPU needed) ------------------------------------------- */
int nbc_num_convs(void);
int nbc_conv_info(int index, nbc_conv_desc* out);
int nbc_num_state_keys(void);                               /* 326 */
int nbc_state_key(int index, const char** name, int64_t shape[4], int32_t* ndim, int32_t* dtype);
/* Low-resolution logits size for an HxW input (three stride-2 stages). */
int nbc_lowres_size(int H, int W, int* h, int* w);

/* ---- weights (host side; no GPU needed) ------------------------------------------------ */
/* The two f16 pieces NBC_PREC_F16X2 keeps of each of n f32 ACTIVATION values (host arithmetic, bit patterns of IEEE
 * binary16): h0 = f16(x) rounded to nearest even, h1 = f16((x - h0) * 2^11).  What the kernels do to every activation
 * (nbc_pack_weights uses the same rounding on the normalised weight rows, with an unscaled low piece); exported so that
 * the conversion can be checked against another implementation of binary16 rounding. */
int nbc_split_f16x2(const float* x, size_t n, uint16_t* h0, uint16_t* h1);

/* Size in bytes of the packed weight blob for a precision (same on every rank). */
size_t nbc_packed_weights_bytes(int precision);
/* Strict key/shape check like nn.Module.load_state_dict (models.py:222): NBC_ERR_KEYS with a
 * message listing missing / unexpected / mis-shaped entries.  Folds each BatchNorm into an f32
 * (scale, shift) pair, reorders conv weights OIHW -> [O][kh][kw][I] (K-major panels, zero padded
 * to whole 128-byte K-steps), converts to the precision's element type, writes `blob`. */
int nbc_pack_weights(const nbc_tensor* tensors, int n, int precision, void* blob, size_t blob_bytes);
/* NBC_PACK_* bits of a packed blob in HOST memory (>= 0), or a negative error.  The bits ride in the blob's trailer, so
 * a rank that received the blob by broadcast reads the same ones (nbc_weights_flags on its context). */
int nbc_packed_weights_flags(const void* blob, size_t blob_bytes, int precision);

/* ---- context --------------------------------------------------------------------------- */
int nbc_create(nbc_ctx** out, int hip_device);
int nbc_destroy(nbc_ctx* ctx);
/* Attach a packed blob that already lives in DEVICE memory and stays owned by the caller
 * (e.g. a torch tensor that was the target of an RCCL broadcast).  Must outlive the context
 * or the next attach. */
int nbc_attach_weights(nbc_ctx* ctx, const void* dev_blob, size_t bytes, int precision);
/* Convenience: pack on the host, allocate device memory owned by the context, upload. */
int nbc_load_weights(nbc_ctx* ctx, const nbc_tensor* tensors, int n, int precision);
/* NBC_PACK_* bits of the attached blob (read from its trailer when it was attached: a 1-KiB device-to-host copy), >= 0,
 * or a negative error (no weights attached). */
int nbc_weights_flags(nbc_ctx* ctx);
/* The power of two a the output tensor of conv unit `name` (or "backbone.maxpool") is STORED with in the attached blob's
 * precision: stored = 2^a x the tensor the network defines (0 outside NBC_PREC_F16X2 and for every tensor of an ordinary
 * checkpoint; see NBC_PREC_F16X2).  Introspection / tests; NBC_ERR_INVALID for an unknown name. */
int nbc_activation_exponent(nbc_ctx* ctx, const char* name, int32_t* exponent);
/* Multi-GPU start-up (the reference has no counterpart: it is single-device,
 * predict.py:66-70): RCCL broadcast of the packed weight blob from rank `root` of `rccl_comm` (an
 * ncclComm_t of the host process, one rank per GPU) on `hip_stream`.  The root must have weights of
 * `precision` attached (nbc_load_weights / nbc_attach_weights: it alone read the checkpoint,
 * predict.py:57); every other rank allocates a blob the context owns, receives into it and attaches
 * it.  The RCCL entry points are looked up in the host process at call time (the library links
 * libamdhip64 only); NBC_ERR_STATE when the process holds no RCCL.  The blob is valid once the stream
 * has been synchronised. */
int nbc_bcast_weights(nbc_ctx* ctx, void* rccl_comm, int root, int precision, void* hip_stream);
/* mean/std used for NBC_IN_U8_NHWC input; defaults are models.py:208-209. */
int nbc_set_normalization(nbc_ctx* ctx, const float mean[3], const float std[3]);
/* Pre-size the workspace for an (N,H,W) so that the first nbc_forward does not allocate.  Buffers only grow, and a
 * growth frees and reallocates (which synchronises the device): a caller that will see many shapes reserves the
 * largest one first, as the folder driver does. */
int nbc_reserve(nbc_ctx* ctx, int N, int H, int W);

/* ---- the hot path ------------------------------------------------------------------------
 * x_dev                 device pointer, layout per x_dtype
 * logits_lowres_dev     nullable, float32 [N,3,h,w] (output of classifier.4, models.py:121)
 * logits_full_dev       nullable, float32 [N,3,H,W] (what self.model(x) returns, models.py:269)
 * labels_dev            nullable, [N,H,W] per labels_dtype (models.py:270; ties -> lowest index,
 *                       NaN counts as the maximum, like torch.argmax)
 * counts_dev            nullable, int64 [N,3]: pixels per class after the optional remap
 * exclude_nodes         non-zero: label 2 -> 1 after the argmax (models.py:273-276)
 * hip_stream            hipStream_t (NULL = default stream)
 */
int nbc_forward(nbc_ctx* ctx, const void* x_dev, int x_dtype, int N, int H, int W,
                float* logits_lowres_dev, float* logits_full_dev,
                void* labels_dev, int labels_dtype, int64_t* counts_dev,
                int exclude_nodes, void* hip_stream);

/* The tail of the path on its own: bicubic upsample (models.py:38-41) of caller-supplied
 * low-resolution logits float32 [N,3,h,w] to HxW + argmax (models.py:270) + remap + counts.
 * Same output arguments as nbc_forward.  Used to test tie / NaN behaviour with crafted logits. */
int nbc_upsample_argmax(nbc_ctx* ctx, const float* logits_lowres_dev, int N, int h, int w, int H, int W,
                        float* logits_full_dev, void* labels_dev, int labels_dtype,
                        int64_t* counts_dev, int exclude_nodes, void* hip_stream);

/* remove_small_zones of utils.py:135-148 (called at models.py:271, between the argmax and the
 * statistics) on device labels, in place: with m = (labels == 0), 8-connected components of ~m smaller
 * than min_pixels join m (skimage remove_small_holes, connectivity=2), then 8-connected components of
 * the filled m smaller than min_pixels leave it (remove_small_objects); pixels that left m and were
 * class 0 become class 1, pixels that joined it become class 0.  The reference uses 150 pixels.
 * labels_dev: uint8 or int64 [N,H,W] (labels_dtype NBC_LABEL_U8 / NBC_LABEL_I64).  exclude_nodes
 * applies the 2 -> 1 remap of models.py:273-276 afterwards; counts_dev (nullable, int64 [N,3]) receives
 * the pixels per class of the result (the counting of models.py:324-331).  N <= 65535. */
int nbc_remove_small_zones(nbc_ctx* ctx, void* labels_dev, int labels_dtype, int N, int H, int W, int min_pixels,
                           int exclude_nodes, int64_t* counts_dev, void* hip_stream);

/* The resize of the reference's preprocessor (models.py:191-198): uint8 RGB [H,W,3] on the device ->
 * ToTensor (u8 / 255 in float32) -> skimage.transform.resize(order=3, mode='reflect',
 * anti_aliasing=False) to out_h x out_w (4-tap Catmull-Rom, all arithmetic in float32 like scikit-image's
 * compiled warp, reflected borders, clipped to the input range) -> float32 [out_h,out_w,3] on the device.
 * Bit-identical to the numpy restatement in neuralbarkcalculator_amd/predict.py, which equals
 * scikit-image 0.18.3's output value for value on the committed fixtures. */
int nbc_resize_cubic_u8(nbc_ctx* ctx, const uint8_t* src_dev, int H, int W, float* dst_dev, int out_h, int out_w,
                        void* hip_stream);
/* The same resize followed by what the reference does with its result (models.py:199-203), for a square
 * target: dst_u8_dev uint8 [out_h,out_w,3] = the bytes skimage.io.imsave writes for that float image
 * (uint8(float64(x) * 255 + 0.499999999), imageio's float -> uint8 path), and row_lit_dev (nullable, int32
 * [out_h]) = per row the number of pixels whose float32 channel sum exceeds 1e-3, i.e. trim_black's `lit`
 * test (models.py:158-159); the caller drops the leading / trailing rows with count / out_w <= 0.85. */
int nbc_preprocess_u8(nbc_ctx* ctx, const uint8_t* src_dev, int H, int W, uint8_t* dst_u8_dev, int32_t* row_lit_dev,
                      int out_h, int out_w, void* hip_stream);

/* Tuning / test knob for the convolution kernel: tile = -1 (per-layer choice) or 0..20 (pixels x channels): 0 128x64,
 * 1 128x128 (4 waves of 64x64, 2 stages, two blocks per CU), 2 256x128, 3 256x256 (bf16), 4 128x128 (4 stages), 5 128x256,
 * 6 256x64, 7 128x64 (2 stages), 8 64x128, 9 / 10 the 8-wave 128x128 / 128x64, 11 the 16-wave 256x128, 12 the 16-wave
 * 256x256 (bf16), 13 the 8-wave 128x128 of 64x32 wave tiles, and f16x2 only: 14 / 15 = 13 / 10 with four loader waves,
 * 16 = 128x128 of four 64x64 waves + four loader waves, 17 = 13 with two stages (two blocks per CU).  f16x2 has no tile
 * 2, 3, 4, 11, 12; f32 no 3, 12.  18 / 19 / 20 are the row-resident 3x3 kernel of f16x2 (csrc/conv3x3_rows.hip: an image
 * row x 128 / 64 channels, or two rows x 64 channels, the rows in LDS for their three taps, one barrier per (channel block,
 * kernel row)): the stride-1 3x3 layers of 128-pixel-wide maps run on 18 or 20 (256 output channels or more: same K order,
 * same bits on both) or 19 (64 / 128) AND ON NOTHING ELSE -- their K order (channel block, kh, kw) is the layer's, so a
 * forced or measured generic tile leaves them where they are -- and no other layer runs on them.  Forced wherever the
 * layer's kind, Cout and the precision allow it; where they do not, the planned tile runs. */
int nbc_set_conv_tile(nbc_ctx* ctx, int tile);

/* Per-layer tile choice by measurement: runs one forward on x (so that the workspace holds real
 * activations), then times every tile shape of the LDS-DMA kernel on every convolution of the
 * current (N,H,W) plan (`reps` launches each, HIP events) and keeps the fastest.  Results do not
 * depend on the tile (same K order, one accumulator per output), only speed does.  Without this call a
 * plan runs on per-layer defaults from a cost model (rounds of blocks per CU x per-block time, fitted to
 * measurements), within 0.1-0.6 % (f32) / 0.5-4.4 % (bf16) of the measured best at any image height; the
 * measurement costs 0.5-0.9 s per shape.  The choice is
 * part of the plan; the context keeps the plans (and choices) of the last 64 shapes it has seen, so a
 * folder of height-trimmed images tunes each distinct (N,H,W) once.
 * nbc_get_plan_tiles copies the tile id of each conv launch of the plan; returns their number. */
int nbc_autotune(nbc_ctx* ctx, const void* x_dev, int x_dtype, int N, int H, int W, int reps, int objective,
                 void* hip_stream);   /* objective 0: time of the launch alone; 1: time x fraction of the 256
                                         CUs it occupies (forwards overlapped on several streams) */
int nbc_get_plan_tiles(nbc_ctx* ctx, int32_t* tiles, int capacity);
/* The default tile id the cost model gives a convolution with M = N*Ho*Wo output pixels, Cout output channels and
 * K = Cin*kh*kw products per output (host arithmetic only: no device needed); -1 for an unknown precision or a Cout
 * no tile divides. */
int nbc_default_conv_tile(int M, int Cout, int K, int precision);
/* Install a tile choice (one id per conv launch of the current plan, as nbc_get_plan_tiles returns them),
 * e.g. one measured in an earlier process: NBC_ERR_INVALID when the count or a tile does not fit. */
int nbc_set_plan_tiles(nbc_ctx* ctx, const int32_t* tiles, int n);

/* 1 when a forward of this context since the last reset produced a logit that is NaN or infinite, else 0 (negative:
 * error).  Sticky, raised by classifier.4's launch at no measurable cost.  NaN / inf in the input or the weights do
 * that in every mode, like in the reference; in NBC_PREC_F16X2 so does an activation beyond f16's range (+-65504), which
 * that mode cannot represent: a caller that runs unknown weights in f16x2 checks this after its last forward and falls
 * back to NBC_PREC_FP32 when it is raised (the folder driver does).  Synchronises the device. */
int nbc_nonfinite_seen(nbc_ctx* ctx, int reset);
/* The same word without a synchronisation: enqueues on hip_stream a copy of it to *host_dst, valid
 * once work enqueued behind it on that stream has been waited for; non-zero = raised by a forward that ran on this
 * context ahead of the copy.  host_dst must be PINNED host memory (hipHostMalloc / hipHostRegister): a copy to pageable
 * memory is staged and may block, which is what this call exists to avoid -- NBC_ERR_INVALID otherwise.  The folder driver sends it along with every batch's labels, so that an f16x2 run of
 * weights that mode cannot carry is abandoned at the first batch that shows it, not after the last. */
int nbc_nonfinite_peek_async(nbc_ctx* ctx, uint32_t* host_dst, void* hip_stream);

/* ---- debugging / measurement ----------------------------------------------------------- */
/* Copy the activation written by conv unit `name` during the last forward to `dst_host` as
 * float32 NCHW.  `capacity` is in elements.  Only valid when keep-activations is on. */
int nbc_set_keep_activations(nbc_ctx* ctx, int on);
/* The calibration guard of NBC_PREC_F16X2 (any precision answers): after a forward with keep-activations on, the largest
 * finite |value| of every conv unit's output tensor AS STORED on the device (with the power of two nbc_pack_weights gave it),
 * one float per conv unit in nbc_conv_info order (0 for classifier.4, whose output is the logits); returns their number
 * (nbc_num_convs()) or a negative error.  nbc_pack_weights places a tensor by its BatchNorm's promise; this is what the
 * data did.  A tensor that peaks below 2^-8 has most of its values under f16x2's 2^-12 floor (absolute error 2^-36), one
 * beyond 2^14 is a factor four from f16's range: a caller that sees either on a frame of its data runs NBC_PREC_FP32
 * (the folder driver checks its first image: neuralbarkcalculator_amd/predict.py).  Synchronises the device. */
int nbc_activation_peaks(nbc_ctx* ctx, float* peaks_host, int capacity);
int nbc_read_activation(nbc_ctx* ctx, const char* name, float* dst_host, size_t capacity,
                        int64_t shape[4]);
/* When on, every launch of the next forwards is bracketed by HIP events on the forward's stream
 * (no synchronisation inside nbc_forward).  nbc_num_op_records() waits for the last profiled
 * forward, averages each launch over all forwards profiled since the previous call, resets the
 * accumulation and returns the number of records (one per launch of the plan). */
int nbc_set_profiling(nbc_ctx* ctx, int on);
int nbc_num_op_records(nbc_ctx* ctx);
int nbc_get_op_record(nbc_ctx* ctx, int index, nbc_op_record* out);

#ifdef __cplusplus
}
#endif
#endif /* NBC_H */
