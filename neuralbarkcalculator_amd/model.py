"""Host-side mirror of the reference's model object for the accelerated path.

``FCNResNet50`` stands where ``fcn_resnet50(pretrained=False)`` is bound to ``self.model`` in
``NeuralBarkCalculator.__init__`` (/root/reference/src/bark_calculator/models.py:221-223) and is
called at ``models.py:269``.  It keeps the reference's surface -- ``load_state_dict``, ``to``,
``eval``, ``__call__`` -- and adds the fused ``predict_labels`` (``models.py:269-270`` plus the
``--exclude_nodes`` remap ``models.py:273-276`` and the per-class counts ``models.py:324-331``).

All arithmetic happens in libnbc_hip.so (hand-written gfx950 kernels) through the C ABI of
``include/nbc.h``; torch only supplies device memory, the current stream and, for multi-GPU
runs, ``torch.distributed`` (RCCL) for the one-off weight broadcast.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib, topology
from .topology import NUM_CLASSES, out_hw

_PRECISIONS = {"fp32": _lib.PREC_FP32, "f32": _lib.PREC_FP32, "float32": _lib.PREC_FP32,
               "bf16": _lib.PREC_BF16, "bfloat16": _lib.PREC_BF16, "f16x2": _lib.PREC_F16X2}


def _as_numpy(v) -> np.ndarray:
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def pack_state_dict(state_dict: Mapping[str, object], precision: str = "fp32") -> np.ndarray:
    """Check keys like ``nn.Module.load_state_dict`` (strict) and return the packed weight blob
    (uint8 numpy array).  Raises ``RuntimeError`` listing missing / unexpected keys."""
    lib = _lib.load()
    prec = _PRECISIONS[precision]
    items = []
    keep = []   # keep numpy arrays / byte strings alive during the call
    for name, value in state_dict.items():
        a = _as_numpy(value)
        if a.dtype == np.int64:
            dt = 1
        else:
            a = a.astype(np.float32, copy=False)
            dt = 0
        shape = a.shape                      # ascontiguousarray promotes 0-dim to 1-dim
        a = np.ascontiguousarray(a)
        if len(shape) > 4:
            raise RuntimeError(f"state_dict entry {name!r} has {len(shape)} dims")
        t = _lib.NbcTensor()
        bname = name.encode()
        t.name = bname
        t.data = a.ctypes.data if a.size else None
        for i in range(4):
            t.shape[i] = shape[i] if i < len(shape) else 1
        t.ndim = len(shape)
        t.dtype = dt
        items.append(t)
        keep.append((a, bname))
    arr = (_lib.NbcTensor * len(items))(*items)
    nbytes = lib.nbc_packed_weights_bytes(prec)
    blob = np.zeros(nbytes, dtype=np.uint8)
    _lib.check(lib.nbc_pack_weights(arr, len(items), prec, blob.ctypes.data, nbytes), "load_state_dict")
    return blob


class FCNResNet50:
    """MI355X-native ``fcn_resnet50`` (3 classes, output stride 8, bicubic upsample), eval mode.

    precision: ``"fp32"`` -- f32 MFMA, the parity mode; ``"f16x2"`` -- f32-grade on the f16 matrix pipe: every f32
    value kept as two f16 pieces (to 2^-23 relative for |x| >= 2^-12, an absolute 2^-36 below; weight rows normalised by
    a power of two per output channel, so their magnitude does not matter), three exact f16 products per product, f32
    two-level sums (include/nbc.h, NBC_PREC_F16X2; same tolerances as "fp32" in the tests); ``"bf16"`` -- bf16 MFMA with
    f32 accumulation and f32 BatchNorm epilogue, the throughput mode.
    """

    def __init__(self, precision: str = "fp32"):
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self._lib = _lib.load()            # fails loudly when libnbc_hip.so is not built
        self.precision = precision
        self._prec = _PRECISIONS[precision]
        self._blob_host: Optional[np.ndarray] = None
        self._blob_dev: Optional[torch.Tensor] = None
        self._ctx = C.c_void_p()
        self.device: Optional[torch.device] = None
        self.training = False

    # ---- nn.Module-like surface ---------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, object], strict: bool = True):
        if not strict:
            raise NotImplementedError("only strict=True is supported (the reference never passes strict=False)")
        self._blob_host = pack_state_dict(state_dict, self.precision)
        if self.device is not None:
            self._upload()
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("FCNResNet50 runs on MI355X only (device 'cuda[:i]'); the CPU path is the reference itself")
        index = device.index if device.index is not None else torch.cuda.current_device()
        device = torch.device("cuda", index)
        if self.device is not None and self.device != device:
            self._destroy()
        if not self._ctx:
            ctx = C.c_void_p()
            _lib.check(self._lib.nbc_create(C.byref(ctx), index), "nbc_create")
            self._ctx = ctx
        self.device = device
        if self._blob_host is not None:
            self._upload()
        return self

    def cuda(self, index: Optional[int] = None):
        return self.to(torch.device("cuda", index if index is not None else torch.cuda.current_device()))

    def eval(self):
        """No-op: the path is eval-only (BN running stats, Dropout identity)."""
        return self

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("training mode is not part of the inference path")
        return self

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """``self.model(x)`` of models.py:269: f32 ``[N,3,H,W]`` -> f32 logits ``[N,3,H,W]``."""
        n, h, w = self._check_input(x)
        logits = torch.empty((n, NUM_CLASSES, h, w), dtype=torch.float32, device=self.device)
        self._forward(x, n, h, w, logits_full=logits)
        return logits

    forward = __call__

    # ---- fused extras ---------------------------------------------------------------------
    def predict_labels(self, x: torch.Tensor, exclude_nodes: bool = False,
                       labels_dtype: torch.dtype = torch.int64,
                       return_lowres: bool = False, small_zones: bool = False):
        """Model call + argmax (+ optional 2->1 remap) + per-class pixel counts in one pass.
        ``small_zones=True`` also applies ``remove_small_zones`` (models.py:271) on the device, before
        the remap like the reference does (models.py:271-276); the counts are those of the final labels.

        Returns ``(labels [N,H,W], counts int64 [N,3])`` (+ ``lowres f32 [N,3,h,w]``)."""
        n, h, w = self._check_input(x)
        if labels_dtype not in (torch.int64, torch.uint8):
            raise ValueError("labels_dtype must be torch.int64 or torch.uint8")
        labels = torch.empty((n, h, w), dtype=labels_dtype, device=self.device)
        counts = torch.empty((n, NUM_CLASSES), dtype=torch.int64, device=self.device)
        lowres = None
        if return_lowres:
            lh, lw = out_hw(h, w)
            lowres = torch.empty((n, NUM_CLASSES, lh, lw), dtype=torch.float32, device=self.device)
        self._forward(x, n, h, w, labels=labels, counts=counts, lowres=lowres,
                      exclude_nodes=exclude_nodes and not small_zones)
        if small_zones:
            labels, counts = self.remove_small_zones(labels, exclude_nodes=exclude_nodes)
        return (labels, counts, lowres) if return_lowres else (labels, counts)

    def lowres_logits(self, x: torch.Tensor) -> torch.Tensor:
        """Output of ``classifier.4`` (models.py:121) before the upsample: f32 ``[N,3,h,w]``."""
        n, h, w = self._check_input(x)
        lh, lw = out_hw(h, w)
        lowres = torch.empty((n, NUM_CLASSES, lh, lw), dtype=torch.float32, device=self.device)
        self._forward(x, n, h, w, lowres=lowres)
        return lowres

    def remove_small_zones(self, labels: torch.Tensor, exclude_nodes: bool = False, min_pixels: int = 150):
        """``utils.remove_small_zones`` (utils.py:135-148, called at models.py:271) on the device, IN PLACE
        on ``labels`` (uint8 or int64 ``[N,H,W]`` / ``[H,W]`` on this model's device): 8-connected zones
        of fewer than ``min_pixels`` pixels of the non-background, then of the filled background, flip
        (class 0 <-> class 1).  ``exclude_nodes`` applies the 2 -> 1 remap of models.py:273-276
        afterwards.  Returns ``(labels, counts int64 [N,3])`` with the pixels per class of the result."""
        self._require_ctx()
        if labels.device != self.device or labels.dtype not in (torch.uint8, torch.int64) or not labels.is_contiguous():
            raise ValueError("labels must be a contiguous uint8 or int64 tensor on %s" % (self.device,))
        if labels.dim() not in (2, 3):
            raise ValueError("labels must be [H,W] or [N,H,W]")
        n = 1 if labels.dim() == 2 else int(labels.shape[0])
        h, w = int(labels.shape[-2]), int(labels.shape[-1])
        counts = torch.empty((n, 3), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.nbc_remove_small_zones(self._ctx, labels.data_ptr(),
                                                        _lib.LABEL_I64 if labels.dtype == torch.int64 else _lib.LABEL_U8,
                                                        n, h, w, int(min_pixels), int(bool(exclude_nodes)),
                                                        counts.data_ptr(), stream), "nbc_remove_small_zones")
        return labels, counts

    def resize_cubic_u8(self, image: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
        """The resize of the reference's preprocessor (models.py:191-198) on the device: uint8 RGB
        ``[H,W,3]`` -> ToTensor -> ``skimage.transform.resize(order=3, mode='reflect',
        anti_aliasing=False)`` -> float32 ``[out_h,out_w,3]``; bit-identical to
        ``predict.resize_bicubic_reflect(image.astype(float32) / 255, out_h, out_w)``."""
        self._require_ctx()
        if image.device != self.device or image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3 \
                or not image.is_contiguous():
            raise ValueError("image must be a contiguous uint8 [H,W,3] tensor on %s" % (self.device,))
        out = torch.empty((int(out_h), int(out_w), 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.nbc_resize_cubic_u8(self._ctx, image.data_ptr(), int(image.shape[0]), int(image.shape[1]),
                                                     out.data_ptr(), int(out_h), int(out_w), stream), "nbc_resize_cubic_u8")
        return out

    def preprocess_u8(self, image: torch.Tensor, out_h: int, out_w: int):
        """``resize_cubic_u8`` followed by the float -> uint8 conversion of ``skimage.io.imsave`` (models.py:203)
        and ``trim_black``'s per-row lit-pixel counts (models.py:158-161), all on the device.  Returns
        ``(uint8 [out_h,out_w,3], int32 [out_h])``."""
        self._require_ctx()
        if image.device != self.device or image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3 \
                or not image.is_contiguous():
            raise ValueError("image must be a contiguous uint8 [H,W,3] tensor on %s" % (self.device,))
        out = torch.empty((int(out_h), int(out_w), 3), dtype=torch.uint8, device=self.device)
        lit = torch.empty((int(out_h),), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.nbc_preprocess_u8(self._ctx, image.data_ptr(), int(image.shape[0]), int(image.shape[1]),
                                                   out.data_ptr(), lit.data_ptr(), int(out_h), int(out_w), stream), "nbc_preprocess_u8")
        return out, lit

    def upsample_argmax(self, lowres: torch.Tensor, size: Tuple[int, int], exclude_nodes: bool = False,
                        labels_dtype: torch.dtype = torch.int64, return_logits: bool = False):
        """Tail of the path on caller-supplied low-res logits f32 ``[N,3,h,w]``:
        bicubic to ``size`` (models.py:38-41), argmax (models.py:270), remap, counts."""
        self._require_ctx()
        if lowres.dtype != torch.float32 or lowres.dim() != 4 or lowres.shape[1] != NUM_CLASSES:
            raise RuntimeError("expected float32 [N,3,h,w]")
        lowres = lowres.contiguous()
        n, _, lh, lw = lowres.shape
        H, W = int(size[0]), int(size[1])
        labels = torch.empty((n, H, W), dtype=labels_dtype, device=self.device)
        counts = torch.empty((n, NUM_CLASSES), dtype=torch.int64, device=self.device)
        logits = torch.empty((n, NUM_CLASSES, H, W), dtype=torch.float32, device=self.device) if return_logits else None
        ldt = _lib.LABEL_I64 if labels_dtype == torch.int64 else _lib.LABEL_U8
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._lib.nbc_upsample_argmax(self._ctx, lowres.data_ptr(), n, lh, lw, H, W,
                                               logits.data_ptr() if logits is not None else None,
                                               labels.data_ptr(), ldt, counts.data_ptr(),
                                               int(bool(exclude_nodes)), stream)
        _lib.check(rc, "nbc_upsample_argmax")
        return (labels, counts, logits) if return_logits else (labels, counts)

    def set_normalization(self, mean, std):
        """mean/std applied to uint8 NHWC input (defaults: models.py:208-209)."""
        m = (C.c_float * 3)(*[float(v) for v in mean])
        s = (C.c_float * 3)(*[float(v) for v in std])
        _lib.check(self._lib.nbc_set_normalization(self._require_ctx(), m, s), "set_normalization")

    def reserve(self, n: int, h: int, w: int):
        """Size the activation workspace ahead of the first call."""
        self._require_weights()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nbc_reserve(self._ctx, n, h, w), "nbc_reserve")

    def clone_shared(self) -> "FCNResNet50":
        """A second model object on the same device that shares this one's packed weights (one copy
        in HBM) but owns its own context and activation workspace, so the two can run concurrently
        on different HIP streams (pipelined batch-1 serving)."""
        self._require_weights()
        other = FCNResNet50(self.precision)
        other.to(self.device)
        other._attach(self._blob_dev)
        return other

    # ---- multi-GPU: one process per GPU, weights read by one rank only ---------------------
    def broadcast_weights(self, src: int = 0, group=None):
        """RCCL broadcast of the packed weight blob from rank ``src`` (the only rank that needs
        ``load_state_dict``); the other ranks call this right after ``to(device)``."""
        import torch.distributed as dist
        if self.device is None:
            raise RuntimeError("call .to(device) before broadcast_weights")
        nbytes = self._lib.nbc_packed_weights_bytes(self._prec)
        if dist.get_rank(group) == src:
            self._require_weights()
            blob = self._blob_dev
        else:
            blob = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        dist.broadcast(blob, src=src, group=group)
        self._attach(blob)
        return self

    # ---- measurement / debugging ------------------------------------------------------------
    def set_profiling(self, on: bool):
        _lib.check(self._lib.nbc_set_profiling(self._require_ctx(), int(on)))

    def op_records(self):
        """Per-launch (name, kernel, ms, flops, bytes, k) of the last profiled forward."""
        out = []
        rec = _lib.NbcOpRecord()
        n = self._lib.nbc_num_op_records(self._require_ctx())
        if n < 0:
            _lib.check(n, "nbc_num_op_records")
        for i in range(n):
            _lib.check(self._lib.nbc_get_op_record(self._ctx, i, C.byref(rec)))
            out.append(dict(name=rec.name.decode(), kernel=rec.kernel.decode(), ms=float(rec.ms),
                            calls=int(rec.calls), flops=float(rec.flops), bytes=float(rec.bytes),
                            k=int(rec.kh), cout=int(rec.cout), launches=int(rec.launches)))
        return out

    @property
    def pack_flags(self) -> int:
        """NBC_PACK_* bits of the weights this model holds (0 = nothing given up): in "f16x2" mode a weight row beyond the
        reach of the row normalisation (``_lib.PACK_ROW_CLAMPED``) or a BatchNorm scale / shift pushed out of f32's normal
        range by the powers of two folded into it (``_lib.PACK_SCALE_RANGE``) -- run such a checkpoint in "fp32".  Read from
        the packed blob's trailer, so a rank that received the blob by broadcast sees the same bits."""
        if self._ctx and self._blob_dev is not None:
            rc = self._lib.nbc_weights_flags(self._ctx)
            if rc < 0:
                _lib.check(rc, "nbc_weights_flags")
            return int(rc)
        if self._blob_host is None:
            raise RuntimeError("no weights loaded")
        rc = self._lib.nbc_packed_weights_flags(self._blob_host.ctypes.data, self._blob_host.size, self._prec)
        if rc < 0:
            _lib.check(rc, "nbc_packed_weights_flags")
        return int(rc)

    def activation_exponent(self, name: str) -> int:
        """Power of two the output tensor of op ``name`` is stored with on the device (0 outside "f16x2" and for every
        tensor of an ordinary checkpoint; nbc_activation_exponent)."""
        e = C.c_int32(0)
        _lib.check(self._lib.nbc_activation_exponent(self._require_ctx(), name.encode(), C.byref(e)), "nbc_activation_exponent")
        return int(e.value)

    def activation_peaks(self, x: torch.Tensor) -> dict:
        """One forward of ``x`` with every activation kept, then the largest finite |value| of each conv unit's output AS
        STORED on the device (nbc_activation_peaks): ``{conv unit name: peak}``, classifier.4 left out.  The calibration
        guard of "f16x2": a tensor that peaks below 2^-8 (or beyond 2^14) on real data belongs in "fp32"
        (``f16x2_range_ok``)."""
        n, h, w = self._check_input(x)
        self.set_keep_activations(True)
        try:
            self._forward(x, n, h, w, lowres=torch.empty((n, NUM_CLASSES) + out_hw(h, w), dtype=torch.float32, device=self.device))
            torch.cuda.synchronize(self.device)
            k = int(self._lib.nbc_num_convs())
            buf = (C.c_float * k)()
            rc = self._lib.nbc_activation_peaks(self._require_ctx(), buf, k)
            if rc < 0:
                _lib.check(rc, "nbc_activation_peaks")
        finally:
            self.set_keep_activations(False)
        names = [u.name for u in topology.conv_units()]
        return {names[i]: float(buf[i]) for i in range(k) if names[i] != "classifier.4"}

    @staticmethod
    def f16x2_range_ok(peaks: dict, low: float = 2.0 ** -8, high: float = 2.0 ** 14):
        """(ok, offenders): whether every stored tensor of ``activation_peaks`` lies where the f16 pieces hold f32 grade."""
        bad = {k: v for k, v in peaks.items() if not (low <= v <= high)}
        return (not bad), bad

    def nonfinite_seen(self, reset: bool = True) -> bool:
        """True when a forward since the last reset produced a NaN / infinite logit (nbc_nonfinite_seen; synchronises).
        In "f16x2" mode that also means an activation left f16's range: rerun such weights in "fp32"."""
        rc = self._lib.nbc_nonfinite_seen(self._require_ctx(), int(reset))
        if rc < 0:
            _lib.check(rc, "nbc_nonfinite_seen")
        return rc == 1

    def nonfinite_peek_async(self, host_word: torch.Tensor):
        """Enqueue, on the current stream, a copy of the sticky non-finite word into ``host_word`` (pinned int32 [1]); no
        synchronisation (nbc_nonfinite_peek_async).  Non-zero once the stream has caught up = a forward of this context
        ahead of the copy produced a NaN / infinite logit."""
        if host_word.dtype != torch.int32 or host_word.numel() != 1 or host_word.is_cuda or not host_word.is_pinned():
            raise ValueError("host_word must be a pinned int32 CPU tensor with one element")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.nbc_nonfinite_peek_async(self._require_ctx(), host_word.data_ptr(), stream), "nbc_nonfinite_peek_async")

    def set_conv_tile(self, tile: int = -1):
        """Tuning/test knob: tile -1 = per-layer choice, 0..20 = one tile shape of the conv kernel (18 / 19 / 20: the row-resident 3x3
        kernel of f16x2, which its layers never leave; the menu:
        include/nbc.h, nbc_set_conv_tile; e.g. 5 = 128x256, 14 = 128x128 with four loader waves and 17 = 128x128 at two
        blocks per CU, both f16x2 only).  A tile that does not exist for the precision, or does not divide a layer's
        Cout, is ignored for that layer: the planned tile runs."""
        _lib.check(self._lib.nbc_set_conv_tile(self._require_ctx(), int(tile)), "nbc_set_conv_tile")

    def autotune(self, x: torch.Tensor, reps: int = 3, objective: str = "latency"):
        """Measure every conv tile shape on every layer for x's (N,H,W) and keep the fastest per
        layer (results are tile-independent).  objective "latency" minimises each launch alone;
        "throughput" weighs a launch by the share of the chip it occupies (use it when several
        forwards overlap on different streams).  Returns the chosen tile ids in launch order."""
        n, h, w = self._check_input(x)
        x = x.contiguous()
        x_dtype = _lib.IN_F32_NCHW if x.dtype == torch.float32 else _lib.IN_U8_NHWC
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.nbc_autotune(self._ctx, x.data_ptr(), x_dtype, n, h, w, int(reps),
                                              1 if objective == "throughput" else 0, stream),
                       "nbc_autotune")
        return self.plan_tiles()

    def plan_tiles(self):
        buf = (C.c_int32 * 64)()
        n = self._lib.nbc_get_plan_tiles(self._require_ctx(), buf, 64)
        return [int(buf[i]) for i in range(min(n, 64))]

    def set_plan_tiles(self, tiles):
        """Install a per-layer tile choice (as ``plan_tiles`` / ``autotune`` return it) for the current
        (N,H,W) plan, e.g. one measured by an earlier process."""
        arr = (C.c_int32 * len(tiles))(*[int(t) for t in tiles])
        _lib.check(self._lib.nbc_set_plan_tiles(self._require_ctx(), arr, len(tiles)), "nbc_set_plan_tiles")

    def set_keep_activations(self, on: bool):
        _lib.check(self._lib.nbc_set_keep_activations(self._require_ctx(), int(on)))

    def read_activation(self, name: str, numel_hint: int) -> np.ndarray:
        """f32 NCHW copy of the activation the op ``name`` wrote in the last forward."""
        buf = np.empty(numel_hint, dtype=np.float32)
        shape = (C.c_int64 * 4)()
        torch.cuda.synchronize(self.device)
        _lib.check(self._lib.nbc_read_activation(self._require_ctx(), name.encode(), buf.ctypes.data,
                                                 buf.size, C.byref(shape)), "read_activation")
        shp = tuple(int(s) for s in shape)
        return buf[: int(np.prod(shp))].reshape(shp)

    # ---- internals --------------------------------------------------------------------------
    def _require_ctx(self):
        if not self._ctx:
            raise RuntimeError("call .to('cuda[:i]') first")
        return self._ctx

    def _require_weights(self):
        self._require_ctx()
        if self._blob_dev is None:
            raise RuntimeError("no weights: call load_state_dict (or broadcast_weights) first")

    def _upload(self):
        blob = torch.from_numpy(self._blob_host).to(self.device, non_blocking=False)
        self._attach(blob)

    def _attach(self, blob: torch.Tensor):
        assert blob.dtype == torch.uint8 and blob.is_contiguous() and blob.device == self.device
        _lib.check(self._lib.nbc_attach_weights(self._require_ctx(), blob.data_ptr(), blob.numel(), self._prec),
                   "nbc_attach_weights")
        self._blob_dev = blob    # keep the device memory alive as long as it is attached

    def _check_input(self, x: torch.Tensor) -> Tuple[int, int, int]:
        self._require_weights()
        if not isinstance(x, torch.Tensor):
            raise TypeError("input must be a torch.Tensor")
        if x.device != self.device:
            raise RuntimeError(f"input is on {x.device}, model is on {self.device}")
        if x.dtype == torch.float32:
            if x.dim() != 4 or x.shape[1] != 3:
                raise RuntimeError(f"expected float32 [N,3,H,W], got {tuple(x.shape)}")
            n, _, h, w = x.shape
        elif x.dtype == torch.uint8:
            if x.dim() != 4 or x.shape[3] != 3:
                raise RuntimeError(f"expected uint8 [N,H,W,3], got {tuple(x.shape)}")
            n, h, w, _ = x.shape
        else:
            raise RuntimeError(f"unsupported input dtype {x.dtype}")
        if h < 8 or w < 8:
            raise RuntimeError("H and W must be >= 8")
        return int(n), int(h), int(w)

    def _forward(self, x, n, h, w, logits_full=None, labels=None, counts=None, lowres=None,
                 exclude_nodes=False):
        x = x.contiguous()
        x_dtype = _lib.IN_F32_NCHW if x.dtype == torch.float32 else _lib.IN_U8_NHWC
        ldt = _lib.LABEL_I64 if (labels is None or labels.dtype == torch.int64) else _lib.LABEL_U8
        ptr = lambda t: (t.data_ptr() if t is not None else None)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._lib.nbc_forward(self._ctx, x.data_ptr(), x_dtype, n, h, w, ptr(lowres), ptr(logits_full),
                                       ptr(labels), ldt, ptr(counts), int(bool(exclude_nodes)), stream)
        _lib.check(rc, "nbc_forward")

    def _destroy(self):
        if self._ctx:
            self._lib.nbc_destroy(self._ctx)
            self._ctx = C.c_void_p()
        self._blob_dev = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass


def fcn_resnet50(pretrained: bool = False, dropout: float = 0.1, precision: str = "fp32") -> FCNResNet50:
    """Factory with the reference's name and arguments (models.py:127).  ``pretrained=True``
    would download ImageNet weights in the reference; there is no network here."""
    if pretrained:
        raise RuntimeError("pretrained=True needs a download; load a local state_dict instead (predict.py:57)")
    del dropout  # identity in eval mode
    return FCNResNet50(precision=precision)
