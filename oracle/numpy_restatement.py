"""Second, independent restatement of the path in float64 numpy.  TEST INFRASTRUCTURE ONLY.

``fcn_resnet50_oracle.py`` leans on torch's own CPU operators for the arithmetic.  This file writes the
same forward from the published formulas with nothing but numpy slicing and ``einsum``, so that the
torch-based oracle is itself checked against something that shares no code with it
(tests/test_oracle.py::test_numpy_restatement_agrees_with_torch_oracle).  Small inputs only.

What it follows (paths relative to /root/reference/src/bark_calculator):
  * models.py:27-43    forward: backbone["out"] -> classifier -> bicubic to the input size (align_corners=False)
  * models.py:113-124  FCNHead: 3x3 conv (no bias) -> BN -> ReLU -> Dropout (identity in eval) -> 1x1 conv + bias
  * models.py:127-139  resnet50(replace_stride_with_dilation=[False, True, True]) cut at layer4
    (torchvision 0.3.0, absent from the reference tree; its published definition: Bottleneck = 1x1 ->
    3x3(stride, padding = dilation) -> 1x1 x4, downsample = 1x1(stride) + BN on the first block of a
    stage, stage stride converted to dilation where requested, the first block of a dilated stage keeps
    the previous dilation)
  * models.py:270      argmax over the class axis (first maximum wins)
  * ATen semantics: conv output size floor((n + 2p - d(k-1) - 1)/s) + 1; eval BatchNorm
    (x - mean) / sqrt(var + 1e-5) * gamma + beta; MaxPool2d(3, 2, 1) pads with -inf; bicubic A = -0.75,
    source coordinate (o + 0.5) * in/out - 0.5 (not clamped), taps i0-1..i0+2 clamped to the image.
"""
from __future__ import annotations

import math

import numpy as np

EPS = 1e-5


def conv2d(x, w, stride=1, padding=0, dilation=1, bias=None):
    """x [C,H,W], w [O,C,kh,kw] -> [O,Ho,Wo]; one einsum per tap."""
    c, h, wd = x.shape
    o, _, kh, kw = w.shape
    ho = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    wo = (wd + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    xp = np.zeros((c, h + 2 * padding, wd + 2 * padding), dtype=np.float64)
    xp[:, padding:padding + h, padding:padding + wd] = x
    out = np.zeros((o, ho, wo), dtype=np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i * dilation:i * dilation + stride * (ho - 1) + 1:stride,
                       j * dilation:j * dilation + stride * (wo - 1) + 1:stride]
            out += np.einsum("oc,chw->ohw", w[:, :, i, j], patch)
    if bias is not None:
        out += bias[:, None, None]
    return out


def batch_norm(x, sd, prefix):
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - m[:, None, None]) / np.sqrt(v[:, None, None] + EPS) * g[:, None, None] + b[:, None, None]


def relu(x):
    return np.maximum(x, 0.0)


def maxpool3x3s2p1(x):
    c, h, w = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    xp = np.full((c, h + 2, w + 2), -np.inf)
    xp[:, 1:1 + h, 1:1 + w] = x
    out = np.full((c, ho, wo), -np.inf)
    for i in range(3):
        for j in range(3):
            out = np.maximum(out, xp[:, i:i + 2 * (ho - 1) + 1:2, j:j + 2 * (wo - 1) + 1:2])
    return out


def cubic_taps(t, a=-0.75):
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    u = 1 - t
    w2 = ((a + 2) * u - (a + 3)) * u * u + 1
    w3 = ((a * (u + 1) - 5 * a) * (u + 1) + 8 * a) * (u + 1) - 4 * a
    return (w0, w1, w2, w3)


def bicubic_matrix(n_in, n_out):
    """[n_out, n_in] interpolation matrix of one axis (align_corners=False)."""
    m = np.zeros((n_out, n_in))
    scale = n_in / n_out
    for o in range(n_out):
        s = (o + 0.5) * scale - 0.5
        i0 = math.floor(s)
        for k, wk in enumerate(cubic_taps(s - i0)):
            m[o, min(max(i0 - 1 + k, 0), n_in - 1)] += wk
    return m


def forward(sd, x):
    """sd: state_dict as numpy arrays (326 keys), x [3,H,W] normalised input -> (lowres [3,h,w], logits [3,H,W], labels [H,W])."""
    sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    t = relu(batch_norm(conv2d(x.astype(np.float64), sd["backbone.conv1.weight"], stride=2, padding=3), sd, "backbone.bn1"))
    t = maxpool3x3s2p1(t)
    dilation = 1
    for li, (blocks, stride, dilate) in enumerate(((3, 1, False), (4, 2, False), (6, 2, True), (3, 2, True)), start=1):
        prev = dilation
        if dilate:                                   # the stage's stride becomes dilation
            dilation *= stride
            stride = 1
        for bi in range(blocks):
            p = f"backbone.layer{li}.{bi}"
            d = prev if bi == 0 else dilation        # first block of a dilated stage keeps the previous dilation
            s = stride if bi == 0 else 1
            a = relu(batch_norm(conv2d(t, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
            b = relu(batch_norm(conv2d(a, sd[p + ".conv2.weight"], stride=s, padding=d, dilation=d), sd, p + ".bn2"))
            c = batch_norm(conv2d(b, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            idt = t
            if bi == 0:
                idt = batch_norm(conv2d(t, sd[p + ".downsample.0.weight"], stride=s), sd, p + ".downsample.1")
            t = relu(c + idt)
    t = relu(batch_norm(conv2d(t, sd["classifier.0.weight"], padding=1), sd, "classifier.1"))
    low = conv2d(t, sd["classifier.4.weight"], bias=sd["classifier.4.bias"])
    my, mx = bicubic_matrix(low.shape[1], x.shape[1]), bicubic_matrix(low.shape[2], x.shape[2])
    # ATen evaluates the x taps first, then the y taps; in float64 the order is immaterial here
    logits = np.einsum("yh,chw,xw->cyx", my, low, mx)
    return low, logits, np.argmax(logits, axis=0)
