"""Torch-CPU restatement of the reference's per-image forward.  TEST INFRASTRUCTURE ONLY.

What it restates (paths relative to /root/reference):

* ``src/bark_calculator/models.py:27-43``  ``SimpleSegmentationModel.forward``
  (backbone -> ["out"] -> classifier -> bicubic interpolate to the input size,
  ``align_corners=False``)
* ``src/bark_calculator/models.py:113-124`` ``FCNHead`` (3x3 conv no bias, BN, ReLU,
  Dropout, 1x1 conv with bias)
* ``src/bark_calculator/models.py:127-139`` ``fcn_resnet50`` (torchvision
  ``resnet50(replace_stride_with_dilation=[False, True, True])`` truncated at
  ``layer4`` by ``IntermediateLayerGetter``; 3 classes)
* ``src/bark_calculator/models.py:269-276`` model call + ``argmax(dim=1)`` +
  ``--exclude_nodes`` remap 2 -> 1

Third-party dependency that holds the topology and is ABSENT from /root/reference
and from this image: **torchvision 0.3.0** (pinned in the reference's
``README.md:24``).  Its ResNet-50 / Bottleneck / ``_make_layer`` algorithm is
restated below from its published definition (stride on conv2, ``padding=dilation``,
``dilate`` turns the stage's stride into dilation, first block keeps the previous
dilation).  The arithmetic itself is torch's own ATen CPU ops (conv2d, batch_norm,
relu, max_pool2d, interpolate(bicubic), argmax), i.e. the same library the
reference runs on ``--device=cpu``.

Pinning status: the reference ships no tests, golden vectors or weights for this
path, and its modules cannot be imported here (ordinary
``ModuleNotFoundError: torchvision``).  **parity unpinned** by reference fixtures; the
independent pins are (tests/test_oracle.py): the 326 state_dict key names, the
32 947 779 parameter count, the feature-map shapes, and the closed-form bicubic
weights, and ``oracle/numpy_restatement.py`` — the same forward written again in
float64 numpy from the published formulas, sharing no code with this file — which
agrees with this file to 2e-5 of the logit range on a ragged 40x56 input.  Goldens
under tests/golden/ are produced by THIS file (drift guards).

Decision D1: the oracle runs in eval mode (BN running stats,
Dropout identity).  ``predict.py`` as shipped never calls ``.eval()`` and is therefore
non-deterministic; eval is the author's intent (``__main__.py:300``).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

NUM_CLASSES = 3


class Bottleneck(nn.Module):
    """torchvision Bottleneck: 1x1 -> 3x3(stride, dilation) -> 1x1 (x4), residual add."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class DilatedResNet50Trunk(nn.Module):
    """resnet50(replace_stride_with_dilation=[False, True, True]) up to layer4.

    Child names match torchvision's so that the ``backbone.*`` state_dict keys are
    the ones ``load_state_dict`` at models.py:222 expects.
    """

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.dilation = 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, stride=1, dilate=False)
        self.layer2 = self._make_layer(128, 4, stride=2, dilate=False)
        self.layer3 = self._make_layer(256, 6, stride=2, dilate=True)
        self.layer4 = self._make_layer(512, 3, stride=2, dilate=True)

    def _make_layer(self, planes, blocks, stride, dilate):
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        return self.layer4(x)


class FCNHead(nn.Sequential):
    """models.py:113-124."""

    def __init__(self, in_channels, channels, dropout=0.1):
        inter = in_channels // 4
        super().__init__(
            nn.Conv2d(in_channels, inter, 3, padding=1, bias=False),
            nn.BatchNorm2d(inter),
            nn.ReLU(),
            nn.Dropout(dropout),
            nn.Conv2d(inter, channels, 1),
        )


class OracleFCNResNet50(nn.Module):
    """models.py:27-43 + 127-139, eval mode by construction (D1)."""

    def __init__(self, dropout=0.1):
        super().__init__()
        self.backbone = DilatedResNet50Trunk()
        self.classifier = FCNHead(2048, NUM_CLASSES, dropout)
        self.eval()

    def features(self, x):
        return self.backbone(x)

    def lowres_logits(self, x):
        return self.classifier(self.backbone(x))

    def forward(self, x):
        input_shape = x.shape[-2:]
        y = self.classifier(self.backbone(x))
        return F.interpolate(y, size=input_shape, mode="bicubic", align_corners=False)


@torch.no_grad()
def predict_labels(model: OracleFCNResNet50, x: torch.Tensor, exclude_nodes: bool = False):
    """models.py:269-276 minus remove_small_zones: returns (labels int64 [N,H,W],
    counts int64 [N,3], logits f32 [N,3,H,W], lowres f32 [N,3,h,w])."""
    lowres = model.lowres_logits(x)
    logits = F.interpolate(lowres, size=x.shape[-2:], mode="bicubic", align_corners=False)
    labels = torch.argmax(logits, dim=1)
    if exclude_nodes:
        labels[labels == 2] = 1
    counts = torch.stack([(labels == c).flatten(1).sum(1) for c in range(NUM_CLASSES)], dim=1)
    return labels, counts, logits, lowres


@torch.no_grad()
def layer_outputs(model: OracleFCNResNet50, x: torch.Tensor):
    """Output tensor of every conv unit (after BN / residual / ReLU as fused by the
    HIP path), keyed by the conv's state_dict prefix.  Used for layer-by-layer parity."""
    outs = {}
    bb = model.backbone
    t = bb.relu(bb.bn1(bb.conv1(x)))
    outs["backbone.conv1"] = t
    t = bb.maxpool(t)
    outs["backbone.maxpool"] = t
    for li in (1, 2, 3, 4):
        layer = getattr(bb, f"layer{li}")
        for bi, blk in enumerate(layer):
            p = f"backbone.layer{li}.{bi}"
            a = blk.relu(blk.bn1(blk.conv1(t)))
            outs[p + ".conv1"] = a
            b = blk.relu(blk.bn2(blk.conv2(a)))
            outs[p + ".conv2"] = b
            idt = t
            if blk.downsample is not None:
                idt = blk.downsample(t)
                outs[p + ".downsample.0"] = idt
            t = blk.relu(blk.bn3(blk.conv3(b)) + idt)
            outs[p + ".conv3"] = t
    h = model.classifier
    t = h[2](h[1](h[0](t)))
    outs["classifier.0"] = t
    t = h[4](t)
    outs["classifier.4"] = t
    return outs


def cubic_weights(t: float, A: float = -0.75):
    """Closed-form cubic-convolution taps (Keys, A=-0.75) for fractional offset t,
    the formula ATen's upsample_bicubic2d uses; known-answer pin for the HIP kernel."""
    def c1(x):  # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A

    return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]
