"""Static description of the network the path runs: the reference's ``fcn_resnet50``.

Restates the layer list of ``/root/reference/src/bark_calculator/models.py:127-139``
(torchvision ``resnet50(replace_stride_with_dilation=[False, True, True])`` cut at
``layer4`` + ``FCNHead(2048, 3)`` from ``models.py:113-124``) as plain data, so that
the host side can (a) check a state_dict's keys the way ``load_state_dict``
(``models.py:222``) does and (b) walk the conv units in execution order.

The same table exists in C++ (``csrc/nbc_net.cpp``); ``tests/test_abi.py::test_topology_through_the_abi``
checks that the two agree through the C-ABI (``nbc_num_convs`` / ``nbc_conv_info``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

NUM_CLASSES = 3
BN_EPS = 1e-5


@dataclass(frozen=True)
class ConvUnit:
    """One convolution with what is fused behind it."""
    name: str                 # state_dict prefix of the conv ("backbone.layer1.0.conv1")
    bn: Optional[str]         # state_dict prefix of its BatchNorm, None for classifier.4
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    dil: int
    relu: bool
    bias: bool = False
    residual: bool = False    # conv3: += identity before the ReLU


def conv_units() -> List[ConvUnit]:
    units = [ConvUnit("backbone.conv1", "backbone.bn1", 3, 64, 7, 2, 3, 1, True)]
    inplanes, dilation = 64, 1
    for li, (planes, blocks, stride, dilate) in enumerate(
            [(64, 3, 1, False), (128, 4, 2, False), (256, 6, 2, True), (512, 3, 2, True)], start=1):
        prev_dil = dilation
        if dilate:
            dilation *= stride
            stride = 1
        for bi in range(blocks):
            p = f"backbone.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            d = prev_dil if bi == 0 else dilation
            units.append(ConvUnit(p + ".conv1", p + ".bn1", inplanes, planes, 1, 1, 0, 1, True))
            units.append(ConvUnit(p + ".conv2", p + ".bn2", planes, planes, 3, s, d, d, True))
            if bi == 0:
                units.append(ConvUnit(p + ".downsample.0", p + ".downsample.1",
                                      inplanes, planes * 4, 1, s, 0, 1, False))
            units.append(ConvUnit(p + ".conv3", p + ".bn3", planes, planes * 4, 1, 1, 0, 1, True,
                                  residual=True))
            inplanes = planes * 4
    units.append(ConvUnit("classifier.0", "classifier.1", 2048, 512, 3, 1, 1, 1, True))
    units.append(ConvUnit("classifier.4", None, 512, NUM_CLASSES, 1, 1, 0, 1, False, bias=True))
    return units


def state_dict_spec() -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, dtype) in ``nn.Module.state_dict()`` order."""
    spec = []

    def bn(prefix, c):
        spec.append((prefix + ".weight", (c,), "float32"))
        spec.append((prefix + ".bias", (c,), "float32"))
        spec.append((prefix + ".running_mean", (c,), "float32"))
        spec.append((prefix + ".running_var", (c,), "float32"))
        spec.append((prefix + ".num_batches_tracked", (), "int64"))

    # nn.Module order inside a Bottleneck: conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}
    units = {u.name: u for u in conv_units()}
    ordered = []
    for u in conv_units():
        if u.name.endswith(".downsample.0"):
            continue
        ordered.append(u)
        if u.name.endswith(".conv3"):
            ds = u.name[:-len("conv3")] + "downsample.0"
            if ds in units:
                ordered.append(units[ds])
    for u in ordered:
        spec.append((u.name + ".weight", (u.cout, u.cin, u.k, u.k), "float32"))
        if u.bias:
            spec.append((u.name + ".bias", (u.cout,), "float32"))
        if u.bn is not None:
            bn(u.bn, u.cout)
    return spec


def out_hw(h: int, w: int) -> Tuple[int, int]:
    """Spatial size of the low-resolution logits for an ``h x w`` input (three stride-2 stages)."""
    for _ in range(3):
        h = (h - 1) // 2 + 1
        w = (w - 1) // 2 + 1
    return h, w
