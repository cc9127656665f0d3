"""Whole-bottleneck fusion of the f16x2 mode (csrc/bottleneck_fused.hip; nbc_set_fusion): ONE launch per bottleneck
without downsample of layer1 / layer2 instead of three convolutions, bit-identical to them.

The fused kernel restates the conv kernel's f16x2 arithmetic product for product, so the check is exact equality of whole
forwards with the fusion off (mask 0): phase by phase (the fused launch stops behind conv1 / behind conv2 and the
unfused convolutions finish the block), bottleneck by bottleneck (one mask bit at a time), and all together; on image sizes
whose layer1 / layer2 maps are not multiples of the patch (tails in both directions), batches, and under the
sub-batched tail."""
import numpy as np
import pytest
import torch

from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BLOCKS = ["backbone.layer1.1", "backbone.layer1.2", "backbone.layer2.1", "backbone.layer2.2", "backbone.layer2.3"]


@pytest.fixture(scope="module")
def model(built_lib, sd_np):
    return FCNResNet50("f16x2").load_state_dict(sd_np).to(DEV)


def frames(idx, h, w):
    return torch.from_numpy(np.stack([synth.make_input(int(i), h, w) for i in idx]))


def run(model, x):
    out = model.predict_labels(x, labels_dtype=torch.uint8, return_lowres=True)
    torch.cuda.synchronize()
    return out


def op_names(model, x):
    model.set_profiling(True)
    model.predict_labels(x, labels_dtype=torch.uint8)
    recs = model.op_records()
    model.set_profiling(False)
    return [r["name"] for r in recs], {r["name"]: r for r in recs}


@pytest.mark.parametrize("shape", [(1, 200, 328), (2, 136, 264), (1, 64, 64), (3, 72, 40)])
def test_fused_bottlenecks_equal_the_three_convolutions(model, shape):
    n, h, w = shape
    x = frames(range(90, 90 + n), h, w).to(DEV)
    try:
        model.set_fusion(0)
        ref = run(model, x)
        names, _ = op_names(model, x)
        assert not any("fused" in s for s in names) and "backbone.layer1.1.conv2" in names
        for stop in (1, 2, 0):                                     # behind conv1, behind conv2, the whole bottleneck
            for mask in (1, 2, 4, 8, 16, 31):
                model.set_fusion(mask, stop)
                got = run(model, x)
                for a, b, what in zip(ref, got, ("labels", "counts", "low-res logits")):
                    assert torch.equal(a, b), f"{what} differ: fusion mask {mask:#x}, stop_after {stop}, shape {shape}"
        model.set_fusion(31)
        names, recs = op_names(model, x)
        for b in BLOCKS:
            assert b + ".fused" in names and b + ".conv1" not in names and b + ".conv3" not in names
            assert recs[b + ".fused"]["kernel"] == "bottleneck_x2" and recs[b + ".fused"]["flops"] > 0
        assert "backbone.layer1.0.conv1" in names and "backbone.layer2.0.downsample.0" in names      # blocks with a downsample stay
        model.set_fusion(-1)                                       # the library's default
        got = run(model, x)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    finally:
        model.set_fusion(-1)


def test_fusion_with_sub_batches_and_other_modes(model, sd_np):
    x = frames(range(95, 100), 104, 136).to(DEV)
    try:
        model.set_fusion(0)
        ref = run(model, x)
        model.set_fusion(31)
        for first, k in (("backbone.layer1.1.conv1", 2), ("backbone.layer2.2.conv1", 3), ("backbone.layer1.0.conv1", 1)):
            model.set_sub_batch(first, k)                          # the tail starts AT a fused bottleneck, or in front of it
            got = run(model, x)
            for a, b in zip(ref, got):
                assert torch.equal(a, b), (first, k)
        model.set_sub_batch(None, 0)
        with pytest.raises(RuntimeError):
            model.set_fusion(64)
    finally:
        model.set_sub_batch(None, 0)
        model.set_fusion(-1)
    # the f32 MFMA and bf16 modes have no fused kernel: the knob changes nothing there
    for mode in ("fp32", "bf16"):
        m = FCNResNet50(mode).load_state_dict(sd_np).to(DEV)
        a = run(m, x)
        m.set_fusion(31)
        names, _ = op_names(m, x)
        assert not any("fused" in s for s in names)
        b = run(m, x)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
