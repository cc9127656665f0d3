import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: compares timings (needs a GPU; never part of -m gpu: run with -m perf)")


def marker_expression_asks_for_perf(expr: str) -> bool:
    """True when the -m expression selects a test BECAUSE it is marked `perf`: it accepts an item marked {gpu, perf}
    and rejects the same item without the perf marker.  `-m perf` and `-m "gpu and perf"` do; `-m gpu`,
    `-m "gpu and not perf"`, `-m "not perf"` and a marker that merely contains the letters (`perfect`) do not."""
    if not expr.strip():
        return False
    from _pytest.mark.expression import Expression
    try:
        e = Expression.compile(expr)
    except Exception:
        return False
    with_perf = e.evaluate(lambda name, **kw: name in ("gpu", "perf"))
    without = e.evaluate(lambda name, **kw: name == "gpu")
    return bool(with_perf and not without)


def pytest_collection_modifyitems(config, items):
    """A test marked `perf` asserts on measured durations: it runs only when the marker expression asks for it, so a
    noisy box cannot turn the correctness suite (-m gpu) red."""
    if marker_expression_asks_for_perf(config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="timing comparison: run with -m perf")
    for item in items:
        if "perf" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_lib():
    """libnbc_hip.so, built in-tree if a source is newer (hipcc cross-compiles without a GPU)."""
    from neuralbarkcalculator_amd import build, _lib
    build.build(verbose=False)          # no-op when the .so is newer than every source
    return _lib.load()


@pytest.fixture(scope="session")
def sd_np():
    """Synthetic 'trained-like' state_dict (numpy), seed 7: the weights of every golden."""
    from neuralbarkcalculator_amd import synth
    return synth.make_state_dict("trained_like", seed=7)


@pytest.fixture(scope="session")
def oracle_model(sd_np):
    from oracle.fcn_resnet50_oracle import OracleFCNResNet50
    # the GPU box reports 256 CPUs but grants a 16-core share: oversubscribing makes the oracle crawl
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))
    m = OracleFCNResNet50()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    return m


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def load_golden_labels(name, b=0):
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(GOLDEN, f"{name}_labels{b}.png")))
    out = np.zeros(a.shape, dtype=np.uint8)
    out[a == 127] = 1
    out[a == 255] = 2
    assert ((a == 0) | (a == 127) | (a == 255)).all()
    return out


def _scale_consumer_bn(out, sd, bn, s, eps=1e-5):
    """The BatchNorm behind a convolution whose INPUT (or weights) grew by the power of two `s`: running_mean x s and
    running_var x s^2 (both exact), gamma x sqrt(var s^2 + eps) / (s sqrt(var + eps)) (float64, rounded once), which undoes
    what eps does to a rescaled variance: the layer computes the same function as before up to that one rounding."""
    var = sd[bn + ".running_var"]
    var_s = var * np.float32(s * s)
    out[bn + ".running_mean"] = out[bn + ".running_mean"] * np.float32(s)
    out[bn + ".running_var"] = var_s
    fix = np.sqrt(var_s.astype(np.float64) + eps) / (s * np.sqrt(var.astype(np.float64) + eps))
    out[bn + ".weight"] = (out[bn + ".weight"].astype(np.float64) * fix).astype(np.float32)


def rescale_conv_weights(sd, s):
    """Every convolution in front of a BatchNorm (all 54 of them) with its weights multiplied by the power of two `s`, the
    BatchNorm's running_mean scaled by s and its running_var by s^2 (both exact), and its weight (gamma) by
    sqrt(var s^2 + eps) / (s sqrt(var + eps)) (float64, rounded once), which undoes what eps does to a rescaled variance:
    the network computes the same function as before up to rounding, every activation keeps its magnitude, and the
    convolution weights sit 2^-12 / 2^-16 lower (or 2^10 higher) than a Kaiming initialisation puts them.  A convolution
    in front of a BatchNorm is scale-free, so a trained checkpoint may look like this (models.py:222 takes any)."""
    from neuralbarkcalculator_amd import topology
    out = dict(sd)
    for u in topology.conv_units():
        if u.bn is None:
            continue
        out[u.name + ".weight"] = sd[u.name + ".weight"] * np.float32(s)
        _scale_consumer_bn(out, sd, u.bn, s)
    return out


def rescale_activations(sd, s, where):
    """The same function computed through ACTIVATIONS that are the power of two `s` times their usual size.  A tensor
    between a BatchNorm (+ ReLU, + max-pool: positively homogeneous) and the next convolution is scale-free: its producing
    BatchNorm's gamma and beta times s (exact) make it s times larger, and the BatchNorm behind every convolution that
    reads it takes the scale off again (running_mean x s, running_var x s^2, gamma compensating eps: `_scale_consumer_bn`).
    `where`:
      "internal": the two tensors inside every bottleneck (produced by bn1 and bn2, read by conv2 and conv3);
      "stream":   the residual stream of every stage (produced by bn3 of every block and downsample.1 of the first, read by
                  the following blocks' conv1, by the next stage's first conv1 and downsample.0, and -- layer4's -- by
                  classifier.0);
      "all":      both, and the stem's output (bn1 of the backbone, read through the max-pool by layer1.0) as well.
    A trained checkpoint may look like any of these (models.py:222 takes any state_dict)."""
    from neuralbarkcalculator_amd import topology
    assert where in ("internal", "stream", "all")
    units = topology.conv_units()
    out = dict(sd)
    s32 = np.float32(s)

    def produce(bn):
        out[bn + ".weight"] = out[bn + ".weight"] * s32
        out[bn + ".bias"] = out[bn + ".bias"] * s32

    blocks = [u.name[:-len(".conv1")] for u in units if u.name.endswith(".conv1") and "layer" in u.name]
    if where in ("internal", "all"):
        for b in blocks:
            produce(b + ".bn1"); _scale_consumer_bn(out, sd, b + ".bn2", s)
            produce(b + ".bn2"); _scale_consumer_bn(out, sd, b + ".bn3", s)
    if where in ("stream", "all"):
        for li in range(1, 5):
            mine = [b for b in blocks if b.startswith("backbone.layer%d." % li)]
            for b in mine:
                produce(b + ".bn3")
            produce(mine[0] + ".downsample.1")
            for b in mine[1:]:
                _scale_consumer_bn(out, sd, b + ".bn1", s)
            if li < 4:
                nxt = "backbone.layer%d.0" % (li + 1)
                _scale_consumer_bn(out, sd, nxt + ".bn1", s)
                _scale_consumer_bn(out, sd, nxt + ".downsample.1", s)
            else:
                _scale_consumer_bn(out, sd, "classifier.1", s)
    if where == "all":
        produce("backbone.bn1")
        _scale_consumer_bn(out, sd, "backbone.layer1.0.bn1", s)
        _scale_consumer_bn(out, sd, "backbone.layer1.0.downsample.1", s)
    return out


def overflowing_state_dict(sd):
    """Weights whose activations leave f16's range (+-65 504) in "f16x2" mode: the stem convolution 3e4 times larger than the
    running statistics of the BatchNorm behind it say (a BatchNorm that never saw such data), so its output sits in the 1e5
    range.  A large BatchNorm gamma no longer does that: nbc_pack_weights stores a tensor whose BatchNorm promises such
    magnitudes with a power of two that brings it back (tests/test_abi.py::test_f16x2_activation_powers_of_two)."""
    big = dict(sd)
    big["backbone.conv1.weight"] = sd["backbone.conv1.weight"] * np.float32(3e4)
    return big
