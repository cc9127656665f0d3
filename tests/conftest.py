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
