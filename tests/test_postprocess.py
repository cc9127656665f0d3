"""remove_small_zones (utils.py:135-148) against fixtures generated with scikit-image 0.18.3."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from neuralbarkcalculator_amd.postprocess import remove_small_zones


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "small_zones_*.npz"))))
def test_remove_small_zones_matches_skimage_fixture(path):
    g = np.load(path, allow_pickle=False)
    out = remove_small_zones(g["labels"])
    np.testing.assert_array_equal(out, g["expected"])
    assert out.dtype == g["labels"].dtype
    # idempotent, and batch form equals per-image form
    np.testing.assert_array_equal(remove_small_zones(out), out)
    np.testing.assert_array_equal(remove_small_zones(np.stack([g["labels"]] * 2))[1], out)


def test_fixture_count():
    assert len(glob.glob(os.path.join(GOLDEN, "small_zones_*.npz"))) >= 5
