#!/opt/conda/bin/python3.9
"""Fixtures for remove_small_zones, produced with scikit-image 0.18.3 (the /opt/conda interpreter
of the build container; the reference pins scikit_image==0.15.0, requirements.txt:5, whose
remove_small_holes / remove_small_objects have the same semantics).  The calls below are the ones
at /root/reference/src/bark_calculator/utils.py:135-148, applied to a [1,H,W] array like the
reference does; the torch indexing of utils.py:145-146 is restated with numpy.

Run:  /opt/conda/bin/python3.9 scripts/make_small_zones_goldens.py
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.morphology import remove_small_holes, remove_small_objects  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def reference_small_zones(img):
    img = img.copy()                      # [1,H,W] integer labels
    np_image = (img == 0)
    remove_small_holes(np_image, area_threshold=150, connectivity=2, in_place=True)
    remove_small_objects(np_image, min_size=150, connectivity=2, in_place=True)
    img[(np_image == 0) & (img == 0)] = 1
    img[(np_image != 0) & (img != 0)] = 0
    return img


def blobs(seed, h, w, n, rmax):
    rng = np.random.RandomState(seed)
    lab = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cy, cx, r = rng.randint(0, h), rng.randint(0, w), rng.randint(1, rmax)
        lab[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = rng.randint(0, 3)
    # diagonal chains test 8- vs 4-connectivity; 149/150/151-pixel zones test the threshold
    for k in range(40):
        if 10 + k < h and 20 + k < w:
            lab[10 + k, 20 + k] = 2
    lab[h - 12:h - 2, 2:17] = 1          # 150 px block of class 1 (stays)
    lab[h - 12:h - 2, 30:45] = 1
    lab[h - 3, 44] = 0                   # 149 px: removed
    return lab


def main():
    cases = {"a": blobs(1, 96, 128, 60, 9), "b": blobs(2, 200, 160, 150, 12),
             "c": (np.random.RandomState(3).rand(64, 64) > 0.6).astype(np.uint8) * 2,
             "empty": np.zeros((32, 32), np.uint8), "full": np.ones((32, 32), np.uint8)}
    for name, lab in cases.items():
        out = reference_small_zones(lab[None])[0]
        np.savez_compressed(os.path.join(OUT, f"small_zones_{name}.npz"), labels=lab, expected=out)
        print(name, lab.shape, "changed", int((out != lab).sum()))


if __name__ == "__main__":
    main()
