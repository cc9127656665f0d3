"""Post-processing that sits between the argmax and the statistics in the reference.

``remove_small_zones`` restates /root/reference/src/bark_calculator/utils.py:135-148 (called at
``models.py:271``) on top of ``scipy.ndimage.label`` -- scikit-image, which the reference uses, is not
importable in the GPU image.  Semantics (pinned by fixtures generated with scikit-image 0.18.3,
``tests/golden/small_zones_*.npz``): with ``m = (labels == 0)`` (the "Nothing" mask),

1. ``remove_small_holes(m, area_threshold=150, connectivity=2)``: 8-connected components of ``~m``
   smaller than 150 pixels are filled (become background);
2. ``remove_small_objects(m, min_size=150, connectivity=2)``: 8-connected components of the
   *filled* ``m`` smaller than 150 pixels are removed;
3. background pixels that left the mask become class 1 (Bark); non-background pixels that
   joined it become class 0.

The reference applies it to the whole ``[1,H,W]`` batch array with a 3-D structuring element whose
in-plane slice is the 8-neighbourhood; with batch size 1 (``models.py:249-250``) that equals the
per-image 2-D operation done here.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

SMALL_ZONE_PIXELS = 150          # utils.py:140,143 (the README says 100; the code wins)
_EIGHT = np.ones((3, 3), dtype=bool)


def _drop_small_components(mask: np.ndarray, min_size: int) -> np.ndarray:
    """skimage.morphology.remove_small_objects(mask, min_size, connectivity=2) for a 2-D bool array."""
    lab, n = ndimage.label(mask, structure=_EIGHT)
    if n == 0:
        return mask.copy()
    sizes = np.bincount(lab.ravel())
    too_small = sizes < min_size
    too_small[0] = False
    out = mask.copy()
    out[too_small[lab]] = False
    return out


def remove_small_zones(labels: np.ndarray, min_pixels: int = SMALL_ZONE_PIXELS) -> np.ndarray:
    """labels: integer ``[H,W]`` or ``[N,H,W]`` class map; returns a new array of the same dtype."""
    labels = np.asarray(labels)
    if labels.ndim == 3:
        return np.stack([remove_small_zones(l, min_pixels) for l in labels])
    if labels.ndim != 2:
        raise ValueError("labels must be [H,W] or [N,H,W]")
    bg = labels == 0
    filled = ~_drop_small_components(~bg, min_pixels)        # remove_small_holes
    kept = _drop_small_components(filled, min_pixels)         # remove_small_objects
    out = labels.copy()
    out[(~kept) & (labels == 0)] = 1
    out[kept & (out != 0)] = 0
    return out
