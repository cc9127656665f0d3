#!/opt/conda/bin/python3.9
"""Fixtures for the reference's Preprocessor (/root/reference/src/bark_calculator/models.py:191-203),
produced with scikit-image 0.18.3 (the /opt/conda interpreter; the reference pins 0.15.0, whose
resize/warp/imsave behave the same for this call).  The calls below are the reference's own:
ToTensor (u8 -> float32 / 255), HWC numpy, `resize(image, (T, T), order=3, mode='reflect',
anti_aliasing=False)` when max(shape) > T, `trim_black` on square images, `imsave` (float -> uint8).
T is the reference's target_size (1024 there; small here to keep the fixtures small).

Run:  /opt/conda/bin/python3.9 scripts/make_preprocess_goldens.py
"""
import os
import tempfile
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.io import imread, imsave  # noqa: E402
from skimage.transform import resize  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def trim_black(image):                                 # models.py:157-166
    summed_image = np.sum(image, axis=-1)
    summed_image = summed_image > 1e-3
    clear_enough_lines_idx = np.mean(summed_image, axis=-1) > 0.85
    first_idx = np.argmax(clear_enough_lines_idx)
    last_idx = image.shape[0] - np.argmax(clear_enough_lines_idx[::-1])
    return image[first_idx:last_idx]


def reference_preprocess(img_u8, target):
    """Returns (what imread gives back from the saved file, the float32 image handed to imsave)."""
    image = (img_u8.astype(np.float32) / np.float32(255))           # ToTensor, then .numpy().transpose(1,2,0)
    if max(image.shape) > target:
        image = resize(image, (target, target), order=3, mode='reflect', anti_aliasing=False)
    if image.shape[0] == image.shape[1]:
        image = trim_black(image)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.png")
        imsave(p, image)
        return imread(p), np.ascontiguousarray(image)


def main():
    rng = np.random.RandomState(11)
    cases = {}
    a = rng.randint(0, 256, (96, 96, 3)).astype(np.uint8)
    a[:10] = 0                                         # black rows on top: trimmed after the resize
    cases["sq96_to24"] = (a, 24)
    b = rng.randint(0, 256, (160, 200, 3)).astype(np.uint8)   # (a linear ramp would put most outputs on x.5 ties)
    cases["rect160x200_to40"] = (b, 40)                # non-square input is squashed to T x T, then trimmed
    c = rng.randint(0, 256, (20, 24, 3)).astype(np.uint8)
    cases["small_untouched"] = (c, 32)                 # max(shape) <= T: no resize, no trim (not square)
    d = rng.randint(30, 256, (64, 64, 3)).astype(np.uint8)
    d[-7:, ::2] = 0                                    # 50 % dark pixels in the last rows: trimmed
    cases["sq64_trim_only"] = (d, 64)
    e = rng.randint(0, 256, (133, 177, 3)).astype(np.uint8)   # non-integer zoom factors 2.66 x 3.54: the sample
    cases["rect133x177_to50"] = (e, 50)                # coordinates themselves are rounded float32 products
    f = rng.randint(0, 256, (61, 53, 3)).astype(np.uint8)
    f[:9] //= 16                                       # dark (not black) band: the clip range matters, nothing is trimmed
    cases["rect61x53_to48"] = (f, 48)                  # factors 1.27 x 1.10, like a 1300 x 1100 scan going to 1024
    for name, (img, t) in cases.items():
        out, pre = reference_preprocess(img, t)
        # `float32` = the image right before imsave's float -> uint8 conversion: lets the restatement be
        # compared value for value, in front of the quantisation
        np.savez_compressed(os.path.join(OUT, f"preprocess_{name}.npz"), image=img, target=np.asarray(t), expected=out,
                            float32=pre.astype(np.float32))
        print(name, img.shape, "->", out.shape, out.dtype, pre.dtype)


if __name__ == "__main__":
    main()
