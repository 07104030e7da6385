"""TEST INFRASTRUCTURE (oracle): CPU restatement of the tail of the reference's input pipeline -- crop, horizontal flip, rotation by
a multiple of 90 degrees, ToTensor, Normalize -- for checking `sgan_image_prep` (supervised-gan_amd/csrc/sgan_ew.hip).

Reference: data/base_dataset.py:17-55 (get_transform: RandomCrop -> RandomHorizontalFlip -> __rotate -> ToTensor -> Normalize),
data/aligned_dataset.py:31-42 (shared crop offsets and flip for the A|B halves).  The operations themselves live in third-party
code the reference calls: Pillow (`Image.crop`, `Image.transpose`, `Image.rotate`; the version installed in this image) and
torchvision's thin wrappers around them (absent here; `ToTensor` = uint8 / 255 as float32, `Normalize` = (x - mean) / std).
`prep_pil` runs the Pillow calls in the reference's order; `prep_numpy` restates them as index arithmetic; tests/test_oracle_golden.py
pins the second against the first on seeded images.  Only tests import this module."""
import numpy as np


def prep_pil(img_u8, x0, y0, n, flip, rot):
    """img_u8: [H0, W0, 3] uint8.  Returns float32 [3, n, n]."""
    from PIL import Image
    img = Image.fromarray(img_u8, "RGB")
    img = img.crop((x0, y0, x0 + n, y0 + n))                       # transforms.RandomCrop(fineSize) with these offsets
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)                 # transforms.RandomHorizontalFlip
    img = img.rotate(90 * rot, resample=Image.BILINEAR, expand=0)  # base_dataset.py:52-55 (exact transpose path: square image)
    t = np.asarray(img, dtype=np.uint8).astype(np.float32) / np.float32(255.0)      # ToTensor
    t = (t - np.float32(0.5)) / np.float32(0.5)                    # Normalize((.5,.5,.5), (.5,.5,.5))
    return np.ascontiguousarray(t.transpose(2, 0, 1))


def prep_numpy(img_u8, x0, y0, n, flip, rot):
    c = img_u8[y0:y0 + n, x0:x0 + n]
    if flip:
        c = c[:, ::-1]
    c = np.rot90(c, rot, axes=(0, 1))                              # counter-clockwise, as PIL's ROTATE_90
    t = c.astype(np.float32) / np.float32(255.0)
    t = (t - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(t.transpose(2, 0, 1))


# ------------------------------------------------------------------------------------------------------------------------------
# Image.resize(size, BILINEAR | BICUBIC) on 8-bit images: the step in front of the crop (data/base_dataset.py:19-21,43-50
# transforms.Scale / __scale_width, BILINEAR; data/aligned_dataset.py:25 AB.resize(..., BICUBIC)).  The arithmetic is Pillow's
# (third-party, the version installed in this image; src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
# ImagingResampleHorizontal_8bpc / Vertical_8bpc): a separable two-pass convolution, horizontal first, the filter stretched by the
# down-scale factor, coefficients normalised in double and rounded to 22-bit fixed point, each pass rounded and clipped to 8 bits.
# `resize_pil` is the Pillow call; `resize_numpy` restates it; tests/test_oracle_golden.py pins the second against the first.
# ------------------------------------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2
FILTERS = {"bilinear": 1.0, "bicubic": 2.0}      # filter support


def _filter(name, x):
    x = -x if x < 0.0 else x
    if name == "bilinear":
        return 1.0 - x if x < 1.0 else 0.0
    a = -0.5
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size, name):
    """(bounds [out_size, 2] = first source index and tap count, kk [out_size, ksize] fixed-point taps) of one axis."""
    import math
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = FILTERS[name] * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_filter(name, (x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, out_size, name, axis):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, kk = resample_coeffs(img.shape[0], out_size, name)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), img[x0:x0 + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize_numpy(img_u8, wo, ho, name):
    """img_u8 [H, W, C] uint8 -> [ho, wo, C] uint8, Pillow's Image.resize((wo, ho), name)."""
    h, w = img_u8.shape[:2]
    out = img_u8
    if wo != w:
        out = _pass(out, wo, name, 1)
    if ho != h:
        out = _pass(out, ho, name, 0)
    return np.ascontiguousarray(out)


def resize_pil(img_u8, wo, ho, name):
    from PIL import Image
    mode = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC}[name]
    return np.asarray(Image.fromarray(img_u8, "RGB").resize((wo, ho), mode), dtype=np.uint8)
