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
