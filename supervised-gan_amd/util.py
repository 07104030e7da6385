"""Image helpers of the drivers (util/util.py:15-28, :41-43 of the reference): tensor -> uint8 image -> PNG."""
import os

import numpy as np


def tensor2im(image_tensor, imtype=np.uint8):
    """[1, C, H, W] in [-1, 1] -> [H, W, 3] uint8; 1 channel is repeated, 2 channels get a zero blue plane (util.py:15-24)."""
    image_numpy = image_tensor[0].detach().cpu().float().numpy()
    image_numpy = (image_numpy + 1) / 2.0 * 255.0
    if image_numpy.shape[0] == 1:
        image_numpy = image_numpy.repeat(3, 0)
    elif image_numpy.shape[0] == 2:
        image_numpy = np.concatenate((image_numpy, np.zeros((1,) + image_numpy.shape[1:], dtype=image_numpy.dtype)), axis=0)
    return np.transpose(image_numpy, (1, 2, 0)).clip(0, 255).astype(imtype)


def save_image(image_numpy, image_path):
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(image_path)), exist_ok=True)
    Image.fromarray(image_numpy).save(image_path)
