"""History buffer of generated images for the discriminator step -- the policy of util/image_pool.py:6-42 (Shrivastava et al.):
the first `pool_size` images pass through and are remembered; afterwards an incoming image is, with probability
1 - reject, swapped against a uniformly drawn remembered one.  Python's `random` supplies the draws in the reference's order
(one uniform(0, 1), then one randint), so a seeded run picks the same slots.

MI355X layout: the history is ONE preallocated device tensor [pool_size, C, H, W] (allocated on the first query), slots are
overwritten in place; nothing is re-allocated per step and the whole history is a single contiguous 100 MB block at 512x512."""
import random

import torch


class ImagePool:
    def __init__(self, pool_size=0, reject=0.5):
        self.pool_size = int(pool_size)
        self.reject = reject
        self.num_imgs = 0
        self._slots = None          # [pool_size, C, H, W], lazily allocated like the first stored image

    def _store(self, idx, image):
        if self._slots is None:
            self._slots = torch.empty((self.pool_size,) + tuple(image.shape), dtype=image.dtype, device=image.device)
        self._slots[idx].copy_(image)

    def query(self, images):
        if self.pool_size == 0:
            return images
        out, swapped = [], False
        for image in images.detach().unbind(0):
            if self.num_imgs < self.pool_size:              # still filling: remember it, hand it on
                self._store(self.num_imgs, image)
                self.num_imgs += 1
                out.append(image)
            elif random.uniform(0, 1) > self.reject:        # swap against a remembered image
                idx = random.randint(0, self.pool_size - 1)
                old = self._slots[idx].clone()
                self._store(idx, image)
                out.append(old)
                swapped = True
            else:
                out.append(image)
        if not swapped:
            return images.detach()        # untouched batch: keep the caller's (zero-copy NHWC-backed) tensor
        return torch.stack(out, 0)

    def sample(self, batchSize=1):
        """`batchSize` remembered images drawn with replacement (util/image_pool.py:35-42)."""
        ids = [random.randint(0, self.pool_size - 1) for _ in range(batchSize)]
        return self._slots[ids].clone() if len(ids) > 1 else self._slots[ids[0]].clone().unsqueeze(0)
