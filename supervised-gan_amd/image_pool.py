"""ImagePool (util/image_pool.py:6-42): 50-image history of generated images for the D step.
Same policy and the same python `random` draws as the reference; images stay device resident."""
import random

import torch


class ImagePool:
    def __init__(self, pool_size=0, reject=0.5):
        self.pool_size = pool_size
        if self.pool_size > 0:
            self.num_imgs = 0
            self.reject = reject
            self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        return_images = []
        for image in images.detach():
            image = torch.unsqueeze(image, 0)
            if self.num_imgs < self.pool_size:
                self.num_imgs = self.num_imgs + 1
                self.images.append(image.clone())
                return_images.append(image)
            else:
                p = random.uniform(0, 1)
                if p > self.reject:
                    random_id = random.randint(0, self.pool_size - 1)
                    tmp = self.images[random_id]
                    self.images[random_id] = image.clone()
                    return_images.append(tmp)
                else:
                    return_images.append(image)
        if len(return_images) == 1:
            return return_images[0]
        return torch.cat(return_images, 0)

    def sample(self, batchSize=1):
        return_images = []
        for _ in range(batchSize):
            random_id = random.randint(0, self.pool_size - 1)
            return_images.append(self.images[random_id].clone())
        return torch.cat(return_images, 0)
