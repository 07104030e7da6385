"""Image-folder feeders with the reference's transforms (data/single_dataset.py, data/aligned_dataset.py, data/base_dataset.py:17-55,
data/image_folder.py): file listing and PIL decode on the host; the decoded uint8 image is uploaded once and everything after it
runs on the device -- Image.resize (`sgan_image_resize`, bit-exact with Pillow's two-pass resampler) and ONE kernel for
crop -> flip -> rot90 -> ToTensor -> Normalize (`sgan_image_prep`) -- so neither the resized nor the float image ever exists in
host memory and the trainers' set_input reads a device tensor.

Yields what the reference's DataLoader yields at batchSize 1: {'A': [1, 3, H, W] in [-1, 1], 'A_paths': [path]} (single) or
{'A', 'B', 'A_paths', 'B_paths'} (aligned, unaligned).  Random draws use Python's `random` like the reference; the aligned feeder makes them
in the reference's order (aligned_dataset.py:31-38), the single feeder's crop offsets are torchvision-internal there and drawn
here as x then y."""
import os
import random

import numpy as np
import torch

from . import ops

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP']      # image_folder.py:14-17


def make_dataset(dir):
    """All image files under `dir`, subdirectories included (data/image_folder.py:24-35)."""
    assert os.path.isdir(dir), '%s is not a valid directory' % dir
    images = []
    for root, _, fnames in sorted(os.walk(dir)):
        for fname in fnames:
            if any(fname.endswith(e) for e in IMG_EXTENSIONS):
                images.append(os.path.join(root, fname))
    return images


def _scale_width_size(ow, oh, target_width):
    """(w, h) after __scale_width (base_dataset.py:43-50)."""
    if ow == target_width:
        return ow, oh
    return target_width, int(target_width * oh / ow)


class _FolderDataset:
    def __init__(self, opt, device=None):
        self.opt = opt
        self.device = device if device is not None else torch.device('cuda', opt.gpu_ids[0])
        self.paths = sorted(make_dataset(os.path.join(opt.dataroot, opt.phase)))
        if not self.paths:
            raise RuntimeError('no images under %s' % os.path.join(opt.dataroot, opt.phase))
        self.order = list(range(len(self.paths)))

    def __len__(self):
        return min(len(self.paths), self.opt.max_dataset_size) if getattr(self.opt, 'max_dataset_size', None) else len(self.paths)

    def _to_device(self, img):
        return torch.from_numpy(np.array(img, dtype=np.uint8)).to(self.device, non_blocking=True)      # [H, W, 3], a writable copy

    # An item is made in two halves: _host(index) -- file read, decode, resize: no random draws, safe on worker threads (PIL drops
    # the GIL while it decodes) -- and _device(index, host) -- the random draws in the reference's order, the upload, the kernel.
    def __getitem__(self, index):
        return self._device(index, self._host(index))

    def __iter__(self):
        """--nThreads host decodes run ahead of the consumer (the reference's DataLoader workers, data/custom_dataset_data_loader.py:
        32-37); the random draws and the device work stay on the calling thread, so an epoch is the same with or without them."""
        if not self.opt.serial_batches:
            random.shuffle(self.order)
        order = self.order[:len(self)]
        workers = int(getattr(self.opt, 'nThreads', 0) or 0)
        if workers <= 0:
            for i in order:
                yield self[i]
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as pool:
            ahead, todo = deque(), iter(order)
            for i in todo:
                ahead.append((i, pool.submit(self._host, i)))
                if len(ahead) >= 2 * workers:
                    break
            while ahead:
                i, fut = ahead.popleft()
                nxt = next(todo, None)
                if nxt is not None:
                    ahead.append((nxt, pool.submit(self._host, nxt)))
                yield self._device(i, fut.result())


class SingleFolderDataset(_FolderDataset):
    """SingleDataset (data/single_dataset.py:8-31) + get_transform (data/base_dataset.py:17-41)."""

    def _host(self, index):
        return self._decode(self.paths[index])

    def _device(self, index, img):
        path = self.paths[index]
        return {'A': ops.logical_view(self._draw_and_prep(img, path), 3), 'A_paths': [path]}

    def _decode(self, path):
        from PIL import Image
        img = Image.open(path).convert('RGB')
        img.load()
        return img

    def _resized_size(self, ow, oh):
        opt = self.opt
        if opt.resize_or_crop == 'resize_and_crop':
            return opt.loadSize, opt.loadSize                  # transforms.Scale([loadSize, loadSize], BILINEAR)
        if opt.resize_or_crop == 'scale_width':
            return _scale_width_size(ow, oh, opt.fineSize)
        if opt.resize_or_crop == 'scale_width_and_crop':
            return _scale_width_size(ow, oh, opt.loadSize)
        if opt.resize_or_crop != 'crop':
            raise ValueError('--resize_or_crop %s' % opt.resize_or_crop)
        return ow, oh

    def _draw_and_prep(self, img, path):
        opt, n = self.opt, self.opt.fineSize
        crop = opt.resize_or_crop != 'scale_width'
        w, h = self._resized_size(*img.size)
        if crop:
            if w < n or h < n:
                raise ValueError('image %s is %dx%d after resizing, smaller than --fineSize %d' % (path, w, h, n))
            x0, y0 = random.randint(0, w - n), random.randint(0, h - n)
        else:
            if w != h:
                raise NotImplementedError('scale_width of a non-square image gives a non-square tensor; the MI355X feeder crops squares')
            x0 = y0 = 0
        flip = opt.isTrain and not opt.no_flip and random.random() < 0.5
        rot = random.randint(0, 3) if (opt.isTrain and not opt.no_rotate) else 0
        dev = self._to_device(img)
        if (w, h) != img.size:
            dev = ops.image_resize(dev, w, h, "bilinear")      # Image.resize((w, h), BILINEAR), on the device
        return ops.image_prep(dev, x0, y0, n, flip, rot)


class UnalignedFolderDataset(SingleFolderDataset):
    """UnalignedDataset (data/unaligned_dataset.py:10-41): <root>/<phase>A and <root>/<phase>B, the i-th file of each (modulo its
    length), each through its own draw of the single-image transform."""

    def __init__(self, opt, device=None):
        self.opt = opt
        self.device = device if device is not None else torch.device('cuda', opt.gpu_ids[0])
        self.A_paths = sorted(make_dataset(os.path.join(opt.dataroot, opt.phase + 'A')))
        self.B_paths = sorted(make_dataset(os.path.join(opt.dataroot, opt.phase + 'B')))
        if not self.A_paths or not self.B_paths:
            raise RuntimeError('no images under %s{A,B}' % os.path.join(opt.dataroot, opt.phase))
        self.paths = list(range(max(len(self.A_paths), len(self.B_paths))))
        self.order = list(self.paths)

    def _host(self, index):
        a, b = self.A_paths[index % len(self.A_paths)], self.B_paths[index % len(self.B_paths)]
        return self._decode(a), self._decode(b)

    def _device(self, index, imgs):
        a, b = self.A_paths[index % len(self.A_paths)], self.B_paths[index % len(self.B_paths)]
        A = self._draw_and_prep(imgs[0], a)          # A's draws first, then B's (unaligned_dataset.py:32-33)
        B = self._draw_and_prep(imgs[1], b)
        return {'A': ops.logical_view(A, 3), 'B': ops.logical_view(B, 3), 'A_paths': [a], 'B_paths': [b]}


class AlignedFolderDataset(_FolderDataset):
    """AlignedDataset (data/aligned_dataset.py:10-46): A|B side by side in one file, bicubic resize to (2 loadSize, loadSize), the
    same crop offsets and flip for both halves, no rotation."""

    def __init__(self, opt, device=None):
        assert opt.resize_or_crop == 'resize_and_crop'      # aligned_dataset.py:17
        super().__init__(opt, device)

    def _host(self, index):
        from PIL import Image
        img = Image.open(self.paths[index]).convert('RGB')
        img.load()
        return img

    def _device(self, index, AB):
        opt, path = self.opt, self.paths[index]
        w, h, n = opt.loadSize, opt.loadSize, opt.fineSize
        w_offset = random.randint(0, max(0, w - n - 1))
        h_offset = random.randint(0, max(0, h - n - 1))
        flip = (not opt.no_flip) and random.random() < 0.5
        dev = self._to_device(AB)
        if AB.size != (2 * w, h):
            dev = ops.image_resize(dev, 2 * w, h, "bicubic")   # AB.resize((loadSize * 2, loadSize), Image.BICUBIC), on the device
        A = ops.image_prep(dev, w_offset, h_offset, n, flip, 0)
        B = ops.image_prep(dev, w + w_offset, h_offset, n, flip, 0)
        return {'A': ops.logical_view(A, 3), 'B': ops.logical_view(B, 3), 'A_paths': [path], 'B_paths': [path]}


def create_dataset(opt, device=None):
    """CreateDataLoader / CreateDataset (data/custom_dataset_data_loader.py:6-25) at batchSize 1."""
    if opt.dataset_mode == 'single':
        return SingleFolderDataset(opt, device)
    if opt.dataset_mode == 'aligned':
        return AlignedFolderDataset(opt, device)
    if opt.dataset_mode == 'unaligned':
        return UnalignedFolderDataset(opt, device)
    raise ValueError("Dataset [%s] not recognized." % opt.dataset_mode)
