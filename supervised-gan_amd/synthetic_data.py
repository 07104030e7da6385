"""`--dataroot synthetic`: the feeder the drivers use when no image folder is given (the reference's data/ package -- file
listing, PIL decode, crop / flip / rotate -- is outside the hot path).  Yields the dict the trainers' set_input
reads: {'A': [1, 3, H, W] in [-1, 1], 'B': ..., 'A_paths': [...], 'B_paths': [...]}, device resident."""
import torch


class SyntheticDataset:
    def __init__(self, opt, length=64, device=None):
        self.opt, self.length = opt, length
        self.device = device if device is not None else (torch.device('cuda', opt.gpu_ids[0]) if opt.gpu_ids else torch.device('cpu'))
        g = torch.Generator().manual_seed(123 + (opt.manualSeed or 0))
        n = min(length, 16)
        hw = opt.fineSize
        self.ring = [{'A': (torch.rand(1, 3, hw, hw, generator=g) * 2 - 1).to(self.device),
                      'B': (torch.rand(1, 3, hw, hw, generator=g) * 2 - 1).to(self.device),
                      'A_paths': ['synthetic_%04d.png' % i], 'B_paths': ['synthetic_%04d.png' % i]} for i in range(n)]

    def __len__(self):
        return self.length

    def __iter__(self):
        for i in range(self.length):
            yield self.ring[i % len(self.ring)]
