"""Test-set loader.  Disk datasets (torchvision ImageFolder + Grayscale/Resize/ToTensor,
dataset_helper/chest_x_ray_dataset.py) are a 'next' row of SURVEY 8f; the accelerated path ships a
synthetic loader of the same tensor contract: batches (images [B,3,224,224] float32 in [0,1), target [B])."""
from __future__ import annotations

import torch


class SyntheticLoader:
    def __init__(self, n_batches: int, batch_size: int, num_classes: int, seed: int = 1234, size: int = 224):
        self.n, self.B, self.C, self.seed, self.size = n_batches, batch_size, num_classes, seed, size

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            yield (torch.rand(self.B, 3, self.size, self.size, generator=g),
                   torch.randint(0, self.C, (self.B,), generator=g))


def get_test_loader(args, config):
    n = int(getattr(args, "synthetic_batches", 0) or 0)
    if n > 0:
        size = int(round((config.model.data_dim / 3) ** 0.5))          # 224 for data_dim = 150528 (3 x 224 x 224)
        if 3 * size * size != config.model.data_dim:
            raise ValueError(f"model.data_dim={config.model.data_dim} is not 3 x S x S")
        return SyntheticLoader(n, config.testing.batch_size, config.data.num_classes, seed=getattr(args, "seed", 0) or 0, size=size)
    raise NotImplementedError(
        "disk datasets need torchvision's ImageFolder pipeline, which is outside the accelerated hot path; "
        "run with --synthetic_batches N (see INTEGRATION.md for wiring the reference's DataLoader in)")
