"""Test-set loaders: the reference's disk layout (ImageFolder + Grayscale/Resize/ToTensor,
dataset_helper/chest_x_ray_dataset.py; host-side decode with PIL, as torchvision does) and a synthetic loader
of the same tensor contract: batches (images [B,3,224,224] float32, target [B] int64)."""
from __future__ import annotations

import os

import torch


class SyntheticLoader:
    """shard = (lo, hi): yield only rows [lo, hi) of every global batch (a rank's slice; the generator still runs over the whole
    batch, so row i holds the same pixels at any world size)."""

    def __init__(self, n_batches: int, batch_size: int, num_classes: int, seed: int = 1234, size: int = 224, shard=None):
        self.n, self.B, self.C, self.seed, self.size = n_batches, batch_size, num_classes, seed, size
        self.shard = tuple(shard) if shard is not None else None
        self.global_batch = batch_size

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        lo, hi = self.shard if self.shard is not None else (0, self.B)
        for _ in range(self.n):
            x = torch.rand(self.B, 3, self.size, self.size, generator=g)
            t = torch.randint(0, self.C, (self.B,), generator=g)
            yield (x, t) if self.shard is None else (x[lo:hi].contiguous(), t[lo:hi].contiguous())


class ShardBatchSampler:
    """Batch sampler of ONE rank: for every global batch b of `batch_size` consecutive samples (shuffle=False, drop_last=True:
    classification_train_separately.py:675-681, quirk Q12) the indices of its rows [lo, hi) only -- the rank decodes, pins and
    uploads its own slice of each batch and nothing else.  (lo, hi) = (0, batch_size) is the reference's loader."""

    def __init__(self, n_samples: int, batch_size: int, lo: int, hi: int):
        if not (0 <= lo <= hi <= batch_size):
            raise ValueError(f"shard [{lo}, {hi}) outside the batch of {batch_size}")
        self.n, self.B, self.lo, self.hi = int(n_samples), int(batch_size), int(lo), int(hi)

    def __len__(self):
        return self.n // self.B

    def __iter__(self):
        for b in range(self.n // self.B):
            yield list(range(b * self.B + self.lo, b * self.B + self.hi))


IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")   # torchvision.datasets.folder
PRECAL = {"ChestXRay": ((0.5094, 0.5234, 0.5289), (0.2189, 0.2225, 0.2244)),          # chest_x_ray_dataset.py:73-74
          "ISICSkinCancer": ((0.7187, 0.5684, 0.5464), (0.1212, 0.1325, 0.1434))}      # :139-140


class ImageFolderDataset(torch.utils.data.Dataset):
    """torchvision.datasets.ImageFolder semantics (classes = sorted sub-directories, samples sorted by path, RGB
    loader) followed by the reference's test-time transforms (dataset_helper/chest_x_ray_dataset.py:31-51, 76-96):
      grayscaled:   Grayscale(num_output_channels=3) -> Resize((224, 224)) -> ToTensor
      standardized: Resize((224, 224)) -> ToTensor -> Normalize(pre-calculated mean, std)
    written with the PIL calls torchvision makes for PIL inputs (L conversion, BILINEAR resize, /255)."""

    def __init__(self, root: str, dataset_name: str, preprocess: str, image_size=(224, 224)):
        from PIL import Image  # noqa: F401  (fail early if absent)
        if not os.path.isdir(root):
            raise FileNotFoundError(f"dataset directory not found: {root}")
        if preprocess not in ("grayscaled", "standardized"):
            raise ValueError("Invalid preprocess type")
        if dataset_name not in PRECAL:
            raise ValueError("Dataset name is not valid")
        self.classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for d, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(d, f), self.class_to_idx[c]))
        self.preprocess, self.size = preprocess, tuple(image_size)
        self.mean, self.std = (torch.tensor(v).view(3, 1, 1) for v in PRECAL[dataset_name])

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        import numpy as np
        from PIL import Image
        path, target = self.samples[i]
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB")                                  # default_loader / pil_loader
        if self.preprocess == "grayscaled":
            g = np.array(img.convert("L"), dtype=np.uint8)                      # F.to_grayscale(img, 3)
            img = Image.fromarray(np.dstack([g, g, g]), "RGB")
        img = img.resize(self.size[::-1], Image.BILINEAR)                       # Resize((h, w)) on a PIL image
        x = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).contiguous().float().div(255)   # ToTensor
        if self.preprocess == "standardized":
            x = (x - self.mean) / self.std                                      # Normalize
        return x, target


def get_dataset(args, config):
    """The ChestXRay / ISICSkinCancer (+ *Validate) branches of diffusion/utils.py:146-164: returns the split the
    test / calibration loop iterates (testing, or validation for the *Validate names)."""
    name = config.data.dataset
    base = {"ChestXRay": "ChestXRay", "ISICSkinCancer": "ISICSkinCancer", "ChestXRayValidate": "ChestXRay",
            "ISICSkinCancerValidate": "ISICSkinCancer"}.get(name)
    if base is None:
        raise NotImplementedError(f"dataset '{name}' is outside the accelerated path")
    split = "validation" if name.endswith("Validate") else "testing"
    return ImageFolderDataset(os.path.join(config.data.dataroot, split), base, args.preprocess)


def get_test_loader(args, config, shard=None):
    """The test loader of classification_train_separately.py:674-681 (batch_size from the config, no shuffle, drop_last).
    shard = (lo, hi): this rank's rows of every global batch -- the loader then yields [hi - lo] images and their targets per
    batch (attributes .shard / .global_batch say so), and only those files are opened."""
    n = int(getattr(args, "synthetic_batches", 0) or 0)
    B = config.testing.batch_size
    if n > 0:
        size = int(round((config.model.data_dim / 3) ** 0.5))          # 224 for data_dim = 150528 (3 x 224 x 224)
        if 3 * size * size != config.model.data_dim:
            raise ValueError(f"model.data_dim={config.model.data_dim} is not 3 x S x S")
        return SyntheticLoader(n, B, config.data.num_classes, seed=getattr(args, "seed", 0) or 0, size=size, shard=shard)
    ds = get_dataset(args, config)
    lo, hi = shard if shard is not None else (0, B)
    # pinned batches: the runner uploads them with a non-blocking copy on a side stream (runner._rank_batches) without a staging copy
    loader = torch.utils.data.DataLoader(ds, batch_sampler=ShardBatchSampler(len(ds), B, lo, hi),
                                         num_workers=int(getattr(config.data, "num_workers", 0) or 0), pin_memory=torch.cuda.is_available())
    loader.shard = (lo, hi) if shard is not None else None
    loader.global_batch = B
    return loader
