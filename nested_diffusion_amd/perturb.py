"""Input-side perturbations of the robustness protocol: mirror of diffusion/utils.py:272-414 (applied by
test_atk at classification_train_separately.py:726-737), same function names and argument meaning.
The random choices (rectangle corners, crop corners) are made on the host with the reference's own RNG calls,
so a seeded run picks the same windows; the pixel work runs in libnd_hip.so."""
from __future__ import annotations

import random
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import check, ptr


def _img(images: torch.Tensor) -> torch.Tensor:
    if not images.is_cuda or images.dtype != torch.float32 or images.dim() != 4:
        raise _lib.NdError("images must be a float32 GPU tensor [B, C, H, W] (no CPU fallback)")
    return images.contiguous()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def add_noise(images_in: torch.Tensor, noise_std: float, z: Optional[torch.Tensor] = None) -> torch.Tensor:
    """utils.py:272-279.  z (optional) replaces the internal torch.randn_like draw."""
    x = _img(images_in)
    z = torch.randn_like(x) if z is None else _img(z)
    out = torch.empty_like(x)
    check(_lib.load().nd_img_add_noise(ptr(x), ptr(z), ptr(out), x.numel(), float(noise_std), _stream(x)), "nd_img_add_noise")
    return out


def adjust_brightness(images_in: torch.Tensor, k: float) -> torch.Tensor:
    """utils.py:390-399."""
    x = _img(images_in)
    out = torch.empty_like(x)
    check(_lib.load().nd_img_brightness(ptr(x), ptr(out), x.numel(), float(k), _stream(x)), "nd_img_brightness")
    return out


def adjust_contrast(images_in: torch.Tensor, k: float) -> torch.Tensor:
    """utils.py:402-414."""
    x = _img(images_in)
    out = torch.empty_like(x)
    ws = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.load().nd_img_contrast(ptr(x), ptr(out), ptr(ws), x.shape[0], x[0].numel(), float(k), _stream(x)), "nd_img_contrast")
    return out


def _resize(x: torch.Tensor, Ho: int, Wo: int, crop: Optional[torch.Tensor] = None, crop_size: int = 0) -> torch.Tensor:
    B, C, Hi, Wi = x.shape
    out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=x.device)
    check(_lib.load().nd_img_resize_bilinear(ptr(x), ptr(out), B, C, Hi, Wi, Ho, Wo, ptr(crop), int(crop_size), _stream(x)),
          "nd_img_resize_bilinear")
    return out


def down_up_sample(images_in: torch.Tensor, k: int) -> torch.Tensor:
    """utils.py:372-387: bilinear down to (H//k, W//k) and back up, align_corners=False."""
    x = _img(images_in)
    H, W = x.shape[-2:]
    return _resize(_resize(x, H // k, W // k), H, W)


def pick_cover_regions(n_images: int, H: int, W: int, k: float, n: int):
    """The rejection sampling of random_cover_new (utils.py:321-343), python `random` calls in the same order."""
    side = int((k * W * H) ** 0.5)
    top_range, left_range = H - side, W - side
    rects = []
    for _ in range(n_images):
        regions = []
        for _ in range(int(n)):
            while True:
                top = random.randint(0, top_range)
                left = random.randint(0, left_range)
                new = (top, left, top + side, left + side)
                if any((max(r[0], new[0]) < min(r[2], new[2]) and max(r[1], new[1]) < min(r[3], new[3])) for r in regions):
                    continue
                regions.append(new)
                break
        rects.append([(r[0], r[1]) for r in regions])
    return side, rects


def random_cover_new(images_in: torch.Tensor, params: Sequence[float], rects=None) -> torch.Tensor:
    """utils.py:315-349.  params = (area fraction k, number of squares n)."""
    x = _img(images_in).clone()
    B, C, H, W = x.shape
    k, n = params[0], int(params[1])
    if rects is None:
        side, rects = pick_cover_regions(B, H, W, k, n)
    else:
        side = int((k * W * H) ** 0.5)
    if n < 1 or side < 1:
        return x
    r = torch.tensor(rects, dtype=torch.int32, device=x.device).reshape(B, n, 2).contiguous()
    check(_lib.load().nd_img_cover(ptr(x), B, C, H, W, ptr(r), n, side, _stream(x)), "nd_img_cover")
    return x


def pick_crop_corners(n_images: int, W: int, k: float):
    """The corner draws of random_crop_and_resize (utils.py:296-300): per image `left` then `top`, torch.randint on the host
    generator, in the reference's order."""
    crop = int(W * (1 - k))
    corners = []
    for _ in range(n_images):
        left = torch.randint(0, W - crop + 1, (1,)).item()
        top = torch.randint(0, W - crop + 1, (1,)).item()
        corners.append((top, left))
    return corners


def random_crop_and_resize(images_in: torch.Tensor, k: float, corners=None) -> torch.Tensor:
    """utils.py:282-312: per image a random square of side int(W * (1 - k)) (left drawn before top, torch.randint),
    resized back to (H, W) with torchvision's tensor Resize = bilinear interpolate, align_corners=False."""
    x = _img(images_in)
    B, C, H, W = x.shape
    crop = int(W * (1 - k))
    if corners is None:
        corners = pick_crop_corners(B, W, k)
    c = torch.tensor(corners, dtype=torch.int32, device=x.device).reshape(B, 2).contiguous()
    return _resize(x, H, W, crop=c, crop_size=crop)
