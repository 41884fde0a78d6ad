"""Synthetic weights / inputs of the reference's shapes, generated directly on the GPU (no datasets
or checkpoints exist offline).  Distributions follow SURVEY 8d: nn.Linear default init, embed ~ U(0,1)
(latent_model.py:99), BatchNorm running stats randomised so the eval-BN fold is exercised."""
from __future__ import annotations

import math
from typing import Dict

import torch


def cond_model_state(data_dim: int, hidden: int, feature: int, y_dim: int, n_steps: int, seed: int, device="cuda",
                     denoiser: bool = False) -> Dict[str, torch.Tensor]:
    """denoiser=True: see make_denoiser (long schedules, T = 1000: keeps the synthetic chain O(1) like a trained model's)."""
    g = torch.Generator(device=device).manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}

    def lin(name, n_in, n_out):
        bound = 1.0 / math.sqrt(n_in)
        p[name + ".weight"] = (torch.rand(n_out, n_in, generator=g, device=device) * 2 - 1) * bound
        p[name + ".bias"] = (torch.rand(n_out, generator=g, device=device) * 2 - 1) * bound

    def bn(name, n):
        p[name + ".weight"] = torch.rand(n, generator=g, device=device) + 0.5
        p[name + ".bias"] = torch.randn(n, generator=g, device=device) * 0.5
        p[name + ".running_mean"] = torch.randn(n, generator=g, device=device) * 0.5
        p[name + ".running_var"] = torch.rand(n, generator=g, device=device) * 1.5 + 0.5
        p[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long, device=device)

    lin("encoder_x.0", data_dim, hidden); bn("encoder_x.1", hidden)
    lin("encoder_x.3", hidden, hidden); bn("encoder_x.4", hidden)
    lin("encoder_x.6", hidden, feature); bn("norm", feature)
    lin("lin1.lin", 2 * y_dim, feature)
    p["lin1.embed.weight"] = torch.rand(n_steps + 1, feature, generator=g, device=device)
    bn("unetnorm1", feature)
    for name in ("lin2", "lin3"):
        lin(name + ".lin", feature, feature)
        p[name + ".embed.weight"] = torch.rand(n_steps + 1, feature, generator=g, device=device)
        bn("unetnorm" + name[-1], feature)
    lin("lin4", feature, y_dim)
    if denoiser:
        make_denoiser(p, y_dim, n_steps)
    return p


def make_denoiser(p: Dict[str, torch.Tensor], y_dim: int, n_steps: int, gain: float = 0.7, beta_start: float = 1e-4,
                  beta_end: float = 0.02) -> None:
    """Give a random ConditionalModel state_dict (in place) the behaviour of a TRAINED noise estimator:
    eps_theta(y_t, t) = gain * (y_t - yhat) / sqrt(1 - abar_t) + (what the random features contribute).  A random-weight eps_theta
    ignores the noise it should predict, so its reverse chain multiplies y_T by 1/sqrt(abar_T) (~160 at T = 1000, linear
    schedule) and the samples leave the range where convert_to_prob means anything; with this signal path the chain is
    contractive and |y_t| stays O(1).  The first 2C features of each layer are +/- pairs: lin1 emits +/-(y - yhat) with the
    per-timestep gain 1/sqrt(1 - abar_t) in its embedding rows (latent_model.py:101-105), lin2 / lin3 / lin4 take pair
    differences -- exactly linear since softplus(p) - softplus(-p) = p -- and xe is 1 on those features.  All other features
    stay random.  Same construction as the oracle's initialiser that golden fixture `sampler_s4` was generated with."""
    from .diffusion_utils import make_beta_schedule
    C, S = y_dim, 2 * y_dim
    dev = p["lin4.weight"].device
    betas = make_beta_schedule(schedule="linear", num_timesteps=n_steps, start=beta_start, end=beta_end).float()
    omabs = torch.sqrt(1 - (1.0 - betas).cumprod(dim=0)).to(dev)

    def bn_identity(name, shift=0.0):
        p[name + ".weight"][:S] = 1.0
        p[name + ".bias"][:S] = shift
        p[name + ".running_mean"][:S] = 0.0
        p[name + ".running_var"][:S] = 1.0
    p["encoder_x.6.weight"][:S] = 0.0
    p["encoder_x.6.bias"][:S] = 0.0
    bn_identity("norm", 1.0)
    eye = torch.eye(C, device=dev)
    w1 = torch.cat([eye, -eye], dim=1)                                   # row c: y_c - yhat_c
    p["lin1.lin.weight"][:S] = torch.cat([w1, -w1], dim=0)
    p["lin1.lin.bias"][:S] = 0.0
    p["lin1.embed.weight"][:n_steps, :S] = (1.0 / omabs)[:, None]
    p["lin1.embed.weight"][n_steps, :S] = 1.0
    bn_identity("unetnorm1")
    for name in ("lin2", "lin3"):
        w = p[name + ".lin.weight"]
        w[:S] = 0.0
        w[:S, :S] = torch.cat([w1, -w1], dim=0)                          # [[I, -I], [-I, I]]
        p[name + ".lin.bias"][:S] = 0.0
        p[name + ".embed.weight"][:, :S] = 1.0
        bn_identity("unetnorm" + name[-1])
    p["lin4.weight"][:, :S] = gain * w1


def classifier_state(in_features: int, seed: int, widths=(4096, 2048, 128), num_classes: int = 2, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    dims = [in_features, *widths, num_classes]
    p = {}
    for i in range(4):
        bound = 1.0 / math.sqrt(dims[i])
        p[f"linear{i + 1}.weight"] = (torch.rand(dims[i + 1], dims[i], generator=g, device=device) * 2 - 1) * bound
        p[f"linear{i + 1}.bias"] = (torch.rand(dims[i + 1], generator=g, device=device) * 2 - 1) * bound
    return p


def vit_state(seed: int, embed: int = 768, depth: int = 12, mlp_ratio: int = 4, patch: int = 16, in_chans: int = 3,
              img: int = 224, num_classes: int = 2, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    n_tok = (img // patch) ** 2
    vp = {}

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, device=device) * std

    def lin(name, n_in, n_out):
        vp[name + ".weight"] = rn(n_out, n_in, std=1.0 / math.sqrt(n_in))
        vp[name + ".bias"] = rn(n_out, std=0.02)

    def ln(name):
        vp[name + ".weight"] = 1.0 + rn(embed, std=0.1)
        vp[name + ".bias"] = rn(embed, std=0.05)

    vp["cls_token"] = rn(1, 1, embed, std=0.02)
    vp["pos_embed"] = rn(1, n_tok + 1, embed, std=0.02)
    vp["patch_embed.proj.weight"] = rn(embed, in_chans, patch, patch, std=1.0 / math.sqrt(in_chans * patch * patch))
    vp["patch_embed.proj.bias"] = rn(embed, std=0.02)
    for i in range(depth):
        pre = f"blocks.{i}."
        ln(pre + "norm1"); lin(pre + "attn.qkv", embed, 3 * embed); lin(pre + "attn.proj", embed, embed)
        ln(pre + "norm2"); lin(pre + "mlp.fc1", embed, mlp_ratio * embed); lin(pre + "mlp.fc2", mlp_ratio * embed, embed)
    ln("norm")
    lin("head", embed, num_classes)
    return vp


def images(batch: int, seed: int = 1234, device="cuda", chans: int = 3, size: int = 224) -> torch.Tensor:
    """U[0,1) like ToTensor output (dataset_helper/chest_x_ray_dataset.py:31-51)."""
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.rand(batch, chans, size, size, generator=g, device=device)
