"""Drop-in mirror of the reference's diffusion/diffusion_utils.py for the inference path:
same function names, argument order and return conventions; the arithmetic runs in libnd_hip.so.

Training-only functions of the reference module (q_sample :39-50, y_0_reparam :114-130) are out
of scope.
"""
from __future__ import annotations

import math
from typing import List, Optional, Union

import torch


def _cosine_alpha_bar(u: float, s: float = 0.008) -> float:
    return math.cos((u + s) / (1 + s) * math.pi / 2) ** 2


def make_beta_schedule(schedule="linear", num_timesteps=1000, start=1e-5, end=1e-2):
    """Noise schedule beta_1..beta_T (reference: diffusion_utils.py:5-28; same names, same fp32 torch ops per branch, so
    the tables are bit-identical -- pinned by tests/golden/schedule.npz).  Init-time host table of T floats."""
    T = num_timesteps
    if schedule == "linear":
        return torch.linspace(start, end, T)
    if schedule == "const":
        return end * torch.ones(T)
    if schedule == "quad":
        return torch.linspace(start ** 0.5, end ** 0.5, T) ** 2
    if schedule == "jsd":
        return 1.0 / torch.linspace(T, 1, T)
    if schedule == "sigmoid":
        return torch.sigmoid(torch.linspace(-6, 6, T)) * (end - start) + start
    if schedule in ("cosine", "cosine_reverse"):
        # beta_i = min(1 - abar((i+1)/T) / abar(i/T), 0.999), abar(u) = cos^2(((u + 0.008) / 1.008) * pi / 2)
        return torch.tensor([min(1 - _cosine_alpha_bar((i + 1) / T) / _cosine_alpha_bar(i / T), 0.999) for i in range(T)])
    if schedule == "cosine_anneal":
        return torch.tensor([start + 0.5 * (end - start) * (1 - math.cos(t / (T - 1) * math.pi)) for t in range(T)])
    raise ValueError(f"unknown beta schedule '{schedule}'")


def extract(input, t, x):
    """diffusion_utils.py:31-35 (table lookup reshaped for broadcasting; host plumbing)."""
    shape = x.shape
    out = torch.gather(input, 0, t.to(input.device))
    reshape = [t.shape[0]] + [1] * (len(shape) - 1)
    return out.reshape(*reshape)


def draw_reference_noise(n_steps: int, like: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The T draws of p_sample_loop in the reference's order (:139 then :67 for t = T-1 .. 1), made on
    the CPU generator exactly as the reference's CPU path makes them: n_steps successive
    randn of like.shape.  With torch.manual_seed(s) before both, the noise is identical to what the
    reference consumes on CPU -- this is how parity runs are driven."""
    shape = tuple(like.shape)
    return torch.stack([torch.randn(shape, generator=generator) for _ in range(n_steps)])


def _engine_of(model):
    eng = getattr(model, "hip_engine", None)
    if eng is None:
        raise TypeError("model must be a nested_diffusion_amd.latent_model.ConditionalModel (HIP-backed)")
    return eng()


def p_sample(model, x, y, y_0_hat, y_T_mean, t, alphas, one_minus_alphas_bar_sqrt, output_detach=True, z=None):
    """Reverse step y_t -> y_{t-1} (diffusion_utils.py:54-92).  `z` (optional) replaces the internal
    torch.randn_like(y) draw."""
    eng = _engine_of(model)
    model.encode(x)
    eng.set_schedule(alphas, one_minus_alphas_bar_sqrt)
    if z is None:
        z = torch.randn_like(y)
    return eng.p_sample(0, y, y_0_hat, y_T_mean, int(t), z)


def p_sample_t_1to0(model, x, y, y_0_hat, y_T_mean, one_minus_alphas_bar_sqrt, output_detach=True):
    """diffusion_utils.py:96-111."""
    eng = _engine_of(model)
    model.encode(x)
    if eng._sched is None or eng._sched[1].numel() != one_minus_alphas_bar_sqrt.numel():
        eng.set_schedule(torch.ones_like(one_minus_alphas_bar_sqrt), one_minus_alphas_bar_sqrt)
    return eng.p_sample(0, y, y_0_hat, y_T_mean, 0, None)


def p_sample_loop(model, x, y_0_hat, y_T_mean, n_steps, alphas, one_minus_alphas_bar_sqrt, only_last_sample=False,
                  input_model_original_version=True, output_detach=True, noise: Optional[torch.Tensor] = None
                  ) -> Union[torch.Tensor, List[torch.Tensor]]:
    """diffusion_utils.py:133-163.  Returns y_0 [B, C] when only_last_sample, else the list
    [y_T, y_{T-1}, ..., y_1, y_0] of n_steps + 1 tensors.

    noise: optional [n_steps, B, C] draws in the reference order (draw_reference_noise); when None the
    n_steps draws are made on y_T_mean's device with one torch.randn call."""
    if not input_model_original_version:
        model = model.conditional_model
    eng = _engine_of(model)
    B, C = y_T_mean.shape
    if noise is None:
        noise = torch.randn((n_steps, B, C), device=y_T_mean.device, dtype=torch.float32)
    elif tuple(noise.shape) != (n_steps, B, C):
        raise ValueError(f"noise must be [{n_steps}, {B}, {C}]")
    model.encode(x)
    eng.set_schedule(alphas, one_minus_alphas_bar_sqrt)
    dev = eng.device
    out = eng.sample(y_0_hat.to(dev)[None], y_T_mean.to(dev)[None], noise.to(dev)[None], member0=0, n_members=1, mc=1,
                     T=n_steps, return_seq=not only_last_sample)
    if only_last_sample:
        return out[0]
    seq = list(out[0].unbind(0))
    assert len(seq) == n_steps + 1
    return seq
