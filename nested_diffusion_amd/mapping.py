"""Mapping network: ViT prefix blocks + mapping MLPs -> latent conditions (guiding predictions).

Mirrors Diffusion.compute_guiding_prediction (classification_train_separately.py:330-348),
mapping/models/mlp.py::Classifier and the timm 0.4.12 vit_base_patch16_224 pieces it calls
(patch_embed, pos_drop, blocks[j], full forward).  All arithmetic runs in libnd_hip.so.

Difference from the reference that does not change results: member i's prefix blocks[0..i-1] reuse
member i-1's tokens instead of recomputing from patch_embed (15 -> 5 block evaluations; eval-mode
blocks are deterministic functions of their input).
"""
from __future__ import annotations

import os
import sys
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib, ops

LN_EPS = 1e-6  # timm 0.4.12: norm_layer = partial(nn.LayerNorm, eps=1e-6)


class Classifier:
    """mapping/models/mlp.py:4-29.  forward: reshape(-1, in_features) -> 3x (Linear, ReLU) -> Linear.
    (dropout is declared by the reference but never applied in forward.)"""

    KEYS = [f"linear{i}.{s}" for i in range(1, 5) for s in ("weight", "bias")]

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", dtype="f32"):
        missing = [k for k in self.KEYS if k not in state_dict]
        if missing:
            raise KeyError(f"Classifier state_dict is missing {missing}")
        self.device = torch.device(device)
        self.in_features = state_dict["linear1.weight"].shape[1]
        self.p = {}
        for k in self.KEYS:
            t = state_dict[k].detach().to(self.device, torch.float32).contiguous()
            # weights are repacked once into the streaming kernels' fragment order; the row-major copy is dropped
            self.p[k] = ops.PackedWeight(t, dtype) if k.endswith("weight") else t

    def __call__(self, x: torch.Tensor, dataset: str = "any") -> torch.Tensor:
        return self.forward(x, dataset)

    def forward(self, x: torch.Tensor, dataset: str = "any") -> torch.Tensor:
        x = x.reshape(-1, self.in_features)
        p = self.p
        x = ops.linear(x, p["linear1.weight"], p["linear1.bias"], act="relu")
        x = ops.linear(x, p["linear2.weight"], p["linear2.bias"], act="relu")
        x = ops.linear(x, p["linear3.weight"], p["linear3.bias"], act="relu")
        return ops.linear(x, p["linear4.weight"], p["linear4.bias"])


class VisionTransformer:
    """The subset of timm 0.4.12 VisionTransformer the path touches, over a timm state_dict."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], num_heads: int = 12, device="cuda", dtype="f32"):
        """dtype 'f16' (fp16 mode, not a reference mode): the Linear / patch-embedding weights are held in fp16 and the GEMMs and
        the attention contractions run on fp16 operands (fp32 accumulation, LayerNorm, softmax, residual stream)."""
        self.device = torch.device(device)
        self.dtype = "f16" if _lib.dtype_code(dtype) == _lib.ND_DTYPE_F16 else "f32"
        self.p = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in state_dict.items()
                  if torch.is_tensor(v) and v.is_floating_point()}
        w = self.p["patch_embed.proj.weight"]
        self.embed_dim, self.in_chans, self.patch = w.shape[0], w.shape[1], w.shape[-1]
        self.pe_w = w.reshape(self.embed_dim, -1).contiguous()
        if self.dtype == "f16":
            self.pe_w = self.pe_w.half()
            for k in list(self.p):
                if k.startswith("blocks.") and k.endswith(("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")):
                    self.p[k] = self.p[k].half()
        self.num_heads = num_heads
        self.depth = 1 + max(int(k.split(".")[1]) for k in self.p if k.startswith("blocks."))
        if self.embed_dim // num_heads != 64:
            raise ValueError("head dim must be 64 (vit_base_patch16_224: 768 / 12)")

    def patch_embed(self, x: torch.Tensor) -> torch.Tensor:
        """PatchEmbed.forward -> tokens [B*N, embed]; pos_drop is the identity in eval."""
        cols = ops.patchify(x, self.patch)
        return ops.gemm_bias_act(cols, self.pe_w, self.p["patch_embed.proj.bias"])

    def block(self, i: int, tok: torch.Tensor, B: int) -> torch.Tensor:
        """Block.forward on tokens [B*N, embed]."""
        p, pre = self.p, f"blocks.{i}."
        N = tok.shape[0] // B
        h = ops.layernorm(tok, p[pre + "norm1.weight"], p[pre + "norm1.bias"], LN_EPS)
        qkv = ops.gemm_bias_act(h, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"])
        a = ops.attention(qkv, B, N, self.num_heads, self.dtype)
        tok = ops.gemm_bias_act(a, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"], residual=tok)
        h = ops.layernorm(tok, p[pre + "norm2.weight"], p[pre + "norm2.bias"], LN_EPS)
        h = ops.gemm_bias_act(h, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], act="gelu")
        return ops.gemm_bias_act(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"], residual=tok)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Full VisionTransformer.forward (cls token + pos_embed, all blocks, norm, head on cls).
        Only feeds the never-sampled last element of compute_guiding_prediction."""
        B = x.shape[0]
        tok = self.patch_embed(x).reshape(B, -1, self.embed_dim)
        tok = torch.cat((self.p["cls_token"].expand(B, -1, -1), tok), dim=1) + self.p["pos_embed"]
        N = tok.shape[1]
        tok = tok.reshape(B * N, self.embed_dim).contiguous()
        for i in range(self.depth):
            tok = self.block(i, tok, B)
        cls = tok.reshape(B, N, self.embed_dim)[:, 0].contiguous()
        cls = ops.layernorm(cls, self.p["norm.weight"], self.p["norm.bias"], LN_EPS)
        return ops.linear(cls, self.p["head.weight"], self.p["head.bias"])

    __call__ = forward


class GuidingConditioner:
    """cond_pred_model {'vit', 'mlps'} of the reference runner (classification_train_separately.py:249-275)."""

    def __init__(self, vit: VisionTransformer, mlps: Sequence[Classifier]):
        self.vit, self.mlps = vit, list(mlps)
        self._side = None

    def compute_guiding_prediction(self, x: torch.Tensor, include_full_vit: bool = True, side_stream=None,
                                   side_work=None) -> List[torch.Tensor]:
        """classification_train_separately.py:330-348: list of K (+1) logits [B, C].

        The ViT blocks are MFMA-bound, the mapping MLPs (2.5 GB of weights each) HBM-bound and independent of
        the later blocks, so with `side_stream` the MLPs (and `side_work`, e.g. the noise estimators' encoder
        hoist) run on a second HIP stream beside the ViT; the current stream waits for them at the end."""
        B = x.shape[0]
        out: List[torch.Tensor] = [None] * len(self.mlps)
        main = torch.cuda.current_stream(x.device)
        if side_stream is not None:
            side_stream.wait_stream(main)
            if side_work is not None:
                with torch.cuda.stream(side_stream):
                    side_work()
        elif side_work is not None:
            side_work()
        tok = self.vit.patch_embed(x)
        for i in range(1, len(self.mlps) + 1):
            tok = self.vit.block(i - 1, tok, B)       # prefix shared across members
            if side_stream is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                tok.record_stream(side_stream)
                with torch.cuda.stream(side_stream):
                    side_stream.wait_event(ev)
                    out[i - 1] = self.mlps[i - 1](tok)
            else:
                out[i - 1] = self.mlps[i - 1](tok)
        if include_full_vit:
            out.append(self.vit.forward(x))
        if side_stream is not None:
            main.wait_stream(side_stream)
        return out


# ---- checkpoint readers (SURVEY section 5 'checkpoint / resume') -----------------------------------
def _to_state_dict(obj) -> Dict[str, torch.Tensor]:
    if isinstance(obj, dict):
        return obj.get("state_dict", obj)
    if hasattr(obj, "state_dict"):
        return obj.state_dict()
    raise TypeError(f"cannot extract a state_dict from {type(obj)}")


def load_pickled(path: str) -> Dict[str, torch.Tensor]:
    """Whole-module pickles as the reference writes them (mapping/train_transformer.py:166,
    mapping/train_mapping.py:160) or plain state_dicts.  torch >= 2.6 needs weights_only=False for
    module pickles; the classes must be importable (timm; mlp.py next to the checkpoints)."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        obj = torch.load(path, map_location="cpu", weights_only=False)
    return _to_state_dict(obj)


def load_conditioner(trained_path: str, dataset: str, device="cuda", num_heads: int = 12, dtype="f32") -> GuidingConditioner:
    """classification_train_separately.py:252-275: <path>/vit_base_patch16_224_<Dataset>.pth and every
    file of sorted(os.listdir(<path>/MLPs))."""
    if trained_path not in sys.path:
        sys.path.append(trained_path)                      # so the pickled `mlp.Classifier` resolves (:255)
    vit_sd = load_pickled(os.path.join(trained_path, f"vit_base_patch16_224_{dataset}.pth"))
    mlp_dir = os.path.join(trained_path, "MLPs")
    mlps = [Classifier(load_pickled(os.path.join(mlp_dir, f)), device, dtype) for f in sorted(os.listdir(mlp_dir))]
    return GuidingConditioner(VisionTransformer(vit_sd, num_heads, device, dtype), mlps)
