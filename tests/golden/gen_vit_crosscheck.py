#!/usr/bin/env python3
"""Cross-check fixture for the ViT prefix: the same seeded ViT-B/16 weights run through an implementation this repo did
not write -- HuggingFace `transformers` (5.15.0 here) `ViTPatchEmbeddings` / `ViTLayer` / `ViTModel` -- on the CPU.

What it is NOT: the reference's own timm 0.4.12 (requirements.txt:58; absent from this image and from the wheelhouse),
so the ViT piece of the oracle stays "parity unpinned" (DESIGN.md section 3).  What it gives: the oracle's restatement of
the published ViT block (oracle/ref_cpu.py::vit_patch_embed / vit_block / vit_full_forward) and the HIP kernels agree
with an independent implementation of the same published algorithm at the real size (768 wide, 12 heads, 196 tokens).

Two paths, as compute_guiding_prediction uses them (classification_train_separately.py:336-346):
  * mapping path: patch_embed -> blocks[0..4], NO cls token and NO pos_embed (quirk Q3); tokens after every block
  * full forward: cls + pos_embed, 12 blocks, final LayerNorm, head on the cls token
Weights: oracle.ref_cpu.init_vit_params(seed=SEED) (a seeded random initialiser: data, not arithmetic), copied tensor by
tensor into the HF modules (timm's fused qkv weight is split into HF's q/k/v projections).
Only seeds, index lists and output arrays are written (tests/golden/vit_hf.npz); fp32 results of the HF modules plus an
fp64 run of the same modules (error-growth yardstick).

Usage:  python tests/golden/gen_vit_crosscheck.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "vit_hf.npz")
sys.path.insert(0, ROOT)

from oracle import ref_cpu  # seeded synthetic-parameter initialiser only  # noqa: E402

SEED_W, SEED_X, B, HEADS, DEPTH, N_PREFIX = 11, 12, 2, 12, 12, 5
TOKENS = [0, 1, 97, 195]          # token rows kept per image (all 768 channels)


def build_hf(vp, dtype):
    from transformers import ViTConfig
    from transformers.models.vit import modeling_vit as mv
    cfg = ViTConfig(hidden_size=768, num_hidden_layers=DEPTH, num_attention_heads=HEADS, intermediate_size=3072,
                    hidden_act="gelu", layer_norm_eps=1e-6, image_size=224, patch_size=16, num_channels=3, qkv_bias=True,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg._attn_implementation = "eager"
    model = mv.ViTModel(cfg, add_pooling_layer=False).eval()
    sd = {}
    E = 768
    sd["embeddings.cls_token"] = vp["cls_token"]
    sd["embeddings.position_embeddings"] = vp["pos_embed"]
    sd["embeddings.patch_embeddings.projection.weight"] = vp["patch_embed.proj.weight"]
    sd["embeddings.patch_embeddings.projection.bias"] = vp["patch_embed.proj.bias"]
    for i in range(DEPTH):
        s, d = f"blocks.{i}.", f"layers.{i}."
        qkv_w, qkv_b = vp[s + "attn.qkv.weight"], vp[s + "attn.qkv.bias"]     # timm: rows [q | k | v]
        for j, n in enumerate("qkv"):
            sd[d + f"attention.{n}_proj.weight"] = qkv_w[j * E:(j + 1) * E]
            sd[d + f"attention.{n}_proj.bias"] = qkv_b[j * E:(j + 1) * E]
        sd[d + "attention.o_proj.weight"] = vp[s + "attn.proj.weight"]
        sd[d + "attention.o_proj.bias"] = vp[s + "attn.proj.bias"]
        sd[d + "layernorm_before.weight"] = vp[s + "norm1.weight"]
        sd[d + "layernorm_before.bias"] = vp[s + "norm1.bias"]
        sd[d + "layernorm_after.weight"] = vp[s + "norm2.weight"]
        sd[d + "layernorm_after.bias"] = vp[s + "norm2.bias"]
        for n in ("fc1", "fc2"):
            sd[d + f"mlp.{n}.weight"] = vp[s + f"mlp.{n}.weight"]
            sd[d + f"mlp.{n}.bias"] = vp[s + f"mlp.{n}.bias"]
    sd["layernorm.weight"] = vp["norm.weight"]
    sd["layernorm.bias"] = vp["norm.bias"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("mask_token" in m or "pooler" in m for m in missing), missing
    return model.to(dtype)


@torch.no_grad()
def run(model, vp, x, dtype):
    x = x.to(dtype)
    # mapping path: no cls / pos (classification_train_separately.py:337-340)
    tok = model.embeddings.patch_embeddings(x)
    prefix = []
    for i in range(N_PREFIX):
        tok = model.layers[i](tok)
        prefix.append(tok[:, TOKENS].clone())
    # full forward (:346): HF ViTModel = embeddings (cls + pos) -> 12 layers -> final LayerNorm; timm's head on cls
    hs = model(pixel_values=x).last_hidden_state
    logits = torch.nn.functional.linear(hs[:, 0], vp["head.weight"].to(dtype), vp["head.bias"].to(dtype))
    return torch.stack(prefix), logits            # [5, B, len(TOKENS), 768], [B, 2]


def main():
    import transformers
    vp = ref_cpu.init_vit_params(seed=SEED_W)                 # ViT-B/16 defaults: 768 / 12 blocks / patch 16 / 224
    x = torch.rand(B, 3, 224, 224, generator=torch.Generator().manual_seed(SEED_X))
    p32, l32 = run(build_hf(vp, torch.float32), vp, x, torch.float32)
    p64, l64 = run(build_hf(vp, torch.float64), vp, x, torch.float64)
    print("fp32 vs fp64 HF: prefix max rel", float(((p32.double() - p64).abs().amax(dim=(1, 2, 3)) / p64.abs().amax(dim=(1, 2, 3))).max()),
          " logits max abs", float((l32.double() - l64).abs().max()))
    np.savez_compressed(OUT, seed_w=SEED_W, seed_x=SEED_X, batch=B, heads=HEADS, depth=DEPTH, tokens=np.array(TOKENS),
                        prefix_f32=p32.numpy(), prefix_f64=p64.numpy(), full_logits_f32=l32.numpy(), full_logits_f64=l64.numpy(),
                        transformers_version=transformers.__version__)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
