#!/usr/bin/env python3
"""Whole-module pickles -> plain state_dict checkpoints, keeping the reference's directory layout.

The reference saves the mapping network as pickled nn.Module OBJECTS:
    mapping/train_transformer.py:166   torch.save(model, .../vit_base_patch16_224_<Dataset>.pth)      (timm 0.4.12 ViT)
    mapping/train_mapping.py:160       torch.save(classifier, .../MLPs/<name>.pth)                    (mlp.Classifier)
and loads them with torch.load at classification_train_separately.py:257, 266, which needs the defining packages importable
(timm==0.4.12, requirements.txt:58; `mlp.py` on sys.path, :255).  nested_diffusion_amd.mapping.load_pickled does NOT need them: it
rebuilds such a pickle as a skeleton module tree and reads the tensors (mapping._SkeletonUnpickler), so the reference's files can be
pointed at directly.  This script is the optional one-off that rewrites them as plain state_dicts -- smaller to audit, loadable with
weights_only=True by anything -- on any machine, with or without the reference's environment:

    python tools/convert_checkpoints.py --src <trained_aux_cls_ckpt_path> --dst <new_dir>

Every *.pth under --src (the ViT file and MLPs/*.pth) is rewritten under --dst with the same relative path as a plain
{name: tensor} state_dict -- point `diffusion.trained_aux_cls_ckpt_path` of the YAML at --dst.  Files that already are state_dicts
(or dicts holding one under 'state_dict') are copied through unchanged in content.  The noise-estimator checkpoints (dict with key
'noise_estimator', classification_train_separately.py:1120-1126) are plain tensors already and need no conversion.
"""
from __future__ import annotations

import argparse
import os
import sys
from collections import OrderedDict

import torch


def to_state_dict(obj):
    """nn.Module -> its state_dict; {'state_dict': ...} -> that; a {name: tensor} mapping -> itself."""
    if isinstance(obj, torch.nn.Module):
        return obj.state_dict()
    if isinstance(obj, dict):
        inner = obj.get("state_dict", obj)
        if all(torch.is_tensor(v) for v in inner.values()):
            return inner
    raise TypeError(f"cannot extract a state_dict from {type(obj).__name__}")


def _load_any(src: str):
    """Plain files with the restricted unpickler; module pickles as skeleton trees (no class of theirs is imported or run)."""
    try:
        return torch.load(src, map_location="cpu", weights_only=True)
    except Exception:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from nested_diffusion_amd.mapping import load_pickled
        return load_pickled(src)


def convert_file(src: str, dst: str) -> int:
    sd = OrderedDict((k, v.detach().clone().contiguous()) for k, v in to_state_dict(_load_any(src)).items())
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    torch.save(sd, dst)
    torch.load(dst, map_location="cpu", weights_only=True)             # must be readable without any class on sys.path
    return len(sd)


def convert_tree(src_root: str, dst_root: str, sys_paths=()) -> dict:
    done = {}
    for dirpath, _, files in os.walk(src_root):
        for f in sorted(files):
            if not f.endswith((".pth", ".pt")):
                continue
            src = os.path.join(dirpath, f)
            rel = os.path.relpath(src, src_root)
            done[rel] = convert_file(src, os.path.join(dst_root, rel))
    return done


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--src", required=True, help="the reference's trained_aux_cls_ckpt_path (holds vit_base_patch16_224_*.pth and MLPs/)")
    ap.add_argument("--dst", required=True, help="output directory (same layout, plain state_dicts)")
    ap.add_argument("--sys-path", action="append", default=[], help="accepted for old command lines; no longer needed (nothing is imported)")
    a = ap.parse_args(argv)
    if os.path.abspath(a.src) == os.path.abspath(a.dst):
        raise SystemExit("--dst must differ from --src (the originals are kept)")
    done = convert_tree(a.src, a.dst, a.sys_path)
    if not done:
        raise SystemExit(f"no .pth files under {a.src}")
    for rel, n in done.items():
        print(f"{rel}: {n} tensors")
    return 0


if __name__ == "__main__":
    sys.exit(main())
