"""Mapping network: ViT prefix blocks + mapping MLPs -> latent conditions (guiding predictions).

Mirrors Diffusion.compute_guiding_prediction (classification_train_separately.py:330-348),
mapping/models/mlp.py::Classifier and the timm 0.4.12 vit_base_patch16_224 pieces it calls
(patch_embed, pos_drop, blocks[j], full forward).  All arithmetic runs in libnd_hip.so.

Difference from the reference that does not change results: member i's prefix blocks[0..i-1] reuse
member i-1's tokens instead of recomputing from patch_embed (15 -> 5 block evaluations; eval-mode
blocks are deterministic functions of their input).
"""
from __future__ import annotations

import _compat_pickle
import copyreg
import ctypes as C
import functools
import os
import pickle
import sys
import types
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops
from ._lib import check, ptr

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
        gemm_keys = [k for k in self.p if k.startswith("blocks.") and
                     k.endswith(("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"))]
        # fp32 mode: the Linear layers run on the bf16 matrix pipe with exact fp32 products (nd_gemm_split) wherever the shapes allow
        # it (every GEMM depth a multiple of 32: 768 / 3072 / 3*16*16 for ViT-B/16); the weights are then held as frag32b3 images
        # ONLY.  ND_GEMM_F32=mfma_f32 keeps the f32-input-MFMA kernels (and row-major fp32 weights).
        self.split = (self.dtype == "f32" and os.environ.get("ND_GEMM_F32", "b9") != "mfma_f32" and self.pe_w.shape[1] % 32 == 0
                      and all(self.p[k].shape[1] % 32 == 0 for k in gemm_keys))
        if self.split:
            self.pe_w = ops.split_rows(self.pe_w)
            for k in gemm_keys:
                self.p[k] = ops.split_rows(self.p[k])
        if self.dtype == "f16":
            self.pe_w = self.pe_w.half()
            for k in gemm_keys:
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
        if self.split and ops.qkv_images_supported(N, self.num_heads) and not os.environ.get("ND_ATT_F32"):
            # attention on the bf16 matrix pipe as well: the qkv Linear writes the attention kernel's operand images (nd_vit_block's sequence)
            img = ops.gemm_split_qkv(ops.split_rows(h), p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"], B, N, self.num_heads)
            a = ops.attention_images(img, B, N, self.num_heads)
        else:
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
    """cond_pred_model {'vit', 'mlps'} of the reference runner (classification_train_separately.py:249-275).

    compute_guiding_prediction is ONE call into the library (nd_guiding_prediction): patch_embed, the shared prefix blocks and
    every mapping MLP are sequenced in C on the current stream.  The handle (`nd_cond`) holds pointers into the tensors of
    `vit` and `mlps` (kept alive here) and one workspace for the activations."""

    def __init__(self, vit: VisionTransformer, mlps: Sequence[Classifier]):
        self.vit, self.mlps = vit, list(mlps)
        if not self.mlps:
            raise ValueError("at least one mapping MLP is required")
        if len(self.mlps) > vit.depth:
            raise ValueError(f"{len(self.mlps)} mapping MLPs need as many ViT blocks, the ViT has {vit.depth}")
        self._h = None
        self._key = None
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().nd_cond_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def handle(self, B: int, img_size: int):
        """nd_cond for batches of up to B images of img_size x img_size (created on first use, re-created when either grows)."""
        if self._h is not None and self._key[0] >= B and self._key[1] == img_size:
            return self._h
        lib = _lib.load()
        vit, m0 = self.vit, self.mlps[0]
        if img_size % vit.patch:
            raise ValueError(f"image size {img_size} is not a multiple of the patch size {vit.patch}")
        n_tok = (img_size // vit.patch) ** 2
        widths = [m0.p[f"linear{i}.weight"].N for i in (1, 2, 3)]
        n_cls = m0.p["linear4.weight"].N
        dt = _lib.dtype_code(vit.dtype)
        for m in self.mlps:
            if [m.p[f"linear{i}.weight"].N for i in (1, 2, 3)] != widths or m.p["linear4.weight"].N != n_cls \
                    or m.in_features != n_tok * vit.embed_dim or m.p["linear1.weight"].dtype != dt:
                raise ValueError("mapping MLPs must share one shape / dtype and read all tokens of a ViT block")
        cfg = _lib.NdCondConfig()
        cfg.img_size, cfg.patch, cfg.in_chans, cfg.embed_dim, cfg.num_heads = img_size, vit.patch, vit.in_chans, vit.embed_dim, vit.num_heads
        cfg.mlp_hidden = vit.p["blocks.0.mlp.fc1.weight"].shape[0]
        cfg.n_blocks, cfg.n_mlps = vit.depth, len(self.mlps)
        cfg.mlp_widths[0], cfg.mlp_widths[1], cfg.mlp_widths[2] = widths
        cfg.num_classes, cfg.max_batch, cfg.max_tokens, cfg.ln_eps = n_cls, max(B, 1), n_tok + 1, LN_EPS
        cfg.operand_dtype = _lib.ND_DTYPE_F32_SPLIT if vit.split else dt

        def dptr(v):
            return v.data.data_ptr() if isinstance(v, ops.SplitMatrix) else v.data_ptr()
        h = C.c_void_p()
        check(lib.nd_cond_create(C.byref(cfg), C.byref(h)), "nd_cond_create")
        nbytes = lib.nd_cond_workspace_bytes(C.byref(cfg))
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=vit.device)
        check(lib.nd_cond_bind_workspace(h, (ws.data_ptr() + 255) & ~255, nbytes), "nd_cond_bind_workspace")
        pe = _lib.NdPatchEmbedWeights(dptr(vit.pe_w), vit.p["patch_embed.proj.bias"].data_ptr())
        check(lib.nd_cond_set_patch_embed(h, C.byref(pe)), "nd_cond_set_patch_embed")
        for i in range(vit.depth):
            w = _lib.NdVitBlockWeights()
            for field, key in _lib.VIT_BLOCK_FIELDS:
                setattr(w, field, dptr(vit.p[f"blocks.{i}.{key}"]))
            check(lib.nd_cond_set_block(h, i, C.byref(w)), "nd_cond_set_block")
        for i, m in enumerate(self.mlps):
            w = _lib.NdMlpWeights()
            for l in range(4):
                w.w_packed[l] = m.p[f"linear{l + 1}.weight"].data.data_ptr()
                w.bias[l] = m.p[f"linear{l + 1}.bias"].data_ptr()
            check(lib.nd_cond_set_mlp(h, i, C.byref(w)), "nd_cond_set_mlp")
        if self._h is not None:
            lib.nd_cond_destroy(self._h)
        self._h, self._key, self._ws = h, (max(B, 1), img_size), ws
        return h

    def compute_guiding_prediction(self, x: torch.Tensor, include_full_vit: bool = True) -> List[torch.Tensor]:
        """classification_train_separately.py:330-348: list of K (+1) logits [B, C]."""
        x = ops._f32(x, "x")
        B, K = x.shape[0], len(self.mlps)
        h = self.handle(B, x.shape[-1])
        n_cls = self.mlps[0].p["linear4.weight"].N
        logits = torch.empty(K, B, n_cls, dtype=torch.float32, device=x.device)
        check(_lib.load().nd_guiding_prediction(h, ptr(x), ptr(logits), None, B, torch.cuda.current_stream(x.device).cuda_stream),
              "nd_guiding_prediction")
        out = list(logits.unbind(0))
        if include_full_vit:
            out.append(self.vit.forward(x))          # the never-sampled (K+1)-th element (:346, quirk Q1)
        return out

    def compute_guiding_prediction_py(self, x: torch.Tensor, include_full_vit: bool = True, side_stream=None,
                                      side_work=None) -> List[torch.Tensor]:
        """The same sequence launched operator by operator from Python (tests: must equal the C-level call bit for bit; tools:
        the two-stream experiment of DESIGN 7b-6 -- with `side_stream` the HBM-bound mapping MLPs and `side_work` run beside
        the MFMA-bound ViT blocks; measured slower, not used by the product path)."""
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
def _module_tree_state_dict(mod, prefix: str = "", out=None) -> Dict[str, torch.Tensor]:
    """state_dict of a rebuilt module tree, read from the pickled attribute dicts themselves (`_parameters`, `_buffers` minus the
    non-persistent ones, `_modules`): what nn.Module.state_dict() returns, without relying on hook tables that a module pickled by
    torch 1.10 (requirements.txt:59) does not carry."""
    out = OrderedDict() if out is None else out
    d = mod.__dict__
    for k, v in (d.get("_parameters") or {}).items():
        if v is not None:
            out[prefix + k] = v.detach()
    skip = d.get("_non_persistent_buffers_set") or ()
    for k, v in (d.get("_buffers") or {}).items():
        if v is not None and k not in skip:
            out[prefix + k] = v.detach()
    for k, sub in (d.get("_modules") or {}).items():
        if sub is not None:
            _module_tree_state_dict(sub, prefix + k + ".", out)
    return out


def _to_state_dict(obj) -> Dict[str, torch.Tensor]:
    if isinstance(obj, dict):
        return obj.get("state_dict", obj)
    if isinstance(obj, torch.nn.Module):
        return _module_tree_state_dict(obj)
    raise TypeError(f"cannot extract a state_dict from {type(obj)}")


class _SkeletonUnpickler(pickle.Unpickler):
    """Rebuilds a pickled nn.Module TREE without the classes that defined it.  The reference saves the mapping network as whole
    module objects (mapping/train_transformer.py:166: the timm 0.4.12 ViT; mapping/train_mapping.py:160: mlp.Classifier) and reads
    them back with torch.load at classification_train_separately.py:255-269, which needs `timm` and `mlp.py` importable.  Only the
    tensors matter here, so:
      * torch's own weights_only allow-list (tensor / storage / parameter rebuilders, OrderedDict, dtypes ...) resolves as usual;
      * `copyreg._reconstructor` + `builtins.object` (how protocol-2 pickles rebuild any plain object) and `functools.partial`
        (an inert constructor) are let through;
      * a class under torch.nn.modules (Linear, LayerNorm, Conv2d, Sequential, GELU ...) and any global of a THIRD-PARTY or missing
        module (timm.*, mlp.*, models.* ...) becomes a bare nn.Module subclass of the same name, WITHOUT importing anything: its
        instances are inert containers of `_parameters` / `_buffers` / `_modules`, built by object.__new__ + a __dict__ update.  The
        stock pickle machine leaves REDUCE unrestricted for every global it resolves, so the REAL torch.nn classes are not handed
        out: a crafted file could otherwise call e.g. Linear(10**9, 10**9) (an allocation); a stub's constructor takes no arguments;
      * everything else -- a global of the standard library, of builtins, of torch or numpy that is not on the allow-list
        (os.system, builtins.eval, torch.hub.load ...) -- is refused, as the weights_only unpickler refuses it."""
    _PASS = {("copyreg", "_reconstructor"): copyreg._reconstructor, ("builtins", "object"): object,
             ("functools", "partial"): functools.partial}
    _stubs: Dict[Tuple[str, str], type] = {}

    _allow_list: Optional[Dict[str, object]] = None          # built once per process, not per global

    @classmethod
    def _allowed(cls):
        if cls._allow_list is None:
            try:
                from torch._weights_only_unpickler import _get_allowed_globals
                cls._allow_list = dict(_get_allowed_globals())
            except Exception:                                # private API moved: fall back to the few globals a module pickle needs
                import collections
                cls._allow_list = {"collections.OrderedDict": collections.OrderedDict, "torch._utils._rebuild_tensor_v2": torch._utils._rebuild_tensor_v2,
                                   "torch._utils._rebuild_parameter": torch._utils._rebuild_parameter, "torch.FloatStorage": torch.FloatStorage,
                                   "torch.LongStorage": torch.LongStorage, "torch.nn.parameter.Parameter": torch.nn.Parameter,
                                   "builtins.set": set, "torch.Size": torch.Size}
        return cls._allow_list

    def find_class(self, module, name):
        # protocol < 3 pickles (what torch.save writes) carry Python-2 names: __builtin__.set, copy_reg._reconstructor ...
        if (module, name) in _compat_pickle.NAME_MAPPING:
            module, name = _compat_pickle.NAME_MAPPING[(module, name)]
        elif module in _compat_pickle.IMPORT_MAPPING:
            module = _compat_pickle.IMPORT_MAPPING[module]
        allowed = self._allowed()
        full = f"{module}.{name}"
        if full in allowed:
            return allowed[full]
        if (module, name) in self._PASS:
            return self._PASS[(module, name)]
        top = module.split(".")[0]
        is_nn_class = False
        if module.startswith("torch.nn.modules"):
            cls = getattr(sys.modules.get(module), name, None)
            is_nn_class = isinstance(cls, type) and issubclass(cls, torch.nn.Module)      # looked up to classify, never returned
        if not is_nn_class and (top in sys.stdlib_module_names or top in ("builtins", "torch", "numpy", "nested_diffusion_amd")):
            raise pickle.UnpicklingError(f"refusing global {full}: not needed to rebuild a module tree (only tensors are read; "
                                         f"see nested_diffusion_amd.mapping._SkeletonUnpickler)")
        key = (module, name)
        if key not in self._stubs:
            self._stubs[key] = type(name, (torch.nn.Module,), {"__module__": module, "__doc__": f"skeleton of {full} (state only)"})
        return self._stubs[key]


def _skeleton_pickle_module():
    """A `pickle_module` for torch.load whose Unpickler is _SkeletonUnpickler."""
    m = types.ModuleType("nd_skeleton_pickle")
    for k in dir(pickle):
        if not k.startswith("__"):
            setattr(m, k, getattr(pickle, k))
    m.Unpickler = _SkeletonUnpickler
    m.load = lambda f, **kw: _SkeletonUnpickler(f, **kw).load()
    return m


def load_checkpoint_object(path: str):
    """torch.load with the restricted unpickler, falling back to the skeleton unpickler (never to arbitrary-code unpickling)."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_skeleton_pickle_module())


def load_pickled(path: str) -> Dict[str, torch.Tensor]:
    """Whole-module pickles as the reference writes them (mapping/train_transformer.py:166, mapping/train_mapping.py:160) or plain
    state_dicts (also under a 'state_dict' key).  Plain files load with weights_only=True; module pickles are rebuilt as skeleton
    trees by _SkeletonUnpickler and their tensors collected -- neither `timm` nor `mlp.py` has to be importable (the reference
    needs both, :255-269), so the reference's own checkpoint files open on a box that has only this package."""
    return _to_state_dict(load_checkpoint_object(path))


def load_conditioner(trained_path: str, dataset: str, device="cuda", num_heads: Optional[int] = None, dtype="f32") -> GuidingConditioner:
    """classification_train_separately.py:252-275: <path>/vit_base_patch16_224_<Dataset>.pth and every
    file of sorted(os.listdir(<path>/MLPs))."""
    vit_sd = load_pickled(os.path.join(trained_path, f"vit_base_patch16_224_{dataset}.pth"))
    mlp_dir = os.path.join(trained_path, "MLPs")
    mlps = [Classifier(load_pickled(os.path.join(mlp_dir, f)), device, dtype) for f in sorted(os.listdir(mlp_dir))]
    if num_heads is None:
        # the head count is not in a state_dict; the attention kernels take head dim 64 only (vit_base_patch16_224: 768 / 12), so it
        # follows from the embedding width
        num_heads = max(1, vit_sd["patch_embed.proj.weight"].shape[0] // 64)
    return GuidingConditioner(VisionTransformer(vit_sd, num_heads, device, dtype), mlps)
