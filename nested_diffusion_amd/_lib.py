"""ctypes binding of libnd_hip.so (include/nested_diffusion.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, every
operator raises.  `load()` is lazy so that host-only logic (CLI parsing, sharding arithmetic,
checkpoint readers) stays importable on machines without the library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ND_LIB_PATH") or os.path.join(HERE, "libnd_hip.so")   # ND_LIB_PATH: a debug build (tools/wg_times.py)

ND_ACT_NONE, ND_ACT_SOFTPLUS, ND_ACT_RELU, ND_ACT_GELU = 0, 1, 2, 3
ND_DTYPE_F32, ND_DTYPE_F16 = 0, 1
ND_DTYPE_F32_SPLIT = 2      # nd_cond_config only: fp32 arithmetic, ViT Linear layers on the bf16 pipe (weights as frag32b3 images)


def dtype_code(dtype) -> int:
    """'f32' | 'f16' (also 'fp32'/'fp16', torch.float32/float16) -> ND_DTYPE_*."""
    name = str(dtype).replace("torch.", "").lower()
    if name in ("f32", "fp32", "float32", "0"):
        return ND_DTYPE_F32
    if name in ("f16", "fp16", "float16", "half", "1"):
        return ND_DTYPE_F16
    raise ValueError(f"operand dtype must be 'f32' or 'f16', got {dtype!r}")

# field order MUST match nd_member_weights in include/nested_diffusion.h
MEMBER_WEIGHT_FIELDS = [
    ("enc0_w", "encoder_x.0.weight"), ("enc0_b", "encoder_x.0.bias"),
    ("bn0_w", "encoder_x.1.weight"), ("bn0_b", "encoder_x.1.bias"),
    ("bn0_mean", "encoder_x.1.running_mean"), ("bn0_var", "encoder_x.1.running_var"),
    ("enc3_w", "encoder_x.3.weight"), ("enc3_b", "encoder_x.3.bias"),
    ("bn1_w", "encoder_x.4.weight"), ("bn1_b", "encoder_x.4.bias"),
    ("bn1_mean", "encoder_x.4.running_mean"), ("bn1_var", "encoder_x.4.running_var"),
    ("enc6_w", "encoder_x.6.weight"), ("enc6_b", "encoder_x.6.bias"),
    ("norm_w", "norm.weight"), ("norm_b", "norm.bias"),
    ("norm_mean", "norm.running_mean"), ("norm_var", "norm.running_var"),
    ("lin1_w", "lin1.lin.weight"), ("lin1_b", "lin1.lin.bias"), ("emb1", "lin1.embed.weight"),
    ("un1_w", "unetnorm1.weight"), ("un1_b", "unetnorm1.bias"),
    ("un1_mean", "unetnorm1.running_mean"), ("un1_var", "unetnorm1.running_var"),
    ("lin2_w", "lin2.lin.weight"), ("lin2_b", "lin2.lin.bias"), ("emb2", "lin2.embed.weight"),
    ("un2_w", "unetnorm2.weight"), ("un2_b", "unetnorm2.bias"),
    ("un2_mean", "unetnorm2.running_mean"), ("un2_var", "unetnorm2.running_var"),
    ("lin3_w", "lin3.lin.weight"), ("lin3_b", "lin3.lin.bias"), ("emb3", "lin3.embed.weight"),
    ("un3_w", "unetnorm3.weight"), ("un3_b", "unetnorm3.bias"),
    ("un3_mean", "unetnorm3.running_mean"), ("un3_var", "unetnorm3.running_var"),
    ("lin4_w", "lin4.weight"), ("lin4_b", "lin4.bias"),
]


class NdConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("y_dim", "data_dim", "hidden_dim", "feature_dim", "n_steps",
                                          "n_members", "max_batch", "max_rows", "operand_dtype")]


class NdMemberWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in MEMBER_WEIGHT_FIELDS]


class NdCondConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("img_size", "patch", "in_chans", "embed_dim", "num_heads", "mlp_hidden", "n_blocks", "n_mlps")] + \
               [("mlp_widths", C.c_int32 * 3)] + \
               [(n, C.c_int32) for n in ("num_classes", "max_batch", "max_tokens", "operand_dtype")] + [("ln_eps", C.c_float)]


class NdPatchEmbedWeights(C.Structure):
    _fields_ = [("proj_w", C.c_void_p), ("proj_b", C.c_void_p)]


VIT_BLOCK_FIELDS = [("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("qkv_w", "attn.qkv.weight"), ("qkv_b", "attn.qkv.bias"),
                    ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"), ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"),
                    ("fc1_w", "mlp.fc1.weight"), ("fc1_b", "mlp.fc1.bias"), ("fc2_w", "mlp.fc2.weight"), ("fc2_b", "mlp.fc2.bias")]


class NdVitBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in VIT_BLOCK_FIELDS]


class NdMlpWeights(C.Structure):
    _fields_ = [("w_packed", C.c_void_p * 4), ("bias", C.c_void_p * 4)]


class NdBatchOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("samples", "prob", "vote", "probs", "yhat")]


# every symbol include/nested_diffusion.h declares: name -> (restype, argtypes)
_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
SIGNATURES = {
    "nd_last_error": (C.c_char_p, []),
    "nd_version": (C.c_char_p, []),
    "nd_create": (_i, [C.POINTER(NdConfig), C.POINTER(_vp)]),
    "nd_destroy": (_i, [_vp]),
    "nd_workspace_bytes": (_sz, [C.POINTER(NdConfig)]),
    "nd_bind_workspace": (_i, [_vp, _vp, _sz]),
    "nd_load_member": (_i, [_vp, _i, C.POINTER(NdMemberWeights), _vp]),
    "nd_set_schedule": (_i, [_vp, _vp, _vp, _i, _vp]),
    "nd_encode": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "nd_eps_theta": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "nd_p_sample": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "nd_sample": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "nd_set_profiling": (_i, [_vp, _i]),
    "nd_set_input_flag": (_i, [_vp, _vp]),
    "nd_set_loop_form": (_i, [_vp, _i, C.c_float]),
    "nd_loop_form": (_i, [_vp]),
    "nd_persist_status": (_i, [_vp, _i]),
    "nd_profile_read": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(_i)]),
    "nd_profile_probe_nodes": (_i, [_vp]),
    "nd_resident_weight_bytes": (C.c_longlong, [_vp, _i]),
    "nd_member_buffer": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "nd_seed": (_i, [_vp, C.c_uint64, C.c_uint32]),
    "nd_philox_normal": (_i, [_vp, _i, _i, _i, _i, _i, C.c_uint64, C.c_uint32, C.c_uint32, _vp]),
    "nd_philox_raw": (_i, [_vp, _vp, _i, C.c_uint32, C.c_uint32, _vp]),
    "nd_cond_workspace_bytes": (_sz, [C.POINTER(NdCondConfig)]),
    "nd_cond_create": (_i, [C.POINTER(NdCondConfig), C.POINTER(_vp)]),
    "nd_cond_destroy": (_i, [_vp]),
    "nd_cond_get_config": (C.POINTER(NdCondConfig), [_vp]),
    "nd_cond_bind_workspace": (_i, [_vp, _vp, _sz]),
    "nd_cond_set_patch_embed": (_i, [_vp, C.POINTER(NdPatchEmbedWeights)]),
    "nd_cond_set_block": (_i, [_vp, _i, C.POINTER(NdVitBlockWeights)]),
    "nd_cond_set_mlp": (_i, [_vp, _i, C.POINTER(NdMlpWeights)]),
    "nd_vit_block": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "nd_guiding_prediction": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "nd_predict_batch": (_i, [_vp, _vp, _vp, _vp, C.POINTER(NdBatchOut), _i, _i, _i, _f, _i, _vp]),
    "nd_packed_bytes": (_sz, [_i, _i, _i]),
    "nd_pack_rows": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "nd_linear_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "nd_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "nd_skinny_plan": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_i)]),
    "nd_step_plan": (_i, [_i, _i, _i, _i, C.POINTER(_i)]),
    "nd_skinny_row_fragments": (_i, [_i]),
    "nd_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "nd_gemm_bias_act": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "nd_split_bytes": (_sz, [_i, _i]),
    "nd_split_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "nd_join_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "nd_gemm_split_workspace_bytes": (_sz, [_i, _i, _i]),
    "nd_gemm_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "nd_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "nd_layernorm_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "nd_attention": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nd_attention_split": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "nd_qkv_images_supported": (_i, [_i, _i]),
    "nd_qkv_images_bytes": (_sz, [_i, _i, _i]),
    "nd_gemm_split_qkv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "nd_attention_images": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "nd_patchify": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nd_patchify_split": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nd_softmax_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "nd_aggregate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "nd_sample_stats": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "nd_img_add_noise": (_i, [_vp, _vp, _vp, _sz, _f, _vp]),
    "nd_img_brightness": (_i, [_vp, _vp, _sz, _f, _vp]),
    "nd_img_contrast": (_i, [_vp, _vp, _vp, _i, _sz, _f, _vp]),
    "nd_img_resize_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "nd_img_cover": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "nd_report": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
}

_lib: Optional[C.CDLL] = None


class NdError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libnd_hip.so (built in-tree by nested_diffusion_amd.build).  Raises loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NdError(
            f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
            "(run `python -m nested_diffusion_amd.build` or __graft_entry__.build()). "
            "There is no CPU fallback on the product path.")
    # torch bundles its own libamdhip64.so.7; libnd_hip.so NEEDs the same soname.  Import torch first so
    # both share ONE HIP runtime (device pointers and streams are exchanged between them); loading the
    # system runtime first would leave two runtimes in the process.
    import torch  # noqa: F401
    torch_hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(torch_hip):
        C.CDLL(torch_hip, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().nd_last_error().decode("utf-8", "replace")
        raise NdError(f"{what} failed (rc={rc}): {msg}")


def ptr(t) -> int:
    """Device (or host) address of a contiguous float32/int64 torch tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise NdError("tensor handed to the C ABI must be contiguous")
    return t.data_ptr()


def current_stream_ptr(device=None) -> Optional[int]:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
