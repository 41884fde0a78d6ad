"""Thin torch-tensor wrappers over the standalone operators of libnd_hip.so.
Every function launches HIP kernels on torch's current stream; inputs must live on the GPU."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ND_ACT_GELU, ND_ACT_NONE, ND_ACT_RELU, ND_ACT_SOFTPLUS, check, ptr

ACT = {"none": ND_ACT_NONE, None: ND_ACT_NONE, "softplus": ND_ACT_SOFTPLUS, "relu": ND_ACT_RELU, "gelu": ND_ACT_GELU}


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.NdError(f"{name} must be a GPU tensor (no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.NdError(f"{name} must be float32")
    return t.contiguous()


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)   # one per stream: streams may run concurrently
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class PackedWeight:
    """An nn.Linear weight [N, K] repacked once into the fragment order the streaming kernels read.
    dtype 'f16': the fp16-operand image (half the bytes; K % 32 == 0) -- the fp16 mode, not the reference's arithmetic."""

    def __init__(self, weight: torch.Tensor, dtype="f32"):
        lib = _lib.load()
        weight = _f32(weight, "weight")
        self.N, self.K = weight.shape
        self.dtype = _lib.dtype_code(dtype)
        nbytes = lib.nd_packed_bytes(self.N, self.K, self.dtype)
        if nbytes == 0:
            raise _lib.NdError(f"cannot pack a [{self.N}, {self.K}] weight: K must be a positive multiple of "
                               f"{32 if self.dtype == _lib.ND_DTYPE_F16 else 16}")
        self.data = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
        check(lib.nd_pack_rows(ptr(weight), ptr(self.data), self.N, self.K, self.dtype, _stream(weight)), "nd_pack_rows")


def linear(x: torch.Tensor, weight, bias: Optional[torch.Tensor] = None, act=None,
           scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(scale * (x @ weight.T) + bias) for small row counts (weight streamed once).
    nn.Linear + ReLU of mapping/models/mlp.py:25-28.  `weight` is a PackedWeight (packed once) or a
    plain [N, K] tensor (packed on the fly: tests / one-off calls)."""
    lib = _lib.load()
    x = _f32(x, "x")
    M, K = x.shape
    if not isinstance(weight, PackedWeight):
        if weight.dim() != 2 or weight.shape[1] != K:
            raise ValueError(f"weight is {tuple(weight.shape)}, x is {tuple(x.shape)}")
        weight = PackedWeight(weight)
    N = weight.N
    if weight.K != K:
        raise ValueError(f"weight is [{weight.N}, {weight.K}], x is {tuple(x.shape)}")
    wdata = weight.data
    bias = _f32(bias, "bias") if bias is not None else None
    scale = _f32(scale, "scale") if scale is not None else None
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    nbytes = lib.nd_linear_workspace_bytes(M, K, N, weight.dtype)
    ws = _workspace(nbytes, x.device)
    check(lib.nd_linear(ptr(x), ptr(wdata), ptr(scale), ptr(bias), ptr(out), M, K, N, ACT[act], weight.dtype, ptr(ws),
                        ws.numel(), _stream(x)), "nd_linear")
    return out


class SplitMatrix:
    """frag32b3 image of an fp32 [rows, K] matrix (csrc/nd_b9.hpp): every value as its three exact bf16 pieces, in the lane order of
    v_mfma_f32_16x16x32_bf16 -- the operand form of nd_gemm_split.  Weights are converted once (at load); activations are written in
    this form by the operator that produces them."""

    def __init__(self, rows: int, K: int, device, data: Optional[torch.Tensor] = None):
        nbytes = _lib.load().nd_split_bytes(int(rows), int(K))
        if nbytes == 0:
            raise ValueError(f"no frag32b3 image for a [{rows}, {K}] matrix (K must be a positive multiple of 32)")
        self.rows, self.K = int(rows), int(K)
        self.data = data if data is not None else torch.empty(nbytes, dtype=torch.uint8, device=device)
        if self.data.numel() < nbytes or not self.data.is_cuda:
            raise ValueError("image buffer too small or not on the GPU")

    @property
    def shape(self):
        return (self.rows, self.K)

    @property
    def device(self):
        return self.data.device


def split_rows(x: torch.Tensor, out: Optional[SplitMatrix] = None) -> SplitMatrix:
    """fp32 [rows, K] -> its frag32b3 image (exact)."""
    x = _f32(x, "x")
    rows, K = x.shape
    out = out if out is not None else SplitMatrix(rows, K, x.device)
    check(_lib.load().nd_split_rows(ptr(x), ptr(out.data), rows, K, _stream(x)), "nd_split_rows")
    return out


def join_rows(s: SplitMatrix) -> torch.Tensor:
    """frag32b3 image -> fp32 [rows, K] (the three pieces summed: exact)."""
    out = torch.empty(s.rows, s.K, dtype=torch.float32, device=s.device)
    check(_lib.load().nd_join_rows(ptr(s.data), ptr(out), s.rows, s.K, _stream(out)), "nd_join_rows")
    return out


def gemm_split(x, weight: SplitMatrix, bias: Optional[torch.Tensor] = None, act=None, residual: Optional[torch.Tensor] = None,
               want_out: bool = True, want_split: bool = False, use_workspace: bool = True):
    """act(x @ weight.T + bias) + residual with exact fp32 products on the bf16 matrix pipe (nd_gemm_split).  x: fp32 [M, K] (split
    here) or a SplitMatrix.  Returns the fp32 result, its frag32b3 image (want_split; N % 32 == 0), or both as a tuple."""
    lib = _lib.load()
    xs = x if isinstance(x, SplitMatrix) else split_rows(x)
    M, K = xs.shape
    N = weight.rows
    if weight.K != K:
        raise ValueError(f"weight is {weight.shape}, x is {xs.shape}")
    dev = xs.device
    bias = _f32(bias, "bias") if bias is not None else None
    residual = _f32(residual, "residual") if residual is not None else None
    if residual is not None and tuple(residual.shape) != (M, N):
        raise ValueError("residual must be [M, N]")
    out = torch.empty(M, N, dtype=torch.float32, device=dev) if want_out else None
    osp = SplitMatrix(M, N, dev) if want_split else None
    nbytes = lib.nd_gemm_split_workspace_bytes(M, K, N) if use_workspace else 0
    ws = _workspace(nbytes, dev) if nbytes else None
    check(lib.nd_gemm_split(ptr(xs.data), ptr(weight.data), ptr(bias), ptr(residual), ptr(out), ptr(osp.data) if osp else None, M, K, N,
                            ACT[act], ptr(ws), ws.numel() if ws is not None else 0, _stream(xs.data)), "nd_gemm_split")
    if want_out and want_split:
        return out, osp
    return osp if want_split else out


def gemm_bias_act(x: torch.Tensor, weight, bias: Optional[torch.Tensor] = None, act=None,
                  residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x @ weight.T + bias) + residual for large row counts (ViT token matrices).  The kernel follows the WEIGHT the caller holds:
    a SplitMatrix (frag32b3 image made once with split_rows, as mapping.VisionTransformer does): exact fp32 products on the bf16
    matrix pipe (gemm_split); a plain fp32 tensor: the f32-input-MFMA kernel (any K % 16 == 0) -- never an implicit re-split of the
    whole weight per call; a float16 tensor: the fp16-operand kernel (x rounded to fp16 on the fly, fp32 accumulate / out; K % 32 == 0)."""
    lib = _lib.load()
    if isinstance(weight, SplitMatrix):
        return gemm_split(x, weight, bias, act, residual)
    x = _f32(x, "x")
    if not weight.is_cuda:
        raise _lib.NdError("weight must be a GPU tensor (no CPU fallback)")
    if weight.dtype == torch.float16:
        weight, dt = weight.contiguous(), _lib.ND_DTYPE_F16
    else:
        weight, dt = _f32(weight, "weight"), _lib.ND_DTYPE_F32
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight is {tuple(weight.shape)}, x is {tuple(x.shape)}")
    bias = _f32(bias, "bias") if bias is not None else None
    residual = _f32(residual, "residual") if residual is not None else None
    if residual is not None and tuple(residual.shape) != (M, N):
        raise ValueError("residual must be [M, N]")
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    nbytes = lib.nd_gemm_workspace_bytes(M, K, N, dt)
    ws = _workspace(nbytes, x.device) if nbytes else None
    check(lib.nd_gemm_bias_act(ptr(x), ptr(weight), ptr(bias), ptr(residual), ptr(out), M, K, N, ACT[act], dt, ptr(ws),
                               ws.numel() if ws is not None else 0, _stream(x)), "nd_gemm_bias_act")
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    lib = _lib.load()
    x = _f32(x, "x")
    dim = x.shape[-1]
    rows = x.numel() // dim
    out = torch.empty_like(x)
    check(lib.nd_layernorm(ptr(x), ptr(_f32(weight, "weight")), ptr(_f32(bias, "bias")), ptr(out), rows, dim, float(eps),
                           _stream(x)), "nd_layernorm")
    return out


def layernorm_split(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> SplitMatrix:
    """LayerNorm with its result written as a frag32b3 image (the input of the gemm_split that follows it in a ViT block)."""
    x = _f32(x, "x")
    dim = x.shape[-1]
    rows = x.numel() // dim
    out = SplitMatrix(rows, dim, x.device)
    check(_lib.load().nd_layernorm_split(ptr(x), ptr(_f32(weight, "weight")), ptr(_f32(bias, "bias")), ptr(out.data), rows, dim, float(eps),
                                         _stream(x)), "nd_layernorm_split")
    return out


def attention_split(qkv: torch.Tensor, B: int, N: int, heads: int) -> SplitMatrix:
    """fp32 attention with its [B*N, heads*64] result written as a frag32b3 image (the input of the proj gemm_split)."""
    qkv = _f32(qkv, "qkv")
    d = qkv.shape[-1] // (3 * heads)
    if qkv.numel() != B * N * 3 * heads * d:
        raise ValueError("qkv has the wrong number of elements")
    out = SplitMatrix(B * N, heads * d, qkv.device)
    check(_lib.load().nd_attention_split(ptr(qkv), ptr(out.data), B, N, heads, d, _stream(qkv)), "nd_attention_split")
    return out


def qkv_images_supported(N: int, heads: int) -> bool:
    return bool(_lib.load().nd_qkv_images_supported(int(N), int(heads)))


def gemm_split_qkv(x: SplitMatrix, weight: SplitMatrix, bias: Optional[torch.Tensor], B: int, N: int, heads: int,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The qkv Linear of a ViT block (x: image of [B*N, K], weight: image of [3*heads*64, K]) with its result written as the attention's
    operand images (per image and head: q, k as frag32b3 blocks, v transposed; csrc/nd_b9.hpp).  Returns the opaque image buffer.
    out: an existing uint8 buffer of nd_qkv_images_bytes(B, N, heads) bytes to write into (rows / keys past N are never written:
    whatever it held stays there)."""
    lib = _lib.load()
    if x.rows != B * N or weight.rows != 3 * heads * 64 or weight.K != x.K:
        raise ValueError(f"x is {x.shape}, weight {weight.shape}: expected [{B * N}, K] and [{3 * heads * 64}, K]")
    nbytes = lib.nd_qkv_images_bytes(B, N, heads)
    if out is None:
        img = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    else:
        if out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous() or out.numel() != nbytes or out.device != x.device:
            raise ValueError(f"out must be a contiguous uint8 GPU buffer of {nbytes} bytes on {x.device}")
        img = out
    check(lib.nd_gemm_split_qkv(ptr(x.data), ptr(weight.data), ptr(_f32(bias, "bias")) if bias is not None else None, ptr(img), B, N, heads, x.K,
                                _stream(x.data)), "nd_gemm_split_qkv")
    return img


def attention_images(img: torch.Tensor, B: int, N: int, heads: int, want_split: bool = False):
    """softmax(q k^T / 8) v on the bf16 matrix pipe with exact fp32 products, from the images gemm_split_qkv wrote: fp32 [B*N, heads*64], or
    (want_split) its frag32b3 image for the proj gemm_split."""
    out = SplitMatrix(B * N, heads * 64, img.device) if want_split else torch.empty(B * N, heads * 64, dtype=torch.float32, device=img.device)
    check(_lib.load().nd_attention_images(ptr(img), ptr(out.data if want_split else out), 1 if want_split else 0, B, N, heads, _stream(img)),
          "nd_attention_images")
    return out


def patchify_split(img: torch.Tensor, p: int) -> SplitMatrix:
    """im2col written as the frag32b3 image of [B*(H/p)*(W/p), Cin*p*p] (the input of the patch-embedding gemm_split)."""
    img = _f32(img, "img")
    B, Cin, H, W = img.shape
    out = SplitMatrix(B * (H // p) * (W // p), Cin * p * p, img.device)
    check(_lib.load().nd_patchify_split(ptr(img), ptr(out.data), B, Cin, H, W, p, _stream(img)), "nd_patchify_split")
    return out


def attention(qkv: torch.Tensor, B: int, N: int, heads: int, dtype="f32") -> torch.Tensor:
    """qkv: [B*N, 3*heads*64] (output of the qkv Linear) -> [B*N, heads*64].  dtype 'f16': fp16-operand contractions."""
    lib = _lib.load()
    qkv = _f32(qkv, "qkv")
    d = qkv.shape[-1] // (3 * heads)
    if qkv.numel() != B * N * 3 * heads * d:
        raise ValueError("qkv has the wrong number of elements")
    out = torch.empty(B * N, heads * d, dtype=torch.float32, device=qkv.device)
    check(lib.nd_attention(ptr(qkv), ptr(out), B, N, heads, d, _lib.dtype_code(dtype), _stream(qkv)), "nd_attention")
    return out


def patchify(img: torch.Tensor, p: int) -> torch.Tensor:
    """[B, Cin, H, W] -> [B*(H/p)*(W/p), Cin*p*p] (im2col for Conv2d(k=p, s=p))."""
    lib = _lib.load()
    img = _f32(img, "img")
    B, Cin, H, W = img.shape
    out = torch.empty(B * (H // p) * (W // p), Cin * p * p, dtype=torch.float32, device=img.device)
    check(lib.nd_patchify(ptr(img), ptr(out), B, Cin, H, W, p, _stream(img)), "nd_patchify")
    return out


def softmax_rows(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = _f32(x, "x")
    C = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.nd_softmax_rows(ptr(x), ptr(out), x.numel() // C, C, _stream(x)), "nd_softmax_rows")
    return out


def aggregate(samples: torch.Tensor, temperature: float, return_probs: bool = False
              ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """samples [S, B, C] -> (prob [B, C], vote [B] int64, per-sample probs [S, B, C] or None).
    convert_to_prob + compute_ensemble_confidence + majority_voting_for_mc_samples
    (classification_train_separately.py:392-398, 425-447, 51-68)."""
    lib = _lib.load()
    samples = _f32(samples, "samples")
    S, B, C = samples.shape
    prob = torch.empty(B, C, dtype=torch.float32, device=samples.device)
    vote = torch.empty(B, dtype=torch.int64, device=samples.device)
    probs = torch.empty_like(samples) if return_probs else None
    check(lib.nd_aggregate(ptr(samples), ptr(prob), ptr(vote), ptr(probs), S, B, C, float(temperature), _stream(samples)),
          "nd_aggregate")
    return prob, vote, probs


def sample_stats(probs: torch.Tensor, q_lo: float = 0.025, q_hi: float = 0.975) -> Tuple[torch.Tensor, torch.Tensor]:
    """probs [S, B, C] -> (PIW [B, C] = quantile(q_hi) - quantile(q_lo) over S, unbiased variance [B, C]).
    compute_mean_piws_for_class :108-114, calculate_variances :166-172."""
    lib = _lib.load()
    probs = _f32(probs, "probs")
    S, B, C = probs.shape
    piw = torch.empty(B, C, dtype=torch.float32, device=probs.device)
    var = torch.empty(B, C, dtype=torch.float32, device=probs.device)
    check(lib.nd_sample_stats(ptr(probs), ptr(piw), ptr(var), S, B, C, float(q_lo), float(q_hi), _stream(probs)), "nd_sample_stats")
    return piw, var


def report(piw: torch.Tensor, var: torch.Tensor, prob_mean: torch.Tensor, vote: torch.Tensor, target: torch.Tensor,
           temperature: float, n_bins: int = 10) -> dict:
    """The numbers test_atk prints (classification_train_separately.py:801-838)."""
    lib = _lib.load()
    piw, var, prob_mean = _f32(piw, "piw"), _f32(var, "var"), _f32(prob_mean, "prob_mean")
    N, C = prob_mean.shape
    vote = vote.to(device=piw.device, dtype=torch.int64).contiguous()
    target = target.to(device=piw.device, dtype=torch.int64).contiguous()
    out = torch.empty(2 + 4 * C, dtype=torch.float32, device=piw.device)
    check(lib.nd_report(ptr(piw), ptr(var), ptr(prob_mean), ptr(vote), ptr(target), ptr(out), N, C, float(temperature), int(n_bins),
                        _stream(piw)), "nd_report")
    o = out.cpu()
    return {"accuracy": o[0], "ece": o[1], "piw_correct": o[2:2 + C], "piw_incorrect": o[2 + C:2 + 2 * C],
            "var_correct": o[2 + 2 * C:2 + 3 * C], "var_incorrect": o[2 + 3 * C:2 + 4 * C]}
