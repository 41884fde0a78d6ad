"""CPU oracle for the nested-diffusion (LaDiNE) inference hot path.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement (plain torch fp32 on the
host) of the reference algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
reported baseline -- never as the thing shipped.  The product path
(``nested_diffusion_amd``) never imports this file and fails loudly when the HIP
library is missing.

Pinning: the sampler, the eps_theta network, the schedule tables, the mapping MLP and
the aggregation functions are pinned against outputs of the reference itself, imported
in the build container by ``tests/golden/gen_golden.py`` (fixtures under ``tests/golden/``,
checked by ``tests/test_oracle_golden.py``).  The ViT prefix (timm 0.4.12, third-party,
source absent from /root/reference and not installed) is restated from timm 0.4.12's
published ``vision_transformer.py`` semantics: PARITY UNPINNED for that one piece.

All ``file:line`` citations are relative to /root/reference/.
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5          # nn.BatchNorm1d default (diffusion/latent_model.py:129,155,162-166)
LN_EPS = 1e-6          # timm 0.4.12 vit_base_patch16_224: partial(nn.LayerNorm, eps=1e-6)


# ----------------------------------------------------------------------------
# schedule  (diffusion/diffusion_utils.py:5-28; classification_train_separately.py:215-226)
# ----------------------------------------------------------------------------
def make_beta_schedule(schedule: str = "linear", num_timesteps: int = 1000,
                       start: float = 1e-5, end: float = 1e-2) -> Tensor:
    """diffusion/diffusion_utils.py:5-28, every branch, same torch ops (pinned bit-exact by schedule.npz)."""
    T = num_timesteps
    lin = torch.linspace

    def abar(u, s=0.008):                       # cosine schedule's alpha-bar (:18-23)
        return math.cos((u + s) / (1 + s) * math.pi / 2) ** 2

    table = {
        "linear": lambda: lin(start, end, T),
        "const": lambda: end * torch.ones(T),
        "quad": lambda: lin(start ** 0.5, end ** 0.5, T) ** 2,
        "jsd": lambda: 1.0 / lin(T, 1, T),
        "sigmoid": lambda: torch.sigmoid(lin(-6, 6, T)) * (end - start) + start,
        "cosine": lambda: torch.tensor([min(1 - abar((i + 1) / T) / abar(i / T), 0.999) for i in range(T)]),
        "cosine_anneal": lambda: torch.tensor([start + 0.5 * (end - start) * (1 - math.cos(t / (T - 1) * math.pi))
                                               for t in range(T)]),
    }
    table["cosine_reverse"] = table["cosine"]
    if schedule not in table:
        raise ValueError(schedule)
    return table[schedule]()


def schedule_tables(schedule: str, num_timesteps: int, start: float, end: float):
    """(alphas, one_minus_alphas_bar_sqrt) exactly as the runner derives them
    (classification_train_separately.py:215-226): fp32 throughout, cumprod in fp32."""
    betas = make_beta_schedule(schedule, num_timesteps, start, end).float()
    alphas = 1.0 - betas
    alphas_cumprod = alphas.cumprod(dim=0)
    omabs = torch.sqrt(1 - alphas_cumprod)
    if schedule == "cosine":
        omabs = omabs * 0.9999
    return alphas, omabs


# ----------------------------------------------------------------------------
# fp16-operand mode (BASELINE config 5).  NOT a mode of the reference, which is fp32 only: this models what the
# build's fp16 mode computes so that mode has a checker too -- both operands of the large Linear layers
# (encoder_x.{0,3,6}, lin2, lin3, the mapping MLP) rounded to fp16 (round to nearest even), exact products, fp32
# accumulation; everything else (BatchNorm, gains, softplus, lin1, lin4, the sampler state) stays fp32.
# ----------------------------------------------------------------------------
_FP16_OPERANDS = False
_FP16_VIT = False


@contextlib.contextmanager
def fp16_operands(enable: bool = True, vit: bool = False):
    """vit=True additionally models the build's fp16 ViT prefix: patch embedding and the four Linear layers of a block on
    fp16-rounded operands, attention with q, k, v and the normalised probabilities rounded to fp16 (fp32 softmax)."""
    global _FP16_OPERANDS, _FP16_VIT
    prev = (_FP16_OPERANDS, _FP16_VIT)
    _FP16_OPERANDS, _FP16_VIT = bool(enable), bool(enable and vit)
    try:
        yield
    finally:
        _FP16_OPERANDS, _FP16_VIT = prev


def _vit_linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    if _FP16_VIT:
        x, w = x.half().float(), w.half().float()
    return F.linear(x, w, b)


def _h(x: Tensor) -> Tensor:
    return x.half().float() if _FP16_VIT else x


def _big_linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    if _FP16_OPERANDS:
        x, w = x.half().float(), w.half().float()
    return F.linear(x, w, b)


# ----------------------------------------------------------------------------
# eps_theta network  (diffusion/latent_model.py:93-184, arch == 'linear')
# ----------------------------------------------------------------------------
def _bn_eval(u: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    """nn.BatchNorm1d in eval mode (latent_model.py:129,132,155,162-166)."""
    return F.batch_norm(u, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], False, 0.0, BN_EPS)


def encoder_x(p: Dict[str, Tensor], x: Tensor) -> Tensor:
    """``self.norm(self.encoder_x(x))`` -- latent_model.py:127-135,155,170-171.
    t-invariant part of ConditionalModel.forward."""
    h = _big_linear(x, p["encoder_x.0.weight"], p["encoder_x.0.bias"])
    h = F.softplus(_bn_eval(h, p, "encoder_x.1"))
    h = _big_linear(h, p["encoder_x.3.weight"], p["encoder_x.3.bias"])
    h = F.softplus(_bn_eval(h, p, "encoder_x.4"))
    h = _big_linear(h, p["encoder_x.6.weight"], p["encoder_x.6.bias"])
    return _bn_eval(h, p, "norm")


def _cond_linear(p: Dict[str, Tensor], name: str, h: Tensor, t: Tensor) -> Tensor:
    """ConditionalLinear.forward -- latent_model.py:101-105."""
    lin = F.linear if name == "lin1" else _big_linear          # lin1 (2C -> F) is tiny and stays fp32 in every mode
    out = lin(h, p[name + ".lin.weight"], p[name + ".lin.bias"])
    gamma = F.embedding(t, p[name + ".embed.weight"])
    return gamma.view(-1, out.shape[-1]) * out


def trunk(p: Dict[str, Tensor], xe: Tensor, y: Tensor, t: Tensor, yhat: Optional[Tensor]) -> Tensor:
    """t-dependent part of ConditionalModel.forward -- latent_model.py:172-184."""
    if yhat is not None:                                   # guidance=True (:172-173)
        y = torch.cat([y, yhat], dim=-1)
    y = F.softplus(_bn_eval(_cond_linear(p, "lin1", y, t), p, "unetnorm1"))   # :174-176
    y = xe * y                                                                 # :177
    y = F.softplus(_bn_eval(_cond_linear(p, "lin2", y, t), p, "unetnorm2"))   # :178-180
    y = F.softplus(_bn_eval(_cond_linear(p, "lin3", y, t), p, "unetnorm3"))   # :181-183
    return F.linear(y, p["lin4.weight"], p["lin4.bias"])                      # :184


def cond_model_forward(p: Dict[str, Tensor], x: Tensor, y: Tensor, t: Tensor,
                       yhat: Optional[Tensor] = None) -> Tensor:
    """ConditionalModel.forward as written (encoder re-evaluated) -- latent_model.py:169-184."""
    return trunk(p, encoder_x(p, x), y, t, yhat)


def init_cond_model_params(data_dim: int, hidden: int, feature: int, y_dim: int, n_steps: int,
                           guidance: bool = True, seed: int = 0,
                           randomize_bn: bool = True, denoiser: bool = False, denoiser_gain: float = 0.7) -> Dict[str, Tensor]:
    """Synthetic state_dict with the reference's key names/shapes (latent_model.py:108-167;
    SURVEY 8c key list).  nn.Linear default init; embed ~ U(0,1) (latent_model.py:99);
    BN running stats randomised so the eval-BN fold is exercised.

    denoiser=True: the same random network, but with an embedded signal path that makes eps_theta behave like a TRAINED noise
    estimator, eps = denoiser_gain * (y_t - yhat) / sqrt(1 - abar_t) + e(random part), so that the reverse chain is contractive
    (|y_t| stays O(1) over T = 1000 steps) instead of amplifying its start by 1/sqrt(abar_T) ~ 160 as a random-weight network
    does.  The first 2C features of every layer form +/- pairs: lin1 emits +/-(y - yhat), lin1's per-timestep gain
    (embed row t, latent_model.py:101-105) carries 1/sqrt(1 - abar_t), and lin2 / lin3 / lin4 take pair DIFFERENCES, which is
    exactly linear because softplus(p) - softplus(-p) = p -- while every softplus still works in its nonlinear range.  All
    other features keep their random weights (they read the signal features too) and reach eps through lin4 as before.  The
    schedule assumed is the shipped one (linear, 1e-4 .. 0.02; configs/*.yml:24-28)."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}

    def lin(name, n_in, n_out):
        bound = 1.0 / math.sqrt(n_in)
        p[name + ".weight"] = (torch.rand(n_out, n_in, generator=g) * 2 - 1) * bound
        p[name + ".bias"] = (torch.rand(n_out, generator=g) * 2 - 1) * bound

    def bn(name, n):
        if randomize_bn:
            p[name + ".weight"] = torch.rand(n, generator=g) + 0.5
            p[name + ".bias"] = torch.randn(n, generator=g) * 0.5
            p[name + ".running_mean"] = torch.randn(n, generator=g) * 0.5
            p[name + ".running_var"] = torch.rand(n, generator=g) * 1.5 + 0.5
        else:
            p[name + ".weight"] = torch.ones(n)
            p[name + ".bias"] = torch.zeros(n)
            p[name + ".running_mean"] = torch.zeros(n)
            p[name + ".running_var"] = torch.ones(n)
        p[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    lin("encoder_x.0", data_dim, hidden); bn("encoder_x.1", hidden)
    lin("encoder_x.3", hidden, hidden);   bn("encoder_x.4", hidden)
    lin("encoder_x.6", hidden, feature);  bn("norm", feature)
    lin("lin1.lin", y_dim * 2 if guidance else y_dim, feature)
    p["lin1.embed.weight"] = torch.rand(n_steps + 1, feature, generator=g)
    bn("unetnorm1", feature)
    for name in ("lin2", "lin3"):
        lin(name + ".lin", feature, feature)
        p[name + ".embed.weight"] = torch.rand(n_steps + 1, feature, generator=g)
        bn("unetnorm" + name[-1], feature)
    lin("lin4", feature, y_dim)
    if denoiser:
        if not guidance:
            raise ValueError("denoiser init needs guidance=True (lin1 sees [y, yhat])")
        C, S = y_dim, 2 * y_dim
        if feature < S + 1:
            raise ValueError("feature dim too small for the signal path")
        _, omabs = schedule_tables("linear", n_steps, 1e-4, 0.02)
        sign = torch.tensor([1.0] * C + [-1.0] * C)

        def bn_identity(name, shift=0.0):
            p[name + ".weight"][:S] = 1.0
            p[name + ".bias"][:S] = shift
            p[name + ".running_mean"][:S] = 0.0
            p[name + ".running_var"][:S] = 1.0
        # xe = 1 on the signal features: encoder_x.6 contributes nothing there, norm adds 1
        p["encoder_x.6.weight"][:S] = 0.0
        p["encoder_x.6.bias"][:S] = 0.0
        bn_identity("norm", 1.0)
        # lin1: +/-(y_c - yhat_c), gain 1/sqrt(1 - abar_t) in the embedding rows
        w1 = torch.zeros(S, 2 * C)
        for c in range(C):
            w1[c, c], w1[c, C + c] = 1.0, -1.0
            w1[C + c] = -w1[c]
        p["lin1.lin.weight"][:S] = w1
        p["lin1.lin.bias"][:S] = 0.0
        p["lin1.embed.weight"][:n_steps, :S] = (1.0 / omabs)[:, None]
        p["lin1.embed.weight"][n_steps, :S] = 1.0
        bn_identity("unetnorm1")
        # lin2 / lin3: pair differences -> +/-p again
        pair = torch.zeros(S, feature)
        for c in range(C):
            pair[c, c], pair[c, C + c] = 1.0, -1.0
            pair[C + c] = -pair[c]
        for name in ("lin2", "lin3"):
            p[name + ".lin.weight"][:S] = pair
            p[name + ".lin.bias"][:S] = 0.0
            p[name + ".embed.weight"][:, :S] = 1.0
            bn_identity("unetnorm" + name[-1])
        # lin4: eps_c = gain * (softplus(p_c) - softplus(-p_c)) + random part
        p["lin4.weight"][:, :S] = 0.0
        for c in range(C):
            p["lin4.weight"][c, c], p["lin4.weight"][c, C + c] = denoiser_gain, -denoiser_gain
        del sign
    return p


# ----------------------------------------------------------------------------
# sampler  (diffusion/diffusion_utils.py:31-35, 54-111, 133-163)
# ----------------------------------------------------------------------------
def extract(table: Tensor, t: Tensor, x: Tensor) -> Tensor:
    """diffusion_utils.py:31-35."""
    out = torch.gather(table, 0, t)
    return out.reshape([t.shape[0]] + [1] * (x.dim() - 1))


def p_sample_given_eps(y: Tensor, y_T_mean: Tensor, eps_theta: Tensor, t: int, alphas: Tensor,
                       omabs: Tensor, z: Tensor) -> Tensor:
    """Posterior update of p_sample with eps_theta and the noise draw supplied --
    diffusion_utils.py:68-92, operation order preserved."""
    tt = torch.tensor([t])
    alpha_t = extract(alphas, tt, y)
    s_t = extract(omabs, tt, y)
    s_tm1 = extract(omabs, tt - 1, y)
    sab_t = (1 - s_t.square()).sqrt()
    sab_tm1 = (1 - s_tm1.square()).sqrt()
    gamma_0 = (1 - alpha_t) * sab_tm1 / (s_t.square())
    gamma_1 = (s_tm1.square()) * (alpha_t.sqrt()) / (s_t.square())
    gamma_2 = 1 + (sab_t - 1) * (alpha_t.sqrt() + sab_tm1) / (s_t.square())
    y_0_reparam = 1 / sab_t * (y - (1 - sab_t) * y_T_mean - eps_theta * s_t)
    y_t_m_1_hat = gamma_0 * y_0_reparam + gamma_1 * y + gamma_2 * y_T_mean
    beta_t_hat = (s_tm1.square()) / (s_t.square()) * (1 - alpha_t)
    return y_t_m_1_hat + beta_t_hat.sqrt() * z


def p_sample_t_1to0_given_eps(y: Tensor, y_T_mean: Tensor, eps_theta: Tensor, omabs: Tensor) -> Tensor:
    """diffusion_utils.py:99-111 with eps_theta supplied."""
    tt = torch.tensor([0])
    s_t = extract(omabs, tt, y)
    sab_t = (1 - s_t.square()).sqrt()
    return 1 / sab_t * (y - (1 - sab_t) * y_T_mean - eps_theta * s_t)


def p_sample_loop(p: Dict[str, Tensor], x: Tensor, y_0_hat: Tensor, y_T_mean: Tensor, n_steps: int,
                  alphas: Tensor, omabs: Tensor, noise: Tensor, only_last_sample: bool = True,
                  hoist: bool = True, guidance: bool = True):
    """p_sample_loop (diffusion_utils.py:133-163) with the RNG draws supplied as
    ``noise[n_steps, B, C]`` in the reference's draw order: row 0 is the initial
    ``randn_like(y_T_mean)`` (:139), row i (i>=1) is the draw inside p_sample for
    t = n_steps - i (:67).  ``hoist=True`` evaluates the t-invariant encoder once
    (bit-identical on CPU); ``hoist=False`` is the as-written cost model.
    ``guidance=False``: the model was built without guidance and ignores the y_0_hat it is handed (latent_model.py:157-158, 172)."""
    assert noise.shape[0] == n_steps
    yh = y_0_hat if guidance else None
    xe = encoder_x(p, x) if hoist else None
    cur_y = noise[0] + y_T_mean                                     # :139-140
    seq = [cur_y]
    for i, t in enumerate(reversed(range(1, n_steps)), start=1):    # :145
        tt = torch.tensor([t])
        xe_t = xe if hoist else encoder_x(p, x)
        eps = trunk(p, xe_t, cur_y, tt, yh)                         # :81
        cur_y = p_sample_given_eps(cur_y, y_T_mean, eps, t, alphas, omabs, noise[i])
        seq.append(cur_y)
    xe_t = xe if hoist else encoder_x(p, x)
    eps = trunk(p, xe_t, cur_y, torch.tensor([0]), yh)              # :103
    y_0 = p_sample_t_1to0_given_eps(cur_y, y_T_mean, eps, omabs)    # :155
    if only_last_sample:
        return y_0
    seq.append(y_0)
    return seq


# ----------------------------------------------------------------------------
# mapping network  (mapping/models/mlp.py:23-29; timm 0.4.12 ViT, call sites
# classification_train_separately.py:337-346)
# ----------------------------------------------------------------------------
def classifier_forward(p: Dict[str, Tensor], x: Tensor) -> Tensor:
    """mapping/models/mlp.py:23-29 (dropout declared, unused in forward).
    The reference hard-codes reshape(-1, 196*768); small-dim tests use linear1's
    in_features instead, identical at config dims."""
    x = x.reshape(-1, p["linear1.weight"].shape[1])
    x = F.relu(_big_linear(x, p["linear1.weight"], p["linear1.bias"]))
    x = F.relu(_big_linear(x, p["linear2.weight"], p["linear2.bias"]))
    x = F.relu(_big_linear(x, p["linear3.weight"], p["linear3.bias"]))
    return _big_linear(x, p["linear4.weight"], p["linear4.bias"])


def init_classifier_params(in_features: int, widths: Sequence[int] = (4096, 2048, 128),
                           num_classes: int = 2, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}
    dims = [in_features, *widths, num_classes]
    for i in range(4):
        bound = 1.0 / math.sqrt(dims[i])
        p[f"linear{i + 1}.weight"] = (torch.rand(dims[i + 1], dims[i], generator=g) * 2 - 1) * bound
        p[f"linear{i + 1}.bias"] = (torch.rand(dims[i + 1], generator=g) * 2 - 1) * bound
    return p


def vit_patch_embed(vp: Dict[str, Tensor], x: Tensor) -> Tensor:
    """timm 0.4.12 PatchEmbed.forward: proj(x).flatten(2).transpose(1, 2); norm=Identity.
    pos_drop is identity in eval.  NOTE no cls token and no pos_embed on the mapping path
    (classification_train_separately.py:337-338)."""
    w = vp["patch_embed.proj.weight"]
    ps = w.shape[-1]
    return F.conv2d(_h(x), _h(w), vp["patch_embed.proj.bias"], stride=ps).flatten(2).transpose(1, 2)


def vit_block(vp: Dict[str, Tensor], i: int, x: Tensor, num_heads: int) -> Tensor:
    """timm 0.4.12 Block.forward (pre-LN): x += attn(norm1(x)); x += mlp(norm2(x));
    Attention: qkv -> [3,B,h,N,d]; softmax((q k^T) * d^-0.5) v; proj.  Mlp: fc1, exact GELU, fc2."""
    pre = f"blocks.{i}."
    B, N, C = x.shape
    d = C // num_heads
    h = F.layer_norm(x, (C,), vp[pre + "norm1.weight"], vp[pre + "norm1.bias"], LN_EPS)
    qkv = _vit_linear(h, vp[pre + "attn.qkv.weight"], vp[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = _h(qkv[0]), _h(qkv[1]), _h(qkv[2])
    attn = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    attn = attn.softmax(dim=-1)
    h = (_h(attn) @ v).transpose(1, 2).reshape(B, N, C)
    h = _vit_linear(h, vp[pre + "attn.proj.weight"], vp[pre + "attn.proj.bias"])
    x = x + h
    h = F.layer_norm(x, (C,), vp[pre + "norm2.weight"], vp[pre + "norm2.bias"], LN_EPS)
    h = F.gelu(_vit_linear(h, vp[pre + "mlp.fc1.weight"], vp[pre + "mlp.fc1.bias"]))
    h = _vit_linear(h, vp[pre + "mlp.fc2.weight"], vp[pre + "mlp.fc2.bias"])
    return x + h


def vit_full_forward(vp: Dict[str, Tensor], x: Tensor, num_heads: int, depth: int) -> Tensor:
    """timm 0.4.12 VisionTransformer.forward (cls token + pos_embed, final norm, head on cls)."""
    h = vit_patch_embed(vp, x)
    cls = vp["cls_token"].expand(h.shape[0], -1, -1)
    h = torch.cat((cls, h), dim=1) + vp["pos_embed"]
    for i in range(depth):
        h = vit_block(vp, i, h, num_heads)
    C = h.shape[-1]
    h = F.layer_norm(h, (C,), vp["norm.weight"], vp["norm.bias"], LN_EPS)
    return F.linear(h[:, 0], vp["head.weight"], vp["head.bias"])


def init_vit_params(embed: int = 768, depth: int = 12, mlp_ratio: int = 4, patch: int = 16,
                    in_chans: int = 3, img: int = 224, num_classes: int = 2, seed: int = 0
                    ) -> Dict[str, Tensor]:
    """Synthetic timm-0.4.12-shaped ViT state_dict (random init; LN affine randomised)."""
    g = torch.Generator().manual_seed(seed)
    n_tok = (img // patch) ** 2
    vp: Dict[str, Tensor] = {}

    def lin(name, n_in, n_out, std=None):
        s = std if std is not None else 1.0 / math.sqrt(n_in)
        vp[name + ".weight"] = torch.randn(n_out, n_in, generator=g) * s
        vp[name + ".bias"] = torch.randn(n_out, generator=g) * 0.02

    def ln(name):
        vp[name + ".weight"] = 1.0 + 0.1 * torch.randn(embed, generator=g)
        vp[name + ".bias"] = 0.05 * torch.randn(embed, generator=g)

    vp["cls_token"] = torch.randn(1, 1, embed, generator=g) * 0.02
    vp["pos_embed"] = torch.randn(1, n_tok + 1, embed, generator=g) * 0.02
    fan = in_chans * patch * patch
    vp["patch_embed.proj.weight"] = torch.randn(embed, in_chans, patch, patch, generator=g) / math.sqrt(fan)
    vp["patch_embed.proj.bias"] = torch.randn(embed, generator=g) * 0.02
    for i in range(depth):
        pre = f"blocks.{i}."
        ln(pre + "norm1"); lin(pre + "attn.qkv", embed, 3 * embed); lin(pre + "attn.proj", embed, embed)
        ln(pre + "norm2"); lin(pre + "mlp.fc1", embed, mlp_ratio * embed)
        lin(pre + "mlp.fc2", mlp_ratio * embed, embed)
    ln("norm")
    lin("head", embed, num_classes)
    return vp


def compute_guiding_prediction(vp: Dict[str, Tensor], mlps: List[Dict[str, Tensor]], x: Tensor,
                               num_heads: int, depth: int, full_vit: bool = True,
                               share_prefix: bool = True) -> List[Tensor]:
    """classification_train_separately.py:330-348: member i (1..K) = patch_embed -> blocks[0..i-1]
    -> mlps[i-1]; last element = full vit(x).  share_prefix=True reuses block j's output across
    members (identical values: eval mode, deterministic ops); False recomputes as written."""
    out: List[Tensor] = []
    if share_prefix:
        tok = vit_patch_embed(vp, x)
        for i in range(1, len(mlps) + 1):
            tok = vit_block(vp, i - 1, tok, num_heads)
            out.append(classifier_forward(mlps[i - 1], tok))
    else:
        for i in range(1, len(mlps) + 1):
            tok = vit_patch_embed(vp, x)
            for j in range(i):
                tok = vit_block(vp, j, tok, num_heads)
            out.append(classifier_forward(mlps[i - 1], tok))
    if full_vit:
        out.append(vit_full_forward(vp, x, num_heads, depth))
    return out


# ----------------------------------------------------------------------------
# aggregation  (classification_train_separately.py:51-68, 392-398, 425-447)
# ----------------------------------------------------------------------------
def convert_to_prob(y: Tensor, temperature: float) -> Tensor:
    """classification_train_separately.py:392-398."""
    logits = ((y - 1.0) ** 2) * (-1.0) / temperature
    return torch.softmax(logits, dim=-1)


def compute_ensemble_confidence(samples: List[Tensor], temperature: float) -> Tensor:
    """classification_train_separately.py:425-447 (mutates the caller's list, quirk Q4)."""
    for i in range(len(samples)):
        samples[i] = convert_to_prob(samples[i], temperature)
    return torch.mean(torch.stack(samples), dim=0)


def majority_voting_for_mc_samples(samples: List[Tensor]) -> Tensor:
    """classification_train_separately.py:51-68: mode of per-sample argmax; ties -> smallest label
    (torch.unique returns sorted labels, counts.argmax returns the first maximum)."""
    votes = torch.stack([torch.argmax(s, dim=1) for s in samples]).transpose(0, 1)
    out = []
    for i in range(votes.shape[0]):
        labels, counts = torch.unique(votes[i], return_counts=True)
        out.append(labels[counts.argmax()])
    return torch.stack(out)


# ----------------------------------------------------------------------------
# whole hot path  (classification_train_separately.py:749-794)
# ----------------------------------------------------------------------------
def ensemble_predict(members: List[Dict[str, Tensor]], x_flat: Tensor, yhat_list: List[Tensor],
                     n_steps: int, alphas: Tensor, omabs: Tensor, noise: Tensor, temperature: float,
                     hoist: bool = True):
    """test_atk :767-789 with noise[K, mc, T, B, C] supplied.  Returns (samples list member-major
    then trial, vote, prob)."""
    K, mc = noise.shape[0], noise.shape[1]
    samples: List[Tensor] = []
    for k in range(K):
        for j in range(mc):
            samples.append(p_sample_loop(members[k], x_flat, yhat_list[k], yhat_list[k], n_steps,
                                         alphas, omabs, noise[k, j], True, hoist))
    vote = majority_voting_for_mc_samples(samples)            # :786 (raw y_0)
    raw = [s.clone() for s in samples]
    prob = compute_ensemble_confidence(samples, temperature)  # :789
    return raw, vote, prob


# ----------------------------------------------------------------------------
# reporting tail of test_atk  (classification_train_separately.py:102-174, 413-423, 801-815)
# ----------------------------------------------------------------------------
def compute_mean_piws_for_class(prediction_tensors: List[Tensor], mv: Tensor, label: Tensor):
    """classification_train_separately.py:102-140: 2.5 / 97.5 % quantiles over the S samples (torch.quantile,
    linear interpolation), PIW of the voted class, mean per class over correct / incorrect predictions
    (mean of an empty selection is NaN, as in the reference)."""
    stacked = torch.stack(prediction_tensors, dim=0)
    lower = torch.quantile(stacked, q=0.025, dim=0)
    upper = torch.quantile(stacked, q=0.975, dim=0)
    piw = upper - lower
    predicted_piw = piw[torch.arange(piw.size(0)), mv]
    C = piw.size(1)
    correct_piw, incorrect_piw = torch.zeros(C), torch.zeros(C)
    for c in range(C):
        indices = (mv == c)
        correct_piw[c] = predicted_piw[indices & (mv == label)].mean()
        incorrect_piw[c] = predicted_piw[indices & (mv != label)].mean()
    return correct_piw, incorrect_piw


def calculate_variances(model_probs: List[Tensor], predicted_classes: Tensor, ground_truth: Tensor):
    """classification_train_separately.py:143-174: unbiased variance over the S samples of the class-c entry,
    averaged over the instances predicted as c (correct / incorrect); 0 when the selection is empty."""
    N, C = model_probs[0].shape
    correct_variances, incorrect_variances = torch.zeros(C), torch.zeros(C)
    for c in range(C):
        ci = (predicted_classes == c) & (ground_truth == c)
        ii = (predicted_classes == c) & (ground_truth != c)
        cp = torch.stack([p[ci, c] for p in model_probs])
        ip = torch.stack([p[ii, c] for p in model_probs])
        if cp.shape[1] > 0:
            correct_variances[c] = cp.var(dim=0).mean()
        if ip.shape[1] > 0:
            incorrect_variances[c] = ip.var(dim=0).mean()
    return correct_variances, incorrect_variances


def compute_accuracy(predictions: Tensor, labels: Tensor) -> Tensor:
    """classification_train_separately.py:801-807."""
    return torch.sum(predictions == labels).float() / predictions.numel()


def multiclass_calibration_error_l1(probs: Tensor, target: Tensor, n_bins: int = 10) -> Tensor:
    """torchmetrics==0.11.4 MulticlassCalibrationError(n_bins, norm='l1') (requirements.txt:61; package absent here:
    PARITY UNPINNED, restated from the published source): confidence = max prob, prediction = argmax,
    bins (b[i-1], b[i]] over linspace(0, 1, n_bins + 1) via torch.bucketize(...) - 1,
    ECE = sum_bins |acc_bin - conf_bin| * count_bin / N."""
    conf, pred = probs.max(dim=1)
    acc = (pred == target).to(conf.dtype)
    bounds = torch.linspace(0, 1, n_bins + 1, dtype=conf.dtype)
    idx = torch.bucketize(conf, bounds) - 1
    count = torch.zeros(n_bins, dtype=conf.dtype).scatter_add_(0, idx, torch.ones_like(conf))
    conf_bin = torch.nan_to_num(torch.zeros(n_bins, dtype=conf.dtype).scatter_add_(0, idx, conf) / count)
    acc_bin = torch.nan_to_num(torch.zeros(n_bins, dtype=conf.dtype).scatter_add_(0, idx, acc) / count)
    return torch.sum(torch.abs(acc_bin - conf_bin) * (count / count.sum()))


def compute_ece_as_reference(prob_mc: Tensor, target: Tensor, temperature: float) -> Tensor:
    """Diffusion.compute_ece as test_atk calls it (:413-423, :812): prob_in defaults to False, so the already
    averaged probabilities go through convert_to_prob once more before the calibration error (quirk kept)."""
    return multiclass_calibration_error_l1(convert_to_prob(prob_mc, temperature), target, 10)


# ----------------------------------------------------------------------------
# input perturbations  (diffusion/utils.py:272-414)
# ----------------------------------------------------------------------------
def add_noise(images: Tensor, noise_std: float, z: Tensor) -> Tensor:
    """utils.py:272-279 with the randn_like draw supplied."""
    return images + z * noise_std


def adjust_brightness(images: Tensor, k: float) -> Tensor:
    """utils.py:390-399."""
    return torch.clamp(images + k, 0, 1)


def adjust_contrast(images: Tensor, k: float) -> Tensor:
    """utils.py:402-414."""
    means = images.mean(dim=[1, 2, 3], keepdim=True)
    return torch.clamp(means + (images - means) * k, 0, 1)


def down_up_sample(images: Tensor, k: int) -> Tensor:
    """utils.py:372-387."""
    H, W = images.shape[-2:]
    down = F.interpolate(images, size=(H // k, W // k), mode="bilinear", align_corners=False)
    return F.interpolate(down, size=(H, W), mode="bilinear", align_corners=False)


def cover_regions(images: Tensor, rects, side: int) -> Tensor:
    """The pixel part of random_cover_new (utils.py:345-347) for given (top, left) corners per image."""
    out = images.clone()
    for b, rs in enumerate(rects):
        for top, left in rs:
            out[b, :, top:top + side, left:left + side] = 0
    return out


def crop_and_resize(images: Tensor, corners, crop: int) -> Tensor:
    """The pixel part of random_crop_and_resize (utils.py:292-300): crop, then torchvision 0.11 tensor Resize =
    interpolate(bilinear, align_corners=False) (torchvision absent here: that equivalence is from its published source)."""
    H, W = images.shape[-2:]
    outs = []
    for b, (top, left) in enumerate(corners):
        c = images[b:b + 1, :, top:top + crop, left:left + crop]
        outs.append(F.interpolate(c, size=(H, W), mode="bilinear", align_corners=False)[0])
    return torch.stack(outs)


# ----------------------------------------------------------------------------
# in-library noise of the throughput mode (NO reference counterpart: the reference draws with torch's global CPU/CUDA
# generator, diffusion_utils.py:67,139).  Restatement of Philox4x32-10 (Salmon et al., SC'11; Random123) and of the
# Box-Muller mapping csrc/nd_rng.hip uses, for known-answer and bit-level checks of the HIP generator.
# ----------------------------------------------------------------------------
def philox4x32_10(ctr, key0: int, key1: int):
    """ctr: uint32 array [n, 4] -> uint32 [n, 4]."""
    import numpy as np
    c = np.array(ctr, dtype=np.uint64).reshape(-1, 4).copy()
    k0, k1 = np.uint64(key0), np.uint64(key1)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c.astype(np.uint32)


def philox_normal(n_members: int, T: int, B: int, mc: int, C: int, seed: int, batch_counter: int = 0, first_image: int = 0) -> Tensor:
    """[n_members, T, mc*B, C] standard normals, row = trial*B + image: counter = (first_image + image, trial | member << 16 |
    class-quad << 24, draw index, batch counter), key = seed; Box-Muller on (x0, x1) and (x2, x3) in float64."""
    import numpy as np
    Q = (C + 3) // 4
    k, i, trial, b, q = np.meshgrid(np.arange(n_members), np.arange(T), np.arange(mc), np.arange(B), np.arange(Q), indexing="ij")
    ctr = np.stack([(first_image + b) & 0xFFFFFFFF, trial | (k << 16) | (q << 24), i, np.full_like(i, batch_counter & 0xFFFFFFFF)], axis=-1)
    x = philox4x32_10(ctr.reshape(-1, 4), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF).astype(np.float64)
    two32 = 4294967296.0
    u1 = np.minimum((x[:, [0, 2]].astype(np.float32).astype(np.float64) + 1.0), two32) / two32      # as the kernel: float(x) + 1, fp32
    u2 = x[:, [1, 3]].astype(np.float32).astype(np.float64) / two32
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.stack([r[:, 0] * np.cos(2 * np.pi * u2[:, 0]), r[:, 0] * np.sin(2 * np.pi * u2[:, 0]),
                  r[:, 1] * np.cos(2 * np.pi * u2[:, 1]), r[:, 1] * np.sin(2 * np.pi * u2[:, 1])], axis=1)
    z = z.reshape(n_members, T, mc * B, Q * 4)[..., :C]
    return torch.from_numpy(z.astype(np.float32))
