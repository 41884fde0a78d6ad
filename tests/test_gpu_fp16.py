"""fp16-operand mode (BASELINE config 5) through the C ABI.

The reference has no fp16 path, so this mode has no reference-generated fixtures: PARITY UNPINNED.  It is checked
(a) against the CPU oracle's fp16-operand mode (operands rounded to fp16, exact products, fp32 accumulation) and
(b) against the fp32 HIP path, which IS pinned to the reference.  Tolerances are stated per check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu

pytestmark = pytest.mark.gpu


def _close(got, ref, tol):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,K,N,act", [(1, 64, 16, None), (5, 32, 7, "gelu"), (32, 4096, 4096, "softplus"), (17, 160, 100, "relu"),
                                       (70, 256, 48, None), (4, 16384, 64, "relu"), (32, 150528, 256, "relu"), (33, 20000 - 32, 130, None)])
def test_linear_fp16_operands(M, K, N, act):
    """nd_linear with dtype f16 == F.linear on fp16-rounded operands (products exact in fp32, fp32 accumulation):
    only the summation order differs -> 2e-5 relative, as for the fp32 kernel."""
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    s = torch.rand(N, generator=g) + 0.5
    pw = ops.PackedWeight(w.cuda(), dtype="f16")
    assert pw.data.numel() * 4 == ((N + 15) // 16) * 16 * K * 2          # half the bytes of the fp32 image
    out = ops.linear(x.cuda(), pw, b.cuda(), act=act, scale=s.cuda())
    ref = s * (x.half().double() @ w.half().double().T).float() + b
    ref = {None: lambda v: v, "softplus": F.softplus, "relu": F.relu, "gelu": F.gelu}[act](ref)
    _close(out, ref, 2e-5)
    # and it is a different (coarser) result than the fp32 path: the mode is really on
    full = s * (x.double() @ w.double().T).float() + b
    full = {None: lambda v: v, "softplus": F.softplus, "relu": F.relu, "gelu": F.gelu}[act](full)
    if K >= 4096:
        assert (out.cpu() - full).abs().max().item() > 1e-6


def test_fp16_rejects_k_not_multiple_of_32():
    from nested_diffusion_amd import _lib, ops
    with pytest.raises(_lib.NdError):
        ops.PackedWeight(torch.zeros(4, 48).cuda(), dtype="f16")
    from nested_diffusion_amd.engine import EnsembleEngine
    with pytest.raises(_lib.NdError):
        EnsembleEngine(2, 48, 64, 64, 10, dtype="f16")                      # data_dim % 32 != 0


def _member(D, H, Fd, C, T, seed):
    return ref_cpu.init_cond_model_params(D, H, Fd, C, T, seed=seed)


@pytest.mark.parametrize("D,H,Fd,C,T,B,mc", [(64, 64, 64, 2, 10, 3, 1), (96, 128, 160, 3, 12, 5, 2), (32, 32, 32, 2, 6, 33, 1)])
def test_sampler_fp16_vs_oracle_fp16_mode(D, H, Fd, C, T, B, mc):
    """encoder, eps_theta and the whole p_sample_loop in fp16-operand mode against the oracle's fp16-operand mode.
    Activations are rounded to fp16 after an fp32 epilogue whose operation order differs from the oracle's (folded
    BatchNorm), so an element can land on the neighbouring fp16 value: tolerance 2e-3 relative on xe / eps,
    5e-3 absolute on y_0 (fp32 path: 5e-5)."""
    from nested_diffusion_amd.engine import EnsembleEngine
    p = _member(D, H, Fd, C, T, seed=11 + D)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(B, C, generator=g), -1)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=1, max_batch=B, max_rows=B * mc, dtype="f16")
    eng.load_member(0, p)
    eng.set_schedule(alphas, omabs)
    eng.encode(x)
    with ref_cpu.fp16_operands():
        xe_ref = ref_cpu.encoder_x(p, x)
    xe = eng.member_buffer(0, 0, B).cpu()
    _close(xe, xe_ref, 2e-3)
    y = torch.randn(B, C, generator=g)
    for t in (0, T // 2, T - 1):
        with ref_cpu.fp16_operands():
            eps_ref = ref_cpu.trunk(p, xe_ref, y, torch.tensor([t]), yhat)
        _close(eng.eps_theta(0, y, yhat, t).cpu(), eps_ref, 2e-3)
    noise = torch.randn(T, B * mc, C, generator=g)
    y0 = eng.sample(yhat[None], yhat[None], noise[None], mc=mc, T=T)[0].cpu()
    with ref_cpu.fp16_operands():
        ref = ref_cpu.p_sample_loop(p, x.repeat(mc, 1), yhat.repeat(mc, 1), yhat.repeat(mc, 1), T, alphas, omabs, noise, True)
    assert (y0 - ref).abs().max().item() < 5e-3
    # the fp16 mode stays close to the reference's fp32 arithmetic
    ref32 = ref_cpu.p_sample_loop(p, x.repeat(mc, 1), yhat.repeat(mc, 1), yhat.repeat(mc, 1), T, alphas, omabs, noise, True)
    assert (y0 - ref32).abs().max().item() < 2e-2


def test_sampler_fp16_config_dims_isic_shape():
    """BASELINE configs[4] in its real dimensions (ISICSkinCancer YAML dims = chest_x_ray dims: D = 150528, F = H = 4096,
    temperature 0.3162), fp16-operand mode, one member, T = 20 of the 1000-step schedule's length class, B = 8, mc = 2 -- against the
    oracle's fp16-operand mode and against the fp32 reference arithmetic.  (The fp16 mode is not a reference mode: unpinned by
    nature; this checks the full-size split-K encoder stream, the frag32h step kernels and the Infinity-Cache residency split at
    config dims.)  Tolerances as in the small-shape test: 2e-3 relative on xe, 5e-3 absolute on y_0, 2e-2 against fp32."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, Fd, C, T, B, mc = 150528, 4096, 4096, 2, 20, 8, 2
    p = ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=77)
    g = torch.Generator().manual_seed(15)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(B, C, generator=g), -1)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=1, max_batch=B, max_rows=B * mc, dtype="f16")
    eng.load_member(0, p)
    eng.set_schedule(alphas, omabs)
    eng.encode(x)
    with ref_cpu.fp16_operands():
        xe_ref = ref_cpu.encoder_x(p, x)
    _close(eng.member_buffer(0, 0, B).cpu(), xe_ref, 2e-3)
    noise = torch.randn(T, B * mc, C, generator=g)
    y0 = eng.sample(yhat[None], yhat[None], noise[None], mc=mc, T=T)[0].cpu()
    with ref_cpu.fp16_operands():
        ref = ref_cpu.p_sample_loop(p, x.repeat(mc, 1), yhat.repeat(mc, 1), yhat.repeat(mc, 1), T, alphas, omabs, noise, True)
    ref32 = ref_cpu.p_sample_loop(p, x.repeat(mc, 1), yhat.repeat(mc, 1), yhat.repeat(mc, 1), T, alphas, omabs, noise, True)
    d16, d32 = (y0 - ref).abs().max().item(), (y0 - ref32).abs().max().item()
    pr = (ref_cpu.convert_to_prob(y0, 0.3162) - ref_cpu.convert_to_prob(ref32, 0.3162)).abs().max().item()
    print(f"fp16 mode at config dims: max |y0 - oracle fp16| = {d16:.2e}, max |y0 - fp32 reference arithmetic| = {d32:.2e}, class-prob delta vs fp32 = {pr:.2e}")
    assert d16 < 5e-3 and d32 < 2e-2


def test_classifier_fp16_vs_oracle_fp16_mode():
    from nested_diffusion_amd.mapping import Classifier
    p = ref_cpu.init_classifier_params(64 * 6, widths=(96, 64, 32), num_classes=2, seed=3)
    x = torch.randn(5, 6, 64, generator=torch.Generator().manual_seed(1))
    out = Classifier(p, "cuda", dtype="f16")(x.cuda()).cpu()
    with ref_cpu.fp16_operands():
        ref = ref_cpu.classifier_forward(p, x)
    _close(out, ref, 2e-3)


def test_config5_pipeline_fp16_flag_end_to_end():
    """BASELINE config 5 end to end in shape (ISICSkinCancer temperature, K = 5, T = 1000) with the --fp16 switch of the runner:
    ViT prefix (GEMMs + MFMA attention), mapping MLPs, encoder and sampler blocks on fp16 operands.  The members carry the
    denoiser-structured init (contractive chains, pinned by golden s4), so every row is tame and every row is checked: against
    the oracle in its fp16-operand mode and against the fp32 HIP path on the same inputs (class probabilities within 1e-3 --
    the yhat from the fp16 conditioner is what moves them, the chain itself forgets fp16 rounding)."""
    import argparse
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    ns = argparse.Namespace
    embed, heads, depth, img, patch, K, B, T, mc, C = 128, 2, 5, 32, 16, 5, 4, 1000, 1, 2
    D, H, Fd = 3 * img * img, 64, 64
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=13)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(64, 32, 32), seed=120 + i) for i in range(K)]
    members = [ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=140 + i, denoiser=True) for i in range(K)]
    cfg = ns(data=ns(dataset="ISICSkinCancer", num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=Fd, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=B))
    g = torch.Generator().manual_seed(19)
    x = torch.rand(B, 3, img, img, generator=g)
    noise = torch.randn(K, mc, T, B, C, generator=g)
    nz = noise.permute(0, 2, 1, 3, 4).reshape(K, T, mc * B, C).cuda()
    outs = {}
    for mode in ("f16", "f32"):
        cond = GuidingConditioner(VisionTransformer(vp, heads, dtype=mode), [Classifier(m, dtype=mode) for m in mlps])
        runner = Diffusion(ns(seed=1, mc_trials=mc, fp16=(mode == "f16")), cfg, device="cuda", conditioner=cond,
                           noise_estimator_states=members)
        assert runner.operand_dtype == mode and runner.temperature == 0.3162
        runner.load_noise_estimators(max_batch=B)
        assert runner.engine.dtype == (1 if mode == "f16" else 0)
        outs[mode] = runner.predict_batch(x.cuda(), noise=nz)
    with ref_cpu.fp16_operands(vit=True):
        logits = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=False, share_prefix=True)
        yhat = [torch.softmax(l, dim=1) for l in logits]
        alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
        raw, vote, prob = ref_cpu.ensemble_predict(members, x.flatten(1), yhat, T, alphas, omabs, noise, 0.3162, hoist=True)
    ref, got = torch.stack(raw), outs["f16"]["samples"].cpu()
    assert float(ref.abs().max()) < 8.0                                   # tame: every row counts
    d_y0 = float((got - ref).abs().max())
    d_pr = float((outs["f16"]["prob"].cpu() - prob).abs().max())
    d_32 = float((outs["f16"]["prob"] - outs["f32"]["prob"]).abs().max())
    print(f"fp16 T=1000: max |y0 - fp16 oracle| = {d_y0:.2e}, class-prob delta vs fp16 oracle {d_pr:.2e}, vs fp32 HIP {d_32:.2e}")
    assert d_y0 < 2e-3 and d_pr < 1e-3 and d_32 < 5e-3


@pytest.mark.parametrize("M,K,N,act,res", [(6272, 768, 768, None, True), (392, 768, 2304, None, False), (6250, 768, 3070, "gelu", True),
                                           (6272, 3072, 768, None, True), (200, 64, 256, "gelu", False), (130, 32, 70, None, True),
                                           (1, 32, 1, "relu", False)])
def test_gemm_fp16_operands(M, K, N, act, res):
    """nd_gemm_bias_act with an fp16 weight == the fp32 GEMM on fp16-rounded operands (exact products, fp32 accumulation):
    2e-5 relative, as for the fp32 kernel."""
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    out = ops.gemm_bias_act(x.cuda(), w.half().cuda(), b.cuda(), act=act, residual=r.cuda() if res else None)
    ref = (x.half().double() @ w.half().double().T).float() + b
    ref = {None: lambda v: v, "relu": F.relu, "gelu": F.gelu}[act](ref)
    if res:
        ref = ref + r
    _close(out, ref, 2e-5)


@pytest.mark.parametrize("B,N,heads", [(2, 196, 12), (1, 197, 12), (3, 16, 2), (2, 5, 1), (1, 256, 3), (2, 50, 4), (32, 196, 12)])
def test_attention_fp16_operands(B, N, heads):
    """nd_attention dtype f16 against torch on fp16-rounded q, k, v and fp16-rounded probabilities (fp32 softmax and sums).
    A probability can land on the neighbouring fp16 value (the kernel's fp32 softmax differs from torch's in the last bits):
    5e-4 relative (fp32 kernel: 2e-5)."""
    from nested_diffusion_amd import ops
    d = 64
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 3 * heads * d, generator=g)
    out = ops.attention(qkv.cuda(), B, N, heads, dtype="f16")
    t = qkv.half().double().reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    attn = ((t[0] @ t[1].transpose(-2, -1)) * d ** -0.5).float().softmax(-1)
    ref = (attn.half().double() @ t[2]).transpose(1, 2).reshape(B * N, heads * d).float()
    _close(out, ref, 5e-4)


def test_vit_prefix_fp16_vs_oracle_fp16_vit_mode():
    """The whole mapping network in fp16 mode (ViT GEMMs + attention + MLPs) against the oracle with fp16_operands(vit=True),
    and against the fp32 HIP path: logits within 2e-2 of the fp16 oracle / 5e-2 of fp32 at test dims."""
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    embed, heads, depth, img, patch, K, B = 128, 2, 5, 32, 16, 5, 3
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=5)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(64, 32, 32), seed=60 + i) for i in range(K)]
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(2))
    c16 = GuidingConditioner(VisionTransformer(vp, heads, dtype="f16"), [Classifier(m, dtype="f16") for m in mlps])
    c32 = GuidingConditioner(VisionTransformer(vp, heads), [Classifier(m) for m in mlps])
    l16 = [t.cpu() for t in c16.compute_guiding_prediction(x.cuda(), include_full_vit=False)]
    l32 = [t.cpu() for t in c32.compute_guiding_prediction(x.cuda(), include_full_vit=False)]
    with ref_cpu.fp16_operands(vit=True):
        ref = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=False, share_prefix=True)
    for k in range(K):
        scale = max(1.0, float(ref[k].abs().max()))
        assert (l16[k] - ref[k]).abs().max() < 2e-2 * scale
        assert (l16[k] - l32[k]).abs().max() < 5e-2 * scale
        assert (l16[k] - l32[k]).abs().max() > 0            # the mode is really on
