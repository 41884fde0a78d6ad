"""The C-level conditioner (nd_vit_block, nd_guiding_prediction), the in-library noise generator (nd_seed, noise_dev == NULL) and
the one-call batch path (nd_predict_batch), all through the C ABI.

Reference anchors: classification_train_separately.py:330-348 (compute_guiding_prediction), :749-794 (hot loop),
diffusion_utils.py:67,139 (the draws the generator stands in for).  Philox4x32-10 known answers: Random123 kat_vectors."""
import argparse
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ref_cpu

pytestmark = pytest.mark.gpu
ns = argparse.Namespace

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_philox_known_answers_and_random_counters():
    from nested_diffusion_amd import _lib
    lib = _lib.load()
    for ctr, key, want in KAT:
        c = torch.tensor(np.array(ctr, dtype=np.uint32).view(np.int32), dtype=torch.int32).cuda()
        o = torch.empty(4, dtype=torch.int32, device="cuda")
        _lib.check(lib.nd_philox_raw(c.data_ptr(), o.data_ptr(), 1, key[0], key[1], _stream()), "nd_philox_raw")
        assert tuple(int(v) & 0xFFFFFFFF for v in o.cpu().tolist()) == want
    rng = np.random.default_rng(3)
    ctr = rng.integers(0, 2 ** 32, size=(1000, 4), dtype=np.uint64).astype(np.uint32)
    o = torch.empty(4000, dtype=torch.int32, device="cuda")
    c = torch.from_numpy(ctr.view(np.int32).reshape(-1)).cuda()
    _lib.check(lib.nd_philox_raw(c.data_ptr(), o.data_ptr(), 1000, 0x12345678, 0x9abcdef0, _stream()), "nd_philox_raw")
    assert np.array_equal(o.cpu().numpy().view(np.uint32).reshape(1000, 4), ref_cpu.philox4x32_10(ctr, 0x12345678, 0x9abcdef0))


def _normal(K, T, B, mc, Cc, seed, batch=0, first=0):
    from nested_diffusion_amd import _lib
    out = torch.empty(K, T, B * mc, Cc, device="cuda")
    _lib.check(_lib.load().nd_philox_normal(out.data_ptr(), K, T, B, mc, Cc, seed, batch, first, _stream()), "nd_philox_normal")
    return out


@pytest.mark.parametrize("K,T,B,mc,Cc", [(5, 100, 32, 1, 2), (2, 7, 5, 3, 5), (1, 3, 1, 1, 1), (3, 4, 6, 2, 8)])
def test_philox_normal_equals_restatement_and_is_shard_independent(K, T, B, mc, Cc):
    seed = 0x1234_5678_9abc_def1
    got = _normal(K, T, B, mc, Cc, seed, batch=7, first=100).cpu()
    ref = ref_cpu.philox_normal(K, T, B, mc, Cc, seed, 7, 100)
    assert (got - ref).abs().max() < 5e-6                                  # fp32 log / sincos vs float64
    if B >= 4:                                                             # rows [lo, hi) of the full batch == a shard drawn alone
        lo, hi = 1, B - 1
        part = _normal(K, T, hi - lo, mc, Cc, seed, batch=7, first=100 + lo).cpu()
        full = got.reshape(K, T, mc, B, Cc)[:, :, :, lo:hi].reshape(K, T, mc * (hi - lo), Cc)
        assert torch.equal(part, full)
    assert not torch.equal(got, _normal(K, T, B, mc, Cc, seed, batch=8, first=100).cpu())     # next batch: new draws


def test_philox_normal_moments():
    z = _normal(5, 1000, 64, 4, 2, 99).double()
    n = z.numel()
    assert abs(z.mean()) < 4 / n ** 0.5 and abs(z.var() - 1) < 4 * (2 / n) ** 0.5
    assert abs((z ** 3).mean()) < 4 * (15 / n) ** 0.5 and abs((z ** 4).mean() - 3) < 4 * (96 / n) ** 0.5
    a, b = z[..., 0].flatten(), z[..., 1].flatten()
    assert abs((a * b).mean()) < 4 / a.numel() ** 0.5                      # the two classes of a row are uncorrelated


def _small_engine(K=2, T=6, B=5, mc=2, Cc=2, D=48, H=64, F=64):
    from nested_diffusion_amd.engine import EnsembleEngine
    eng = EnsembleEngine(Cc, D, H, F, T, n_members=K, max_batch=B, max_rows=B * mc)
    ps = [ref_cpu.init_cond_model_params(D, H, F, Cc, T, True, seed=70 + k) for k in range(K)]
    for k, p in enumerate(ps):
        eng.load_member(k, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    return eng, ps, (alphas, omabs)


def test_sampler_draws_its_own_noise_and_advances_the_batch_counter():
    K, T, B, mc, Cc = 2, 6, 5, 2, 2
    eng, ps, _ = _small_engine(K, T, B, mc, Cc)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 48, generator=g)
    yhat = torch.softmax(torch.randn(K, B, Cc, generator=g), -1).cuda()
    eng.encode(x)
    eng.seed(4242, first_image=16)
    a0 = eng.sample(yhat, yhat, None, mc=mc)                        # batch counter 0 (graph)
    a1 = eng.sample(yhat, yhat, None, mc=mc, use_graph=False)       # batch counter 1 (eager)
    a2 = eng.sample(yhat, yhat, None, mc=mc)                        # batch counter 2 (graph replay)
    for ctr, got in enumerate((a0, a1, a2)):
        want = eng.sample(yhat, yhat, _normal(K, T, B, mc, Cc, 4242, batch=ctr, first=16), mc=mc)
        assert torch.equal(got, want), ctr
    eng.seed(4242, first_image=16)                                   # re-seeding restarts the sequence
    assert torch.equal(eng.sample(yhat, yhat, None, mc=mc), a0)
    eng.seed(4243, first_image=16)
    assert not torch.equal(eng.sample(yhat, yhat, None, mc=mc), a0)


def test_member_range_draws_the_members_own_noise():
    """A member's in-library draws depend on its index in the ENSEMBLE, not on its position in the call: sampling member 1 alone
    equals member 1 of the full-range noise (nd_rng.hip counter word 1 = trial | (m0 + k) << 16 | quad << 24), graph and eager."""
    K, T, B, mc, Cc = 2, 6, 5, 2, 2
    eng, _, _ = _small_engine(K, T, B, mc, Cc)
    g = torch.Generator().manual_seed(2)
    eng.encode(torch.rand(B, 48, generator=g))
    yhat = torch.softmax(torch.randn(K, B, Cc, generator=g), -1).cuda()
    for ctr, use_graph in enumerate((True, False)):
        eng.seed(777, first_image=3)
        for _ in range(ctr):                                          # advance to batch counter `ctr`
            eng.sample(yhat, yhat, None, mc=mc)
        alone = eng.sample(yhat[1:], yhat[1:], None, member0=1, n_members=1, mc=mc, use_graph=use_graph)
        full = _normal(K, T, B, mc, Cc, 777, batch=ctr, first=3)
        want = eng.sample(yhat[1:], yhat[1:], full[1:].contiguous(), member0=1, n_members=1, mc=mc)
        assert torch.equal(alone, want), use_graph
        zero = eng.sample(yhat[1:], yhat[1:], full[:1].contiguous(), member0=1, n_members=1, mc=mc)
        assert not torch.equal(alone, zero)                           # and NOT member 0's stream


def _vit_and_mlps(embed=128, heads=2, depth=5, img=32, patch=16, K=5, widths=(64, 32, 32), seed=3):
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=seed)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=widths, seed=20 + i) for i in range(K)]
    return vp, mlps


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_c_level_conditioner_equals_operator_by_operator_launches_and_the_oracle(dtype, monkeypatch):
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    heads, depth, img = 2, 5, 32
    vp, mlps = _vit_and_mlps(heads=heads, depth=depth, img=img, widths=(64, 32, 32))
    cond = GuidingConditioner(VisionTransformer(vp, heads, dtype=dtype), [Classifier(m, dtype=dtype) for m in mlps])
    x = torch.rand(6, 3, img, img, generator=torch.Generator().manual_seed(5))
    batched = cond.compute_guiding_prediction(x.cuda(), include_full_vit=True)       # layers 2..4 of the five MLPs in three shared launches
    monkeypatch.setenv("ND_MLP_TAIL_PER_MEMBER", "1")                                 # every MLP's four layers one after the other
    got = cond.compute_guiding_prediction(x.cuda(), include_full_vit=True)
    py = cond.compute_guiding_prediction_py(x.cuda(), include_full_vit=True)
    monkeypatch.delenv("ND_MLP_TAIL_PER_MEMBER")
    assert len(got) == len(py) == len(batched) == 6
    for a, b, c in zip(got, py, batched):
        assert torch.equal(a, b)                                           # same kernels in the same order: bitwise
        # the shared launches deal the k-chunks to the waves in another order: same values to rounding (fp16 operands: a last-bit
        # difference of a hidden activation can flip its rounding to fp16, 2^-11 of that value)
        assert (a - c).abs().max().item() <= (2e-5 if dtype == "f32" else 5e-4) * max(1.0, float(a.abs().max())), dtype
    smaller = cond.compute_guiding_prediction(x[:3].cuda(), include_full_vit=False)      # B below the handle's max_batch
    if dtype == "f32":
        ref = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=True, share_prefix=False)
        for k in range(6):
            assert (got[k].cpu() - ref[k]).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), k
        for k in range(5):
            assert (smaller[k].cpu() - ref[k][:3]).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), k


def test_conditioner_split_mode_equals_f32_mfma_mode(monkeypatch):
    """The two fp32 forms of the ViT's Linear layers -- exact products from three bf16 pieces per operand on the bf16 matrix pipe
    (ND_DTYPE_F32_SPLIT: the default where every GEMM depth is a multiple of 32) and the f32-input MFMA kernels (ND_DTYPE_F32:
    ND_GEMM_F32=mfma_f32) -- compute the same arithmetic in different summation orders: same guiding predictions to fp32 rounding,
    both within the oracle's tolerance; and the C-level call equals the operator-by-operator launches bitwise in either mode."""
    from nested_diffusion_amd import _lib
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    heads, depth, img = 2, 5, 32
    vp, mlps = _vit_and_mlps(heads=heads, depth=depth, img=img, widths=(64, 32, 32))
    x = torch.rand(6, 3, img, img, generator=torch.Generator().manual_seed(8)).cuda()
    ref = ref_cpu.compute_guiding_prediction(vp, mlps, x.cpu(), heads, depth, full_vit=True, share_prefix=False)
    outs = {}
    for mode in ("b9", "mfma_f32"):
        monkeypatch.setenv("ND_GEMM_F32", mode)
        vit = VisionTransformer(vp, heads)
        assert vit.split == (mode == "b9")
        cond = GuidingConditioner(vit, [Classifier(m) for m in mlps])
        got = cond.compute_guiding_prediction(x, include_full_vit=True)
        assert _lib.load().nd_cond_get_config(cond._h).contents.operand_dtype == (_lib.ND_DTYPE_F32_SPLIT if mode == "b9" else _lib.ND_DTYPE_F32)
        monkeypatch.setenv("ND_MLP_TAIL_PER_MEMBER", "1")             # the launch sequence the operator-by-operator form makes
        for a, b in zip(cond.compute_guiding_prediction(x, include_full_vit=True), cond.compute_guiding_prediction_py(x, include_full_vit=True)):
            assert torch.equal(a, b), mode
        monkeypatch.delenv("ND_MLP_TAIL_PER_MEMBER")
        for k in range(6):
            assert (got[k].cpu() - ref[k]).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), (mode, k)
        outs[mode] = got
    for a, b in zip(outs["b9"], outs["mfma_f32"]):
        assert (a - b).abs().max().item() < 2e-5 * max(1.0, float(b.abs().max()))
        assert not torch.equal(a, b)                                  # two kernels, two summation orders: not the same code path


@pytest.mark.parametrize("dtype,B,K", [("f32", 32, 5), ("f32", 70, 5), ("f16", 32, 5), ("f32", 5, 2), ("f32", 32, 1), ("f32", 140, 3)])
def test_mapping_mlp_tails_share_launches(dtype, B, K, monkeypatch):
    """mapping/models/mlp.py:23-29 for the K members of compute_guiding_prediction (:336-345) at the reference's layer widths
    (4096 -> 2048 -> 128 -> classes): layer 1 of each MLP runs behind its prefix block, layers 2..4 of ALL members as three launches of
    K weight matrices each (nd_mlp_chain_tail) instead of 3 K launches.  Against the per-member sequence (ND_MLP_TAIL_PER_MEMBER=1) the
    logits agree to fp32 rounding (the k-chunks are dealt to the waves in another order), against the oracle within its tolerance.
    K = 1 and more than 128 rows (the LDS-tiled form of a layer) keep the per-member sequence: identical bits."""
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    heads, depth, img = 2, 5, 32
    vp, mlps = _vit_and_mlps(heads=heads, depth=depth, img=img, K=K, widths=(4096, 2048, 128))
    cond = GuidingConditioner(VisionTransformer(vp, heads, dtype=dtype), [Classifier(m, dtype=dtype) for m in mlps])
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(15))
    shared = cond.compute_guiding_prediction(x.cuda(), include_full_vit=False)
    again = cond.compute_guiding_prediction(x.cuda(), include_full_vit=False)
    monkeypatch.setenv("ND_MLP_TAIL_PER_MEMBER", "1")
    single = cond.compute_guiding_prediction(x.cuda(), include_full_vit=False)
    monkeypatch.delenv("ND_MLP_TAIL_PER_MEMBER")
    assert len(shared) == len(single) == K
    for k in range(K):
        assert torch.equal(shared[k], again[k])                                          # reproducible
        # (fp16 operands: a last-bit difference of a hidden activation can flip its rounding to fp16, 2^-11 of that value)
        assert (shared[k] - single[k]).abs().max().item() <= (2e-5 if dtype == "f32" else 5e-4) * max(1.0, float(single[k].abs().max())), k
        if K == 1 or B > 128:
            assert torch.equal(shared[k], single[k])
    if dtype == "f32":
        ref = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=False, share_prefix=True)
        for k in range(K):
            assert (shared[k].cpu() - ref[k]).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), k


@pytest.mark.parametrize("heads,embed", [(2, 128), (3, 192)])
def test_conditioner_attention_forms_agree(heads, embed, monkeypatch):
    """The ViT blocks' attention core on the bf16 matrix pipe (qkv images + k_attention_b9: the default where N % 4 == 0 and the head
    count is even) against the f32-input-MFMA kernel on the fp32 qkv (ND_ATT_F32=1; also what a ViT with an odd head count gets, here 3
    heads of 64: the image form does not apply and the conditioner falls back by itself): the same guiding predictions to fp32 rounding,
    both within the oracle's tolerance, and the C-level call equals the operator-by-operator launches bitwise in either form."""
    from nested_diffusion_amd import ops
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    depth, img = 3, 32
    vp, mlps = _vit_and_mlps(embed=embed, heads=heads, depth=depth, img=img, K=3, widths=(64, 32, 32))
    x = torch.rand(5, 3, img, img, generator=torch.Generator().manual_seed(18)).cuda()
    ref = ref_cpu.compute_guiding_prediction(vp, mlps, x.cpu(), heads, depth, full_vit=False, share_prefix=True)
    assert ops.qkv_images_supported(4, heads) == (heads % 2 == 0)
    outs = {}
    for form in ("b9", "f32"):
        if form == "f32":
            monkeypatch.setenv("ND_ATT_F32", "1")
        vit = VisionTransformer(vp, heads)
        assert vit.split
        cond = GuidingConditioner(vit, [Classifier(m) for m in mlps])
        got = cond.compute_guiding_prediction(x, include_full_vit=False)
        monkeypatch.setenv("ND_MLP_TAIL_PER_MEMBER", "1")
        for a, b in zip(cond.compute_guiding_prediction(x, include_full_vit=False), cond.compute_guiding_prediction_py(x, include_full_vit=False)):
            assert torch.equal(a, b), form
        monkeypatch.delenv("ND_MLP_TAIL_PER_MEMBER")
        for k in range(3):
            assert (got[k].cpu() - ref[k]).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), (form, k)
        outs[form] = got
    monkeypatch.delenv("ND_ATT_F32")
    for a, b in zip(outs["b9"], outs["f32"]):
        assert (a - b).abs().max().item() < 2e-5 * max(1.0, float(b.abs().max()))
        if heads % 2 == 0:
            assert not torch.equal(a, b)                      # two kernels, two summation orders
        else:
            assert torch.equal(a, b)                          # odd head count: both runs took the f32-input-MFMA attention


def test_conditioner_driven_through_ctypes_only():
    """What a C caller of include/nested_diffusion.h does, spelled out with ctypes and raw device pointers: no
    nested_diffusion_amd.mapping / ops orchestration anywhere -- pack the MLP weights, create the conditioner, hand over the
    pointers, one nd_guiding_prediction call, one nd_vit_block call."""
    from nested_diffusion_amd import _lib
    lib = _lib.load()
    embed, heads, depth, img, patch, K, B = 128, 2, 3, 32, 16, 3, 4
    widths = (64, 32, 16)
    vp, mlps = _vit_and_mlps(embed, heads, depth, img, patch, K, widths)
    n_tok = (img // patch) ** 2
    st = _stream()
    dev = {k: v.float().contiguous().cuda() for k, v in vp.items()}
    cfg = _lib.NdCondConfig()
    cfg.img_size, cfg.patch, cfg.in_chans, cfg.embed_dim, cfg.num_heads, cfg.mlp_hidden = img, patch, 3, embed, heads, 4 * embed
    cfg.n_blocks, cfg.n_mlps = depth, K
    for i in range(3):
        cfg.mlp_widths[i] = widths[i]
    cfg.num_classes, cfg.max_batch, cfg.max_tokens, cfg.operand_dtype, cfg.ln_eps = 2, B, n_tok + 1, 0, 1e-6
    h = C.c_void_p()
    _lib.check(lib.nd_cond_create(C.byref(cfg), C.byref(h)), "nd_cond_create")
    assert lib.nd_cond_get_config(h).contents.n_mlps == K
    ws = torch.empty(lib.nd_cond_workspace_bytes(C.byref(cfg)) + 256, dtype=torch.uint8, device="cuda")
    _lib.check(lib.nd_cond_bind_workspace(h, (ws.data_ptr() + 255) & ~255, ws.numel() - 256), "bind")
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(8))
    xd = x.cuda()
    logits = torch.empty(K, B, 2, device="cuda")
    yhat = torch.empty(K, B, 2, device="cuda")
    assert lib.nd_guiding_prediction(h, xd.data_ptr(), logits.data_ptr(), yhat.data_ptr(), B, st) == -3      # nothing set yet: ND_ERR_STATE
    pe_w = dev["patch_embed.proj.weight"].reshape(embed, -1).contiguous()
    pe = _lib.NdPatchEmbedWeights(pe_w.data_ptr(), dev["patch_embed.proj.bias"].data_ptr())
    _lib.check(lib.nd_cond_set_patch_embed(h, C.byref(pe)), "pe")
    for i in range(depth):
        w = _lib.NdVitBlockWeights()
        for field, key in _lib.VIT_BLOCK_FIELDS:
            setattr(w, field, dev[f"blocks.{i}.{key}"].data_ptr())
        _lib.check(lib.nd_cond_set_block(h, i, C.byref(w)), "block")
    keep = []
    for i, m in enumerate(mlps):
        w = _lib.NdMlpWeights()
        for l in range(4):
            wt = m[f"linear{l + 1}.weight"].float().contiguous().cuda()
            N, Kd = wt.shape
            pk = torch.empty(lib.nd_packed_bytes(N, Kd, 0) // 4, device="cuda")
            _lib.check(lib.nd_pack_rows(wt.data_ptr(), pk.data_ptr(), N, Kd, 0, st), "pack")
            b = m[f"linear{l + 1}.bias"].float().cuda()
            keep += [pk, b]
            w.w_packed[l], w.bias[l] = pk.data_ptr(), b.data_ptr()
        _lib.check(lib.nd_cond_set_mlp(h, i, C.byref(w)), "mlp")
    _lib.check(lib.nd_guiding_prediction(h, xd.data_ptr(), logits.data_ptr(), yhat.data_ptr(), B, st), "nd_guiding_prediction")
    ref = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=False, share_prefix=False)
    for k in range(K):
        assert (logits[k].cpu() - ref[k]).abs().max() < 2e-5 * max(1.0, float(ref[k].abs().max())), k
        assert (yhat[k].cpu() - torch.softmax(ref[k], 1)).abs().max() < 1e-5
    # one Block.forward on tokens with a cls row (N = n_tok + 1), in place
    tok = torch.randn(B, n_tok + 1, embed, generator=torch.Generator().manual_seed(2))
    td = tok.cuda().contiguous()
    _lib.check(lib.nd_vit_block(h, 1, td.data_ptr(), td.data_ptr(), B, n_tok + 1, st), "nd_vit_block")
    want = ref_cpu.vit_block(vp, 1, tok, heads)
    assert (td.cpu() - want).abs().max() < 2e-5 * float(want.abs().max())
    assert lib.nd_vit_block(h, depth, td.data_ptr(), td.data_ptr(), B, n_tok, st) != 0                       # no such block
    assert lib.nd_vit_block(h, 0, td.data_ptr(), td.data_ptr(), B, n_tok + 2, st) != 0                       # more tokens than max_tokens
    torch.cuda.synchronize()
    lib.nd_cond_destroy(h)


def _runner(K=5, T=10, B=4, mc=2, dtype="f32"):
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    embed, heads, depth, img, patch, Cc = 128, 2, 5, 32, 16, 2
    D, H, F = 3 * img * img, 64, 64
    vp, mlps = _vit_and_mlps(embed, heads, depth, img, patch, K)
    members = [ref_cpu.init_cond_model_params(D, H, F, Cc, T, True, seed=40 + i) for i in range(K)]
    cfg = ns(data=ns(dataset="ChestXRay", num_classes=Cc), model=ns(data_dim=D, hidden_dim=H, feature_dim=F, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=B))
    cond = GuidingConditioner(VisionTransformer(vp, heads, dtype=dtype), [Classifier(m, dtype=dtype) for m in mlps])
    r = Diffusion(ns(seed=1, mc_trials=mc, fp16=(dtype == "f16")), cfg, device="cuda", conditioner=cond, noise_estimator_states=members)
    r.load_noise_estimators(max_batch=B)
    return r, (vp, mlps, members, heads, depth, img)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_predict_batch_graph_equals_eager_equals_the_separate_calls(dtype):
    from nested_diffusion_amd import ops
    K, T, B, mc, Cc = 5, 10, 4, 2, 2
    r, (vp, mlps, members, heads, depth, img) = _runner(K, T, B, mc, dtype)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 3, img, img, generator=g).cuda()
    nz = torch.randn(K, T, mc * B, Cc, generator=g).cuda()
    first = r.predict_batch(x, noise=nz)                         # first call of a shape: eager pass + graph recording
    replay = r.predict_batch(x, noise=nz)                        # graph replay
    eager = r.predict_batch(x, noise=nz, use_graph=False)
    # the same path as separate library calls
    logits = r.compute_guiding_prediction(x, include_full_vit=False)
    yhat = torch.stack([ops.softmax_rows(l) for l in logits])
    r.engine.encode(torch.flatten(x, 1))
    y0 = r.engine.sample(yhat, yhat, nz, mc=mc, T=T).reshape(K * mc, B, Cc)
    prob, vote, probs = ops.aggregate(y0, r.temperature, return_probs=True)
    for out in (first, replay, eager):
        assert torch.equal(out["yhat"], yhat) and torch.equal(out["samples"], y0)
        assert torch.equal(out["prob"], prob) and torch.equal(out["vote"], vote) and torch.equal(out["probs"], probs)
    # another batch through the recorded graph: inputs are read at replay time, nothing is baked in
    x2 = torch.rand(B, 3, img, img, generator=g).cuda()
    a = r.predict_batch(x2, noise=nz)
    b = r.predict_batch(x2, noise=nz, use_graph=False)
    assert torch.equal(a["samples"], b["samples"]) and not torch.equal(a["samples"], first["samples"])
    # clone=False hands out the fixed buffers
    c1 = r.predict_batch(x, noise=nz, clone=False)
    assert c1["prob"].data_ptr() == r.predict_batch(x2, noise=nz, clone=False)["prob"].data_ptr()


def test_predict_batch_in_library_noise_graph_and_eager_follow_the_same_sequence():
    K, T, B, mc, Cc = 5, 10, 4, 2, 2
    r, (_, _, _, _, _, img) = _runner(K, T, B, mc)
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(3)).cuda()
    r.engine.seed(77, first_image=8)
    outs = [r.predict_batch(x) for _ in range(3)]               # batch counters 0 (eager + recording), 1, 2 (replays)
    assert not torch.equal(outs[0]["samples"], outs[1]["samples"]) and not torch.equal(outs[1]["samples"], outs[2]["samples"])
    for ctr, out in enumerate(outs):
        want = r.predict_batch(x, noise=_normal(K, T, B, mc, Cc, 77, batch=ctr, first=8))
        assert torch.equal(out["samples"], want["samples"]) and torch.equal(out["prob"], want["prob"]), ctr
    r.engine.seed(77, first_image=8)
    assert torch.equal(r.predict_batch(x, use_graph=False)["samples"], outs[0]["samples"])


def test_predict_batch_argument_checks():
    from nested_diffusion_amd import _lib
    r, (_, _, _, _, _, img) = _runner(5, 4, 3, 1)
    x = torch.rand(3, 3, img, img).cuda()
    with pytest.raises(ValueError):
        r.predict_batch(x, noise=torch.zeros(5, 4, 7, 2).cuda())          # wrong row count
    with pytest.raises(_lib.NdError):
        r.predict_batch(torch.rand(4, 3, img, img).cuda())                 # B above max_batch
    with pytest.raises(_lib.NdError):
        r.predict_batch(x.cpu())                                           # no CPU fallback


def test_recorded_graphs_do_not_outlive_a_replaced_conditioner():
    """GuidingConditioner.handle() re-creates its nd_cond when the batch grows; the allocator may hand the new handle the address
    of the destroyed one.  Batch graphs are keyed on the conditioner's serial (it changes with every state change), so a graph
    recorded against the old handle's workspace is never replayed."""
    K, T, B, mc, Cc = 5, 4, 4, 1, 2
    r, (_, _, _, _, _, img) = _runner(K, T, B, mc)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, 3, img, img, generator=g).cuda()
    nz3, nz4 = torch.randn(K, T, 3, Cc, generator=g).cuda(), torch.randn(K, T, 4, Cc, generator=g).cuda()
    for _ in range(2):
        small = r.predict_batch(x[:3], noise=nz3)               # handle sized for 3 images; graph recorded, then replayed
    h_small = r.cond_pred_model._h.value
    for _ in range(2):
        big = r.predict_batch(x, noise=nz4)                      # batch of 4: the conditioner handle is re-created
    assert r.cond_pred_model._key[0] == 4
    again = r.predict_batch(x[:3], noise=nz3)                    # the (B = 3) graph of the destroyed handle must not be replayed
    assert torch.equal(again["samples"], small["samples"]) and torch.equal(again["prob"], small["prob"])
    assert torch.equal(big["samples"], r.predict_batch(x, noise=nz4, use_graph=False)["samples"])
    print("conditioner handle address reused:", r.cond_pred_model._h.value == h_small)


def test_probes_do_not_change_results_and_report_intervals():
    """nd_set_profiling: event-record nodes inside the batch graph (bench.py's roofline probes) leave every output bit-identical;
    nd_profile_read returns positive intervals for min(8, (T - 1) // 2) probed pairs of steps; what a record node adds is the smallest."""
    K, T, B, mc, Cc = 5, 12, 4, 1, 2
    r, (_, _, _, _, _, img) = _runner(K, T, B, mc)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(B, 3, img, img, generator=g).cuda()
    nz = torch.randn(K, T, B, Cc, generator=g).cuda()
    plain = r.predict_batch(x, noise=nz)
    r.engine.set_profiling(True)
    for _ in range(2):                                            # eager + recording, then a replay
        probed = r.predict_batch(x, noise=nz)
    head, pair, rec, n = r.engine.profile_read()
    # the probes are NODES of the recorded batch graph (stamped on every replay), not leftovers of the eager first call
    assert r.engine.probe_nodes() == 4 * 5                        # T = 12: five pairs of steps, four record nodes each
    r.engine.set_profiling(False)
    assert n == 5 and head > 0 and pair > 0 and rec >= 0 and rec < head and rec < pair
    for k in ("samples", "prob", "vote", "probs", "yhat"):
        assert torch.equal(plain[k], probed[k]), k
    assert torch.equal(r.predict_batch(x, noise=nz)["samples"], plain["samples"])
    assert r.engine.probe_nodes() == 0                            # rebuilt with profiling off: no record node


def test_inputs_consumed_count_is_published_per_call_and_replay():
    """nd_set_input_flag: every nd_predict_batch call -- the eager first call of a shape, graph replays, use_graph = 0 -- stores the count of
    calls whose input buffer has been read to the caller's pinned host word (right behind the encoder hoist's pack, the last reader of
    images_dev); results are unchanged by the signal; NULL switches it off; the count restarts with the next flag."""
    import ctypes
    K, T, B, mc = 5, 6, 4, 2
    r, (_, _, _, _, _, img) = _runner(K, T, B, mc)
    eng = r.engine
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(2)).cuda()
    nz = torch.randn(K, T, mc * B, 2, generator=torch.Generator().manual_seed(3)).cuda()
    before = r.predict_batch(x, noise=nz)["prob"]
    eng.enable_input_flag()
    assert int(eng._input_flag[0]) == 0
    outs = [r.predict_batch(x, noise=nz)["prob"] for _ in range(3)]            # eager + recording, then two replays
    outs.append(r.predict_batch(x, noise=nz, use_graph=False)["prob"])
    eng.inputs_consumed()                                                       # host wait: returns once the 4th batch has read its inputs
    torch.cuda.synchronize()
    assert int(eng._input_flag[0]) == 4 == eng._batch_calls
    for o in outs:
        assert torch.equal(o, before)
    assert eng.lib.nd_set_input_flag(eng.h, None) == 0                          # off: the word stays where it is
    r.predict_batch(x, noise=nz)
    torch.cuda.synchronize()
    assert int(eng._input_flag[0]) == 4
    word = torch.zeros(1, dtype=torch.int32).pin_memory()
    assert eng.lib.nd_set_input_flag(eng.h, ctypes.c_void_p(word.data_ptr())) == 0
    r.predict_batch(x, noise=nz)
    torch.cuda.synchronize()
    assert int(word[0]) == 1                                                     # a new flag counts from one
    assert eng.lib.nd_set_input_flag(eng.h, ctypes.c_void_p(word.data_ptr() + 2)) != 0   # misaligned
    eng._input_flag = None                                                       # (this test drove the flag by hand)
    assert eng.lib.nd_set_input_flag(eng.h, None) == 0
