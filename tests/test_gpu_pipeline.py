"""End-to-end parity on the GPU: conditioner, the nn.Module / p_sample_loop drop-ins, the runner's hot
path and the config-dim golden, all through the C ABI, against the CPU oracle / reference goldens."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def ns(**kw):
    return argparse.Namespace(**kw)


def small_config(D, H, Fd, C, T, B, dataset="ChestXRay"):
    return ns(data=ns(dataset=dataset, num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=Fd, arch="linear"),
              diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                           trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
              testing=ns(batch_size=B))


def test_conditioner_vs_oracle_small_vit():
    """ViT prefix (timm 0.4.12 semantics; parity unpinned upstream) + mapping MLPs."""
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    embed, heads, depth, img, patch, K, B = 128, 2, 6, 64, 16, 5, 3
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=5)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(256, 128, 64), seed=10 + i) for i in range(K)]
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(1))
    ref = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=True, share_prefix=False)
    cond = GuidingConditioner(VisionTransformer(vp, heads), [Classifier(m) for m in mlps])
    got = cond.compute_guiding_prediction(x.cuda(), include_full_vit=True)
    assert len(got) == K + 1
    for k in range(K + 1):
        err = (got[k].cpu() - ref[k]).abs().max().item()
        assert err < 2e-5 * max(1.0, ref[k].abs().max().item()), (k, err)


def test_conditioner_arena_poisoned_with_nans_changes_no_bit():
    """The conditioner's activation arena (nd_cond workspace: token buffers, the attention's operand images, the mapping MLPs'
    packed intermediates) filled with 0xFF bytes -- fp32 and bf16 NaNs everywhere -- before a call: every byte a kernel reads is
    written first or masked (token rows past N = 196 in the last 16-row fragment and keys past N in the last key block of the qkv
    images are NEVER written), so the logits must come out bit-identical to the call on a clean arena.  196 tokens: ragged last
    fragment; B = 3: ragged last row tile of the mapping MLPs."""
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    embed, heads, depth, img, patch, K, B = 128, 2, 2, 224, 16, 2, 3
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=15)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(256, 128, 64), seed=30 + i) for i in range(K)]
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(2)).cuda()
    cond = GuidingConditioner(VisionTransformer(vp, heads), [Classifier(m) for m in mlps])
    assert cond.vit.split                                                      # the bf16 x 9 path with attention images
    clean = [t.clone() for t in cond.compute_guiding_prediction(x, include_full_vit=False)]
    assert all(torch.isfinite(t).all() for t in clean)
    ref = ref_cpu.compute_guiding_prediction(vp, mlps, x.cpu(), heads, depth, full_vit=False, share_prefix=False)
    for k in range(K):
        assert (clean[k].cpu() - ref[k]).abs().max().item() < 2e-5 * max(1.0, ref[k].abs().max().item())
    torch.cuda.synchronize()
    cond._ws.fill_(0xFF)
    poisoned = cond.compute_guiding_prediction(x, include_full_vit=False)
    for a, b in zip(clean, poisoned):
        assert torch.equal(a, b)


def test_loader_loop_uploads_straight_into_the_input_buffer_and_matches_per_batch_calls():
    """runner._rank_batches without device-side perturbations: every host batch is copied on a side stream STRAIGHT into the library's
    input buffer as soon as the previous batch's graph has read it (nd_set_input_flag: a count published to pinned host memory right
    behind the last kernel that reads the images), i.e. while the previous batch's sampler is still running.  Five DISTINCT batches
    through test_atk must give exactly the probabilities of five plain predict_batch calls on explicitly uploaded tensors (same seed,
    same in-library noise sequence): a refill that came too early would corrupt the batch still in flight, one that came too late or
    skipped the wait for its own copy would feed stale pixels.  Also: the flag has counted every batch, only host bytes of the five
    batches crossed PCIe, and with a perturbation flag on the loop takes the staged path and still agrees with per-batch calls."""
    import argparse
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    ns = argparse.Namespace
    embed, heads, depth, img, patch, K, T, C, B = 128, 2, 5, 32, 16, 5, 6, 2, 6
    D, H, F = 3 * img * img, 64, 64
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=3)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(64, 32, 16), seed=20 + i) for i in range(K)]
    members = [ref_cpu.init_cond_model_params(D, H, F, C, T, True, seed=40 + i) for i in range(K)]
    cfg = ns(data=ns(dataset="ChestXRay", num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=F, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=B))
    g = torch.Generator().manual_seed(17)
    batches = [(torch.rand(B, 3, img, img, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(5)]

    def make(**flags):
        cond = GuidingConditioner(VisionTransformer(vp, heads, "cuda"), [Classifier(m, "cuda") for m in mlps])
        return Diffusion(ns(seed=5, mc_trials=2, **flags), cfg, device="cuda", conditioner=cond, noise_estimator_states=[dict(m) for m in members])

    for flags in ({}, {"brightness": 0.1}):
        loop = make(**flags)
        loop.test_atk(test_loader=batches)
        direct = not flags
        if direct:
            torch.cuda.synchronize()
            assert loop.engine._batch_calls == 5 and int(loop.engine._input_flag[0]) == 5
        else:
            assert getattr(loop.engine, "_input_flag", None) is None                 # the staged path never asked for the signal
        assert loop.bytes_uploaded == 5 * B * 3 * img * img * 4
        plain = make(**flags)
        plain.load_noise_estimators(max_batch=B)
        plain._seed_noise(0)
        want = torch.cat([plain.predict_batch(plain.perturb(x.cuda()))["prob"] for x, _ in batches])
        assert torch.equal(loop.last_probs, want), flags
        assert not torch.equal(want[:B], want[B:2 * B])                               # the batches really differ


def test_module_and_p_sample_loop_dropin():
    """ConditionalModel.load_state_dict(reference state) + p_sample_loop / p_sample / p_sample_t_1to0 with
    the reference's signatures reproduce the golden trajectory."""
    from nested_diffusion_amd import diffusion_utils as du
    from nested_diffusion_amd.latent_model import ConditionalModel
    z = np.load(os.path.join(G, "sampler_s0.npz"))
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    model = ConditionalModel(small_config(D, H, Fd, C, T, B), guidance=True, max_batch=8)
    model.load_state_dict(p, strict=True)
    model = model.to("cuda").eval()
    x, yhat, noise = (torch.from_numpy(z[k]).cuda() for k in ("x", "yhat", "noise"))
    alphas, omabs = torch.from_numpy(z["alphas"]).cuda(), torch.from_numpy(z["omabs"]).cuda()
    seq = du.p_sample_loop(model, x, yhat, yhat, T, alphas, omabs, only_last_sample=False, noise=noise)
    assert isinstance(seq, list) and len(seq) == T + 1
    ref = z["seq"]
    got = torch.stack(seq).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    y0 = du.p_sample_loop(model, x, yhat, yhat, T, alphas, omabs, only_last_sample=True, noise=noise)
    assert np.array_equal(y0.cpu().numpy(), got[-1])
    # forward() == eps_theta of the reference at t = T-1 on y_T
    eps = model(x, torch.from_numpy(ref[0]).cuda(), torch.tensor([T - 1]), yhat)
    i = list(z["eps_ts"]).index(T - 1)
    assert np.abs(eps.cpu().numpy() - z["eps"][i]).max() < 5e-5 * max(1.0, np.abs(z["eps"][i]).max())
    # single reverse steps: y_T -> y_{T-1} with the golden draw, and the last step y_1 -> y_0
    y1 = du.p_sample(model, x, torch.from_numpy(ref[0]).cuda(), yhat, yhat, T - 1, alphas, omabs, z=noise[1])
    assert np.abs(y1.cpu().numpy() - ref[1]).max() < 2e-5 * max(1.0, np.abs(ref[1]).max())
    yl = du.p_sample_t_1to0(model, x, torch.from_numpy(ref[T - 1]).cuda(), yhat, yhat, omabs)
    assert np.abs(yl.cpu().numpy() - ref[T]).max() < 2e-5 * max(1.0, np.abs(ref[T]).max())
    # without supplied noise it still runs (device RNG) and returns the right shapes
    y_rand = du.p_sample_loop(model, x, yhat, yhat, T, alphas, omabs, only_last_sample=True)
    assert tuple(y_rand.shape) == (B, C) and torch.isfinite(y_rand).all()
    with pytest.raises(Exception):
        model.train()(x, yhat, torch.tensor([0]), yhat)        # inference only


def test_new_batch_at_a_recycled_address_is_re_encoded():
    """The encoder hoist is cached per input tensor (ConditionalModel.encode).  A NEW batch that the caching allocator places at the
    freed address of the previous one (same shape, _version 0) must not get the previous batch's xe: the reference evaluates
    encoder_x on every call (latent_model.py:169-171), and a per-batch `x = img.cuda().flatten(1)` (classification_train_separately.py:
    744-747 without the old batch kept alive) recycles exactly like this.  (a) free x1, allocate x2 at the same address, sample
    again: y_0 must be the oracle's for x2;  (b) an in-place edit of x re-encodes (version counter)."""
    from nested_diffusion_amd import diffusion_utils as du
    from nested_diffusion_amd.latent_model import ConditionalModel
    z = np.load(os.path.join(G, "sampler_s0.npz"))
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    model = ConditionalModel(small_config(D, H, Fd, C, T, B), guidance=True, max_batch=8)
    model.load_state_dict(p, strict=True)
    model = model.to("cuda").eval()
    yhat, noise = (torch.from_numpy(z[k]).cuda() for k in ("yhat", "noise"))
    alphas, omabs = torch.from_numpy(z["alphas"]).cuda(), torch.from_numpy(z["omabs"]).cuda()
    x1_host = torch.from_numpy(z["x"])
    x2_host = torch.randn(x1_host.shape, generator=torch.Generator().manual_seed(77)) * 4.0
    params = {k: v.clone() for k, v in p.items()}

    def oracle_y0(xh):
        return ref_cpu.p_sample_loop(params, xh, yhat.cpu(), yhat.cpu(), T, alphas.cpu(), omabs.cpu(), noise.cpu()).numpy()

    def tol(ref):
        return 2e-5 * max(1.0, np.abs(ref).max())

    def run(x):
        return du.p_sample_loop(model, x, yhat, yhat, T, alphas, omabs, only_last_sample=True, noise=noise).cpu().numpy()

    ref1, ref2 = oracle_y0(x1_host), oracle_y0(x2_host)
    assert np.abs(ref1 - ref2).max() > 20 * max(tol(ref1), tol(ref2))       # the two batches differ in y_0 far beyond the tolerance
    x1 = x1_host.cuda()
    assert np.abs(run(x1) - ref1).max() < tol(ref1)
    old_ptr = x1.data_ptr()
    del x1
    # The Python engine happens to hold the last encoded batch (EnsembleEngine._x_keepalive), which would keep the block from ever
    # being freed and hide the case; the cache key must not depend on that courtesy (a C-ABI caller has no such reference).
    model.hip_engine()._x_keepalive = None
    # (a) a fresh tensor of the same shape at the freed address; the allocator usually hands the block straight back
    x2 = None
    for _ in range(8):
        cand = torch.empty(x1_host.shape, device="cuda")
        if cand.data_ptr() == old_ptr:
            x2 = cand
            break
        del cand
    if x2 is None:
        pytest.skip("the caching allocator never recycled the freed block")
    x2.copy_(x2_host)                                            # version counter 1 -- so also the harder variant below
    got = run(x2)
    assert np.abs(got - ref2).max() < tol(ref2), "stale xe: the new batch was sampled with the previous batch's encoder output"
    # the same with _version == 0 on both sides: build the batch elsewhere, free, and let torch.clone land on the old address
    del x2
    src = x1_host.cuda()
    assert np.abs(run(src) - ref1).max() < tol(ref1)
    ptr = src.data_ptr()
    stage = x2_host.cuda()
    del src
    model.hip_engine()._x_keepalive = None
    x3 = stage.clone()
    if x3.data_ptr() == ptr:
        assert x3._version == 0
        assert np.abs(run(x3) - ref2).max() < tol(ref2), "stale xe at _version 0"
    # (b) an in-place edit re-encodes
    x4 = x1_host.cuda()
    assert np.abs(run(x4) - ref1).max() < tol(ref1)
    x4.copy_(x2_host.cuda())
    assert np.abs(run(x4) - ref2).max() < tol(ref2)
    # ... and a view of the same, unchanged storage does NOT (the T calls of a loop and re-flattened batches stay cached)
    calls = []
    eng = model.hip_engine()
    real = eng.encode
    eng.encode = lambda t: (calls.append(1), real(t))[1]
    try:
        assert np.abs(run(x4.view(B, -1)) - ref2).max() < tol(ref2)
        assert np.abs(run(x4.reshape(B, D)) - ref2).max() < tol(ref2)
    finally:
        eng.encode = real
    assert not calls, "a view of the cached batch was re-encoded"


def test_guidance_false_model_vs_reference_golden():
    """The nn.Module mirror built with guidance=False against golden s5 (the reference's own run of such a model): whole
    p_sample_loop trajectory through the drop-in diffusion_utils.p_sample_loop, and a forward with a per-row t vector."""
    import os
    import numpy as np
    from nested_diffusion_amd import diffusion_utils as du
    from nested_diffusion_amd.latent_model import ConditionalModel
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampler_s5.npz"))
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    model = ConditionalModel(small_config(D, H, Fd, C, T, B), guidance=False, max_batch=8)
    model.load_state_dict(p)
    model = model.cuda().eval()
    x, yhat, noise = (torch.from_numpy(z[k]).cuda() for k in ("x", "yhat", "noise"))
    alphas, omabs = torch.from_numpy(z["alphas"]).cuda(), torch.from_numpy(z["omabs"]).cuda()
    seq = du.p_sample_loop(model, x, yhat, yhat, T, alphas, omabs, only_last_sample=False, noise=noise)
    got = torch.stack(seq).cpu().numpy()
    assert got.shape == z["seq"].shape
    assert np.abs(got - z["seq"]).max() < 5e-5 * max(1.0, np.abs(z["seq"]).max())
    e = model(x, torch.from_numpy(z["seq"][1]).cuda(), torch.from_numpy(z["t_rows"]).cuda(), yhat).cpu().numpy()
    assert np.abs(e - z["eps_rows"]).max() < 5e-5 * max(1.0, np.abs(z["eps_rows"]).max())


@pytest.mark.parametrize("name", ["s0", "s1", "s2", "s3", "s4"])
def test_per_row_timesteps_vs_reference_golden(name):
    """`eps_rows` of golden s0-s4: the REFERENCE's ConditionalModel.forward called with a [B] vector of timesteps
    (latent_model.py:101-105, 169-184) -- guidance=True models, C = 2 and 3, B = 1..32, T = 10..1000 (s3 / s4 index embedding rows
    up to 999) -- against the nn.Module mirror on the GPU."""
    import os
    import numpy as np
    from nested_diffusion_amd.latent_model import ConditionalModel
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"sampler_{name}.npz"))
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    model = ConditionalModel(small_config(D, H, Fd, C, T, B), guidance=True, max_batch=max(B, 8))
    model.load_state_dict(p)
    model = model.cuda().eval()
    x, yhat = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["yhat"]).cuda()
    e = model(x, torch.from_numpy(z["seq"][1]).cuda(), torch.from_numpy(z["t_rows"]).cuda(), yhat).cpu().numpy()
    assert e.shape == z["eps_rows"].shape
    assert np.abs(e - z["eps_rows"]).max() < 5e-5 * max(1.0, np.abs(z["eps_rows"]).max()), name


def test_guidance_false_and_per_row_timesteps():
    """The two call shapes of ConditionalModel.forward the inference loop never uses (latent_model.py:157-158: lin1 on y_t
    alone; :101-105: gamma = embed(t) with one t PER ROW, the training-time call) against the oracle, and a whole reverse
    loop of a guidance=False model."""
    from nested_diffusion_amd import diffusion_utils as du
    from nested_diffusion_amd.latent_model import ConditionalModel
    D, H, Fd, C, T, B = 192, 64, 96, 3, 12, 7
    g = torch.Generator().manual_seed(4)
    x, y, yhat = torch.rand(B, D, generator=g), torch.randn(B, C, generator=g), torch.rand(B, C, generator=g)
    t_rows = torch.tensor([0, 11, 3, 3, 7, 0, 9])          # the reference draws t in [0, timesteps) (:randint), so row T of the embedding is never read
    a, s = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    for guidance in (True, False):
        p = ref_cpu.init_cond_model_params(D, H, Fd, C, T, guidance, seed=70 + guidance)
        model = ConditionalModel(small_config(D, H, Fd, C, T, B), guidance=guidance, max_batch=8)
        model.load_state_dict(p, strict=True)
        model = model.to("cuda").eval()
        ref = ref_cpu.cond_model_forward(p, x, y, t_rows, yhat if guidance else None)
        got = model(x.cuda(), y.cuda(), t_rows.cuda(), yhat.cuda() if guidance else None)
        assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), guidance
        one = model(x.cuda(), y.cuda(), torch.tensor([5]), yhat.cuda())          # yhat handed over and ignored when guidance=False
        ref1 = ref_cpu.cond_model_forward(p, x, y, torch.tensor([5]), yhat if guidance else None)
        assert (one.cpu() - ref1).abs().max().item() < 2e-5 * max(1.0, ref1.abs().max().item()), guidance
    with pytest.raises(ValueError):
        model(x.cuda(), y.cuda(), torch.tensor([1, 2]), yhat.cuda())
    # reverse loop of the guidance=False model: yhat is still the prior mean (diffusion_utils.py:139-140), not an input of eps_theta
    noise = torch.randn(T, B, C, generator=g)
    ref_seq = ref_cpu.p_sample_loop(p, x, yhat, yhat, T, a, s, noise, only_last_sample=False, guidance=False)
    seq = du.p_sample_loop(model, x.cuda(), yhat.cuda(), yhat.cuda(), T, a.cuda(), s.cuda(), only_last_sample=False,
                           noise=noise.cuda())
    ref_t, got_t = torch.stack(ref_seq), torch.stack(seq).cpu()
    assert (got_t - ref_t).abs().max().item() < 2e-5 * max(1.0, ref_t.abs().max().item())


def test_runner_hot_path_vs_oracle():
    """Diffusion.predict_batch (test_atk :749-794) == oracle ensemble on identical weights and noise:
    class probabilities within 1e-3 (the north-star criterion), votes equal."""
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    embed, heads, depth, img, patch, K, B, T, mc, C = 128, 2, 5, 32, 16, 5, 6, 20, 3, 2
    D, H, Fd = 3 * img * img, 128, 128
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=3)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(128, 64, 32), seed=20 + i) for i in range(K)]
    members = [ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=40 + i) for i in range(K)]
    cfg = small_config(D, H, Fd, C, T, B)
    cond = GuidingConditioner(VisionTransformer(vp, heads), [Classifier(m) for m in mlps])
    runner = Diffusion(ns(seed=1, mc_trials=mc), cfg, device="cuda", conditioner=cond, noise_estimator_states=members)
    runner.load_noise_estimators(max_batch=B)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 3, img, img, generator=g)
    noise = torch.randn(K, mc, T, B, C, generator=g)                      # oracle layout [K, mc, T, B, C]
    out = runner.predict_batch(x.cuda(), noise=noise.permute(0, 2, 1, 3, 4).reshape(K, T, mc * B, C).cuda())
    # oracle: as-written conditioner (no prefix sharing), as-written sampler
    logits = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=False, share_prefix=False)
    yhat = [torch.softmax(l, dim=1) for l in logits]
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    assert torch.equal(runner.alphas.cpu(), alphas) and torch.equal(runner.one_minus_alphas_bar_sqrt.cpu(), omabs)
    raw, vote, prob = ref_cpu.ensemble_predict(members, x.flatten(1), yhat, T, alphas, omabs, noise, runner.temperature, hoist=False)
    got = out["samples"].cpu()
    ref = torch.stack(raw)
    assert got.shape == ref.shape == (K * mc, B, C)
    assert (got - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max())
    assert (out["prob"].cpu() - prob).abs().max() < 1e-3
    assert torch.equal(out["vote"].cpu(), vote)
    # reference-API aggregation helpers
    lst = [r.clone() for r in raw]
    p2 = runner.compute_ensemble_confidence(lst)
    assert (p2 - prob).abs().max() < 1e-5 and (lst[0] - ref_cpu.convert_to_prob(raw[0], runner.temperature)).abs().max() < 1e-5
    assert (runner.convert_to_prob(raw[1]) - ref_cpu.convert_to_prob(raw[1], runner.temperature)).abs().max() < 1e-5


def test_config_dims_vs_reference_golden():
    """D=150528, F=H=4096 (configs/chest_x_ray.yml dims), T=10, B=4: trajectory produced by the REFERENCE on CPU."""
    from nested_diffusion_amd import synthetic
    from nested_diffusion_amd.engine import EnsembleEngine
    z = np.load(os.path.join(G, "sampler_full.npz"))
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    p = ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=seed)       # same seeded weights as the fixture
    x = torch.rand(B, D, generator=torch.Generator().manual_seed(int(z["x_seed"])))
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=1, max_batch=B)
    eng.load_member(0, p)
    del p
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    eng.encode(x)
    xe = eng.member_buffer(0, 0, B).cpu().numpy()
    assert np.abs(xe - z["xe"]).max() < 2e-5 * max(1.0, np.abs(z["xe"]).max())
    yhat, noise = torch.from_numpy(z["yhat"]).cuda(), torch.from_numpy(z["noise"]).cuda()
    seq = eng.sample(yhat[None], yhat[None], noise[None], return_seq=True)[0].cpu().numpy()
    ref = z["seq"]
    assert np.abs(seq - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), np.abs(seq - ref).max()
    pr = ref_cpu.convert_to_prob(torch.from_numpy(seq[-1]), 0.1737)
    pr_ref = ref_cpu.convert_to_prob(torch.from_numpy(ref[-1]), 0.1737)
    assert (pr - pr_ref).abs().max() < 1e-3


def test_isic_config_t1000_vs_oracle():
    """BASELINE config 5 in shape (ISICSkinCancer: temperature 0.3162; K = 5 members; T = 1000 steps), fp32, small dims.
    The members carry the denoiser-structured init (oracle/ref_cpu.py; pinned by golden s4): their chains are contractive
    like a trained estimator's, every sample stays O(1), and the 1e-3 class-probability criterion is asserted on EVERY row."""
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    embed, heads, depth, img, patch, K, B, T, mc, C = 128, 2, 5, 32, 16, 5, 4, 1000, 1, 2
    D, H, Fd = 3 * img * img, 64, 64
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=13)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(64, 32, 16), seed=120 + i) for i in range(K)]
    members = [ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=140 + i, denoiser=True) for i in range(K)]
    cfg = small_config(D, H, Fd, C, T, B, dataset="ISICSkinCancer")
    cond = GuidingConditioner(VisionTransformer(vp, heads), [Classifier(m) for m in mlps])
    runner = Diffusion(ns(seed=1, mc_trials=mc), cfg, device="cuda", conditioner=cond, noise_estimator_states=members)
    assert runner.temperature == 0.3162
    runner.load_noise_estimators(max_batch=B)
    g = torch.Generator().manual_seed(19)
    x = torch.rand(B, 3, img, img, generator=g)
    noise = torch.randn(K, mc, T, B, C, generator=g)
    out = runner.predict_batch(x.cuda(), noise=noise.permute(0, 2, 1, 3, 4).reshape(K, T, mc * B, C).cuda())
    logits = ref_cpu.compute_guiding_prediction(vp, mlps, x, heads, depth, full_vit=False, share_prefix=True)
    yhat = [torch.softmax(l, dim=1) for l in logits]
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    raw, vote, prob = ref_cpu.ensemble_predict(members, x.flatten(1), yhat, T, alphas, omabs, noise, runner.temperature, hoist=True)
    ref, got = torch.stack(raw), out["samples"].cpu()
    assert float(ref.abs().max()) < 8.0                                   # tame: every row counts
    assert (got - ref).abs().max() < 2e-4, float((got - ref).abs().max())
    assert (out["prob"].cpu() - prob).abs().max() < 1e-3
    assert torch.equal(out["vote"].cpu(), vote)


def test_calibrate_ece_equals_oracle_on_cached_samples():
    """Diffusion.test_calibrate(temp) (:449-629) for three temperatures == the oracle's compute_ensemble_confidence (:612)
    -> compute_ece (:619, with its second convert_to_prob) on the SAME raw samples (the cached draws of the first call).
    ECE is a sum of per-bin |acc - conf| * share: tolerance 2e-6 (fp32 means of <= 48 values in [0,1])."""
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    embed, heads, depth, img, patch, K, B, T, mc, C = 128, 2, 5, 32, 16, 5, 16, 8, 3, 2
    D, H, Fd = 3 * img * img, 64, 64
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=23)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(64, 32, 16), seed=220 + i) for i in range(K)]
    members = [ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=240 + i) for i in range(K)]
    cfg = small_config(D, H, Fd, C, T, B, dataset="ChestXRayValidate")
    cond = GuidingConditioner(VisionTransformer(vp, heads), [Classifier(m) for m in mlps])
    runner = Diffusion(ns(seed=5, mc_trials=mc), cfg, device="cuda", conditioner=cond, noise_estimator_states=members)
    g = torch.Generator().manual_seed(31)
    loader = [(torch.rand(B, 3, img, img, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(3)]
    eces = {}
    for temp in (0.1737, 0.2555, 0.9):
        eces[temp] = runner.test_calibrate([temp], test_loader=loader)
        assert runner.temperature == pytest.approx(temp)
    samples, targets = runner._calib_cache                               # [K*mc, 3*B, C] raw y_0, [3*B]
    assert samples.shape == (K * mc, 3 * B, C) and targets.shape == (3 * B,)
    for temp, got in eces.items():
        conf = ref_cpu.compute_ensemble_confidence([s.clone() for s in samples.cpu()], temp)
        ref = float(ref_cpu.compute_ece_as_reference(conf, targets.cpu(), temp))
        assert abs(got - ref) < 2e-6, (temp, got, ref)
    assert len({round(v, 6) for v in eces.values()}) > 1                  # the objective does depend on the temperature
