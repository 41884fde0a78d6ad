"""GPU parity of the sampler path (through the C ABI) against the golden fixtures produced by the
reference itself and against the CPU oracle.  Tolerances are stated per check."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(G, name))
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    return z, p


def _engine(p, dims, max_rows=None, n_members=1):
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, Fd, C, T, B = dims
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=n_members, max_batch=B, max_rows=max_rows or B)
    for k in range(n_members):
        eng.load_member(k, p)
    return eng


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("name", ["s0", "s1", "s2", "s3"])
def test_encoder_and_eps_theta_vs_golden(name):
    z, p = _load(f"sampler_{name}.npz")
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    eng = _engine(p, (D, H, Fd, C, T, B))
    x, yhat = torch.from_numpy(z["x"]), torch.from_numpy(z["yhat"])
    eng.encode(x)
    xe = eng.member_buffer(0, 0, B).cpu().numpy()
    xe_ref = ref_cpu.encoder_x(p, x).numpy()
    assert _rel(xe, xe_ref) < 2e-5, _rel(xe, xe_ref)           # fp32 GEMM, different summation order
    seq = torch.from_numpy(z["seq"])
    for i, t in enumerate(z["eps_ts"]):
        yy = seq[min(T - 1 - int(t), T - 1)]
        eps = eng.eps_theta(0, yy, yhat, int(t)).cpu().numpy()
        assert _rel(eps, z["eps"][i]) < 5e-5, (name, t, _rel(eps, z["eps"][i]))
    # softplus threshold branch (> 20): large activations
    eng.encode(x * 40.0)
    big = eng.eps_theta(0, seq[0] * 30.0, yhat, T - 1).cpu().numpy()
    assert _rel(big, z["eps_big"]) < 5e-5


@pytest.mark.parametrize("name,tol", [("s0", 2e-5), ("s1", 1e-4), ("s2", 5e-5)])
def test_p_sample_loop_trajectory_vs_golden(name, tol):
    z, p = _load(f"sampler_{name}.npz")
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    eng = _engine(p, (D, H, Fd, C, T, B))
    eng.set_schedule(torch.from_numpy(z["alphas"]), torch.from_numpy(z["omabs"]))
    x, yhat, noise = (torch.from_numpy(z[k]) for k in ("x", "yhat", "noise"))
    eng.encode(x)
    yh = yhat[None].cuda()
    nz = noise[None].cuda()
    seq_g = eng.sample(yh, yh, nz, return_seq=True, use_graph=True)[0].cpu().numpy()
    seq_e = eng.sample(yh, yh, nz, return_seq=True, use_graph=False)[0].cpu().numpy()
    assert np.array_equal(seq_g, seq_e)                        # hipGraph replay == eager launches, bitwise
    ref = z["seq"]
    assert seq_g.shape == ref.shape
    assert np.array_equal(seq_g[0], ref[0])                    # y_T = noise + mean: exact
    err = np.abs(seq_g - ref).max()
    assert err < tol * max(1.0, np.abs(ref).max()), err
    y0 = eng.sample(yh, yh, nz, return_seq=False)[0].cpu().numpy()
    assert np.array_equal(y0, seq_g[-1])
    # class probabilities (the 1e-3 criterion applies here)
    pr = ref_cpu.convert_to_prob(torch.from_numpy(y0), 0.1737).numpy()
    pr_ref = ref_cpu.convert_to_prob(torch.from_numpy(ref[-1]), 0.1737).numpy()
    assert np.abs(pr - pr_ref).max() < 1e-3


def test_t1000_amplified_trajectory():
    """T=1000: 1/sqrt(abar_t) ~ 160 at t=999 amplifies rounding differences; compare relative to the
    trajectory's own scale (the golden y_0 reaches |y| ~ 4e2 with random weights)."""
    z, p = _load("sampler_s3.npz")
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    eng = _engine(p, (D, H, Fd, C, T, B))
    eng.set_schedule(torch.from_numpy(z["alphas"]), torch.from_numpy(z["omabs"]))
    x, yhat, noise = (torch.from_numpy(z[k]) for k in ("x", "yhat", "noise"))
    eng.encode(x)
    yh, nz = yhat[None].cuda(), noise[None].cuda()
    seq = eng.sample(yh, yh, nz, return_seq=True)[0].cpu().numpy()
    ref = z["seq"]
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    assert (np.abs(seq - ref) / scale).max() < 2e-3


def test_mc_trials_and_members_batched():
    """M = B*mc rows and several members in one launch equal the one-at-a-time calls."""
    z, p = _load("sampler_s0.npz")
    D, H, Fd, C, T, B, seed = [int(v) for v in z["dims"]]
    p2 = ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=99)
    from nested_diffusion_amd.engine import EnsembleEngine
    mc = 3
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=2, max_batch=B, max_rows=B * mc)
    eng.load_member(0, p); eng.load_member(1, p2)
    alphas, omabs = torch.from_numpy(z["alphas"]), torch.from_numpy(z["omabs"])
    eng.set_schedule(alphas, omabs)
    x = torch.from_numpy(z["x"])
    eng.encode(x)
    g = torch.Generator().manual_seed(3)
    yhat = torch.softmax(torch.randn(2, B, C, generator=g), -1)
    noise = torch.randn(2, T, B * mc, C, generator=g)
    y0 = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda(), mc=mc).cpu()
    for k, pk in enumerate((p, p2)):
        for j in range(mc):
            nz = noise[k, :, j * B:(j + 1) * B]
            ref = ref_cpu.p_sample_loop(pk, x, yhat[k], yhat[k], T, alphas, omabs, nz)
            got = y0[k, j * B:(j + 1) * B]
            assert (got - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max()), (k, j)
    # single-member call on member 1 reproduces the batched result bitwise
    y1 = eng.sample(yhat[1:2].cuda(), yhat[1:2].cuda(), noise[1:2].cuda(), member0=1, n_members=1, mc=mc).cpu()
    assert torch.equal(y1[0], y0[1])


def test_more_members_than_inline_descriptors():
    """Up to 8 members' descriptors travel by value in the kernel arguments; a launch over more members reads them from the
    device tables instead (nd_sampler.hip emit_loop).  K = 11 in one graph, against the oracle member by member, graph == eager,
    and a sub-range of 3 members (the by-value form) reproduces its slice bitwise."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, Fd, C, T, B, K = 64, 48, 80, 2, 5, 6, 11
    members = [ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=300 + k) for k in range(K)]
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=K, max_batch=B)
    for k in range(K):
        eng.load_member(k, members[k])
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, D, generator=g)
    eng.encode(x)
    yhat = torch.softmax(torch.randn(K, B, C, generator=g), -1)
    noise = torch.randn(K, T, B, C, generator=g)
    y0 = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda()).cpu()
    y0_eager = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda(), use_graph=False).cpu()
    assert torch.equal(y0, y0_eager)
    for k in range(K):
        ref = ref_cpu.p_sample_loop(members[k], x, yhat[k], yhat[k], T, alphas, omabs, noise[k])
        assert (y0[k] - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max()), k
    sub = eng.sample(yhat[4:7].cuda(), yhat[4:7].cuda(), noise[4:7].cuda(), member0=4, n_members=3).cpu()
    assert torch.equal(sub, y0[4:7])


def test_error_paths():
    from nested_diffusion_amd import _lib
    from nested_diffusion_amd.engine import EnsembleEngine
    with pytest.raises(_lib.NdError):
        EnsembleEngine(2, 50, 64, 64, 10)                       # data_dim not a multiple of 16
    eng = EnsembleEngine(2, 48, 64, 64, 10, max_batch=4)
    with pytest.raises(_lib.NdError):                           # member not loaded
        eng.encode(torch.zeros(2, 48))
    z, p = _load("sampler_s0.npz")
    eng.load_member(0, p)
    with pytest.raises(_lib.NdError):                           # B > max_batch
        eng.encode(torch.zeros(5, 48))
    eng.encode(torch.zeros(3, 48))
    with pytest.raises(_lib.NdError):                           # schedule not set
        eng.sample(torch.zeros(1, 3, 2).cuda(), torch.zeros(1, 3, 2).cuda(), torch.zeros(1, 10, 3, 2).cuda())
    bad = dict(p); bad["lin2.lin.weight"] = torch.zeros(8, 8)
    with pytest.raises(ValueError):
        eng.load_member(0, bad)


def test_reference_default_batch_70_and_row_groups():
    """B = 70 (configs/*.yml testing.batch_size) is not a multiple of 16: exercises padded row tiles, the
    MT = 4 kernels and two row groups; M = B*mc = 140."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, Fd, C, T, B, mc = 96, 64, 96, 2, 6, 70, 2
    p = ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=5)
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=1, max_batch=B, max_rows=B * mc)
    eng.load_member(0, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(1, B, C, generator=g), -1)
    noise = torch.randn(1, T, B * mc, C, generator=g)
    eng.encode(x)
    xe = eng.member_buffer(0, 0, B).cpu()
    assert (xe - ref_cpu.encoder_x(p, x)).abs().max() < 2e-5
    y0 = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda(), mc=mc).cpu()[0]
    for j in range(mc):
        ref = ref_cpu.p_sample_loop(p, x, yhat[0], yhat[0], T, alphas, omabs, noise[0, :, j * B:(j + 1) * B])
        assert (y0[j * B:(j + 1) * B] - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("K,Fd,C", [(3, 3200, 2), (5, 1376, 2), (2, 4112, 2), (3, 3200, 7), (5, 4096, 5)])
def test_members_with_uneven_fragment_dealing(K, Fd, C):
    """Multi-member launches whose fragments do not divide evenly over a member's workgroups: F = 3200 with 3 members gives
    85 workgroups per member, 30 of them with 3 fragments and 55 with 2; F = 1376 with 5 members 51 workgroups with 2 or 1;
    F = 4112 with 2 members 128 workgroups with 3 or 2 -- the NF / NF-1 forms of k_skinny side by side in one launch.
    C = 7 / 5 classes with 3 / 6 fragments per workgroup: more lin4 entries (fragments x C x 16) than threads, the form of the
    lin3 + lin4 epilogue that fetches them after the main loop, at the headline's dealing (5 members x 51 workgroups, F = 4096)."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, T, B = 64, 48, 3, 6
    ps = [ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=170 + k) for k in range(K)]
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=K, max_batch=B)
    for k, p in enumerate(ps):
        eng.load_member(k, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(K, B, C, generator=g), -1)
    noise = torch.randn(K, T, B, C, generator=g)
    eng.encode(x)
    y0 = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda()).cpu()
    for k, p in enumerate(ps):
        ref = ref_cpu.p_sample_loop(p, x, yhat[k], yhat[k], T, alphas, omabs, noise[k])
        assert (y0[k] - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max()), k


def test_five_members_uneven_fragment_ranges():
    """K = 5 members with F = 80 (5 fragments each): every member's fragments go to its own workgroups (one fragment each
    here); a workgroup never mixes members."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, Fd, C, T, B = 64, 48, 80, 2, 5, 9
    ps = [ref_cpu.init_cond_model_params(D, H, Fd, C, T, True, seed=70 + k) for k in range(5)]
    eng = EnsembleEngine(C, D, H, Fd, T, n_members=5, max_batch=B)
    for k, p in enumerate(ps):
        eng.load_member(k, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(5, B, C, generator=g), -1)
    noise = torch.randn(5, T, B, C, generator=g)
    eng.encode(x)
    y0 = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda()).cpu()
    for k, p in enumerate(ps):
        ref = ref_cpu.p_sample_loop(p, x, yhat[k], yhat[k], T, alphas, omabs, noise[k])
        assert (y0[k] - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max()), k


@pytest.mark.parametrize("B,mc,F_,C_", [(35, 3, 96, 2), (70, 2, 144, 3), (16, 9, 256, 2)])
def test_large_m_row_groups(B, mc, F_, C_):
    """M = B*mc > 64 rows (the reference's mc_trials = 20 regime): several 64-row groups per launch (MT = 4),
    ragged row and column fragments, C = 3, two members."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, T = 64, 48, 5
    ps = [ref_cpu.init_cond_model_params(D, H, F_, C_, T, True, seed=90 + k) for k in range(2)]
    eng = EnsembleEngine(C_, D, H, F_, T, n_members=2, max_batch=B, max_rows=B * mc)
    for k, p in enumerate(ps):
        eng.load_member(k, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(2, B, C_, generator=g), -1)
    noise = torch.randn(2, T, B * mc, C_, generator=g)
    eng.encode(x)
    y0 = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda(), mc=mc).cpu()
    assert torch.equal(eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda(), mc=mc, use_graph=False).cpu(), y0)
    for k, p in enumerate(ps):
        for j in range(mc):
            ref = ref_cpu.p_sample_loop(p, x, yhat[k], yhat[k], T, alphas, omabs, noise[k, :, j * B:(j + 1) * B])
            got = y0[k, j * B:(j + 1) * B]
            assert (got - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max()), (k, j)
    # single eps_theta evaluation through the same path
    yy = torch.randn(B * mc, C_, generator=g)
    eps = eng.eps_theta(1, yy, yhat[1], 2, mc=mc).cpu()
    ref = ref_cpu.trunk(ps[1], ref_cpu.encoder_x(ps[1], x).repeat(mc, 1), yy, torch.tensor([2]), yhat[1].repeat(mc, 1))
    assert (eps - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("B,mc,K,C_", [(32, 20, 5, 2), (70, 20, 2, 2), (33, 5, 1, 2), (16, 9, 2, 5)])
def test_reference_default_mc20_rows_at_feature_dim_4096(B, mc, K, C_):
    """The reference's own operating point: mc_trials = 20 (classification_train_separately.py:770-771) x batch 32 / 70
    (configs/chest_x_ray.yml:66) -> M = 640 / 1400 rows per member through lin2 / lin3 at F = 4096 -- the LDS-tiled
    k_cond_gemm (M > 128), incl. its k-split tail + fixup (K = 5, M = 640: 800 tiles = 768 whole + 32 cut 8 ways) and a ragged
    last row tile (M = 165).  y_0 of every (member, trial) vs the oracle; graph == eager bitwise."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, F_, T = 64, 64, 4096, 3
    ps = [ref_cpu.init_cond_model_params(D, H, F_, C_, T, True, seed=300 + k) for k in range(K)]
    eng = EnsembleEngine(C_, D, H, F_, T, n_members=K, max_batch=B, max_rows=B * mc)
    plan = eng.step_plan(B * mc)
    assert plan["kernel"] == "k_cond_gemm"
    if (B, mc, K) == (32, 20, 5):
        assert (plan["whole_tiles"], plan["remainder_tiles"], plan["split"]) == (768, 32, 8)
    for k, p in enumerate(ps):
        eng.load_member(k, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(14)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(K, B, C_, generator=g), -1)
    noise = torch.randn(K, T, B * mc, C_, generator=g)
    eng.encode(x)
    y0 = eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda(), mc=mc).cpu()
    assert torch.equal(eng.sample(yhat.cuda(), yhat.cuda(), noise.cuda(), mc=mc, use_graph=False).cpu(), y0)
    worst = 0.0
    for k, p in enumerate(ps):
        xe = ref_cpu.encoder_x(p, x)
        # all mc trials of a member at once on the CPU: rows = trial * B + image, as the engine lays them out
        cur = noise[k, 0] + yhat[k].repeat(mc, 1)
        ymean, yh, xer = yhat[k].repeat(mc, 1), yhat[k].repeat(mc, 1), xe.repeat(mc, 1)
        for i, t in enumerate(reversed(range(1, T)), start=1):
            eps = ref_cpu.trunk(p, xer, cur, torch.tensor([t]), yh)
            cur = ref_cpu.p_sample_given_eps(cur, ymean, eps, t, alphas, omabs, noise[k, i])
        eps = ref_cpu.trunk(p, xer, cur, torch.tensor([0]), yh)
        ref = ref_cpu.p_sample_t_1to0_given_eps(cur, ymean, eps, omabs)
        err = (y0[k] - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        worst = max(worst, err)
        assert err < 5e-5, (k, err)
    # one eps_theta evaluation (nd_eps_theta: eager launches of the same kernels)
    yy = torch.randn(B * mc, C_, generator=g)
    eps = eng.eps_theta(K - 1, yy, yhat[K - 1], 1, mc=mc).cpu()
    ref = ref_cpu.trunk(ps[-1], ref_cpu.encoder_x(ps[-1], x).repeat(mc, 1), yy, torch.tensor([1]), yhat[K - 1].repeat(mc, 1))
    assert (eps - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max())
    print(f"M={B * mc} K={K}: max rel y0 err {worst:.2e}, plan {plan}")


@pytest.mark.parametrize("T,B,C_", [(1, 2, 2), (2, 1, 2), (3, 5, 1), (4, 1, 5)])
def test_edge_sizes(T, B, C_):
    """Smallest loops (T = 1: no reverse step, only the t = 0 reparameterisation), single image, one and five classes."""
    from nested_diffusion_amd.engine import EnsembleEngine
    D, H, Fd = 32, 16, 32
    p = ref_cpu.init_cond_model_params(D, H, Fd, C_, T, True, seed=T * 10 + B)
    eng = EnsembleEngine(C_, D, H, Fd, T, n_members=1, max_batch=B)
    eng.load_member(0, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", max(T, 2), 1e-4, 0.02)
    alphas, omabs = alphas[:T].contiguous(), omabs[:T].contiguous()
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(T)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(B, C_, generator=g), -1)
    noise = torch.randn(T, B, C_, generator=g)
    eng.encode(x)
    seq = eng.sample(yhat[None].cuda(), yhat[None].cuda(), noise[None].cuda(), return_seq=True)[0].cpu()
    ref = torch.stack(ref_cpu.p_sample_loop(p, x, yhat, yhat, T, alphas, omabs, noise, only_last_sample=False))
    assert seq.shape == ref.shape == (T + 1, B, C_)
    assert (seq - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max())
