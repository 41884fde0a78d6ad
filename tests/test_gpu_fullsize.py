"""BASELINE-size checks (D=150528, F=H=4096, K=5, T=100, B=32) through size-independent properties -- the CPU
oracle cannot run this size in seconds -- plus one oracle comparison at config dims with a short T."""
import pytest
import torch

from oracle import ref_cpu

pytestmark = pytest.mark.gpu
D, H, F, C = 150528, 4096, 4096, 2


@pytest.fixture(scope="module")
def engine5():
    from nested_diffusion_amd import synthetic
    from nested_diffusion_amd.engine import EnsembleEngine
    K, T, B = 5, 100, 32
    eng = EnsembleEngine(C, D, H, F, T, n_members=K, max_batch=B, max_rows=2 * B)
    for k in range(K):
        eng.load_member(k, synthetic.cond_model_state(D, H, F, C, T, seed=1000 + k))
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    return eng


def test_fullsize_determinism_permutation_and_batching(engine5):
    from nested_diffusion_amd import ops, synthetic
    eng, K, T, B = engine5, 5, 100, 32
    g = torch.Generator(device="cuda").manual_seed(5)
    x = synthetic.images(B, seed=77).flatten(1)
    yhat = torch.softmax(torch.randn(K, B, C, generator=g, device="cuda"), -1)
    noise = torch.randn(K, T, B, C, generator=g, device="cuda")
    eng.encode(x)
    y0 = eng.sample(yhat, yhat, noise)
    assert torch.isfinite(y0).all()
    # 1. determinism: replaying the graph and re-encoding give bitwise identical results
    assert torch.equal(eng.sample(yhat, yhat, noise), y0)
    eng.encode(x)
    assert torch.equal(eng.sample(yhat, yhat, noise), y0)
    assert torch.equal(eng.sample(yhat, yhat, noise, use_graph=False), y0)
    # 2. images are independent: permuting the batch permutes the outputs, bitwise (each output element is
    #    its own fixed-order dot product regardless of the row's position in a tile)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    eng.encode(x[perm])
    yp = eng.sample(yhat[:, perm], yhat[:, perm], noise[:, :, perm])
    assert torch.equal(yp, y0[:, perm])
    # 3. members are independent: one member at a time == all five in one launch.  The two launches use different
    #    kernel geometries (NF=1/U=2 vs NF=5/U=1: the k-chunks are dealt to the 16 waves differently), so the fp32
    #    summation order differs: equal to rounding, not bitwise.
    eng.encode(x)
    for k in (0, 3):
        yk = eng.sample(yhat[k:k + 1], yhat[k:k + 1], noise[k:k + 1], member0=k, n_members=1)
        assert (yk[0] - y0[k]).abs().max() < 1e-4 * max(1.0, float(y0[k].abs().max())), k
    # 4. Monte-Carlo rows: two trials fed the same draws give bitwise the same sample as each other, and the
    #    single-trial result to rounding (M = 64 rows selects the MT = 4 kernel geometry)
    nz2 = torch.cat([noise, noise], dim=2)
    y2 = eng.sample(yhat, yhat, nz2, mc=2)
    assert torch.equal(y2[:, :B], y2[:, B:])
    assert (y2[:, :B] - y0).abs().max() < 1e-4 * max(1.0, float(y0.abs().max()))
    # 5. aggregation: probabilities are a distribution; vote is in range
    prob, vote, probs = ops.aggregate(y0.reshape(K, B, C).contiguous(), 0.1737, return_probs=True)
    assert torch.allclose(prob.sum(-1), torch.ones(B, device="cuda"), atol=1e-6)
    assert int(vote.min()) >= 0 and int(vote.max()) < C and probs.shape == (K, B, C)


def test_fullsize_linearity_of_the_encoder_first_layer(engine5):
    """encoder_x.0 is linear before its BatchNorm/softplus: check the 150528-wide split-K GEMM on unit vectors
    against the weight columns themselves (exact: one non-zero product per output)."""
    from nested_diffusion_amd import ops, synthetic
    w = synthetic.cond_model_state(D, 64, 64, C, 2, seed=3)["encoder_x.0.weight"]      # [64, D]
    cols = torch.tensor([0, 1, 15, 16, 4095, 75263, 150527])
    x = torch.zeros(len(cols), D, device="cuda")
    x[torch.arange(len(cols)), cols] = 1.0
    out = ops.linear(x, w)
    assert torch.equal(out, w[:, cols.cuda()].T.contiguous())


def test_config_dims_one_member_vs_oracle_T8():
    """K=1, T=8, B=3 at config dims against the CPU oracle (seconds on the host)."""
    from nested_diffusion_amd.engine import EnsembleEngine
    T, B = 8, 3
    p = ref_cpu.init_cond_model_params(D, H, F, C, T, True, seed=321)
    eng = EnsembleEngine(C, D, H, F, T, n_members=1, max_batch=B)
    eng.load_member(0, p)
    alphas, omabs = ref_cpu.schedule_tables("linear", T, 1e-4, 0.02)
    eng.set_schedule(alphas, omabs)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, D, generator=g)
    yhat = torch.softmax(torch.randn(B, C, generator=g), -1)
    noise = torch.randn(T, B, C, generator=g)
    eng.encode(x)
    y0 = eng.sample(yhat[None].cuda(), yhat[None].cuda(), noise[None].cuda())[0].cpu()
    ref = ref_cpu.p_sample_loop(p, x, yhat, yhat, T, alphas, omabs, noise)
    assert (y0 - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max())
    pr, pr_ref = ref_cpu.convert_to_prob(y0, 0.1737), ref_cpu.convert_to_prob(ref, 0.1737)
    assert (pr - pr_ref).abs().max() < 1e-3
