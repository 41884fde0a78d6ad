"""Standalone HIP operators (through the C ABI) against plain torch fp32 on the CPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _close(got, ref, tol):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,K,N,act", [(1, 64, 16, None), (3, 128, 2, None), (32, 4096, 4096, "softplus"),
                                       (32, 2048, 128, "relu"), (17, 160, 100, "relu"), (70, 256, 48, None),
                                       (5, 16, 7, "gelu")])
def test_linear_fused_skinny(M, K, N, act):
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    s = torch.rand(N, generator=g) + 0.5
    out = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=act, scale=s.cuda())
    ref = s * (x.double() @ w.double().T).float() + b
    ref = {None: lambda v: v, "softplus": F.softplus, "relu": F.relu, "gelu": F.gelu}[act](ref)
    _close(out, ref, 2e-5)
    out2 = ops.linear(x.cuda(), w.cuda(), None, act=None)                 # no scale / shift
    _close(out2, (x.double() @ w.double().T).float(), 2e-5)


@pytest.mark.parametrize("M,K,N,act", [(129, 64, 16, None), (640, 4096, 4096, "softplus"), (1400, 4096, 2048, "relu"), (200, 160, 130, "gelu"),
                                       (257, 48, 7, None), (1000, 1024, 1000, "relu"), (300, 16, 300, None), (2048, 512, 136, "softplus")])
def test_linear_large_m_tiled(M, K, N, act):
    """More than 128 rows: nd_linear runs the LDS-tiled k_cond_gemm (MODE 0, row-major output), with ragged row / column
    tiles, N not a multiple of 4, a single k-step (K = 16), odd k-step counts, and the k-split tail + fixup."""
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    s = torch.rand(N, generator=g) + 0.5
    out = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=act, scale=s.cuda())
    ref = s * (x.double() @ w.double().T).float() + b
    ref = {None: lambda v: v, "softplus": F.softplus, "relu": F.relu, "gelu": F.gelu}[act](ref)
    _close(out, ref, 2e-5)
    assert torch.equal(ops.linear(x.cuda(), w.cuda(), b.cuda(), act=act, scale=s.cuda()), out)     # reproducible


@pytest.mark.parametrize("M,K,N", [(4, 16384, 64), (32, 150528, 256), (2, 32768 + 16, 130), (33, 20000 - 16 * 2, 128)])
def test_linear_splitk(M, K, N):
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(K + N)
    x = torch.rand(M, K, generator=g)
    w = (torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5
    b = torch.randn(N, generator=g)
    out = ops.linear(x.cuda(), w.cuda(), b.cuda(), act="relu")
    ref = F.relu((x.double() @ w.double().T).float() + b)
    _close(out, ref, 2e-5)


@pytest.mark.parametrize("M,K,N,act,res", [(6272, 768, 768, None, True), (392, 768, 2304, None, False),
                                           (6250, 768, 3070, "gelu", True), (6272, 3072, 768, None, True),   # k-split tail tiles

                                           (200, 64, 256, "gelu", False), (130, 48, 70, None, True), (1, 16, 1, "relu", False)])
def test_gemm_bias_act(M, K, N, act, res):
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    out = ops.gemm_bias_act(x.cuda(), w.cuda(), b.cuda(), act=act, residual=r.cuda() if res else None)
    ref = (x.double() @ w.double().T).float() + b
    ref = {None: lambda v: v, "relu": F.relu, "gelu": F.gelu}[act](ref)
    if res:
        ref = ref + r
    _close(out, ref, 2e-5)


@pytest.mark.parametrize("rows,K", [(6272, 768), (130, 64), (1, 32), (17, 3072)])
def test_split_rows_is_exact(rows, K):
    """frag32b3 (csrc/nd_b9.hpp): an fp32 value IS the sum of its three bf16 pieces -- bit for bit, including values whose low
    pieces vanish, negative zero aside -- so nothing is rounded on the way onto the bf16 matrix pipe."""
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, K, generator=g) * torch.exp(4 * torch.randn(rows, 1, generator=g))
    x[0, :8] = torch.tensor([0.0, 1.0, -1.0, 2.0 ** -100, 1.0 + 2.0 ** -23, 3.0e38, -7.5e-30, 1.0 / 3.0])
    x[0, 8:11] = torch.tensor([float("inf"), float("-inf"), 3.4e38])      # infinities stay infinities (3.4e38 rounds to an infinite bf16)
    s = ops.split_rows(x.cuda())
    assert s.data.numel() == ((rows + 15) // 16) * (K // 32) * 3072
    back = ops.join_rows(s).cpu()
    want = x.clone(); want[0, 10] = float("inf")
    assert torch.equal(back, want)
    x[0, 11] = float("nan")
    assert torch.isnan(ops.join_rows(ops.split_rows(x.cuda()))[0, 11])


@pytest.mark.parametrize("M,K,N,act,res", [(6272, 768, 2304, None, False), (6272, 768, 768, None, True), (6272, 768, 3072, "gelu", False),
                                           (6272, 3072, 768, None, True),                      # 8-wave shape, k-split tail + fixup
                                           (392, 768, 2304, None, False), (6250, 768, 3040, "gelu", True),   # ragged M and N edges
                                           (130, 64, 70, None, True), (1, 32, 1, "relu", False), (200, 2048, 96, "gelu", True)])
def test_gemm_split_exact_products_on_the_bf16_pipe(M, K, N, act, res):
    """nd_gemm_split against an fp64 product of the same fp32 inputs at the tolerance of the f32-MFMA kernel's test (2e-5), with
    and without the k-split workspace, reproducible run to run; its frag32b3 output is exactly the fp32 output."""
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    ws, xs = ops.split_rows(w.cuda()), ops.split_rows(x.cuda())
    rc = r.cuda() if res else None
    out = ops.gemm_split(xs, ws, b.cuda(), act=act, residual=rc)
    ref = (x.double() @ w.double().T).float() + b
    ref = {None: lambda v: v, "relu": F.relu, "gelu": F.gelu}[act](ref)
    if res:
        ref = ref + r
    _close(out, ref, 2e-5)
    assert torch.equal(out, ops.gemm_split(xs, ws, b.cuda(), act=act, residual=rc))
    whole = ops.gemm_split(xs, ws, b.cuda(), act=act, residual=rc, use_workspace=False)
    _close(whole, ref, 2e-5)
    if N % 32 == 0:
        o2, osp = ops.gemm_split(xs, ws, b.cuda(), act=act, residual=rc, want_out=True, want_split=True)
        assert torch.equal(o2, out) and torch.equal(ops.join_rows(osp), out)
        only = ops.gemm_split(xs, ws, b.cuda(), act=act, residual=rc, want_out=False, want_split=True)
        assert torch.equal(ops.join_rows(only), out)


@pytest.mark.parametrize("rows,dim", [(6272, 768), (5, 768), (70, 1024), (100, 2048), (64, 64), (6257, 128)])
def test_layernorm_split_is_the_layernorm(rows, dim):
    """nd_layernorm_split (both forms: 16 rows per workgroup through LDS, and the direct one) writes exactly nd_layernorm's values;
    rows of the last 16-row block past `rows` read back as zeros in the LDS form."""
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(rows + dim)
    x = (torch.randn(rows, dim, generator=g) * 3 + 1).cuda()
    w, b = torch.randn(dim, generator=g).cuda(), torch.randn(dim, generator=g).cuda()
    assert torch.equal(ops.join_rows(ops.layernorm_split(x, w, b, 1e-6)), ops.layernorm(x, w, b, 1e-6))


def test_attention_and_patchify_split_outputs_are_exact():
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, N, heads in ((3, 196, 12), (2, 5, 2), (1, 197, 12)):
        qkv = torch.randn(B * N, 3 * heads * 64, generator=g).cuda()
        assert torch.equal(ops.join_rows(ops.attention_split(qkv, B, N, heads)), ops.attention(qkv, B, N, heads))
    for B, C, H, p in ((3, 3, 224, 16), (2, 3, 32, 16), (1, 2, 16, 8), (2, 2, 16, 4), (5, 3, 48, 16)):     # p = 4: element-wise form
        img = torch.rand(B, C, H, H, generator=g).cuda()
        assert torch.equal(ops.join_rows(ops.patchify_split(img, p)), ops.patchify(img, p))


def test_gemm_split_argument_checks():
    from nested_diffusion_amd import _lib, ops
    lib = _lib.load()
    assert lib.nd_split_bytes(16, 48) == 0 and lib.nd_split_bytes(0, 32) == 0 and lib.nd_split_bytes(17, 64) == 2 * 2 * 3072
    x = ops.split_rows(torch.randn(32, 64).cuda())
    w = ops.split_rows(torch.randn(48, 64).cuda())
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(32, 48, device="cuda")
    assert lib.nd_gemm_split(_lib.ptr(x.data), _lib.ptr(w.data), None, None, None, None, 32, 64, 48, 0, None, 0, st) != 0      # no output at all
    assert lib.nd_gemm_split(_lib.ptr(x.data), _lib.ptr(w.data), None, None, _lib.ptr(out), _lib.ptr(out), 32, 64, 48, 0, None, 0, st) != 0
    assert b"N % 32" in lib.nd_last_error()                                                                              # split output needs N % 32 == 0
    assert lib.nd_gemm_split(_lib.ptr(x.data), _lib.ptr(w.data), None, None, _lib.ptr(out), None, 32, 48, 48, 0, None, 0, st) != 0   # K % 32
    with pytest.raises(ValueError):
        ops.SplitMatrix(4, 40, "cuda")


def test_gemm_without_workspace_matches_split_path():
    """nd_gemm_bias_act with a NULL workspace computes every tile whole: same values up to summation order."""
    from nested_diffusion_amd import _lib, ops
    lib = _lib.load()
    M, K, N = 6272, 768, 768
    assert lib.nd_gemm_workspace_bytes(M, K, N, 0) > 0                     # this shape takes the k-split tail
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    a = ops.gemm_bias_act(x, w, b)
    out = torch.empty(M, N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.nd_gemm_bias_act(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), M, K, N, 0, 0, None, 0, st), "gemm")
    torch.cuda.synchronize()
    assert (a - out).abs().max().item() < 2e-5
    assert torch.equal(a, ops.gemm_bias_act(x, w, b))                    # reproducible run to run


@pytest.mark.parametrize("rows,dim", [(5, 768), (197 * 3, 768), (7, 64), (3, 1024), (2, 2048)])
def test_layernorm(rows, dim):
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, dim, generator=g) * 3 + 1
    w, b = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    out = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6)
    _close(out, F.layer_norm(x, (dim,), w, b, 1e-6), 1e-5)


@pytest.mark.parametrize("B,N,heads", [(2, 196, 12), (1, 197, 12), (3, 16, 2), (2, 5, 1), (1, 256, 3), (2, 50, 4)])
def test_attention(B, N, heads):
    from nested_diffusion_amd import ops
    d = 64
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 3 * heads * d, generator=g)
    out = ops.attention(qkv.cuda(), B, N, heads)
    t = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    a = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1)
    ref = (a @ v).transpose(1, 2).reshape(B * N, heads * d)
    _close(out, ref, 1e-5)


@pytest.mark.parametrize("B,N,heads", [(2, 196, 12), (3, 16, 2), (1, 4, 2), (2, 100, 2), (1, 256, 4), (2, 36, 2), (5, 132, 2), (1, 64, 2), (32, 196, 12),
                                       (10, 196, 12)])          # B = 10: 558 tiles on 512 slots -> the qkv GEMM's HALF-TILE tail writes images too
def test_attention_on_the_bf16_pipe_from_qkv_images(B, N, heads):
    """timm Attention.forward (call sites classification_train_separately.py:339-340) with BOTH contractions on the bf16 matrix pipe, exact
    fp32 products: the qkv Linear writes the attention's operand images (q, k as frag32b3 blocks per head, v transposed with the keys of
    a 32-block in score-accumulator order; nd_gemm_split_qkv), k_attention_b9 consumes them.  Against torch in fp64 on the same fp32
    inputs (qkv Linear included), at the tolerance of the f32-input-MFMA kernel's test; against that kernel on the fp32 qkv; the split
    output is the exact image of the fp32 output.  N = 4 / 16 / 36: one partial fragment, one workgroup per head; 100 / 132 / 196:
    ragged last fragment and a half-empty last key block (never-written image rows behind it are poisoned with NaNs first);
    256: the largest N; B = 32 x 196 x 12: the conditioner's own shape."""
    from nested_diffusion_amd import _lib, ops
    E = heads * 64
    g = torch.Generator().manual_seed(N + B)
    x = torch.randn(B * N, E, generator=g)
    w = torch.randn(3 * E, E, generator=g) / E ** 0.5
    b = torch.randn(3 * E, generator=g) * 0.1
    assert ops.qkv_images_supported(N, heads)
    xs, ws = ops.split_rows(x.cuda()), ops.split_rows(w.cuda())
    # poison THE buffer the images land in (handed over explicitly: no reliance on the allocator recycling a block): rows / keys past N
    # are never written and must not reach the result.  0xFF bytes = bf16 NaNs in every piece.
    nbytes = _lib.load().nd_qkv_images_bytes(B, N, heads)
    buf = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda")
    img = ops.gemm_split_qkv(xs, ws, b.cuda(), B, N, heads, out=buf)
    assert img.data_ptr() == buf.data_ptr()
    if N % 16:                                                                            # a ragged last fragment: poison must survive in it
        assert (buf == 0xFF).any()
    out = ops.attention_images(img, B, N, heads)
    qkv64 = x.double() @ w.double().T + b.double()
    t = qkv64.reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    a = ((t[0] @ t[1].transpose(-2, -1)) * 0.125).softmax(-1)
    ref = (a @ t[2]).transpose(1, 2).reshape(B * N, E)
    _close(out, ref.float(), 1e-5)
    assert torch.isfinite(out).all()
    # the f32-input-MFMA kernel on the fp32 qkv of the same GEMM: the same arithmetic in another summation order
    qkv32 = ops.gemm_split(xs, ws, b.cuda())
    old = ops.attention(qkv32, B, N, heads)
    _close(out, old.cpu(), 1e-5)
    assert torch.equal(ops.join_rows(ops.attention_images(img, B, N, heads, want_split=True)), out)
    assert torch.equal(ops.attention_images(img, B, N, heads), out)                       # reproducible


def test_qkv_images_argument_checks():
    from nested_diffusion_amd import _lib, ops
    lib = _lib.load()
    assert lib.nd_qkv_images_supported(196, 12) == 1 and lib.nd_qkv_images_supported(197, 12) == 0        # the full forward's cls token
    assert lib.nd_qkv_images_supported(196, 1) == 0 and lib.nd_qkv_images_supported(260, 2) == 0
    assert lib.nd_qkv_images_bytes(32, 196, 12) == 32 * 12 * (4 * 13 + 4 * 7) * 3072
    x = ops.split_rows(torch.randn(2 * 197, 128).cuda())
    w = ops.split_rows(torch.randn(384, 128).cuda())
    buf = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.nd_gemm_split_qkv(_lib.ptr(x.data), _lib.ptr(w.data), None, _lib.ptr(buf), 2, 197, 2, 128, st) != 0
    assert b"qkv images" in lib.nd_last_error()
    assert lib.nd_attention_images(_lib.ptr(buf), _lib.ptr(buf), 0, 2, 197, 2, st) != 0
    assert lib.nd_attention_images(None, _lib.ptr(buf), 0, 2, 196, 2, st) != 0


def test_patchify_matches_conv2d():
    from nested_diffusion_amd import ops
    g = torch.Generator().manual_seed(0)
    img = torch.rand(2, 3, 64, 48, generator=g)
    w = torch.randn(32, 3, 16, 16, generator=g) / 28
    b = torch.randn(32, generator=g)
    cols = ops.patchify(img.cuda(), 16)
    out = ops.gemm_bias_act(cols, w.reshape(32, -1).cuda(), b.cuda())
    ref = F.conv2d(img, w, b, stride=16).flatten(2).transpose(1, 2).reshape(-1, 32)
    _close(out, ref, 1e-5)


def test_softmax_and_aggregate_vs_golden():
    from nested_diffusion_amd import ops
    z = np.load(os.path.join(G, "aggregation.npz"))
    for name in ("a0", "a1", "a2"):
        temp = float(z[name + "_temp"])
        s = torch.from_numpy(z[name + "_samples"])
        prob, vote, probs = ops.aggregate(s.cuda(), temp, return_probs=True)
        assert np.array_equal(vote.cpu().numpy(), z[name + "_vote"])          # integer result: exact
        assert np.abs(prob.cpu().numpy() - z[name + "_prob"]).max() < 2e-6
        assert np.abs(probs.cpu().numpy() - z[name + "_mutated"]).max() < 2e-6
    x = torch.randn(37, 2)
    _close(ops.softmax_rows(x.cuda()), torch.softmax(x, 1), 1e-6)
    x3 = torch.randn(5, 3) * 10
    _close(ops.softmax_rows(x3.cuda()), torch.softmax(x3, 1), 1e-6)


def test_ops_reject_cpu_tensors_and_bad_shapes():
    from nested_diffusion_amd import _lib, ops
    with pytest.raises(_lib.NdError):
        ops.linear(torch.zeros(2, 16), torch.zeros(4, 16).cuda())              # CPU tensor: no fallback
    with pytest.raises(_lib.NdError):
        ops.linear(torch.zeros(2, 24).cuda(), torch.zeros(4, 24).cuda())       # K not a multiple of 16
    with pytest.raises(_lib.NdError):
        ops.attention(torch.zeros(4, 3 * 32).cuda(), 1, 4, 1)                  # head dim 32
    with pytest.raises(ValueError):
        ops.linear(torch.zeros(2, 16).cuda(), torch.zeros(4, 32).cuda())


def test_sample_stats_and_report_vs_golden_and_oracle():
    """Reporting tail of test_atk: per-image PIW / variance over the samples, then accuracy, ECE and the per-class
    means -- against the reference's own functions (report.npz) and the oracle (ECE: torchmetrics semantics)."""
    from nested_diffusion_amd import ops
    z = np.load(os.path.join(G, "report.npz"))
    for n in ("r0", "r1", "r2"):
        probs = torch.from_numpy(z[n + "_probs"])
        mv, lab = torch.from_numpy(z[n + "_mv"]), torch.from_numpy(z[n + "_label"])
        piw, var = ops.sample_stats(probs.cuda())
        lo, hi = torch.quantile(probs, 0.025, dim=0), torch.quantile(probs, 0.975, dim=0)
        assert (piw.cpu() - (hi - lo)).abs().max() < 1e-6
        assert (var.cpu() - probs.var(dim=0)).abs().max() < 1e-6
        pm = probs.mean(0)
        rep = ops.report(piw, var, pm.cuda(), mv.cuda(), lab.cuda(), 0.1737, 10)
        assert abs(float(rep["accuracy"]) - float(ref_cpu.compute_accuracy(mv, lab))) < 1e-7
        for key, ref in (("piw_correct", z[n + "_piw_c"]), ("piw_incorrect", z[n + "_piw_i"]),
                         ("var_correct", z[n + "_var_c"]), ("var_incorrect", z[n + "_var_i"])):
            got = rep[key].numpy()
            assert np.array_equal(np.isnan(got), np.isnan(ref)), (n, key)
            assert np.nanmax(np.abs(got - ref), initial=0.0) < 2e-6, (n, key, got, ref)
        ece = ref_cpu.compute_ece_as_reference(pm, lab, 0.1737)
        assert abs(float(rep["ece"]) - float(ece)) < 2e-6
    # quantile interpolation edge cases: S = 1, S = 2, ties
    for S in (1, 2, 3):
        p = torch.tensor([0.25, 0.25, 0.75])[:S].reshape(S, 1, 1).repeat(1, 2, 2).contiguous()
        piw, var = ops.sample_stats(p.cuda())
        ref = torch.quantile(p, 0.975, dim=0) - torch.quantile(p, 0.025, dim=0)
        assert (piw.cpu() - ref).abs().max() < 1e-6


def test_perturbation_ops_vs_reference_goldens():
    """diffusion/utils.py perturbations: HIP kernels against outputs of the reference's own functions (perturb.npz);
    crop-and-resize against the oracle (torchvision's tensor Resize = bilinear interpolate, unpinned)."""
    import random
    from nested_diffusion_amd import perturb as P
    z = np.load(os.path.join(G, "perturb.npz"))
    x = torch.from_numpy(z["x"]).cuda()
    def close(got, ref, tol=2e-7):
        assert np.abs(got.cpu().numpy() - ref).max() <= tol, np.abs(got.cpu().numpy() - ref).max()
    close(P.add_noise(x, 0.3, z=torch.from_numpy(z["z"]).cuda()), z["noise_0p3"], 0)          # exact
    close(P.adjust_brightness(x, 0.4), z["bright_p0p4"], 0)
    close(P.adjust_brightness(x, -0.3), z["bright_m0p3"], 0)
    close(P.adjust_contrast(x, 1.7), z["contrast_1p7"], 5e-7)        # the per-image mean is a differently ordered sum
    close(P.adjust_contrast(x, 0.4), z["contrast_0p4"], 5e-7)
    close(P.down_up_sample(x, 2), z["downup_2"], 3e-7)
    close(P.down_up_sample(x, 3), z["downup_3"], 3e-7)
    random.seed(9)
    close(P.random_cover_new(x, (0.05, 2)), z["cover_0p05_2"], 0)    # same python-random picks as the reference
    corners = [(1, 0), (3, 2), (0, 4)]
    got = P.random_crop_and_resize(x, 0.25, corners=corners)
    ref = ref_cpu.crop_and_resize(x.cpu(), corners, int(20 * 0.75))
    close(got, ref.numpy(), 3e-7)
    torch.manual_seed(3)
    out = P.random_crop_and_resize(x, 0.25)
    assert out.shape == x.shape and torch.isfinite(out).all()
    # config-size image batch: shapes / ranges
    big = torch.rand(4, 3, 224, 224, device="cuda")
    assert float(P.adjust_contrast(big, 2.0).max()) <= 1.0 and float(P.adjust_brightness(big, -0.5).min()) >= 0.0
    d = P.down_up_sample(big, 4)
    assert d.shape == big.shape
    assert (d.cpu() - ref_cpu.down_up_sample(big.cpu(), 4)).abs().max() < 3e-7


def test_linear_random_shapes_cover_launch_geometry():
    """40 random (M, K, N) shapes in both operand dtypes: every combination of fragments-per-workgroup (1..6, with and without
    a remainder), leftover k-chunks, k-slabs and row groups the launcher can pick must give the same numbers as torch."""
    import random
    from nested_diffusion_amd import ops
    rnd = random.Random(7)
    for it in range(40):
        half = it % 2 == 1
        kq = 32 if half else 16
        M = rnd.choice([1, 2, 7, 16, 17, 31, 32, 33, 48, 64, 65, 70])
        K = kq * rnd.randint(1, 80) if it % 5 else kq * rnd.randint(1024 // kq, 40000 // kq)
        N = rnd.choice([1, 2, 15, 16, 17, 100, 255, 256, 257, 600, 1000, 1536, 4096]) if K < 5000 else rnd.choice([3, 64, 130, 256])
        g = torch.Generator().manual_seed(it)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        s = torch.rand(N, generator=g) + 0.5
        pw = ops.PackedWeight(w.cuda(), dtype="f16" if half else "f32")
        out = ops.linear(x.cuda(), pw, b.cuda(), act="relu", scale=s.cuda())
        xr, wr = (x.half().double(), w.half().double()) if half else (x.double(), w.double())
        ref = F.relu(s * (xr @ wr.T).float() + b)
        err = (out.cpu().double() - ref.double()).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (it, M, K, N, half, err)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_linear_four_and_five_row_fragments_with_up_to_six_weight_fragments(dtype):
    """33..128 rows: a k_skinny workgroup keeps four or five row fragments (five where that saves a pass over the weights: 65..80
    rows in one pass) against up to six weight fragments, its cross-wave sum in dynamic LDS (96 / 120 KiB).  Wide N so that a
    single-member launch really gets five or six fragments per workgroup, and a split-K shape (K >= 16384); against fp64."""
    import ctypes as C
    from nested_diffusion_amd import _lib, ops
    lib = _lib.load()
    half = dtype == "f16"
    out6 = (C.c_int * 6)()
    assert lib.nd_skinny_plan(64, 24576, 70, 1, 1 if half else 0, 0, out6) == 0
    assert out6[1] == 1 and out6[3] == 6            # one pass over the weights for 70 rows, six fragment slots per workgroup
    assert lib.nd_skinny_plan(64, 24576, 64, 1, 1 if half else 0, 0, out6) == 0 and out6[1] == 1 and out6[3] == 6
    for M in (40, 64, 70, 80, 100, 128):
        for K, N in ((64, 24576), (16384, 1536), (96, 20000)):
            g = torch.Generator().manual_seed(M + K)
            x = torch.randn(M, K, generator=g)
            w = torch.randn(N, K, generator=g) / K ** 0.5
            b = torch.randn(N, generator=g)
            out = ops.linear(x.cuda(), ops.PackedWeight(w.cuda(), dtype=dtype), b.cuda(), act="softplus")
            xr, wr = (x.half().double(), w.half().double()) if half else (x.double(), w.double())
            ref = F.softplus((xr @ wr.T).float() + b)
            err = (out.cpu().double() - ref.double()).abs().max().item()
            assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (M, K, N, err)


def test_gelu_exact_erf_accuracy_through_an_identity_gemm():
    """timm's Mlp uses the exact-erf GELU (fc1 epilogue).  nd_erf is a branch-free < 1 ulp erf (csrc/nd_common.hpp); a GEMM against the
    identity (exact products, exact sums) exposes it element by element: against torch in fp64 the error stays within fp32 rounding of
    the three steps (argument product, erf, 1 + erf) -- absolute 2e-7 max(1, |v|) -- over [-9, 9], denormal-small and large arguments."""
    from nested_diffusion_amd import ops
    K = 64
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.linspace(-9, 9, 4096 * K).reshape(4096, K), torch.randn(2048, K, generator=g) * 1.5,
                   torch.tensor([[0.0, -0.0, 1e-30, -1e-30, 0.927734375 * 2 ** 0.5, -0.927734375 * 2 ** 0.5, 30.0, -30.0, 1e4, -1e4] + [0.5] * (K - 10)])])
    eye = torch.eye(K)
    for out in (ops.gemm_bias_act(x.cuda(), eye.cuda(), None, act="gelu"),                       # f32-input MFMA kernel's epilogue
                ops.gemm_split(x.cuda(), ops.split_rows(eye.cuda()), None, act="gelu")):        # bf16-pipe kernel's epilogue (fc1)
        ref = torch.nn.functional.gelu(x.double())
        err = (out.cpu().double() - ref).abs()
        assert torch.isfinite(out).all()
        assert bool((err <= 2e-7 * x.double().abs().clamp(min=1.0)).all()), float(err.max())
