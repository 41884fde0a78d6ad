"""CPU-only checks: the C-ABI library loads and exports every symbol include/nested_diffusion.h declares
(no compute calls), the ctypes table matches the header, and the host-side logic (CLI, config, sharding)."""
import argparse
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nested_diffusion.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = header_functions()
    for must in ("nd_create", "nd_load_member", "nd_encode", "nd_sample", "nd_eps_theta", "nd_p_sample", "nd_linear",
                 "nd_gemm_bias_act", "nd_layernorm", "nd_attention", "nd_aggregate", "nd_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from nested_diffusion_amd import _lib, build
    build.build()                                   # hipcc cross-compiles for gfx950 without a GPU
    lib = _lib.load()
    names = header_functions()
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)
    for n in names:
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.nd_version()


def test_abi_argument_validation_without_gpu():
    """Pure host-side argument checks: they return error codes before any HIP call."""
    from nested_diffusion_amd import _lib
    lib = _lib.load()
    cfg = _lib.NdConfig(2, 50, 64, 64, 10, 1, 4, 4)                 # data_dim not a multiple of 16
    assert lib.nd_workspace_bytes(ctypes.byref(cfg)) == 0
    h = ctypes.c_void_p()
    assert lib.nd_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"data_dim" in lib.nd_last_error()
    # library limits the reference does not have, stated in nd_config's comment: y_dim <= 8, n_members <= 255
    for bad_cfg, word in ((_lib.NdConfig(9, 48, 64, 64, 10, 1, 4, 4), b"y_dim must be in [1,8]"),
                          (_lib.NdConfig(2, 48, 64, 64, 10, 256, 4, 4), b"n_members")):
        assert lib.nd_create(ctypes.byref(bad_cfg), ctypes.byref(h)) != 0
        assert word in lib.nd_last_error(), lib.nd_last_error()
    assert lib.nd_create(ctypes.byref(_lib.NdConfig(8, 48, 64, 64, 10, 255, 4, 4)), ctypes.byref(h)) == 0
    assert lib.nd_destroy(h) == 0
    good = _lib.NdConfig(2, 150528, 4096, 4096, 100, 5, 32, 32)
    nbytes = lib.nd_workspace_bytes(ctypes.byref(good))
    # packed weights (5 x 2.74 GB) + tables + activations: 13..16 GB at the headline config
    assert 13e9 < nbytes < 17e9, nbytes
    assert lib.nd_create(ctypes.byref(good), ctypes.byref(h)) == 0
    assert lib.nd_encode(h, 0, 1, None, 4, None) != 0              # workspace not bound
    # form of the T-step loop (host state only here): 0 = per-step kernel nodes, 1 = one launch; nothing else; a fresh handle reports
    # the graph form, and the status query needs a bound workspace
    assert lib.nd_loop_form(h) == 0
    assert lib.nd_set_loop_form(h, 1, ctypes.c_float(12.5)) == 0 and lib.nd_set_loop_form(h, 0, ctypes.c_float(-1.0)) == 0
    assert lib.nd_set_loop_form(h, 2, ctypes.c_float(0.0)) != 0 and b"mode" in lib.nd_last_error()
    assert lib.nd_set_loop_form(h, 1, ctypes.c_float(2e6)) != 0 and lib.nd_set_loop_form(None, 1, ctypes.c_float(0.0)) != 0
    assert lib.nd_persist_status(h, 0) < 0 and b"workspace" in lib.nd_last_error()
    assert lib.nd_set_input_flag(h, None) < 0 and b"workspace" in lib.nd_last_error()          # needs a bound workspace (it resets a device counter)
    assert lib.nd_skinny_row_fragments(32) == 2 and lib.nd_skinny_row_fragments(70) == 5 and lib.nd_skinny_row_fragments(64) == 4
    assert lib.nd_skinny_row_fragments(16) == 1 and lib.nd_skinny_row_fragments(0) < 0
    assert lib.nd_destroy(h) == 0
    assert lib.nd_linear(None, None, None, None, None, 1, 16, 1, 0, 0, None, 0, None) != 0
    assert lib.nd_packed_bytes(3, 32, 0) == 16 * 32 * 4            # rows padded to 16
    assert lib.nd_packed_bytes(3, 30, 0) == 0
    assert lib.nd_packed_bytes(3, 64, 1) == 16 * 64 * 2            # fp16 image
    assert lib.nd_packed_bytes(3, 48, 1) == 0                      # fp16 needs K % 32 == 0
    assert lib.nd_packed_bytes(3, 32, 7) == 0                      # unknown dtype
    # frag32b3 images (csrc/nd_b9.hpp) and the split-GEMM's launch plan: host-side arithmetic only
    assert lib.nd_split_bytes(6272, 768) == 392 * 24 * 3072 and lib.nd_split_bytes(17, 64) == 2 * 2 * 3072
    assert lib.nd_split_bytes(16, 48) == 0 and lib.nd_split_bytes(0, 32) == 0
    assert lib.nd_gemm_split_workspace_bytes(6272, 3072, 768) > 0          # fc2: 294 tiles of 128 x 128 on 256 CUs: the remainder is cut along K
    assert lib.nd_gemm_split_workspace_bytes(6272, 768, 2304) == 0         # qkv: K = 768 tiles are short against the fixup: left whole
    assert lib.nd_gemm_split_workspace_bytes(64, 768, 768) == 0 and lib.nd_gemm_split_workspace_bytes(64, 48, 768) == 0
    assert lib.nd_gemm_split(None, None, None, None, None, None, 1, 32, 1, 0, None, 0, None) != 0
    assert lib.nd_split_rows(None, None, 16, 32, None) != 0 and lib.nd_layernorm_split(None, None, None, None, 4, 64, 1e-6, None) != 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nested_diffusion_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NdError, match="no CPU fallback"):
        _lib.load()


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    from nested_diffusion_amd import _lib, ops
    from nested_diffusion_amd.engine import EnsembleEngine
    with pytest.raises(_lib.NdError):
        EnsembleEngine(2, 48, 64, 64, 10)
    with pytest.raises(_lib.NdError):
        ops.linear(torch.zeros(2, 16), torch.zeros(4, 16))
    with pytest.raises(_lib.NdError):
        ops.aggregate(torch.zeros(3, 2, 2), 0.1737)


def test_cli_accepts_the_reference_invocation():
    """Flags of diffusion/testing_scripts/test.sh:24."""
    from nested_diffusion_amd.main import build_parser, dict2namespace
    argv = ("--test --device 0 --thread 8 --loss card_onehot_conditional --config configs/chest_x_ray.yml "
            "--exp ./results/chest_x_ray/run --doc chest_x_ray --n_splits 1 --noise_perturbation 0 --low_resolution 0 "
            "--brightness 0 --contrast 1 --crop 0 --attack_name None --eps 0 --ni --preprocess grayscaled").split()
    a = build_parser().parse_args(argv)
    assert a.test and a.ni and a.loss == "card_onehot_conditional" and a.preprocess == "grayscaled"
    assert a.timesteps is None and a.mc_trials == 20 and a.covered == (0.0, 0.0)
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--test", "--doc", "x"])          # --config / --preprocess are required
    ns = dict2namespace({"a": {"b": 1, "c": [[1, 2]]}, "d": "x"})
    assert ns.a.b == 1 and ns.a.c[0][1] == 2 and ns.d == "x"


def test_schedule_and_temperature_match_pinned_tables():
    import numpy as np
    from nested_diffusion_amd.diffusion_utils import make_beta_schedule
    from nested_diffusion_amd.runner import temperature_for
    z = np.load(os.path.join(ROOT, "tests", "golden", "schedule.npz"))
    for T in (10, 100, 1000):
        assert np.array_equal(make_beta_schedule("linear", T, 1e-4, 0.02).float().numpy(), z[f"betas_{T}"])
    for sched in ("cosine", "cosine_anneal", "quad", "sigmoid", "const", "jsd"):
        assert np.array_equal(make_beta_schedule(sched, 50, 1e-4, 0.02).float().numpy(), z[f"betas_{sched}_50"])
    with pytest.raises(ValueError):
        make_beta_schedule("nope")
    assert temperature_for("ChestXRay") == 0.1737 and temperature_for("ISICSkinCancerAtkPGD") == 0.3162
    with pytest.raises(NotImplementedError):
        temperature_for("MNIST")


def test_shard_bounds_cover_and_balance():
    from nested_diffusion_amd.dist import shard_bounds
    for n in (0, 1, 7, 32, 70, 256):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_synthetic_loader_contract():
    from nested_diffusion_amd.data import SyntheticLoader, get_test_loader
    batches = list(SyntheticLoader(2, 3, 2, seed=5, size=32))
    assert len(batches) == 2 and batches[0][0].shape == (3, 3, 32, 32) and batches[0][1].dtype == torch.int64
    assert float(batches[0][0].min()) >= 0.0 and float(batches[0][0].max()) < 1.0
    again = list(SyntheticLoader(2, 3, 2, seed=5, size=32))
    assert torch.equal(batches[1][0], again[1][0])
    cfg = argparse.Namespace(model=argparse.Namespace(data_dim=3 * 32 * 32), testing=argparse.Namespace(batch_size=2),
                             data=argparse.Namespace(num_classes=2))
    x, y = next(iter(get_test_loader(argparse.Namespace(synthetic_batches=1, seed=3), cfg)))
    assert x.shape == (2, 3, 32, 32)


def test_image_folder_pipeline(tmp_path):
    """ImageFolder ordering + Grayscale(3)/Resize(224)/ToTensor, drop_last loader (dataset_helper/chest_x_ray_dataset.py,
    classification_train_separately.py:674-681) on a tiny PNG/JPEG tree."""
    import numpy as np
    from PIL import Image
    from nested_diffusion_amd.data import get_test_loader, ImageFolderDataset
    rng = np.random.default_rng(0)
    root = tmp_path / "data"
    for split in ("testing", "validation"):
        for cls, n in (("PNEUMONIA", 3), ("NORMAL", 2)):
            d = root / split / cls
            d.mkdir(parents=True)
            for i in range(n):
                arr = rng.integers(0, 255, size=(37 + i, 53, 3), dtype=np.uint8)
                Image.fromarray(arr, "RGB").save(d / f"img_{i}.png")
        (root / split / "NORMAL" / "notes.txt").write_text("not an image")
    args = argparse.Namespace(synthetic_batches=0, preprocess="grayscaled", seed=0)
    cfg = argparse.Namespace(data=argparse.Namespace(dataset="ChestXRay", dataroot=str(root), num_workers=0, num_classes=2),
                             testing=argparse.Namespace(batch_size=2), model=argparse.Namespace(data_dim=150528))
    loader = get_test_loader(args, cfg)
    batches = list(loader)
    assert len(batches) == 2                                      # 5 images, batch 2, drop_last -> 2 batches
    x, y = batches[0]
    assert x.shape == (2, 3, 224, 224) and x.dtype == torch.float32 and y.tolist() == [0, 0]   # NORMAL sorts first
    assert torch.equal(x[:, 0], x[:, 1]) and torch.equal(x[:, 1], x[:, 2])                     # grayscale replicated
    assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    # value check against the PIL calls written out by hand
    ds = loader.dataset
    assert ds.classes == ["NORMAL", "PNEUMONIA"] and len(ds) == 5
    img = Image.open(ds.samples[0][0]).convert("RGB").convert("L")
    g = np.array(img)
    ref = np.array(Image.fromarray(np.dstack([g, g, g]), "RGB").resize((224, 224), Image.BILINEAR)).astype(np.float32) / 255
    assert np.array_equal(ds[0][0].numpy(), ref.transpose(2, 0, 1))
    # validation split + standardized preprocess
    cfg.data.dataset = "ChestXRayValidate"; args.preprocess = "standardized"
    xv, _ = next(iter(get_test_loader(args, cfg)))
    assert xv.shape == (2, 3, 224, 224) and float(xv.min()) < 0.0
    with pytest.raises(FileNotFoundError):
        ImageFolderDataset(str(root / "nope"), "ChestXRay", "grayscaled")
    cfg.data.dataset = "MNIST"
    with pytest.raises(NotImplementedError):
        get_test_loader(args, cfg)


def test_image_folder_transforms_against_hand_arithmetic(tmp_path):
    """Independent expectations for the reference's test-time transforms (dataset_helper/chest_x_ray_dataset.py:31-51; torchvision
    on PIL images), written out in numpy instead of calling PIL again:
      Grayscale(3): ITU-R 601-2 luma in PIL's 16-bit fixed point, L = (19595 R + 38470 G + 7471 B + 32768) >> 16, replicated;
      Resize((224,224)) of a 448x448 image: PIL BILINEAR is a triangle filter whose support scales with the reduction, i.e. taps
      (1, 3, 3, 1)/8 per axis at a factor of 2 (each pass rounds to 8 bits with fixed-point taps: tolerance 1.5 grey levels, mean error < 0.35);  ToTensor: /255."""
    import numpy as np
    from PIL import Image
    from nested_diffusion_amd.data import ImageFolderDataset
    rng = np.random.default_rng(7)
    root = tmp_path / "testing"
    (root / "NORMAL").mkdir(parents=True)
    same = rng.integers(0, 256, size=(224, 224, 3), dtype=np.uint8)          # already 224 x 224: the resize is the identity
    big = rng.integers(0, 256, size=(448, 448, 3), dtype=np.uint8)
    Image.fromarray(same, "RGB").save(root / "NORMAL" / "a_same.png")
    Image.fromarray(big, "RGB").save(root / "NORMAL" / "b_big.png")
    ds = ImageFolderDataset(str(root), "ChestXRay", "grayscaled")
    x_same, x_big = ds[0][0].numpy(), ds[1][0].numpy()
    def luma(a):
        a = a.astype(np.int64)
        return ((19595 * a[..., 0] + 38470 * a[..., 1] + 7471 * a[..., 2] + 32768) >> 16).astype(np.float64)
    exp_same = luma(same) / 255.0
    for c in range(3):
        assert np.array_equal(x_same[c].astype(np.float64), exp_same.astype(np.float32).astype(np.float64)), c
    g = luma(big)
    taps = np.array([1, 3, 3, 1], dtype=np.float64) / 8.0

    def down2(a, axis):                                                   # output i takes inputs 2i-1 .. 2i+2 (edges: renormalised)
        a = np.moveaxis(a, axis, 0)
        n = a.shape[0] // 2
        out = np.zeros((n,) + a.shape[1:])
        for i in range(n):
            idx = np.arange(2 * i - 1, 2 * i + 3)
            ok = (idx >= 0) & (idx < a.shape[0])
            wgt = taps[ok] / taps[ok].sum()
            out[i] = np.tensordot(wgt, a[idx[ok]], axes=(0, 0))
        return np.moveaxis(out, 0, axis)
    exp_big = down2(np.rint(down2(g, 1)), 0) / 255.0                      # PIL: horizontal pass (rounded to 8 bits), then vertical
    assert np.abs(x_big[0].astype(np.float64) - exp_big).max() <= 1.5 / 255
    assert np.abs(x_big[0].astype(np.float64) - exp_big).mean() < 0.35 / 255
    assert np.array_equal(x_big[0], x_big[1]) and np.array_equal(x_big[1], x_big[2])
    # standardized: no grayscale; (x/255 - mean) / std with the reference's pre-calculated constants (:73-74)
    xs = ImageFolderDataset(str(root), "ChestXRay", "standardized")[0][0].numpy()
    mean, std = np.array([0.5094, 0.5234, 0.5289]), np.array([0.2189, 0.2225, 0.2244])
    exp = (same.transpose(2, 0, 1).astype(np.float32) / 255 - mean[:, None, None].astype(np.float32)) / std[:, None, None].astype(np.float32)
    assert np.abs(xs - exp).max() < 1e-6


def test_launch_plans_on_the_host():
    """The launch-plan helpers are host code: the headline shapes must give at most one 4-wave workgroup per CU, every
    fragment covered, every k-chunk covered, and the ViT GEMM must ask for a tail workspace only where a partial last round
    exists."""
    import ctypes as C
    from nested_diffusion_amd import _lib
    lib = _lib.load()

    def plan(K, N, M, nm, dtype=0, mode=0):
        out = (C.c_int * 6)()
        assert lib.nd_skinny_plan(K, N, M, nm, dtype, mode, out) == 0, lib.nd_last_error()
        return list(out)

    for dtype in (0, 1):
        gx, gy, gz, nf, cps, threads = plan(4096, 4096, 32, 5, dtype, 0)          # lin2 of K = 5 members
        assert threads == 256 and gy == 1 and gz == 1
        assert gx <= 256 and gx % 5 == 0 and nf * (gx // 5) >= 256 and nf <= 6      # one workgroup per CU, no member mixing
        assert cps == 4096 // (32 if dtype else 16)
    gx, gy, gz, nf, cps, _ = plan(150528, 4096, 32, 1, 0, 2)                        # mapping linear1: split-K fills the CUs
    assert gx * gy * gz <= 256 and gx * gy * gz >= 192 and nf >= 3 and cps * gz >= 150528 // 16
    gx, gy, gz, nf, cps, _ = plan(4096, 4096, 640, 5, 0, 0)                         # mc = 20: 40 row fragments = 8 passes of five
    assert gy == 8 and nf <= 6
    for M, passes in ((64, 1), (70, 1), (80, 1), (81, 2), (128, 2), (140, 2), (161, 3), (1400, 18)):
        # row fragments per pass: four, or five where that saves a whole pass over the weights (70 rows, the reference's batch: one)
        assert plan(4096, 4096, M, 5, 1, 0)[1] == passes, M
    gx, gy, gz, nf, cps, _ = plan(16, 7, 1, 1, 0, 0)                                # tiny
    assert (gx, gy, gz, nf) == (1, 1, 1, 1)
    out = (C.c_int * 6)()
    assert lib.nd_skinny_plan(48, 16, 1, 1, 1, 0, out) != 0                         # fp16 needs K % 32 == 0
    assert lib.nd_gemm_workspace_bytes(6272, 768, 768, 0) > 0                          # 588 tiles: 76 left over after 2 rounds
    assert lib.nd_gemm_workspace_bytes(8192, 4096, 4096, 0) == 0                       # 4096 tiles: whole rounds only
    assert lib.nd_gemm_workspace_bytes(8, 16, 8, 0) == 0
    assert lib.nd_gemm_workspace_bytes(6272, 768, 768, 1) > 0 and lib.nd_gemm_workspace_bytes(6272, 48, 768, 1) == 0   # fp16: K % 32


def test_step_block_dispatch_threshold_and_tile_plan():
    """nd_step_plan (host only): which kernel a ConditionalLinear block runs at M = B*mc rows.  Up to 128 rows the
    weight-streaming k_skinny (one or two 64-row passes), above that the LDS-tiled k_cond_gemm; fp16 operands always stream.
    The reference's own points: mc = 20 x B = 32 / 70 (classification_train_separately.py:770-771, configs/chest_x_ray.yml:66)."""
    import ctypes as C
    from nested_diffusion_amd import _lib
    lib = _lib.load()

    def plan(F, M, nm, dtype=0):
        out = (C.c_int * 8)()
        assert lib.nd_step_plan(F, M, nm, dtype, out) == 0, lib.nd_last_error()
        return dict(zip(("tile", "wgs", "whole", "rem", "split", "partials", "TM", "TN"), out))

    for M in (1, 32, 64, 70, 128):
        p = plan(4096, M, 5)
        assert p["tile"] == 0 and p["partials"] == 256, (M, p)              # one partial per 16 columns
    p = plan(4096, 129, 5)
    assert p["tile"] == 1 and p["TM"] == 2 and p["TN"] == 32 and p["partials"] == 64
    p = plan(4096, 640, 5)                                                      # mc = 20, B = 32, K = 5: 5 x 5 x 32 tiles
    assert (p["tile"], p["TM"], p["TN"], p["whole"], p["rem"], p["split"]) == (1, 5, 32, 768, 32, 8)
    assert p["wgs"] == 768 + 32 * 8 and p["rem"] * p["split"] <= 512            # slab workspace bound (32 MB)
    p = plan(4096, 1400, 5)                                                     # mc = 20, B = 70: 11 x 32 x 5 = 1760 tiles
    assert p["tile"] == 1 and p["TM"] == 11 and p["whole"] + p["rem"] == 1760
    if p["rem"]:
        assert p["rem"] * p["split"] <= 512
    assert plan(4096, 1400, 5, 1)["tile"] == 0                                  # fp16 operands: streaming kernel
    assert plan(96, 144, 2)["tile"] == 1 and plan(96, 144, 2)["TN"] == 1        # tiny feature_dim: one ragged column tile
    out = (C.c_int * 8)()
    assert lib.nd_step_plan(4096, 0, 5, 0, out) != 0 and lib.nd_step_plan(40, 32, 1, 0, out) != 0


def test_step_gemm_loops_keep_counted_waits():
    """ISA check (no GPU): every software-pipelined k_skinny loop must wait with a counted vmcnt and contain no flat_load --
    a pending flat load (pointer read out of a descriptor table without an address-space cast) or an uncountable load before
    the loop degrades every wait to vmcnt(0) and silently collapses the pipeline (measured: +5..8 us per launch)."""
    import shutil
    import subprocess
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 degraded" in r.stdout


def test_conditioner_and_batch_abi_argument_validation_without_gpu():
    """Host-side checks of the round-3 entry points: configuration validation, workspace sizing and NULL handling return error
    codes (with an nd_last_error text) before any HIP call."""
    from nested_diffusion_amd import _lib
    lib = _lib.load()

    def cond_cfg(**over):
        c = _lib.NdCondConfig()
        c.img_size, c.patch, c.in_chans, c.embed_dim, c.num_heads, c.mlp_hidden, c.n_blocks, c.n_mlps = 224, 16, 3, 768, 12, 3072, 12, 5
        c.mlp_widths[0], c.mlp_widths[1], c.mlp_widths[2] = 4096, 2048, 128
        c.num_classes, c.max_batch, c.max_tokens, c.operand_dtype, c.ln_eps = 2, 32, 197, 0, 1e-6
        for k, v in over.items():
            setattr(c, k, v)
        return c
    good = cond_cfg()
    nbytes = lib.nd_cond_workspace_bytes(ctypes.byref(good))
    # activations of ViT-B/16 at B = 32: qkv [6304, 2304] + fc1 [6304, 3072] + five [6304, 768] + im2col + GEMM / linear workspaces
    assert 250e6 < nbytes < 600e6, nbytes
    h = ctypes.c_void_p()
    assert lib.nd_cond_create(ctypes.byref(good), ctypes.byref(h)) == 0 and h.value
    assert lib.nd_cond_get_config(h).contents.embed_dim == 768
    assert lib.nd_cond_bind_workspace(h, None, nbytes) != 0 and b"NULL" in lib.nd_last_error()
    assert lib.nd_cond_bind_workspace(h, 0x1001, nbytes) != 0 and b"aligned" in lib.nd_last_error()
    assert lib.nd_cond_set_block(h, 12, ctypes.byref(_lib.NdVitBlockWeights())) != 0             # block index out of range
    assert lib.nd_cond_set_block(h, 0, ctypes.byref(_lib.NdVitBlockWeights())) != 0 and b"NULL" in lib.nd_last_error()
    assert lib.nd_guiding_prediction(h, 0x1000, 0x1000, None, 4, None) == -3                       # ND_ERR_STATE: workspace not bound
    assert lib.nd_vit_block(h, 0, 0x1000, 0x1000, 4, 196, None) == -3
    assert lib.nd_cond_destroy(h) == 0
    for bad in (cond_cfg(embed_dim=760), cond_cfg(patch=15), cond_cfg(n_mlps=13), cond_cfg(max_tokens=100), cond_cfg(max_tokens=300),
                cond_cfg(operand_dtype=3), cond_cfg(ln_eps=0.0), cond_cfg(max_batch=0)):
        assert lib.nd_cond_workspace_bytes(ctypes.byref(bad)) == 0
        assert lib.nd_cond_create(ctypes.byref(bad), ctypes.byref(ctypes.c_void_p())) != 0
    mw = cond_cfg()
    mw.mlp_widths[2] = 100                                                                           # not a multiple of 16
    assert lib.nd_cond_workspace_bytes(ctypes.byref(mw)) == 0
    assert lib.nd_philox_normal(None, 1, 1, 1, 1, 1, 0, 0, 0, None) != 0
    assert lib.nd_philox_normal(0x1000, 0, 1, 1, 1, 1, 0, 0, 0, None) != 0
    assert lib.nd_philox_raw(None, None, 1, 0, 0, None) != 0
    assert lib.nd_predict_batch(None, None, None, None, None, 1, 1, 1, 0.5, 1, None) != 0
    assert lib.nd_seed(None, 1, 0) != 0
    assert lib.nd_resident_weight_bytes(None, 0) == -1


def test_documents_quote_what_the_committed_profiles_say():
    """tools/check_docs.py: every tagged figure of profiles/README.md and DESIGN.md equals the value in the CSV / log / JSON it names; the
    stage table of DESIGN 7b and the first section of profiles/README.md cite the newest round's kernel stats; the header's convention
    block describes the operand storage the library uses (round 4's verdict found all three drifted)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_docs.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 problem(s)" in r.stdout
