"""Host-side drop-in claims of the CLI (nested_diffusion_amd/main.py mirrors diffusion/main.py:16-161, 166-296, 299-380).
CPU tests: the reference's own shipped YAMLs load and carry every key the runner reads; exit codes / exceptions on the failure
paths match the reference for one process and tear the job down for several."""
import argparse
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = "/root/reference/diffusion/configs"


def _args(config, extra=()):
    from nested_diffusion_amd.main import build_parser
    return build_parser().parse_args(["--test", "--loss", "card_onehot_conditional", "--config", config, "--doc", "d", "--ni",
                                      "--preprocess", "grayscaled", *extra])


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="the reference checkout only exists in the build container")
@pytest.mark.parametrize("name,dataset,temp", [("chest_x_ray.yml", "ChestXRay", 0.1737), ("isic_skin_cancer.yml", "ISICSkinCancer", 0.3162)])
def test_reference_yaml_loads_with_every_key_the_runner_reads(name, dataset, temp):
    """diffusion/configs/{chest_x_ray,isic_skin_cancer}.yml through main.load_config: the keys runner.py / data.py read
     are present with the shipped values; --timesteps overrides as at main.py:192-193."""
    from nested_diffusion_amd.main import load_config
    from nested_diffusion_amd.runner import temperature_for
    cfg = load_config(_args(os.path.join(REF_CFG, name)))
    assert cfg.data.dataset == dataset and temperature_for(cfg.data.dataset) == temp
    assert cfg.data.num_classes == 2 and isinstance(cfg.data.dataroot, str)
    assert (cfg.model.data_dim, cfg.model.hidden_dim, cfg.model.feature_dim, cfg.model.arch) == (150528, 4096, 4096, "linear")
    d = cfg.diffusion
    assert d.beta_schedule == "linear" and d.beta_start == 1e-4 and d.beta_end == 0.02 and d.timesteps == 1000
    assert d.aux_cls.arch == "sevit" and d.include_guidance is True
    assert isinstance(d.trained_aux_cls_ckpt_path, str)
    paths = d.trained_diffusion_ckpt_path
    assert isinstance(paths, list) and isinstance(paths[0], list) and len(paths[0]) == 5          # indexed [0][i] (:689), quirk Q1
    assert cfg.testing.batch_size == 70
    assert d.noise_prior is False and cfg.model.cat_y_pred is True                                   # main.py:189-190
    cfg2 = load_config(_args(os.path.join(REF_CFG, name), ["--timesteps", "100", "--dataroot", "/x"]))
    assert cfg2.diffusion.timesteps == 100 and cfg2.data.dataroot == "/x"


def test_invalid_loss_raises_like_the_reference(tmp_path, monkeypatch):
    """main.py:305-311: a loss other than card_onehot_conditional raises NotImplementedError OUTSIDE the reference's
    log-and-swallow block, so the process dies non-zero even as a single process."""
    import torch
    import yaml
    from nested_diffusion_amd import main as nd_main
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)         # get past parse_config's device pick on a CPU box
    cfg = {"data": {"dataset": "ChestXRay", "num_classes": 2, "dataroot": "."}, "model": {"data_dim": 48, "hidden_dim": 16, "feature_dim": 16, "arch": "linear"},
           "diffusion": {"timesteps": 4, "beta_schedule": "linear", "beta_start": 1e-4, "beta_end": 0.02, "aux_cls": {"arch": "sevit"},
                         "trained_aux_cls_ckpt_path": str(tmp_path / "none"), "trained_diffusion_ckpt_path": [[]]}, "testing": {"batch_size": 2}}
    y = tmp_path / "c.yml"
    y.write_text(yaml.safe_dump(cfg))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    argv = ["--test", "--loss", "ddpm", "--config", str(y), "--exp", str(tmp_path / "e"), "--doc", "d", "--ni", "--preprocess", "grayscaled", "--seed", "1"]
    with pytest.raises(NotImplementedError, match="Invalid loss option"):
        nd_main.main(argv)
    # a failure INSIDE the block (here: no conditioner checkpoint) is logged and swallowed for one process: rc 0 (main.py:377-380)
    argv[2] = "card_onehot_conditional"
    assert nd_main.main(argv) == 0


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import torch
    torch.cuda.is_available = lambda: True      # CPU rehearsal: lets parse_config pick a device; nothing is launched
    import torch.distributed as td
    from nested_diffusion_amd import main as nd_main
    rc = nd_main.main({argv!r})
    assert not td.is_initialized(), "process group must be left in main()'s finally"
    sys.exit(rc)
""")


def test_failing_rank_of_a_multi_rank_run_returns_1_and_leaves_the_group(tmp_path):
    """WORLD_SIZE = 2 (gloo): both ranks fail inside the dispatch block (no checkpoint to load).  Unlike the single-process
    case the exit code must be 1 and the process group must be gone, so a launcher tears the job down instead of peers waiting
    in the batch's all-gather (diffusion/main.py:377-380 is single-process only)."""
    import yaml
    cfg = {"data": {"dataset": "ChestXRay", "num_classes": 2, "dataroot": "."}, "model": {"data_dim": 48, "hidden_dim": 16, "feature_dim": 16, "arch": "linear"},
           "diffusion": {"timesteps": 4, "beta_schedule": "linear", "beta_start": 1e-4, "beta_end": 0.02, "aux_cls": {"arch": "sevit"},
                         "trained_aux_cls_ckpt_path": str(tmp_path / "none"), "trained_diffusion_ckpt_path": [[]]}, "testing": {"batch_size": 2}}
    y = tmp_path / "c.yml"
    y.write_text(yaml.safe_dump(cfg))
    argv = ["--test", "--loss", "card_onehot_conditional", "--config", str(y), "--exp", str(tmp_path / "e"), "--doc", "d", "--ni",
            "--preprocess", "grayscaled", "--seed", "1"]
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT, argv=argv))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ND_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 1, out[-2000:]
        assert "process group must be left" not in out


def test_bench_launcher_propagates_a_failing_rank(tmp_path, monkeypatch):
    """bench.launch_ranks: children are started with RANK / WORLD_SIZE / MASTER_* set; a non-zero rank makes the launcher
    return non-zero and the other children are terminated (no GPU needed: the children are stand-ins)."""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    fake = tmp_path / "fake_bench.py"
    fake.write_text(textwrap.dedent("""
        import os, sys, time
        assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'
        r = int(os.environ['RANK'])
        if '--fail' in sys.argv and r == 1:
            sys.exit(3)
        if '--fail' in sys.argv:
            time.sleep(60)
        if r == 0:
            print('{"rank": 0}')
    """))
    monkeypatch.setattr(bench, "__file__", str(fake))
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    import time
    t0 = time.time()
    assert bench.launch_ranks(2, ["--gpus", "2", "--fail"]) == 3
    assert time.time() - t0 < 30                      # rank 0 was terminated, not waited for
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("ND_DIST_BACKEND", raising=False)
    with pytest.raises(SystemExit, match="visible"):
        bench.launch_ranks(2, ["--gpus", "2"])


def test_tensor_key_cannot_be_forged_by_a_recycled_address():
    """latent_model._TensorKey (the cache key of ConditionalModel.encode and of the engine's parameter signature): a new tensor at
    the freed address of the old one, same shape and version counter, is NOT the old tensor; a view of the same live storage is;
    an in-place edit is not.  CPU tensors: the host allocator recycles freed blocks just like the device's caching allocator."""
    import torch
    from nested_diffusion_amd.latent_model import _TensorKey
    hit = False
    for _ in range(32):
        a = torch.zeros(1 << 16)
        key, ptr = _TensorKey(a), a.data_ptr()
        assert key.matches(a) and key.matches(a.view(256, 256).view(-1))
        assert not key.matches(a[1:])                    # same storage, another window
        del a
        b = torch.zeros(1 << 16)
        if b.data_ptr() == ptr:
            hit = True
            assert b._version == 0 and tuple(b.shape) == (1 << 16,)
            assert not key.matches(b), "a recycled address passed for the freed tensor"
            break
    if not hit:
        pytest.skip("the host allocator never handed the freed block back")
    c = torch.zeros(8)
    k = _TensorKey(c)
    c.add_(1)
    assert not k.matches(c)                              # version counter
    k = _TensorKey(c)
    c.view(2, 4)[0, 0] = 5                               # ... shared with every view
    assert not k.matches(c)


def test_loader_path_selection_follows_the_perturbation_flags():
    """runner.Diffusion._perturbs: the batch loop uploads straight into the library's input buffer only when no flag of the robustness
    protocol (test.sh:24: --noise_perturbation, --low_resolution, --brightness, --contrast, --covered, --crop) changes pixels on the
    device; test.sh's own neutral values (0 / 0 / 0 / 1 / unset / 0) select the direct path."""
    import argparse
    from nested_diffusion_amd.runner import Diffusion
    ns = argparse.Namespace
    neutral = dict(noise_perturbation=0.0, low_resolution=0, brightness=0.0, contrast=1.0, covered=None, crop=0.0)
    assert not Diffusion._perturbs(ns(args=ns(**neutral)))
    assert not Diffusion._perturbs(ns(args=ns()))                                   # flags absent altogether
    assert not Diffusion._perturbs(ns(args=ns(**{**neutral, "low_resolution": 1, "covered": (0.0, 3.0)})))
    for k, v in (("noise_perturbation", 0.05), ("low_resolution", 2), ("brightness", -0.2), ("contrast", 0.8), ("covered", (0.1, 2.0)), ("crop", 0.25)):
        assert Diffusion._perturbs(ns(args=ns(**{**neutral, k: v}))), k
