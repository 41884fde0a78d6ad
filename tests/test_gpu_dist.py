"""N>1 path with real GPU compute.  On a box with >= 2 GPUs the two ranks take one device each and the collective is
RCCL (`backend="nccl"`, dist.py:init_from_env's production branch); on a 1-GPU box the two processes share cuda:0 and the one
collective runs over gloo, staged through the host.  Each rank runs the hot path on its batch shard; the gathered result
must equal the single-process full-batch result -- with explicit noise (kernels) and with the runner's own seeded draws
and input perturbations (--noise_perturbation / --covered / --crop: the windows and draws of image i must not depend on the
world size; SURVEY section 4 "1/2/4/8-rank runs must produce identical gathered logits")."""
import argparse
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _build(device, args_ns, B):
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    from oracle import ref_cpu
    ns = argparse.Namespace
    embed, heads, depth, img, patch, K, T, C = 128, 2, 5, 32, 16, 5, 6, 2
    D, H, F = 3 * img * img, 64, 64
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=3)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(64, 32, 16), seed=20 + i) for i in range(K)]
    members = [ref_cpu.init_cond_model_params(D, H, F, C, T, True, seed=40 + i) for i in range(K)]
    cfg = ns(data=ns(dataset="ChestXRay", num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=F, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=B))
    cond = GuidingConditioner(VisionTransformer(vp, heads, device), [Classifier(m, device) for m in mlps])
    return Diffusion(args_ns, cfg, device=device, conditioner=cond, noise_estimator_states=members), (K, T, C, img)


def _worker(rank, world, port, out_dir, backend):
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local))
    import torch.distributed as td
    from nested_diffusion_amd import dist as nd_dist
    ns = argparse.Namespace
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    B, mc = 7, 2                                                                      # B = 7: ragged shards (4 + 3)
    pert = dict(noise_perturbation=0.05, covered=(0.05, 2.0), crop=0.1, low_resolution=0, brightness=0.0, contrast=1.0)
    g = torch.Generator().manual_seed(9)
    x_cpu = torch.rand(B, 3, 32, 32, generator=g)
    target = torch.randint(0, 2, (B,), generator=g)

    # ---- world = 1 (no process group yet): full batch through test_atk with the runner's own seeded draws ----
    full_runner, (K, T, C, img) = _build(dev, ns(seed=11, mc_trials=mc, **pert), B)
    full_runner.test_atk(test_loader=[(x_cpu, target), (x_cpu.flip(0), target)])
    probs_full, acc_full = full_runner.last_probs.clone(), full_runner.last_report["accuracy"]
    del full_runner

    nd_dist.init_from_env(backend=backend)
    assert td.get_backend() == backend

    # ---- (1) explicit noise: kernels on a shard == kernels on the full batch ----
    runner, _ = _build(dev, ns(seed=1, mc_trials=mc), B)
    runner.load_noise_estimators(max_batch=B)
    x = x_cpu.to(dev)
    noise = torch.randn(K, T, mc, B, C, generator=g).to(dev)                          # [K, T, trial, image, C]
    full = runner.predict_batch(x, noise=noise.reshape(K, T, mc * B, C))
    lo, hi = nd_dist.shard_bounds(B, rank, world)
    part = runner.predict_batch(x[lo:hi], noise=noise[:, :, :, lo:hi].reshape(K, T, mc * (hi - lo), C).contiguous())
    prob = nd_dist.all_gather_rows(part["prob"], B, world)
    vote = nd_dist.all_gather_rows(part["vote"], B, world)
    ok1 = torch.equal(vote, full["vote"]) and torch.allclose(prob, full["prob"], rtol=0, atol=1e-6)
    err1 = float((prob - full["prob"]).abs().max())
    del runner

    # ---- (2) same --seed, world = 2, perturbations on: gathered probabilities == the world-1 run above ----
    shard_runner, _ = _build(dev, ns(seed=11, mc_trials=mc, **pert), B)
    shard_runner.test_atk(test_loader=[(x_cpu, target), (x_cpu.flip(0), target)])
    err2 = float((shard_runner.last_probs - probs_full).abs().max())
    ok2 = err2 <= 1e-6 and shard_runner.last_report["accuracy"] == acc_full
    # only the rank's rows of each of the two batches crossed PCIe (the loader hands whole batches here: sliced on the host)
    ok2 = ok2 and shard_runner.bytes_uploaded == 2 * (hi - lo) * 3 * 32 * 32 * 4
    torch.save({"ok": bool(ok1 and ok2), "err_kernels": err1, "err_seeded_run": err2, "backend": backend, "bytes_uploaded": shard_runner.bytes_uploaded},
               os.path.join(out_dir, f"r{rank}.pt"))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharded_batch_equals_full_batch(tmp_path, world):
    """world = 2: shards of 4 + 3 images; world = 4: 2 + 2 + 2 + 1 (one rank per device over RCCL when the box has that many GPUs,
    else the ranks share cuda:0 and gather over gloo; at most 5 GPU processes at once)."""
    port = _free_port()
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    mp.spawn(_worker, args=(world, port, str(tmp_path), backend), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        print(res)
        assert res["ok"], res


def _rccl_one_rank(port, out_path):
    """dist.py's production branch on whatever one GPU offers: backend 'nccl' (= RCCL), communicator bound to the device,
    all_gather_into_tensor on device tensors -- in a one-rank group."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as td
    from nested_diffusion_amd import dist as nd_dist
    rank, local, world = nd_dist.init_from_env(backend="nccl", force=True)
    assert (rank, local, world) == (0, 0, 1) and td.get_backend() == "nccl"
    x = torch.arange(7 * 2, dtype=torch.float32, device="cuda:0").reshape(7, 2)
    got = nd_dist.all_gather_rows(x, 7, force_collective=True)
    td.barrier(device_ids=[0])
    torch.cuda.synchronize()
    ok = got.is_cuda and torch.equal(got, x)
    nd_dist.shutdown()
    with open(out_path, "w") as f:
        f.write("ok" if ok else "mismatch")


def test_rccl_initialises_and_gathers_on_one_rank(tmp_path):
    """The builder's boxes have one GPU, so the two-rank test above gathers over gloo there.  This one makes sure the RCCL
    branch itself (init_process_group('nccl', device_id=...) + all_gather_into_tensor on HBM tensors) has run on the hardware."""
    out = tmp_path / "rccl.txt"
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_one_rank, args=(_free_port(), str(out)))
    p.start()
    p.join(180)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("RCCL one-rank group did not finish in 180 s")
    assert p.exitcode == 0
    assert out.read_text() == "ok"
