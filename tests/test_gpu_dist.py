"""N>1 path with real GPU compute: two processes share cuda:0 (gloo for the one collective, staged through the host;
the production backend is RCCL), each runs the hot path on its batch shard; the gathered result must equal the
single-process full-batch result."""
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


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as td
    from nested_diffusion_amd import dist as nd_dist
    from nested_diffusion_amd.mapping import Classifier, GuidingConditioner, VisionTransformer
    from nested_diffusion_amd.runner import Diffusion
    from oracle import ref_cpu
    nd_dist.init_from_env(backend="gloo")
    ns = argparse.Namespace
    embed, heads, depth, img, patch, K, B, T, mc, C = 128, 2, 5, 32, 16, 5, 7, 6, 2, 2      # B = 7: ragged shards (4 + 3)
    D, H, F = 3 * img * img, 64, 64
    vp = ref_cpu.init_vit_params(embed=embed, depth=depth, patch=patch, img=img, seed=3)
    n_tok = (img // patch) ** 2
    mlps = [ref_cpu.init_classifier_params(n_tok * embed, widths=(64, 32, 16), seed=20 + i) for i in range(K)]
    members = [ref_cpu.init_cond_model_params(D, H, F, C, T, True, seed=40 + i) for i in range(K)]
    cfg = ns(data=ns(dataset="ChestXRay", num_classes=C), model=ns(data_dim=D, hidden_dim=H, feature_dim=F, arch="linear"),
             diffusion=ns(timesteps=T, beta_schedule="linear", beta_start=1e-4, beta_end=0.02, aux_cls=ns(arch="sevit"),
                          trained_aux_cls_ckpt_path="", trained_diffusion_ckpt_path=[[]], include_guidance=True),
             testing=ns(batch_size=B))
    cond = GuidingConditioner(VisionTransformer(vp, heads, "cuda:0"), [Classifier(m, "cuda:0") for m in mlps])
    runner = Diffusion(ns(seed=1, mc_trials=mc), cfg, device="cuda:0", conditioner=cond, noise_estimator_states=members)
    runner.load_noise_estimators(max_batch=B)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 3, img, img, generator=g).cuda()
    noise = torch.randn(K, T, mc, B, C, generator=g).cuda()                       # [K, T, trial, image, C]
    full = runner.predict_batch(x, noise=noise.reshape(K, T, mc * B, C))
    lo, hi = nd_dist.shard_bounds(B, rank, world)
    part = runner.predict_batch(x[lo:hi], noise=noise[:, :, :, lo:hi].reshape(K, T, mc * (hi - lo), C).contiguous())
    prob = nd_dist.all_gather_rows(part["prob"], B, world)
    vote = nd_dist.all_gather_rows(part["vote"], B, world)
    ok = torch.equal(vote, full["vote"]) and torch.allclose(prob, full["prob"], rtol=0, atol=1e-6)
    torch.save({"ok": bool(ok), "err": float((prob - full["prob"]).abs().max())}, os.path.join(out_dir, f"r{rank}.pt"))
    td.barrier()
    td.destroy_process_group()


def test_two_ranks_sharded_batch_equals_full_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert res["ok"], res
