"""N>1 path on CPU: two gloo ranks shard a batch, compute rank-local results and exchange them with the
single all-gather the GPU path uses (nested_diffusion_amd.dist).  The per-sample arithmetic is replaced by
the CPU oracle here -- what is under test is the sharding / gather logic, which is device-independent."""
import os
import socket

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from oracle import ref_cpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nested_diffusion_amd import dist as nd_dist
    r, _, w = nd_dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and nd_dist.rank_world() == (rank, world)
    g = torch.Generator().manual_seed(0)
    K, mc, C = 3, 2, 2
    samples = torch.randn(K * mc, n_total, C, generator=g)            # same on every rank (same seed)
    lo, hi = nd_dist.shard_bounds(n_total, rank, world)
    local = [s[lo:hi].clone() for s in samples]
    vote_l = ref_cpu.majority_voting_for_mc_samples(local) if hi > lo else torch.zeros(0, dtype=torch.int64)
    prob_l = ref_cpu.compute_ensemble_confidence([x.clone() for x in local], 0.1737) if hi > lo else torch.zeros(0, C)
    prob = nd_dist.all_gather_rows(prob_l, n_total, world)
    vote = nd_dist.all_gather_rows(vote_l, n_total, world)
    full_vote = ref_cpu.majority_voting_for_mc_samples(list(samples))
    full_prob = ref_cpu.compute_ensemble_confidence([x.clone() for x in samples], 0.1737)
    # votes are integers: exact.  Probabilities: torch's CPU softmax takes a vector or a scalar-tail path
    # depending on the row count, so shard-vs-full agree to an ulp, not bitwise.
    ok = torch.equal(vote, full_vote) and torch.allclose(prob, full_prob, rtol=0, atol=2e-7) and prob.shape == (n_total, C)
    torch.save({"ok": ok, "rank": rank}, os.path.join(out_dir, f"r{rank}.pt"))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 33])
def test_batch_sharding_and_single_all_gather_world2(tmp_path, n_total):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert res["ok"], (r, n_total)


def test_configs3_group_shape_world8_256_rows(tmp_path):
    """BASELINE configs[3]'s group: 8 ranks x 32 rows = one batch of 256, gathered with the single all-gather (gloo, CPU tensors)."""
    world, n_total = 8, 256
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))["ok"], r


@pytest.mark.parametrize("world,n_total", [(8, 256), (3, 10), (4, 7)])
def test_emulated_ranks_assemble_what_the_collective_returns(world, n_total):
    """tests/_emulated_dist.py (the one-process rehearsal the GPU tests use for world sizes the box cannot host as processes): after the
    last rank's pass all_gather_rows returns exactly the concatenation of the shards -- ragged splits included -- and rank_world()
    reported each (rank, world) on the way; switched off, the process is a single rank again."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _emulated_dist import emulate_rank
    from nested_diffusion_amd import dist as nd_dist
    full = torch.arange(n_total * 3, dtype=torch.float32).reshape(n_total, 3) + 1.0
    sink, out = {}, None
    try:
        for r in range(world):
            emulate_rank(r, world, sink)
            assert nd_dist.rank_world() == (r, world)
            lo, hi = nd_dist.shard_bounds(n_total, r, world)
            out = nd_dist.all_gather_rows(full[lo:hi].clone(), n_total)
            assert out.shape == full.shape and torch.equal(out[:hi], full[:hi]) and not out[hi:].any()
    finally:
        emulate_rank()
    assert torch.equal(out, full) and sorted(sink) == list(range(world))
    assert nd_dist.rank_world() == (0, 1)


def test_single_process_is_a_noop():
    from nested_diffusion_amd import dist as nd_dist
    assert nd_dist.rank_world() == (0, 1)
    x = torch.arange(6.0).reshape(3, 2)
    assert nd_dist.all_gather_rows(x, 3, 1) is x


# ---- the DATA path is sharded too: a rank decodes, stages and uploads its own rows of every batch and nothing else ----------------
class _CountingDataset(torch.utils.data.Dataset):
    """23 'images' whose pixels encode their index; counts which samples were opened."""

    def __init__(self, n=23):
        self.n, self.opened = n, []

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.opened.append(i)
        return torch.full((3, 4, 4), float(i)), i % 2


@pytest.mark.parametrize("world", [1, 2, 8])
def test_shard_batch_sampler_opens_only_the_ranks_rows(world):
    """data.ShardBatchSampler through a real DataLoader: batch b of the reference's loader (batch_size B, shuffle=False, drop_last=True,
    classification_train_separately.py:675-681) is rows b*B .. b*B + B - 1; rank r is handed rows [lo, hi) of it in order, opens exactly
    those samples, and the ranks' slices concatenated in rank order are the world-1 batches.  Ragged split (B = 10 over 8 ranks)."""
    from nested_diffusion_amd import dist as nd_dist
    from nested_diffusion_amd.data import ShardBatchSampler
    B, n = 10, 23
    ref_ds = _CountingDataset(n)
    ref = [x for x in torch.utils.data.DataLoader(ref_ds, batch_size=B, shuffle=False, drop_last=True)]
    assert len(ref) == 2
    pieces = [[] for _ in ref]
    for r in range(world):
        lo, hi = nd_dist.shard_bounds(B, r, world)
        ds = _CountingDataset(n)
        loader = torch.utils.data.DataLoader(ds, batch_sampler=ShardBatchSampler(len(ds), B, lo, hi))
        got = list(loader)
        assert len(got) == len(ref) == len(loader)                                  # drop_last on the GLOBAL batch
        for b, (x, t) in enumerate(got):
            assert tuple(x.shape) == (hi - lo, 3, 4, 4) and tuple(t.shape) == (hi - lo,)
            pieces[b].append((x, t))
        assert ds.opened == [b * B + i for b in range(len(ref)) for i in range(lo, hi)]      # nothing but its own rows was decoded
        assert len(ds.opened) * 3 * 4 * 4 * 4 == sum(x.numel() * 4 for x, _ in got)          # bytes staged = the shard's bytes
    for b, (x_ref, t_ref) in enumerate(ref):
        assert torch.equal(torch.cat([x for x, _ in pieces[b]]), x_ref) and torch.equal(torch.cat([t for _, t in pieces[b]]), t_ref)
    with pytest.raises(ValueError):
        ShardBatchSampler(23, 10, 4, 11)


@pytest.mark.parametrize("world", [2, 8])
def test_synthetic_loader_shard_is_a_slice_of_the_global_batch(world):
    from nested_diffusion_amd import dist as nd_dist
    from nested_diffusion_amd.data import SyntheticLoader
    B = 16
    full = list(SyntheticLoader(2, B, 2, seed=5, size=8))
    for r in range(world):
        lo, hi = nd_dist.shard_bounds(B, r, world)
        mine = SyntheticLoader(2, B, 2, seed=5, size=8, shard=(lo, hi))
        assert mine.shard == (lo, hi) and mine.global_batch == B
        for (x, t), (xf, tf) in zip(mine, full):
            assert torch.equal(x, xf[lo:hi]) and torch.equal(t, tf[lo:hi])


@pytest.mark.parametrize("world", [2, 8])
def test_perturbations_of_image_i_do_not_depend_on_the_world_size(world, monkeypatch):
    """runner.Diffusion.perturb on a shard: every rank (same --seed: set_seed, classification_train_separately.py:31-38) makes the
    reference's random calls for the WHOLE batch -- the draw of add_noise (utils.py:274), python `random` for the cover rectangles
    (:321-343), torch.randint for the crop corners (:296-300) -- and applies rows [lo, hi) of them.  The pixel kernels are replaced by
    recorders here (no GPU in this suite): what is checked is that rank r is handed exactly rows [lo, hi) of the world-1 parameters,
    with --noise_perturbation / --covered / --crop all on, over two consecutive batches."""
    import argparse
    from nested_diffusion_amd import dist as nd_dist, perturb as P
    from nested_diffusion_amd.runner import Diffusion, set_seed
    B, H = 16, 12
    args = argparse.Namespace(noise_perturbation=0.1, low_resolution=0, brightness=0.0, contrast=1.0, covered=(0.05, 2.0), crop=0.25)
    log = []
    monkeypatch.setattr(P, "add_noise", lambda x, std, z=None: (log.append(("noise", z.clone())), x)[1])
    monkeypatch.setattr(P, "random_cover_new", lambda x, params, rects=None: (log.append(("cover", list(rects))), x)[1])
    monkeypatch.setattr(P, "random_crop_and_resize", lambda x, k, corners=None: (log.append(("crop", list(corners))), x)[1])
    fake = argparse.Namespace(args=args)

    def run(lo, hi):
        log.clear()
        set_seed(123)
        for _ in range(2):                                                           # two batches: the generators run on between them
            Diffusion.perturb(fake, torch.zeros(hi - lo, 3, H, H), lo, hi, B)
        return list(log)

    whole = run(0, B)
    assert [k for k, _ in whole] == ["noise", "cover", "crop"] * 2
    for r in range(world):
        lo, hi = nd_dist.shard_bounds(B, r, world)
        mine = run(lo, hi)
        for (kind, got), (_, ref) in zip(mine, whole):
            if kind == "noise":
                assert tuple(got.shape) == (hi - lo, 3, H, H) and torch.equal(got, ref[lo:hi])
            else:
                assert got == ref[lo:hi] and len(got) == hi - lo
    with pytest.raises(ValueError):
        Diffusion.perturb(fake, torch.zeros(3, 3, H, H), 2, 6, B)                    # 3 rows handed for a shard of 4


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_disk_loader_decodes_only_the_ranks_files(tmp_path, world, monkeypatch):
    """data.get_test_loader(shard=(lo, hi)) over an image tree on disk (ImageFolder ordering + the reference's transforms,
    dataset_helper/chest_x_ray_dataset.py:28-51): rank r's loader yields rows [lo, hi) of every global batch with their targets, opens
    exactly those files, and the ranks' pieces concatenated in rank order are the unsharded loader's batches (drop_last on the GLOBAL
    batch: 7 images, batch 3 -> two batches)."""
    import argparse
    import numpy as np
    from PIL import Image
    import nested_diffusion_amd.data as data
    from nested_diffusion_amd import dist as nd_dist
    rng = np.random.default_rng(0)
    names = {"NORMAL": ["a.png", "b.png", "c.png", "d.png"], "PNEUMONIA": ["e.png", "f.png", "g.png"]}
    for cls, files in names.items():
        os.makedirs(os.path.join(str(tmp_path), "testing", cls))
        for f in files:
            Image.fromarray(rng.integers(0, 256, size=(40, 40, 3), dtype=np.uint8), "RGB").save(os.path.join(str(tmp_path), "testing", cls, f))
    ns = argparse.Namespace
    args = ns(synthetic_batches=0, preprocess="grayscaled", seed=0)
    cfg = ns(data=ns(dataset="ChestXRay", dataroot=str(tmp_path), num_classes=2, num_workers=0), model=ns(data_dim=3 * 224 * 224), testing=ns(batch_size=3))
    opened = []
    orig = data.ImageFolderDataset.__getitem__
    monkeypatch.setattr(data.ImageFolderDataset, "__getitem__", lambda self, i: (opened.append(i), orig(self, i))[1])
    full = list(data.get_test_loader(args, cfg))
    assert len(full) == 2 and opened == [0, 1, 2, 3, 4, 5] and torch.cat([t for _, t in full]).tolist() == [0, 0, 0, 0, 1, 1]
    pieces = [[], []]
    for r in range(world):
        lo, hi = nd_dist.shard_bounds(3, r, world)
        opened.clear()
        loader = data.get_test_loader(args, cfg, shard=(lo, hi))
        assert loader.shard == (lo, hi) and loader.global_batch == 3
        got = list(loader)
        assert opened == [b * 3 + i for b in range(2) for i in range(lo, hi)]
        for b, (x, t) in enumerate(got):
            assert tuple(x.shape) == (hi - lo, 3, 224, 224)
            pieces[b].append((x, t))
    for b in range(2):
        assert torch.equal(torch.cat([x for x, _ in pieces[b]]), full[b][0]) and torch.equal(torch.cat([t for _, t in pieces[b]]), full[b][1])
