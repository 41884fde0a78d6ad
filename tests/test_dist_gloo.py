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
