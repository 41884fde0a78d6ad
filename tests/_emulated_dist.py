"""Test scaffolding: rehearse an N-rank job inside ONE process, with no process group.

A GPU box admits at most 6 processes on its card, so BASELINE configs[3]'s real shape (8 ranks x 32 rows) cannot run there as 8
processes sharing the device.  `emulate_rank(r, world, sink)` swaps two names of nested_diffusion_amd.dist -- `rank_world` (reports
(r, world)) and `_gather_padded` (the collective: deposits this rank's padded shard in sink[r] and assembles whatever shards the sink
holds so far; rows of ranks that have not run yet are zero) -- so the product's sharding, padding and unpadding code runs unchanged.
Running ranks 0 .. world-1 one after the other leaves, after the last one, exactly the tensor the real all-gather returns on every
rank (checked against the real gloo gather in tests/test_dist_gloo.py).  `emulate_rank()` restores the product functions.
Nothing under nested_diffusion_amd/ knows about this file."""
import torch

from nested_diffusion_amd import dist as nd_dist

_REAL = (nd_dist.rank_world, nd_dist._gather_padded)


def emulate_rank(rank: int = None, world: int = None, sink: dict = None) -> None:
    if rank is None:
        nd_dist.rank_world, nd_dist._gather_padded = _REAL
        return
    rank, world = int(rank), int(world)
    sink = sink if sink is not None else {}

    def gather_padded(pad: torch.Tensor, w: int) -> torch.Tensor:
        assert w == world
        sink[rank] = pad
        return torch.cat([sink[k].to(pad.device) if k in sink else torch.zeros_like(pad) for k in range(world)], dim=0)

    nd_dist.rank_world = lambda: (rank, world)
    nd_dist._gather_padded = gather_padded
