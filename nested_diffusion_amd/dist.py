"""Multi-GPU: one process per GPU (torch.distributed; backend 'nccl' = RCCL over xGMI on ROCm).

The path shards with no data-path collective: every (image, member, trial) sample is independent, so
the test batch is split across ranks, every rank holds all K members, and the only exchange is ONE
all-gather of the per-image results at the end of a batch.  CPU tests run the same code
over the gloo backend.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as td

# The hosts of this pool support only dmabuf IPC: with the legacy IPC mode RCCL's (and torch's) cross-process device-memory handles
# fail with `hipIpcGetMemHandle: invalid argument`.  The variable is read when the HSA runtime starts, so it has to be in the
# environment BEFORE the first GPU call of every rank -- whichever launcher started it (bench.py's own children, the tests'
# children, torch.distributed.run): set here, at import, and again at the top of init_from_env; an explicit setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def rank_world() -> Tuple[int, int]:
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def init_from_env(backend: str = None, force: bool = False) -> Tuple[int, int, int]:
    """Initialise from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, local_rank, world).  No-op for single-process runs unless `force` (a one-rank group: lets a 1-GPU box run the
    RCCL initialisation and collective of the production branch)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before any GPU call (see the note at the top of this file)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # ND_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the ranks then share devices)
            backend = os.environ.get("ND_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if backend == "nccl":
            # one rank per GPU: two ranks on one device make RCCL fail with a duplicate-GPU error (or hang)
            if local >= n_dev:
                raise RuntimeError(f"LOCAL_RANK={local} but only {n_dev} GPU(s) are visible: the nccl (RCCL) backend needs one "
                                   "device per rank (ND_DIST_BACKEND=gloo lets ranks share a device for rehearsal)")
            torch.cuda.set_device(local)
            td.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            if n_dev:
                local = local % n_dev            # rehearsal only: more ranks than devices share them
            td.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def shutdown() -> None:
    """Leave the process group (no-op for single-process runs); called from main()'s finally block so that a rank that
    failed does not leave its peers blocked in a collective."""
    if td.is_available() and td.is_initialized():
        try:
            td.destroy_process_group()
        except Exception:
            pass


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n rows: the first n % world ranks get one extra row."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _gather_padded(pad: torch.Tensor, world: int) -> torch.Tensor:
    """The collective itself: every rank contributes `pad` ([widest, ...]), every rank receives [world * widest, ...] in rank order.
    RCCL gathers device tensors directly; gloo (CPU rehearsal of the N>1 path) is staged through the host."""
    dev = pad.device
    stage = torch.device("cpu") if (td.get_backend() == "gloo" and pad.is_cuda) else dev
    out = torch.empty((world * pad.shape[0],) + tuple(pad.shape[1:]), dtype=pad.dtype, device=stage)
    td.all_gather_into_tensor(out, pad.to(stage))
    return out.to(dev)


def all_gather_rows(local: torch.Tensor, n_total: int, world: int = None, force_collective: bool = False) -> torch.Tensor:
    """Concatenate the ranks' row shards (dim 0) in rank order with one all_gather.  Shards may be
    ragged by one row (shard_bounds); they are padded to the widest shard for the collective.
    force_collective: issue the collective even in a one-rank group (tests)."""
    rank, w = rank_world()
    world = w if world is None else world
    if world == 1 and not (force_collective and td.is_initialized()):
        return local
    q, r = divmod(n_total, world)
    widest = q + (1 if r else 0)
    pad = torch.zeros((widest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = _gather_padded(pad, world)
    if r == 0:
        return out
    pieces = []
    for k in range(world):
        lo, hi = shard_bounds(n_total, k, world)
        pieces.append(out[k * widest: k * widest + (hi - lo)])
    return torch.cat(pieces, dim=0)
