"""Batch sharding of independent 512-bp windows across the GPUs of one node (one process per GPU).

The reference is single-process / single-device (`src/zero_shot_score.py:27,97,112`); windows are
independent (`:111-120` is a plain loop over batches), so the path shards with NO data-path collective:
rank r owns the contiguous block [r*ceil(N/W), (r+1)*ceil(N/W)) of the window list, the tail padded with
a dummy window so every rank runs the same number of rows, and ONE all-gather (RCCL over xGMI on the GPU
box, gloo in the CPU tests) per result chunk reassembles the per-SNP rows in the single-GPU order
(SURVEY.md §8e).  Payloads are tiny (16 B/window of probabilities, 4*D B/window of embeddings).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(device: str) -> str:
    """Under `torchrun` (WORLD_SIZE set in the environment, any value >= 1): join the process group (RCCL for ROCm devices, gloo
    for cpu) and return this rank's device (`cuda:LOCAL_RANK`); otherwise (plain `python`) return `device` unchanged.  A
    one-rank launch still creates the group, so `torchrun --nproc-per-node 1` runs the same collectives as 8 ranks do."""
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC, before the first GPU call (RCCL needs it here)
    if "WORLD_SIZE" not in os.environ or "RANK" not in os.environ or not dist.is_available():
        return device
    if str(device).startswith("cuda"):
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(torch.device(device))
    if not dist.is_initialized():
        global _OWN_GROUP
        _OWN_GROUP = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if str(device).startswith("cuda"):
            dist.init_process_group(backend="nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend="gloo")
    return device


_OWN_GROUP = False


def shutdown():
    """Destroy the process group if `init_from_env` created it (end of a command-line run)."""
    global _OWN_GROUP
    if _OWN_GROUP and dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    _OWN_GROUP = False


def rank0_decides(flag: bool, device="cpu") -> bool:
    """A yes / no that selects between a branch WITH collectives and one without (a cache file exists -> load it, else embed =
    all-gather) must be the same on every rank, whatever each rank's view of the file system: rank 0's value, broadcast (one int).
    Outside a process group: the caller's own value."""
    if not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    dev = torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(bool(flag))], dtype=torch.int32, device=dev)
    dist.broadcast(t, src=0)
    return bool(int(t.item()))


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int, int]:
    """-> (start, stop, per_rank): contiguous block of rank `rank`; per_rank = ceil(n / world_size)."""
    per = -(-n // world_size) if n > 0 else 0
    start = min(rank * per, n)
    stop = min(start + per, n)
    return start, stop, per


def pad_rows(x: torch.Tensor, rows: int) -> torch.Tensor:
    """Pad dim 0 to `rows` by repeating the last row (a dummy window; stripped after the gather)."""
    if x.shape[0] == rows:
        return x
    if x.shape[0] == 0:
        return x.new_zeros((rows,) + tuple(x.shape[1:]))
    return torch.cat([x, x[-1:].expand(rows - x.shape[0], *x.shape[1:])], dim=0)


def _all_gather(local: torch.Tensor, ws: int) -> torch.Tensor:
    """the collective itself: local [rows, ...] of each of `ws` ranks -> [ws * rows, ...] in rank order."""
    local = local.contiguous()
    if local.is_cuda and dist.get_backend() != "nccl":
        # device tensors on a gloo group (several ranks sharing one GPU in tests): stage through the host
        return _all_gather(local.cpu(), ws).to(local.device)
    out = torch.empty((ws * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda:
        dist.all_gather_into_tensor(out, local)            # RCCL
    else:
        parts = list(out.chunk(ws, dim=0))                 # gloo (CPU tests)
        dist.all_gather(parts, local)
    return out


def all_gather_blocks(local: torch.Tensor) -> torch.Tensor:
    """local [rows, ...] (the same `rows` on every rank) -> [world * rows, ...] in rank order (nothing stripped): the raw collective
    for callers that place the blocks themselves (plantcad2_eval's chunked gathers).  Outside a process group: `local`."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    return _all_gather(local, dist.get_world_size())


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """local [per_rank, ...] (equal on every rank) -> [n_total, ...] in rank order, padding stripped."""
    rank, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return local[:n_total]
    return _all_gather(local, ws)[:n_total]                 # also at world size 1 (a one-rank torchrun): same code path as N ranks
