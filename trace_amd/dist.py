"""Data-parallel video sharding (one process per GPU) + the single collective of the path.

The reference runs inference in one process on one device, batch 1 (trace/eval/evaluate.py:298-357); videos are
independent requests, so the MI355X design replicates the model per GPU (15 GB of 288 GB), shards the video list by
rank with no data-path collective, and gathers the packed event outputs (int32 token ids, <= 4 KiB per video) with ONE
RCCL all-gather per batch over xGMI so rank 0 (or every rank) can run the drivers' unchanged id-stream parser.
The payload is latency-bound; ring/direct and link topology are irrelevant at this size."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run contract); returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("TRACE_FORCE_PG") == "1"      # exercise the RCCL path with a single rank (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin: rank r takes videos r, r+W, ... (balanced to within one video)."""
    return list(range(rank, n_items, world))


def pack_ids(ids: Sequence[Sequence[int]], max_new: int) -> torch.Tensor:
    """[n, 1 + max_new] int32: length then ids, zero padded."""
    out = torch.zeros((len(ids), 1 + max_new), dtype=torch.int32)
    for i, row in enumerate(ids):
        row = list(row)[:max_new]
        out[i, 0] = len(row)
        out[i, 1:1 + len(row)] = torch.tensor(row, dtype=torch.int32)
    return out


def unpack_ids(packed: torch.Tensor) -> List[List[int]]:
    return [row[1:1 + int(row[0])].tolist() for row in packed.cpu()]


def gather_outputs(local_ids: Sequence[Sequence[int]], max_new: int, per_rank: int, device=None) -> List[List[List[int]]]:
    """All-gather of the packed ids; returns [world][per_rank] id lists (ranks with fewer videos pad with empties)."""
    packed = pack_ids(list(local_ids) + [[]] * (per_rank - len(local_ids)), max_new)
    if not (dist.is_available() and dist.is_initialized()):
        return [unpack_ids(packed)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    packed = packed.to(device)
    out = torch.empty((dist.get_world_size(),) + tuple(packed.shape), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, packed) if hasattr(dist, "all_gather_into_tensor") and device.type == "cuda" else \
        dist.all_gather(list(out.unbind(0)), packed)
    return [unpack_ids(out[r]) for r in range(out.shape[0])]


def merge_round_robin(gathered: List[List[List[int]]], n_items: int) -> List[List[int]]:
    """Inverse of shard_indices: video i lives at gathered[i % W][i // W]."""
    W = len(gathered)
    return [gathered[i % W][i // W] for i in range(n_items)]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
