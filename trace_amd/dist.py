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


def gather_floats(vec: Sequence[float]) -> List[List[float]]:
    """All-gather of a short float vector -> [world][len(vec)] (per-rank timings for the bench line: a sub-linear scaling curve
    has to show WHICH rank was slow, and in which stage)."""
    v = [float(x) for x in vec]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [v]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(v, dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]


# ---- host placement of the ranks -----------------------------------------------------------------------------------------
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]"""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def numa_node_of_pci(bdf: str, sysfs: str = "/sys") -> int:
    """NUMA node of a PCI function ('0000:c1:00.0'); -1 when the platform does not say (single node, VM)."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", bdf.lower(), "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def cpus_of_numa_node(node: int, sysfs: str = "/sys") -> List[int]:
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError):
        return []


def gpu_pci_bdf(local_rank: int) -> str | None:
    """PCI address of HIP device `local_rank` as sysfs spells it, or None (no GPU / torch too old to say)."""
    if not torch.cuda.is_available():
        return None
    try:
        p = torch.cuda.get_device_properties(local_rank)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


def plan_affinity(allowed: Sequence[int], local_rank: int, local_world: int, node_cpus: Sequence[int]) -> Tuple[List[int], str]:
    """Which CPUs rank `local_rank` of `local_world` on this host should run on -> (cpus, how).  With a known NUMA node for its GPU: that
    node's CPUs (those the process may use), shared by the ranks whose GPUs sit on the node.  Otherwise the allowed CPUs are cut into
    `local_world` equal contiguous slices, so that eight ranks' Python threads do not pile onto the same cores.  One rank alone is left as it is."""
    allowed = sorted(set(int(c) for c in allowed))
    if local_world <= 1 or not allowed:
        return list(allowed), "unchanged (single rank)"
    on_node = [c for c in allowed if c in set(node_cpus)]
    if on_node:
        return on_node, "cpus of the GPU's NUMA node"
    n = len(allowed) // local_world
    if n < 1:
        return list(allowed), "unchanged (fewer CPUs than ranks)"
    return allowed[local_rank * n:(local_rank + 1) * n], f"even slice {local_rank + 1}/{local_world} of the allowed CPUs (NUMA node unknown)"


def bind_rank(local_rank: int, local_world: int, sysfs: str = "/sys") -> dict:
    """Pins this process to the CPUs plan_affinity picks for its GPU; never fails the run (returns what it did)."""
    info = {"numa_node": -1, "cpus": None, "how": "unchanged"}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return info
    bdf = gpu_pci_bdf(local_rank)
    node = numa_node_of_pci(bdf, sysfs) if bdf else -1
    cpus, how = plan_affinity(allowed, local_rank, local_world, cpus_of_numa_node(node, sysfs) if node >= 0 else [])
    info.update(numa_node=node, how=how, pci=bdf)
    try:
        if cpus and cpus != allowed:
            os.sched_setaffinity(0, cpus)
        info["cpus"] = f"{len(cpus)} cpus: {cpus[0]}..{cpus[-1]}" if cpus else None
    except OSError as e:
        info["how"] = f"unchanged (sched_setaffinity: {e})"
    return info
