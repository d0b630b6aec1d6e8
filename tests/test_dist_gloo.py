"""Data-parallel sharding + gather of packed token ids with 2 ranks on CPU (gloo).  The union of per-rank outputs
must equal the single-rank order (no reference behaviour exists for multi-GPU inference: SURVEY.md §8e)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from trace_amd import dist as tdist


def fake_generate(video_idx: int, max_new: int):
    n = 3 + (video_idx * 7) % (max_new - 3)
    return [(video_idx * 31 + i) % 32027 for i in range(n)]


def _worker(rank, world, port, n_videos, max_new, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = tdist.init_from_env("gloo")
    mine = tdist.shard_indices(n_videos, r, w)
    per_rank = (n_videos + w - 1) // w
    local = [fake_generate(i, max_new) for i in mine]
    gathered = tdist.gather_outputs(local, max_new, per_rank, torch.device("cpu"))
    merged = tdist.merge_round_robin(gathered, n_videos)
    t = tdist.max_over_ranks(float(rank + 1))
    q.put((rank, merged, t))
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_videos, max_new = 7, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_videos, max_new, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    expect = [fake_generate(i, max_new) for i in range(n_videos)]
    for rank, merged, t in res:
        assert merged == expect, rank
        assert t == 2.0


def test_pack_roundtrip_and_single_process():
    ids = [[1, 2, 3], [], [32026] * 5]
    assert tdist.unpack_ids(tdist.pack_ids(ids, 8)) == ids
    assert tdist.gather_outputs(ids, 8, 3) == [ids]
    assert tdist.shard_indices(10, 1, 4) == [1, 5, 9]
