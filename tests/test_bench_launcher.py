"""`python bench.py --gpus N` launches its own ranks (round-1 VERDICT: a driver-style `python3 bench.py --gpus 8`
silently ran one rank).  CPU / gloo: the launcher, the rendezvous on 127.0.0.1, WORLD_SIZE checks and the packed-id all-gather
run for real through `--plumbing-check` (no kernels); without that flag the launcher refuses a GPU count the node does not have."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TRACE_FORCE_PG"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)


def test_self_launch_two_ranks_gloo():
    r = _run(["--gpus", "2", "--plumbing-check"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["gather_ok"] is True


def test_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "visible" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]        # no JSON line with a wrong n_gpus


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--plumbing-check"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
