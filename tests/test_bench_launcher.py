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
    # the per-rank collective of the real line: one entry per rank, in rank order; each rank pinned to its own slice of the CPUs
    pr = d["per_rank"]
    assert pr["rank"] == [0, 1] and len(pr["ms_per_step"]) == 2 and all(x > 0 for x in pr["ms_per_step"])
    n_cpu = len(os.sched_getaffinity(0))
    if n_cpu >= 2:
        assert pr["cpus"] == [n_cpu // 2, n_cpu // 2], pr
        assert "slice 1/2" in pr["host_placement_rank0"]["how"]


def test_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "visible" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]        # no JSON line with a wrong n_gpus


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--plumbing-check"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_affinity_plan_and_sysfs_parsing(tmp_path):
    """Host placement of the ranks (trace_amd/dist.py): NUMA node of a GPU's PCI function and that node's CPUs from a (fake) sysfs tree; the plan =
    the node's CPUs when known, else an even slice of the allowed CPUs per rank; a single rank is left alone."""
    from trace_amd import dist as td
    assert td.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and td.parse_cpulist("") == []
    pci = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    pci.mkdir(parents=True)
    (pci / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("16-31,80-95\n")
    assert td.numa_node_of_pci("0000:C1:00.0", str(tmp_path)) == 1
    assert td.numa_node_of_pci("0000:ff:00.0", str(tmp_path)) == -1
    cpus = td.cpus_of_numa_node(1, str(tmp_path))
    assert cpus == list(range(16, 32)) + list(range(80, 96)) and td.cpus_of_numa_node(7, str(tmp_path)) == []
    allowed = list(range(0, 128))
    got, how = td.plan_affinity(allowed, 3, 8, cpus)
    assert got == cpus and "NUMA node" in how
    got, how = td.plan_affinity(list(range(0, 24)), 3, 8, cpus)        # cgroup-limited: only CPUs 16..23 of the node are allowed
    assert got == list(range(16, 24))
    got, how = td.plan_affinity(allowed, 3, 8, [])                      # node unknown: slice 4 of 8
    assert got == list(range(48, 64)) and "slice 4/8" in how
    slices = [td.plan_affinity(allowed, r, 8, [])[0] for r in range(8)]
    assert sorted(c for sl in slices for c in sl) == allowed            # disjoint, complete
    assert td.plan_affinity(allowed, 0, 1, cpus)[0] == allowed          # one rank: unchanged
    assert td.plan_affinity([0, 1, 2], 1, 8, [])[0] == [0, 1, 2]        # fewer CPUs than ranks: unchanged
    assert td.gather_floats([1.5, 2]) == [[1.5, 2.0]]                   # no process group: this rank only


def test_c1_cpu_leg_runs_here():
    """BASELINE config 1's CPU leg ("C1 runs fully on CPU end-to-end"): with no HIP device the line still comes out, from the
    oracle alone — clip file -> process_video -> llama_2 prompt -> tokenizer_MMODAL_token_all -> 32 greedy ids starting on the time head."""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: tests/test_gpu_bench_launch.py runs both legs")
    r = _run(["--config", "c1", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 0 and d["value"] is None and "CPU leg only" in d["config"]["note"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and len(cb["ids"]) == 32
    assert 320 < cb["ids"][0] <= 320 + 13                       # generation starts on the time head (heads=[1]); tiny vocabulary V = 320
