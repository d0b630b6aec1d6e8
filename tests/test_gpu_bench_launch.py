"""bench.py under the driver's launcher (torch.distributed.run, one rank per GPU) on the single GPU of the test box:
the process group is forced on (TRACE_FORCE_PG=1) so RCCL init + the all-gather of packed ids really run."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_tiny_under_torchrun():
    env = dict(os.environ, TRACE_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--tiny",
           "--frames", "4", "--videos-per-step", "3", "--max-new", "12"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "videos/s" and d["scaling"] == "weak"
    assert d["config"]["videos_per_step_per_gpu"] == 3
    assert "roofline" in d and "cpu_baseline" in d
