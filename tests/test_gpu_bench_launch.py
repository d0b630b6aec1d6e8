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
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tiny",
           "--frames", "4", "--videos-per-step", "3", "--max-new", "12", "--pipeline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "videos/s" and d["scaling"] == "weak"
    assert d["config"]["videos_per_step_per_gpu"] == 3
    assert "roofline" in d and "cpu_baseline" in d
    assert d["steps_repeat_exactly"] is True, d.get("steps_repeat_detail")


def test_bench_c1_single_clip_from_a_file():
    """BASELINE config 1 on the GPU box: a clip FILE (y4m) -> process_video -> prompt -> generate through the drop-in surface, 8 frames, 32 greedy
    tokens; the ids equal the oracle's (the oracle that rounds where the engine stores bf16) at least up to its first near-tie, the CPU leg (the
    oracle end to end, fp32) is timed beside it, and the line carries the new per-shape roofline fields of the default config's line format."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["baseline_config"] == "c1" and d["config"]["frames"] == 8
    ids = d["ids"]
    assert len(ids["hip"]) == 32 == len(ids["oracle_16bit_emulating"])
    if not ids["equal"]:
        assert ids["oracle_top2_margin_there"] < 0.1, ids                 # a near-tie of the random-weight text head may fall either way
    assert ids["first_difference"] is None or ids["first_difference"] >= 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and "no extrapolation" in d["cpu_baseline"]["sample"]
    assert set(d["parsed"]) >= {"timestamps", "scores", "captions"}


def test_evaluate_driver_under_torchrun(tmp_path):
    """python -m trace_amd.evaluate under torch.distributed.run: sharding, device preprocessing, batched decode, the RCCL
    gather of packed ids and the parser, end to end on a synthetic tiny checkpoint and .npy frame files."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from trace_amd import config as tcfg
    from trace_amd.model.builder import save_synthetic_checkpoint
    ckpt = save_synthetic_checkpoint(str(tmp_path / "trace-tiny"), tcfg.tiny(num_frames=4))
    items = []
    for i in range(3):
        f = str(tmp_path / f"v{i}.npy")
        np.save(f, np.random.RandomState(i).randint(0, 255, size=(24, 40, 56, 3), dtype=np.uint8))
        items.append({"id": i, "video": f, "fps": 8.0})
    (tmp_path / "items.json").write_text(json.dumps(items))
    env = dict(os.environ, TRACE_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", "-m", "trace_amd.evaluate", "--model", ckpt, "--items", str(tmp_path / "items.json"),
           "--out", str(tmp_path / "out.json"), "--num-frames", "4", "--max-new-tokens", "8", "--batch-size", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads((tmp_path / "out.json").read_text())
    assert [x["id"] for x in res] == [0, 1, 2] and all(1 <= len(x["output_ids"]) <= 8 for x in res)
    assert all(x["video"].endswith(".npy") for x in res)
