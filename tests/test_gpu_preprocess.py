"""On-device frame preprocessing (trace_preprocess_frames) vs the fixtures captured from the reference's expand2square +
the HF CLIPImageProcessor it calls, and vs the oracle (Pillow restatement) at the real 336-pixel geometry: fp32 output is
bit-exact; bf16 output is the round-to-nearest-even of it."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from oracle import trace_oracle as O  # noqa: E402  (checker only)
from trace_amd import config as tcfg, mm_utils  # noqa: E402
from trace_amd.engine import TraceEngine  # noqa: E402


@pytest.fixture(scope="module")
def eng56():
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), vision_image_size=56)          # fixtures are 56 px
    e = TraceEngine(cfg, max_batch=1, max_ctx=128, max_frames=4, max_new_tokens=8)    # no weights needed for preprocessing
    yield e
    e.close()


def test_matches_reference_fixture(eng56, golden_dir):
    G = np.load(os.path.join(golden_dir, "preprocess.npz"))
    mean, std = G["image_mean"].tolist(), G["image_std"].tolist()
    for tag in ("land", "port", "square", "up"):
        for mode in ("pad", "plain"):
            ref = torch.from_numpy(G[f"{tag}_{mode}"])
            got = eng56.preprocess_frames(G[f"{tag}_frames"], pad=(mode == "pad"), image_mean=mean, image_std=std, dtype=torch.float32)
            assert torch.equal(got.cpu(), ref), f"{tag}/{mode}: max diff {(got.cpu() - ref).abs().max().item()}"
            got16 = eng56.preprocess_frames(G[f"{tag}_frames"], pad=(mode == "pad"), image_mean=mean, image_std=std)
            assert torch.equal(got16.cpu(), ref.to(torch.bfloat16))


@pytest.mark.parametrize("H,W", [(360, 640), (480, 270), (100, 100), (37, 1000)])
def test_336_geometry_vs_oracle(H, W):
    cfg = tcfg.tiny(num_frames=2)
    cfg = dataclasses.replace(cfg, vision_image_size=336, vision_patch_size=14)
    eng = TraceEngine(cfg, max_batch=1, max_ctx=2048, max_frames=2, max_new_tokens=8)
    rng = np.random.RandomState(H * 7 + W)
    frames = rng.randint(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    frames[1] = (np.add.outer(np.arange(H), np.arange(W)) % 256)[..., None].astype(np.uint8)
    for pad in (True, False):
        ref = torch.from_numpy(O.preprocess_frames(frames, eng.CLIP_MEAN, eng.CLIP_STD, 336, pad))
        got = eng.preprocess_frames(frames, pad=pad, dtype=torch.float32)
        assert torch.equal(got.cpu(), ref), f"pad={pad}: {(got.cpu() - ref).abs().max().item()}"
    eng.close()


def test_process_video_device_path(eng56):
    """mm_utils.process_video(engine=...) returns the same frames/timestamps as the host path (bf16 of it)."""
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={"shortest_edge": 56}, crop_size={"height": 56, "width": 56})
    raw = np.random.RandomState(3).randint(0, 256, size=(20, 48, 80, 3), dtype=np.uint8)
    host, ts_h = mm_utils.process_video(raw, proc, aspect_ratio="pad", num_frames=4, fps=5.0)
    dev, ts_d = mm_utils.process_video(raw, proc, aspect_ratio="pad", num_frames=4, fps=5.0, engine=eng56)
    assert ts_h == ts_d and dev.is_cuda and dev.dtype == torch.bfloat16
    assert torch.equal(dev.cpu(), host.to(torch.bfloat16))


def test_plain_c_client_matches_python(eng56, tmp_path):
    """examples/preprocess_demo.c (gcc, no Python, no torch) calls trace_preprocess_frames through the C ABI; its output
    checksum equals the one computed from the Python path on the same pixels."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    exe = str(tmp_path / "preprocess_demo")
    r = subprocess.run(["gcc", "-O1", "-std=c99", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                        os.path.join(root, "examples", "preprocess_demo.c"), "-L", os.path.join(root, "trace_amd"), "-ltrace_hip",
                        "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(root, "trace_amd"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    T, H, W = 3, 70, 120
    run = subprocess.run([exe, str(T), str(H), str(W)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    line = [l for l in run.stdout.splitlines() if "checksum" in l][0]
    c_sum = int(line.split()[-1])
    i = np.arange(T * H * W * 3, dtype=np.uint64)
    frames = (((i * np.uint64(2654435761)) >> np.uint64(24)) & np.uint64(0xFF)).astype(np.uint8).reshape(T, H, W, 3)
    out = eng56.preprocess_frames(frames, pad=True).cpu().view(torch.int16).numpy().astype(np.uint16).reshape(-1)
    s = 0
    for v in out.tolist():
        s = (s * 1000003 + v) & 0xFFFFFFFFFFFFFFFF
    assert s == c_sum
    assert "error path" in run.stdout
