"""STC connector (legacy trace.infer() path; north_star names it) on the GPU against the oracle's restatement.
Partly pinned (round 4): tests/golden/stc_connector.npz is the output of the REFERENCE's own STCConnector.forward — its layouts, Conv3d sampler,
GELU readout and token order — with timm's RegStage (not importable in the build container) replaced by the restated block; the RegStage block
arithmetic itself follows timm 0.6.x from memory and stays UNPINNED."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from oracle import trace_oracle as O  # noqa: E402
from trace_amd import config as tcfg, synth  # noqa: E402
from trace_amd.engine import TraceEngine  # noqa: E402


def test_stc_connector_vs_oracle():
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), mm_projector_type="stc_connector", vision_image_size=84,   # 6x6 patch grid
                              vision_hidden_size=256, vision_num_heads=4, mm_hidden_size=256)             # SE reduce width 64
    sd = synth.state_dict(cfg)
    eng = TraceEngine(cfg, max_batch=1, max_ctx=256, max_frames=4, max_new_tokens=8)
    eng.load_weights(sd.items())
    ora = O.Oracle(cfg, sd, emulate_bf16=True)
    g = torch.Generator().manual_seed(3)
    feats = (torch.randn(4, cfg.vision_patches, cfg.vision_hidden_size, generator=g)).to(torch.bfloat16)
    got = eng.stc_connector(feats.cuda(), 4)
    ref = ora.stc_connector(feats.float())
    assert got.shape == ref.shape == ((4 // 2 + 1) * 4 * 4, cfg.hidden_size)
    err = (got.float().cpu() - ref).abs()
    tol = 6e-2 + 6e-2 * ref.abs()
    assert torch.isfinite(got).all()
    assert (err > tol).float().mean().item() < 5e-3, f"max err {err.max().item()}, ref max {ref.abs().max().item()}"
    # and against the reference's own STCConnector.forward on the same input (tests/golden/stc_connector.npz: the reference's layouts, Conv3d sampler
    # and GELU readout; RegStage restated) — the same budget
    import os
    import numpy as np
    M = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stc_connector.npz"))
    assert int(M["seed"]) == 3
    fx = torch.from_numpy(M["out"])
    gc = got.float().cpu()[:, torch.from_numpy(M["out_cols"])]
    errf = (gc - fx).abs()
    assert (errf > 6e-2 + 6e-2 * fx.abs()).float().mean().item() < 5e-3, f"vs reference forward: max err {errf.max().item()}"
    # legacy flow: ViT -> STC -> splice (no time tokens) -> text-head decode
    frames = synth.synth_frames(cfg, 0).to(torch.bfloat16)
    eng.vit_forward(frames)
    vid = eng.stc_connector(None, 4)
    ids = synth.synth_prompt_ids(cfg, n_text=16, video_pos=5).tolist()
    ids[-1] = 7                                   # legacy prompts carry no <sync>
    L = eng.splice(ids)
    assert L == 15 + vid.shape[0]
    eng.prefill(0, L)
    eng.decode_begin([0], [0], 6)
    eng.decode_steps(5)
    out, heads = eng.decode_read()
    assert len(out[0]) == 6 and all(0 <= t <= cfg.vocab_size for t in out[0])
    eng.close()


def test_stc_connector_at_real_width_vs_oracle():
    """Round 5: the connector at the widths the reference builds it with (mm_hidden 1024 -> hidden 4096, 24 x 24 patch grid; builder.py:138-205), two
    frames (-> 2 x 13 x 13 = 338 tokens): every GEMM of the path at its production K / N — the 1 x 1 convolutions (K = 1024 and 4096), the Conv3d sampler as
    a K = 32768 GEMM, the readout — the depthwise convolutions and squeeze-excite GEMVs at 4096 channels, and the fused LayerNorm + shortcut + SiLU tail
    of every block, against the bf16-emulating oracle (the RegStage block itself restated: DESIGN section 2)."""
    cfg = dataclasses.replace(tcfg.tiny(num_frames=2), mm_projector_type="stc_connector", vision_image_size=336, vision_patch_size=14,
                              vision_hidden_size=1024, vision_intermediate_size=1024, vision_num_heads=16, vision_num_layers=2, mm_hidden_size=1024)
    assert cfg.hidden_size == 4096 and cfg.vision_grid == 24
    sd = synth.state_dict(cfg)
    eng = TraceEngine(cfg, max_batch=1, max_ctx=512, max_frames=2, max_new_tokens=8)
    eng.load_weights(sd.items())
    ora = O.Oracle(cfg, {k: v for k, v in sd.items() if "mm_projector" in k}, emulate_bf16=True)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2, cfg.vision_patches, cfg.vision_hidden_size, generator=g).to(torch.bfloat16)
    got = eng.stc_connector(feats.cuda(), 2)
    again = eng.stc_connector(feats.cuda(), 2)
    assert torch.equal(got, again)
    with torch.no_grad():
        ref = ora.stc_connector(feats.float())
    assert got.shape == ref.shape == (2 * 13 * 13, 4096)
    err = (got.float().cpu() - ref).abs()
    tol = 6e-2 + 6e-2 * ref.abs()
    frac = (err > tol).float().mean().item()
    print(f"STC at real width: max err {err.max().item():.4f} (ref max {ref.abs().max().item():.3f}, ref std {ref.std().item():.3f}), outside the budget: {frac:.5f}")
    assert torch.isfinite(got).all()
    assert frac < 5e-3, (err.max().item(), ref.abs().max().item(), frac)
    eng.close()


def test_legacy_infer_api(tmp_path):
    """trace.model_init / trace.infer (trace/__init__.py:13-75) with an STC checkpoint: text-head ids only."""
    import numpy as np
    import trace_amd
    from trace_amd.model.builder import save_synthetic_checkpoint
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), mm_projector_type="stc_connector", vision_image_size=84,
                              vision_hidden_size=256, vision_num_heads=4, mm_hidden_size=256)
    path = save_synthetic_checkpoint(str(tmp_path / "trace-stc-tiny"), cfg)
    model, processor, tokenizer = trace_amd.model_init(path, max_new_tokens=16)
    raw = np.random.RandomState(0).randint(0, 255, size=(12, 40, 56, 3), dtype=np.uint8)
    video, _ = processor(raw, fps=4.0)
    assert video.shape == (4, 3, 84, 84)
    out = trace_amd.infer(model, video, "what happens?", tokenizer, max_new_tokens=6)
    assert out.shape[0] == 1 and out.shape[1] <= 6
    assert int(out.max()) <= cfg.vocab_size            # text head + <sync> only: no time/score ids on the legacy path
    model.engine.close()
