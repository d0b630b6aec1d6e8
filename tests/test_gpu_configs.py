"""BASELINE.json's configurations beyond C2 on the GPU, each against logits captured from the reference
(oracle/make_goldens.py; fixtures under tests/golden/):

  C4  Charades-STA moment retrieval  — 64 frames, prefill L = 1086, 32 new tokens, 18 of them on the time / score heads
      (trace/eval/evaluate.py:298-357 with prompts/mr.txt)                                         charades_ctx.npz
  C5  VideoMME long video (shape)    — 256 frames (past constants.MAX_FRAMES, as trace/eval/videomme/evaluate.py:215-258
      samples), prefill L = 3834, 16 tokens, bf16; the fp8 weight path of C5 is checked against this
      bf16 result (it has no reference counterpart)                                                 videomme_ctx.npz
  full depth — all 32 decoder layers at the real Mistral-7B widths, teacher-forced                  full_depth_llm.npz

Tolerances: LOGIT_TOL = 0.15 for the one-layer shapes (the budget of tests/test_gpu_parity.py); FULL_DEPTH_TOL for the
32-layer stack — independent bf16 roundings (weights, activations, KV cache; fp32 accumulation) grow like sqrt(depth):
0.13-0.14 measured at 8 layers -> 0.27 expected at 32, 0.32 measured (gpurun_out/parity_measured.txt); the budget is 0.40
absolute on logits of std 1.29 — anchored (round 4) on the reference's OWN bf16 run of the same stream, which deviates 0.49 max / 0.116 rms
from its fp32 run (`tf_logits_ref_bf16` in the fixture): the HIP path must stay below both figures.  Greedy ids must equal the reference's wherever its own top-2 margin exceeds twice the budget;
the 13-way time / score head ids are what the timestamps are made of and are checked at every step that qualifies."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from trace_amd import config as tcfg, synth  # noqa: E402
from trace_amd.engine import TraceEngine  # noqa: E402

LOGIT_TOL = 0.15
FULL_DEPTH_TOL = 0.40      # round 4: below the reference's own bf16-vs-fp32 deviation on this stream (0.49 max, tests/golden/full_depth_llm.npz:tf_logits_ref_bf16); HIP measured 0.29-0.32


def _teacher_forced(eng, nb, n, forced, graph_tail=False):
    """logits per step [n][nb, NV] (cpu) + emitted ids, stepping eagerly with the logits read back"""
    lgs = [eng.decode_begin(list(range(nb)), [1] * nb, n, eos=-1, forced=[forced] * nb, want_logits=True).float().cpu()]
    for _ in range(n - 1):
        lgs.append(eng.decode_steps(1, use_graph=False, want_logits=True).float().cpu())
    ids, heads = eng.decode_read()
    return lgs, ids, heads


def _check(lgs, ids, M, rows, tol, tag):
    ref_lg, ref_ids = torch.from_numpy(M["tf_logits"]), M["tf_argmax"].tolist()
    fin = torch.isfinite(ref_lg)
    srt = torch.sort(torch.where(fin, ref_lg, torch.full_like(ref_lg, -1e30)), dim=-1, descending=True).values
    margin = (srt[:, 0] - srt[:, 1]).tolist()
    worst = 0.0
    for b in rows:
        lg = torch.stack([x[b] for x in lgs])
        assert torch.equal(torch.isfinite(lg), fin), f"{tag}: head mask pattern differs (row {b})"
        err = (lg[fin] - ref_lg[fin]).abs().max().item()
        worst = max(worst, err)
        assert err < tol, (tag, b, err)
        for i, (a, r, m) in enumerate(zip(ids[b], ref_ids, margin)):
            if m > 2 * tol:
                assert a == r, (tag, b, i, a, r, m)
    print(f"{tag}: max |dlogit| = {worst:.4f} (budget {tol})")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):                                   # measured values travel back from the GPU box with gpurun_out/
        with open(os.path.join(out_dir, "parity_measured.txt"), "a") as f:
            f.write(f"{tag}: max |dlogit| = {worst:.4f} (budget {tol})\n")
    return worst


def test_c4_charades_moment_retrieval_vs_reference_fixture(golden_dir):
    cfg = dataclasses.replace(tcfg.tiny(num_frames=64), intermediate_size=14336, num_hidden_layers=1)
    M = np.load(os.path.join(golden_dir, "charades_ctx.npz"))
    eng = TraceEngine(cfg, max_batch=20, max_ctx=1152, max_frames=64, max_new_tokens=32)
    eng.load_weights(synth.state_dict(cfg).items())
    frames = synth.synth_frames(cfg, int(M["video_idx"])).to(torch.bfloat16)
    ts, ids, forced = M["timestamps"].tolist(), M["input_ids"].tolist(), M["forced_ids"].tolist()
    n = len(forced) + 1
    assert n == 32
    for nb in (1, 20):
        eng.encode_video(frames, ts)
        L, emb = eng.splice(ids, want_output=True)
        assert L == int(M["prefill_len"]) == 1086
        if nb == 1:
            eng.prefill(0, L)
        else:
            for b in range(0, nb, 2):
                eng.prefill_pair(b, emb, emb)
        lgs, out, heads = _teacher_forced(eng, nb, n, forced)
        _check(lgs, out, M, (0, nb - 1), LOGIT_TOL, f"C4 batch {nb}")
        assert all(len(o) == 32 for o in out)
        assert all(h == 1 for h in heads)                       # the stream ends on the text <sync>: back to the time head
    # the drivers' one-call path (hipGraph decode, batch of 3 videos through generate()): the reference's ids wherever its margin
    # allows (another batch size sums the GEMV partial rows in another grouping, so near-ties may fall either way)
    out_g, _ = eng.generate([frames] * 3, [ts] * 3, [ids] * 3, [1] * 3, n, eos=-1, use_graph=True, forced=[forced] * 3)
    ref_lg, ref_ids = torch.from_numpy(M["tf_logits"]), M["tf_argmax"].tolist()
    srt = torch.sort(torch.where(torch.isfinite(ref_lg), ref_lg, torch.full_like(ref_lg, -1e30)), dim=-1, descending=True).values
    checked = 0
    for b in (0, 2):
        assert len(out_g[b]) == 32
        for i, (a, r) in enumerate(zip(out_g[b], ref_ids)):
            if float(srt[i, 0] - srt[i, 1]) > 2 * LOGIT_TOL:
                assert a == r, (b, i, a, r)
                checked += 1
    assert checked >= 20
    eng.close()


def test_c5_shape_256_frames_vs_reference_fixture(golden_dir):
    cfg = dataclasses.replace(tcfg.tiny(num_frames=256), intermediate_size=14336, num_hidden_layers=1)
    M = np.load(os.path.join(golden_dir, "videomme_ctx.npz"))
    eng = TraceEngine(cfg, max_batch=4, max_ctx=3904, max_frames=256, max_new_tokens=16)
    eng.load_weights(synth.state_dict(cfg).items())
    frames = synth.synth_frames(cfg, int(M["video_idx"])).to(torch.bfloat16)
    ts, ids, forced = M["timestamps"].tolist(), M["input_ids"].tolist(), M["forced_ids"].tolist()
    n = len(forced) + 1
    for nb in (1, 4):
        eng.encode_video(frames, ts)
        L, emb = eng.splice(ids, want_output=True)
        assert L == int(M["prefill_len"]) == 3834
        if nb == 1:
            eng.prefill(0, L)
        else:
            for b in range(0, nb, 2):
                eng.prefill_pair(b, emb, emb)                      # M = 7668
        lgs, out, _ = _teacher_forced(eng, nb, n, forced)
        _check(lgs, out, M, (0, nb - 1), LOGIT_TOL, f"C5 shape batch {nb}")
    eng.close()


def test_full_depth_32_layers_vs_reference_fixture(golden_dir):
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=32)
    M = np.load(os.path.join(golden_dir, "full_depth_llm.npz"))
    eng = TraceEngine(cfg, max_batch=2, max_ctx=192, max_frames=4, max_new_tokens=64)
    eng.load_weights(synth.iter_weights(cfg))                       # the canonical (CPU-stream) weights the fixture was made with
    frames = synth.synth_frames(cfg, 0).to(torch.bfloat16)
    ts, ids, forced = M["timestamps"].tolist(), M["input_ids"].tolist(), M["forced_ids"].tolist()
    n = len(forced) + 1
    for b in range(2):
        eng.encode_video(frames, ts)
        eng.prefill(b, eng.splice(ids))
    lgs, out, _ = _teacher_forced(eng, 2, n, forced)
    worst = _check(lgs, out, M, (0, 1), FULL_DEPTH_TOL, "full depth (32 layers)")
    # time / score head ids (13-way): where the reference's margin allows, every one of them equals the reference's
    V = cfg.vocab_size
    ref_ids = M["tf_argmax"].tolist()
    assert sum(r > V for r in ref_ids) >= 20
    assert worst > 0.0
    # the anchor for the budget (round 4): the reference's OWN bf16 run of this stream (model.to(bfloat16), layer-streamed) is 0.49 max / 0.116 rms
    # from its fp32 run; the HIP path (fp32 accumulation, bf16 only at the storage points) must be no further from fp32 than that, and must not
    # change more 13-way decisions than the reference's bf16 run does
    from conftest import bf16_anchor_report
    r = bf16_anchor_report(torch.stack([x[0] for x in lgs]), M, "full depth (32 layers), bf16 anchor")
    assert r["hip_rms"] <= r["ref_bf16_rms"] and r["hip_max"] <= r["ref_bf16_max"], r
    assert r["hip_flips_13way"] <= r["ref_bf16_flips_13way"] + 2 and r["hip_flips_13way_margin_gt_0.5"] <= r["ref_bf16_flips_13way_margin_gt_0.5"] + 1, r
    eng.close()
