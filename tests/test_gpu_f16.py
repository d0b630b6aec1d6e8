"""The fp16 library (libtrace_hip_f16.so: the same kernels compiled with -DTRACE_F16 — IEEE half elements, fp32 accumulation; the reference's own
inference dtype, trace/model/builder.py:50,127,147) against the oracle and the fixture captured from the reference with fp16-rounded weights
(tests/golden/tiny_e2e_f16.npz: the reference in fp32 arithmetic, plus its own model.half() run).

Tolerances, a priori: an fp16 rounding is 2^-11 relative, a bf16 one 2^-8, so every budget of tests/test_gpu_parity.py shrinks by 8: activations
against the fp16-emulating oracle atol = rtol = 3e-2 / 8 ~ 4e-3 (used: 5e-3); logits against the fp32-arithmetic fixture 0.15 / 8 ~ 0.02 (used: 0.03,
the same number that bounds the reference's own fp16 run against its fp32 run in tests/test_oracle_golden.py).

First run on hardware at the end of round 3: 9 passed (gpurun_out -> profiles/r03_f16_tests.txt)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from oracle import trace_oracle as O  # noqa: E402  (checker only)
from trace_amd import config as tcfg, synth  # noqa: E402
from trace_amd.engine import TraceEngine, ops, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, EPI_QUICKGELU  # noqa: E402

F16 = torch.float16
ACT_ATOL, ACT_RTOL = 5e-3, 5e-3
LOGIT_TOL = 0.03


def report(name, got, ref, atol, rtol):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    frac = bad.float().mean().item()
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    assert frac < 2e-3 and err.max().item() < 20 * (atol + rtol * ref.abs().max().item()), (
        f"{name}: {frac:.4%} elements off, max err {err.max().item():.4g}, ref max {ref.abs().max().item():.4g}, "
        f"first bad {torch.nonzero(bad)[0].tolist() if bad.any() else None}")


@pytest.fixture(scope="module")
def setup(golden_dir):
    cfg = tcfg.tiny(num_frames=4)
    sd = synth.state_dict(cfg, dtype=F16)
    eng = TraceEngine(cfg, max_batch=4, max_ctx=256, max_frames=4, max_new_tokens=64, dtype=F16)
    assert eng.lib.trace_element_type() == 1
    eng.load_weights(sd.items())
    ora = O.Oracle(cfg, sd, emulate_bf16=F16)
    E = np.load(os.path.join(golden_dir, "tiny_e2e_f16.npz"))
    frames = synth.synth_frames(cfg, 0).to(F16)
    yield cfg, eng, ora, E, frames
    eng.close()


def test_vit_slots_splice_prefill(setup):
    cfg, eng, ora, E, frames = setup
    got = eng.vit_forward(frames)
    assert got.dtype == F16
    report("vit vs oracle(fp16)", got, ora.vit_forward(frames.float()), ACT_ATOL, ACT_RTOL)
    report("vit vs reference fixture", got, torch.from_numpy(E["vit_feats"]), 1e-2, 1e-2)
    slots = eng.slot_pool(None, frames.shape[0])
    report("slot pool vs reference fixture", slots, torch.from_numpy(E["slots"]), 1e-2, 1e-2)
    ts = E["timestamps"].tolist()
    vid = eng.encode_video(frames, ts, want_output=True)
    ref_vid = ora.encode_video(frames.float(), ts)
    report("video rows", vid, ref_vid, ACT_ATOL, ACT_RTOL)
    T, S = frames.shape[0], cfg.num_slots
    v3, r3 = vid.view(T, cfg.tokens_per_frame, -1).float().cpu(), ref_vid.view(T, cfg.tokens_per_frame, -1)
    assert torch.equal(v3[:, S:], r3[:, S:].to(F16).float())                      # time-token rows are pure gathers
    ids = E["input_ids"].tolist()
    L, emb = eng.splice(ids, want_output=True)
    assert L == int(E["prefill_len"])
    report("spliced embeds", emb, ora.splice(torch.tensor(ids), ref_vid), ACT_ATOL, ACT_RTOL)
    hid = eng.prefill(0, L, want_hidden=True)
    ref_hid, _ = ora.llm_forward(ora.splice(torch.tensor(ids), ref_vid))
    report("prefill hidden", hid, ref_hid, 1e-2, 1e-2)
    report("prefill hidden vs reference fixture", hid[-4:], torch.from_numpy(E["hidden_last_rows"]), 2e-2, 2e-2)


def _run_forced(eng, E, frames, forced):
    eng.encode_video(frames, E["timestamps"].tolist())
    L = eng.splice(E["input_ids"].tolist())
    eng.prefill(1, L)
    n = len(forced) + 1
    logits = [eng.decode_begin([1], [1], n, eos=-1, forced=[forced], want_logits=True).cpu()]
    for _ in range(n - 1):
        logits.append(eng.decode_steps(1, use_graph=False, want_logits=True).cpu())
    ids, _ = eng.decode_read()
    return ids[0], torch.cat(logits)


def test_teacher_forced_logits_and_ids(setup):
    cfg, eng, ora, E, frames = setup
    forced = E["forced_ids"].tolist()
    ids, lg = _run_forced(eng, E, frames, forced)
    ref_lg, ref_ids = torch.from_numpy(E["tf_logits"]), E["tf_argmax"].tolist()
    fin = torch.isfinite(ref_lg)
    assert torch.equal(torch.isfinite(lg), fin), "head mask (-inf pattern) differs from the reference"
    err = (lg[fin] - ref_lg[fin]).abs().max().item()
    assert err < LOGIT_TOL, f"logits differ from the reference (fp32 arithmetic, fp16 weights) by {err}"
    if "tf_logits_ref_fp16" in E:
        e16 = (lg[fin] - torch.from_numpy(E["tf_logits_ref_fp16"])[fin]).abs().max().item()
        assert e16 < 0.04, f"logits differ from the reference's own fp16 run by {e16}"
    srt = torch.sort(torch.where(fin, ref_lg, torch.full_like(ref_lg, -1e30)), dim=-1, descending=True).values
    margin = (srt[:, 0] - srt[:, 1]).tolist()
    checked = 0
    for i, (a, b, m) in enumerate(zip(ids, ref_ids, margin)):
        if m > 2 * LOGIT_TOL:
            assert a == b, f"step {i}: id {a} != reference {b} (margin {m:.3f})"
            checked += 1
    assert checked >= len(ref_ids) // 2
    o_ids, o_lg = ora.generate(torch.from_numpy(E["input_ids"]), frames.float(), E["timestamps"].tolist(), head=1,
                               max_new_tokens=len(forced) + 1, forced_ids=forced, return_logits=True)
    V = cfg.vocab_size
    osrt = torch.sort(torch.where(torch.isfinite(o_lg), o_lg, torch.full_like(o_lg, -1e30)), dim=-1, descending=True).values
    for i, (a, b) in enumerate(zip(ids, o_ids)):
        if b > V and (osrt[i, 0] - osrt[i, 1]) > 0.005:
            assert a == b, f"step {i}: time/score id {a} != oracle {b}"
    assert (lg[fin] - o_lg[fin]).abs().max().item() < 0.015


def test_free_run_graph_equals_eager_and_batches(setup):
    cfg, eng, ora, E, frames = setup
    ts, ids = E["timestamps"].tolist(), E["input_ids"].tolist()
    f2 = synth.synth_frames(cfg, 1).to(F16)
    n = 16
    a, _ = eng.generate([frames], [ts], [ids], [1], n, use_graph=True)
    a2, _ = eng.generate([frames], [ts], [ids], [1], n, use_graph=False)
    assert a == a2
    ref = E["free_ids"].tolist()
    assert a[0][:4] == ref[:4]
    b, _ = eng.generate([f2], [ts], [ids], [1], n)
    ab, _ = eng.generate([frames, f2], [ts, ts], [ids, ids], [1, 1], n)
    assert ab[0] == a[0] and ab[1] == b[0]
    for nb in (40, 100):                          # NB = 4 decode GEMV; the wide decode step (projections as small-M MFMA GEMMs)
        big = TraceEngine(cfg, max_batch=nb, max_ctx=192, max_frames=4, max_new_tokens=32, dtype=F16)
        big.load_weights(synth.state_dict(cfg, dtype=F16).items())
        vids = [frames if i % 2 == 0 else f2 for i in range(nb)]
        out, _ = big.generate(vids, [ts] * nb, [ids] * nb, [1] * nb, n, use_graph=True)
        for i in range(nb):
            assert out[i] == (a[0] if i % 2 == 0 else b[0]), (nb, i)
        big.close()


def test_both_libraries_side_by_side(setup, golden_dir):
    """One process, two element types: a bf16 engine created and run while the fp16 engine is alive gives the bf16 fixture's answer, and the
    fp16 engine still gives its own afterwards (the two .so files export the same symbols; RTLD_LOCAL keeps their globals apart)."""
    cfg, eng, ora, E, frames = setup
    ts, ids = E["timestamps"].tolist(), E["input_ids"].tolist()
    a16, _ = eng.generate([frames], [ts], [ids], [1], 12)
    e_bf = TraceEngine(cfg, max_batch=1, max_ctx=192, max_frames=4, max_new_tokens=16)
    assert e_bf.lib is not eng.lib and e_bf.lib.trace_element_type() == 0
    e_bf.load_weights(synth.state_dict(cfg).items())
    Eb = np.load(os.path.join(golden_dir, "tiny_e2e.npz"))
    got = e_bf.vit_forward(synth.synth_frames(cfg, 0).to(torch.bfloat16))
    report("bf16 library next to the fp16 one", got, torch.from_numpy(Eb["vit_feats"]), 6e-2, 6e-2)
    abf, _ = e_bf.generate([synth.synth_frames(cfg, 0).to(torch.bfloat16)], [ts], [ids], [1], 12)
    assert abf[0][:4] == Eb["free_ids"].tolist()[:4]
    e_bf.close()
    again, _ = eng.generate([frames], [ts], [ids], [1], 12)
    assert again == a16
    with pytest.raises(ValueError, match="bf16 library only"):
        TraceEngine(cfg, max_batch=1, max_ctx=192, max_frames=4, max_new_tokens=16, dtype=F16, llm_fp8=True)


@pytest.fixture()
def f16_ops():
    ops.use("f16")
    yield ops
    ops.use("bf16")


@pytest.mark.parametrize("M,N,K,epi", [(300, 256, 192, EPI_NONE), (1111, 512, 256, EPI_QUICKGELU), (2304, 1024, 1024, EPI_RESIDUAL),
                                       (2100, 512, 1024, EPI_SWIGLU), (98, 384, 128, EPI_NONE)])
def test_gemm_kernels_f16(f16_ops, M, N, K, epi):
    """every GEMM kernel family at its own shapes (128^2 LDS-DMA tiles, the 256^2 loader-wave and persistent kernels): fp16 operands, fp32
    accumulation, one fp16 rounding of the result — against torch in fp32 on the same fp16 operands."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(F16)
    W = (torch.randn((N, K), device="cuda", generator=g) * (1.0 / K ** 0.5)).to(F16)
    bias = (torch.randn((N,), device="cuda", generator=g) * 0.1).to(F16) if epi != EPI_SWIGLU else None
    R = (torch.randn((M, N), device="cuda", generator=g)).to(F16) if epi == EPI_RESIDUAL else None
    got = f16_ops.gemm(A, W, bias=bias, R=R, epilogue=epi)
    assert got.dtype == F16
    ref = A.float() @ W.float().t()
    if bias is not None:
        ref = ref + bias.float()
    if epi == EPI_QUICKGELU:
        ref = ref * torch.sigmoid(1.702 * ref)
    elif epi == EPI_RESIDUAL:
        ref = ref + R.float()
    elif epi == EPI_SWIGLU:
        # gate|up interleaved in 16-row groups (the engine's load-time layout): out[:, 16 j + c] = silu(acc[:, 32 j + c]) * acc[:, 32 j + 16 + c]
        r3 = ref.view(M, N // 32, 2, 16)
        ref = (torch.nn.functional.silu(r3[:, :, 0]) * r3[:, :, 1]).reshape(M, N // 2)
    err = (got.float() - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err


def test_vit_large_geometry_f16():
    """The real CLIP-ViT-L/14-336 geometry in fp16: one frame, and the frame 24 times over (the 256-wide kernels' path), against the fp32 oracle on the
    same fp16-rounded weights (the oracle is pinned to the reference at this geometry by tests/golden/medium_vit.npz).  Budget: the bf16 test's
    (max 0.5, relative L2 2 %) divided by 4 — the a-priori factor is 8, half of it is kept as margin."""
    import dataclasses
    cfg = dataclasses.replace(tcfg.tiny(num_frames=1), vision_hidden_size=1024, vision_intermediate_size=4096, vision_num_layers=24,
                              vision_num_heads=16, vision_image_size=336, vision_patch_size=14, mm_hidden_size=1024)
    sd = {k: v for k, v in synth.state_dict(cfg, dtype=F16).items()}
    frames = synth.synth_frames(cfg, 7, num_frames=1).to(F16)
    ora = O.Oracle(cfg, {k: v for k, v in sd.items() if "vision_tower" in k or "mm_projector" in k}, emulate_bf16=False)
    ref = ora.vit_forward(frames.float())[0]
    eng = TraceEngine(cfg, max_batch=1, max_ctx=512, max_frames=24, max_new_tokens=8, dtype=F16)
    eng.load_weights(sd.items())
    one = eng.vit_forward(frames).float().cpu()[0]
    many = eng.vit_forward(frames.expand(24, -1, -1, -1).contiguous()).float().cpu()
    assert all(torch.equal(many[0], many[i]) for i in range(1, 24))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    for name, f in (("1 frame", one), ("24 frames", many[0])):
        e = (f - ref).abs()
        rl2 = (f - ref).norm().item() / ref.norm().item()
        msg = f"fp16 ViT-L/14-336 real geometry, {name}: rel L2 vs fp32 oracle {rl2:.5f}, max {e.max().item():.4f}, mean {e.mean().item():.5f} (ref abs mean {ref.abs().mean().item():.3f})"
        print(msg)
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "parity_measured.txt"), "a") as fh:
                fh.write(msg + "\n")
        assert torch.isfinite(f).all() and e.max().item() < 0.125 and rl2 < 0.005, msg
    eng.close()


@pytest.mark.parametrize("name,layers,tol,tol16", [("medium_llm_f16.npz", 1, 0.03, 0.05), ("deep_llm_f16.npz", 8, 0.06, 0.10)])
def test_real_width_layers_f16(golden_dir, name, layers, tol, tol16):
    """1 and 8 decoder layers at the real Mistral-7B widths in fp16, teacher-forced, alone and inside batches of 40 / 100 (decode GEMV with four row
    groups; the wide decode step): logits against the reference's fp32 arithmetic on the same fp16 weights, and against the reference's own
    model.half() run.  Budgets: the reference's fp16 run is itself 0.019 / 0.037 from its fp32 run (tests/test_oracle_golden.py); the library, with
    fp32 accumulation everywhere, gets 1.5x that against fp32 and the sum of both against the reference's fp16."""
    import dataclasses
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=layers)
    M = np.load(os.path.join(golden_dir, name))
    eng = TraceEngine(cfg, max_batch=100, max_ctx=192, max_frames=4, max_new_tokens=64, dtype=F16)
    eng.load_weights(synth.iter_weights(cfg, dtype=F16))
    frames = synth.synth_frames(cfg, 0).to(F16)
    forced, ref_lg, ref_ids = M["forced_ids"].tolist(), torch.from_numpy(M["tf_logits"]), M["tf_argmax"].tolist()
    ref16 = torch.from_numpy(M["tf_logits_ref_fp16"]) if "tf_logits_ref_fp16" in M else None
    n = len(forced) + 1
    fin = torch.isfinite(ref_lg)
    srt = torch.sort(torch.where(fin, ref_lg, torch.full_like(ref_lg, -1e30)), dim=-1, descending=True).values
    margin = (srt[:, 0] - srt[:, 1]).tolist()
    for nb in (1, 40, 100):
        for b in range(nb):
            eng.encode_video(frames, M["timestamps"].tolist())
            eng.prefill(b, eng.splice(M["input_ids"].tolist()))
        lgs = [eng.decode_begin(list(range(nb)), [1] * nb, n, eos=-1, forced=[forced] * nb, want_logits=True).float().cpu()]
        for _ in range(n - 1):
            lgs.append(eng.decode_steps(1, use_graph=False, want_logits=True).float().cpu())
        ids, _ = eng.decode_read()
        for b in (0, nb - 1):
            lg = torch.stack([x[b] for x in lgs])
            assert torch.equal(torch.isfinite(lg), fin)
            err = (lg[fin] - ref_lg[fin]).abs().max().item()
            assert err < tol, (name, nb, b, err)
            if ref16 is not None:
                e16 = (lg[fin] - ref16[fin]).abs().max().item()
                assert e16 < tol16, (name, nb, b, e16)
            for i, (a, r, m) in enumerate(zip(ids[b], ref_ids, margin)):
                if m > 2 * tol:
                    assert a == r, (name, nb, b, i, a, r, m)
    eng.close()


@pytest.mark.parametrize("name,layers", [("medium_llm_f16.npz", 1), ("deep_llm_f16.npz", 8)])
def test_fp16_checkpoint_argmax_agreement(golden_dir, name, layers):
    """What the fp16 library is for: an fp16 checkpoint, and the reference's own fp16 run (model.half()) as the answer key.  The same fp16 weights go
    into (a) the fp16 engine as they are and (b) the bf16 engine, which casts them at load (what round 2 shipped).  Teacher-forced over a stream that
    visits all three heads, count the steps whose arg-max equals the reference's fp16 arg-max: the fp16 engine must agree on at least as many steps as
    the bf16 engine, and on every step whose reference top-2 margin exceeds its budget."""
    import dataclasses
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=layers)
    M = np.load(os.path.join(golden_dir, name))
    if "tf_argmax_ref_fp16" not in M:
        pytest.skip("fixture generated on a host without CPU half kernels")
    forced, ref_ids = M["forced_ids"].tolist(), M["tf_argmax_ref_fp16"].tolist()
    ref_lg = torch.from_numpy(M["tf_logits_ref_fp16"])
    fin = torch.isfinite(ref_lg)
    srt = torch.sort(torch.where(fin, ref_lg, torch.full_like(ref_lg, -1e30)), dim=-1, descending=True).values
    margin = (srt[:, 0] - srt[:, 1]).tolist()
    n = len(forced) + 1
    agree = {}
    for dt in (F16, torch.bfloat16):
        eng = TraceEngine(cfg, max_batch=1, max_ctx=192, max_frames=4, max_new_tokens=64, dtype=dt)
        eng.load_weights(synth.iter_weights(cfg, dtype=F16))                 # the fp16 checkpoint; the bf16 engine casts it
        eng.encode_video(synth.synth_frames(cfg, 0).to(F16).to(dt), M["timestamps"].tolist())
        eng.prefill(0, eng.splice(M["input_ids"].tolist()))
        eng.decode_begin([0], [1], n, eos=-1, forced=[forced])
        eng.decode_steps(n - 1, use_graph=False)
        ids, _ = eng.decode_read()
        agree[dt] = sum(int(a == r) for a, r in zip(ids[0], ref_ids))
        if dt == F16:
            tol = 0.05 if layers == 1 else 0.10
            for i, (a, r, m) in enumerate(zip(ids[0], ref_ids, margin)):
                if m > 2 * tol:
                    assert a == r, (name, i, a, r, m)
        eng.close()
    msg = f"fp16 checkpoint, {layers} real-width layer(s), arg-max equal to the reference's own fp16 run: fp16 library {agree[F16]}/{n}, bf16 library (cast at load) {agree[torch.bfloat16]}/{n}"
    print(msg)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_measured.txt"), "a") as fh:
            fh.write(msg + "\n")
    assert agree[F16] >= agree[torch.bfloat16], msg
