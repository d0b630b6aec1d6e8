"""The fp8 (OCP e4m3fn) weight path of the decoder — BASELINE config 5's "fp8 MFMA weight path".  The reference has no fp8 mode
(bitsandbytes loading only), so parity is anchored on this build's bf16 path and, through it, on the reference
fixtures: every kernel against an exact restatement of the W8A8 arithmetic in torch (float8_e4m3fn), and the engine end to end
against the reference's logits with a stated, wider budget.

Budget — stated BEFORE measuring, from the number format (round 3; rounds 1-2 used "measured + 25 %"):
  * e4m3 keeps 3 mantissa bits: a value m 2^e (1 <= m < 2) lands on a grid of spacing 2^(e-3), error uniform in +-2^(e-4), i.e. a
    relative rms error of 2^-4 / (sqrt(3) m) — between 3.6 % (m = 1) and 1.8 % (m -> 2), independent of any scale (floating
    point).  The budget takes the upper end: E8 = 3.6 % per element.
  * a W8A8 product of two quantised factors is off by sqrt(2) E8 = 5.1 % rms; K such terms with independent errors add to a dot
    product error of 5.1 % of the dot product's own rms — per projection, whatever K and the scale granularity are (EPS_PROJ).
    Weight-only (W8A16: e4m3 weights, bf16 activations) is E8 = 3.6 % per projection.
  * one decoder layer chains them: attention branch qkv -> o = sqrt(2) EPS_PROJ, MLP branch (gate x up) -> down = sqrt(3) EPS_PROJ;
    both add into the residual stream: EPS_LAYER = sqrt(5) EPS_PROJ = 11.4 % (W8A8) / 8.1 % (weight-only) of a branch's rms.
  * the logits are a linear read-out of the normalised stream, so their error is sigma_logit x (relative error of the stream); L layers
    of independent injections grow like sqrt(L), and on RANDOM weights nothing damps an injection on its way through the later layers
    (a trained network's residual stream is dominated by a few directions; a random one is a chaotic map), so every injection is taken
    at full size — no 1/(1 + 2l) dilution by the growing stream — times a stated safety factor AMP = 1.5 for the non-linear stages:
        rms budget(L) = sigma_logit x EPS_LAYER x sqrt(L) x AMP,          sigma_logit = 1.3 on these fixtures
        max budget(L) = 4.5 x rms budget(L)     (the largest of ~10^4 near-Gaussian deviations is ~3.9 sigma)
    -> W8A8: 0.22 / 1.0 after one layer, 0.63 / 2.8 after eight, 1.26 / 5.7 after 32.   Weight-only: x 0.71.
  What rounds 1-2 measured against these: one layer rms 0.14-0.16 / max 0.40-0.58, eight layers 0.39 / 1.4 — inside, with the margin
  the safety factor was meant to leave.  At 32 layers the budget is of the order of the logit spread itself: W8A8 on random weights
  is noise-limited there, which is what the 13-way arg-max flip rate recorded in profiles/ says in the units that matter.
The greedy id must equal the reference's wherever the reference's top-2 margin exceeds twice the maximum budget.  The kernels
themselves are exact: they match a torch float8_e4m3fn restatement of the same arithmetic to fp32 round-off (first three tests)."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from trace_amd import config as tcfg, synth  # noqa: E402
from trace_amd.engine import TraceEngine, ops, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU  # noqa: E402

E8 = 0.036                       # relative rms rounding error of e4m3 (3 mantissa bits)
SIGMA_LOGIT, AMP, MAX_OVER_RMS = 1.3, 1.5, 4.5


def fp8_budget(layers: int, weight_only: bool = False):
    """(max, rms) logit-error budget of the fp8 decoder path after `layers` layers — derived in the module docstring, not measured"""
    eps_proj = E8 * (1.0 if weight_only else 2.0 ** 0.5)
    rms = SIGMA_LOGIT * (5.0 ** 0.5) * eps_proj * (layers ** 0.5) * AMP
    return MAX_OVER_RMS * rms, rms


FP8_MAX_TOL = {n: fp8_budget(n)[0] for n in (1, 8, 32)}
FP8_RMS_TOL = {n: fp8_budget(n)[1] for n in (1, 8, 32)}
dev = torch.device("cuda", 0)


def ref_quant(x):
    """per-row dynamic e4m3 quantisation exactly as quant_rows_fp8 defines it"""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    inv = torch.where(amax > 0, 448.0 / amax, torch.ones_like(amax))
    q = (xf * inv[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, scale


@pytest.mark.parametrize("rows,K", [(5, 256), (64, 4096), (33, 14336), (300, 1024)])
def test_row_quantiser_bit_exact(rows, K):
    torch.manual_seed(rows + K)
    x = (torch.randn(rows, K, device=dev) * torch.rand(rows, 1, device=dev) * 3).to(torch.bfloat16)
    x[min(2, rows - 1)] = 0                                    # an all-zero row: scale 1, bytes 0
    q, sx = ops.quant_rows_fp8(x)
    rq, rs = ref_quant(x)
    torch.testing.assert_close(sx, rs, rtol=3e-7, atol=0)      # amax / 448: at most the last bit of the division
    # the codes: identical except a few 1e-4 of the elements, one code apart — x * (448 / amax) within an ulp of a rounding boundary
    # (the reciprocal's last bit), and e4m3's subnormal range (|v| < 2^-6), where v_cvt_pk_fp8_f32 and torch round differently
    diff = q != rq.view(torch.uint8)
    assert diff.float().mean().item() < 1e-3
    assert ((q.int() - rq.view(torch.uint8).int()).abs()[diff] <= 1).all()
    assert (q[min(2, rows - 1)] == 0).all() and float(sx[min(2, rows - 1)]) == 1.0


@pytest.mark.parametrize("M,N,K,epi", [(300, 256, 256, EPI_NONE), (1967, 512, 4096, EPI_RESIDUAL), (2100, 1024, 1024, EPI_SWIGLU),
                                       (3934, 6144, 4096, EPI_NONE), (64, 128, 14336, EPI_RESIDUAL), (1086, 28672, 4096, EPI_SWIGLU)])
def test_gemm_fp8_vs_exact_restatement(M, N, K, epi):
    torch.manual_seed(M + N)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    a8, sa = ops.quant_rows_fp8(a)
    w8, sw = ops.quant_rows_fp8(w)
    No = N // 2 if epi == EPI_SWIGLU else N
    R = torch.randn(M, No, device=dev).to(torch.bfloat16) if epi == EPI_RESIDUAL else None
    got = ops.gemm_fp8(a8, sa, w8, sw, R=R, epilogue=epi).float()
    acc = (a8.view(torch.float8_e4m3fn).float() @ w8.view(torch.float8_e4m3fn).float().t()) * sa[:, None] * sw[None, :]
    if epi == EPI_SWIGLU:                                      # 16-row interleaved gate|up layout of the engine's packed weight
        g = acc.view(M, N // 32, 2, 16)
        ref = (torch.nn.functional.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, No)
    elif epi == EPI_RESIDUAL:
        ref = acc.to(torch.bfloat16).float() + R.float()
    else:
        ref = acc
    err = (got - ref).abs()
    tol = 2e-2 * ref.abs().max().item() + 1e-3
    assert torch.isfinite(got).all() and err.max().item() < tol, (err.max().item(), tol)


@pytest.mark.parametrize("B", [1, 20, 40, 64])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (28672, 4096), (256, 512)])
def test_decode_gemv_fp8_vs_exact_restatement(B, N, K):
    torch.manual_seed(B + N)
    x = torch.randn(B, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    x8, sx = ops.quant_rows_fp8(x)
    w8, sw = ops.quant_rows_fp8(w)
    got = ops.skinny_fp8(x8, sx, w8, sw)
    ref = (x8.view(torch.float8_e4m3fn).float() @ w8.view(torch.float8_e4m3fn).float().t()) * sx[:, None] * sw[None, :]
    err = (got - ref).abs().max().item()
    assert err < 1e-3 * ref.abs().max().item() + 1e-4, err      # fp32 accumulation in another order


@pytest.mark.parametrize("B", [1, 20, 40, 64])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (28672, 4096), (256, 512)])
def test_decode_gemv_weight_only_vs_exact_restatement(B, N, K):
    """The weight-only (W8A16) decode GEMV: e4m3 weights widened to bf16 in registers (exact), bf16 activations, bf16 MFMA."""
    torch.manual_seed(B + N + 1)
    x = torch.randn(B, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    w8, sw = ops.quant_rows_fp8(w)
    got = ops.skinny_w8(x, w8, sw)
    ref = (x.float() @ w8.view(torch.float8_e4m3fn).float().t()) * sw[None, :]
    err = (got - ref).abs().max().item()
    assert err < 1e-3 * ref.abs().max().item() + 1e-4, err      # fp32 accumulation in another order


def _run(eng, frames, M, nb, paired):
    ts, ids, forced = M["timestamps"].tolist(), M["input_ids"].tolist(), M["forced_ids"].tolist()
    n = len(forced) + 1
    eng.encode_video(frames, ts)
    L, emb = eng.splice(ids, want_output=True)
    if paired:
        for b in range(0, nb, 2):
            eng.prefill_pair(b, emb, emb)
    else:
        for b in range(nb):
            eng.prefill(b, L, embeds=emb)
    lgs = [eng.decode_begin(list(range(nb)), [1] * nb, n, eos=-1, forced=[forced] * nb, want_logits=True).float().cpu()]
    for _ in range(n - 1):
        lgs.append(eng.decode_steps(1, use_graph=False, want_logits=True).float().cpu())
    out, _ = eng.decode_read()
    return lgs, out


def _check(lgs, out, M, rows, tag, layers=1):
    ref_lg, ref_ids = torch.from_numpy(M["tf_logits"]), M["tf_argmax"].tolist()
    fin = torch.isfinite(ref_lg)
    srt = torch.sort(torch.where(fin, ref_lg, torch.full_like(ref_lg, -1e30)), dim=-1, descending=True).values
    margin = (srt[:, 0] - srt[:, 1]).tolist()
    worst, rms = 0.0, 0.0
    for b in rows:
        lg = torch.stack([x[b] for x in lgs])
        assert torch.equal(torch.isfinite(lg), fin)
        d = lg[fin] - ref_lg[fin]
        worst, rms = max(worst, d.abs().max().item()), max(rms, d.pow(2).mean().sqrt().item())
        for i, (a, r, m) in enumerate(zip(out[b], ref_ids, margin)):
            if m > 2 * FP8_MAX_TOL[layers]:
                assert a == r, (tag, b, i, a, r, m)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_measured.txt"), "a") as f:
            f.write(f"{tag}: max |dlogit| = {worst:.4f} (budget {FP8_MAX_TOL[layers]}), rms = {rms:.4f} (budget {FP8_RMS_TOL[layers]}), "
                    f"logit std {ref_lg[fin].std().item():.3f}\n")
    assert worst < FP8_MAX_TOL[layers] and rms < FP8_RMS_TOL[layers], (tag, worst, rms)
    return worst


def test_fp8_engine_one_real_width_layer_vs_reference_fixture(golden_dir):
    """medium_llm.npz: one decoder layer at the real Mistral-7B widths; fp8 prefill GEMMs (M = 79: the 128x128 kernel) and fp8 decode
    GEMVs at batch 1 and 40 against the reference's logits."""
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=1)
    M = np.load(os.path.join(golden_dir, "medium_llm.npz"))
    eng = TraceEngine(cfg, max_batch=40, max_ctx=192, max_frames=4, max_new_tokens=64, llm_fp8=True)
    eng.load_weights(synth.state_dict(cfg).items())
    frames = synth.synth_frames(cfg, 0).to(torch.bfloat16)
    for nb in (1, 40):
        lgs, out = _run(eng, frames, M, nb, paired=False)
        _check(lgs, out, M, (0, nb - 1), f"fp8 one real-width layer, batch {nb}")
    eng.close()


def test_fp8_c5_256_frames_vs_reference_fixture(golden_dir):
    """BASELINE config 5 as specified: 256 frames -> prefill L = 3834 (paired: M = 7668 on the 256x256 loader-wave fp8 GEMM), 16 tokens,
    fp8 weight path — against the reference's own (fp32) logits for that shape (videomme_ctx.npz)."""
    cfg = dataclasses.replace(tcfg.tiny(num_frames=256), intermediate_size=14336, num_hidden_layers=1)
    M = np.load(os.path.join(golden_dir, "videomme_ctx.npz"))
    eng = TraceEngine(cfg, max_batch=4, max_ctx=3904, max_frames=256, max_new_tokens=16, llm_fp8=True)
    eng.load_weights(synth.state_dict(cfg).items())
    frames = synth.synth_frames(cfg, int(M["video_idx"])).to(torch.bfloat16)
    for nb, paired in ((1, False), (4, True)):
        lgs, out = _run(eng, frames, M, nb, paired)
        _check(lgs, out, M, (0, nb - 1), f"fp8 C5 (256 frames, L=3834), batch {nb}")
    eng.close()


def test_fp8_engine_tracks_bf16_engine_on_eight_layers(golden_dir):
    """Depth: eight real-width layers (deep_llm.npz): the fp8 engine's logits stay within the fp8 budget of the reference at batch 1,
    and the bf16 engine run beside it shows what the fp8 operands add."""
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=8)
    M = np.load(os.path.join(golden_dir, "deep_llm.npz"))
    frames = synth.synth_frames(cfg, 0).to(torch.bfloat16)
    res = {}
    for f8 in (True, False):
        eng = TraceEngine(cfg, max_batch=2, max_ctx=192, max_frames=4, max_new_tokens=64, llm_fp8=f8)
        eng.load_weights(synth.iter_weights(cfg))
        lgs, out = _run(eng, frames, M, 2, paired=False)
        res[f8] = torch.stack([x[0] for x in lgs])
        if f8:
            _check(lgs, out, M, (0, 1), "fp8 eight real-width layers", layers=8)
        eng.close()
    fin = torch.isfinite(res[False])
    d = res[True][fin] - res[False][fin]
    assert d.abs().max().item() < FP8_MAX_TOL[8] and d.pow(2).mean().sqrt().item() < FP8_RMS_TOL[8]


def _flip_stats(lgs, M, row=0):
    """13-way (time / score head) teacher-forced steps: how many engine arg-maxes differ from the reference's, over all such steps and over
    those where the reference's own top-2 margin exceeds 0.5 (a decision a trained model would call clear)"""
    ref_lg = torch.from_numpy(M["tf_logits"])
    fin = torch.isfinite(ref_lg)
    lg = torch.stack([x[row] for x in lgs])
    steps13 = [i for i in range(ref_lg.shape[0]) if int(fin[i].sum()) <= 13]
    neg = torch.full_like(ref_lg, -1e30)
    ra, ea = torch.where(fin, ref_lg, neg).argmax(-1), torch.where(fin, lg, neg).argmax(-1)
    srt = torch.sort(torch.where(fin, ref_lg, neg), dim=-1, descending=True).values
    margin = srt[:, 0] - srt[:, 1]
    flips = sum(int(ra[i] != ea[i]) for i in steps13)
    clear = [i for i in steps13 if float(margin[i]) > 0.5]
    flips_clear = sum(int(ra[i] != ea[i]) for i in clear)
    d = (lg[fin] - ref_lg[fin])
    return dict(steps=len(steps13), flips=flips, clear=len(clear), flips_clear=flips_clear, max=d.abs().max().item(), rms=d.pow(2).mean().sqrt().item())


@pytest.mark.parametrize("fixture,layers", [("deep_llm.npz", 8), ("full_depth_llm.npz", 32)])
def test_fp8_schemes_vs_reference_fixture(golden_dir, fixture, layers):
    """Both fp8 schemes (1: W8A8 everywhere; 2: W8A8 prefill + weight-only decode GEMVs) and the bf16 engine against the reference's
    teacher-forced logits at 8 and at all 32 real-width layers: logit error inside the a-priori budget (fp8_budget), and the number of
    13-way arg-max decisions that flip — written to parity_measured.txt; the scheme C5 runs on is chosen from these numbers."""
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=layers)
    M = np.load(os.path.join(golden_dir, fixture))
    frames = synth.synth_frames(cfg, 0).to(torch.bfloat16)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    res = {}
    for scheme in ((False, "w8a8", "weight_only") if layers <= 8 else ("w8a8", "weight_only")):     # (bf16 at 32 layers: tests/test_gpu_configs.py)
        eng = TraceEngine(cfg, max_batch=2, max_ctx=192, max_frames=4, max_new_tokens=64, llm_fp8=scheme)
        eng.load_weights(synth.iter_weights(cfg))
        lgs, _ = _run(eng, frames, M, 1, paired=False)
        eng.close()
        st = _flip_stats(lgs, M)
        res[scheme] = st
        line = (f"fp8 schemes, {layers} real-width layers, scheme {scheme or 'bf16'}: max |dlogit| = {st['max']:.4f}, rms = {st['rms']:.4f}; 13-way steps "
                f"{st['steps']}: arg-max flips vs reference {st['flips']} ({st['flips_clear']} of the {st['clear']} with reference margin > 0.5)")
        print(line)
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "parity_measured.txt"), "a") as f:
                f.write(line + "\n")
    mx, rm = fp8_budget(layers)
    for scheme in ("w8a8", "weight_only"):      # (scheme 2's prefill is W8A8 too, so the W8A8 budget is the one that applies to both)
        assert res[scheme]["max"] < mx and res[scheme]["rms"] < rm, (scheme, res[scheme])
    assert res["weight_only"]["rms"] <= res["w8a8"]["rms"] * 1.1          # one rounded operand instead of two where the tokens are chosen
