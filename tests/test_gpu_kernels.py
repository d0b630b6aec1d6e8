"""Kernel-level parity on the GPU: every HIP kernel against a plain PyTorch fp32 reference of the same op
(bf16 inputs, fp32 math), called through the C ABI (trace_op_* entry points).  Tolerances are stated per test:
bf16 outputs carry 2^-8 relative rounding, accumulation is fp32."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from trace_amd import engine as E  # noqa: E402
from trace_amd.engine import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def check(name, got, ref, atol, rtol):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        idx = torch.nonzero(bad)
        rows = torch.unique(idx[:, 0])[:16].tolist() if idx.numel() else []
        cols = torch.unique(idx[:, -1])[:16].tolist() if idx.numel() else []
        first = idx[0].tolist() if idx.numel() else None
        msg = (f"{name}: {int(bad.sum())}/{bad.numel()} elements off; max err {err.max().item():.4g} "
               f"(ref max {ref.abs().max().item():.4g}); first bad {first} got "
               f"{got[tuple(first)].item() if first else None} ref {ref[tuple(first)].item() if first else None}; "
               f"bad rows {rows} cols {cols}; finite={torch.isfinite(got).all().item()}")
        pytest.fail(msg)


@pytest.fixture(params=[0, 2, 3, 4, 5, 8], ids=["auto", "tile128", "tile256", "ldr", "persistent", "w4"])
def gemm_variant(request):
    ops.set_gemm_variant(request.param)
    yield request.param
    ops.set_gemm_variant(0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 256, 128), (577, 384, 1024), (1, 128, 64), (1000, 1024, 640),
                                   (2100, 512, 192), (9000, 2048, 256), (20000, 1024, 64), (70000, 1024, 128)])
def test_gemm_plain_bias(M, N, K, gemm_variant):
    A, W, b = rnd(M, K), rnd(N, K, scale=0.05), rnd(N, scale=0.5)
    ref = A.float() @ W.float().t() + b.float()
    check("gemm+bias", ops.gemm(A, W, bias=b), ref, 2e-2, 1e-2)
    check("gemm", ops.gemm(A, W), A.float() @ W.float().t(), 2e-2, 1e-2)


def test_gemm_asymmetric_layout(gemm_variant):
    # transpose-detecting: A rows and W rows carry different, non-symmetric patterns
    M, N, K = 256, 256, 128
    A = torch.zeros(M, K)
    W = torch.zeros(N, K)
    for i in range(M):
        A[i, i % K] = 1.0 + (i % 7)
    for j in range(N):
        W[j, (3 * j) % K] = 0.5 + (j % 5)
    A, W = A.to(torch.bfloat16).to(DEV), W.to(torch.bfloat16).to(DEV)
    check("gemm layout", ops.gemm(A, W), A.float() @ W.float().t(), 1e-3, 1e-3)


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (64, 128, 4096), (1300, 512, 256)])
def test_gemm_epilogues(M, N, K, gemm_variant):
    A, W, b, R = rnd(M, K), rnd(N, K, scale=0.05), rnd(N, scale=0.5), rnd(M, N)
    lin = A.float() @ W.float().t() + b.float()
    check("residual", ops.gemm(A, W, bias=b, R=R, epilogue=E.EPI_RESIDUAL), lin.to(torch.bfloat16).float() + R.float(), 3e-2, 1e-2)
    check("quickgelu", ops.gemm(A, W, bias=b, epilogue=E.EPI_QUICKGELU), lin * torch.sigmoid(1.702 * lin), 2e-2, 1e-2)
    # SwiGLU on 16-row interleaved [gate|up] weights
    Wg, Wu = rnd(N // 2, K, scale=0.05, seed=1), rnd(N // 2, K, scale=0.05, seed=2)
    Wp = torch.stack([Wg.view(-1, 16, K), Wu.view(-1, 16, K)], dim=1).reshape(N, K).contiguous()
    g, u = A.float() @ Wg.float().t(), A.float() @ Wu.float().t()
    check("swiglu", ops.gemm(A, Wp, epilogue=E.EPI_SWIGLU), torch.nn.functional.silu(g) * u, 2e-2, 1e-2)


def test_gemm_residual_in_place():
    M, N, K = 130, 128, 64
    A, W, R = rnd(M, K), rnd(N, K, scale=0.1), rnd(M, N)
    ref = (A.float() @ W.float().t()).to(torch.bfloat16).float() + R.float()
    lib = E._lib.load()
    Rc = R.clone()
    E._lib.check(lib.trace_op_gemm(E._ptr(A), K, E._ptr(W), K, E._ptr(Rc), N, None, E._ptr(Rc), N, M, N, K, E.EPI_RESIDUAL, E._stream()))
    check("residual in place", Rc, ref, 3e-2, 1e-2)


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (9000, 2048, 256), (70000, 1024, 128), (66000, 512, 192)])
def test_gemm_persistent_equals_loader_wave_kernel(M, N, K):
    """gemm_pers.hip (variant 5 ticketed, 6 static deal) against gemm_ldr.hip (variant 4), bit for bit, all four epilogues: ragged M, fewer
    tiles than CUs, more than one tile per workgroup (the ticket path), the in-place residual, and back-to-back launches (the ticket counters
    re-arm themselves at the end of every launch)."""
    A, W, b, R = rnd(M, K), rnd(N, K, scale=0.05), rnd(N, scale=0.5), rnd(M, N)
    lib = E._lib.load()
    try:
        for epi, kw in ((E.EPI_NONE, dict(bias=b)), (E.EPI_QUICKGELU, dict(bias=b)), (E.EPI_RESIDUAL, dict(bias=b, R=R)), (E.EPI_SWIGLU, {})):
            ops.set_gemm_variant(4)
            ref = ops.gemm(A, W, epilogue=epi, **kw)
            for v in (5, 6):
                ops.set_gemm_variant(v)
                for rep in range(3):
                    got = ops.gemm(A, W, epilogue=epi, **kw)
                    assert torch.equal(got, ref), (epi, v, rep, (got.float() - ref.float()).abs().max().item())
        ops.set_gemm_variant(4)
        ref = ops.gemm(A, W, R=R, epilogue=E.EPI_RESIDUAL)
        ops.set_gemm_variant(5)
        Rc = R.clone()
        E._lib.check(lib.trace_op_gemm(E._ptr(A), K, E._ptr(W), K, E._ptr(Rc), N, None, E._ptr(Rc), N, M, N, K, E.EPI_RESIDUAL, E._stream()))
        assert torch.equal(Rc, ref)
    finally:
        ops.set_gemm_variant(0)


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (9000, 2048, 256), (70000, 1024, 192), (66000, 512, 320), (3934, 6144, 1024)])
def test_gemm_four_wave_kernel_equals_loader_wave_kernel(M, N, K):
    """gemm_w4.hip (variant 8: the persistent 256x256 tile on 4 waves of 128x128, accumulators in AGPRs, the waves issue their own LDS-DMA pieces) against
    gemm_ldr.hip (variant 4), bit for bit, all four epilogues: ragged M, fewer tiles than CUs, several tiles per workgroup (tickets and the static deal),
    its A/B builds (one barrier per K-tile; L2 touches; the check build in which every hand-counted s_waitcnt vmcnt(N) is vmcnt(0): the counted build must give the same bits), the in-place residual, back-to-back launches, and launches alternating with gemm_pers.hip on one
    stream (the two kernels share the stream's ticket counters)."""
    A, W, b, R = rnd(M, K), rnd(N, K, scale=0.05), rnd(N, scale=0.5), rnd(M, N)
    lib = E._lib.load()
    try:
        for epi, kw in ((E.EPI_NONE, dict(bias=b)), (E.EPI_NONE, {}), (E.EPI_QUICKGELU, dict(bias=b)), (E.EPI_RESIDUAL, dict(bias=b, R=R)), (E.EPI_SWIGLU, {})):
            ops.set_gemm_variant(4)
            ref = ops.gemm(A, W, epilogue=epi, **kw)
            for walk, opt in ((500, 0), (501, 0), (500, 1), (500, 4), (500, 5), (500, 2)):      # (opt 2: the check build whose counted waits are all vmcnt(0))
                ops.set_gemm_variant(walk)
                ops.set_gemm_variant(540 + opt)
                for rep in range(3):
                    ops.set_gemm_variant(8)
                    got = ops.gemm(A, W, epilogue=epi, **kw)
                    assert torch.equal(got, ref), (epi, walk, opt, rep, (got.float() - ref.float()).abs().max().item())
                    if rep == 1:
                        ops.set_gemm_variant(5)
                        assert torch.equal(ops.gemm(A, W, epilogue=epi, **kw), ref)
            ops.set_gemm_variant(500)
            ops.set_gemm_variant(540)
        ops.set_gemm_variant(4)
        ref = ops.gemm(A, W, R=R, epilogue=E.EPI_RESIDUAL)
        ops.set_gemm_variant(8)
        Rc = R.clone()
        E._lib.check(lib.trace_op_gemm(E._ptr(A), K, E._ptr(W), K, E._ptr(Rc), N, None, E._ptr(Rc), N, M, N, K, E.EPI_RESIDUAL, E._stream()))
        assert torch.equal(Rc, ref)
    finally:
        ops.set_gemm_variant(500)
        ops.set_gemm_variant(540)
        ops.set_gemm_variant(0)


def test_gemm_auto_routing_takes_the_four_wave_kernel_and_can_be_switched_back():
    """auto mode (what the engine runs): shapes without a residual go to gemm_w4.hip, trace_op_set_gemm_variant(530) sends them back to gemm_pers.hip —
    same bits either way, and equal to the forced kernels"""
    M, N, K = 20000, 1024, 512
    A, W, b = rnd(M, K), rnd(N, K, scale=0.05), rnd(N, scale=0.5)
    try:
        ops.set_gemm_variant(4)
        ref = ops.gemm(A, W, bias=b, epilogue=E.EPI_QUICKGELU)
        ops.set_gemm_variant(0)
        for sw in (531, 530, 531):
            ops.set_gemm_variant(sw)
            assert torch.equal(ops.gemm(A, W, bias=b, epilogue=E.EPI_QUICKGELU), ref), sw
    finally:
        ops.set_gemm_variant(531)
        ops.set_gemm_variant(0)


def test_gemm_persistent_two_streams():
    """Two persistent launches in flight on different streams: each stream has its own ticket counters (both persistent kernels)."""
    M, N, K = 40000, 1024, 256
    A1, W1, A2, W2 = rnd(M, K), rnd(N, K, scale=0.05), rnd(M, K, seed=3), rnd(N, K, scale=0.05, seed=4)
    try:
        ops.set_gemm_variant(4)
        r1, r2 = ops.gemm(A1, W1), ops.gemm(A2, W2)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for v in (5, 8):                               # gemm_pers.hip, gemm_w4.hip
            ops.set_gemm_variant(v)
            torch.cuda.synchronize()
            outs = []
            for _ in range(4):
                with torch.cuda.stream(s1):
                    o1 = ops.gemm(A1, W1)
                with torch.cuda.stream(s2):
                    o2 = ops.gemm(A2, W2)
                outs.append((o1, o2))
            torch.cuda.synchronize()
            for o1, o2 in outs:
                assert torch.equal(o1, r1) and torch.equal(o2, r2), v
    finally:
        ops.set_gemm_variant(0)


@pytest.mark.parametrize("rows,D", [(5, 128), (577, 1024), (33, 4096), (4618, 1024)])
def test_norms(rows, D):
    x, w, b = rnd(rows, D, scale=2.0), 1 + rnd(D, scale=0.1), rnd(D, scale=0.1)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5)
    check("layernorm", ops.layernorm(x, w, b, 1e-5), ref, 2e-2, 1e-2)
    xf = x.float()
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()
    check("rmsnorm", ops.rmsnorm(x, w, 1e-5), ref, 2e-2, 1e-2)


def _attn_ref(q, k, v, causal, scale):
    Bn, nq, heads, hd = q.shape
    nkv, kvh = k.shape[1], k.shape[2]
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(heads // kvh, dim=1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(heads // kvh, dim=1)
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        off = nkv - nq
        mask = torch.arange(nkv, device=q.device)[None, :] > (torch.arange(nq, device=q.device)[:, None] + off)
        s = s.masked_fill(mask, float("-inf"))
    return (torch.softmax(s, dim=-1) @ vf).permute(0, 2, 1, 3)


@pytest.mark.parametrize("Bn,n,heads", [(2, 17, 2), (3, 577, 16), (1, 64, 1), (1, 130, 4)])
def test_attention_vit(Bn, n, heads):
    q, k, v = rnd(Bn, n, heads, 64, seed=1), rnd(Bn, n, heads, 64, seed=2), rnd(Bn, n, heads, 64, seed=3)
    check("attn vit", ops.attention(q, k, v, False, 0.125), _attn_ref(q, k, v, False, 0.125), 2e-2, 2e-2)


def test_attention_vit_rowmajor_v_equals_transposed_v():
    """Round 3: the LDS-DMA ViT attention reads V row-major (ds_read_b64_tr_b16) instead of a transposed, permuted copy: the same values
    in the same MFMA k-slots, so the two paths agree bit for bit (trace_op_set_gemm_variant(116) = the transposed-V path)."""
    q, k, v = rnd(3, 577, 16, 64, seed=4), rnd(3, 577, 16, 64, seed=5), rnd(3, 577, 16, 64, seed=6)
    try:
        ops.set_gemm_variant(190)                  # the 4 x 32-row kernel (577 tokens go to the 192-row kernel since round 5)
        a = ops.attention(q, k, v, False, 0.125)
        ops.set_gemm_variant(116)
        b = ops.attention(q, k, v, False, 0.125)
    finally:
        ops.set_gemm_variant(110)
        ops.set_gemm_variant(192)
    assert torch.equal(a, b)
    check("attn vit 577 keys", a, _attn_ref(q, k, v, False, 0.125), 2e-2, 2e-2)


@pytest.mark.parametrize("Bn,n,heads", [(3, 577, 16), (2, 193, 3), (1, 384, 5), (1, 769, 2), (5, 196, 1), (7, 577, 1)])
@pytest.mark.parametrize("ring", [191, 192])
def test_attention_vit_192_row_kernel(Bn, n, heads, ring):
    """Round 5: the 192-row ViT attention (16x16x32 MFMA, three query tiles per wave, row sums on the matrix pipe, the rows 192 does not divide on
    a VALU path) against torch and against the 4 x 32-row kernel: whole workgroups only (384), one leftover row (193, 577, 769), four (196), no
    key tail (384) and a key tail (the others), head counts that leave the last quad of (frame, head) pairs partly empty; both ring depths."""
    q, k, v = rnd(Bn, n, heads, 64, seed=11), rnd(Bn, n, heads, 64, seed=12), rnd(Bn, n, heads, 64, seed=13)
    ref = _attn_ref(q, k, v, False, 0.125)
    try:
        ops.set_gemm_variant(ring)
        got = ops.attention(q, k, v, False, 0.125)
        again = ops.attention(q, k, v, False, 0.125)
        ops.set_gemm_variant(190)
        old = ops.attention(q, k, v, False, 0.125)
    finally:
        ops.set_gemm_variant(192)
    assert torch.equal(got, again)
    check("attn vit 192-row", got, ref, 2e-2, 2e-2)
    check("attn vit 192-row vs 32-row kernel", got, old, 2e-2, 2e-2)


def test_attention_vit_192_row_kernel_moves_its_reference():
    """Keys that beat everything before them by more than the lazy-rescale threshold (2^8), in the first tile, in the middle, in the last tile and as
    the tail key, for rows of different waves and query tiles — and a row whose largest score is the very first key: the slow path must rescale
    exactly the rows that need it, once.  Tolerance: the kernels fold scale * log2(e) into a bf16 copy of q, so a score s carries an error of
    ~2^-9 |s| (exp2 domain); with self-matches of ~25 planted here two near-tied large scores can shift each other's weight by a few per cent (a CPU
    replay of the kernel's arithmetic reproduces the GPU's value to the last digit) — 5e-2 absolute, against 2e-2 on ordinary data."""
    Bn, n, heads = 2, 577, 2
    q, k, v = rnd(Bn, n, heads, 64, seed=21), rnd(Bn, n, heads, 64, seed=22), rnd(Bn, n, heads, 64, seed=23)
    for key, row, gain in ((3, 5, 2.0), (70, 50, 2.5), (150, 200, 2.0), (300, 17, 3.0), (500, 383, 2.0), (575, 100, 2.5), (576, 20, 3.0), (64, 191, 2.5)):
        k[:, key] = q[:, row] * gain
    k[:, 0] = q[:, 300] * 3.0                       # row 300: its largest score is the very first key
    outs = {}
    try:
        for var in (191, 192, 190):
            ops.set_gemm_variant(var)
            outs[var] = ops.attention(q, k, v, False, 0.125)
    finally:
        ops.set_gemm_variant(192)
    ref = _attn_ref(q, k, v, False, 0.125)
    for var in (191, 192, 190):
        check(f"attn vit spikes, variant {var}", outs[var], ref, 5e-2, 2e-2)
    assert torch.equal(outs[191], outs[192])        # the ring depth changes no arithmetic


def test_attention_vit_spiked_scores():
    # forces large running-max jumps between kv tiles (online-softmax rescale path)
    Bn, n, heads = 1, 200, 2
    q, k, v = rnd(Bn, n, heads, 64, seed=1), rnd(Bn, n, heads, 64, seed=2), rnd(Bn, n, heads, 64, seed=3)
    k[:, 150] = q[:, 10] * 4
    k[:, 70] = q[:, 11] * 6
    check("attn spike", ops.attention(q, k, v, False, 0.125), _attn_ref(q, k, v, False, 0.125), 3e-2, 2e-2)


@pytest.mark.parametrize("L,kvh", [(79, 8), (200, 2), (1, 1), (64, 1), (333, 8)])
def test_attention_prefill_causal_gqa(L, kvh):
    q = rnd(1, L, 4 * kvh, 128, seed=1)
    k, v = rnd(1, L, kvh, 128, seed=2), rnd(1, L, kvh, 128, seed=3)
    sc = 1 / math.sqrt(128)
    check("attn prefill", ops.attention(q, k, v, True, sc), _attn_ref(q, k, v, True, sc), 2e-2, 2e-2)


@pytest.mark.parametrize("Bn,N,K", [(1, 128, 256), (1, 4096, 4096), (3, 6144, 4096), (16, 256, 14336), (2, 512, 128),
                                    (17, 256, 4096), (32, 512, 14336), (25, 6144, 4096), (20, 16384, 256), (40, 512, 4096), (64, 4096, 14336),
                                    (50, 16384, 512)])
def test_skinny_gemm(Bn, N, K):
    X, W, R = rnd(Bn, K), rnd(N, K, scale=0.05), rnd(Bn, N)
    lin = X.float() @ W.float().t()
    check("skinny", ops.skinny_gemm(X, W), lin, 3e-2, 1e-2)
    check("skinny+res", ops.skinny_gemm(X, W, R=R, epilogue=E.EPI_RESIDUAL), lin.to(torch.bfloat16).float() + R.float(), 4e-2, 1e-2)
    if N % 32 == 0:
        Wg, Wu = W[: N // 2], W[N // 2:]
        Wp = torch.stack([Wg.reshape(-1, 16, K), Wu.reshape(-1, 16, K)], dim=1).reshape(N, K).contiguous()
        g, u = X.float() @ Wg.float().t(), X.float() @ Wu.float().t()
        check("skinny swiglu", ops.skinny_gemm(X, Wp, epilogue=E.EPI_SWIGLU), torch.nn.functional.silu(g) * u, 3e-2, 1e-2)
        if K % 64 == 0:
            sw = ops.skinny_gemm(X, ops.tile_pack(Wp), epilogue=E.EPI_SWIGLU, tiled=True)
            assert torch.equal(sw, ops.skinny_gemm(X, Wp, epilogue=E.EPI_SWIGLU)), "tile layout changes the result"
            pg = ops.skinny_gemm(X, ops.tile_pack(Wp), epilogue=E.EPI_PARTIAL, tiled=True)     # the engine's path: partial rows + combine
            check("skinny partial + swiglu combine", ops.swiglu_combine(pg, Bn), torch.nn.functional.silu(g) * u, 3e-2, 1e-2)
    # the decode weight layout is a pure permutation: bit-identical results
    Wt = ops.tile_pack(W)
    assert torch.equal(ops.skinny_gemm(X, Wt, tiled=True), ops.skinny_gemm(X, W))
    # fp32 k-chunk partial rows + the fused residual add / RMSNorm consumer (the o-proj / down-proj path of a decode step)
    if N <= 4096:
        part = ops.skinny_gemm(X, Wt, epilogue=E.EPI_PARTIAL, tiled=True)
        tot = part.sum(0)[:Bn]
        check("skinny partial", tot, lin, 2e-3, 1e-3)
        w = rnd(N, seed=9)
        x, y = ops.add_rmsnorm(part, R, w, 1e-5)
        xr = (tot.to(torch.bfloat16).float() + R.float()).to(torch.bfloat16)
        assert (x.float() - xr.float()).abs().max().item() <= 2 ** -6 * xr.float().abs().max().item()
        xf = x.float()
        yr = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-5) * w.float()
        check("add_rmsnorm", y, yr, 2e-2, 1e-2)


@pytest.mark.parametrize("M,N,K", [(128, 6144, 4096), (100, 4096, 4096), (65, 4096, 14336), (128, 4096, 256), (77, 512, 4096), (256, 4096, 4096), (200, 6144, 4096)])
def test_gemm_partial_rows(M, N, K):
    """Decode batches above 64 rows: the split-K MFMA GEMM leaves fp32 k-chunk partial rows [ks][sk_rows][N]; their sum is the product,
    and the decode consumer (sum + residual + RMSNorm) takes them exactly as it takes the GEMV's."""
    X, W = rnd(M, K, seed=3), rnd(N, K, scale=0.05, seed=4)
    part = ops.gemm_partial(X, W)
    lin = X.float() @ W.float().t()
    SKR = part.shape[1]
    assert SKR >= M and part.shape[0] >= 1
    check("partial gemm", part[:, :M].sum(0), lin, 3e-2, 1e-2)
    if K % 64 == 0:      # weights from the decode tile copy: each MFMA then sums another 32 of a K-tile's 64 k (fp32 order differs: tolerance, not bits);
        Wt = ops.tile_pack(W)                               # the 4-stage ring changes nothing
        pt = ops.gemm_partial(X, Wt, tiled=1)
        check("partial gemm, tiled weights", pt[:, :M].sum(0), lin, 3e-2, 1e-2)
        assert torch.equal(ops.gemm_partial(X, Wt, tiled=5), pt)    # 4-stage K-tile ring: the same sums in the same order
        if N % 256 == 0:
            sw = ops.gemm_swiglu_tiled(X, Wt)              # rows read as 16-row interleaved gate|up
            check("tiled swiglu gemm", sw, ops.gemm(X, W, epilogue=E.EPI_SWIGLU), 3e-2, 1e-2)
    assert float(part[:, M:].abs().max()) == 0.0 if M < SKR else True          # rows beyond M are never written
    if N <= 4096:
        R, w = rnd(M, N, seed=5), rnd(N, seed=6)
        x, y = ops.add_rmsnorm(part, R, w, 1e-5)
        xr = part[:, :M].sum(0).to(torch.bfloat16).float() + R.float()
        check("partial gemm + add", x, xr, 3e-2, 1e-2)
        xb = x.float()
        check("partial gemm + rmsnorm", y, xb * torch.rsqrt((xb * xb).mean(-1, keepdim=True) + 1e-5) * w.float(), 4e-2, 2e-2)


@pytest.mark.parametrize("Bn,N,K,ks_in", [(1, 6144, 4096, 0), (1, 28672, 4096, 8), (3, 6144, 4096, 14), (4, 512, 4096, 2), (2, 256, 1024, 3)])
def test_skinny_gemv_with_fused_add_rmsnorm(Bn, N, K, ks_in):
    """Batches of 1..4: the decode GEMV sums the previous GEMV's partial rows, adds the residual, RMS-normalises and multiplies in one launch;
    against the unfused pair (add_rmsnorm kernel + GEMV): the new residual rows bit for bit, the product within fp32 re-ordering of one sum."""
    R, w, W = rnd(Bn, K, seed=1), (1 + 0.2 * rnd(K, seed=2).float()).to(torch.bfloat16), rnd(N, K, scale=0.05, seed=3)
    part = None
    if ks_in:
        part = torch.zeros((ks_in, ops.skinny_ks.__self__ if False else E._lib.load().trace_op_sk_rows(), K), dtype=torch.float32, device=DEV)
        part[:, :Bn] = torch.randn(ks_in, Bn, K, device=DEV) * 0.3
    xout, out = ops.skinny_fused_norm(part, R, w, 1e-5, W)
    if ks_in:
        x_ref, y_ref = ops.add_rmsnorm(part, R, w, 1e-5)
    else:
        x_ref, y_ref = R, ops.rmsnorm(R, w, 1e-5)
    assert torch.equal(xout, x_ref)
    ref = ops.skinny_gemm(y_ref, ops.tile_pack(W), epilogue=E.EPI_PARTIAL, tiled=True)
    # the normalised activations can differ by one bf16 ulp where the two sums of squares round differently: compare the products with that slack
    check("fused norm gemv", out[:, :Bn].sum(0), ref[:, :Bn].sum(0), 2e-2, 1e-2)


@pytest.mark.parametrize("Bn,ctxs", [(1, [80]), (3, [1, 200, 2047]), (2, [16, 17])])
def test_attn_decode(Bn, ctxs):
    nq, nkv, max_ctx = 32, 8, 2048
    q = rnd(Bn, nq * 128, seed=4)
    kc, vc = rnd(Bn, nkv, max_ctx, 128, seed=5), rnd(Bn, nkv, max_ctx, 128, seed=6)
    pos = torch.tensor([c - 1 for c in ctxs], dtype=torch.int32, device=DEV)
    sc = 1 / math.sqrt(128)
    out = ops.attn_decode(q, kc, vc, pos, 16, sc)
    out2 = ops.attn_decode(q, kc, vc, pos, 32, sc)      # second launch re-uses the ticket counters
    assert torch.equal(out, out2) or (out.float() - out2.float()).abs().max() < 2e-2
    for ns in (1, 2, 5):                                 # 1 = the no-merge shortcut; 5 = ragged chunks
        o3 = ops.attn_decode(q, kc, vc, pos, ns, sc)
        assert (out.float() - o3.float()).abs().max() < 2e-2, ns
    for b, ctx in enumerate(ctxs):
        qq = q[b].view(1, 1, nq, 128)
        kk = kc[b, :, :ctx].permute(1, 0, 2).unsqueeze(0)
        vv = vc[b, :, :ctx].permute(1, 0, 2).unsqueeze(0)
        ref = _attn_ref(qq, kk, vv, False, sc).reshape(-1)
        check(f"attn decode b{b}", out[b], ref, 2e-2, 2e-2)
