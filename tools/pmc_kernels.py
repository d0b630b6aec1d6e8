"""Small fixed workload for rocprofv3 --pmc passes: the two roofline kernels of bench.py at the launch shapes it brackets — the ViT fc1 GEMM of one
170-frame tower call (gemm_w4_kernel<2, 0> since round 5; rounds 2-4: gemm_pers_kernel<2, ..>) and the decode attention of a 128-sequence step at ctx 2100 — plus
the ViT attention and the batch-64 gate|up GEMV of the earlier rounds; round 5: `tower` (the four GEMMs of a ViT layer + both attention kernels at 170 frames) and
`decgemm` (the wide decode step's weight-side GEMMs at 128 rows).   python tools/pmc_kernels.py [attn] [gemm] [attn_decode] [gemv] [tower] [decgemm]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
what = sys.argv[1:] or ["all"]
if "all" in what or "attn" in what:
    q, k, v = rnd(32, 577, 16, 64), rnd(32, 577, 16, 64), rnd(32, 577, 16, 64)
    for _ in range(3):
        ops.attention(q, k, v, False, 0.125)
if "all" in what or "gemm" in what:
    A, W, b = rnd(170 * 577, 1024), rnd(4096, 1024, scale=0.02), rnd(4096)      # the bench's probe shape: one 170-frame ViT call
    for _ in range(3):                                   # as shipped: the LayerNorm is a kernel of its own, the GEMM is gemm_w4_kernel<2, 0>
        ops.gemm(A, W, bias=b, epilogue=E.EPI_QUICKGELU)
if "tower" in what:
    # round 5: one launch set of a ViT layer at the bench's call shape (170 frames): the four GEMMs as the tower routes them (qkv and fc1 on the 4-wave persistent
    # kernel gemm_w4 — and once more on gemm_pers, for the counters side by side —, out-proj and fc2 with the residual on the loader-wave kernel) and the attention (the 192-row kernel and, for comparison, the 4 x 32-row one)
    M = 170 * 577
    X1, X4 = rnd(M, 1024), rnd(M, 4096)
    Wq, bq = rnd(3072, 1024, scale=0.02), rnd(3072)
    Wo, bo = rnd(1024, 1024, scale=0.02), rnd(1024)
    W1, b1 = rnd(4096, 1024, scale=0.02), rnd(4096)
    W2, b2 = rnd(1024, 4096, scale=0.02), rnd(1024)
    R = rnd(M, 1024)
    for _ in range(3):
        ops.gemm(X1, Wq, bias=bq)
        ops.gemm(X1, Wo, bias=bo, R=R, epilogue=E.EPI_RESIDUAL)
        ops.gemm(X1, W1, bias=b1, epilogue=E.EPI_QUICKGELU)
        ops.gemm(X4, W2, bias=b2, R=R, epilogue=E.EPI_RESIDUAL)
    del X4
    ops.set_gemm_variant(530)                            # the same two shapes on gemm_pers.hip
    for _ in range(3):
        ops.gemm(X1, Wq, bias=bq)
        ops.gemm(X1, W1, bias=b1, epilogue=E.EPI_QUICKGELU)
    ops.set_gemm_variant(531)
    q, k, v = rnd(170, 577, 16, 64), rnd(170, 577, 16, 64), rnd(170, 577, 16, 64)
    for var in (190, 192):
        ops.set_gemm_variant(var)
        for _ in range(3):
            ops.attention(q, k, v, False, 0.125)
    ops.set_gemm_variant(192)
if "decgemm" in what:
    # the wide decode step's weight-side kernels at 128 rows: the three split-K GEMMs (qkv, o, down; fp32 partial rows) and gate|up with the SwiGLU epilogue,
    # tile-packed weights, 4-stage ring (GemmArgs::w_tiled = 5), three rotating weight copies so that a launch streams from HBM
    X, XI = rnd(128, 4096), rnd(128, 14336)
    for N_, K_, x_ in ((6144, 4096, X), (4096, 4096, X), (4096, 14336, XI)):
        Ws = [ops.tile_pack(rnd(N_, K_, scale=0.02)) for _ in range(3)]
        for i in range(6):
            ops.gemm_partial(x_, Ws[i % 3], tiled=5)
        del Ws
    Ws = [ops.tile_pack(rnd(28672, 4096, scale=0.02)) for _ in range(3)]
    for i in range(6):
        ops.gemm_swiglu_tiled(X, Ws[i % 3], ring=True)
    del Ws
if "all" in what or "attn_decode" in what:      # the wide decode step's dominant kernel: 128 sequences x ctx 2100, 8 kv heads (1.10 GB of K + V^T rows)
    Bn, ctx, mc = 128, 2100, 2112
    kc, vt = rnd(Bn, 8, mc, 128, scale=0.5), rnd(Bn, 8, 128, mc, scale=0.5)
    q = rnd(Bn, 32 * 128)
    pos = torch.full((Bn,), ctx - 1, dtype=torch.int32, device=dev)
    for _ in range(4):
        ops.attn_decode(q, kc, None, pos, 1, 1 / math.sqrt(128), vtcache=vt)
if "all" in what or "gemv" in what:      # batches below 32: gate|up GEMV, tile-layout weights, fp32 partial rows (the round-2 probe, batch 64)
    Ws = [ops.tile_pack(rnd(28672, 4096, scale=0.02)) for _ in range(3)]
    X = rnd(64, 4096)
    for i in range(6):
        ops.skinny_gemm(X, Ws[i % 3], epilogue=E.EPI_PARTIAL, tiled=True, want_partial=False)
torch.cuda.synchronize()
