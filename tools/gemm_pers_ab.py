"""Interleaved A/B of the 256x256 GEMM kernels on the ViT / prefill shapes: gemm_ldr.hip (variant 4), gemm_pers.hip with ticketed tiles
(5) and with a static deal (6), optionally with K-loop build options (trace_op_set_gemm_variant(300 + opt)).  The variants are timed
round-robin (the part's clock drifts over a run: back-to-back blocks of one variant are not comparable), median of the rounds.
python tools/gemm_pers_ab.py [opt ...]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def timed(fn, n=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
opts = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0]
variants = [("ldr", 4, 0), ("ldr/noA", 4, -2), ("ldr/none", 4, -3)] + [("pers%s" % ("" if o == 0 else "/opt%d" % o), 5, o) for o in opts] + [("static", 6, 0), ("1tile/wg", 7, 0)]
shapes = [("vit fc1 gelu", 170 * 577, 4096, 1024, E.EPI_QUICKGELU, True), ("vit qkv", 170 * 577, 3072, 1024, E.EPI_NONE, True),
          ("vit fc2 res", 170 * 577, 1024, 4096, E.EPI_RESIDUAL, True), ("vit out res", 170 * 577, 1024, 1024, E.EPI_RESIDUAL, True),
          ("prefill qkv pair", 3934, 6144, 4096, E.EPI_NONE, False), ("prefill o res", 3934, 4096, 4096, E.EPI_RESIDUAL, False),
          ("prefill gateup pair", 3934, 28672, 4096, E.EPI_SWIGLU, False), ("prefill down res", 3934, 4096, 14336, E.EPI_RESIDUAL, False),
          ("prefill qkv one", 1967, 6144, 4096, E.EPI_NONE, False), ("prefill gateup one", 1967, 28672, 4096, E.EPI_SWIGLU, False),
          ("vit out none", 170 * 577, 1024, 1024, E.EPI_NONE, True), ("vit fc2 none", 170 * 577, 1024, 4096, E.EPI_NONE, True),
          ("down shape none", 3934, 4096, 14336, E.EPI_NONE, False), ("o shape none", 3934, 4096, 4096, E.EPI_NONE, False),
          ("c5 qkv pair", 7668, 6144, 4096, E.EPI_NONE, False), ("c5 gateup pair", 7668, 28672, 4096, E.EPI_SWIGLU, False)]
only = [a for a in sys.argv[1:] if not a.isdigit()]
for name, M, N, K, epi, has_bias in shapes:
    if only and not any(o in name for o in only): continue
    A, W = rnd(M, K), rnd(N, K, scale=0.03)
    bias = rnd(N) if has_bias else None
    R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
    run = lambda: ops.gemm(A, W, bias=bias, R=R, epilogue=epi)
    ts = {v[0]: [] for v in variants}
    ref, same = None, {}
    for rnd_i in range(7):
        for vn, v, o in variants:
            ops.set_gemm_variant(400 + (-o if o < 0 else 0))      # gemm_ldr A/B: residual touches off
            ops.set_gemm_variant(300 + max(o, 0))
            ops.set_gemm_variant(v)
            if rnd_i == 0:
                out = run()
                if ref is None: ref = out
                same[vn] = torch.equal(out, ref) or "max|d| %.3g (%d of %d differ)" % ((out.float() - ref.float()).abs().max().item(), int((out != ref).sum()), out.numel())
            ts[vn].append(timed(run))
    ops.set_gemm_variant(400)
    ops.set_gemm_variant(300)
    ops.set_gemm_variant(0)
    med = {k: statistics.median(v[1:]) for k, v in ts.items()}
    tf = lambda t: 2.0 * M * N * K / t / 1e6
    print("%-20s M=%6d N=%5d K=%5d | " % (name, M, N, K) + " | ".join("%s %7.1f us %6.1f TF %+5.1f%% %s" % (
        k, med[k], tf(med[k]), (med[k] / med["ldr"] - 1) * 100, "" if same[k] is True else same[k]) for k in med), flush=True)
