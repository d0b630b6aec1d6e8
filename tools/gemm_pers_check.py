"""gemm_pers.hip (variant 5 = persistent workgroups with ticketed tiles, 6 = the same with a static tile deal) against gemm_ldr.hip
(variant 4): results must be bit-identical — including ragged M, grids smaller than the chip, repeated launches (the ticket counters
re-arm themselves) and the in-place residual; time per launch and TFLOP/s of each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def timed(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [("tiny 300x256", 300, 256, 128, E.EPI_NONE, True), ("small 700x512", 700, 512, 1024, E.EPI_QUICKGELU, True),
          ("ragged 5000x1024 res", 5000, 1024, 256, E.EPI_RESIDUAL, True), ("glu 2100x1024", 2100, 1024, 512, E.EPI_SWIGLU, False),
          ("vit fc1 gelu", 170 * 577, 4096, 1024, E.EPI_QUICKGELU, True), ("vit qkv", 170 * 577, 3072, 1024, E.EPI_NONE, True),
          ("vit fc2 res", 170 * 577, 1024, 4096, E.EPI_RESIDUAL, True), ("vit out res", 170 * 577, 1024, 1024, E.EPI_RESIDUAL, True),
          ("prefill qkv pair", 3934, 6144, 4096, E.EPI_NONE, False), ("prefill o res", 3934, 4096, 4096, E.EPI_RESIDUAL, False),
          ("prefill gateup pair", 3934, 28672, 4096, E.EPI_SWIGLU, False), ("prefill down res", 3934, 4096, 14336, E.EPI_RESIDUAL, False),
          ("prefill gateup one", 1967, 28672, 4096, E.EPI_SWIGLU, False)]
only = sys.argv[1:]
bad = 0
for name, M, N, K, epi, has_bias in shapes:
    if only and not any(o in name for o in only): continue
    A, W = rnd(M, K), rnd(N, K, scale=0.03)
    bias = rnd(N) if has_bias else None
    R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
    out = {}
    for v in (4, 5, 6):
        ops.set_gemm_variant(v)
        out[v] = ops.gemm(A, W, bias=bias, R=R, epilogue=epi)
        for _ in range(3):                                  # repeated launches: the counters must come back to zero every time
            again = ops.gemm(A, W, bias=bias, R=R, epilogue=epi)
            if not torch.equal(again, out[v]): out[("unstable", v)] = True
        out[("t", v)] = timed(lambda: ops.gemm(A, W, bias=bias, R=R, epilogue=epi))
    ops.set_gemm_variant(0)
    tf = lambda t: 2.0 * M * N * K / t / 1e6
    e5, e6 = torch.equal(out[4], out[5]), torch.equal(out[4], out[6])
    bad += (not e5) + (not e6) + len([k for k in out if isinstance(k, tuple) and k[0] == "unstable"])
    d5 = (out[4].float() - out[5].float()).abs().max().item()
    print("%-22s M=%6d N=%5d K=%5d  ldr %7.1f us (%6.1f TF)  pers %7.1f us (%6.1f TF)  static %7.1f us (%6.1f TF)  %s %s  max|d| %.4g %s"
          % (name, M, N, K, out[("t", 4)], tf(out[("t", 4)]), out[("t", 5)], tf(out[("t", 5)]), out[("t", 6)], tf(out[("t", 6)]),
             "BIT-EQUAL" if e5 else "DIFFERS", "BIT-EQUAL" if e6 else "DIFFERS", d5,
             "UNSTABLE" if any(isinstance(k, tuple) and k[0] == "unstable" for k in out) else ""), flush=True)
    if not e5:
        x = (out[4] != out[5]).nonzero()
        print("   first mismatches (row, col):", x[:6].tolist(), " count", x.shape[0], flush=True)
print("MISMATCHES", bad)
