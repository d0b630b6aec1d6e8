"""gemm_ldr.hip (variant 4 = the default for 256x256 tiles: 8 MFMA + 4 loader waves) against gemm.hip's own 256x256 kernel
(variant 3) on the ViT / prefill shapes: results must be bit-identical; time per launch and TFLOP/s of both."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [("vit fc1 gelu", 170 * 577, 4096, 1024, E.EPI_QUICKGELU, True), ("vit qkv", 170 * 577, 3072, 1024, E.EPI_NONE, True),
          ("vit fc2 res", 170 * 577, 1024, 4096, E.EPI_RESIDUAL, True), ("vit out res", 170 * 577, 1024, 1024, E.EPI_RESIDUAL, True),
          ("prefill o res", 3934, 4096, 4096, E.EPI_RESIDUAL, False), ("prefill down res", 3934, 4096, 14336, E.EPI_RESIDUAL, False),
          ("small 700x512", 700, 512, 1024, E.EPI_NONE, True),
          # shapes the dispatcher currently gives to the 128x128 kernel (variant 2): is the loader-wave kernel faster there now?
          ("prefill qkv pair", 3934, 6144, 4096, E.EPI_NONE, False), ("prefill qkv one", 1967, 6144, 4096, E.EPI_NONE, False),
          ("prefill o one", 1967, 4096, 4096, E.EPI_RESIDUAL, False), ("prefill gateup one", 1967, 28672, 4096, E.EPI_SWIGLU, False),
          ("prefill down one", 1967, 4096, 14336, E.EPI_RESIDUAL, False), ("patch embed", 170 * 576, 1024, 640, E.EPI_NONE, False),
          ("slot readout", 170 * 8, 4096, 1024, E.EPI_NONE, False)]
only = sys.argv[1:]
for name, M, N, K, epi, has_bias in shapes:
    if only and not any(o in name for o in only): continue
    A, W = rnd(M, K), rnd(N, K, scale=0.03)
    bias = rnd(N) if has_bias else None
    R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
    out = {}
    for v in (2, 3, 4):
        ops.set_gemm_variant(v)
        out[v] = ops.gemm(A, W, bias=bias, R=R, epilogue=epi)
        t = timed(lambda: ops.gemm(A, W, bias=bias, R=R, epilogue=epi), n=10)
        out[("t", v)] = t
    ops.set_gemm_variant(0)
    d = (out[3].float() - out[4].float()).abs()
    same = torch.equal(out[3], out[4])
    tf = lambda t: 2.0 * M * N * K / t / 1e6
    print("%-18s M=%6d N=%5d K=%5d  v2(128^2) %7.1f us  v3 %7.1f us (%6.1f TF)  v4 %7.1f us (%6.1f TF)  %s  max|d| %.4g  v2==v4 %s"
          % (name, M, N, K, out[("t", 2)], out[("t", 3)], tf(out[("t", 3)]), out[("t", 4)], tf(out[("t", 4)]), "BIT-EQUAL" if same else "differs",
             d.max().item(), torch.equal(out[2], out[4])), flush=True)
