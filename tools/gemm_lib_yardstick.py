"""Yardstick, not product: the library GEMM (hipBLASLt behind torch.nn.functional.linear) on the tower's and the prefill's shapes next to this repo's
kernels, interleaved rounds, medians.  Says how far the hand-written 256x256 kernels are from what the vendor's tuned kernels reach on the same
part at the same moment (clock, power).  The product never calls the library.      python tools/gemm_lib_yardstick.py"""
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
shapes = [("vit qkv", 170 * 577, 3072, 1024, True), ("vit out", 170 * 577, 1024, 1024, True), ("vit fc1", 170 * 577, 4096, 1024, True), ("vit fc2", 170 * 577, 1024, 4096, True),
          ("prefill qkv pair", 3934, 6144, 4096, False), ("prefill gateup pair", 3934, 28672, 4096, False), ("prefill down", 3934, 4096, 14336, False),
          ("square 8192", 8192, 8192, 8192, False)]
for name, M, N, K, has_bias in shapes:
    A, W = rnd(M, K), rnd(N, K, scale=0.03)
    bias = rnd(N) if has_bias else None
    arms = {"ours": lambda: ops.gemm(A, W, bias=bias, epilogue=E.EPI_NONE), "lib": lambda: torch.nn.functional.linear(A, W, bias)}
    ts = {k: [] for k in arms}
    for r in range(7):
        for k, f in arms.items():
            ts[k].append(timed(f))
    med = {k: statistics.median(v[1:]) for k, v in ts.items()}
    d = (arms["ours"]().float() - arms["lib"]().float()).abs().max().item()
    tf = lambda t: 2.0 * M * N * K / t / 1e6
    print("%-20s M=%6d N=%5d K=%5d | " % (name, M, N, K) + " | ".join("%s %7.1f us %6.1f TF" % (k, med[k], tf(med[k])) for k in med) + " | max|d| %.3g" % d, flush=True)
# the ViT attention shape through the library's fused attention, for the same reason
q, k, v = (rnd(170, 16, 577, 64) for _ in range(3))
t = statistics.median([timed(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v)) for _ in range(5)])
print("vit attention 170 x 16 x 577 x 64 via F.scaled_dot_product_attention: %.1f us (%.1f TF)" % (t, 4.0 * 170 * 16 * 577 * 577 * 64 / t / 1e6))
