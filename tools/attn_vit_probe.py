"""ViT attention launch anatomy (170 frames x 16 heads x 577 tokens, head_dim 64): full kernel and knock-outs
(variant 111 = no K/V loads after the first tile, 112 = loads + LDS staging + barriers only), several batch sizes."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
def timeit(fn, iters=8, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
for Bn in (170, 32):
    q, k, v = rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64)
    # transpose-only time: run with a 1-row query (attention negligible)? measured separately by rocprof; here: total incl. transpose
    for var, name in ((110, "persistent LDS-resident kernel"), (111, "persistent, no DMA"), (112, "persistent, no tile math"),
                      (113, "4-wave LDS-DMA ring kernel"), (115, "register-staged kernel (round 1 structure, uniform mask branch)")):
        ops.set_gemm_variant(var)
        t = timeit(lambda: ops.attention(q, k, v, False, 0.125))
        fl = 4.0 * Bn * 16 * 577 * 577 * 64
        print(f"frames={Bn} {name}: {t:.1f} us (incl. ~{63 * Bn / 170:.0f} us transpose)  {fl / t / 1e6:.0f} TFLOP/s", flush=True)
    ops.set_gemm_variant(110)
