"""ViT attention at the bench's launch shape (170 frames x 16 heads x 577 tokens, head_dim 64): the 4 x 32-row LDS-DMA kernel (variant 190) against the
round-5 192-row kernel with a 4-stage (191) and a 3-stage (192) K/V ring; interleaved rounds, median; outputs compared."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
def timed(fn, n=6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
names = {190: "4 x 32-row kernel", 191: "192-row kernel, 4-stage ring", 192: "192-row kernel, 3-stage ring"}
KO = {110: "full kernel", 111: "no K/V loads after the first tiles", 112: "no tile math (loads + barriers only)", 113: "without the trailing key on the VALU",
      114: "without the output stores", 115: "without trailing key, output stores and Q fetch"}
if "--knockout" in sys.argv:          # where the time goes: knock-out runs of both kernels (trace_op_set_gemm_variant(111 / 112))
    q, k, v = rnd(170, 577, 16, 64), rnd(170, 577, 16, 64), rnd(170, 577, 16, 64)
    for var in (190, 191, 192):
        ops.set_gemm_variant(var)
        for ko, kn in KO.items():
            if ko > 112 and var != 192: continue          # (the round-6 knock-outs exist in the 192-row kernel only)
            ops.set_gemm_variant(ko)
            print(f"frames=170 {names[var]}, {kn}: {timed(lambda: ops.attention(q, k, v, False, 0.125)):.1f} us", flush=True)
    ops.set_gemm_variant(110); ops.set_gemm_variant(192)
    sys.exit(0)
for Bn in (170, 32):
    q, k, v = rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64)
    outs, ts = {}, {v_: [] for v_ in names}
    for r in range(6):
        for var in names:
            ops.set_gemm_variant(var)
            if r == 0: outs[var] = ops.attention(q, k, v, False, 0.125)
            ts[var].append(timed(lambda: ops.attention(q, k, v, False, 0.125)))
    ops.set_gemm_variant(192)
    fl = 4.0 * Bn * 16 * 577 * 577 * 64
    for var, name in names.items():
        t = statistics.median(ts[var][1:])
        d = (outs[var].float() - outs[190].float()).abs().max().item()
        print(f"frames={Bn} {name}: {t:.1f} us  {fl / t / 1e6:.0f} TFLOP/s = {fl / t / 1e6 / 2500:.3f} of peak  max|d| vs 32-row kernel {d:.4f}", flush=True)
