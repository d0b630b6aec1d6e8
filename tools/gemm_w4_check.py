"""gemm_w4.hip (variant 8: the persistent 256x256 tile on 4 waves of 128x128) against gemm_ldr.hip (variant 4) and gemm_pers.hip (variant 5): bit equality
on small / ragged / large shapes with every epilogue, then interleaved timings on the ViT and prefill shapes (the library GEMM beside them as a yardstick).\npython tools/gemm_w4_check.py [--time] [--soak N]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def run(v, A, W, bias, R, epi):
    ops.set_gemm_variant(v)
    try:
        return ops.gemm(A, W, bias=bias, R=R, epilogue=epi)
    finally:
        ops.set_gemm_variant(0)
bad = 0
for (M, N, K) in [(300, 256, 192), (256, 512, 256), (1000, 1024, 1024), (5000, 512, 4096), (2049, 2048, 320), (98090, 1024, 1024), (70000, 4096, 1024)]:
    for epi, nm in [(E.EPI_NONE, "none"), (E.EPI_QUICKGELU, "gelu"), (E.EPI_RESIDUAL, "res"), (E.EPI_SWIGLU, "swiglu")]:
        for has_bias in ([True, False] if epi != E.EPI_SWIGLU else [False]):
            A, W = rnd(M, K), rnd(N, K, scale=0.03)
            bias = rnd(N) if has_bias else None
            R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
            ref = run(4, A, W, bias, R, epi)
            for rep in range(3):
                out = run(8, A, W, bias, R, epi)
                ok = torch.equal(out, ref)
                if not ok:
                    bad += 1
                    d = (out.float() - ref.float()).abs()
                    rows = torch.nonzero((out != ref).any(dim=1)).flatten()
                    print(f"MISMATCH M={M} N={N} K={K} {nm} bias={has_bias} rep={rep}: {int((out != ref).sum())} of {out.numel()} differ, max {d.max().item():.3g}, rows {rows[:6].tolist()} .. {rows[-3:].tolist()}", flush=True)
                    break
            else:
                print(f"ok M={M} N={N} K={K} {nm} bias={has_bias}", flush=True)
print("mismatching cases:", bad)
if "--soak" in sys.argv:
    # many launches of the multi-tile walks (6 and 17 tiles per workgroup), every result compared: the kernel's counted waits are the kind of code that fails rarely
    n = int(sys.argv[sys.argv.index("--soak") + 1])
    for (M, N, K, epi) in [(98090, 1024, 1024, E.EPI_NONE), (70000, 4096, 1024, E.EPI_QUICKGELU), (3934, 28672, 4096, E.EPI_SWIGLU), (98090, 1024, 1024, E.EPI_RESIDUAL)]:
        A, W = rnd(M, K), rnd(N, K, scale=0.03)
        bias = rnd(N) if epi != E.EPI_SWIGLU else None
        R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
        ref = run(4, A, W, bias, R, epi)
        wrong = 0
        ops.set_gemm_variant(8)
        for i in range(n):
            out = ops.gemm(A, W, bias=bias, R=R, epilogue=epi)
            if i % 2:                                        # (every other launch runs back to back with the next: the compare is a different kernel mix in between)
                wrong += int(not torch.equal(out, ref))
        ops.set_gemm_variant(0)
        print(f"soak M={M} N={N} K={K} epi={epi}: {wrong} wrong of {n // 2} compared launches ({n} launched)", flush=True)
if "--time" in sys.argv:
    def timed(fn, n=5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(); a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    shapes = [("vit fc1 gelu", 170 * 577, 4096, 1024, E.EPI_QUICKGELU, True), ("vit qkv", 170 * 577, 3072, 1024, E.EPI_NONE, True),
              ("vit fc2 res", 170 * 577, 1024, 4096, E.EPI_RESIDUAL, True), ("vit out res", 170 * 577, 1024, 1024, E.EPI_RESIDUAL, True),
              ("vit fc2 none", 170 * 577, 1024, 4096, E.EPI_NONE, True),
              ("prefill qkv pair", 3934, 6144, 4096, E.EPI_NONE, False), ("prefill o res", 3934, 4096, 4096, E.EPI_RESIDUAL, False),
              ("prefill gateup pair", 3934, 28672, 4096, E.EPI_SWIGLU, False), ("prefill down res", 3934, 4096, 14336, E.EPI_RESIDUAL, False),
              ("square 8192", 8192, 8192, 8192, E.EPI_NONE, False)]
    for name, M, N, K, epi, has_bias in shapes:
        A, W = rnd(M, K), rnd(N, K, scale=0.03)
        bias = rnd(N) if has_bias else None
        R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
        # (variant, opt): gemm_w4 opt bit 0 = without the re-aligning barrier, bit 2 = with L2 touches of the A panel
        arms = {"ldr": (4, 0), "pers": (5, 0), "w4": (8, 0), "w4 one barrier": (8, 1), "w4 + touches": (8, 4)}
        if epi == E.EPI_NONE:
            arms["lib"] = (-1, 0)
        ts = {k: [] for k in arms}
        for r in range(7):
            for k, (v, o) in arms.items():
                if v < 0:
                    ts[k].append(timed(lambda: torch.nn.functional.linear(A, W, bias)))
                else:
                    ops.set_gemm_variant(540 + o)
                    ops.set_gemm_variant(v)
                    ts[k].append(timed(lambda: ops.gemm(A, W, bias=bias, R=R, epilogue=epi)))
                    ops.set_gemm_variant(0)
                    ops.set_gemm_variant(540)
        med = {k: statistics.median(v[1:]) for k, v in ts.items()}
        tf = lambda t: 2.0 * M * N * K / t / 1e6
        print("%-20s M=%6d N=%5d K=%5d | " % (name, M, N, K) + " | ".join("%s %7.1f us %6.1f TF" % (k, med[k], tf(med[k])) for k in med), flush=True)
