import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M = 73856
for N, epi in ((4096, E.EPI_QUICKGELU), (1024, E.EPI_RESIDUAL), (4096, E.EPI_NONE)):
    for K in (64, 128, 256, 512, 1024, 2048):
        A = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        R = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == E.EPI_RESIDUAL else None
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        lib = E._lib.load()
        fn = lambda: E._lib.check(lib.trace_op_gemm(E._ptr(A), K, E._ptr(W), K, E._ptr(C), N, E._ptr(b), E._ptr(R), N, M, N, K, epi, E._stream()))
        us = timeit(fn)
        nblk = ((M + 255) // 256) * (N // 256)
        print(f"N={N} epi={epi} K={K}: {us:.1f} us  ({us / (nblk / 256):.2f} us per block-round, {2.0*M*N*K/us/1e6:.0f} TF)", flush=True)
