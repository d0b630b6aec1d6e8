import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
nq, nkv, max_ctx = 32, 8, 2304
for Bn in (1, 8):
    q = (torch.randn(Bn, nq * 128, device=dev)).to(torch.bfloat16)
    kc = torch.randn(Bn, nkv, max_ctx, 128, device=dev).to(torch.bfloat16)
    vc = torch.randn(Bn, nkv, max_ctx, 128, device=dev).to(torch.bfloat16)
    pos = torch.full((Bn,), 2000, dtype=torch.int32, device=dev)
    lib = ops
    import ctypes as C
    from trace_amd import _lib
    L = _lib.load()
    ws = torch.zeros((Bn * nq * 64 * 130,), dtype=torch.float32, device=dev)
    o = torch.empty_like(q)
    from trace_amd.engine import _ptr, _stream
    for nsplit in (32, 64):
        for dbg in (0, 3, 1, 2):
            ops.set_gemm_variant(100 + dbg)
            fn = lambda: _lib.check(L.trace_op_attn_decode(_ptr(q), _ptr(kc), _ptr(vc), _ptr(pos), _ptr(o), _ptr(ws), Bn, nq, nkv, max_ctx, nsplit, 0.088, _stream()))
            print(f"B={Bn} nsplit={nsplit} dbg={dbg}: {timeit(fn):.1f} us", flush=True)
            ops.set_gemm_variant(100)
            torch.cuda.synchronize()
            # reset tickets possibly left non-zero by cut-off runs
    empty = lambda: None
