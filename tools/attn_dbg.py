"""Decode-attention microbenchmark: time per launch and effective cache-stream bandwidth for B x nsplit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _lib
from trace_amd.engine import ops, _ptr, _stream
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
L = _lib.load()
for Bn in (1, 16, 32, 64):
    q = torch.randn(Bn, nq * 128, device=dev).to(torch.bfloat16)
    kc = torch.randn(Bn, nkv, max_ctx, 128, device=dev).to(torch.bfloat16)
    vt = torch.randn(Bn, nkv, 128, max_ctx, device=dev).to(torch.bfloat16)
    pos = torch.full((Bn,), 2100, dtype=torch.int32, device=dev)
    ws = torch.zeros((Bn * nq * 64 * 130,), dtype=torch.float32, device=dev)
    o = torch.empty_like(q)
    byts = Bn * nkv * 2101 * 128 * 2 * 2
    for nsplit in ((1, 2, 4) if Bn >= 32 else (2, 4, 8, 16)):
        fn = lambda: _lib.check(L.trace_op_attn_decode(_ptr(q), _ptr(kc), _ptr(vt), _ptr(pos), _ptr(o), _ptr(ws), Bn, nq, nkv, max_ctx, nsplit, 0.088, _stream()))
        t = timeit(fn)
        print(f"B={Bn} nsplit={nsplit}: {t:.1f} us  {byts / t / 1e6:.2f} TB/s", flush=True)
