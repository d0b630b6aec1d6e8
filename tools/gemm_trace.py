"""Where does a GEMM tile's time go?  Per-workgroup phase stamps (s_memrealtime, 100 MHz) written by gemm_glds_kernel
when a trace buffer is set: [0] entry, [1] first K-tile landed, [2] main loop done, [3] epilogue staged in LDS,
[4] stores drained, [5] HW_ID, [6] XCC_ID.  Prints mean phase durations and the gap between consecutive workgroups
on the same CU (dispatch + launch overhead the next tile pays before its first instruction)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E, _lib
from trace_amd.engine import ops, _ptr
dev = torch.device("cuda", 0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
L = _lib.load()
for name, M, N, K, epi in [("vit_fc1", 73856, 4096, 1024, E.EPI_QUICKGELU), ("vit_qkv", 73856, 3072, 1024, E.EPI_NONE),
                           ("vit_fc2", 73856, 1024, 4096, E.EPI_RESIDUAL)]:
    A, W, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
    R = rnd(M, N) if epi == E.EPI_RESIDUAL else None
    nblk = ((M + 255) // 256) * (N // 256)
    for _ in range(2): ops.gemm(A, W, bias=b, R=R, epilogue=epi)
    buf = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
    _lib.check(L.trace_op_set_gemm_trace(_ptr(buf)))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.gemm(A, W, bias=b, R=R, epilogue=epi); e.record()
    torch.cuda.synchronize()
    _lib.check(L.trace_op_set_gemm_trace(None))
    t = buf.cpu().numpy()
    us = lambda a: a / 100.0
    t0 = t[:, 0].min()
    span = us(t[:, 4].max() - t0)
    ph = [us((t[:, i + 1] - t[:, i]).astype("float64")).mean() for i in range(4)]
    print(f"{name}: kernel {s.elapsed_time(e)*1e3:.0f} us, stamps span {span:.0f} us, {nblk} tiles")
    print(f"   mean per tile: prologue(first tile landed) {ph[0]:.2f} | main loop {ph[1]:.2f} | epilogue math+LDS {ph[2]:.2f} | stores {ph[3]:.2f} | total {sum(ph):.2f} us")
    # consecutive workgroups on one CU: key = (xcc, se, cu) ; 1 WG per CU at a time (128 KB LDS)
    per = collections.defaultdict(list)
    for i in range(nblk):
        hw, xcc = int(t[i, 5]), int(t[i, 6]) & 0xf
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        per[key].append((int(t[i, 0]), int(t[i, 4])))
    gaps, tiles = [], []
    for k, v in per.items():
        v.sort()
        tiles.append(len(v))
        for (a0, a1), (b0, b1) in zip(v, v[1:]): gaps.append(us(b0 - a1))
    import numpy as np
    gaps = np.array(gaps)
    print(f"   {len(per)} CUs seen, tiles/CU {min(tiles)}..{max(tiles)}; gap end->next start on a CU: mean {gaps.mean():.2f} us, p10 {np.percentile(gaps,10):.2f}, p90 {np.percentile(gaps,90):.2f}")
    first_start = us(np.array([min(v)[0] for v in per.values()]) - t0)
    last_end = us(np.array([max(x[1] for x in v) for v in per.values()]) - t0)
    print(f"   first-tile start spread {first_start.min():.1f}..{first_start.max():.1f} us; last-tile end {last_end.min():.0f}..{last_end.max():.0f} us")
