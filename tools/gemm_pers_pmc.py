"""Workload for rocprofv3 --pmc passes over the three 256x256 GEMM kernels on one long-K shape (one tile per workgroup: the K loop and
nothing else): gemm_ldr (variant 4), gemm_pers (5), gemm_pers on 32x32x16 MFMAs (5 with build option 1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import engine as E
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
A, W = rnd(3934, 14336), rnd(4096, 14336, scale=0.02)
for v, o in ((4, 0), (5, 0), (5, 1)):
    ops.set_gemm_variant(300 + o)
    ops.set_gemm_variant(v)
    for _ in range(3):
        ops.gemm(A, W)
ops.set_gemm_variant(300)
ops.set_gemm_variant(0)
torch.cuda.synchronize()
