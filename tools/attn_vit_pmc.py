"""rocprofv3 --pmc workload: the ViT attention launch (170 frames) on the LDS-DMA kernel (3 launches) and on the register-staged one (3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 170
q, k, v = rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64)
for var in (110, 115):
    ops.set_gemm_variant(var)
    for _ in range(3):
        ops.attention(q, k, v, False, 0.125)
torch.cuda.synchronize()
