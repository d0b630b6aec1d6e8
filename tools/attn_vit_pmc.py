"""rocprofv3 --pmc workload: the ViT attention launch on the 4 x 32-row LDS-DMA kernel (variant 190, 3 launches) and on the round-5 192-row kernel
(191, 3 launches).   python tools/attn_vit_pmc.py [frames]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.engine import ops
dev = torch.device("cuda", 0)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 170
q, k, v = rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64), rnd(Bn, 577, 16, 64)
for var in (190, 192):
    ops.set_gemm_variant(var)
    for _ in range(3):
        ops.attention(q, k, v, False, 0.125)
ops.set_gemm_variant(192)
torch.cuda.synchronize()
