import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def bf16_anchor_report(lg, M, tag, narrow_width=13):
    """The HIP path's bf16 logits next to the REFERENCE's own bf16 run of the same teacher-forced stream (`tf_logits_ref_bf16`, made by
    oracle/make_goldens.py::bf16_anchor_goldens from model.to(torch.bfloat16), trace/model/builder.py:50's dtype switch), both against the
    reference's fp32 run.  lg: [steps, NV] torch fp32 logits of one row.  Returns a dict of the measured figures and appends them to
    gpurun_out/parity_measured.txt; the caller asserts on them."""
    import numpy as np
    import torch
    ref = torch.from_numpy(M["tf_logits"])
    rb = torch.from_numpy(M["tf_logits_ref_bf16"])
    fin = torch.isfinite(ref)
    e_hip, e_ref = (lg[fin] - ref[fin]).abs(), (rb[fin] - ref[fin]).abs()
    neg = torch.full_like(ref, -1e30)
    am = lambda x: torch.where(fin, x, neg).argmax(-1)
    a32, ahip, ab = am(ref), am(lg), am(rb)
    narrow = fin.sum(-1) == narrow_width                         # steps on the 13-way time / score heads
    srt = torch.sort(torch.where(fin, ref, neg), dim=-1, descending=True).values
    margin = srt[:, 0] - srt[:, 1]
    r = {"hip_max": float(e_hip.max()), "hip_rms": float(e_hip.pow(2).mean().sqrt()), "ref_bf16_max": float(e_ref.max()),
         "ref_bf16_rms": float(e_ref.pow(2).mean().sqrt()), "steps_13way": int(narrow.sum()),
         "hip_flips_13way": int((ahip != a32)[narrow].sum()), "ref_bf16_flips_13way": int((ab != a32)[narrow].sum()),
         "hip_vs_ref_bf16_agree_13way": int((ahip == ab)[narrow].sum()),
         "hip_flips_13way_margin_gt_0.5": int(((ahip != a32) & narrow & (margin > 0.5)).sum()),
         "ref_bf16_flips_13way_margin_gt_0.5": int(((ab != a32) & narrow & (margin > 0.5)).sum())}
    line = (f"{tag}: vs reference fp32 — HIP bf16 max {r['hip_max']:.3f} rms {r['hip_rms']:.4f} | reference's own bf16 run max {r['ref_bf16_max']:.3f} rms "
            f"{r['ref_bf16_rms']:.4f}; 13-way arg-max ({r['steps_13way']} steps): HIP differs from fp32 on {r['hip_flips_13way']} "
            f"({r['hip_flips_13way_margin_gt_0.5']} with reference margin > 0.5), reference-bf16 on {r['ref_bf16_flips_13way']} "
            f"({r['ref_bf16_flips_13way_margin_gt_0.5']}); HIP == reference-bf16 on {r['hip_vs_ref_bf16_agree_13way']}")
    print(line)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_measured.txt"), "a") as f:
            f.write(line + "\n")
    return r
