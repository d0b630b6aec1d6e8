"""fp8 weight path (BASELINE config 5; no reference counterpart): how the logit error grows with depth under each quantisation scheme, on CPU.

Not a test (pytest does not collect it): `python tests/study_fp8_error_growth.py` writes profiles/r04_fp8_error_growth.txt.  Test infrastructure: it uses the
oracle's own pieces (splice, towers, heads) and restates its decoder layer with a quantisation hook at the four projections, streaming the synthetic
weights of the 32 real-width layers one layer at a time (the layer-streamed shape of oracle/make_goldens.py::run_reference_layer_streamed).  The fp32 pass
must reproduce the reference fixtures deep_llm.npz (8 layers) and full_depth_llm.npz (32 layers) before anything else is trusted.

Schemes (e4m3 = torch.float8_e4m3fn, round to nearest even, max 448; scales fp32 = amax / 448):
  bf16          the engine's storage roundings only (the parity path)
  w8a8          weights e4m3, one scale per output row; activations e4m3, one scale per token row          (llm_weights_fp8 = 1)
  scheme2       w8a8 on the prefill rows, weight-only (bf16 activations) on the decode rows               (llm_weights_fp8 = 2, C5's default)
  wonly         weight-only everywhere
  wonly_b128    weight-only, one scale per (output row, 128-wide k block)                                 (not built: measured here to decide whether to)
  w8a8_b128     both operands block-scaled per 128 k                                                     (not built)
The question the round-3 review put: does any scheme keep the 13-way time / score decisions of the 32-layer stack within TWICE the flip count of the
reference's own bf16 run (tests/golden/full_depth_llm.npz: tf_argmax_ref_bf16 vs tf_argmax)?"""
import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import trace_oracle as O  # noqa: E402
from trace_amd import config as tcfg, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
F8 = torch.float8_e4m3fn
DEPTHS = (1, 2, 4, 8, 16, 32)


def rbf(x):
    return x.to(torch.bfloat16).float()


def q_rows(x):                      # one scale per row (dim -1 reduced)
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / 448.0
    return (x / s).to(F8).float() * s


def q_blocks(x, b=128):             # one scale per (row, b-wide block of the last dim)
    sh = x.shape
    y = x.reshape(*sh[:-1], sh[-1] // b, b)
    s = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / 448.0
    return ((y / s).to(F8).float() * s).reshape(sh)


SCHEMES = {
    # name: (weight quantiser or None, activation quantiser for prefill rows or None, activation quantiser for decode rows or None, bf16 storage roundings)
    "fp32": (None, None, None, False),
    "bf16": (None, None, None, True),
    "w8a8": (q_rows, q_rows, q_rows, True),
    "scheme2": (q_rows, q_rows, None, True),
    "wonly": (q_rows, None, None, True),
    "wonly_b128": (q_blocks, None, None, True),
    "w8a8_b128": (q_blocks, q_blocks, q_blocks, True),
}


def main():
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    base = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336)
    cfg32 = dataclasses.replace(base, num_hidden_layers=32)
    cfg1 = dataclasses.replace(base, num_hidden_layers=1)
    F32 = np.load(os.path.join(GOLD, "full_depth_llm.npz"))
    D8 = np.load(os.path.join(GOLD, "deep_llm.npz"))
    sd1 = {k: v.float() for k, v in synth.state_dict(cfg1).items()}
    ora = O.Oracle(cfg1, sd1, emulate_bf16=False)
    frames = synth.synth_frames(base, 0).to(torch.bfloat16).float()
    ts = F32["timestamps"].tolist()
    ids = torch.from_numpy(F32["input_ids"])
    forced = F32["forced_ids"].tolist()
    with torch.no_grad():
        emb = ora.splice(ids, ora.encode_video(frames, ts))
        L = emb.shape[0]
        rows = [emb] + [ora.decode_embed(int(t))[None] for t in forced]
        x0 = torch.cat(rows, 0)                                   # [L + n, H]: every fed token is known in advance -> one causal pass
        heads = [1]
        for t in forced:
            heads.append(O.swap_head(cfg1, int(t), heads[-1]))
    N = x0.shape[0]
    nq, nkv, hd = base.num_attention_heads, base.num_key_value_heads, base.head_dim
    pos = torch.arange(N)
    causal = torch.arange(N)[None, :] > pos[:, None]
    specs = {sp[0]: sp for sp in synth.weight_specs(cfg32)}
    xs = {name: x0.clone() for name in SCHEMES}
    logits_at = {name: {} for name in SCHEMES}

    def head_logits(x, rnd):
        h = ora._rms(x, sd1["model.norm.weight"])
        h = rbf(h) if rnd else h
        lg = []
        for i, hh in enumerate(heads):
            lg.append(ora.logits(h[L - 1 + i], hh))
        return torch.stack(lg)

    with torch.no_grad():
        for l in range(32):
            W = {}
            for k in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj",
                      "input_layernorm", "post_attention_layernorm"):
                sp = specs[f"model.layers.{l}.{k}.weight"]
                W[k] = synth.synth_tensor(sp[0], sp[1], sp[2], torch.bfloat16).float()
            wq = {}
            for name, (qw, qa_pre, qa_dec, rnd) in SCHEMES.items():
                key = qw.__name__ if qw else None
                if key not in wq:
                    wq[key] = {k: (qw(v) if qw and v.dim() == 2 else v) for k, v in W.items()}
                Wl = wq[key]
                r = rbf if rnd else (lambda t: t)

                def act(a):                                         # the GEMM's A operand, quantised per scheme on prefill / decode rows
                    if qa_pre is None and qa_dec is None:
                        return a
                    out = a.clone()
                    if qa_pre is not None:
                        out[:L] = qa_pre(a[:L])
                    if qa_dec is not None:
                        out[L:] = qa_dec(a[L:])
                    return out

                x = xs[name]
                h = r(ora._rms(x, Wl["input_layernorm"]))
                ha = act(h)
                q = r(ha @ Wl["self_attn.q_proj"].t()).view(N, nq, hd)
                k_ = r(ha @ Wl["self_attn.k_proj"].t()).view(N, nkv, hd)
                v = r(ha @ Wl["self_attn.v_proj"].t()).view(N, nkv, hd)
                q, k_ = r(ora._rope(q, pos)), r(ora._rope(k_, pos))
                kk, vv = k_.repeat_interleave(nq // nkv, dim=1), v.repeat_interleave(nq // nkv, dim=1)
                s = torch.einsum("lhd,chd->hlc", q, kk) * (hd ** -0.5)
                s = s.masked_fill(causal[None], float("-inf"))
                o = r(torch.einsum("hlc,chd->lhd", torch.softmax(s, dim=-1), vv).reshape(N, nq * hd))
                x = r(x + r(act(o) @ Wl["self_attn.o_proj"].t()))
                h = r(ora._rms(x, Wl["post_attention_layernorm"]))
                ha = act(h)
                a = r(torch.nn.functional.silu(ha @ Wl["mlp.gate_proj"].t()) * (ha @ Wl["mlp.up_proj"].t()))
                x = r(x + r(act(a) @ Wl["mlp.down_proj"].t()))
                xs[name] = x
                if l + 1 in DEPTHS:
                    logits_at[name][l + 1] = head_logits(x, rnd)
            print(f"layer {l + 1}/32", flush=True)

    # the fp32 pass against the reference fixtures
    ref8, ref32 = torch.from_numpy(D8["tf_logits"]), torch.from_numpy(F32["tf_logits"])
    fin = torch.isfinite(ref32)
    e8 = (logits_at["fp32"][8][fin] - ref8[fin]).abs().max().item()
    e32 = (logits_at["fp32"][32][fin] - ref32[fin]).abs().max().item()
    assert e8 < 5e-4 and e32 < 2e-3, (e8, e32)
    narrow = fin.sum(-1) == 13
    neg = torch.full_like(ref32, -1e30)
    am = lambda t: torch.where(fin, t, neg).argmax(-1)
    out = [f"fp8 error growth with depth (CPU emulation, teacher-forced stream of the reference fixtures: {N - L + 1} steps, {int(narrow.sum())} on the 13-way time / score heads; "
           f"tiny ViT, real-width decoder layers, synthetic N(0, 0.02^2) weights).  The fp32 pass reproduces deep_llm.npz to {e8:.1e} and full_depth_llm.npz to {e32:.1e}.",
           "error = logits(scheme, first d layers + final norm + heads) - logits(fp32, same depth): rms / max over the finite entries; flips = 13-way arg-max "
           "decisions that differ from the fp32 run's (in brackets: those where the fp32 top-2 margin exceeds 0.5)", ""]
    hdr = "scheme      " + "".join(f"| d={d:<2} rms   max  flips  " for d in DEPTHS)
    out.append(hdr)
    summary = {}
    for name in SCHEMES:
        if name == "fp32":
            continue
        line = f"{name:<12}"
        for d in DEPTHS:
            ref, lg = logits_at["fp32"][d], logits_at[name][d]
            e = (lg[fin] - ref[fin])
            srt = torch.sort(torch.where(fin, ref, neg), dim=-1, descending=True).values
            margin = srt[:, 0] - srt[:, 1]
            fl = (am(lg) != am(ref)) & narrow
            line += f"| {e.pow(2).mean().sqrt().item():6.3f} {e.abs().max().item():5.2f} {int(fl.sum()):3d} ({int((fl & (margin > 0.5)).sum())}) "
            summary[(name, d)] = (e.pow(2).mean().sqrt().item(), int(fl.sum()))
        out.append(line)
    ab = torch.from_numpy(F32["tf_argmax_ref_bf16"])
    ref_flips = int(((ab != am(ref32)) & narrow).sum())
    out += ["", f"anchor: the reference's own bf16 run (model.to(bfloat16)) differs from its fp32 run on {ref_flips} of the {int(narrow.sum())} 13-way decisions at 32 layers; "
            f"the bar the review set is twice that = {2 * ref_flips}.",
            "at 32 layers: " + ", ".join(f"{n} {summary[(n, 32)][1]}" for n in SCHEMES if n != "fp32")]
    ok = [n for n in SCHEMES if n not in ("fp32", "bf16") and summary[(n, 32)][1] <= 2 * ref_flips]
    out.append("schemes inside the bar: " + (", ".join(ok) if ok else "none"))
    text = "\n".join(out)
    print(text)
    with open(os.path.join(ROOT, "profiles", "r04_fp8_error_growth.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
