"""Where does the pipelined schedule's rare id difference come from?  Runs many short identical steps through TraceEngine.generate_stream and, per step,
checksums (a) every video's ViT features, (b) the prefilled K / V^T rows and last-position hidden rows of every KV slot of the step's bank — both on the
encode stream, before the decode stage may start — and (c) the decoded ids.  Identical inputs: every step must reproduce step 0; the first level that
does not names the stage.      python tools/pipeline_stress.py [--steps 100] [--frames 32] [--max-new 24] [--B 128] [--sequential]"""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import config as tcfg, synth, _lib
from trace_amd.engine import TraceEngine

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--max-new", type=int, default=24)
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--sequential", action="store_true")
ap.add_argument("--gemm-variant", type=int, default=0, help="trace_op_set_gemm_variant: 4 = every 256^2 GEMM on the loader-wave kernel (no persistent kernel, no LayerNorm fold)")
ap.add_argument("--plan", default="", help="comma-separated phases run in one process, each a '+'-joined list of variant codes, e.g. 500,500,502")
ap.add_argument("--trace", action="store_true", help="per tower call, checksum what every stage of every ViT layer leaves (trace_debug_vit_trace) and, on a "
                "mismatch with step 0, name the first (call, layer, stage, 256-row panel) that differs — LayerNorm-fold path only")
ap.add_argument("--adaptive", action="store_true", help="after the plan: if a phase on the shipped tile walk (500) differed, also run 501 (static deal) and 150 (no LayerNorm fold)")
a = ap.parse_args()
cfg = tcfg.trace_7b(a.frames)
B, n_new = a.B, a.max_new
ids = synth.synth_prompt_ids(cfg, n_text=176, video_pos=150).tolist()
L = 175 + a.frames * cfg.tokens_per_frame
eng = TraceEngine(cfg, max_batch=2 * B, max_ctx=(L + n_new + 63) // 64 * 64, max_frames=a.frames, max_new_tokens=n_new,
                  vit_batch_frames=TraceEngine.full_round_frames(cfg))
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
videos = [synth.synth_frames(cfg, b, num_frames=a.frames, dtype=torch.bfloat16, device="cuda") for b in range(B)]
ts = [[[float(i)] for i in range(a.frames)] for _ in range(B)]
rng = torch.Generator().manual_seed(1)
V = cfg.vocab_size
forced = [[V + 3 + int(x) for x in torch.randint(0, 10, (n_new,), generator=rng)] for _ in range(B)]
batch = (videos, ts, [ids] * B, [1] * B, forced)

kc, vc, xl = C.c_void_p(), C.c_void_p(), C.c_void_p()
st = (C.c_int64 * 8)()
_lib.check(eng.lib.trace_debug_buffers(eng.h, C.byref(kc), C.byref(vc), C.byref(xl), st))
layer_stride, slot_stride, head_stride, ctx_pad, NL, NKV, HD, H = [int(x) for x in st]


class Dev:                       # a [n] int16 view of device memory for torch
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i2", "data": (ptr, False), "version": 3}


nslots = 2 * B
K = torch.as_tensor(Dev(kc.value, layer_stride * NL), device="cuda").view(NL, nslots, NKV, ctx_pad, HD)
VT = torch.as_tensor(Dev(vc.value, layer_stride * NL), device="cuda").view(NL, nslots, NKV, HD, ctx_pad)
XL = torch.as_tensor(Dev(xl.value, max(nslots, 64) * H), device="cuda").view(-1, H)
w = torch.arange(1, 1 + 4096, device="cuda", dtype=torch.int64)

log = {"feats": [], "kv": [], "ids": []}
cur = {"feats": {}}
STAGES = ["qkv out", "attention out", "out-proj out (residual stream)", "row statistics after out-proj", "fc1 out", "fc2 out (residual stream)", "row statistics after fc2"]
tr = {"on": a.trace, "ref": None, "sum": [], "first": []}
if a.trace:
    import ctypes as _C
    tr["words"] = int(eng.lib.trace_debug_vit_trace(eng.h, None, 0))
    tr["cap"] = (B * a.frames + eng.vit_batch_frames - 1) // eng.vit_batch_frames + 2
    tr["buf"] = torch.zeros(tr["cap"] * tr["words"], dtype=torch.int64, device="cuda")
    tr["vL"] = cfg.vision_layers_used
    tr["panels"] = tr["words"] // (tr["vL"] * 7)


def trace_arm():
    if tr["on"]:
        eng.lib.trace_debug_vit_trace(eng.h, C.c_void_p(tr["buf"].data_ptr()), tr["cap"])       # (also resets the call index: the records of a step start at 0)


def trace_collect():
    """on the encode stream, after a step's encode: this step's records against step 0's -> per (call, layer, stage) the number of differing panels and the first one"""
    if not tr["on"]:
        return
    snap = tr["buf"].clone().view(tr["cap"], tr["vL"], 7, tr["panels"])
    if tr["ref"] is None:
        tr["ref"] = snap
        tr["sum"].append(None); tr["first"].append(None)
    else:
        d = snap != tr["ref"]
        tr["sum"].append(d.sum(-1).to(torch.int32))
        tr["first"].append(torch.where(d.any(-1), d.to(torch.int8).argmax(-1), torch.full_like(d[..., 0], -1, dtype=torch.int64)).to(torch.int32))
    trace_arm()


def dbg(tag, idx, t):
    if tag == "feats":
        x = t.view(torch.int16).to(torch.int64)
        cur["feats"][idx % B] = (x.view(-1, 4096) * w).sum()               # position-weighted; stays on the device: a host sync here would change the schedule under test
    else:                        # prefilled: bank's slots idx .. idx + B - 1
        k = K[:, idx:idx + B, :, :L].to(torch.int64).sum(dim=(0, 2, 3, 4))
        v = VT[:, idx:idx + B, :, :, :L].to(torch.int64).sum(dim=(0, 2, 3, 4))
        x = (XL[idx:idx + B].to(torch.int64) * w[:H]).sum(dim=1)
        log["kv"].append(torch.stack([k, v, x], 1))
        log["feats"].append(dict(cur["feats"])); cur["feats"] = {}
        trace_collect()


eng._dbg = dbg


def run_phase(variants, steps):
    """one run of `steps` identical steps under the given trace_op_set_gemm_variant codes -> number of steps that differ from the phase's step 0"""
    for v in variants:
        _lib.check(eng.lib.trace_op_set_gemm_variant(v))
    log["feats"], log["kv"], log["ids"] = [], [], []
    cur["feats"] = {}
    tr["ref"], tr["sum"], tr["first"] = None, [], []
    trace_arm()
    t0 = time.time()
    if a.sequential:
        outs = []
        for _ in range(steps):
            eng.encode_prefill(videos, ts, [ids] * B, 0)
            dbg("prefilled", 0, None)
            outs.append(eng.decode(range(B), [1] * B, n_new, -1, False, forced)[0])
    else:
        outs = [o[0] for o in eng.generate_stream([batch] * steps, n_new, eos=-1, use_graph=False)]
    torch.cuda.synchronize()
    print(f"phase {variants}: {steps} steps in {time.time() - t0:.0f} s ({'sequential' if a.sequential else 'pipelined'})", flush=True)
    kv = [t.cpu() for t in log["kv"]]
    feats = [{b: int(v) for b, v in d.items()} for d in log["feats"]]
    bad = 0
    for k in range(1, steps):
        f_bad = [b for b in range(B) if feats[k].get(b) != feats[0].get(b)]
        kv_bad = torch.nonzero((kv[k] != kv[0]).any(dim=1)).flatten().tolist()
        kv_cols = (kv[k] != kv[0]).any(dim=0).tolist()
        id_bad = [(b, next(i for i, (x, y) in enumerate(zip(outs[k][b], outs[0][b])) if x != y), sum(int(x != y) for x, y in zip(outs[k][b], outs[0][b])))
                  for b in range(B) if outs[k][b] != outs[0][b]]
        if f_bad or kv_bad or id_bad:
            bad += 1
            print(f"  step {k}: ViT features differ for videos {f_bad[:8]}; prefilled state differs for slots {kv_bad[:8]} (K, V^T, last hidden: {kv_cols}); "
                  f"ids differ for (sequence, first token, count) {id_bad[:8]}")
    if tr["on"]:
        eng.lib.trace_debug_vit_trace(eng.h, None, 0)
        for k in range(1, steps):
            sm = tr["sum"][k].cpu()
            if int(sm.sum()) == 0:
                continue
            fp = tr["first"][k].cpu()
            nz = torch.nonzero(sm)                                        # (call, layer, stage), lexicographic = execution order
            c0, l0, s0 = [int(x) for x in nz[0]]
            print(f"  step {k} trace: first difference in tower call {c0}, layer {l0}, stage {s0} ({STAGES[s0]}), panel {int(fp[c0, l0, s0])} "
                  f"(rows {int(fp[c0, l0, s0]) * 256}..): {int(sm[c0, l0, s0])} differing panel(s) there; the stages that follow in that call: " +
                  ", ".join(f"L{int(l)}/{STAGES[int(st)].split()[0]}:{int(sm[int(c), int(l), int(st)])}@{int(fp[int(c), int(l), int(st)])}" for c, l, st in nz[1:14] if int(c) == c0) +
                  f"; calls with differences: {sorted(set(int(c) for c, _, _ in nz))}")
    print(f"phase {variants}: " + ("all steps identical" if not bad else f"{bad} of {steps - 1} steps differ from step 0"), flush=True)
    for v in variants:                      # back to the defaults
        _lib.check(eng.lib.trace_op_set_gemm_variant({500: 500, 501: 500, 502: 500, 510: 510, 511: 510, 150: 150, 151: 150}.get(v, 0)))
    return bad


# --plan "500,500,502" : phases, each a '+'-joined list of variant codes (500 = shipped tile walk, 501 = static deal, 502 = round 3's plain-store re-arm,
# 510 = LayerNorm-fold statistics moved with agent-scope atomics (shipped), 511 = with round 3's plain stores / loads, 150 = no LayerNorm fold,
# 4 = no persistent kernel).  --adaptive: if a shipped-walk phase (500) shows a difference, add the phases that localise it.
# Sensitivity: the wrong tile only ever appeared while the OTHER stage's kernels shared the GPU with the tower; --max-new 200 makes the decode stage as
# long as the encode stage (24 tokens: a tenth of it), i.e. ~10x the exposure per step.
plan = [[int(x) for x in ph.split("+")] for ph in (a.plan.split(",") if a.plan else [str(a.gemm_variant or 0)])]       # 0 = the shipped defaults
res = []
for ph in plan:
    res.append((ph, run_phase(ph, a.steps)))
if a.adaptive and any(b for ph, b in res if ph == [500]):
    for ph in ([501], [150]):
        res.append((ph, run_phase(ph, a.steps)))
print("summary: " + "; ".join(f"{ph}: {b} differing steps of {a.steps - 1}" for ph, b in res))
eng.close()
