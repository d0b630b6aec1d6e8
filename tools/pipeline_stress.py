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
ap.add_argument("--gemm-variant", type=int, default=0, help="trace_op_set_gemm_variant: 4 = every 256^2 GEMM on the loader-wave kernel (no persistent kernel)")
ap.add_argument("--plan", default="", help="comma-separated phases run in one process, each a '+'-joined list of variant codes, e.g. 0,501,4")
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


eng._dbg = dbg


def run_phase(variants, steps):
    """one run of `steps` identical steps under the given trace_op_set_gemm_variant codes -> number of steps that differ from the phase's step 0"""
    for v in variants:
        _lib.check(eng.lib.trace_op_set_gemm_variant(v))
    log["feats"], log["kv"], log["ids"] = [], [], []
    cur["feats"] = {}
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
    print(f"phase {variants}: " + ("all steps identical" if not bad else f"{bad} of {steps - 1} steps differ from step 0"), flush=True)
    for v in variants:                      # back to the defaults
        _lib.check(eng.lib.trace_op_set_gemm_variant({500: 500, 501: 500}.get(v, 0)))
    return bad


# --plan "0,501,4": phases, each a '+'-joined list of variant codes (0 = the shipped defaults, 500 = ticketed tile walk, 501 = static deal, 4 = no persistent
# kernel; the ring GEMM and the big-tile attention have their own codes: trace_hip.h).  History (rounds 3-4, profiles/r0[34]_pipeline_stress_*): the only
# configuration in which a step ever differed from step 0 had the ViT's LayerNorm fold on; the fold was never root-caused and left the product in round 5.
# Sensitivity: --max-new 200 makes the decode stage as long as the encode stage (24 tokens: a tenth of it), i.e. ~10x the overlap exposure per step.
plan = [[int(x) for x in ph.split("+")] for ph in (a.plan.split(",") if a.plan else [str(a.gemm_variant or 0)])]       # 0 = the shipped defaults
res = []
for ph in plan:
    res.append((ph, run_phase(ph, a.steps)))
print("summary: " + "; ".join(f"{ph}: {b} differing steps of {a.steps - 1}" for ph, b in res))
eng.close()
