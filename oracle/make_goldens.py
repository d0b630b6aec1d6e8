"""Golden-vector generator — BUILD-CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Imports the reference TRACE python package through a shim (the package is named `trace`, which
shadows the stdlib module, and it imports timm/decord/... that are absent here), instantiates
`TraceMistralForCausalLM` at the tiny geometry of `trace_amd.config.tiny()`, loads the synthetic
weights of `trace_amd.synth`, runs the reference's own `forward()` for prefill and decode, and writes
input/output vectors to tests/golden/.  Only data is written: no reference source or bytecode.

    python oracle/make_goldens.py            # regenerates tests/golden/*

transformers here is 5.15 (reference pins 4.40.1): `forward()` works, `generate()` does not
(SURVEY.md §8c), so the greedy loop below restates trace_mistral.py:336-344 around forward().
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def install_shim():
    tmp = tempfile.mkdtemp(prefix="trace_shim_")
    os.symlink(REF, os.path.join(tmp, "Trace"))
    sys.path.insert(0, tmp)
    from transformers import (AutoConfig, AutoModelForCausalLM, MistralConfig, MistralModel,  # noqa: F401
                              MistralForCausalLM, CLIPVisionModel, CLIPImageProcessor, CLIPVisionConfig,
                              PreTrainedTokenizer, AutoTokenizer, StoppingCriteria, PretrainedConfig,
                              BitsAndBytesConfig)
    sys.modules["transformers"].TRANSFORMERS_CACHE = "/tmp/hf"

    def mod(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m

    class LN(nn.LayerNorm):  # timm.models.layers.LayerNorm: eps defaults to 1e-6
        def __init__(s, c, eps=1e-6, affine=True):
            super().__init__(c, eps=eps, elementwise_affine=affine)

    mod("timm"); mod("timm.models"); mod("timm.models.regnet", RegStage=None)
    mod("timm.models.layers", LayerNorm=LN, LayerNorm2d=LN)
    mod("decord", VideoReader=None, cpu=None); mod("imageio"); mod("moviepy")
    mod("moviepy.editor", VideoFileClip=None)
    mod("scenedetect", open_video=None, SceneManager=None)
    mod("scenedetect.detectors", ContentDetector=None)
    mod("scenedetect.stats_manager", StatsManager=None)
    return tmp


def build_reference_model(cfg, tmp):
    from transformers import CLIPVisionConfig, CLIPVisionModel, CLIPImageProcessor
    from Trace.trace.model import TraceMistralForCausalLM, TraceMistralConfig
    clip_dir = os.path.join(tmp, "tiny-clip")
    vc = CLIPVisionConfig(hidden_size=cfg.vision_hidden_size, intermediate_size=cfg.vision_intermediate_size,
                          num_hidden_layers=cfg.vision_num_layers, num_attention_heads=cfg.vision_num_heads,
                          image_size=cfg.vision_image_size, patch_size=cfg.vision_patch_size,
                          hidden_act="quick_gelu", layer_norm_eps=cfg.vision_layer_norm_eps)
    CLIPVisionModel(vc).save_pretrained(clip_dir)
    CLIPImageProcessor(size={"shortest_edge": cfg.vision_image_size},
                       crop_size={"height": cfg.vision_image_size, "width": cfg.vision_image_size}
                       ).save_pretrained(clip_dir)
    mc = TraceMistralConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
        max_position_embeddings=cfg.max_position_embeddings, sliding_window=None,
        time_vocab_size=cfg.time_vocab_size, score_vocab_size=cfg.score_vocab_size,
        mm_vision_tower=clip_dir, mm_vision_select_layer=cfg.mm_vision_select_layer,
        mm_vision_select_feature="patch", mm_projector_type="spatial_slot", mm_hidden_size=cfg.mm_hidden_size,
        num_frames=cfg.num_frames, downsample_num=1, attn_implementation="eager",
    )
    model = TraceMistralForCausalLM(mc).float().eval()
    return model


def load_synth(model, cfg, dtype=torch.bfloat16):
    from trace_amd import synth
    sd_ref = model.state_dict()
    mine = synth.state_dict(cfg, dtype=dtype)
    used = set()
    new = {}
    for k in sd_ref:
        cand = [k, k.replace("vision_tower.vision_tower.", "vision_tower.vision_tower.vision_model.")]
        hit = next((c for c in cand if c in mine), None)
        if hit is None:
            if "position_ids" in k or "inv_freq" in k or "cached" in k:
                new[k] = sd_ref[k]
                continue
            raise KeyError(f"no synthetic tensor for reference key {k}")
        assert tuple(mine[hit].shape) == tuple(sd_ref[k].shape), (k, mine[hit].shape, sd_ref[k].shape)
        new[k] = mine[hit].float()
        used.add(hit)
    unused = [k for k in mine if k not in used]
    assert not unused, f"synthetic tensors with no reference counterpart: {unused[:5]}"
    model.load_state_dict(new, strict=True)
    return sorted(sd_ref.keys())


def scripted_ids(cfg):
    """A DVC-shaped stream visiting all three heads: time digits, <sep>, digits, time-<sync>,
    score digits, score-<sync>, some text, text-<sync>, then time again."""
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    t0, s0 = V + 1, V + Tv + 1
    dig = lambda base, s: [base + {"<sync>": 0, "<sep>": 1, ".": 12, **{str(i): i + 2 for i in range(10)}}[c] for c in s]
    seq = dig(t0, "0012.5") + [t0 + 1] + dig(t0, "0031.0") + [t0]
    seq += dig(s0, "4.5") + [s0]
    seq += [17, 45, 203, 99, 7, 311, 28] + [V]
    seq += dig(t0, "0040.2") + [t0 + 1] + dig(t0, "0055.9") + [t0]
    seq += dig(s0, "3.0") + [s0] + [5, 150, 62] + [V]
    return seq


@torch.no_grad()
def run_reference(model, cfg, input_ids, frames, ts, forced=None, n_new=24):
    """prefill + decode loop over the reference forward(); returns per-step masked logits + argmax ids."""
    heads = [1]
    ids = input_ids.view(1, -1)
    out = model(input_ids=ids, attention_mask=torch.ones_like(ids), images=[[frames], ["video"]], times=[[]],
                scores=[[]], video_timestamps=[ts], heads=heads, use_cache=True, return_dict=True)
    pkv = out.past_key_values
    L = out.logits.shape[1]
    step_logits, toks = [], []
    lg = out.logits[0, -1]
    mask_len = L
    n = len(forced) + 1 if forced is not None else n_new
    for step in range(n):
        step_logits.append(lg.clone())
        tok = int(torch.argmax(lg))
        toks.append(tok)
        if step == n - 1:
            break
        feed = tok if forced is None else int(forced[step])
        heads[0] = model.swap_tokens.get(feed, heads[0])            # trace_mistral.py:336-344
        mask_len += 1
        o = model(input_ids=torch.tensor([[feed]]), attention_mask=torch.ones(1, mask_len, dtype=torch.long),
                  past_key_values=pkv, heads=heads, use_cache=True, return_dict=True)
        pkv = o.past_key_values
        lg = o.logits[0, -1]
    return torch.stack(step_logits), toks, L


def _own_lowp_run(model, cfg, input_ids, frames, ts, forced, tf_logits, dtype=torch.float16):
    """The reference ITSELF in a 16-bit dtype on the same teacher-forced stream: model.half() is its own inference dtype (trace/model/builder.py:50),
    model.to(torch.bfloat16) the dtype BASELINE's configurations name (the same `torch_dtype` switch) — the anchor for the HIP path's bf16 tolerance:
    how far the reference's own 16-bit run lands from its fp32 run.  {} when this host's CPU kernels cannot run it."""
    tag = "fp16" if dtype == torch.float16 else "bf16"
    try:
        m16 = model.to(dtype)
        l16, a16, _ = run_reference(m16, cfg, input_ids, frames.to(dtype), ts, forced=forced)
        print(f"reference {tag} run: max |logit - fp32 run| =", float((l16.float() - tf_logits)[torch.isfinite(tf_logits)].abs().max()))
        return {f"tf_logits_ref_{tag}": l16.float().numpy().astype(np.float32), f"tf_argmax_ref_{tag}": np.array(a16)}
    except Exception as e:
        print(f"reference {tag} run not possible on this host:", repr(e)[:200])
        return {}
    finally:
        model.float()


def _own_fp16_run(model, cfg, input_ids, frames, ts, forced, tf_logits):
    return _own_lowp_run(model, cfg, input_ids, frames, ts, forced, tf_logits, torch.float16)


def add_arrays(npz_name, **arrays):
    """adds arrays to a committed fixture without touching what it already holds"""
    path = os.path.join(OUT, npz_name)
    old = dict(np.load(path))
    old.update(arrays)
    np.savez_compressed(path, **old)


def fp_goldens(tmp, dtype=torch.bfloat16, name="tiny_e2e.npz"):
    """dtype: what the synthetic weights and frames are rounded to before the reference (fp32 arithmetic) sees them — bf16 for the default
    library, fp16 for libtrace_hip_f16.so (tiny_e2e_f16.npz)."""
    from trace_amd import config as tcfg, synth
    cfg = tcfg.tiny(num_frames=4)
    model = build_reference_model(cfg, tmp)
    keys = load_synth(model, cfg, dtype)
    frames = synth.synth_frames(cfg, 0).to(dtype).float()
    ts = [[float(i) * 2.5] for i in range(cfg.num_frames)]
    input_ids = synth.synth_prompt_ids(cfg, n_text=24, video_pos=10)

    # stage captures through the reference's own modules
    with torch.no_grad():
        vt = model.get_model().get_vision_tower()
        feats = vt(frames)                                                           # clip_encoder.py:41-53
        slots = model.get_model().mm_projector(feats[None])                          # [1,T,8,H]
        vfeat = model.encode_images_or_videos([frames], ["video"], [ts])             # [1,T*14,H]
        (_, _, _, embeds, _, _, _) = model.prepare_inputs_labels_for_multimodal(
            input_ids.view(1, -1), torch.ones(1, input_ids.numel(), dtype=torch.long), None, None,
            [[frames], ["video"]], [[]], [[]], video_timestamps=[ts])
        # NB: never pass output_hidden_states=True to the outer model here: on transformers 5.x that
        # installs capture hooks which also fire inside the nested CLIP tower, duplicating entries of
        # its hidden_states tuple so that hidden_states[-2] silently becomes the LAST layer's output
        # (4.40.1 semantics, pinned by the reference's requirements.txt, is N+1 entries -> layer N-1).
        cap = {}
        h0 = model.model.layers[0].register_forward_hook(lambda m, i, o: cap.__setitem__("l0", o[0] if isinstance(o, tuple) else o))
        hs = model.model(inputs_embeds=embeds, use_cache=False, return_dict=True)
        h0.remove()
        clip_hs = vt.vision_tower(frames, output_hidden_states=True).hidden_states
        assert len(clip_hs) == cfg.vision_num_layers + 1, len(clip_hs)
    free_logits, free_ids, L = run_reference(model, cfg, input_ids, frames, ts, n_new=40)
    forced = scripted_ids(cfg)
    tf_logits, tf_argmax, _ = run_reference(model, cfg, input_ids, frames, ts, forced=forced)

    extra = _own_fp16_run(model, cfg, input_ids, frames, ts, forced, tf_logits) if dtype == torch.float16 else {}
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(
        os.path.join(OUT, name), **extra,
        input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
        vit_feats=feats.numpy().astype(np.float32),
        slots=slots[0].numpy().astype(np.float32),
        video_feats_rows=vfeat[0, ::7].numpy().astype(np.float32),
        embeds_rows=embeds[0, ::5].numpy().astype(np.float32),
        hidden_last_rows=hs.last_hidden_state[0, -4:].numpy().astype(np.float32),
        layer0_last_row=cap["l0"][0, -1].numpy().astype(np.float32),
        prefill_len=np.array(L),
        free_logits=free_logits.numpy().astype(np.float32), free_ids=np.array(free_ids),
        forced_ids=np.array(forced), tf_logits=tf_logits.numpy().astype(np.float32),
        tf_argmax=np.array(tf_argmax),
    )
    with torch.no_grad():
        assert torch.equal(vt(frames), feats), "reference CLIP feature selection changed state mid-run"
    print("%s: L=%d free_ids=%s" % (name, L, free_ids))
    print("tf_argmax", tf_argmax)
    if dtype != torch.bfloat16:
        return
    with open(os.path.join(OUT, "reference_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)

    # B=2 equal-length batch (reference supports it): second video + same prompt
    frames2 = synth.synth_frames(cfg, 1).to(torch.bfloat16).float()
    l2, ids2, _ = run_reference(model, cfg, input_ids, frames2, ts, n_new=16)
    np.savez_compressed(os.path.join(OUT, "tiny_video1.npz"), free_logits=l2.numpy().astype(np.float32),
                        free_ids=np.array(ids2))


def int_goldens():
    """Integer / string functions of the path captured from the reference modules."""
    from Trace.trace.model.multimodal_encoder.time_encoder import TimeTower, TimeTokenizer
    from Trace.trace.model.multimodal_encoder.score_encoder import ScoreTower, ScoreTokenizer
    from Trace.trace import conversation as conv_mod
    from Trace.trace import mm_utils as ref_mm
    from Trace.trace import constants as ref_const
    G = {}
    tt, st = TimeTower(TimeTokenizer(), hidden_dim=8), ScoreTower(ScoreTokenizer(), hidden_dim=8)
    tcases = [[12.3, 45.6], [0.0], [1234.56], [], [9999.0], [0.04], [0.05], [0.15], [99999.9], [7.0, 8.25, 100.0], [3.14159]]
    scases = [[4.5], [10.0], [], [0.0], [3.0, 2.5], [9.96], [0.04]]
    G["time_encode"] = [{"in": c, "out": tt.encode(c).tolist()} for c in tcases]
    G["score_encode"] = [{"in": c, "out": st.encode(c).tolist()} for c in scases]
    G["time_decode"] = [{"in": i, "out": TimeTokenizer().decode(i)} for i in range(13)]
    G["score_decode"] = [{"in": i, "out": ScoreTokenizer().decode(torch.tensor(i))} for i in range(13)]
    G["time_vocab"] = TimeTokenizer().get_vocab()

    # prompts: llama_2 template with each task prompt file (evaluate.py:327-332)
    prompts = {}
    pdir = os.path.join(REF, "trace", "prompts")
    for fn in sorted(os.listdir(pdir)):
        q = open(os.path.join(pdir, fn)).read()
        conv = conv_mod.conv_templates["llama_2"].copy()
        conv.append_message(conv.roles[0], "<video>\n" + q)
        conv.append_message(conv.roles[1], None)
        prompts[fn] = {"question": q, "prompt": conv.get_prompt() + "<sync>"}
    conv = conv_mod.conv_templates["llama_2"].copy()
    conv.append_message(conv.roles[0], "<video>\nhello")
    conv.append_message(conv.roles[1], "an answer")
    conv.append_message(conv.roles[0], "second turn")
    conv.append_message(conv.roles[1], None)
    prompts["_multi_turn"] = {"prompt": conv.get_prompt()}
    G["llama2_prompts"] = prompts
    G["llama2_sep"] = [conv_mod.conv_templates["llama_2"].sep, conv_mod.conv_templates["llama_2"].sep2]

    # tokenizer_MMODAL_token_all with a fake whitespace tokenizer (no sentencepiece model offline)
    class FakeTok:
        bos_token_id = 1

        def __call__(self, text):
            return types.SimpleNamespace(input_ids=[1] + [10 + len(w) for w in text.split()])

    cases = ["a bb <video>\nccc dd [/INST]<sync>", "<video>\nxx", "no modal here", "x <time> y <score> z <sync>",
             "<image> a <video> b <audio> c", "<sync>", "lead <sync><sync> tail"]
    G["tokenizer_MMODAL_token_all"] = [
        {"in": c, "out": ref_mm.tokenizer_MMODAL_token_all(c, FakeTok(), return_tensors="pt").tolist()} for c in cases]
    G["tokenizer_MMODAL_token_video"] = [
        {"in": c, "out": ref_mm.tokenizer_MMODAL_token(c, FakeTok(), -201, return_tensors="pt").tolist()}
        for c in ["a <video> b", "<video>\nq", "plain"]]
    G["get_model_name_from_path"] = [{"in": p, "out": ref_mm.get_model_name_from_path(p)}
                                     for p in ["/a/b/trace-7b/", "x/checkpoint-100", "trace"]]
    G["constants"] = {k: getattr(ref_const, k) for k in
                      ["NUM_FRAMES", "MAX_FRAMES", "IGNORE_INDEX", "IMAGE_TOKEN_INDEX", "MMODAL_TOKEN_INDEX",
                       "DEFAULT_MMODAL_TOKEN", "NUM_FRAMES_PER_SECOND"]}

    # frame sampling arithmetic of process_video (mm_utils.py:379-437): uniform linspace + timestamps
    fs = []
    for duration, fps, n in [(300, 30.0, 8), (3000, 25.0, 128), (100, 29.97, 64), (50, 10.0, 128), (7, 1.0, 8), (1, 24.0, 4)]:
        idx = np.linspace(0, duration - 1, n, dtype=int)
        if len(idx) > ref_const.MAX_FRAMES:
            idx = np.linspace(0, duration - 1, ref_const.MAX_FRAMES, dtype=int)
        fs.append({"duration": duration, "fps": fps, "num_frames": n, "indices": idx.tolist(),
                   "timestamps": [[float(i / fps)] for i in idx]})
    G["frame_sample_uniform"] = fs

    # expand2square on small synthetic images (mm_utils.py:259-270)
    from PIL import Image
    e2s = []
    rng = np.random.RandomState(0)
    for (w, h) in [(6, 4), (3, 7), (5, 5)]:
        arr = rng.randint(0, 255, size=(h, w, 3), dtype=np.uint8)
        bg = tuple(int(x * 255) for x in [0.48145466, 0.4578275, 0.40821073])
        out = ref_mm.expand2square(Image.fromarray(arr), bg)
        e2s.append({"in": arr.tolist(), "bg": list(bg), "out": np.array(out).tolist()})
    G["expand2square"] = e2s

    # process_video itself (mm_utils.py:379-471), driven through fake readers: frame k is a 4x6 image filled with k, the fake
    # processor reports (fill value, width) of every image it is handed -> which frames, in which order, after which padding
    class FakeVR:
        def __init__(self, uri=None, ctx=None, num_threads=0):
            self.n, self.fps = FakeVR.spec

        def __len__(self):
            return self.n

        def get_avg_fps(self):
            return self.fps

        def get_batch(self, ids):
            arr = np.stack([np.full((4, 6, 3), int(i) % 251, dtype=np.uint8) for i in ids])
            return types.SimpleNamespace(numpy=lambda: arr)

    class FakeProc:
        image_mean = [0.48145466, 0.4578275, 0.40821073]

        def preprocess(self, images, return_tensors="pt"):
            return {"pixel_values": torch.tensor([[int(np.asarray(im)[im.size[1] // 2, im.size[0] // 2, 0]), im.size[0], im.size[1]]
                                                  for im in images])}

    ref_mm.VideoReader, ref_mm.cpu = FakeVR, (lambda i: None)
    pv = []
    import random as _random
    for (n, fps, nf, scheme, grid, aspect) in [
            (300, 30.0, 8, "uniform", False, "pad"), (50, 10.0, 128, "uniform", False, "pad"), (3000, 25.0, 128, "uniform", False, "no"),
            (900, 30.0, 8, "fps", False, "pad"), (100, 24.0, 8, "fps", False, "pad"), (400, 25.0, 8, "rand", False, "pad"),
            (64, 8.0, 9, "uniform", True, "pad"), (400000, 30.0, 8, "uniform", False, "pad"), (20, 5.0, 8, "bogus", False, "pad")]:
        FakeVR.spec = (n, fps)
        _random.seed(1234)
        case = {"kind": "frames", "duration": n, "fps": fps, "num_frames": nf, "scheme": scheme, "image_grid": grid, "aspect": aspect}
        try:
            v, ts = ref_mm.process_video("x.mp4", FakeProc(), aspect_ratio=aspect, num_frames=nf, image_grid=grid, sample_scheme=scheme)
            case.update(picked=v.tolist(), timestamps=ts)
        except Exception as e:
            case.update(error=type(e).__name__ + ": " + str(e))
        pv.append(case)
    for (n, nf) in [(12, 8), (5, 8), (40, 6)]:
        frames = [np.full((4, 6, 3), 10 * (k % 25), dtype=np.uint8) for k in range(n)]
        ref_mm.imageio.get_reader = lambda path, frames=frames: list(frames)
        v, ts = ref_mm.process_video("x.gif", FakeProc(), aspect_ratio="pad", num_frames=nf)
        pv.append({"kind": "gif", "duration": n, "num_frames": nf, "picked": v.tolist(), "timestamps": ts})
    G["process_video"] = pv
    grids = []
    for (t, rows, cols) in [(5, None, None), (4, 2, 2), (7, None, 3), (6, 2, None), (9, 3, 3)]:
        arr = np.arange(t * 2 * 3 * 3, dtype=np.uint8).reshape(t, 2, 3, 3)
        grids.append({"t": t, "rows": rows, "cols": cols, "out": ref_mm.create_photo_grid(arr, rows, cols).tolist()})
    G["create_photo_grid"] = grids

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "host_functions.json"), "w") as f:
        json.dump(G, f, indent=0)
    print("host_functions.json written:", list(G.keys()))


def medium_goldens(tmp):
    """Real CLIP-ViT-L/14-336 geometry (hidden 1024, 24 layers, 16 heads, 577 tokens) for ONE frame through the reference's
    own vision tower + SpatialSlotPool: pins the 577-token attention, the 23-layer depth and quick-GELU at the
    size the product runs.  Stored as float16 to keep the fixture near 1 MB (values are O(1))."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=1), vision_hidden_size=1024, vision_intermediate_size=4096, vision_num_layers=24,
                              vision_num_heads=16, vision_image_size=336, vision_patch_size=14, mm_hidden_size=1024)
    model = build_reference_model(cfg, os.path.join(tmp, "medium"))
    load_synth(model, cfg)
    frames = synth.synth_frames(cfg, 7, num_frames=1).to(torch.bfloat16).float()
    with torch.no_grad():
        vt = model.get_model().get_vision_tower()
        feats = vt(frames)                                           # [1,576,1024]  clip_encoder.py:41-53
        slots = model.get_model().mm_projector(feats[None])          # [1,1,8,4096]
        clip_hs = vt.vision_tower(frames, output_hidden_states=True).hidden_states
        assert len(clip_hs) == 25 and torch.equal(clip_hs[-2][:, 1:], feats)
    np.savez_compressed(os.path.join(OUT, "medium_vit.npz"), vit_feats=feats[0].numpy().astype(np.float16),
                        slots=slots[0, 0].numpy().astype(np.float32),
                        feat_abs_mean=np.array(feats.abs().mean().item()), video_idx=np.array(7))
    print("medium_vit: feats", tuple(feats.shape), "abs mean %.4f max %.3f" % (feats.abs().mean().item(), feats.abs().max().item()))


def medium_multi_goldens(tmp):
    """The same real CLIP-ViT-L/14-336 geometry for THREE DISTINCT frames in one call of the reference's vision tower + SpatialSlotPool (round 6: the one-frame
    fixture above pins the arithmetic, this one pins that a frame's result does not depend on its neighbours in the batch or its place in a frame stream).
    To stay small it stores every frame's 8 slot rows in full (float16) and every 24th feature row of every frame (24 of 576 rows, float16)."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=3), vision_hidden_size=1024, vision_intermediate_size=4096, vision_num_layers=24,
                              vision_num_heads=16, vision_image_size=336, vision_patch_size=14, mm_hidden_size=1024)
    model = build_reference_model(cfg, os.path.join(tmp, "medium_multi"))
    load_synth(model, cfg)
    frames = synth.synth_frames(cfg, 11, num_frames=3).to(torch.bfloat16).float()
    assert not torch.equal(frames[0], frames[1]) and not torch.equal(frames[1], frames[2])
    with torch.no_grad():
        vt = model.get_model().get_vision_tower()
        feats = vt(frames)                                           # [3,576,1024]  clip_encoder.py:41-53
        slots = model.get_model().mm_projector(feats[None])          # [1,3,8,4096]
        one = vt(frames[1:2])                                        # the reference itself: a frame alone == the frame in a batch (to fp32 rounding)
        assert (one[0] - feats[1]).abs().max().item() < 1e-3
    np.savez_compressed(os.path.join(OUT, "medium_vit_multi.npz"), feat_rows=np.arange(0, 576, 24, dtype=np.int32),
                        vit_feats_rows=feats[:, ::24].numpy().astype(np.float16), slots=slots[0].numpy().astype(np.float16),
                        feat_abs_mean=np.array(feats.abs().mean().item()), feat_norm=feats.flatten(1).norm(dim=1).numpy().astype(np.float32),
                        video_idx=np.array(11))
    print("medium_vit_multi: feats", tuple(feats.shape), "abs mean %.4f max %.3f" % (feats.abs().mean().item(), feats.abs().max().item()))


def medium_llm_goldens(tmp, dtype=torch.bfloat16):
    """One decoder layer at the REAL Mistral-7B widths (hidden 4096, intermediate 14336, 32/8 heads x 128) behind the tiny ViT:
    teacher-forced logits of the reference over a stream that visits all three heads.  Pins the K = 14336 down-projection
    (the decode GEMV's many-chunk partial rows) and the 28672-wide gate|up product against the reference itself."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=1)
    model = build_reference_model(cfg, os.path.join(tmp, "medllm"))
    load_synth(model, cfg, dtype)
    frames = synth.synth_frames(cfg, 0).to(dtype).float()
    ts = [[float(i) * 2.5] for i in range(cfg.num_frames)]
    input_ids = synth.synth_prompt_ids(cfg, n_text=24, video_pos=10)
    forced = scripted_ids(cfg)
    tf_logits, tf_argmax, L = run_reference(model, cfg, input_ids, frames, ts, forced=forced)
    extra = _own_fp16_run(model, cfg, input_ids, frames, ts, forced, tf_logits) if dtype == torch.float16 else {}
    np.savez_compressed(os.path.join(OUT, "medium_llm" + ("_f16" if dtype == torch.float16 else "") + ".npz"), **extra, input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
                        forced_ids=np.array(forced), tf_logits=tf_logits.numpy().astype(np.float32), tf_argmax=np.array(tf_argmax),
                        prefill_len=np.array(L))
    print("medium_llm: L=%d steps=%d" % (L, len(tf_argmax)))


def long_ctx_goldens(tmp):
    """The C2 context length against the reference: 128 (tiny-ViT) frames -> 1792 visual rows, a 176-id prompt -> prefill
    L = 1967, one decoder layer at the real Mistral-7B widths, then teacher-forced decode (contexts 1968 ..).  Pins the causal
    prefill attention, the 256x256-tile GEMMs at M ~ 2k and the decode attention over a 2k-token KV cache."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=128), intermediate_size=14336, num_hidden_layers=1)
    model = build_reference_model(cfg, os.path.join(tmp, "longctx"))
    load_synth(model, cfg)
    frames = synth.synth_frames(cfg, 0).to(torch.bfloat16).float()
    ts = [[float(i)] for i in range(cfg.num_frames)]
    input_ids = synth.synth_prompt_ids(cfg, n_text=176, video_pos=150)
    forced = scripted_ids(cfg)[:24]
    tf_logits, tf_argmax, L = run_reference(model, cfg, input_ids, frames, ts, forced=forced)
    assert L == 1967, L
    np.savez_compressed(os.path.join(OUT, "long_ctx.npz"), input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
                        forced_ids=np.array(forced), tf_logits=tf_logits.numpy().astype(np.float32), tf_argmax=np.array(tf_argmax),
                        prefill_len=np.array(L))
    print("long_ctx: L=%d steps=%d" % (L, len(tf_argmax)))


def real_vocab_goldens(tmp):
    """The real vocabulary (V = 32000 -> 32027 global ids: lm_head 32000 rows, <sync> 32000, time 32001..32013, score
    32014..32026) on the tiny two-layer model: teacher-forced reference logits.  Full rows are 128 KB per step, so the fixture
    keeps, per step, the top-2 (value, index), 96 fixed sampled columns and the -inf pattern boundaries."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), vocab_size=32000)
    model = build_reference_model(cfg, os.path.join(tmp, "realvocab"))
    load_synth(model, cfg)
    frames = synth.synth_frames(cfg, 0).to(torch.bfloat16).float()
    ts = [[float(i) * 2.5] for i in range(cfg.num_frames)]
    input_ids = synth.synth_prompt_ids(cfg, n_text=24, video_pos=10)
    forced = scripted_ids(cfg)
    forced = [t if t >= cfg.vocab_size else (t * 97 + 13) % cfg.vocab_size for t in forced]     # spread the text ids over the table
    tf_logits, tf_argmax, L = run_reference(model, cfg, input_ids, frames, ts, forced=forced)
    NV = tf_logits.shape[1]
    cols = np.unique(np.concatenate([np.random.RandomState(3).randint(0, NV, size=80), np.arange(NV - 28, NV), [0, 1, 31999]]))
    fin = torch.isfinite(tf_logits)
    masked = torch.where(fin, tf_logits, torch.full_like(tf_logits, -1e30))
    top = torch.topk(masked, 2, dim=-1)
    np.savez_compressed(os.path.join(OUT, "real_vocab.npz"), input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
                        forced_ids=np.array(forced), tf_argmax=np.array(tf_argmax), cols=cols,
                        sampled=tf_logits[:, cols].numpy().astype(np.float32), top_val=top.values.numpy().astype(np.float32),
                        top_idx=top.indices.numpy(), finite_count=fin.sum(-1).numpy(), prefill_len=np.array(L))
    print("real_vocab: NV=%d steps=%d finite counts %s" % (NV, len(tf_argmax), sorted(set(fin.sum(-1).tolist()))))


def deep_llm_goldens(tmp, dtype=torch.bfloat16):
    """Depth: EIGHT decoder layers at the real Mistral-7B widths (1.74 B parameters, a quarter of the real stack) behind the
    tiny ViT, teacher-forced.  Shows how the bf16 noise of the HIP path accumulates with depth against the fp32 reference."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336, num_hidden_layers=8)
    model = build_reference_model(cfg, os.path.join(tmp, "deepllm"))
    load_synth(model, cfg, dtype)
    frames = synth.synth_frames(cfg, 0).to(dtype).float()
    ts = [[float(i) * 2.5] for i in range(cfg.num_frames)]
    input_ids = synth.synth_prompt_ids(cfg, n_text=24, video_pos=10)
    forced = scripted_ids(cfg)
    tf_logits, tf_argmax, L = run_reference(model, cfg, input_ids, frames, ts, forced=forced)
    extra = _own_lowp_run(model, cfg, input_ids, frames, ts, forced, tf_logits, dtype)     # the reference's own run in the fixture's 16-bit dtype (bf16 anchor: round 4)
    np.savez_compressed(os.path.join(OUT, "deep_llm" + ("_f16" if dtype == torch.float16 else "") + ".npz"), **extra, input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
                        forced_ids=np.array(forced), tf_logits=tf_logits.numpy().astype(np.float32), tf_argmax=np.array(tf_argmax),
                        prefill_len=np.array(L))
    print("deep_llm: L=%d steps=%d logit std %.3f" % (L, len(tf_argmax), tf_logits[torch.isfinite(tf_logits)].std().item()))


def mr_forced_ids(cfg, n=31):
    """The moment-retrieval answer shape (trace/eval/evaluate.py:298-357 with prompts/mr.txt: one event): 14 time-head tokens
    ('dddd.d<sep>dddd.d' + time <sync>), 4 score-head tokens ('d.d' + score <sync>), then caption text and the text <sync>."""
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    t0, s0 = V + 1, V + Tv + 1
    dig = lambda base, s: [base + {"<sync>": 0, "<sep>": 1, ".": 12, **{str(i): i + 2 for i in range(10)}}[c] for c in s]
    seq = dig(t0, "0003.7") + [t0 + 1] + dig(t0, "0011.2") + [t0]            # 14 on the time head
    seq += dig(s0, "4.0") + [s0]                                              # 4 on the score head
    seq += [(17 * k + 5) % (V - 3) + 3 for k in range(n - 19)] + [V]          # text, then the text <sync>
    return seq[:n]


def charades_goldens(tmp):
    """BASELINE config 4 (Charades-STA moment retrieval): 64 (tiny-ViT) frames -> 896 visual rows, a 191-id prompt (the llama_2
    template + prompts/mr.txt + a query is ~190 sentencepiece ids) -> prefill L = 1086 through one decoder layer at the real
    Mistral-7B widths, then the 32-token timestamp-heavy answer teacher-forced (18 of 32 steps on the time / score heads)."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=64), intermediate_size=14336, num_hidden_layers=1)
    model = build_reference_model(cfg, os.path.join(tmp, "charades"))
    load_synth(model, cfg)
    frames = synth.synth_frames(cfg, 3).to(torch.bfloat16).float()
    ts = [[float(i) * 0.5] for i in range(cfg.num_frames)]
    input_ids = synth.synth_prompt_ids(cfg, n_text=191, video_pos=150, seed=11)
    forced = mr_forced_ids(cfg, 31)
    tf_logits, tf_argmax, L = run_reference(model, cfg, input_ids, frames, ts, forced=forced)
    assert L == 1086 and len(tf_argmax) == 32, (L, len(tf_argmax))
    np.savez_compressed(os.path.join(OUT, "charades_ctx.npz"), input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
                        forced_ids=np.array(forced), tf_logits=tf_logits.numpy().astype(np.float32), tf_argmax=np.array(tf_argmax),
                        prefill_len=np.array(L), video_idx=np.array(3))
    print("charades_ctx: L=%d steps=%d" % (L, len(tf_argmax)))


def videomme_goldens(tmp):
    """The BASELINE config 5 SHAPE in the reference's own precision: 256 (tiny-ViT) frames (trace/eval/videomme/evaluate.py:215-258
    samples past constants.MAX_FRAMES = 128) -> 3584 visual rows, a 251-id prompt -> prefill L = 3834 through one real-width decoder
    layer, then 16 teacher-forced tokens at contexts 3835...  The fp8 weight path of config 5 has no reference counterpart
    (SURVEY section 5); its parity anchor is this bf16 result."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    cfg = dataclasses.replace(tcfg.tiny(num_frames=256), intermediate_size=14336, num_hidden_layers=1)
    model = build_reference_model(cfg, os.path.join(tmp, "videomme"))
    load_synth(model, cfg)
    frames = synth.synth_frames(cfg, 5).to(torch.bfloat16).float()
    ts = [[float(i) * 2.0] for i in range(cfg.num_frames)]
    input_ids = synth.synth_prompt_ids(cfg, n_text=251, video_pos=200, seed=13)
    forced = scripted_ids(cfg)[:15]
    tf_logits, tf_argmax, L = run_reference(model, cfg, input_ids, frames, ts, forced=forced)
    assert L == 3834 and len(tf_argmax) == 16, (L, len(tf_argmax))
    np.savez_compressed(os.path.join(OUT, "videomme_ctx.npz"), input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
                        forced_ids=np.array(forced), tf_logits=tf_logits.numpy().astype(np.float32), tf_argmax=np.array(tf_argmax),
                        prefill_len=np.array(L), video_idx=np.array(5))
    print("videomme_ctx: L=%d steps=%d" % (L, len(tf_argmax)))


@torch.no_grad()
def run_reference_layer_streamed(cfg_full, tmp, input_ids, frames, ts, forced, dtype=torch.float32):
    """Teacher-forced logits of the FULL-depth stack without holding it in RAM: the reference model is built with ONE decoder
    layer; its own modules do the embedding splice (prepare_inputs_labels_for_multimodal, both branches), the layer
    (`model.model.layers[0]`, the transformers MistralDecoderLayer the reference delegates to), the final norm and the four
    heads with the head mask; only the loop over layers is restated here — layer l's synthetic weights are loaded into that one
    module, applied to the whole teacher-forced sequence (prefill rows + the fed tokens: with every fed token known in advance
    one causal pass over L + n rows gives, in fp32, what L-row prefill + n cached single-token steps give), and discarded.
    Checked against the 8-layer fixture made by the unmodified reference forward() (deep_llm.npz) in `full_depth_goldens`."""
    import dataclasses
    from trace_amd import synth
    cfg1 = dataclasses.replace(cfg_full, num_hidden_layers=1)
    model = build_reference_model(cfg1, tmp)
    load_synth(model, cfg1)                                    # embeddings, towers, ViT, slot pool, norm, heads (+ layer 0)
    if dtype != torch.float32:                                 # the reference's own modules in a 16-bit dtype (model.to(dtype), builder.py:50's switch)
        model = model.to(dtype)
        frames = frames.to(dtype)
    ids = input_ids.view(1, -1)
    (_, _, _, embeds, _, _, _) = model.prepare_inputs_labels_for_multimodal(
        ids, torch.ones_like(ids), None, None, [[frames], ["video"]], [[]], [[]], video_timestamps=[ts])
    L = embeds.shape[1]
    rows = [embeds[0]]
    heads_at = [1]                                             # head that masks the logits at position L-1+i
    h = 1
    for t in forced:                                           # decode branch of the splice (trace_arch.py:345-375) per fed token
        h = model.swap_tokens.get(int(t), h)
        heads_at.append(h)
        (_, _, _, e1, _, _, _) = model.prepare_inputs_labels_for_multimodal(
            torch.tensor([[int(t)]]), torch.ones(1, L + len(rows), dtype=torch.long), [(torch.zeros(1, 1, L, 1),) * 2], None,
            None, None, None, video_timestamps=None)
        rows.append(e1[0])
    x = torch.cat(rows, 0)[None]                               # [1, L+n, H]
    N = x.shape[1]
    pos = torch.arange(N)[None]
    layer, mm = model.model.layers[0], model.model
    pe = mm.rotary_emb(x, pos)
    mask = torch.full((N, N), float("-inf")).triu(1)[None, None].to(dtype)
    keys = [k for k in layer.state_dict().keys()]
    specs = {sp[0]: sp for sp in synth.weight_specs(cfg_full)}
    for l in range(cfg_full.num_hidden_layers):
        sd = {}
        for k in keys:
            spec = specs[f"model.layers.{l}.{k}"]
            sd[k] = synth.synth_tensor(spec[0], spec[1], spec[2], torch.bfloat16).to(dtype)
        layer.load_state_dict(sd, strict=True)
        out = layer(x, attention_mask=mask, position_ids=pos, position_embeddings=pe, use_cache=False)
        x = out[0] if isinstance(out, tuple) else out
        if x.dim() == 2:
            x = x[None]
    hs = mm.norm(x)[0, L - 1:]                                  # positions L-1 .. L-1+n
    lg = torch.cat([model.lm_head(hs), model.sync_head(hs), model.time_head(hs), model.score_head(hs)], -1).float()
    V, Tv, Sv = cfg_full.vocab_size, cfg_full.time_vocab_size, cfg_full.score_vocab_size
    rng = [(0, V + 1), (V + 1, V + 1 + Tv), (V + 1 + Tv, V + 1 + Tv + Sv)]
    for i, hh in enumerate(heads_at):                          # the head mask of trace_mistral.py:244-252
        lo, hi = rng[hh]
        lg[i, :lo] = float("-inf"); lg[i, hi:] = float("-inf")
    return lg, L


def full_depth_goldens(tmp):
    """Depth: ALL 32 decoder layers at the real Mistral-7B widths (6.98 B decoder parameters) behind the tiny ViT, teacher-forced,
    by the layer-streamed reference run above.  The harness is first run at 8 layers and must reproduce deep_llm.npz (made by the
    unmodified reference forward()) to fp32 round-off."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    base = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336)
    frames = synth.synth_frames(base, 0).to(torch.bfloat16).float()
    ts = [[float(i) * 2.5] for i in range(base.num_frames)]
    input_ids = synth.synth_prompt_ids(base, n_text=24, video_pos=10)
    forced = scripted_ids(base)
    D = np.load(os.path.join(OUT, "deep_llm.npz"))
    lg8, L8 = run_reference_layer_streamed(dataclasses.replace(base, num_hidden_layers=8), os.path.join(tmp, "fd8"), input_ids, frames, ts, forced)
    ref8 = torch.from_numpy(D["tf_logits"])
    fin = torch.isfinite(ref8)
    assert torch.equal(fin, torch.isfinite(lg8)), "head mask pattern differs from the reference forward()"
    err8 = (lg8[fin] - ref8[fin]).abs().max().item()
    print("layer-streamed harness vs reference forward() at 8 layers: max |dlogit| = %.3e" % err8)
    assert err8 < 2e-4, err8
    lg, L = run_reference_layer_streamed(dataclasses.replace(base, num_hidden_layers=32), os.path.join(tmp, "fd32"), input_ids, frames, ts, forced)
    masked = torch.where(torch.isfinite(lg), lg, torch.full_like(lg, -1e30))
    np.savez_compressed(os.path.join(OUT, "full_depth_llm.npz"), input_ids=input_ids.numpy(), timestamps=np.array(ts, dtype=np.float64),
                        forced_ids=np.array(forced), tf_logits=lg.numpy().astype(np.float32), tf_argmax=masked.argmax(-1).numpy(),
                        prefill_len=np.array(L), harness_err_8_layers=np.array(err8))
    print("full_depth_llm: L=%d steps=%d logit std %.3f" % (L, lg.shape[0], lg[torch.isfinite(lg)].std().item()))


def bf16_anchor_goldens(tmp):
    """Round 4: the reference's OWN bf16 run next to its fp32 run, at 8 and at all 32 real-width layers, added to deep_llm.npz / full_depth_llm.npz
    (`tf_logits_ref_bf16`, `tf_argmax_ref_bf16`).  8 layers: the unmodified forward() of model.to(torch.bfloat16) (prefill + cached single-token steps).
    32 layers: the layer-streamed harness with the reference's modules in bf16 — first run at 8 layers next to the forward() run to show that its bf16
    error against fp32 has the same size (it cannot be bit-equal: one causal pass over L + n rows rounds differently from L rows + n cached steps)."""
    import dataclasses
    from trace_amd import config as tcfg, synth
    base = dataclasses.replace(tcfg.tiny(num_frames=4), intermediate_size=14336)
    frames = synth.synth_frames(base, 0).to(torch.bfloat16).float()
    ts = [[float(i) * 2.5] for i in range(base.num_frames)]
    input_ids = synth.synth_prompt_ids(base, n_text=24, video_pos=10)
    forced = scripted_ids(base)
    D8 = np.load(os.path.join(OUT, "deep_llm.npz"))
    ref8 = torch.from_numpy(D8["tf_logits"])
    fin = torch.isfinite(ref8)
    if "tf_logits_ref_bf16" not in D8.files:
        cfg8 = dataclasses.replace(base, num_hidden_layers=8)
        model = build_reference_model(cfg8, os.path.join(tmp, "deepllm_bf16"))
        load_synth(model, cfg8)
        chk, _, _ = run_reference(model, cfg8, input_ids, frames, ts, forced=forced)
        assert (chk[fin] - ref8[fin]).abs().max().item() < 1e-4, "the fp32 run no longer reproduces the committed deep_llm.npz"
        add_arrays("deep_llm.npz", **_own_lowp_run(model, cfg8, input_ids, frames, ts, forced, ref8, torch.bfloat16))
        del model
        D8 = np.load(os.path.join(OUT, "deep_llm.npz"))
    fw8 = torch.from_numpy(D8["tf_logits_ref_bf16"])
    lg8, _ = run_reference_layer_streamed(dataclasses.replace(base, num_hidden_layers=8), os.path.join(tmp, "fd8b"), input_ids, frames, ts, forced, torch.bfloat16)
    e_fw, e_st = (fw8[fin] - ref8[fin]).abs(), (lg8.float()[fin] - ref8[fin]).abs()
    print("8 layers, reference bf16 vs fp32: forward() max %.3f rms %.4f | layer-streamed harness max %.3f rms %.4f" %
          (e_fw.max(), e_fw.pow(2).mean().sqrt(), e_st.max(), e_st.pow(2).mean().sqrt()))
    assert 0.5 < e_st.pow(2).mean().sqrt() / e_fw.pow(2).mean().sqrt() < 2.0, "the streamed bf16 harness is not representative of forward() in bf16"
    D32 = np.load(os.path.join(OUT, "full_depth_llm.npz"))
    ref32 = torch.from_numpy(D32["tf_logits"])
    lg32, _ = run_reference_layer_streamed(dataclasses.replace(base, num_hidden_layers=32), os.path.join(tmp, "fd32b"), input_ids, frames, ts, forced, torch.bfloat16)
    lg32 = lg32.float()
    fin32 = torch.isfinite(ref32)
    assert torch.equal(fin32, torch.isfinite(lg32))
    e32 = (lg32[fin32] - ref32[fin32]).abs()
    masked = torch.where(torch.isfinite(lg32), lg32, torch.full_like(lg32, -1e30))
    add_arrays("full_depth_llm.npz", tf_logits_ref_bf16=lg32.numpy().astype(np.float32), tf_argmax_ref_bf16=masked.argmax(-1).numpy(),
               harness_bf16_vs_forward_bf16_rms_ratio_8_layers=np.array(float(e_st.pow(2).mean().sqrt() / e_fw.pow(2).mean().sqrt())))
    flips = int((masked.argmax(-1) != torch.from_numpy(D32["tf_argmax"])).sum())
    print("32 layers, reference bf16 (streamed) vs fp32: max %.3f rms %.4f; arg-max differs on %d of %d steps" % (e32.max(), e32.pow(2).mean().sqrt(), flips, lg32.shape[0]))


def stc_goldens(tmp):
    """STCConnector (north_star names it; multimodal_projector/builder.py:138-249), as far as it can be pinned without timm: the REFERENCE's own
    `STCConnector.__init__` / `.forward` run here — its einops layouts ('b t (h w) d -> b d t h w', '(b t) d h w', '(t h w) d' token order), its
    `Conv3d(k = s = (2, 2, 2), padding = 1) + SiLU` sampler and its `Linear - GELU - Linear` readout are torch / einops code inside the reference — with
    timm's `RegStage` (un-vendored, not installed) replaced by the restatement below.  So two of the four stages and all of the data movement are the
    reference's; the RegStage block arithmetic stays this build's reading of timm 0.6.13 (UNPINNED, said so wherever STC is mentioned)."""
    import dataclasses
    import torch.nn.functional as F
    from trace_amd import config as tcfg, synth
    from Trace.trace.model.multimodal_projector import builder as pb

    class LN2d(nn.LayerNorm):                      # timm LayerNorm2d: LayerNorm over the channels of an NCHW tensor, eps 1e-6
        def __init__(s, c, eps=1e-6):
            super().__init__(c, eps=eps)

        def forward(s, x):
            return F.layer_norm(x.permute(0, 2, 3, 1), s.normalized_shape, s.weight, s.bias, s.eps).permute(0, 3, 1, 2)

    class ConvNormAct(nn.Module):
        def __init__(s, cin, cout, k, groups=1, act=True):
            super().__init__()
            s.conv = nn.Conv2d(cin, cout, k, padding=k // 2, groups=groups, bias=False)
            s.bn = LN2d(cout)
            s.act = act

        def forward(s, x):
            x = s.bn(s.conv(x))
            return F.silu(x) if s.act else x

    class SE(nn.Module):
        def __init__(s, c, rd):
            super().__init__()
            s.fc1, s.fc2 = nn.Conv2d(c, rd, 1), nn.Conv2d(rd, c, 1)

        def forward(s, x):
            g = x.mean((2, 3), keepdim=True)
            return x * torch.sigmoid(s.fc2(F.silu(s.fc1(g))))

    class Bottleneck(nn.Module):                   # regnet Bottleneck at bottle_ratio 1, group_size 1 (depthwise), se_ratio 0.25 of the INPUT width
        def __init__(s, cin, cout):
            super().__init__()
            s.conv1 = ConvNormAct(cin, cout, 1)
            s.conv2 = ConvNormAct(cout, cout, 3, groups=cout)
            s.se = SE(cout, int(round(cin * 0.25)))
            s.conv3 = ConvNormAct(cout, cout, 1, act=False)
            s.downsample = ConvNormAct(cin, cout, 1, act=False) if cin != cout else None

        def forward(s, x):
            sc = x if s.downsample is None else s.downsample(x)
            return F.silu(s.conv3(s.se(s.conv2(s.conv1(x)))) + sc)

    class RegStageRestated(nn.Module):
        def __init__(s, depth, in_chs, out_chs, stride=1, dilation=1, act_layer=None, norm_layer=None):
            super().__init__()
            assert stride == 1 and dilation == 1 and act_layer is nn.SiLU
            for i in range(depth):
                s.add_module(f"b{i + 1}", Bottleneck(in_chs if i == 0 else out_chs, out_chs))

        def forward(s, x):
            for m in s.children():
                x = m(x)
            return x

    pb.RegStage, pb.LayerNorm2d = RegStageRestated, LN2d
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), mm_projector_type="stc_connector", vision_image_size=84, vision_hidden_size=256,
                              vision_num_heads=4, mm_hidden_size=256)                       # the geometry of tests/test_gpu_stc.py: 6 x 6 patches
    conn = pb.STCConnector(types.SimpleNamespace(mm_hidden_size=cfg.mm_hidden_size, hidden_size=cfg.hidden_size, downsample_num=1)).float().eval()
    sd = synth.state_dict(cfg)
    P = "model.mm_projector."
    mine = {k[len(P):]: v.float() for k, v in sd.items() if k.startswith(P)}
    assert set(mine) == set(conn.state_dict()), set(mine) ^ set(conn.state_dict())
    conn.load_state_dict(mine, strict=True)
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(4, cfg.vision_patches, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).float()
    with torch.no_grad():
        out = conn(feats[None], h=6, w=6)[0]                                                 # [3 * 4 * 4, 4096]
        # the same forward with BOTH RegStages switched to identity-width stand-ins is not possible (s1 changes the width), so the pinned part is
        # also captured on its own: sampler + readout applied to a given [1, C, t, h, w] tensor through the reference's modules
        z = torch.randn(1, cfg.hidden_size, 4, 6, 6, generator=g) * 0.5
        zs = conn.sampler(z)
        zr = conn.readout(zs[0].permute(1, 2, 3, 0).reshape(-1, cfg.hidden_size))
    cols = np.arange(0, cfg.hidden_size, 8)
    np.savez_compressed(os.path.join(OUT, "stc_connector.npz"), seed=np.array(3), out_cols=cols, out=out.numpy()[:, cols].astype(np.float32),
                        out_shape=np.array(out.shape), sampler_in_seed_note=np.array("z = randn(1, C, 4, 6, 6, same generator after feats) * 0.5"),
                        sampler_out_shape=np.array(zs.shape), sampler_readout=zr.numpy()[:, cols].astype(np.float32),
                        out_absmax=np.array(float(out.abs().max())))
    print("stc_connector: out", tuple(out.shape), "absmax %.3f" % out.abs().max().item(), "sampler out", tuple(zs.shape))


def tokenizer_goldens():
    """SURVEY 8f.2, tokenizer hookup: a tiny sentencepiece LlamaTokenizer (trained here on this repo's own SURVEY.md + DESIGN.md text,
    vocabulary 512 — the GPU test widens the tiny config's embedding table to match; the trained model file is committed as data) loaded the way the reference loads its tokenizer
    (AutoTokenizer.from_pretrained(path, use_fast=False), trace/model/builder.py:113), then the REFERENCE's tokenizer_MMODAL_token_all /
    tokenizer_MMODAL_token (trace/mm_utils.py:493-554) on the llama_2 prompts of every task file and on multi-placeholder strings."""
    import sentencepiece as spm
    from transformers import AutoTokenizer
    from Trace.trace import mm_utils as ref_mm
    from Trace.trace import conversation as conv_mod
    d = os.path.join(OUT, "sp_tiny")
    os.makedirs(d, exist_ok=True)
    corpus = os.path.join(tempfile.mkdtemp(), "corpus.txt")
    with open(corpus, "w") as f:
        f.write(open(os.path.join(REPO, "SURVEY.md")).read())
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(d, "tokenizer"), vocab_size=512, model_type="bpe", pad_id=-1,
                                   unk_id=0, bos_id=1, eos_id=2, byte_fallback=True, character_coverage=1.0,
                                   normalization_rule_name="identity", add_dummy_prefix=True, minloglevel=2)
    os.remove(os.path.join(d, "tokenizer.vocab"))
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "LlamaTokenizer", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>", "add_bos_token": True,
                   "add_eos_token": False, "legacy": True, "model_max_length": 4096}, f, indent=1)
    tok = AutoTokenizer.from_pretrained(d, use_fast=False)
    cases = []
    pdir = os.path.join(REF, "trace", "prompts")
    for fn in sorted(os.listdir(pdir)):
        q = open(os.path.join(pdir, fn)).read()
        if "{}" in q:
            q = q.format("a person opens the door")
        conv = conv_mod.conv_templates["llama_2"].copy()
        conv.append_message(conv.roles[0], "<video>\n" + q)
        conv.append_message(conv.roles[1], None)
        cases.append(conv.get_prompt() + "<sync>")
    cases += ["a bb <video>\nccc dd [/INST]<sync>", "<video>\nxx", "no modal here", "x <time> y <score> z <sync>", "<image> a <video> b <audio> c",
              "<sync>", "lead <sync><sync> tail", "Ünïcödé <video> 你好 [/INST]<sync>"]
    G = {"vocab_size": len(tok), "bos": tok.bos_token_id, "eos": tok.eos_token_id,
         "all": [{"in": c, "out": ref_mm.tokenizer_MMODAL_token_all(c, tok, return_tensors="pt").tolist()} for c in cases],
         "video": [{"in": c, "out": ref_mm.tokenizer_MMODAL_token(c, tok, -201, return_tensors="pt").tolist()} for c in cases[:7]],
         "decode": [{"in": ids, "out": tok.batch_decode([ids], skip_special_tokens=True)[0]}
                    for ids in ([1, 40, 41, 300, 2], [5, 6, 7], [400, 401, 402, 403, 2, 2])],
         "stop_ids": tok("</s>").input_ids}
    with open(os.path.join(OUT, "tokenizer_sp.json"), "w") as f:
        json.dump(G, f, indent=0)
    print("tokenizer_sp.json:", len(cases), "cases; lens", [len(c["out"]) for c in G["all"]][:6])


def preprocess_goldens():
    """Frame preprocessing of process_video (mm_utils.py:456-462): the reference's own expand2square + the HF
    CLIPImageProcessor it delegates to (PIL backend), on small synthetic frames, 'pad' and plain modes.  The processor is
    built at 56 px so the fixture stays small; the arithmetic (Pillow 8-bit bicubic resample, centre crop, x/255,
    (x-mean)/std) does not depend on the size."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPImageProcessor
    from Trace.trace import mm_utils as ref_mm
    proc = CLIPImageProcessor(size={"shortest_edge": 56}, crop_size={"height": 56, "width": 56})
    rng = np.random.RandomState(5)
    G = {"image_mean": np.array(proc.image_mean, dtype=np.float32), "image_std": np.array(proc.image_std, dtype=np.float32)}
    for tag, (T, H, W) in {"land": (3, 90, 160), "port": (2, 150, 70), "square": (2, 64, 64), "up": (2, 20, 33)}.items():
        frames = rng.randint(0, 256, size=(T, H, W, 3), dtype=np.uint8)
        # smooth half of them so the resampler sees structure, not only noise
        frames[0] = (np.add.outer(np.arange(H) * 255 // max(H - 1, 1), np.arange(W) * 255 // max(W - 1, 1)) // 2)[..., None].astype(np.uint8)
        G[f"{tag}_frames"] = frames
        images = [Image.fromarray(f) for f in frames]
        padded = [ref_mm.expand2square(im, tuple(int(x * 255) for x in proc.image_mean)) for im in images]
        G[f"{tag}_pad"] = proc.preprocess(padded, return_tensors="pt")["pixel_values"].numpy().astype(np.float32)
        G[f"{tag}_plain"] = proc.preprocess(images, return_tensors="pt")["pixel_values"].numpy().astype(np.float32)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "preprocess.npz")
    np.savez_compressed(out, **G)
    print("wrote", out, {k: v.shape for k, v in G.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = install_shim()
    if "--int-only" in sys.argv:
        int_goldens()
        sys.exit(0)
    if "--preprocess-only" in sys.argv:
        preprocess_goldens()
        sys.exit(0)
    if "--medium-only" in sys.argv:
        medium_goldens(tmp)
        sys.exit(0)
    if "--medium-multi-only" in sys.argv:
        medium_multi_goldens(tmp)
        sys.exit(0)
    if "--medium-llm-only" in sys.argv:
        medium_llm_goldens(tmp)
        sys.exit(0)
    if "--long-ctx-only" in sys.argv:
        long_ctx_goldens(tmp)
        sys.exit(0)
    if "--real-vocab-only" in sys.argv:
        real_vocab_goldens(tmp)
        sys.exit(0)
    if "--deep-llm-only" in sys.argv:
        deep_llm_goldens(tmp)
        sys.exit(0)
    if "--tokenizer-only" in sys.argv:
        tokenizer_goldens()
        sys.exit(0)
    if "--charades-only" in sys.argv:
        charades_goldens(tmp)
        sys.exit(0)
    if "--videomme-only" in sys.argv:
        videomme_goldens(tmp)
        sys.exit(0)
    if "--f16-only" in sys.argv:
        fp_goldens(tmp, torch.float16, "tiny_e2e_f16.npz")
        medium_llm_goldens(tmp, torch.float16)
        deep_llm_goldens(tmp, torch.float16)
        sys.exit(0)
    if "--stc-only" in sys.argv:
        stc_goldens(tmp)
        sys.exit(0)
    if "--bf16-anchor-only" in sys.argv:
        bf16_anchor_goldens(tmp)
        sys.exit(0)
    if "--full-depth-only" in sys.argv:
        full_depth_goldens(tmp)
        sys.exit(0)
    int_goldens()
    fp_goldens(tmp)
    fp_goldens(tmp, torch.float16, "tiny_e2e_f16.npz")
    medium_llm_goldens(tmp, torch.float16)
    deep_llm_goldens(tmp, torch.float16)
    medium_goldens(tmp)
    medium_multi_goldens(tmp)
    medium_llm_goldens(tmp)
    long_ctx_goldens(tmp)
    real_vocab_goldens(tmp)
    deep_llm_goldens(tmp)
    charades_goldens(tmp)
    videomme_goldens(tmp)
    full_depth_goldens(tmp)
    bf16_anchor_goldens(tmp)
    stc_goldens(tmp)
    tokenizer_goldens()
    preprocess_goldens()
