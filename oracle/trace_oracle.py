"""ORACLE — test infrastructure, NOT product code.

CPU (PyTorch fp32) restatement of the reference's TRACE inference hot path, written from the
reference's algorithm, each function citing the reference file:line it follows.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product
path (trace_amd/) never does and fails loudly when the HIP library is missing.

Parity pinning: the reference has no tests or golden vectors of its own, so this
oracle is pinned against outputs of the reference itself, run in the build container by
`oracle/make_goldens.py` (imports /root/reference through a shim, loads the same synthetic weights,
dumps `tests/golden/*.npz|json`).  `tests/test_oracle_golden.py` checks every function here against
those fixtures.  The third-party arithmetic the reference delegates to (transformers 4.40.1
CLIPVisionModel / MistralModel; container has 5.15.0, same eager math) is restated from the
published architecture and anchored by the same fixtures.  STC connector: pinned against the
reference's own `STCConnector.forward` (tests/golden/stc_connector.npz: layouts, Conv3d sampler,
readout MLP, token order) EXCEPT for the RegStage block — timm 0.6.13 is not importable here; the
block is restated from its published structure, cross-checked against transformers' independent
RegNet-Y block (tests/test_oracle_golden.py), and that part says "parity unpinned".

`emulate_bf16=True` rounds activations to bf16 at the points where the HIP engine stores bf16
tensors to HBM (same points a bf16 reference model rounds at, minus the ones fused away), so GPU
vs oracle differences reduce to accumulation order.  `emulate_bf16=torch.float16` does the same for
the fp16 library (libtrace_hip_f16.so).  `emulate_bf16=False` is the plain fp32 oracle that is
compared against the fp32 reference fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

VIS = "model.vision_tower.vision_tower.vision_model."

# ----------------------------------------------------------------------------------------------
# integer pieces: tokenizers / encoders / head switching / id-stream parsing
# ----------------------------------------------------------------------------------------------
# vocab of TimeTokenizer / ScoreTokenizer: multimodal_encoder/time_encoder.py:80-88, score_encoder.py:83-96
NUM_VOCAB = {"<sync>": 0, "<sep>": 1, **{str(i): i + 2 for i in range(10)}, ".": 12}
NUM_IDS = {v: k for k, v in NUM_VOCAB.items()}


def _encode_numbers(values: Sequence[float], fmt: str) -> List[int]:
    """time_encoder.py:52-68 / score_encoder.py:52-70: fixed-width format, <sep> between values,
    trailing <sync>.  The reference tokenises with a regex over the vocab keys; for strings made of
    digits and '.', that is one token per character."""
    ids: List[int] = []
    strs = [format(v, fmt) for v in values]
    for i, s in enumerate(strs):
        if i:
            ids.append(NUM_VOCAB["<sep>"])
        for ch in s:
            if ch not in NUM_VOCAB:          # e.g. '-' : regex findall silently drops it
                continue
            ids.append(NUM_VOCAB[ch])
    ids.append(NUM_VOCAB["<sync>"])
    return ids


def time_encode(timestamps: Sequence[float]) -> List[int]:
    return _encode_numbers(timestamps, "0>6.1f")


def score_encode(scores: Sequence[float]) -> List[int]:
    return _encode_numbers(scores, "0>3.1f")


def num_decode(ids: Sequence[int]) -> str:
    """PreTrainedTokenizer.decode on these vocabularies = concatenated tokens (drivers call it with a
    single id: evaluate.py:395,408)."""
    return "".join(NUM_IDS[int(i)] for i in ids)


def swap_head(cfg, last_token: int, head: int) -> int:
    """trace_mistral.py:86-88,336-344: V -> time head(1); V+1 -> score head(2); V+Tv+1 -> text head(0)."""
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    return {V: 1, V + 1: 2, V + Tv + 1: 0}.get(int(last_token), head)


def head_range(cfg, head: int) -> Tuple[int, int]:
    """trace_mistral.py:246-252."""
    V, Tv, Sv = cfg.vocab_size, cfg.time_vocab_size, cfg.score_vocab_size
    return [(0, V + 1), (V + 1, V + Tv + 1), (V + Tv + 1, V + Tv + Sv + 1)][head]


def parse_output_ids(cfg, ids: Sequence[int]) -> Dict[str, list]:
    """evaluate.py:373-410 (same logic in inference.py:82-128): split the global id stream into
    timestamps / scores / caption id lists."""
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    t0, s0 = V + 1, V + Tv + 1
    timestamps, scores, captions = [], [], []
    cur_t, cur_s, cur_c = [], [], []
    tbuf, sbuf = "", ""
    for idx in ids:
        idx = int(idx)
        if idx < V:
            cur_c.append(idx)
        elif idx == V:
            captions.append(cur_c)
            cur_c = []
        elif idx < s0:
            if idx == t0:                      # time <sync>
                if tbuf:
                    cur_t.append(float(tbuf))
                timestamps.append(cur_t)
                cur_t, tbuf = [], ""
            elif idx == t0 + 1:                # time <sep>
                if tbuf:
                    cur_t.append(float(tbuf))
                tbuf = ""
            else:
                tbuf += NUM_IDS[idx - t0]
        else:
            if idx == s0:
                if sbuf:
                    cur_s.append(float(sbuf))
                scores.append(cur_s)
                cur_s, sbuf = [], ""
            elif idx == s0 + 1:
                if sbuf:
                    cur_s.append(float(sbuf))
                sbuf = ""
            else:
                sbuf += NUM_IDS[idx - s0]
    return {"timestamps": timestamps, "scores": scores, "captions": captions}


# ----------------------------------------------------------------------------------------------
# floating-point pieces
# ----------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------
# frame preprocessing (process_video's image branch, trace/mm_utils.py:259-270,456-462 -> HF CLIPImageProcessor.preprocess,
# transformers==4.40.1 image_processing_clip.py: resize(shortest_edge, BICUBIC) -> center_crop -> rescale(1/255) -> normalize;
# the resize itself is Pillow's ImagingResample (src/libImaging/Resample.c, 8-bit path), restated here integer for integer)
# ----------------------------------------------------------------------------------------------
def _bicubic(x: float) -> float:
    a = -0.5                                   # Resample.c bicubic_filter
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pillow_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: per output index (first source index, tap count) and the taps as
    22-bit fixed point."""
    import numpy as np
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << 22)) if k < 0 else int(0.5 + k * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pillow_resize(img, out_w: int, out_h: int):
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C], Pillow Image.resize(BICUBIC): horizontal pass over the source rows the
    vertical pass needs, uint8 in between, both in 22-bit fixed point with round-half-up and clamp (Resample.c
    ImagingResampleHorizontal_8bpc / Vertical_8bpc)."""
    import numpy as np
    H, W, _ = img.shape

    def run(src, bounds, kk):                  # resample axis 1
        s64 = src.astype(np.int64)
        out = np.empty((src.shape[0], bounds.shape[0], src.shape[2]), dtype=np.uint8)
        for xx in range(bounds.shape[0]):
            x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
            acc = (1 << 21) + (s64[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(1)
            out[:, xx, :] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
        return out

    cur = img
    bv, kv = pillow_coeffs(H, out_h)
    if out_w != W:
        bh, kh = pillow_coeffs(W, out_w)
        if out_h != H:
            first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
            cur = cur[first:last]
            bv = bv.copy()
            bv[:, 0] -= first
        cur = run(cur, bh, kh)
    if out_h != H:
        cur = run(cur.transpose(1, 0, 2), bv, kv).transpose(1, 0, 2)
    return cur


def preprocess_frames(frames_u8, image_mean, image_std, size: int, pad: bool):
    """uint8 [T,H,W,3] -> float32 [T,3,size,size] as process_video does it (mm_utils.py:456-462): optional expand2square
    with background int(mean*255) (mm_utils.py:259-270), resize so the shorter side is `size` (longer side int(size*long/short),
    HF get_resize_output_image_size), centre crop (top = (h-size)//2, left = (w-size)//2), float32(float64(u8)*(1/255)),
    (x - float32(mean)) / float32(std)."""
    import numpy as np
    mean = np.asarray(image_mean, dtype=np.float32)
    std = np.asarray(image_std, dtype=np.float32)
    out = []
    for f in np.asarray(frames_u8):
        H, W, _ = f.shape
        if pad and H != W:
            S = max(H, W)
            canvas = np.empty((S, S, 3), dtype=np.uint8)
            canvas[:] = np.array([int(float(m) * 255) for m in image_mean], dtype=np.uint8)
            if W > H:
                y0 = (W - H) // 2
                canvas[y0:y0 + H] = f
            else:
                x0 = (H - W) // 2
                canvas[:, x0:x0 + W] = f
            f, H, W = canvas, S, S
        if W <= H:
            nw, nh = size, int(size * H / W)
        else:
            nh, nw = size, int(size * W / H)
        r = pillow_resize(f, nw, nh)
        top, left = (nh - size) // 2, (nw - size) // 2
        c = r[top:top + size, left:left + size]
        x = (c.astype(np.float64) * (1 / 255)).astype(np.float32)
        x = (x - mean) / std
        out.append(x.transpose(2, 0, 1))
    return np.stack(out).astype(np.float32)


class Oracle:
    def __init__(self, cfg, weights: Dict[str, torch.Tensor], emulate_bf16=False):
        """emulate_bf16: False = plain fp32; True = round to bf16 at the engine's storage points; a torch dtype (torch.float16 for
        libtrace_hip_f16.so) = round to that dtype there instead."""
        self.cfg = cfg
        self.W = {k: v.float() for k, v in weights.items()}
        self.emu = bool(emulate_bf16)
        self.emu_dtype = emulate_bf16 if isinstance(emulate_bf16, torch.dtype) else torch.bfloat16

    def r(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(self.emu_dtype).float() if self.emu else x

    # -- CLIP ViT (clip_encoder.py:31-53 -> HF CLIPVisionModel; HF5 modeling_clip.py:148-385) -------
    def vit_forward(self, frames: torch.Tensor, return_all: bool = False):
        """frames [T,3,S,S] -> features [T, patches, vh] = hidden state after layer `layers_used`
        (hidden_states[-2] for select_layer=-2), CLS dropped (feature_select 'patch')."""
        c, W, r = self.cfg, self.W, self.r
        T = frames.shape[0]
        P, g, vh = c.vision_patch_size, c.vision_grid, c.vision_hidden_size
        x = r(frames.float())
        # patch-embed conv, stride = kernel = P, no bias (modeling_clip.py:148-154,208-211) as a GEMM
        pw = W[VIS + "embeddings.patch_embedding.weight"].reshape(vh, 3 * P * P)
        patches = x.reshape(T, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(T, g * g, 3 * P * P)
        pe = r(patches @ pw.t())
        cls = W[VIS + "embeddings.class_embedding"].reshape(1, 1, vh).expand(T, 1, vh)
        x = torch.cat([cls, pe], dim=1) + W[VIS + "embeddings.position_embedding.weight"][None]
        x = r(x)
        x = r(self._ln(x, W[VIS + "pre_layrnorm.weight"], W[VIS + "pre_layrnorm.bias"], c.vision_layer_norm_eps))
        hs = [x]
        nh, hd = c.vision_num_heads, c.vision_head_dim
        for l in range(c.vision_layers_used):
            p = VIS + f"encoder.layers.{l}."
            h = r(self._ln(x, W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], c.vision_layer_norm_eps))
            q = r(h @ W[p + "self_attn.q_proj.weight"].t() + W[p + "self_attn.q_proj.bias"])
            k = r(h @ W[p + "self_attn.k_proj.weight"].t() + W[p + "self_attn.k_proj.bias"])
            v = r(h @ W[p + "self_attn.v_proj.weight"].t() + W[p + "self_attn.v_proj.bias"])
            N = q.shape[1]
            q = q.view(T, N, nh, hd).transpose(1, 2)
            k = k.view(T, N, nh, hd).transpose(1, 2)
            v = v.view(T, N, nh, hd).transpose(1, 2)
            s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)          # modeling_clip.py:261-277
            pattn = torch.softmax(s, dim=-1)
            o = r((pattn @ v).transpose(1, 2).reshape(T, N, vh))
            x = r(x + r(o @ W[p + "self_attn.out_proj.weight"].t() + W[p + "self_attn.out_proj.bias"]))
            h = r(self._ln(x, W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], c.vision_layer_norm_eps))
            u = h @ W[p + "mlp.fc1.weight"].t() + W[p + "mlp.fc1.bias"]
            u = r(u * torch.sigmoid(1.702 * u))                    # quick_gelu
            x = r(x + r(u @ W[p + "mlp.fc2.weight"].t() + W[p + "mlp.fc2.bias"]))
            hs.append(x)
        feats = x[:, 1:]
        return (feats, hs) if return_all else feats

    @staticmethod
    def _ln(x, w, b, eps):
        mu = x.mean(-1, keepdim=True)
        var = ((x - mu) ** 2).mean(-1, keepdim=True)
        return (x - mu) * torch.rsqrt(var + eps) * w + b

    # -- SpatialSlotPool (multimodal_projector/builder.py:289-359,411-467) --------------------------
    def slot_pool(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [T, n, d] -> [T, num_slots, H]."""
        c, W, r = self.cfg, self.W, self.r
        T, n, d = feats.shape
        x = self._ln(feats, W["model.mm_projector.ln_vision.weight"], W["model.mm_projector.ln_vision.bias"],
                     c.slot_ln_eps)                                             # builder.py:451 (timm LN eps 1e-6)
        inv_freq = 1.0 / (c.slot_rope_base ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))  # :296
        t = torch.arange(n, dtype=torch.int64).float()
        freqs = torch.outer(t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)                                  # :307-309
        cos, sin = emb.cos(), emb.sin()
        x1, x2 = x[..., : d // 2], x[..., d // 2:]
        x = x * cos + torch.cat((-x2, x1), dim=-1) * sin                         # :335-359, position = patch idx
        logits = x @ W["model.mm_projector.slots"]                               # [T, n, s]  :457
        p = torch.softmax(logits, dim=1)                                         # over patches :458
        res = r(torch.einsum("tnd,tns->tsd", x, p))                              # :462
        return r(res @ W["model.mm_projector.readout.weight"].t())               # :467

    # -- STCConnector (multimodal_projector/builder.py:138-249), legacy trace.infer() path -----------
    # stc_connector() as a whole is pinned by tests/golden/stc_connector.npz (the reference's own forward with this block substituted for timm's).
    # PARITY UNPINNED for the block itself: timm.models.regnet.RegStage is not importable in the build container, so the block
    # below follows timm 0.6.x `Bottleneck` from memory (1x1 conv -> LN2d -> SiLU; depthwise 3x3 -> LN2d -> SiLU;
    # SE with rd = round(in_chs/4), SiLU, sigmoid gate; 1x1 conv -> LN2d; + shortcut (1x1 conv + LN2d when the
    # channel count changes); SiLU).  It pins the HIP path against THIS restatement only.
    def _ln2d(self, x, w, b):
        # LayerNorm2d: normalise over channels at every pixel (eps 1e-6); x [N,C,H,W]
        xp = x.permute(0, 2, 3, 1)
        return self._ln(xp, w, b, 1e-6).permute(0, 3, 1, 2)

    def _reg_block(self, x, q):
        import torch.nn.functional as F
        W, r = self.W, self.r
        silu = torch.nn.functional.silu
        sc = x
        y = r(F.conv2d(x, W[q + "conv1.conv.weight"]))
        y = r(silu(r(self._ln2d(y, W[q + "conv1.bn.weight"], W[q + "conv1.bn.bias"]))))
        y = r(F.conv2d(y, W[q + "conv2.conv.weight"], padding=1, groups=y.shape[1]))
        y = r(silu(r(self._ln2d(y, W[q + "conv2.bn.weight"], W[q + "conv2.bn.bias"]))))
        se = r(y.mean((2, 3), keepdim=True))
        se = r(silu(r(F.conv2d(se, W[q + "se.fc1.weight"])) + W[q + "se.fc1.bias"].view(1, -1, 1, 1)))
        se = r(torch.sigmoid(r(F.conv2d(se, W[q + "se.fc2.weight"])) + W[q + "se.fc2.bias"].view(1, -1, 1, 1)))
        y = r(y * se)
        y = r(F.conv2d(y, W[q + "conv3.conv.weight"]))
        y = r(self._ln2d(y, W[q + "conv3.bn.weight"], W[q + "conv3.bn.bias"]))
        if (q + "downsample.conv.weight") in W:
            sc = r(F.conv2d(sc, W[q + "downsample.conv.weight"]))
            sc = r(self._ln2d(sc, W[q + "downsample.bn.weight"], W[q + "downsample.bn.bias"]))
        return r(silu(y + sc))

    def stc_connector(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [T, n, d] (one video) -> [(T//2+1) * (h//2+1)^2, H]   (builder.py:208-231)."""
        import torch.nn.functional as F
        W, r = self.W, self.r
        P = "model.mm_projector."
        T, n, d = feats.shape
        h = int(n ** 0.5)
        x = feats.float().permute(0, 2, 1).reshape(T, d, h, h)                        # (b t) d h w
        for b in range(4):
            x = self._reg_block(x, f"{P}s1.b{b + 1}.")
        x = x.permute(1, 0, 2, 3)[None]                                               # b d t h w
        x = r(F.conv3d(x, W[P + "sampler.0.weight"], W[P + "sampler.0.bias"], stride=2, padding=1))
        x = r(torch.nn.functional.silu(x))
        nt = x.shape[2]
        x = x[0].permute(1, 0, 2, 3)                                                  # (b t) d h w
        for b in range(4):
            x = self._reg_block(x, f"{P}s2.b{b + 1}.")
        x = x.permute(0, 2, 3, 1).reshape(nt * x.shape[2] * x.shape[3], -1)           # (t h w) d
        x = r(torch.nn.functional.gelu(r(x @ W[P + "readout.0.weight"].t() + W[P + "readout.0.bias"])))
        return r(x @ W[P + "readout.2.weight"].t() + W[P + "readout.2.bias"])

    # -- encode_images_or_videos (trace_arch.py:218-266) -------------------------------------------
    def encode_video(self, frames: torch.Tensor, timestamps: Sequence[Sequence[float]]) -> torch.Tensor:
        """frames [T,3,S,S], timestamps [[t]]*T -> [T*(slots+6), H]: per frame slots then time tokens."""
        c, W = self.cfg, self.W
        slots = self.slot_pool(self.vit_forward(frames))                         # [T, 8, H]
        tok = [time_encode(t) for t in timestamps]                               # encode_time, trace_arch.py:271-288
        assert all(len(x) == len(tok[0]) for x in tok), "time token length differs across frames (trace_arch.py:285)"
        ids = torch.tensor([x[:-1] for x in tok], dtype=torch.long)              # drop <sync> :243
        tfeat = W["model.time_tower.embed_tokens.weight"][ids]                   # [T, 6, H]
        return torch.cat([slots, tfeat], dim=1).reshape(-1, c.hidden_size)       # :256-258

    # -- prepare_inputs_labels_for_multimodal: prefill splice (trace_arch.py:377-456) --------------
    def splice(self, input_ids: torch.Tensor, video_feats: torch.Tensor,
               times: Sequence[Sequence[float]] = (), scores: Sequence[Sequence[float]] = ()) -> torch.Tensor:
        c, W = self.cfg, self.W
        ids = input_ids.long()
        vpos = torch.where((ids == -201) | (ids == -200))[0]
        assert len(vpos) == 1, "only have one video inputs!"                     # trace_arch.py:411
        vp = int(vpos[0])
        new_ids = torch.cat([ids[:vp], torch.full((video_feats.shape[0],), -201, dtype=torch.long), ids[vp + 1:]])
        emb = W["model.embed_tokens.weight"][new_ids.clamp(min=0)].clone()        # :417-418
        emb[new_ids == -201] = video_feats
        t_ids = [i for t in times for i in time_encode(t)]
        s_ids = [i for s in scores for i in score_encode(s)]
        if (new_ids == -203).any() or t_ids:
            emb[new_ids == -203] = W["model.time_tower.embed_tokens.weight"][torch.tensor(t_ids, dtype=torch.long)]
        if (new_ids == -204).any() or s_ids:
            emb[new_ids == -204] = W["model.score_tower.embed_tokens.weight"][torch.tensor(s_ids, dtype=torch.long)]
        emb[new_ids == -205] = W["model.sync_tower.embed_tokens.weight"][0]       # sync_encoder.py:15-18
        return self.r(emb)

    # -- decode-branch embedding (trace_arch.py:345-375) -------------------------------------------
    def decode_embed(self, token: int) -> torch.Tensor:
        c, W = self.cfg, self.W
        V, Tv = c.vocab_size, c.time_vocab_size
        token = int(token)
        if token == V:
            return W["model.sync_tower.embed_tokens.weight"][0]
        if V + 1 <= token < V + Tv + 1:
            return W["model.time_tower.embed_tokens.weight"][token - V - 1]
        if token >= V + Tv + 1:
            return W["model.score_tower.embed_tokens.weight"][token - V - Tv - 1]
        return W["model.embed_tokens.weight"][token % V]

    # -- Mistral decoder (trace_mistral.py:178-188 -> HF MistralModel; HF5 modeling_mistral.py:35-245)
    def _rms(self, x, w):
        var = x.pow(2).mean(-1, keepdim=True)
        return x * torch.rsqrt(var + self.cfg.rms_norm_eps) * w

    def _rope(self, x, pos):
        """x [L, heads, hd]; rotate-half RoPE (modeling_mistral.py:52-82)."""
        hd = x.shape[-1]
        inv = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
        f = torch.outer(pos.float(), inv)
        emb = torch.cat((f, f), dim=-1)
        cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
        x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
        return x * cos + torch.cat((-x2, x1), dim=-1) * sin

    def llm_forward(self, embeds: torch.Tensor, kv: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
                    return_layers: bool = False):
        """embeds [L,H] appended after the cached context; returns (hidden [L,H] after final norm, kv)."""
        c, W, r = self.cfg, self.W, self.r
        L = embeds.shape[0]
        nq, nkv, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        past = 0 if kv is None else kv[0][0].shape[0]
        pos = torch.arange(past, past + L)
        new_kv = []
        x = embeds.float()
        layers = []
        for l in range(c.num_hidden_layers):
            p = f"model.layers.{l}."
            h = r(self._rms(x, W[p + "input_layernorm.weight"]))
            q = r(h @ W[p + "self_attn.q_proj.weight"].t()).view(L, nq, hd)
            k = r(h @ W[p + "self_attn.k_proj.weight"].t()).view(L, nkv, hd)
            v = r(h @ W[p + "self_attn.v_proj.weight"].t()).view(L, nkv, hd)
            q, k = r(self._rope(q, pos)), r(self._rope(k, pos))
            if kv is not None:
                k = torch.cat([kv[l][0], k], dim=0)
                v = torch.cat([kv[l][1], v], dim=0)
            new_kv.append((k, v))
            ctx = k.shape[0]
            kk = k.repeat_interleave(nq // nkv, dim=1)                   # repeat_kv, modeling_mistral.py:85-94
            vv = v.repeat_interleave(nq // nkv, dim=1)
            s = torch.einsum("lhd,chd->hlc", q, kk) * (hd ** -0.5)
            mask = torch.arange(ctx)[None, :] > (pos[:, None])             # causal
            s = s.masked_fill(mask[None], float("-inf"))
            pa = torch.softmax(s, dim=-1)
            o = r(torch.einsum("hlc,chd->lhd", pa, vv).reshape(L, nq * hd))
            x = r(x + r(o @ W[p + "self_attn.o_proj.weight"].t()))
            h = r(self._rms(x, W[p + "post_attention_layernorm.weight"]))
            g = h @ W[p + "mlp.gate_proj.weight"].t()
            u = h @ W[p + "mlp.up_proj.weight"].t()
            a = r(torch.nn.functional.silu(g) * u)
            x = r(x + r(a @ W[p + "mlp.down_proj.weight"].t()))
            layers.append(x)
        hidden = r(self._rms(x, W["model.norm.weight"]))
        if return_layers:
            return hidden, new_kv, layers
        return hidden, new_kv

    # -- heads + mask (trace_mistral.py:190-200,244-252) -------------------------------------------
    def logits(self, hidden: torch.Tensor, head: Optional[int]) -> torch.Tensor:
        """hidden [..., H] -> fp32 logits [..., V+1+Tv+Sv] with -inf outside the active head."""
        W = self.W
        lg = torch.cat([hidden @ W["lm_head.weight"].t(), hidden @ W["sync_head.weight"].t(),
                        hidden @ W["time_head.weight"].t(), hidden @ W["score_head.weight"].t()], dim=-1)
        if head is not None:
            lo, hi = head_range(self.cfg, head)
            lg = lg.clone()
            lg[..., :lo] = float("-inf")
            lg[..., hi:] = float("-inf")
        return lg

    # -- generate: greedy with head switching (trace_mistral.py:268-347 + HF greedy) ---------------
    def generate(self, input_ids: torch.Tensor, frames: torch.Tensor, timestamps, head: int = 1,
                 max_new_tokens: int = 32, eos_token_id: Optional[int] = None,
                 forced_ids: Optional[Sequence[int]] = None, return_logits: bool = False):
        """Returns new token ids (global vocabulary) as the reference's generate does (new tokens
        only, trace_mistral.py:309-314).  `forced_ids` = teacher forcing: feed these instead of the
        argmax (the argmax at every step is still returned)."""
        feats = self.encode_video(frames, timestamps)
        emb = self.splice(input_ids, feats)
        hidden, kv = self.llm_forward(emb)
        out, all_logits = [], []
        h_last = hidden[-1]
        for step in range(max_new_tokens):
            lg = self.logits(h_last, head)
            tok = int(torch.argmax(lg))
            out.append(tok)
            if return_logits:
                all_logits.append(lg)
            if eos_token_id is not None and tok == eos_token_id:
                break
            if step == max_new_tokens - 1:
                break
            feed = tok if forced_ids is None else int(forced_ids[step])
            head = swap_head(self.cfg, feed, head)                       # prepare_inputs_for_generation :336-344
            e = self.r(self.decode_embed(feed))[None]
            hidden, kv = self.llm_forward(e, kv)
            h_last = hidden[-1]
        if return_logits:
            return out, torch.stack(all_logits)
        return out
