"""Deterministic synthetic weights with the reference's state-dict key names.

There are no checkpoints in the build or on the GPU box (no network), so parity
and throughput runs use random-init weights of the exact TRACE architecture.
Both sides of every parity test (the oracle, the imported reference in
oracle/make_goldens.py, and the HIP engine) regenerate the *same* tensors from
this module: each tensor's generator is seeded from a hash of its key name, so
no weight file ever has to be committed or shipped.

Key names are the ones `TraceMistralForCausalLM.state_dict()` produces in the
reference (observed by instantiating it):
  model.embed_tokens.weight, model.layers.N.*, model.norm.weight,
  model.vision_tower.vision_tower.vision_model.*  (transformers 4.40 layout),
  model.mm_projector.{slots,ln_vision.weight,ln_vision.bias,readout.weight},
  model.{time,score,sync}_tower.embed_tokens.weight, {lm,time,score,sync}_head.weight
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, List, Tuple

import torch

from .config import TraceConfig

BASE_SEED = 1234
VIS = "model.vision_tower.vision_tower.vision_model."


def weight_specs(cfg: TraceConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for every tensor on the inference path.
    kind: 'w' = N(0, 0.02^2); 'norm' = 1 + 0.1 N(0,1); 'bias' = 0.02 N(0,1); 'slots' = N(0,1)
    (slots ~ randn as in projector/builder.py:418)."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    hd, nq, nkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    vh, vi, P = cfg.vision_hidden_size, cfg.vision_intermediate_size, cfg.vision_patch_size
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    s.append(("model.embed_tokens.weight", (V, H), "w"))
    for l in range(cfg.num_hidden_layers):
        p = f"model.layers.{l}."
        s += [
            (p + "self_attn.q_proj.weight", (nq * hd, H), "w"),
            (p + "self_attn.k_proj.weight", (nkv * hd, H), "w"),
            (p + "self_attn.v_proj.weight", (nkv * hd, H), "w"),
            (p + "self_attn.o_proj.weight", (H, nq * hd), "w"),
            (p + "mlp.gate_proj.weight", (I, H), "w"),
            (p + "mlp.up_proj.weight", (I, H), "w"),
            (p + "mlp.down_proj.weight", (H, I), "w"),
            (p + "input_layernorm.weight", (H,), "norm"),
            (p + "post_attention_layernorm.weight", (H,), "norm"),
        ]
    s.append(("model.norm.weight", (H,), "norm"))
    # vision tower (all layers are synthesised, even the one select_layer=-2 never runs)
    s += [
        (VIS + "embeddings.class_embedding", (vh,), "w"),
        (VIS + "embeddings.patch_embedding.weight", (vh, 3, P, P), "w"),
        (VIS + "embeddings.position_embedding.weight", (cfg.vision_tokens, vh), "w"),
        (VIS + "pre_layrnorm.weight", (vh,), "norm"),
        (VIS + "pre_layrnorm.bias", (vh,), "bias"),
    ]
    for l in range(cfg.vision_num_layers):
        p = VIS + f"encoder.layers.{l}."
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s += [(p + f"self_attn.{proj}.weight", (vh, vh), "w"), (p + f"self_attn.{proj}.bias", (vh,), "bias")]
        s += [
            (p + "layer_norm1.weight", (vh,), "norm"), (p + "layer_norm1.bias", (vh,), "bias"),
            (p + "mlp.fc1.weight", (vi, vh), "w"), (p + "mlp.fc1.bias", (vi,), "bias"),
            (p + "mlp.fc2.weight", (vh, vi), "w"), (p + "mlp.fc2.bias", (vh,), "bias"),
            (p + "layer_norm2.weight", (vh,), "norm"), (p + "layer_norm2.bias", (vh,), "bias"),
        ]
    s += [(VIS + "post_layernorm.weight", (vh,), "norm"), (VIS + "post_layernorm.bias", (vh,), "bias")]
    if cfg.mm_projector_type == "stc_connector":
        s += stc_specs(cfg)
    else:
        s += [
            ("model.mm_projector.slots", (cfg.mm_hidden_size, cfg.num_slots), "slots"),
            ("model.mm_projector.ln_vision.weight", (cfg.mm_hidden_size,), "norm"),
            ("model.mm_projector.ln_vision.bias", (cfg.mm_hidden_size,), "bias"),
            ("model.mm_projector.readout.weight", (H, cfg.mm_hidden_size), "w"),
        ]
    s += [
        ("model.time_tower.embed_tokens.weight", (cfg.time_vocab_size, H), "w"),
        ("model.score_tower.embed_tokens.weight", (cfg.score_vocab_size, H), "w"),
        ("model.sync_tower.embed_tokens.weight", (1, H), "w"),
        ("lm_head.weight", (V, H), "w"),
        ("sync_head.weight", (1, H), "w"),
        ("time_head.weight", (cfg.time_vocab_size, H), "w"),
        ("score_head.weight", (cfg.score_vocab_size, H), "w"),
    ]
    return s


def stc_specs(cfg: TraceConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """STCConnector (reference multimodal_projector/builder.py:138-205): RegStage(depth 4, SiLU, LayerNorm2d) ->
    Conv3d k=s=2 p=1 + SiLU -> RegStage -> MLP(GELU).  Module/key names follow timm 0.6.x `regnet.Bottleneck`
    (conv1/conv2/se/conv3/downsample, each ConvNormAct = .conv + .bn) as recalled — timm is not installed here, so
    these names (like the STC arithmetic) are unpinned."""
    C, Cin = cfg.hidden_size, cfg.mm_hidden_size
    P = "model.mm_projector."
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    for stage, cin0 in (("s1", Cin), ("s2", C)):
        for b in range(4):
            cin = cin0 if b == 0 else C
            rd = int(round(cin * 0.25))
            q = f"{P}{stage}.b{b + 1}."
            s += [
                (q + "conv1.conv.weight", (C, cin, 1, 1), "w"), (q + "conv1.bn.weight", (C,), "norm"), (q + "conv1.bn.bias", (C,), "bias"),
                (q + "conv2.conv.weight", (C, 1, 3, 3), "dw"), (q + "conv2.bn.weight", (C,), "norm"), (q + "conv2.bn.bias", (C,), "bias"),
                (q + "se.fc1.weight", (rd, C, 1, 1), "w"), (q + "se.fc1.bias", (rd,), "bias"),
                (q + "se.fc2.weight", (C, rd, 1, 1), "w"), (q + "se.fc2.bias", (C,), "bias"),
                (q + "conv3.conv.weight", (C, C, 1, 1), "w"), (q + "conv3.bn.weight", (C,), "norm"), (q + "conv3.bn.bias", (C,), "bias"),
            ]
            if cin != C:
                s += [(q + "downsample.conv.weight", (C, cin, 1, 1), "w"), (q + "downsample.bn.weight", (C,), "norm"),
                      (q + "downsample.bn.bias", (C,), "bias")]
    s += [(P + "sampler.0.weight", (C, C, 2, 2, 2), "w3"), (P + "sampler.0.bias", (C,), "bias"),
          (P + "readout.0.weight", (C, C), "w"), (P + "readout.0.bias", (C,), "bias"),
          (P + "readout.2.weight", (C, C), "w"), (P + "readout.2.bias", (C,), "bias")]
    return s


def synth_tensor(name: str, shape: Tuple[int, ...], kind: str, dtype=torch.bfloat16,
                 base_seed: int = BASE_SEED, device: str = "cpu") -> torch.Tensor:
    """device="cpu" is the canonical stream (what the oracle, the goldens and the parity tests use);
    device="cuda" draws from the device generator instead (different values, same distribution) so a
    7B-parameter throughput run does not wait on a host RNG."""
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(name.encode()) + base_seed) & 0x7FFFFFFF)
    x = torch.randn(shape, generator=g, dtype=torch.float32, device=device)
    if kind == "w":
        x = x * 0.02
    elif kind == "norm":
        x = 1.0 + 0.1 * x
    elif kind == "bias":
        x = x * 0.02
    elif kind == "dw":
        x = x * 0.3            # 9-tap depthwise kernels: keep the activation scale O(1)
    elif kind == "w3":
        x = x * 0.006          # fan-in 8*C
    elif kind == "slots":
        # randn as in the reference, scaled: with unit-variance LN output over 1024 channels an
        # N(0,1) slot matrix gives logits of std 32 and a one-hot 576-way softmax whose value is
        # decided by bf16 rounding noise; 0.05 keeps logits at std ~1.6 so parity is meaningful.
        x = x * 0.05
    else:
        raise ValueError(kind)
    return x.to(dtype)


def iter_weights(cfg: TraceConfig, dtype=torch.bfloat16, base_seed: int = BASE_SEED, device: str = "cpu"
                 ) -> Iterator[Tuple[str, torch.Tensor]]:
    """Streams (name, tensor) one at a time so a 7B model never has to sit in host RAM twice."""
    for name, shape, kind in weight_specs(cfg):
        yield name, synth_tensor(name, shape, kind, dtype, base_seed, device)


def state_dict(cfg: TraceConfig, dtype=torch.bfloat16, base_seed: int = BASE_SEED) -> Dict[str, torch.Tensor]:
    return dict(iter_weights(cfg, dtype, base_seed))


def synth_frames(cfg: TraceConfig, video_idx: int = 0, num_frames: int | None = None,
                 dtype=torch.float32, device="cpu") -> torch.Tensor:
    """[T,3,S,S] N(0,1) frames: post-CLIP-normalisation statistics.  device="cpu" is the canonical
    stream (parity tests, goldens); a CUDA device draws from the device generator instead (benchmarks: 64 clips per
    rank would otherwise cost ~20 s of host RNG per rank)."""
    T = num_frames or cfg.num_frames
    g = torch.Generator(device=device)
    g.manual_seed(42 + video_idx)
    S = cfg.vision_image_size
    return torch.randn((T, 3, S, S), generator=g, dtype=torch.float32, device=device).to(dtype)


def synth_prompt_ids(cfg: TraceConfig, n_text: int = 176, video_pos: int = 150, seed: int = 7) -> torch.Tensor:
    """Prompt ids shaped like the DVC prompt (685 chars -> ~176 ids): BOS, VIDEO placeholder (-201)
    at `video_pos`, SYNC (-205) last, everything else uniform in [3, vocab)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ids = torch.randint(3, cfg.vocab_size, (n_text,), generator=g, dtype=torch.long)
    ids[0] = cfg.bos_token_id
    ids[min(video_pos, n_text - 2)] = -201
    ids[-1] = -205
    return ids
