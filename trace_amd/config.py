"""Model geometry for the TRACE inference hot path.

The reference never states these numbers in-tree; it reads them from the
checkpoint's ``config.json`` (keys consumed: reference
trace/model/language_model/trace_mistral.py:84-96, trace/model/trace_arch.py:34,219,
trace/model/multimodal_encoder/clip_encoder.py:15-16,
trace/model/multimodal_projector/builder.py:95).  ``TraceConfig`` carries the same
keys under the same names so ``model.config.<key>`` keeps working for the drivers
(``config.image_aspect_ratio``, ``config.num_frames`` at trace/eval/evaluate.py:315).
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Any, Dict


@dataclasses.dataclass
class TraceConfig:
    # --- Mistral decoder (public Mistral-7B geometry by default) ---
    vocab_size: int = 32000
    hidden_size: int = 4096            # hard requirement: towers are Embedding(13, 4096) (trace_arch.py:38-40)
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    rms_norm_eps: float = 1e-5
    rope_theta: float = 1000000.0
    max_position_embeddings: int = 4096
    time_vocab_size: int = 13
    score_vocab_size: int = 13
    # --- CLIP ViT-L/14-336 vision tower ---
    mm_vision_tower: str = "openai/clip-vit-large-patch14-336"
    mm_vision_select_layer: int = -2
    mm_vision_select_feature: str = "patch"
    vision_hidden_size: int = 1024
    vision_intermediate_size: int = 4096
    vision_num_layers: int = 24
    vision_num_heads: int = 16
    vision_image_size: int = 336
    vision_patch_size: int = 14
    vision_layer_norm_eps: float = 1e-5
    # --- connector ---
    mm_projector_type: str = "spatial_slot"
    mm_hidden_size: int = 1024
    num_slots: int = 8
    slot_ln_eps: float = 1e-6          # timm LayerNorm default (projector/builder.py:419)
    slot_rope_base: float = 10000.0    # SlotRotaryEmbedding (projector/builder.py:291)
    # --- driver-visible knobs ---
    num_frames: int = 128
    image_aspect_ratio: str = "pad"
    mm_use_im_start_end: bool = False
    mm_use_im_patch_token: bool = False
    max_sequence_length: int = 4096
    bos_token_id: int = 1
    eos_token_id: int = 2
    model_type: str = "trace_mistral"

    # derived -----------------------------------------------------------------
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def vision_head_dim(self) -> int:
        return self.vision_hidden_size // self.vision_num_heads

    @property
    def vision_grid(self) -> int:
        return self.vision_image_size // self.vision_patch_size

    @property
    def vision_patches(self) -> int:
        return self.vision_grid * self.vision_grid

    @property
    def vision_tokens(self) -> int:
        return self.vision_patches + 1

    @property
    def vision_layers_used(self) -> int:
        """Encoder layers that must run for hidden_states[select_layer].

        HF returns hidden_states = [embeddings(after pre_layrnorm), layer1_out, ..., layerN_out];
        index -2 is the output of layer N-1 (clip_encoder.py:31-39)."""
        n = self.vision_num_layers
        sel = self.mm_vision_select_layer
        idx = sel if sel >= 0 else n + 1 + sel
        return idx

    @property
    def time_tokens_per_frame(self) -> int:
        return 6  # format(t, '0>6.1f') -> 6 chars; trailing <sync> dropped (trace_arch.py:243)

    @property
    def tokens_per_frame(self) -> int:
        return self.num_slots + self.time_tokens_per_frame

    @property
    def total_vocab(self) -> int:
        return self.vocab_size + 1 + self.time_vocab_size + self.score_vocab_size

    # (de)serialisation --------------------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "TraceConfig":
        names = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})

    @classmethod
    def from_pretrained(cls, path: str) -> "TraceConfig":
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        vis = d.get("vision_config") or {}
        alias = {
            "hidden_size": "vision_hidden_size", "intermediate_size": "vision_intermediate_size",
            "num_hidden_layers": "vision_num_layers", "num_attention_heads": "vision_num_heads",
            "image_size": "vision_image_size", "patch_size": "vision_patch_size",
            "layer_norm_eps": "vision_layer_norm_eps",
        }
        for k, v in vis.items():
            if k in alias:
                d.setdefault(alias[k], v)
        # a real TRACE config.json names the CLIP tower (`mm_vision_tower`) instead of spelling out its geometry: read it
        # from that directory's config.json when it is a local path (CLIPVisionConfig.from_pretrained, clip_encoder.py:21)
        tower_cfg = os.path.join(str(d.get("mm_vision_tower", "")), "config.json")
        if os.path.isfile(tower_cfg):
            with open(tower_cfg) as f:
                c = json.load(f)
            for k, v in (c.get("vision_config") or c).items():
                if k in alias:
                    d.setdefault(alias[k], v)
        return cls.from_dict(d)

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=1)


def trace_7b(num_frames: int = 128) -> TraceConfig:
    """BASELINE.json config 2: TRACE-7B, CLIP-ViT-L/14-336 + Mistral-7B."""
    return TraceConfig(num_frames=num_frames)


def tiny(num_frames: int = 4) -> TraceConfig:
    """Smallest geometry the reference can be instantiated with (hidden 4096 is
    forced by trace_arch.py:38-40) that still satisfies the HIP kernels' tile
    constraints (head_dim 128 / vision head_dim 64, dims multiples of 128)."""
    return TraceConfig(
        vocab_size=320, intermediate_size=256, num_hidden_layers=2,
        vision_hidden_size=128, vision_intermediate_size=256, vision_num_layers=3,
        vision_num_heads=2, vision_image_size=56, vision_patch_size=14,
        mm_hidden_size=128, num_frames=num_frames, mm_vision_tower="tiny-clip",
    )
