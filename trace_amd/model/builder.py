"""`load_pretrained_model` with the reference's signature and return tuple (trace/model/builder.py:29-156):
    tokenizer, model, processor, context_len = load_pretrained_model(model_path, model_base, model_name, ...)

A model directory holds `config.json` (TraceConfig keys = the reference's config keys) and either HF-format weight
shards (`*.safetensors` or `pytorch_model*.bin`, reference state-dict names; both transformers CLIP key layouts are
accepted; when the shards carry no vision tower it is read from the local CLIP directory `mm_vision_tower` names) or
`"synthetic_weights": true` (random-init weights of the exact architecture — what the build container and the GPU box
use, since no checkpoint can be downloaded).  8-bit / 4-bit / LoRA branches of the reference are load-time conveniences
outside the accelerated path and raise NotImplementedError."""
from __future__ import annotations

import glob
import json
import os
from typing import Optional

import torch

from ..config import TraceConfig
from ..constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN
from .trace_mistral import TraceMistralForCausalLM


class ByteTokenizer:
    """Stand-in text tokenizer for synthetic checkpoints (no sentencepiece model offline): UTF-8 bytes shifted past
    the special ids.  Same call surface the drivers use (`__call__().input_ids`, `decode`, `batch_decode`, bos/eos/pad)."""
    bos_token_id, eos_token_id, pad_token_id, unk_token_id = 1, 2, 0, 0

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size

    def __len__(self):
        return self.vocab_size

    def __call__(self, text, **kw):
        from types import SimpleNamespace
        ids = [self.bos_token_id] + [3 + (b % (self.vocab_size - 3)) for b in text.encode("utf-8")]
        return SimpleNamespace(input_ids=ids)

    def decode(self, ids, skip_special_tokens=True, **kw):
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        if isinstance(ids, int):
            ids = [ids]
        bs = bytes([(int(i) - 3) % 256 for i in ids if int(i) >= 3 and int(i) < self.vocab_size])
        return bs.decode("utf-8", errors="replace")

    def batch_decode(self, batch, **kw):
        return [self.decode(x, **kw) for x in batch]

    def add_tokens(self, toks, special_tokens=False):
        return 0


def save_synthetic_checkpoint(path: str, cfg: TraceConfig) -> str:
    os.makedirs(path, exist_ok=True)
    d = cfg.to_dict()
    d["synthetic_weights"] = True
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(d, f, indent=1)
    return path


def _checkpoint_files(path: str):
    """HF weight shards of a model directory: `*.safetensors` if there are any, else `pytorch_model*.bin`."""
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    return files or sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))


def _iter_file(fn: str):
    if fn.endswith(".safetensors"):
        from safetensors import safe_open
        with safe_open(fn, framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)
    else:
        sd = torch.load(fn, map_location="cpu", weights_only=True)
        yield from sd.items()


# buffers older transformers versions serialise; not parameters of the path
_SKIP_SUFFIXES = ("rotary_emb.inv_freq", "embeddings.position_ids")


def _iter_checkpoint(model_path: str, cfg: TraceConfig):
    """(reference state-dict name, tensor) for every tensor of the path.  The reference builds the CLIP tower from
    `config.mm_vision_tower` first (clip_encoder.py:23-29, trace_arch.py:35) and then loads the TRACE checkpoint over it
    (builder.py:114), so tower weights come from the checkpoint when it has them and from the CLIP directory otherwise."""
    files = _checkpoint_files(model_path)
    if not files:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {model_path}")
    seen_vision = False
    for fn in files:
        for k, t in _iter_file(fn):
            if k.endswith(_SKIP_SUFFIXES):
                continue
            seen_vision |= ".vision_tower." in k
            yield k, t
    if seen_vision:
        return
    vdir = cfg.mm_vision_tower
    vfiles = _checkpoint_files(vdir) if os.path.isdir(vdir) else []
    if not vfiles:
        raise FileNotFoundError(f"the checkpoint holds no vision-tower weights and mm_vision_tower={vdir!r} is not a local "
                                "CLIP directory with weight files (there is no hub access: point it at a downloaded copy)")
    for fn in vfiles:
        for k, t in _iter_file(fn):
            # a CLIPModel checkpoint also carries text_model.* / visual_projection / logit_scale: not on the path
            if k.startswith("vision_model.") and not k.endswith(_SKIP_SUFFIXES):
                yield "model.vision_tower.vision_tower." + k, t


def _image_processor(cfg: TraceConfig, model_path: str):
    try:
        from transformers import CLIPImageProcessor
    except Exception:          # the processor is third-party in the reference too (HF CLIPImageProcessor)
        return None
    for cand in (model_path, getattr(cfg, "mm_vision_tower", "")):
        if cand and os.path.exists(os.path.join(cand, "preprocessor_config.json")):
            return CLIPImageProcessor.from_pretrained(cand)
    s = cfg.vision_image_size
    return CLIPImageProcessor(size={"shortest_edge": s}, crop_size={"height": s, "width": s})


def load_tokenizer(model_path: str, cfg: TraceConfig, **kwargs):
    """The reference's `AutoTokenizer.from_pretrained(model_path, use_fast=False)` (trace/model/builder.py:113,135) when the directory
    carries tokenizer files (sentencepiece `tokenizer.model` / `tokenizer.json`); the byte-level stand-in otherwise."""
    has_tok = any(os.path.exists(os.path.join(model_path, f)) for f in ("tokenizer.model", "tokenizer.json"))
    if has_tok:
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(model_path, use_fast=False, token=kwargs.get("token"))
    import warnings
    warnings.warn(f"no tokenizer files under {model_path}: falling back to the byte-level stand-in tokenizer")
    return ByteTokenizer(cfg.vocab_size)


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", use_flash_attn=False, max_batch: int = 1, max_new_tokens: int = 1024, **kwargs):
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes quantised loading is outside the MI355X path")
    if model_base is not None or "lora" in model_name.lower():
        raise NotImplementedError("LoRA merge-at-load is outside the accelerated path; merge the adapter offline")
    from ..engine import TraceEngine      # fails loudly without the HIP library / a GPU
    cfg = TraceConfig.from_pretrained(model_path)
    raw = json.load(open(os.path.join(model_path, "config.json")))
    dev_index = 0
    if isinstance(device, str) and ":" in device:
        dev_index = int(device.split(":")[1])
    elif isinstance(device, torch.device) and device.index is not None:
        dev_index = device.index
    T = cfg.num_frames
    if cfg.mm_projector_type == "stc_connector":
        vis = (T // 2 + 1) * (cfg.vision_grid // 2 + 1) ** 2
    else:
        vis = T * cfg.tokens_per_frame
    max_ctx = min(cfg.max_position_embeddings, vis + 1024 + max_new_tokens)
    # the reference loads with torch_dtype=torch.float16 (trace/model/builder.py:50); north_star's configs say bf16, which stays the default here:
    # torch_dtype=torch.float16 selects the fp16 library (libtrace_hip_f16.so)
    dtype = kwargs.get("torch_dtype") or torch.bfloat16
    eng = TraceEngine(cfg, device=dev_index, max_batch=max_batch, max_ctx=max_ctx, max_frames=max(T, 1),
                      max_new_tokens=max_new_tokens, dtype=dtype)
    if raw.get("synthetic_weights"):
        from .. import synth
        small = cfg.hidden_size * cfg.num_hidden_layers < 4096 * 8
        eng.load_weights(synth.iter_weights(cfg, dtype=dtype, device="cpu" if small else f"cuda:{dev_index}"))
        has_tok = any(os.path.exists(os.path.join(model_path, f)) for f in ("tokenizer.model", "tokenizer.json"))
        tokenizer = load_tokenizer(model_path, cfg, **kwargs) if has_tok else ByteTokenizer(cfg.vocab_size)
    else:
        eng.load_weights(_iter_checkpoint(model_path, cfg))
        tokenizer = load_tokenizer(model_path, cfg, **kwargs)
    processor = _image_processor(cfg, model_path)
    model = TraceMistralForCausalLM(cfg, eng, processor)
    # builder.py:135-149: optional extra tokens must not change the embedding table the engine already holds
    if getattr(cfg, "mm_use_im_patch_token", False):
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if getattr(cfg, "mm_use_im_start_end", False):
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    context_len = getattr(cfg, "max_sequence_length", None) or 2048           # builder.py:151-154
    return tokenizer, model, processor, context_len
