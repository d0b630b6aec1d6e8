"""trace_amd — MI355X-native implementation of the TRACE video-grounding inference hot path.

Public surface mirrors the reference package (`trace`): trace_amd.model.builder.load_pretrained_model,
trace_amd.mm_utils, trace_amd.conversation, trace_amd.constants, plus the legacy `model_init / infer` helpers of
trace/__init__.py.  `trace_amd.compat.install()` registers the reference's import names (`Trace.trace.*`) so its
unchanged drivers resolve to this package."""
from functools import partial

__all__ = ["model_init", "infer"]


def model_init(model_path=None, **kw):
    """trace/__init__.py:13-21."""
    from .constants import NUM_FRAMES
    from .mm_utils import get_model_name_from_path, process_video
    from .model.builder import load_pretrained_model
    if model_path is None:
        raise ValueError("model_path is required (no network access to fetch a default checkpoint)")
    tokenizer, model, processor, _ = load_pretrained_model(model_path, None, get_model_name_from_path(model_path), **kw)
    nf = model.config.num_frames if hasattr(model.config, "num_frames") else NUM_FRAMES
    return model, partial(process_video, aspect_ratio=None, processor=processor, num_frames=nf), tokenizer


def infer(model, video, instruct, tokenizer, do_sample=False, video_timestamps=None, max_new_tokens=128):
    """trace/__init__.py:23-75, routed through the TRACE path (per-frame timestamps are required by it; when the
    caller has none, frames are stamped 1 s apart as the reference's 1-fps sampler would)."""
    import torch
    from .constants import DEFAULT_MMODAL_TOKEN
    from .conversation import conv_templates
    from .mm_utils import tokenizer_MMODAL_token_all
    conv = conv_templates["llama_2"].copy()
    conv.append_message(conv.roles[0], DEFAULT_MMODAL_TOKEN["VIDEO"] + "\n" + instruct)
    conv.append_message(conv.roles[1], None)
    if getattr(model.config, "mm_projector_type", "") == "stc_connector":
        # the reference's legacy flow verbatim: STC connector, no <sync>, no timestamps, text head (trace/__init__.py:46-71)
        from .constants import MMODAL_TOKEN_INDEX
        from .mm_utils import tokenizer_MMODAL_token
        ids = tokenizer_MMODAL_token(conv.get_prompt(), tokenizer, MMODAL_TOKEN_INDEX["VIDEO"], return_tensors="pt").unsqueeze(0)
        with torch.inference_mode():
            return model.generate(ids, images_or_videos=[video], modal_list=["video"], do_sample=do_sample,
                                  temperature=0.2 if do_sample else 0.0, max_new_tokens=max_new_tokens, use_cache=True)
    prompt = conv.get_prompt() + "<sync>"
    ids = tokenizer_MMODAL_token_all(prompt, tokenizer, return_tensors="pt").unsqueeze(0)
    ts = video_timestamps or [[float(i)] for i in range(video.shape[0])]
    with torch.inference_mode():
        out = model.generate(ids, images_or_videos=[video], modal_list=["video"], do_sample=do_sample,
                             temperature=0.2 if do_sample else 0.0, max_new_tokens=max_new_tokens, use_cache=True,
                             video_timestamps=[ts], heads=[1])
    return out
