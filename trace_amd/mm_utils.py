"""Host-side harness of the path with the reference's names and return contracts (trace/mm_utils.py):
frame sampling + timestamps, pad-to-square, CLIP preprocessing call, prompt -> ids with modal placeholders.

Integer behaviour (sampled indices, timestamps, placeholder interleave, padding colour) is pinned by fixtures captured
from the reference (tests/golden/host_functions.json).  Container decoding needs decord / imageio like the reference;
when they are absent, `process_video` also accepts already-decoded frames (uint8 array / tensor / list of PIL images,
or a .npy file) plus an fps, so the drivers' call shape keeps working."""
from __future__ import annotations

import math
import os
import random
import re
from typing import List, Optional, Sequence

import numpy as np
import torch
from PIL import Image

from .constants import (DEFAULT_MMODAL_TOKEN, IMAGE_TOKEN_INDEX, MAX_FRAMES, MMODAL_INDEX_TOKEN, MMODAL_TOKEN_INDEX,
                        NUM_FRAMES, NUM_FRAMES_PER_SECOND)


def expand2square(pil_img, background_color):
    """mm_utils.py:259-270: paste onto a square canvas of the longer side, centred."""
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
    return canvas


def create_photo_grid(arr, rows=None, cols=None):
    """mm_utils.py:303-355: tile T equally sized frames row by row into one [rows*h, cols*w, c] image (zeros where the
    grid has spare cells).  Missing rows/cols default to a near-square layout."""
    if isinstance(arr, (list, tuple)):
        if not len(arr) or not isinstance(arr[0], (Image.Image, np.ndarray)):
            raise ValueError("Invalid input type. Expected list of Images or numpy arrays.")
        arr = np.stack([np.asarray(a) for a in arr])
    t, h, w, c = arr.shape
    if rows is None and cols is None:
        rows = math.ceil(math.sqrt(t))
    if cols is None:
        cols = math.ceil(t / rows)
    if rows is None:
        rows = math.ceil(t / cols)
    if rows * cols < t:
        raise ValueError(f"Not enough grid cells ({rows}x{cols}) to hold all images ({t}).")
    grid = np.zeros((rows * h, cols * w, c), dtype=arr.dtype)
    for i in range(t):
        y, x = divmod(i, cols)
        grid[y * h:(y + 1) * h, x * w:(x + 1) * w] = arr[i]
    return grid


def frame_sample(duration: int, num_frames: int = NUM_FRAMES, mode: str = "uniform", local_fps: Optional[float] = None):
    """mm_utils.py:380-400 (closure inside process_video in the reference)."""
    if mode == "uniform":
        return np.linspace(0, duration - 1, num_frames, dtype=int)
    if mode == "fps":
        assert local_fps is not None
        seg = min(local_fps // NUM_FRAMES_PER_SECOND, duration)
        return np.arange(seg // 2, duration, seg, dtype=int)
    if mode == "rand":
        edges = np.linspace(0, duration - 1, num=num_frames + 1).astype(int)
        return [random.choice(np.linspace(a, b - 1, num=6).astype(int)[:-1]) for a, b in zip(edges[:-1], edges[1:])]
    raise ImportError(f"Unsupported frame sampling mode: {mode}")


def sample_indices_and_timestamps(duration: int, fps: float, num_frames: int, mode: str = "uniform"):
    """Index selection + timestamp arithmetic of process_video (mm_utils.py:428-437): uniform linspace, cap at
    MAX_FRAMES by re-sampling, timestamps = index / fps."""
    idx = frame_sample(duration, num_frames, mode, fps)
    if len(idx) > MAX_FRAMES:
        idx = np.linspace(0, duration - 1, MAX_FRAMES, dtype=int)
    return idx, [[float(i / fps)] for i in idx]


class _GifFrames(list):
    """marker: the reference's GIF branch keeps each sampled index ONCE, in increasing order (`index in frame_id_list`)"""


def _to_frames(video, fps):
    if isinstance(video, str):
        if video.endswith(".npy"):
            return np.load(video), float(fps or 1.0)
        if video.lower().endswith(".gif"):
            # the reference reads GIFs with imageio at a nominal 10 fps (mm_utils.py:404-413); Pillow decodes them as well
            with Image.open(video) as im:
                frames = []
                for i in range(getattr(im, "n_frames", 1)):
                    im.seek(i)
                    frames.append(np.array(im.convert("RGB")))
            return _GifFrames(frames), 10.0
        if os.path.isdir(video):
            names = sorted(n for n in os.listdir(video) if n.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
            return [np.array(Image.open(os.path.join(video, n)).convert("RGB")) for n in names], float(fps or 1.0)
        from .video_io import open_container
        vr = open_container(video)                    # .y4m, Motion-JPEG / uncompressed .avi: read here, decord's VideoReader surface
        if vr is not None:
            return vr, float(vr.get_avg_fps())
        try:
            from decord import VideoReader, cpu       # same reader as the reference (mm_utils.py:421)
        except ImportError as e:
            raise ImportError("decoding this video container needs `decord` (as in the reference); without it: a .y4m file (`ffmpeg -i clip.mp4 clip.y4m`), "
                              "a Motion-JPEG .avi, a .gif, a .npy file, a directory of images, or decoded frames") from e
        vr = VideoReader(uri=video, ctx=cpu(0))
        return vr, float(vr.get_avg_fps())
    if isinstance(video, torch.Tensor):
        video = video.numpy()
    return video, float(fps or 1.0)


def process_video(video_path, processor, aspect_ratio="pad", num_frames=NUM_FRAMES, image_grid=False,
                  sample_scheme="uniform", fps: Optional[float] = None, engine=None):
    """-> (FloatTensor[T,3,S,S], [[t_seconds]] * T)   (mm_utils.py:379-471).
    `engine` (a TraceEngine, or a model carrying one as `.engine`): run the per-frame image work — expand2square, bicubic
    resize, centre crop, rescale, normalise — on the GPU (trace_preprocess_frames) and return a bf16 device tensor the
    model consumes directly; same values as the host path (the resize is Pillow's resampler bit for bit)."""
    src, local_fps = _to_frames(video_path, fps)
    duration = len(src)
    idx, video_timestamps = sample_indices_and_timestamps(duration, local_fps, num_frames, sample_scheme)
    if isinstance(src, _GifFrames):
        idx = sorted(set(int(i) for i in idx))
        video_timestamps = [[i / local_fps] for i in idx]
    if hasattr(src, "get_batch"):
        batch = src.get_batch(idx)
        data = batch.asnumpy() if hasattr(batch, "asnumpy") else batch.numpy()
        frames = [f for f in data]
    else:
        frames = [src[int(i)] for i in idx]
    # mm_utils.py:466-469 (error types kept; the reference checks after preprocessing, which has no side effects)
    if video_timestamps[-1][0] > 9999:
        raise ImportError("The video is too long!")
    if video_timestamps[0][0] < 0:
        raise ImportError("Timestamp can not be less than zero")
    if image_grid:                                                       # mm_utils.py:450-453: the photo grid goes first
        side = math.ceil(math.sqrt(num_frames))
        as_np = [np.asarray(f.convert("RGB") if isinstance(f, Image.Image) else f) for f in frames]
        frames = [create_photo_grid(as_np, side, side), *as_np]
    eng = getattr(engine, "engine", engine)
    if eng is not None and not image_grid:                               # (the grid image has a different size: host path)
        arr = np.stack([np.asarray(f.convert("RGB") if isinstance(f, Image.Image) else f) for f in frames])
        mean = getattr(processor, "image_mean", None) or eng.CLIP_MEAN
        std = getattr(processor, "image_std", None) or eng.CLIP_STD
        return eng.preprocess_frames(arr, pad=(aspect_ratio == "pad"), image_mean=mean, image_std=std), video_timestamps
    images = [f if isinstance(f, Image.Image) else Image.fromarray(np.asarray(f)) for f in frames]
    if aspect_ratio == "pad":
        bg = tuple(int(x * 255) for x in processor.image_mean)           # mm_utils.py:456-458
        images = [expand2square(im, bg) for im in images]
    video = processor.preprocess(images, return_tensors="pt")["pixel_values"]
    return video, video_timestamps


def process_image(image_path, processor, aspect_ratio="pad", num_frames=NUM_FRAMES, image_grid=False):
    image = image_path if isinstance(image_path, Image.Image) else Image.open(image_path).convert("RGB")
    images = [np.array(image)]
    if image_grid:                                                       # mm_utils.py:361-366
        side = math.ceil(math.sqrt(num_frames))
        images = [create_photo_grid(np.stack([images[0]] * num_frames), side, side), images[0]]
    images = [Image.fromarray(f) for f in images]
    if aspect_ratio == "pad":
        images = [expand2square(im, tuple(int(x * 255) for x in processor.image_mean)) for im in images]
    return processor.preprocess(images, return_tensors="pt")["pixel_values"]


def tokenizer_MMODAL_token_all(prompt, tokenizer, return_tensors=None):
    """mm_utils.py:519-554: split on the six modal tags, tokenize the text chunks, keep BOS only from chunk 0,
    interleave the placeholder ids (-200 .. -205)."""
    pattern = "|".join(map(re.escape, DEFAULT_MMODAL_TOKEN.values()))
    chunks = [tokenizer(c).input_ids for c in re.split(pattern, prompt)]
    seps = [MMODAL_TOKEN_INDEX[m[1:-1].upper()] for m in re.findall(pattern, prompt)]
    offset = 1 if (len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id) else 0
    ids = list(chunks[0])
    assert len(chunks) == len(seps) + 1 or len(chunks) == 0
    for x, s in zip(chunks[1:], seps):
        ids.append(s)
        ids.extend(x[offset:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def tokenizer_MMODAL_token(prompt, tokenizer, MMODAL_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """mm_utils.py:493-516: single-tag variant used by the legacy trace.infer API."""
    tag = f"<{MMODAL_INDEX_TOKEN[MMODAL_token_index].lower()}>"
    chunks = [tokenizer(c).input_ids for c in prompt.split(tag)]
    ids: List[int] = []
    offset = 0
    if len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    for i, x in enumerate(chunks):
        if i:
            ids.extend(([MMODAL_token_index] * (offset + 1))[offset:])
        ids.extend(x[offset:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def get_model_name_from_path(model_path):
    return "_".join(model_path.strip("/").split("/"))                    # mm_utils.py:558-564 (both branches equal)


class KeywordsStoppingCriteria:
    """mm_utils.py:567-600.  Callable(output_ids, scores) -> bool; the engine evaluates it between decode chunks."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids, scores=None, **kwargs) -> bool:
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        flat = output_ids[output_ids < self.tokenizer.vocab_size]
        for kid in self.keyword_ids:
            kid = kid.to(flat.device)
            if flat.numel() >= kid.numel() and (flat[-kid.shape[0]:] == kid).all():
                return True
        if offset > 0 and hasattr(self.tokenizer, "batch_decode"):
            text = self.tokenizer.batch_decode(flat[-offset:].view(1, -1), skip_special_tokens=True)[0]
            return any(kw in text for kw in self.keywords)
        return False

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))
