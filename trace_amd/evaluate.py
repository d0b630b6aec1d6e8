"""Batched, data-parallel form of the reference's evaluation loop (trace/eval/evaluate.py:298-417; the same per-video code
is repeated in scripts/inference/inference.py:27-128 and the mvbench / videomme drivers).

The reference handles one video at a time: process_video -> llama_2 prompt + "<sync>" -> tokenizer_MMODAL_token_all ->
model.generate(heads=[1]) -> split the id stream into timestamps / scores / captions.  Videos are independent, so here
they are (1) sharded round-robin over the ranks of a torchrun job, (2) preprocessed on the GPU, (3) decoded `batch_size`
at a time (the engine pairs equal-length prompts in prefill and streams the weights once per decode step for the whole
batch), (4) gathered with ONE all-gather of packed ids per job, and parsed with the drivers' own logic on every rank.
The result list is in input order and identical to processing the videos one by one.

    python -m torch.distributed.run --nproc-per-node 8 -m trace_amd.evaluate --model PATH --items items.json --out out.json

`items.json`: [{"id": ..., "video": "frames.npy" (uint8 [n, H, W, 3]), "fps": 30.0, "query": optional str}, ...]
(container decoding needs decord, which this image lacks: process_video takes decoded frames)."""
from __future__ import annotations

import argparse
import json
from typing import Dict, List, Optional, Sequence

import torch

from . import dist as tdist
from .constants import DEFAULT_MMODAL_TOKEN
from .conversation import SeparatorStyle, conv_templates
from .mm_utils import process_video, tokenizer_MMODAL_token_all


def build_prompt_ids(question: str, tokenizer, conv_mode: str = "llama_2") -> torch.Tensor:
    """evaluate.py:324-332: <video>\\n + question in the conversation template, '<sync>' appended, modal tags -> placeholders."""
    conv = conv_templates[conv_mode].copy()
    conv.append_message(conv.roles[0], DEFAULT_MMODAL_TOKEN["VIDEO"] + "\n" + question)
    conv.append_message(conv.roles[1], None)
    return tokenizer_MMODAL_token_all(conv.get_prompt() + "<sync>", tokenizer, return_tensors="pt")


def stop_string(conv_mode: str = "llama_2") -> str:
    conv = conv_templates[conv_mode]
    return conv.sep if conv.sep_style in [SeparatorStyle.SINGLE] else conv.sep2        # evaluate.py:337


def parse_output_ids(ids: Sequence[int], tokenizer, model, stop_str: Optional[str] = None) -> Dict[str, list]:
    """The drivers' id-stream parser (evaluate.py:360-411), with the vocabulary boundaries taken from the model config
    instead of the literals 32000 / 32001 / 32014: text ids < V collect a caption that a text <sync> (V) flushes; time ids
    V+1 .. V+Tv (V+1 <sync> flushes an event, V+2 <sep> ends a number, the rest are characters) ; score ids likewise."""
    cfg = model.config
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    t_sync, s_sync = V + 1, V + Tv + 1
    tt, st = model.get_model().time_tokenizer, model.get_model().score_tokenizer
    out = {"timestamps": [], "scores": [], "captions": []}
    cur_ts, cur_t, cur_ss, cur_s, cur_c = [], [], [], [], []
    for idx in ids:
        idx = int(idx)
        if idx <= V:
            if idx == V:
                cap = tokenizer.decode(cur_c, skip_special_tokens=True)
                out["captions"].append(cap)
                cur_c = []
                if stop_str and stop_str in cap:
                    break
            else:
                cur_c.append(idx)
        elif idx < s_sync:
            if idx == t_sync:
                if cur_t:
                    cur_ts.append(float("".join(cur_t)))
                out["timestamps"].append(cur_ts)
                cur_ts, cur_t = [], []
            elif idx == t_sync + 1:
                if cur_t:
                    cur_ts.append(float("".join(cur_t)))
                cur_t = []
            else:
                cur_t.append(tt.decode(idx - t_sync))
        else:
            if idx == s_sync:
                if cur_s:
                    cur_ss.append(float("".join(cur_s)))
                out["scores"].append(cur_ss)
                cur_ss, cur_s = [], []
            elif idx == s_sync + 1:
                if cur_s:
                    cur_ss.append(float("".join(cur_s)))
                cur_s = []
            else:
                cur_s.append(st.decode(idx - s_sync))
    if cur_c:
        out["captions"].append(tokenizer.decode(cur_c, skip_special_tokens=True))
    return out


@torch.no_grad()
def evaluate_videos(model, tokenizer, processor, items: Sequence[dict], prompt: str, *, num_frames: Optional[int] = None,
                    max_new_tokens: int = 512, batch_size: Optional[int] = None, conv_mode: str = "llama_2",
                    device_preprocess: bool = True, pipeline: bool = True) -> List[dict]:
    """items: dicts with "video" (decoded frames array / list / a reader with get_batch), optional "fps", "id", "query"
    (formatted into `prompt` as the tvg / vhd tasks do, evaluate.py:303-306).  Returns one result dict per item, in order, on
    every rank: {"video", "id", "timestamps", "scores", "captions", "output_ids"}."""
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    eng = model.engine
    # videos decoded together: the decode batch limit (128; 64 on the fp8 path).  When the engine holds two such banks of KV slots (max_batch >= 2 bs)
    # the chunks go through the two-stage pipeline: chunk k decodes on one stream while chunk k+1 is preprocessed, encoded and prefilled on another
    # (the default since round 4; `pipeline=False`: chunk by chunk.  Bit-identical to the chunk-by-chunk loop in every test.  The rare wrong ViT row
    # panel of round 3 needed the ViT's LayerNorm fold, which was never root-caused and left the product in round 5 — DESIGN.md)
    bs = max(1, min(batch_size or eng.decode_batch_max, eng.decode_batch_max, eng.max_batch))
    pipelined = pipeline and eng.max_batch >= 2 * bs
    nf = num_frames or getattr(model.config, "num_frames", 128)
    aspect = getattr(model.config, "image_aspect_ratio", "pad")
    mine = tdist.shard_indices(len(items), rank, world)
    local: List[List[int]] = []
    eos = tokenizer.eos_token_id if getattr(tokenizer, "eos_token_id", None) is not None else -1

    def chunks():
        for s0 in range(0, len(mine), bs):
            chunk = [items[i] for i in mine[s0: s0 + bs]]
            vids, tss, idl = [], [], []
            for it in chunk:
                v, ts = process_video(it["video"], processor, aspect, nf, fps=it.get("fps"), engine=eng if device_preprocess else None)
                vids.append(v if v.is_cuda else v.to(eng.device, eng.dtype))
                tss.append(ts)
                q = prompt.format(it["query"].strip()) if it.get("query") is not None else prompt
                idl.append(build_prompt_ids(q, tokenizer, conv_mode).tolist())
            yield vids, tss, idl, [1] * len(chunk), None

    if pipelined and len(mine) > bs:
        for out, _ in eng.generate_stream(chunks(), max_new_tokens, eos=eos):
            local.extend(out)
    else:
        for vids, tss, idl, heads, _ in chunks():
            out, _ = eng.generate(vids, tss, idl, heads, max_new_tokens, eos=eos)
            local.extend(out)
    per_rank = (len(items) + world - 1) // world
    if torch.distributed.is_initialized():
        gathered = tdist.gather_outputs(local, max_new_tokens, per_rank, eng.device)
        all_ids = tdist.merge_round_robin(gathered, len(items))
    else:
        all_ids = local
    stop = stop_string(conv_mode)
    results = []
    for it, ids in zip(items, all_ids):
        try:
            r = parse_output_ids(ids, tokenizer, model, stop)
        except ValueError as e:       # a malformed number string ('.0.1'): the reference's float() raises here too and its driver
            r = {"timestamps": [], "scores": [], "captions": [], "error": str(e)}      # gives the video up (evaluate.py:413-416)
        r.update({"video": it["video"] if isinstance(it["video"], str) else None, "id": it.get("id"), "output_ids": list(ids)})
        results.append(r)
    return results


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", required=True)
    ap.add_argument("--items", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--prompt", default="Localize a series of activity events in the video, output the start and end timestamp for "
                                        "each event, and describe each event with sentences.")
    ap.add_argument("--num-frames", type=int, default=None)
    ap.add_argument("--max-new-tokens", type=int, default=512)
    ap.add_argument("--batch-size", type=int, default=128, help="videos decoded together")
    ap.add_argument("--pipeline", dest="pipeline", action="store_true", default=True, help="(default) two-stage pipeline over the chunks: two banks of KV slots, a chunk decodes while the next is encoded")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="chunk by chunk, one bank of KV slots")
    args = ap.parse_args()
    from .mm_utils import get_model_name_from_path
    from .model.builder import load_pretrained_model
    rank, local, world = tdist.init_from_env()
    torch.cuda.set_device(local)
    tok, model, proc, _ = load_pretrained_model(args.model, None, get_model_name_from_path(args.model), device=f"cuda:{local}",
                                                max_batch=min(512, (2 if args.pipeline else 1) * args.batch_size), max_new_tokens=args.max_new_tokens)
    items = json.load(open(args.items))
    res = evaluate_videos(model, tok, proc, items, args.prompt, num_frames=args.num_frames, max_new_tokens=args.max_new_tokens,
                          batch_size=args.batch_size, pipeline=args.pipeline)
    if rank == 0:
        with open(args.out, "w") as f:
            json.dump(res, f)
    tdist.barrier()


if __name__ == "__main__":
    main()
