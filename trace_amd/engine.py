"""Python face of the HIP engine (libtrace_hip.so).  PyTorch is plumbing here: device allocations for the
tensors the caller hands in/out, and the current HIP stream.  All arithmetic happens in the C-ABI library."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .config import TraceConfig
from .model.encoders import TimeTower, ScoreTower

EPI_NONE, EPI_RESIDUAL, EPI_QUICKGELU, EPI_SWIGLU, EPI_PARTIAL = 0, 1, 2, 3, 4


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _i32(seq) -> "C.Array":
    seq = [int(x) for x in seq]
    return (C.c_int32 * len(seq))(*seq)


class TraceEngine:
    def __init__(self, cfg: TraceConfig, device: int = 0, max_batch: int = 1, max_ctx: Optional[int] = None,
                 max_frames: Optional[int] = None, max_new_tokens: int = 1024, vit_batch_frames: Optional[int] = None,
                 llm_fp8=False, dtype: torch.dtype = torch.bfloat16):
        """dtype: the 16-bit element type everything is stored and multiplied in — torch.bfloat16 (libtrace_hip.so; north_star's configs) or
        torch.float16 (libtrace_hip_f16.so: the reference's own inference dtype, trace/model/builder.py:50,127,147); accumulation is fp32 in both."""
        # llm_fp8: False / None = bf16 weights; "w8a8" (True is accepted as its alias) = W8A8 prefill GEMMs and decode GEMVs; "weight_only" = W8A8
        # prefill GEMMs, weight-only decode GEMVs (bf16 activations).  Anything else is an error — a typo must not silently pick a numerics scheme.
        if llm_fp8 in (False, None):
            self.fp8_scheme = None
        elif llm_fp8 is True or llm_fp8 == "w8a8":
            self.fp8_scheme = "w8a8"
        elif llm_fp8 == "weight_only":
            self.fp8_scheme = "weight_only"
        else:
            raise ValueError(f"llm_fp8 must be False, True / 'w8a8' or 'weight_only', got {llm_fp8!r}")
        if self.fp8_scheme and dtype != torch.bfloat16:
            raise ValueError("the fp8 weight path exists in the bf16 library only")
        if not torch.cuda.is_available():
            raise _lib.TraceHipError("no HIP device visible: the TRACE hot path only runs on an MI355X (no CPU fallback)")
        self.dtype = dtype
        self.lib = _lib.load(_lib.element_of(dtype))
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        max_frames = max_frames or cfg.num_frames
        if max_ctx is None:
            vis = ((max_frames // 2 + 1) * (cfg.vision_grid // 2 + 1) ** 2 if cfg.mm_projector_type == "stc_connector"
                   else max_frames * cfg.tokens_per_frame)
            max_ctx = min(cfg.max_position_embeddings, vis + 1024 + max_new_tokens)
        self.max_batch, self.max_ctx, self.max_frames, self.max_new_tokens = max_batch, max_ctx, max_frames, max_new_tokens
        if vit_batch_frames is None:
            vit_batch_frames = self.full_round_frames(cfg) if max_batch > 1 else max_frames
        self.vit_batch_frames = max(int(vit_batch_frames), max_frames)
        c = _lib.TraceConfigC(
            cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads,
            cfg.num_key_value_heads, cfg.time_vocab_size, cfg.score_vocab_size, cfg.rms_norm_eps, cfg.rope_theta,
            cfg.vision_hidden_size, cfg.vision_intermediate_size, cfg.vision_layers_used, cfg.vision_num_heads,
            cfg.vision_image_size, cfg.vision_patch_size, cfg.vision_layer_norm_eps, cfg.num_slots, cfg.slot_ln_eps,
            cfg.slot_rope_base, max_frames, max_ctx, max_batch, max_new_tokens,
            1 if cfg.mm_projector_type == "stc_connector" else 0, self.vit_batch_frames,
            {None: 0, "w8a8": 1, "weight_only": 2}[self.fp8_scheme])
        self.llm_fp8 = self.fp8_scheme is not None
        h = C.c_void_p()
        _lib.check(self.lib.trace_ctx_create(C.byref(c), device, C.byref(h)))
        self.h = h
        self.time_tower, self.score_tower = TimeTower(), ScoreTower()
        self._B = 0
        self._max_new = 0
        self._stage_ev = None          # stage_timing(): [(kind, event, event)] while on
        self._stage_videos = 0
        self._dbg = None               # debugging hook: callable(tag, index, tensor-or-None) called between the stages (tools/pipeline_stress.py)

    @property
    def decode_batch_max(self) -> int:
        """sequences one decode batch can hold: the KV slots, at most 256 (64 on the fp8 weight path)"""
        return min(self.max_batch, 64 if self.llm_fp8 else self.lib.trace_op_sk_rows())

    @staticmethod
    def full_round_frames(cfg: TraceConfig) -> int:
        """Frames per ViT call when several videos are encoded together: the largest count whose token rows fill 384 row
        tiles of 256 (170 frames x 577 tokens = 383.2 tiles), so that every ViT GEMM — 4 / 12 / 16 column tiles — runs in
        whole rounds of the 256 CUs (6 / 18 / 24) while the fc1 input panel (201 MB) still fits the 256 MB Infinity Cache.
        One 128-frame video alone is 289 row tiles: 4.5 / 13.5 / 18.1 rounds, the last round of each launch half empty.
        Measured (tools/vit_chunk_sweep.py, tower ms per video): 128 -> 55.0, 170 -> 53.9, 227 (512 tiles; the fc1 panel
        spills the cache: 931 vs 979 TFLOP/s on that launch) -> 53.6, 64 -> 58.6."""
        return max(1, min(256, (384 * 256) // cfg.vision_tokens))

    def close(self):
        if getattr(self, "h", None):
            self.lib.trace_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def load_weights(self, items: Iterable[Tuple[str, torch.Tensor]]) -> int:
        """items: (reference state-dict name, tensor).  Tensors are converted to the engine's contiguous 16-bit dtype; host or device."""
        n = 0
        for name, t in items:
            t = t.detach().to(self.dtype).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*(list(t.shape) or [1]))
            rc = _lib.check(self.lib.trace_ctx_load_tensor(self.h, name.encode(), C.c_void_p(t.data_ptr()),
                                                          1 if t.is_cuda else 0, shape, t.dim()))
            n += rc == 0
        _lib.check(self.lib.trace_ctx_finalize(self.h))
        return n

    def device_bytes(self) -> int:
        return int(self.lib.trace_ctx_device_bytes(self.h))

    # ---- stages --------------------------------------------------------------------------------
    def _frames(self, frames: torch.Tensor):
        if frames.dim() != 4:
            raise ValueError("frames must be [T,3,H,W]")
        if frames.dtype not in (self.dtype, torch.float32):
            frames = frames.to(self.dtype)
        frames = frames.to(self.device).contiguous()
        return frames, (1 if frames.dtype == torch.float32 else 0)

    CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
    CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

    def preprocess_frames(self, frames_u8, pad: bool = True, image_mean: Sequence[float] = CLIP_MEAN,
                          image_std: Sequence[float] = CLIP_STD, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """process_video's per-frame image work (mm_utils.py:456-462) on the device: uint8 RGB [T,H,W,3] (tensor or numpy,
        host or device) -> [T,3,S,S] `dtype` (bf16 for the engine, fp32 = the reference's FloatTensor bit for bit)."""
        x = torch.as_tensor(frames_u8)
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError(f"expected uint8 [T,H,W,3] RGB frames, got {x.dtype} {tuple(x.shape)}")
        dtype = dtype or self.dtype
        if dtype not in (self.dtype, torch.float32):
            raise ValueError(f"dtype must be the engine's {self.dtype} or float32")
        x = x.to(self.device).contiguous()
        T, H, W, _ = x.shape
        S = self.cfg.vision_image_size
        out = torch.empty((T, 3, S, S), dtype=dtype, device=self.device)
        mean, std = (C.c_float * 3)(*image_mean), (C.c_float * 3)(*image_std)
        _lib.check(self.lib.trace_preprocess_frames(self.h, _ptr(x), T, H, W, int(bool(pad)), mean, std, _ptr(out),
                                                    1 if dtype == torch.float32 else 0, _stream()))
        return out

    def vit_forward(self, frames: torch.Tensor, want_output: bool = True) -> Optional[torch.Tensor]:
        frames, dt = self._frames(frames)
        T = frames.shape[0]
        if not want_output:
            _lib.check(self.lib.trace_vit_forward(self.h, _ptr(frames), dt, T, None, _stream()))
            return None
        out = torch.empty((T, self.cfg.vision_patches, self.cfg.vision_hidden_size), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.trace_vit_forward(self.h, _ptr(frames), dt, T, _ptr(out), _stream()))
        return out

    def slot_pool(self, feats: Optional[torch.Tensor], T: int) -> torch.Tensor:
        out = torch.empty((T, self.cfg.num_slots, self.cfg.hidden_size), dtype=self.dtype, device=self.device)
        if feats is not None:
            feats = feats.to(self.device, self.dtype).contiguous()
        _lib.check(self.lib.trace_slot_pool(self.h, _ptr(feats), T, _ptr(out), _stream()))
        return out

    def stc_connector(self, feats: Optional[torch.Tensor], T: int) -> torch.Tensor:
        """Legacy STC connector (projector_type 'stc_connector'); result also becomes the video rows for splice()."""
        g = self.cfg.vision_grid // 2 + 1
        rows = (T // 2 + 1) * g * g
        out = torch.empty((rows, self.cfg.hidden_size), dtype=self.dtype, device=self.device)
        if feats is not None:
            feats = feats.to(self.device, self.dtype).contiguous()
        n = C.c_int(0)
        _lib.check(self.lib.trace_stc_connector(self.h, _ptr(feats), T, _ptr(out), C.byref(n), _stream()))
        assert n.value == rows
        return out

    def time_ids(self, timestamps: Sequence[Sequence[float]]) -> List[int]:
        """encode_time + [:-1] (trace_arch.py:243,271-288): 6 ids per frame; all frames must agree in length."""
        toks = [self.time_tower.encode_ids(t) for t in timestamps]
        assert all(len(x) == len(toks[0]) for x in toks), f"{timestamps} {[len(x) for x in toks]}"
        if len(toks[0]) - 1 != self.cfg.time_tokens_per_frame:
            raise ValueError("each frame must carry exactly one timestamp (6 time tokens)")
        return [i for x in toks for i in x[:-1]]

    def encode_video(self, frames: torch.Tensor, timestamps, want_output: bool = False):
        frames, dt = self._frames(frames)
        T = frames.shape[0]
        ids = _i32(self.time_ids(timestamps))
        out = None
        if want_output:
            out = torch.empty((T * self.cfg.tokens_per_frame, self.cfg.hidden_size), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.trace_encode_video(self.h, _ptr(frames), dt, T, ids, _ptr(out), _stream()))
        return out

    def encode_features(self, feats: torch.Tensor, timestamps, want_output: bool = False):
        """encode_video from ViT features computed earlier (vit_forward on a frame batch that may span several videos)"""
        T = feats.shape[0]
        assert feats.dtype == self.dtype and feats.is_cuda and feats.is_contiguous()
        ids = _i32(self.time_ids(timestamps))
        out = None
        if want_output:
            out = torch.empty((T * self.cfg.tokens_per_frame, self.cfg.hidden_size), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.trace_encode_features(self.h, _ptr(feats), T, ids, _ptr(out), _stream()))
        return out

    def vit_forward_many(self, videos: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """ViT features of several videos, the frames pushed through the tower `vit_batch_frames` at a time regardless of
        video boundaries (per-frame arithmetic: identical to per-video calls).  Returns one [T_v, patches, v_hidden] view
        per video."""
        vids = [self._frames(v) for v in videos]
        dts = {dt for _, dt in vids}
        if len(dts) != 1:
            raise ValueError("all videos of a batch must share one frame dtype")
        dt = dts.pop()
        counts = [v.shape[0] for v, _ in vids]
        total = sum(counts)
        feats = torch.empty((total, self.cfg.vision_patches, self.cfg.vision_hidden_size), dtype=self.dtype, device=self.device)
        F = self.vit_batch_frames
        pos, chunk, chunk_n = 0, [], 0          # gather frames into chunks of F (a copy only when a chunk spans videos)
        def flush():
            nonlocal chunk, chunk_n, pos
            if not chunk_n:
                return
            x = chunk[0] if len(chunk) == 1 else torch.cat(chunk, dim=0)
            _lib.check(self.lib.trace_vit_forward(self.h, _ptr(x.contiguous()), dt, chunk_n, _ptr(feats[pos:pos + chunk_n]), _stream()))
            pos += chunk_n
            chunk, chunk_n = [], 0
        for v, _ in vids:
            o = 0
            while o < v.shape[0]:
                n = min(F - chunk_n, v.shape[0] - o)
                chunk.append(v[o:o + n])
                chunk_n += n
                o += n
                if chunk_n == F:
                    flush()
        flush()
        out, o = [], 0
        for n in counts:
            out.append(feats[o:o + n])
            o += n
        return out

    def splice(self, input_ids: Sequence[int], time_rows: Sequence[int] = (), score_rows: Sequence[int] = (),
               want_output: bool = False):
        ids = _i32(input_ids)
        tr, sr = _i32(time_rows), _i32(score_rows)
        L = C.c_int(0)
        # length is known up-front: n_ids - 1 + video rows; allocate generously when a copy is requested
        out = None
        if want_output:
            out = torch.empty((self.max_ctx, self.cfg.hidden_size), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.trace_splice_embeds(self.h, ids, len(ids), tr, len(tr), sr, len(sr), C.byref(L), _ptr(out), _stream()))
        return (L.value, out[: L.value]) if want_output else L.value

    def prefill(self, slot: int, L: int, embeds: Optional[torch.Tensor] = None, want_hidden: bool = False):
        hid = torch.empty((L, self.cfg.hidden_size), dtype=self.dtype, device=self.device) if want_hidden else None
        if embeds is not None:
            embeds = embeds.to(self.device, self.dtype).contiguous()
        _lib.check(self.lib.trace_llm_prefill(self.h, slot, _ptr(embeds), L, _ptr(hid), _stream()))
        return hid

    def head_logits(self, hidden: torch.Tensor, head: int) -> torch.Tensor:
        """masked fp32 logits [R, total_vocab] of final-norm hidden rows under one head (forward()'s logits at every position)"""
        assert hidden.dtype == self.dtype and hidden.is_cuda and hidden.is_contiguous()
        out = torch.empty((hidden.shape[0], self.cfg.total_vocab), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trace_llm_head_logits(self.h, _ptr(hidden), hidden.shape[0], int(head), _ptr(out), _stream()))
        return out

    def prefill_pair(self, slot0: int, embeds0: torch.Tensor, embeds1: torch.Tensor):
        """two spliced prompts of equal length -> KV slots slot0, slot0 + 1 in one pass (trace_llm_prefill_pair)"""
        assert embeds0.shape == embeds1.shape and embeds0.dtype == self.dtype and embeds0.is_cuda
        _lib.check(self.lib.trace_llm_prefill_pair(self.h, slot0, _ptr(embeds0.contiguous()), _ptr(embeds1.contiguous()),
                                                   embeds0.shape[0], _stream()))

    PREFILL_GROUP = 4          # equal-length prompts of max_ctx rows one prefill pass takes (trace_llm_prefill_multi); shorter prompts: prefill_group(L)
    PREFILL_GROUP_MAX = 8

    @property
    def prefill_rows(self) -> int:
        """rows the prefill workspaces hold (engine.hip pf_rows): n prompts of L rows fit one pass while n * L <= this"""
        return max(self.PREFILL_GROUP * self.max_ctx, min(8192, self.PREFILL_GROUP_MAX * self.max_ctx))

    def prefill_group(self, L: int) -> int:
        """How many equal-length prompts of L rows to put through one prefill pass: the n <= 8 that fits the workspaces and minimises the projections' cost per
        prompt in whole rounds of the 256 CUs — a 256 x 256 tile occupies a CU for one tile time whether its round is full or not, so cost(n) = sum over the four
        projections of ceil(row panels x column tiles / #CUs) x K.  C2 (L = 1967): 4 (31 row panels; 8 would not fit).  C4 (L = 1086): 7 (30 panels: the o / down
        grid in 1.9 rounds; four prompts' 17 panels take 2 rounds for 1.06 of work: -22 % of the projections' time per prompt)."""
        c = self.cfg
        H, I = c.hidden_size, c.intermediate_size
        qkv = (c.num_attention_heads + 2 * c.num_key_value_heads) * (H // c.num_attention_heads)
        shapes = ((qkv, H), (H, H), (2 * I, H), (H, I))          # (N, K) of qkv, o, gate|up, down
        ncu = 256
        best, best_cost = 1, None
        for n in range(1, self.PREFILL_GROUP_MAX + 1):
            if n * L > self.prefill_rows:
                break
            panels = -(-n * L // 256)
            cost = sum(-(-panels * -(-N // 256) // ncu) * K for N, K in shapes) / n
            if best_cost is None or cost < best_cost * 0.995:      # (ties go to the smaller group: less latency to the first prefilled slot)
                best, best_cost = n, cost
        return best

    def prefill_multi(self, slot0: int, embeds: Sequence[torch.Tensor]):
        """up to 4 spliced prompts of equal length -> KV slots slot0 .. slot0 + n - 1 in one pass"""
        n = len(embeds)
        assert 1 <= n <= self.PREFILL_GROUP_MAX and n * embeds[0].shape[0] <= self.prefill_rows and all(e.shape == embeds[0].shape and e.dtype == self.dtype and e.is_cuda for e in embeds)
        keep = [e.contiguous() for e in embeds]
        ptrs = (C.c_void_p * n)(*[e.data_ptr() for e in keep])
        _lib.check(self.lib.trace_llm_prefill_multi(self.h, slot0, ptrs, n, keep[0].shape[0], _stream()))

    # ---- decode --------------------------------------------------------------------------------
    def decode_begin(self, slots: Sequence[int], heads: Sequence[int], max_new: int, eos: int = -1,
                     forced: Optional[Sequence[Sequence[int]]] = None, want_logits: bool = False):
        B = len(slots)
        self._B, self._max_new = B, max_new
        f = None
        if forced is not None:
            flat = []
            for row in forced:
                row = list(row)[:max_new]
                flat += row + [0] * (max_new - len(row))
            f = _i32(flat)
        lg = torch.empty((B, self.cfg.total_vocab), dtype=torch.float32, device=self.device) if want_logits else None
        _lib.check(self.lib.trace_decode_begin(self.h, _i32(slots), B, _i32(heads), max_new, eos, f, _ptr(lg), _stream()))
        return lg

    def decode_steps(self, n: int, use_graph: bool = True, want_logits: bool = False):
        lg = None
        if want_logits:
            lg = torch.empty((self._B, self.cfg.total_vocab), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trace_decode_steps(self.h, n, 1 if use_graph else 0, _ptr(lg), _stream()))
        return lg

    def decode_read(self):
        B, mn = self._B, self._max_new
        ids = (C.c_int32 * (B * mn))()
        ln = (C.c_int32 * B)()
        hd = (C.c_int32 * B)()
        _lib.check(self.lib.trace_decode_read(self.h, ids, ln, hd, _stream()))
        out = [[ids[b * mn + i] for i in range(ln[b])] for b in range(B)]
        return out, list(hd)

    def host_mode(self, on: bool):
        """Host-driven token selection (sampling / stopping criteria): logits come back every step, ids go in via feed()."""
        _lib.check(self.lib.trace_decode_host_mode(self.h, 1 if on else 0))

    def feed(self, tokens: Sequence[int]):
        _lib.check(self.lib.trace_decode_feed(self.h, _i32(tokens), len(tokens), _stream()))

    def set_profile(self, mode: int):
        """0 off; 1 time decode_steps calls; 2 also bracket one gate|up GEMV launch per decode step with HIP events."""
        _lib.check(self.lib.trace_set_profile(self.h, int(mode)))

    def get_profile(self) -> List[float]:
        buf = (C.c_float * 20)()
        _lib.check(self.lib.trace_get_profile(self.h, buf, 20))
        return list(buf)

    # ---- one-call convenience: what generate() does for one batch of videos ----------------------
    # ---- stage timing inside a run (bench.py): HIP-event pairs around the tower / slot-pool / prefill calls of encode_prefill ----
    def stage_timing(self, on: bool):
        """on: encode_prefill brackets its tower, encode_features and prefill calls with event pairs on its stream (generate(): every batch;
        generate_stream(): only the batch that fills the pipeline, while the encode stage has the GPU to itself); read with stage_times()."""
        self._stage_ev = [] if on else None
        self._stage_videos = 0

    def stage_times(self) -> dict:
        """{'videos': n, 'tower_ms': .., 'slotpool_ms': .., 'prefill_ms': ..} summed over the recorded batches (synchronises the device)"""
        torch.cuda.synchronize(self.device)
        out = {"videos": getattr(self, "_stage_videos", 0), "tower_ms": 0.0, "slotpool_ms": 0.0, "prefill_ms": 0.0}
        for kind, e0, e1 in (getattr(self, "_stage_ev", None) or []):
            out[kind + "_ms"] += e0.elapsed_time(e1)
        return out

    def _bracket(self, kind: str, fn, record: bool):
        if not record:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self._stage_ev.append((kind, e0, e1))
        return r

    def feature_groups(self, videos: Sequence[torch.Tensor]):
        """(first, last + 1) video ranges whose ViT features (12 GB at most) one vit_forward_many call computes — encode_prefill's grouping"""
        per_frame = self.cfg.vision_patches * self.cfg.vision_hidden_size * 2
        groups, g0, B = [], 0, len(videos)
        while g0 < B:
            g1, nbytes = g0, 0
            while g1 < B and (g1 == g0 or nbytes + videos[g1].shape[0] * per_frame <= 12 << 30):
                nbytes += videos[g1].shape[0] * per_frame
                g1 += 1
            groups.append((g0, g1))
            g0 = g1
        return groups

    def encode_prefill(self, videos: Sequence[torch.Tensor], timestamps: Sequence, input_ids: Sequence[Sequence[int]], slot0: int = 0,
                       record_stages: bool = False):
        """Stage 1 of generate() for a batch: CLIP tower (over the batch's frame stream) -> slot pool + time rows -> splice -> prefill
        into KV slots slot0 .. slot0 + B - 1.  Runs on the current stream; touches no decode state, so it may run on another stream
        while an earlier batch (other slots) is being decoded (generate_stream)."""
        B = len(videos)
        if slot0 < 0 or slot0 + B > self.max_batch:
            raise ValueError(f"slots {slot0}..{slot0 + B - 1} exceed the engine's {self.max_batch} KV slots")
        rec = bool(record_stages) and getattr(self, "_stage_ev", None) is not None
        if rec:
            self._stage_videos += B
        # prefill: runs of up to 4 neighbours whose spliced prompts have the same length share one pass (M = 4 L fills the GEMM tile grids in whole
        # rounds of the CUs); the spliced embeddings of a run wait in `held`
        held: List[torch.Tensor] = []                 # spliced embeds of slots held_slot0 .. (equal lengths)
        held_slot0 = 0

        def flush():
            nonlocal held
            if len(held) == 1:
                self._bracket("prefill", lambda: self.prefill(held_slot0, held[0].shape[0], embeds=held[0]), rec)
            elif len(held) == 2:
                self._bracket("prefill", lambda: self.prefill_pair(held_slot0, held[0], held[1]), rec)
            elif held:
                self._bracket("prefill", lambda: self.prefill_multi(held_slot0, held), rec)
            held = []

        feats = None
        if B > 1 and self.vit_batch_frames > self.max_frames and self.cfg.mm_projector_type != "stc_connector":
            groups, feats = self.feature_groups(videos), {}      # the tower runs over groups of videos (whole GEMM rounds); 12 GB of features at a time
        for b in range(B):
            if feats is not None:
                if b not in feats:
                    g = next(g for g in groups if g[0] <= b < g[1])
                    feats = dict(zip(range(g[0], g[1]), self._bracket("tower", lambda: self.vit_forward_many(videos[g[0]:g[1]]), rec)))
                fb = feats.pop(b)
                if self._dbg is not None:
                    self._dbg("feats", slot0 + b, fb)
                self._bracket("slotpool", lambda: self.encode_features(fb, timestamps[b]), rec)
            else:
                self._bracket("tower", lambda: self.encode_video(videos[b], timestamps[b]), rec)      # (tower + slot pool in one call)
            if B == 1:
                L1 = self.splice(input_ids[b])
                self._bracket("prefill", lambda: self.prefill(slot0, L1), rec)
                continue
            L, emb = self.splice(input_ids[b], want_output=True)
            if held and (held[0].shape[0] != L or len(held) == self.prefill_group(held[0].shape[0])):
                flush()
            if not held:
                held_slot0 = slot0 + b
            held.append(emb)
        flush()

    def decode(self, slots: Sequence[int], heads: Sequence[int], max_new_tokens: int, eos: int = -1, use_graph: bool = True,
               forced: Optional[Sequence[Sequence[int]]] = None):
        """Stage 2 of generate(): the greedy loop over prefilled KV slots, on the current stream -> (ids per sequence, final heads)."""
        self.decode_begin(list(slots), heads, max_new_tokens, eos, forced)
        if max_new_tokens > 1:
            if eos < 0:
                self.decode_steps(max_new_tokens - 1, use_graph)
            else:
                done, chunk = 0, 32
                while done < max_new_tokens - 1:
                    n = min(chunk, max_new_tokens - 1 - done)
                    self.decode_steps(n, use_graph)
                    done += n
                    ids, _ = self.decode_read()
                    if all(len(x) and x[-1] == eos for x in ids):
                        break
        return self.decode_read()

    def generate(self, videos: Sequence[torch.Tensor], timestamps: Sequence, input_ids: Sequence[Sequence[int]],
                 heads: Sequence[int], max_new_tokens: int, eos: int = -1, use_graph: bool = True,
                 forced: Optional[Sequence[Sequence[int]]] = None):
        B = len(videos)
        if B > self.decode_batch_max:
            raise ValueError(f"batch {B} exceeds the engine's decode batch {self.decode_batch_max}")
        self.encode_prefill(videos, timestamps, input_ids, 0, record_stages=True)
        return self.decode(range(B), heads, max_new_tokens, eos, use_graph, forced)

    # ---- two-stage pipeline over a stream of batches ---------------------------------------------------
    def make_streams(self, decode_cus: int = 0):
        """The pipeline's (encode stream, decode stream).  decode_cus > 0 confines the decode stream to that many CUs and the encode
        stream to the rest (trace_stream_create: CU-masked HIP streams, spread evenly over the 8 XCDs; the persistent GEMM's grid on such a
        stream is capped at the stream's CU count): the MFMA-bound GEMMs
        and the HBM-bound decode kernels then run side by side instead of taking turns on the whole chip."""
        if decode_cus <= 0:
            return torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        if decode_cus % 8 or decode_cus >= ncu:
            raise ValueError("decode_cus must be a multiple of 8 (the XCD count) below the CU count")
        def masked(lo, n):
            h = C.c_void_p()
            _lib.check(self.lib.trace_stream_create(self.h, lo, n, C.byref(h)))
            return torch.cuda.ExternalStream(h.value, device=self.device)
        dec, enc = masked(0, decode_cus), masked(decode_cus, ncu - decode_cus)      # (trace_stream_create caps each stream's persistent GEMMs at its CU count)
        return enc, dec

    def generate_stream(self, batches: Iterable, max_new_tokens: int, eos: int = -1, use_graph: bool = True, streams=None):
        """generate() over a stream of batches as a two-stage pipeline: while batch k decodes (HBM-bound) on one stream, batch k+1
        runs its ViT + slot pool + prefill (MFMA-bound) on another, into the other half of the KV slots.  `batches` yields
        (videos, timestamps, input_ids, heads, forced-or-None); yields generate()'s result per batch, in order.  Every batch holds at
        most max_batch // 2 videos.  Results are identical to generate() batch by batch: the stages share no buffers (KV banks,
        prefill / ViT scratch vs decode scratch) and every kernel's reductions have a fixed order.

        The decode stage is issued from a worker thread: one decode batch is ~75 k kernel dispatches, far more than a HIP stream's
        queue holds, so the issuing thread blocks until the GPU has consumed most of them — issued from the caller's thread it would
        hold back the encode stage's launches and the two stages would run one after the other (profiles/r03_overlap_probe_single_thread.jsonl)."""
        import concurrent.futures as cf
        half = self.max_batch // 2
        if half < 1:
            raise ValueError("generate_stream needs an engine with max_batch >= 2 (two banks of KV slots)")
        enc_s, dec_s = streams if streams is not None else self.make_streams()
        cur = torch.cuda.current_stream(self.device)
        enc_s.wait_stream(cur); dec_s.wait_stream(cur)

        def dec_job(bank, heads, forced, B, ready):
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(dec_s):
                dec_s.wait_event(ready)
                return self.decode(range(bank * half, bank * half + B), heads, max_new_tokens, eos, use_graph, forced)

        pending = None                                  # (bank, heads, forced, B, ready event): prefilled, waiting for its decode
        bank = 0
        # profiling mode 2 brackets single launches with HIP events: meaningful only while a stage has the GPU to itself — the first batch's
        # encode (pipeline fill) and the last batch's decode (drain); in between the brackets are off
        brackets = lambda m: _lib.check(self.lib.trace_set_profile_brackets(self.h, m))
        try:                                            # whatever ends the generator (exhaustion, an exception, the caller closing it early): both brackets back on
            with cf.ThreadPoolExecutor(max_workers=1, thread_name_prefix="trace-decode") as pool:
                for item in batches:
                    videos, timestamps, input_ids, heads, forced = item
                    if len(videos) > min(half, self.decode_batch_max):
                        raise ValueError(f"batch of {len(videos)} exceeds max_batch // 2 = {half} (two KV banks)")
                    brackets(1 if pending is None else 0)
                    fut = pool.submit(dec_job, *pending) if pending is not None else None
                    enc_s.wait_stream(cur)                  # the batch's frames may have been made on the caller's stream (device preprocessing)
                    try:
                        with torch.cuda.stream(enc_s):
                            self.encode_prefill(videos, timestamps, input_ids, bank * half, record_stages=pending is None)
                            if self._dbg is not None:
                                self._dbg("prefilled", bank * half, None)
                            ready = torch.cuda.Event()
                            ready.record(enc_s)
                    finally:
                        out = fut.result() if fut is not None else None      # also drains the decode stage before an exception propagates
                    if out is not None:
                        yield out
                    pending = (bank, list(heads), forced, len(videos), ready)
                    bank ^= 1
                if pending is not None:
                    brackets(2)
                    out = pool.submit(dec_job, *pending).result()
                    cur.wait_stream(dec_s); cur.wait_stream(enc_s)
                    yield out
        finally:
            brackets(3)


# ---- kernel-level wrappers for unit tests / microbenchmarks -------------------------------------
class ops:
    element = "bf16"               # which library the wrappers call: "bf16" (libtrace_hip.so) or "f16" (libtrace_hip_f16.so); ops.use()

    @staticmethod
    def use(element: str):
        _lib.load(element)
        ops.element = element

    @staticmethod
    def dtype() -> torch.dtype:
        return torch.float16 if ops.element == "f16" else torch.bfloat16

    @staticmethod
    def set_gemm_variant(v: int):
        _lib.check(_lib.load(ops.element).trace_op_set_gemm_variant(v))

    @staticmethod
    def gemm(A, W, bias=None, R=None, epilogue=EPI_NONE):
        lib = _lib.load(ops.element)
        M, K = A.shape
        N = W.shape[0]
        No = N // 2 if epilogue == EPI_SWIGLU else N
        Cc = torch.empty((M, No), dtype=ops.dtype(), device=A.device)
        _lib.check(lib.trace_op_gemm(_ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(Cc), No, _ptr(bias), _ptr(R),
                                     0 if R is None else R.stride(0), M, N, K, epilogue, _stream()))
        return Cc

    @staticmethod
    def layernorm(x, w, b, eps):
        lib = _lib.load(ops.element)
        y = torch.empty_like(x)
        _lib.check(lib.trace_op_layernorm(_ptr(x), _ptr(y), _ptr(w), _ptr(b), x.shape[0], x.shape[1], eps, _stream()))
        return y

    @staticmethod
    def rmsnorm(x, w, eps):
        lib = _lib.load(ops.element)
        y = torch.empty_like(x)
        _lib.check(lib.trace_op_rmsnorm(_ptr(x), _ptr(y), _ptr(w), x.shape[0], x.shape[1], eps, _stream()))
        return y

    @staticmethod
    def attention(q, k, v, causal, scale):
        """q [B, nq, heads, hd]; k, v [B, nkv, kv_heads, hd] -> [B, nq, heads, hd]"""
        lib = _lib.load(ops.element)
        Bn, nq, heads, hd = q.shape
        nkv, kvh = k.shape[1], k.shape[2]
        pad = (nkv + 63) // 64 * 64
        vt = torch.empty((Bn * kvh * hd * pad,), dtype=ops.dtype(), device=q.device)
        o = torch.empty_like(q)
        _lib.check(lib.trace_op_attention(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(vt), Bn, heads, kvh, nq, nkv, hd,
                                          1 if causal else 0, scale, _stream()))
        return o

    @staticmethod
    def tile_pack(W):
        """row-major [N, K] -> the decode GEMV tile layout (same shape/bytes, permuted)"""
        lib = _lib.load(ops.element)
        out = torch.empty_like(W)
        _lib.check(lib.trace_op_tile_pack(_ptr(W), _ptr(out), W.shape[0], W.shape[1], _stream()))
        return out

    @staticmethod
    def skinny_gemm(X, W, R=None, epilogue=EPI_NONE, tiled=False, want_partial=True):
        """decode GEMV; `tiled`: W was passed through tile_pack.  EPI_PARTIAL returns the fp32 partial rows [KS, sk_rows, N]."""
        lib = _lib.load(ops.element)
        Bn, K = X.shape
        N = W.shape[0]
        if epilogue == EPI_PARTIAL:
            ks = lib.trace_op_skinny_ks(N, K, epilogue, Bn)
            out = torch.zeros((ks, lib.trace_op_sk_rows(), N), dtype=torch.float32, device=X.device) if want_partial else None
        else:
            No = N // 2 if epilogue == EPI_SWIGLU else N
            out = torch.empty((Bn, No), dtype=ops.dtype(), device=X.device)
        _lib.check(lib.trace_op_skinny_gemm(_ptr(X), _ptr(W), _ptr(out), _ptr(R), Bn, N, K, epilogue, int(tiled), _stream()))
        return out

    @staticmethod
    def gemm_partial(X, W, tiled: int = 0):
        """decode batches above 64 rows: X [M <= 128, K] . W [N, K]^T as fp32 k-chunk partial rows [ks, sk_rows, N] (split-K MFMA GEMM);
        tiled: W went through tile_pack (1; 5 = + the 4-stage K-tile ring)"""
        lib = _lib.load(ops.element)
        M, K = X.shape
        N = W.shape[0]
        out = torch.zeros((lib.trace_op_gemm_partial_ks(N, K), lib.trace_op_sk_rows(), N), dtype=torch.float32, device=X.device)
        _lib.check(lib.trace_op_gemm_partial(_ptr(X), _ptr(W), _ptr(out), M, N, K, int(tiled), _stream()))
        return out

    @staticmethod
    def gemm_swiglu_tiled(X, Wt, ring: bool = True):
        """gate|up of a wide decode step: X [M <= 128, K], Wt = tile_pack(16-row interleaved gate|up [N, K]) -> [M, N/2] bf16"""
        lib = _lib.load(ops.element)
        M, K = X.shape
        N = Wt.shape[0]
        out = torch.empty((M, N // 2), dtype=ops.dtype(), device=X.device)
        _lib.check(lib.trace_op_gemm_swiglu_tiled(_ptr(X), _ptr(Wt), _ptr(out), M, N, K, int(ring), _stream()))
        return out

    @staticmethod
    def skinny_fused_norm(part_in, R, w, eps, W):
        """(xout, out-partials): xout = bf16(sum_k part_in[k, :B]) + R; partial rows [ks, sk_rows, N] of RMSNorm(xout; w) . W^T  (B <= 4)"""
        lib = _lib.load(ops.element)
        Bn, K = R.shape
        N = W.shape[0]
        ks_in = 0 if part_in is None else part_in.shape[0]
        xout = torch.empty_like(R)
        out = torch.zeros((lib.trace_op_skinny_ks(N, K, EPI_PARTIAL, Bn), lib.trace_op_sk_rows(), N), dtype=torch.float32, device=R.device)
        _lib.check(lib.trace_op_skinny_fused_norm(_ptr(part_in), ks_in, _ptr(R), _ptr(xout), _ptr(w), eps, _ptr(W), _ptr(out), Bn, N, K, _stream()))
        return xout, out

    @staticmethod
    def skinny_ks(N, K, epilogue, B):
        return _lib.load(ops.element).trace_op_skinny_ks(N, K, epilogue, B)

    @staticmethod
    def swiglu_combine(part, Bn):
        lib = _lib.load(ops.element)
        N2 = part.shape[2]
        out = torch.empty((Bn, N2 // 2), dtype=ops.dtype(), device=part.device)
        _lib.check(lib.trace_op_swiglu_combine(_ptr(part), part.shape[0], N2, _ptr(out), Bn, _stream()))
        return out

    @staticmethod
    def add_rmsnorm(part, R, w, eps):
        """(x, y): x = bf16(sum_ks part[ks, b]) + R[b]; y = RMSNorm(x) * w"""
        lib = _lib.load(ops.element)
        Bn, N = R.shape
        x, y = torch.empty_like(R), torch.empty_like(R)
        _lib.check(lib.trace_op_add_rmsnorm(_ptr(part), part.shape[0], _ptr(R), _ptr(x), _ptr(w), _ptr(y), Bn, N, eps, _stream()))
        return x, y

    @staticmethod
    def attn_decode(q, kcache, vcache, pos, nsplit, scale, vtcache=None):
        """q [B, nq*128]; caches [B, nkv, max_ctx, 128]; pos int32 [B] (device).  The kernel reads V transposed
        ([B, nkv, 128, max_ctx], the engine's cache layout): built here unless `vtcache` is passed."""
        lib = _lib.load(ops.element)
        if vtcache is None:
            vtcache = vcache.transpose(2, 3).contiguous()
        vcache = vtcache
        Bn = q.shape[0]
        nkv, max_ctx = kcache.shape[1], kcache.shape[2]
        nq = q.shape[1] // 128
        ws = torch.zeros((Bn * nq * nsplit * 130,), dtype=torch.float32, device=q.device)
        o = torch.empty_like(q)
        _lib.check(lib.trace_op_attn_decode(_ptr(q), _ptr(kcache), _ptr(vcache), _ptr(pos), _ptr(o), _ptr(ws), Bn, nq, nkv,
                                            max_ctx, nsplit, scale, _stream()))
        return o

    # ---- fp8 (e4m3) path pieces ----
    @staticmethod
    def quant_rows_fp8(X):
        """X bf16 [rows, K] -> (uint8 e4m3 bytes [rows, K], fp32 scale [rows])"""
        lib = _lib.load(ops.element)
        q = torch.empty(X.shape, dtype=torch.uint8, device=X.device)
        sx = torch.empty((X.shape[0],), dtype=torch.float32, device=X.device)
        _lib.check(lib.trace_op_quant_rows_fp8(_ptr(X), _ptr(q), _ptr(sx), X.shape[0], X.shape[1], _stream()))
        return q, sx

    @staticmethod
    def gemm_fp8(A8, sa, W8, sw, R=None, epilogue=EPI_NONE):
        lib = _lib.load(ops.element)
        M, K = A8.shape
        N = W8.shape[0]
        No = N // 2 if epilogue == EPI_SWIGLU else N
        Cc = torch.empty((M, No), dtype=ops.dtype(), device=A8.device)
        _lib.check(lib.trace_op_gemm_fp8(_ptr(A8), _ptr(sa), _ptr(W8), _ptr(sw), _ptr(Cc), _ptr(R), M, N, K, epilogue, _stream()))
        return Cc

    @staticmethod
    def skinny_w8(X, W8, sw):
        """weight-only decode GEMV: X bf16 [B, K], W8 uint8 e4m3 [N, K] + row scales -> fp32 [B, N]"""
        lib = _lib.load(ops.element)
        Bn, K = X.shape
        N = W8.shape[0]
        out = torch.empty((Bn, N), dtype=torch.float32, device=X.device)
        _lib.check(lib.trace_op_skinny_w8(_ptr(X), _ptr(W8), _ptr(sw), _ptr(out), Bn, N, K, _stream()))
        return out

    @staticmethod
    def skinny_fp8(X8, sx, W8, sw):
        lib = _lib.load(ops.element)
        Bn, K = X8.shape
        N = W8.shape[0]
        out = torch.empty((Bn, N), dtype=torch.float32, device=X8.device)
        _lib.check(lib.trace_op_skinny_fp8(_ptr(X8), _ptr(sx), _ptr(W8), _ptr(sw), _ptr(out), Bn, N, K, _stream()))
        return out
