"""Host-side `TraceMistralForCausalLM` with the reference's call surface (trace/model/language_model/
trace_mistral.py): `.generate(inputs, images_or_videos=, modal_list=, video_timestamps=, heads=, max_new_tokens=,
do_sample=, ...) -> LongTensor[B, n_new]`, `.forward(...)` -> object with `.logits`, `.config`, `.get_model()`,
`.get_vision_tower()`, `.to()`, `.eval()`.  Everything between the frame tensor and the token ids runs in the HIP
engine (trace_amd/engine.py -> libtrace_hip.so); this class only adapts arguments."""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch

from ..config import TraceConfig
from ..constants import MMODAL_TOKEN_INDEX, NUM_FRAMES
from .encoders import NumberTokenizer, ScoreTower, TimeTower


class _VisionTower:
    """Attribute holder standing in for CLIPVisionTower (clip_encoder.py): drivers read `.image_processor`."""

    def __init__(self, cfg: TraceConfig, image_processor):
        self.config = SimpleNamespace(hidden_size=cfg.vision_hidden_size, image_size=cfg.vision_image_size,
                                      patch_size=cfg.vision_patch_size)
        self.image_processor = image_processor
        self.is_loaded = True
        self.hidden_size = cfg.vision_hidden_size
        self.num_patches = cfg.vision_patches

    def load_model(self):
        return None

    def to(self, *a, **k):
        return self


class _MetaModel:
    """`model.get_model()`: exposes the tokenizers/towers the drivers touch (trace_arch.py:31-40)."""

    def __init__(self, cfg: TraceConfig, vision_tower: _VisionTower):
        self.time_tokenizer = NumberTokenizer()
        self.score_tokenizer = NumberTokenizer()
        self.time_tower = TimeTower(self.time_tokenizer)
        self.score_tower = ScoreTower(self.score_tokenizer)
        self.vision_tower = vision_tower

    def get_vision_tower(self):
        return self.vision_tower

    def get_time_tower(self):
        return self.time_tower

    def get_score_tower(self):
        return self.score_tower


class TraceMistralForCausalLM:
    def __init__(self, config: TraceConfig, engine, image_processor=None):
        self.config = config
        self.engine = engine
        self.vocab_size = config.vocab_size
        self.time_vocab_size = config.time_vocab_size
        self.score_vocab_size = config.score_vocab_size
        # trace_mistral.py:86-88
        self.swap_tokens = {config.vocab_size: 1, config.vocab_size + 1: 2,
                            config.vocab_size + config.time_vocab_size + 1: 0}
        self.model = _MetaModel(config, _VisionTower(config, image_processor))
        self.device = engine.device
        self.dtype = engine.dtype

    # ---- nn.Module-like conveniences the drivers call ----
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.get_vision_tower()

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def half(self):
        return self

    def eval(self):
        return self

    def resize_token_embeddings(self, n):
        if n != self.config.vocab_size:
            raise NotImplementedError("the engine's embedding table is fixed at load time")

    # ---- generate (trace_mistral.py:268-314) ----
    @torch.no_grad()
    def generate(self, inputs=None, images_or_videos=None, times=None, scores=None, video_timestamps=None,
                 modal_list=None, heads=None, max_new_tokens: int = 128, do_sample: bool = False, temperature: float = 1.0,
                 eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None, stopping_criteria=None,
                 use_cache: bool = True, attention_mask=None, position_ids=None, **kwargs):
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")          # trace_mistral.py:282-283
        if images_or_videos is None:
            raise NotImplementedError("text-only generation is outside the accelerated path")
        cfg, eng = self.config, self.engine
        eng.host_mode(False)                    # a previous forward() leaves the engine armed for its decode form
        self._live_kv = None
        ids = inputs if isinstance(inputs, torch.Tensor) else torch.tensor(inputs)
        if ids.dim() == 1:
            ids = ids.unsqueeze(0)
        B = ids.shape[0]
        if len(images_or_videos) != B:
            raise ValueError("one video per prompt row")
        if heads is None:
            heads = [0] * B
        assert len(heads) == B                                                    # trace_mistral.py:245
        legacy_stc = video_timestamps is None and cfg.mm_projector_type == "stc_connector"
        if video_timestamps is None and not legacy_stc:
            raise ValueError("video_timestamps is required on the TRACE path (time tokens per frame)")
        nf = cfg.num_frames if hasattr(cfg, "num_frames") else NUM_FRAMES
        vids = []
        for x, modal in zip(images_or_videos, modal_list or ["video"] * B):
            if modal == "image":                                                  # trace_arch.py:221
                x = x.unsqueeze(0).expand(nf, -1, -1, -1) if x.dim() == 3 else x.expand(nf, -1, -1, -1)
            vids.append(x)
        eos = cfg.eos_token_id if eos_token_id is None else eos_token_id
        if eos is None:
            eos = -1
        id_lists = [row.tolist() for row in ids]
        if B > eng.max_batch:
            raise ValueError(f"batch {B} exceeds the engine's max_batch {eng.max_batch}")
        if legacy_stc:
            # legacy trace.infer() flow (trace/__init__.py:23-75): STC connector, no time tokens, text head only
            for b in range(B):
                eng.vit_forward(vids[b])
                eng.stc_connector(None, vids[b].shape[0])
                eng.prefill(b, eng.splice(id_lists[b]))
            eng.decode_begin(list(range(B)), [0] * B, max_new_tokens, eos)
            if max_new_tokens > 1:
                eng.decode_steps(max_new_tokens - 1)
            out, new_heads = eng.decode_read()
        elif not do_sample and not stopping_criteria:
            out, new_heads = eng.generate(vids, video_timestamps, id_lists, list(heads), max_new_tokens, eos=eos)
        else:
            out, new_heads = self._generate_stepwise(vids, video_timestamps, id_lists, list(heads), max_new_tokens, eos,
                                                     do_sample, temperature, stopping_criteria, ids)
        for b in range(B):                       # the reference mutates `heads` in place (trace_mistral.py:342)
            heads[b] = int(new_heads[b])
        pad = eos if pad_token_id is None else pad_token_id
        n = max(len(x) for x in out)
        res = torch.full((B, n), pad if pad is not None and pad >= 0 else 0, dtype=torch.long)
        for b, row in enumerate(out):
            res[b, : len(row)] = torch.tensor(row, dtype=torch.long)
        return res.to(self.device)

    def _generate_stepwise(self, vids, timestamps, id_lists, heads, max_new, eos, do_sample, temperature, stopping, prompt_ids):
        """Sampling / stopping-criteria path: one device step at a time with the masked logits brought back
        (the reference's HF sampling loop does the same round trip every token)."""
        eng = self.engine
        B = len(vids)
        for b in range(B):
            eng.encode_video(vids[b], timestamps[b])
            eng.prefill(b, eng.splice(id_lists[b]))
        done = [False] * B
        eng.host_mode(True)
        try:
            lg = eng.decode_begin(list(range(B)), heads, max_new, eos=eos, want_logits=True)
            for step in range(max_new):
                if do_sample and temperature and temperature > 0:
                    probs = torch.softmax(lg.float() / temperature, dim=-1)
                    tok = torch.multinomial(probs, 1).view(-1).tolist()
                else:
                    tok = torch.argmax(lg, dim=-1).tolist()
                eng.feed(tok)
                for b in range(B):
                    done[b] = done[b] or (eos >= 0 and tok[b] == eos)
                if all(done) or step == max_new - 1:
                    break
                if stopping:
                    # HF StoppingCriteriaList: a row stops when ANY criterion fires for it; generate(inputs_embeds=...) hands the
                    # criteria the generated ids only (the reference calls super().generate with inputs_embeds, trace_mistral.py:301-312)
                    cur, _ = eng.decode_read()
                    n = max(len(x) for x in cur)
                    gen = torch.tensor([x + [0] * (n - len(x)) for x in cur], dtype=torch.long)
                    fired = torch.zeros(B, dtype=torch.bool)
                    for sc in stopping:
                        r = sc(gen, None)
                        fired |= (r.view(-1).bool().cpu() if isinstance(r, torch.Tensor) else torch.full((B,), bool(r)))
                    for b in range(B):
                        done[b] = done[b] or bool(fired[b])
                    if all(done):
                        break
                lg = eng.decode_steps(1, use_graph=False, want_logits=True)
            return eng.decode_read()
        finally:
            eng.host_mode(False)

    # ---- forward (trace_mistral.py:114-264) ----
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, images=None, times=None, scores=None, video_timestamps=None, heads=None, **kwargs):
        """Both forms of the reference's forward():
          * prefill form (`images=(videos, modal_list)`, `input_ids` with the modal placeholders): logits `[B, L, V']` fp32 for
            EVERY position of the spliced sequence (`heads` given: V' = V+1+Tv+Sv with everything outside the row's head at
            -inf; `heads=None`: the text|sync logits `[B, L, V+1]`, trace_mistral.py:190-193) and `past_key_values` = a handle
            on the KV slots the rows were prefilled into;
          * decode form (`input_ids [B, 1]` + that handle): the next-token embedding by id range (trace_arch.py:345-375), one
            decoder step, logits `[B, 1, V']`.
        Deviations, by design: `past_key_values` is an opaque handle (the cache lives in the engine, never in torch tensors);
        rows of a batch must splice to one length (the reference pads); `labels` / `inputs_embeds` are training / internal
        inputs and raise; in the decode form the active head is tracked on the device by the swap-token rule the reference's
        `prepare_inputs_for_generation` applies (trace_mistral.py:336-344), so a `heads` argument that disagrees with it raises."""
        if labels is not None or inputs_embeds is not None:
            raise NotImplementedError("forward(labels= / inputs_embeds=) belongs to training; outside the accelerated path")
        if input_ids is None:
            raise ValueError("input_ids is required")
        eng, cfg = self.engine, self.config
        ids = input_ids if input_ids.dim() == 2 else input_ids.unsqueeze(0)
        B = ids.shape[0]
        if past_key_values is not None:
            if not isinstance(past_key_values, _KVHandle) or past_key_values is not getattr(self, "_live_kv", None):
                raise ValueError("past_key_values must be the handle returned by the previous forward() of this model")
            if ids.shape[1] != 1 or B != past_key_values.B:
                raise ValueError("decode form takes input_ids [B, 1]")
            eng.feed([int(x) for x in ids[:, 0].tolist()])
            lg = eng.decode_steps(1, use_graph=False, want_logits=True)
            _, cur = eng.decode_read()
            if heads is not None and [int(h) for h in heads] != [int(h) for h in cur]:
                raise ValueError(f"heads={list(heads)} disagrees with the head state the fed tokens imply ({cur})")
            if heads is None:
                if any(cur):
                    raise ValueError("heads=None asks for text logits but the fed tokens switched a row to the time/score head")
                lg = lg[:, : cfg.vocab_size + 1]
            return SimpleNamespace(logits=lg.unsqueeze(1), past_key_values=past_key_values, loss=None, hidden_states=None,
                                   attentions=None)
        if images is None:
            raise NotImplementedError("text-only forward is outside the accelerated path")
        vids, modals = images
        hd = [0] * B if heads is None else [int(h) for h in heads]
        assert len(hd) == B                                                       # trace_mistral.py:245
        if B > eng.max_batch:
            raise ValueError(f"batch {B} exceeds the engine's max_batch {eng.max_batch}")
        rows = []
        for b in range(B):
            x = vids[b]
            if (modals[b] if modals else "video") == "image":
                nf = cfg.num_frames if hasattr(cfg, "num_frames") else NUM_FRAMES
                x = x.unsqueeze(0).expand(nf, -1, -1, -1) if x.dim() == 3 else x.expand(nf, -1, -1, -1)
            eng.encode_video(x, video_timestamps[b])
            trow = [int(i) for ev in (times[b] if times is not None else []) for i in self.model.time_tower.encode(ev)]
            srow = [int(i) for ev in (scores[b] if scores is not None else []) for i in self.model.score_tower.encode(ev)]
            L = eng.splice(ids[b].tolist(), trow, srow)
            hid = eng.prefill(b, L, want_hidden=True)
            rows.append(eng.head_logits(hid, hd[b]))
        if any(r.shape != rows[0].shape for r in rows):
            raise NotImplementedError("rows of one forward() batch must splice to the same length (no padding path in the engine)")
        logits = torch.stack(rows, 0)
        if heads is None:
            logits = logits[..., : cfg.vocab_size + 1]
        # arm the decode form: token selection stays with the caller (host mode), the cache stays in the engine.  The number of steps the
        # KV slots still have room for bounds the arming; a prompt that fills the context gets its logits and no decode form
        # (past_key_values=None) instead of an error after all the work is done.
        room = min(eng.max_new_tokens, eng.max_ctx - logits.shape[1])
        self._live_kv = None
        if room >= 1:
            eng.host_mode(True)
            try:
                eng.decode_begin(list(range(B)), hd, room, eos=-1)
            except Exception:
                eng.host_mode(False)
                raise
            self._live_kv = _KVHandle(B)
        return SimpleNamespace(logits=logits, past_key_values=self._live_kv, loss=None, hidden_states=None, attentions=None)

    __call__ = forward


class _KVHandle:
    """`past_key_values` of this build: the KV cache stays inside the engine (slots 0..B-1); the handle only proves that the
    decode-form call follows the prefill-form call that filled them."""

    def __init__(self, B: int):
        self.B = B

    def get_seq_length(self, *a, **k):
        raise NotImplementedError("the KV cache is engine-resident")
