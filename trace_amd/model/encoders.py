"""Time / score / sync "towers": the integer side (tokenizers + fixed-width number encoding).

Mirrors trace/model/multimodal_encoder/{time_encoder,score_encoder,sync_encoder}.py of the reference:
13-symbol vocabulary {<sync>:0, <sep>:1, '0'..'9':2..11, '.':12} (time_encoder.py:80-88), numbers formatted
'0>6.1f' (time) / '0>3.1f' (score), '<sep>' between values and a trailing '<sync>' (time_encoder.py:52-68).
The embedding tables themselves (Embedding(13,4096) / Embedding(1,4096)) live on the device inside the engine."""
from __future__ import annotations

from typing import List, Sequence

import torch

VOCAB = {"<sync>": 0, "<sep>": 1, **{str(i): i + 2 for i in range(10)}, ".": 12}
IDS_TO_TOKENS = {v: k for k, v in VOCAB.items()}


class _Enc:
    """what a tokenizer call returns: .input_ids"""
    input_ids: List[int]


class NumberTokenizer:
    """Stand-in for TimeTokenizer / ScoreTokenizer (PreTrainedTokenizer subclasses in the reference): the
    drivers only call `.decode(i)` with an int or 0-d tensor (trace/eval/evaluate.py:395,408) and `.get_vocab()`."""

    def __init__(self):
        self.vocab = dict(VOCAB)
        self.ids_to_tokens = dict(IDS_TO_TOKENS)
        self._keys = sorted(self.vocab, key=len, reverse=True)      # longest first, as the reference's regex alternation matches

    def get_vocab(self):
        return self.vocab

    def get_vocab_size(self):
        return len(self.vocab)

    def __len__(self):
        return len(self.vocab)

    def tokenize(self, text: str) -> List[str]:
        out, i = [], 0
        keys = self._keys
        while i < len(text):
            for k in keys:
                if text.startswith(k, i):
                    out.append(k)
                    i += len(k)
                    break
            else:
                i += 1          # characters outside the vocabulary are dropped (regex findall in the reference)
        return out

    def __call__(self, text: str):
        e = _Enc()
        e.input_ids = [self.vocab[t] for t in self.tokenize(text)]
        return e

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self.ids_to_tokens.get(ids, None)
        return [self.ids_to_tokens.get(int(i), None) for i in ids]

    def decode(self, token_ids, skip_special_tokens: bool = False, **kw) -> str:
        if isinstance(token_ids, torch.Tensor):
            token_ids = token_ids.tolist()
        if isinstance(token_ids, int):
            token_ids = [token_ids]
        # PreTrainedTokenizer.decode joins plain tokens with ' ' only between *added* tokens; for this
        # vocabulary (no added tokens) the pieces are concatenated.
        return "".join(self.ids_to_tokens[int(i)] for i in token_ids)


TimeTokenizer = NumberTokenizer
ScoreTokenizer = NumberTokenizer


_TOK = None          # one tokenizer for every encode() (its tables never change; building it per call was most of a call's time)


def _encode_ids(values: Sequence[float], fmt: str) -> List[int]:
    """The ids of _encode() as a plain list (the engine's per-frame path: 128 calls per video; no tensor per call)."""
    global _TOK
    if _TOK is None:
        _TOK = NumberTokenizer()
    tok = _TOK
    sep, sync = tok("<sep>").input_ids, tok("<sync>").input_ids
    ids: List[int] = []
    for i, v in enumerate(values):
        if i:
            ids.extend(sep)
        ids.extend(tok(format(v, fmt)).input_ids)
    ids.extend(sync)
    return ids


def _encode(values: Sequence[float], fmt: str) -> torch.Tensor:
    return torch.tensor(_encode_ids(values, fmt), dtype=torch.long)


class TimeTower:
    """encode() of the reference TimeTower (time_encoder.py:52-68); forward() is a device gather in the engine."""
    fmt = "0>6.1f"

    def __init__(self, tokenizer=None):
        self.tokenizer = tokenizer or NumberTokenizer()

    def encode(self, values: Sequence[float]) -> torch.Tensor:
        return _encode(values, self.fmt)

    def encode_ids(self, values: Sequence[float]) -> List[int]:
        """encode() as a list of ints"""
        return _encode_ids(values, self.fmt)


class ScoreTower(TimeTower):
    fmt = "0>3.1f"
