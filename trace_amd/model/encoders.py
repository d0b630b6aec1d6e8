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


class NumberTokenizer:
    """Stand-in for TimeTokenizer / ScoreTokenizer (PreTrainedTokenizer subclasses in the reference): the
    drivers only call `.decode(i)` with an int or 0-d tensor (trace/eval/evaluate.py:395,408) and `.get_vocab()`."""

    def __init__(self):
        self.vocab = dict(VOCAB)
        self.ids_to_tokens = dict(IDS_TO_TOKENS)

    def get_vocab(self):
        return self.vocab

    def get_vocab_size(self):
        return len(self.vocab)

    def __len__(self):
        return len(self.vocab)

    def tokenize(self, text: str) -> List[str]:
        out, i = [], 0
        keys = sorted(self.vocab, key=len, reverse=True)
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
        class _Enc:
            pass
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


def _encode(values: Sequence[float], fmt: str) -> torch.Tensor:
    tok = NumberTokenizer()
    strs = [format(v, fmt) for v in values]
    ids: List[int] = []
    for i, s in enumerate(strs):
        if i:
            ids.extend(tok("<sep>").input_ids)
        ids.extend(tok(s).input_ids)
    ids.extend(tok("<sync>").input_ids)
    return torch.tensor(ids, dtype=torch.long)


class TimeTower:
    """encode() of the reference TimeTower (time_encoder.py:52-68); forward() is a device gather in the engine."""
    fmt = "0>6.1f"

    def __init__(self, tokenizer=None):
        self.tokenizer = tokenizer or NumberTokenizer()

    def encode(self, values: Sequence[float]) -> torch.Tensor:
        return _encode(values, self.fmt)


class ScoreTower(TimeTower):
    fmt = "0>3.1f"
