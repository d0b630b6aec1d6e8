"""Prompt assembly with the reference's interface (trace/conversation.py): `conv_templates[name].copy()`,
`.append_message(role, msg)`, `.get_prompt()`, `.roles`, `.sep`, `.sep2`, `.sep_style`.

SUPPORTED TEMPLATES: 'llama_2' only is verified — every inference driver selects it (trace/eval/evaluate.py:226,327-337;
scripts/inference/inference.py:31) and its output is pinned string-for-string by tests/golden/host_functions.json for the
reference's five task prompts.  'plain' carries the reference's values (conversation.py:420-428) but is not pinned.  The
reference's other 15 chat templates (vicuna, mistral_instruct, mpt, qwen, llava_*; conversation.py:329-498) belong to
training / chat front ends outside the accelerated path: `conv_templates[name]` raises a KeyError that says so, and
`default_conversation` is llama_2 here (the reference's default, vicuna_v1, is never used by the inference drivers).
The SeparatorStyle members exist because drivers compare against them (evaluate.py:337)."""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()
    QWEN = auto()


def _text(message):
    # multimodal messages are (text, media, mode) tuples in the reference; only the text enters the prompt
    return message[0] if isinstance(message, tuple) else message


@dataclasses.dataclass
class Conversation:
    system: str
    roles: tuple
    messages: list
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False
    modality: str = "image"

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2,
                            version=self.version, modality=self.modality)

    def get_prompt(self) -> str:
        msgs = [(r, m) for r, m in self.messages]
        if msgs and isinstance(msgs[0][1], tuple):                      # conversation.py:39-49
            tag = f"<{self.modality}>"
            first = msgs[0][1][0].replace(tag, "").strip()
            msgs[0] = (msgs[0][0], f"{tag}\n" + first)
        st = self.sep_style
        if st == SeparatorStyle.LLAMA_2:                                # conversation.py:78-98
            out = ""
            for i, (role, message) in enumerate(msgs):
                if i == 0:
                    assert message, "first message should not be none"
                    assert role == self.roles[0], "first message should come from user"
                if not message:
                    continue
                message = _text(message)
                if i == 0:
                    message = f"<<SYS>>\n{self.system}\n<</SYS>>\n\n" + message
                if i % 2 == 0:
                    out += self.sep + f"[INST] {message} [/INST]"
                else:
                    out += " " + message + " " + self.sep2
            return out.lstrip(self.sep)
        if st == SeparatorStyle.PLAIN:
            seps = [self.sep, self.sep2]
            out = self.system
            for i, (role, message) in enumerate(msgs):
                if message:
                    out += _text(message) + seps[i % 2]
            return out
        raise ValueError(f"Invalid style: {st}")


# Llama-2 chat default system prompt (the text Meta ships with Llama-2-chat; the reference embeds it at
# trace/conversation.py:383-394).  Pinned by the golden prompts.
_LLAMA2_SYSTEM = (
    "You are a helpful, respectful and honest assistant. Always answer as helpfully as possible, while being safe.  "
    "Your answers should not include any harmful, unethical, racist, sexist, toxic, dangerous, or illegal content. "
    "Please ensure that your responses are socially unbiased and positive in nature.\n\n"
    "If a question does not make any sense, or is not factually coherent, explain why instead of answering something "
    "not correct. If you don't know the answer to a question, please don't share false information.")

conv_llama_2 = Conversation(system=_LLAMA2_SYSTEM, roles=("USER", "ASSISTANT"), version="llama_v2", messages=[], offset=0,
                            sep_style=SeparatorStyle.LLAMA_2, sep="<s>", sep2="</s>")
conv_plain = Conversation(system="", roles=("", ""), messages=[], offset=0, sep_style=SeparatorStyle.PLAIN, sep="\n", sep2=None)


class _Templates(dict):
    def __missing__(self, name):
        raise KeyError(f"conversation template {name!r} is not provided by trace_amd (supported: {sorted(self)}); the TRACE inference "
                       "drivers use 'llama_2' — see the module docstring")


conv_templates = _Templates({"llama_2": conv_llama_2, "plain": conv_plain})
default_conversation = conv_llama_2
