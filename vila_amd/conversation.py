"""Conversation -> token ids / training labels: the text-side input producer of the path (llava/utils/tokenizer.py), host-only string and
integer work over whatever tokenizer the checkpoint ships.

  * `tokenize_conversation`   (tokenizer.py:70-114, the `SeparatorStyle.AUTO` branch every NVILA script runs: `conv_auto` is the default
                               conversation, llava/conversation.py:114-164) — messages `{"from": "human" | "gpt", "value": ...}` through the
                               tokenizer's own chat template; media tokens are added special tokens, so the plain tokenizer call places
                               their ids (`tokenizer_image_token`, llava/mm_utils.py:574-575)
  * `preprocess_conversation` (tokenizer.py:124-169) — SFT labels: the conversation is tokenised a second time with every assistant reply
                               replaced by a sentinel token; whatever the two id rows do NOT share (the replies and their end-of-turn token)
                               is supervised, the rest is IGNORE_INDEX
  * `infer_stop_tokens`       (tokenizer.py:172-183) — the end-of-turn strings generation should stop on, read off the same template

Pinned by tests/golden/conversation_ref.json = the reference's own three functions taken out of the file with `ast` and executed on the same
tokenizer (oracle/make_golden_conversation.py).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence

import torch

from .configs import IGNORE_INDEX

SENTINEL_TOKEN = "<vila/sentinel>"                                    # llava/constants.py:32

DUMMY_CONVERSATION = [{"from": "human", "value": "question"}, {"from": "gpt", "value": "answer"}] * 10      # tokenizer.py:33-36


_ROLE = {"human": "user", "gpt": "assistant"}

# `--chat_template qwen2` of every NVILA script (scripts/NVILA/stage1_9tile.sh:16, scripts/NVILA-Lite/align.sh:20): the builder replaces the
# tokenizer's template by llava/model/language_model/chat_templates/qwen2.jinja (language_model/builder.py:194-200).  Restated here — a default
# system turn "You are a helpful assistant" (no full stop, unlike Qwen2's own) unless the conversation opens with a system turn, every turn as
# `<|im_start|>role\ncontent<|im_end|>\n`, turns whose content is None left out — and pinned to the reference file's rendering of the same
# conversations (tests/golden/conversation_ref.json: "rendered").
CHAT_TEMPLATES = {
    "qwen2": ("{%- macro turn(role, content) -%}{{ '<|im_start|>' ~ role ~ '\\n' ~ content ~ '<|im_end|>\\n' }}{%- endmacro -%}"
              "{%- if messages[0]['role'] != 'system' -%}{{ turn('system', 'You are a helpful assistant') }}{%- endif -%}"
              "{%- for m in messages -%}{%- if m['content'] is not none -%}{{ turn(m['role'], m['content']) }}{%- endif -%}{%- endfor -%}"
              "{%- if add_generation_prompt -%}{{ '<|im_start|>assistant\\n' }}{%- endif -%}"),
}


def prepare_tokenizer(tokenizer, chat_template: Optional[str] = None, media_tokens: Optional[Dict[str, str]] = None):
    """What `build_llm_and_tokenizer` does to the tokenizer it loaded (language_model/builder.py:194-211), in its order: the named chat template,
    `stop_tokens` / `stop_token_ids` read off that template, the media tokens added as special tokens with their ids in `media_token_ids`."""
    if chat_template is not None:
        if chat_template not in CHAT_TEMPLATES:
            raise ValueError(f"unknown chat template '{chat_template}' (known: {sorted(CHAT_TEMPLATES)})")
        tokenizer.chat_template = CHAT_TEMPLATES[chat_template]
    if getattr(tokenizer, "chat_template", None):
        tokenizer.stop_tokens = infer_stop_tokens(tokenizer)
        tokenizer.stop_token_ids = tokenizer.convert_tokens_to_ids(tokenizer.stop_tokens)
    tokenizer.media_tokens = dict(media_tokens or {"image": "<image>", "video": "<vila/video>"})      # llava/constants.py:34-37
    tokenizer.media_token_ids = {}
    for name, token in tokenizer.media_tokens.items():
        tokenizer.add_tokens([token], special_tokens=True)
        tokenizer.media_token_ids[name] = tokenizer.convert_tokens_to_ids(token)
    return tokenizer


def tokenize_conversation(messages: Sequence[Dict[str, str]], tokenizer, add_generation_prompt: bool = False,
                          overrides: Optional[Dict[str, str]] = None, no_system_prompt: bool = False) -> torch.Tensor:
    """-> 1-D int64 ids.  Every message's text is stripped IN PLACE first, as the reference does (tokenizer.py:77-78); `overrides` replaces
    the text of every message of a sender; `no_system_prompt` puts an EMPTY system turn in front (which silences a template's default one)."""
    overrides = overrides or {}
    turns = [{"role": "system", "content": ""}] if no_system_prompt else []
    for m in messages:
        m["value"] = m["value"].strip()
    for m in messages:
        if m["from"] not in _ROLE:
            raise ValueError(f"Unexpected sender '{m['from']}' in conversation entry.")
        turns.append({"role": _ROLE[m["from"]], "content": overrides.get(m["from"], m["value"])})
    text = tokenizer.apply_chat_template(turns, add_generation_prompt=add_generation_prompt, tokenize=False)
    return tokenizer(text, return_tensors="pt").input_ids[0]          # media tokens are added tokens: mm_utils.py:574-575


def _ensure_sentinel(tokenizer) -> int:
    """tokenizer.py:117-121: the sentinel is an added special token, registered once per tokenizer object."""
    if not hasattr(tokenizer, "sentinel_token"):
        tokenizer.add_tokens([SENTINEL_TOKEN], special_tokens=True)
        tokenizer.sentinel_token = SENTINEL_TOKEN
        tokenizer.sentinel_token_id = tokenizer.convert_tokens_to_ids(SENTINEL_TOKEN)
    return int(tokenizer.sentinel_token_id)


def _frame_of(template: List[int], sentinel: int, also_before: bool) -> List[int]:
    """The template row without its sentinels and without the token AFTER each (the end-of-turn token, which is supervised); `also_before`
    drops the token in front of each sentinel too.  A sentinel in the very last position stays (the reference's loop stops one short)."""
    keep = [True] * len(template)
    for k in range(len(template) - 1):
        if template[k] == sentinel:
            keep[k] = keep[k + 1] = False
            if also_before and k > 0:
                keep[k - 1] = False
    return [t for t, kp in zip(template, keep) if kp]


def _supervised(ids: List[int], frame: List[int]):
    """Greedy in-order match of `frame` inside `ids` -> (which positions of ids are NOT part of the frame, whether the frame was used up)."""
    p, out = 0, []
    for t in ids:
        hit = p < len(frame) and t == frame[p]
        p += int(hit)
        out.append(not hit)
    return out, p == len(frame)


def preprocess_conversation(conversation: Sequence[Dict[str, str]], tokenizer, no_system_prompt: bool = False) -> Dict[str, Any]:
    """-> {"input_ids", "labels"} (1-D int64).  The conversation is tokenised twice — as it is, and with a sentinel in place of every reply;
    whatever the first row has beyond the second row's frame (the replies and their end-of-turn tokens) is its own label, the rest is
    IGNORE_INDEX.  If the frame is not used up the match is tried once more with the token in front of each sentinel dropped as well (a
    tokenizer that glues a space to it); a second failure masks the whole sample (tokenizer.py:124-169)."""
    inputs = tokenize_conversation(conversation, tokenizer, no_system_prompt=no_system_prompt)
    sentinel = _ensure_sentinel(tokenizer)
    template = tokenize_conversation(conversation, tokenizer, overrides={"gpt": SENTINEL_TOKEN}, no_system_prompt=no_system_prompt).tolist()
    ids = inputs.tolist()
    for also_before in (False, True):
        free, used_up = _supervised(ids, _frame_of(template, sentinel, also_before))
        if used_up:
            break
    else:
        logging.getLogger(__name__).error(f"Failed to process the conversation: '{conversation}'. All tokens will be masked in the label.")
        free = [False] * len(ids)
    labels = torch.where(torch.tensor(free, dtype=torch.bool), inputs, torch.full_like(inputs, IGNORE_INDEX))
    return {"input_ids": inputs, "labels": labels}


def infer_stop_tokens(tokenizer) -> List[str]:
    """tokenizer.py:172-183: the tokenizer's EOS plus whatever token follows an assistant reply in its chat template."""
    sentinel = _ensure_sentinel(tokenizer)
    template = tokenize_conversation([dict(m) for m in DUMMY_CONVERSATION], tokenizer, overrides={"gpt": SENTINEL_TOKEN}).tolist()
    after = {tokenizer.decode(template[k + 1]) for k in range(len(template) - 1) if template[k] == sentinel}
    return list({tokenizer.eos_token} | after)
