"""REFERENCE-EXECUTED fixture for the text-side input producer (llava/utils/tokenizer.py) — TEST INFRASTRUCTURE.

`llava/utils/tokenizer.py` cannot be imported here (it pulls llava.mm_utils -> torchvision, llava.utils.logging -> loguru), but
`tokenize_conversation` (:70-114), `_maybe_add_sentinel_token` (:117-121), `preprocess_conversation` (:124-169) and `infer_stop_tokens`
(:172-183) are string / integer code over a tokenizer: the four definitions (and `DUMMY_CONVERSATION`) are taken out of the file with `ast`
and EXECUTED unchanged.  What they need from the package is supplied as it stands in the reference: `tokenizer_image_token` = the plain
tokenizer call (llava/mm_utils.py:574-575), `IGNORE_INDEX = -100`, `SENTINEL_TOKEN = "<vila/sentinel>"` (llava/constants.py:28,32),
`conversation_lib.default_conversation.sep_style = SeparatorStyle.AUTO` (llava/conversation.py:114-118,164: `conv_auto`, the default of every
NVILA script).

No tokenizer files exist offline, so the tokenizer is a byte-level BPE trained HERE on a fixed corpus, with the reference's `qwen2` chat template (`--chat_template qwen2` of every NVILA
script: llava/model/language_model/chat_templates/qwen2.jinja, installed the way language_model/builder.py:194-200 does; the fixture also
keeps that template's own rendering of five conversations), `<|im_end|>` as EOS and the media tokens added as special tokens the way
`build_llm_and_tokenizer` adds them (language_model/builder.py:190-211).  Its serialised form is stored in the fixture, so the test runs the
HIP-side functions over the byte-identical tokenizer without re-training it.

    python oracle/make_golden_conversation.py       # seconds; needs /root/reference; writes tests/golden/conversation_ref.json
"""
from __future__ import annotations

import ast
import copy
import json
import logging
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/llava/utils/tokenizer.py"
OUT = os.path.join(ROOT, "tests", "golden", "conversation_ref.json")

REF_TEMPLATE = "/root/reference/llava/model/language_model/chat_templates/qwen2.jinja"


def reference_chat_template() -> str:
    """`--chat_template qwen2` as `build_llm_and_tokenizer` installs it (language_model/builder.py:194-200): the file, minus indentation and newlines."""
    return open(REF_TEMPLATE).read().replace("    ", "").replace("\n", "")


CORPUS = ["the quick brown fox jumps over the lazy dog", "a red square sits on a blue table\nnext to a green circle", "what is in this picture ?",
          "describe the image in detail , please .", "system user assistant You are a helpful assistant.", "question answer question answer",
          "there are two cats and one dog in the video", "图片里有一个红色的方块", "hello world, hello again!  two  spaces"] * 4
CONVERSATIONS = [
    [{"from": "human", "value": "<image>\nwhat is in this picture ?"}, {"from": "gpt", "value": "a red square on a blue table"}],
    [{"from": "human", "value": "  describe the image in detail , please .  "}, {"from": "gpt", "value": " there are two cats\nand one dog "},
     {"from": "human", "value": "and the video ? <vila/video>"}, {"from": "gpt", "value": "a quick brown fox"}],
    [{"from": "human", "value": "hello"}, {"from": "gpt", "value": ""}],                                     # an empty reply
    [{"from": "human", "value": "<image><image> two pictures"}, {"from": "gpt", "value": "图片里有一个红色的方块"}, {"from": "human", "value": "again"},
     {"from": "gpt", "value": "hello again!"}, {"from": "human", "value": "once more"}, {"from": "gpt", "value": "the lazy dog"}],
    [{"from": "human", "value": "answer with the question itself : what is in this picture ?"}, {"from": "gpt", "value": "what is in this picture ?"}],
]


def build_tokenizer(serialised: str = None, chat_template: str = None):
    """Train (or re-load) the stand-in tokenizer.  -> PreTrainedTokenizerFast with EOS and media tokens; `chat_template` (a jinja string) is
    installed when given — the generator passes the REFERENCE's, the tests install the HIP side's own restatement instead."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    if serialised is None:
        tk = Tokenizer(models.BPE())
        tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
        tk.decoder = decoders.ByteLevel()
        tk.train_from_iterator(CORPUS, trainers.BpeTrainer(vocab_size=400, special_tokens=["<|endoftext|>", "<|im_start|>", "<|im_end|>"],
                                                           initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False))
    else:
        tk = Tokenizer.from_str(serialised)
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|im_end|>", pad_token="<|endoftext|>")
    if chat_template is not None:
        tok.chat_template = chat_template
    tok.add_tokens(["<image>", "<vila/video>"], special_tokens=True)            # builder.py:205-211: the media tokens are added tokens
    return tok


def reference_functions():
    import torch
    import transformers
    from typing import Any, Dict, List, Optional, Sequence
    mod = ast.parse(open(REF).read())
    names = ("tokenize_conversation", "_maybe_add_sentinel_token", "preprocess_conversation", "infer_stop_tokens")
    keep = [n for n in mod.body if (isinstance(n, ast.FunctionDef) and n.name in names)
            or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "DUMMY_CONVERSATION")]
    assert len(keep) == 5
    auto = object()
    ns = {"torch": torch, "transformers": transformers, "Any": Any, "Dict": Dict, "List": List, "Optional": Optional, "Sequence": Sequence,
          "IGNORE_INDEX": -100, "SENTINEL_TOKEN": "<vila/sentinel>", "logger": logging.getLogger("reference"),
          "tokenizer_image_token": lambda prompt, tokenizer, return_tensors=None: tokenizer(prompt, return_tensors=return_tensors).input_ids[0],
          "conversation_lib": SimpleNamespace(default_conversation=SimpleNamespace(sep_style=auto), SeparatorStyle=SimpleNamespace(AUTO=auto))}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns


def main():
    ns = reference_functions()
    template = reference_chat_template()
    tok = build_tokenizer(chat_template=template)
    serialised = tok.backend_tokenizer.to_str()
    out = {"tokenizer": serialised, "chat_template_name": "qwen2", "conversations": CONVERSATIONS, "cases": []}
    # the reference template's own rendering: plain turns, a conversation that opens with a system turn, a turn without content (left out)
    out["turns"] = [[{"role": "user", "content": "hi"}], [{"role": "system", "content": ""}, {"role": "user", "content": " a\nb "}, {"role": "assistant", "content": "x"}],
                    [{"role": "system", "content": "be brief"}, {"role": "user", "content": "q"}], [{"role": "user", "content": "q"}, {"role": "assistant", "content": None}],
                    [{"role": "user", "content": "<image>\nq"}, {"role": "assistant", "content": "a"}, {"role": "user", "content": "q2"}, {"role": "assistant", "content": "a2"}]]
    out["rendered"] = [[tok.apply_chat_template(t, add_generation_prompt=g, tokenize=False) for g in (False, True)] for t in out["turns"]]
    for conv in CONVERSATIONS:
        case = {}
        for key, kw in (("plain", {}), ("gen", {"add_generation_prompt": True}), ("nosys", {"no_system_prompt": True}),
                        ("override", {"overrides": {"gpt": "answer"}})):
            case[f"ids_{key}"] = ns["tokenize_conversation"](copy.deepcopy(conv), tok, **kw).tolist()
        for key, kw in (("sft", {}), ("sft_nosys", {"no_system_prompt": True})):
            r = ns["preprocess_conversation"](copy.deepcopy(conv), tok, **kw)
            case[f"{key}_ids"], case[f"{key}_labels"] = r["input_ids"].tolist(), r["labels"].tolist()
        out["cases"].append(case)
        sup = [t for t in case["sft_labels"] if t != -100]
        print(f"{len(case['ids_plain'])} ids, {len(sup)} supervised: {tok.decode(sup)!r}")
    # what `generate_content` feeds the model (llava_arch.py:843, 921): ONE human turn + the generation prompt
    out["prompts"] = ["<image>what is in this picture ?", "  describe the image in detail , please .\n", "<image>\n<image>\n<image>\ntwo  spaces", "hello"]
    out["prompt_ids"] = [ns["tokenize_conversation"]([{"from": "human", "value": p}], tok, add_generation_prompt=True).tolist() for p in out["prompts"]]
    out["stop_tokens"] = sorted(ns["infer_stop_tokens"](tok))
    out["sentinel_id"] = int(tok.sentinel_token_id)
    print("stop tokens:", out["stop_tokens"])
    # a tokenizer that glues a space in front of the sentinel cannot be built from a byte-level BPE; the retry branch (retried=True) is covered
    # on the HIP side by a hand-made template (tests/test_conversation_cpu.py)
    json.dump(out, open(OUT, "w"), ensure_ascii=False)
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes)")


if __name__ == "__main__":
    main()
