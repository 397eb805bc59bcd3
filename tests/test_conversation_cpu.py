"""The text-side input producer (vila_amd/conversation.py) against the REFERENCE-EXECUTED fixture: `tokenize_conversation`,
`preprocess_conversation` and `infer_stop_tokens` taken out of llava/utils/tokenizer.py with ast and executed over the byte-identical
tokenizer (oracle/make_golden_conversation.py).  Integer work: bit-exact."""
import copy
import json
import os

import pytest
import torch

pytest.importorskip("tokenizers")
pytest.importorskip("transformers")

from oracle.make_golden_conversation import build_tokenizer  # noqa: E402
from vila_amd import conversation as C  # noqa: E402
from vila_amd.configs import IGNORE_INDEX  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "conversation_ref.json")


@pytest.fixture(scope="module")
def fx():
    return json.load(open(GOLDEN))


def _tok(fx):
    """The fixture's tokenizer with the HIP side's OWN `qwen2` template (the reference-executed ids were produced under the reference's file)."""
    return C.prepare_tokenizer(build_tokenizer(fx["tokenizer"]), fx["chat_template_name"])


def test_qwen2_chat_template_renders_like_the_reference_file(fx):
    """`--chat_template qwen2` (language_model/builder.py:194-200): our restated template against the reference file's own rendering — default
    system turn without a full stop, a leading system turn kept, a turn with `None` content left out, the generation prompt."""
    tok = _tok(fx)
    for turns, (plain, gen) in zip(fx["turns"], fx["rendered"]):
        assert tok.apply_chat_template(turns, add_generation_prompt=False, tokenize=False) == plain
        assert tok.apply_chat_template(turns, add_generation_prompt=True, tokenize=False) == gen
    assert fx["rendered"][0][0].startswith("<|im_start|>system\nYou are a helpful assistant<|im_end|>\n<|im_start|>user\nhi")
    with pytest.raises(ValueError, match="unknown chat template"):
        C.prepare_tokenizer(build_tokenizer(fx["tokenizer"]), "vicuna_v9")
    # the builder's other additions: stop tokens off the template, media tokens as added tokens with their ids
    assert tok.stop_tokens == ["<|im_end|>"] and tok.stop_token_ids == [tok.eos_token_id]
    assert tok.media_token_ids == {"image": tok.convert_tokens_to_ids("<image>"), "video": tok.convert_tokens_to_ids("<vila/video>")}


def test_tokenize_conversation_every_switch_bit_exact(fx):
    tok = _tok(fx)
    for conv, case in zip(fx["conversations"], fx["cases"]):
        for key, kw in (("plain", {}), ("gen", {"add_generation_prompt": True}), ("nosys", {"no_system_prompt": True}),
                        ("override", {"overrides": {"gpt": "answer"}})):
            got = C.tokenize_conversation(copy.deepcopy(conv), tok, **kw)
            assert got.dtype == torch.int64 and got.tolist() == case[f"ids_{key}"], key
    # media tokens are ONE id each, wherever they stand; the messages are stripped in place like the reference does
    conv = copy.deepcopy(fx["conversations"][1])
    ids = C.tokenize_conversation(conv, tok).tolist()
    assert ids.count(tok.convert_tokens_to_ids("<vila/video>")) == 1 and conv[0]["value"] == "describe the image in detail , please ."
    with pytest.raises(ValueError, match="Unexpected sender 'system'"):
        C.tokenize_conversation([{"from": "system", "value": "x"}], tok)


def test_preprocess_conversation_labels_bit_exact(fx):
    tok = _tok(fx)
    for conv, case in zip(fx["conversations"], fx["cases"]):
        for key, kw in (("sft", {}), ("sft_nosys", {"no_system_prompt": True})):
            r = C.preprocess_conversation(copy.deepcopy(conv), tok, **kw)
            assert r["input_ids"].tolist() == case[f"{key}_ids"] and r["labels"].tolist() == case[f"{key}_labels"], key
            lab = r["labels"]
            assert bool(((lab == IGNORE_INDEX) | (lab == r["input_ids"])).all())
    assert int(tok.sentinel_token_id) == fx["sentinel_id"]
    # the supervised tokens of the first conversation are the reply and its end-of-turn token, nothing of the prompt or the system turn
    r = C.preprocess_conversation(copy.deepcopy(fx["conversations"][0]), tok)
    sup = r["labels"][r["labels"] != IGNORE_INDEX]
    assert tok.decode(sup) == "a red square on a blue table<|im_end|>"


def test_infer_stop_tokens(fx):
    tok = _tok(fx)
    assert sorted(C.infer_stop_tokens(tok)) == fx["stop_tokens"] == ["<|im_end|>"]


def test_frame_matching_retry_and_failure():
    """tokenizer.py:143-168 on hand-made rows: the token after each sentinel is dropped from the frame (and so supervised); when the frame is
    not used up the token in FRONT of each sentinel goes too; when even that fails nothing is supervised."""
    S = 99
    assert C._frame_of([1, 2, S, 7, 3, S, 7, 4], S, False) == [1, 2, 3, 4]
    assert C._frame_of([1, 2, S, 7, 3, S, 7, 4], S, True) == [1, 4]
    assert C._frame_of([1, S], S, False) == [1, S]                          # a sentinel in the last position is never looked at
    free, used = C._supervised([1, 2, 50, 51, 7, 3, 60, 7, 4], [1, 2, 3, 4])
    assert used and free == [False, False, True, True, True, False, True, True, False]
    free, used = C._supervised([1, 5, 50, 7, 4], [1, 2, 4])
    assert not used                                                          # 2 never shows up: the frame is stuck -> retry

    class _Tok:                                                             # a tokenizer whose reply merges with the token in front of it
        sentinel_token, sentinel_token_id, eos_token = "<vila/sentinel>", S, "</s>"

        def apply_chat_template(self, turns, add_generation_prompt=False, tokenize=False):
            return turns

        def __call__(self, turns, return_tensors=None):
            ids = []
            for t in turns:
                if t["role"] == "user":
                    ids += [1, 10]
                elif t["content"] == "<vila/sentinel>":
                    ids += [20, S, 7]                                       # 20 = a space token in front of the sentinel ...
                else:
                    ids += [21, 30, 7]                                      # ... that fuses with the reply's first piece into 21
            return type("E", (), {"input_ids": torch.tensor([ids])})()
    conv = [{"from": "human", "value": "q"}, {"from": "gpt", "value": "a"}]
    r = C.preprocess_conversation(conv, _Tok())
    assert r["input_ids"].tolist() == [1, 10, 21, 30, 7] and r["labels"].tolist() == [-100, -100, 21, 30, 7]     # second attempt: frame [1, 10]

    class _Bad(_Tok):
        def __call__(self, turns, return_tensors=None):
            if any(t["content"] == "<vila/sentinel>" for t in turns):
                return type("E", (), {"input_ids": torch.tensor([[1, 10, 20, S, 7, 77]])})()            # 77 is nowhere in the real row
            return type("E", (), {"input_ids": torch.tensor([[1, 10, 21, 30, 7]])})()
    r = C.preprocess_conversation(conv, _Bad())
    assert r["labels"].tolist() == [-100] * 5


def test_serving_prompt_ids_equal_the_reference_generate_content_prompt(fx):
    """`generate_content` tokenises ONE human turn with the generation prompt appended (llava_arch.py:843, 921): the serving shim's ids for
    the same text — through the tokenizer's own chat template, Qwen2's default system turn included — are the reference-executed ones."""
    from vila_amd import serving
    tok = _tok(fx)
    image_id = tok.convert_tokens_to_ids("<image>")
    for text, want in zip(fx["prompts"], fx["prompt_ids"]):
        got = serving.encode_with_images(tok, serving.prompt_text(tok, text.strip()), image_id)      # prepare_prompt strips, like tokenizer.py:77-78
        assert got.tolist() == want, text
    assert "You are a helpful assistant<|im_end|>" in serving.prompt_text(tok, "hello")
    assert serving.prompt_text(tok, "hello", system="be brief").count("system") == 1 and "be brief" in serving.prompt_text(tok, "hello", system="be brief")
    # a tokenizer without a template gets the bare chat form
    tok.chat_template = None
    assert serving.prompt_text(tok, "hello") == "<|im_start|>user\nhello<|im_end|>\n<|im_start|>assistant\n"
