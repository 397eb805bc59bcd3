"""Serving shim (vila_amd/serving.py) on CPU with a stub model: prompt assembly, image pre-processing, endpoint schema."""
import base64
import io
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from vila_amd import configs, serving


class _Tok:
    """Whitespace tokenizer over a growing vocabulary (stands in for the checkpoint's tokenizer)."""
    eos_token_id = 1

    def __init__(self):
        self.vocab = {"<eos>": 1}
        self.inv = {1: "<eos>"}

    def __call__(self, text, add_special_tokens=False):
        ids = []
        for w in text.split():
            if w not in self.vocab:
                self.vocab[w] = len(self.vocab) + 2
                self.inv[self.vocab[w]] = w
            ids.append(self.vocab[w])
        return SimpleNamespace(input_ids=ids)

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(self.inv[i] for i in ids if not (skip_special_tokens and i == 1))


class _Model:
    """Records what the shim hands to `generate` and answers with fixed ids."""
    def __init__(self, tok):
        self.cfg = configs.tiny()
        self.device = torch.device("cpu")
        self.tok = tok
        self.calls = []

    def generate(self, input_ids, media, max_new_tokens, eos_token_id, **sampling):
        self.calls.append((input_ids, media, max_new_tokens, eos_token_id))
        self.sampling = sampling
        reply = self.tok("a red square").input_ids + [1, 99]       # EOS then a token that must be dropped
        return torch.tensor([reply])


def _png_data_url(arr):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG")
    return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()


def test_preprocess_image_matches_siglip_processor_semantics():
    arr = (np.arange(100 * 80 * 3) % 256).astype(np.uint8).reshape(100, 80, 3)
    x = serving.preprocess_image(arr, 56)
    assert x.shape == (3, 56, 56) and x.dtype == torch.float32
    assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0
    same = serving.preprocess_image(np.full((56, 56, 3), 255, np.uint8), 56)      # no resize: exact (1 - 0.5) / 0.5
    assert torch.equal(same, torch.ones(3, 56, 56))


def test_generate_content_builds_ids_media_and_decodes():
    tok = _Tok()
    m = _Model(tok)
    img = np.zeros((70, 70, 3), np.uint8)
    out = serving.generate_content(m, tok, [img, "what is this ?"], max_new_tokens=7)
    assert out == "a red square"
    ids, media, n, eos = m.calls[0]
    assert n == 7 and eos == 1
    assert int((ids == m.cfg.image_token_id).sum()) == 1 and len(media["image"]) == 1
    assert media["image"][0].shape == (3, m.cfg.vision.image_size, m.cfg.vision.image_size) and media["image"][0].dtype == torch.bfloat16
    # the <image> id sits where the part was, inside the user turn
    text_ids = tok("<|im_start|>user").input_ids
    pos = int((ids[0] == m.cfg.image_token_id).nonzero()[0])
    assert ids[0, :pos].tolist()[-len(text_ids):] == text_ids


def test_chat_completions_endpoint_schema_and_errors():
    fastapi = pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient
    tok = _Tok()
    m = _Model(tok)
    client = TestClient(serving.create_app(m, tok, model_name="NVILA-8B"))
    url = _png_data_url(np.full((20, 30, 3), 128, np.uint8))
    body = {"model": "NVILA-8B", "max_tokens": 16,
            "messages": [{"role": "user", "content": [{"type": "text", "text": "describe"}, {"type": "image_url", "image_url": {"url": url}}]}]}
    r = client.post("/chat/completions", json=body)
    assert r.status_code == 200
    j = r.json()
    assert j["object"] == "chat.completion" and j["choices"][0]["message"]["content"][0] == {"type": "text", "text": "a red square"}
    assert len(m.calls[-1][1]["image"]) == 1
    r = client.post("/chat/completions", json=dict(body, stream=True))
    events = [l for l in r.text.split("\n\n") if l]
    assert events[-1] == "data: [DONE]" and json.loads(events[0][6:])["object"] == "chat.completion.chunk"
    assert "".join(json.loads(e[6:])["choices"][0]["delta"]["content"] for e in events[:-1]).strip() == "a red square"
    r = client.post("/chat/completions", json=dict(body, model="other"))
    assert r.status_code == 500 and "configured to use the model" in r.json()["error"]
    # temperature > 0 samples (server.py:185-187): do_sample with the request's temperature / top_p reaches generate()
    r = client.post("/chat/completions", json=dict(body, temperature=0.7, top_p=0.8))
    assert r.status_code == 200 and m.sampling["do_sample"] is True
    assert abs(m.sampling["temperature"] - 0.7) < 1e-6 and abs(m.sampling["top_p"] - 0.8) < 1e-6 and m.sampling["top_k"] == 50
    # a request without the fields gets the REFERENCE's defaults (server.py:101-102: temperature 0.2, top_p 0.9 -> sampled)
    r = client.post("/chat/completions", json=body)
    assert r.status_code == 200 and m.sampling["do_sample"] is True
    assert abs(m.sampling["temperature"] - 0.2) < 1e-6 and abs(m.sampling["top_p"] - 0.9) < 1e-6
    r = client.post("/chat/completions", json=dict(body, temperature=0.0))
    assert r.status_code == 200 and m.sampling == {}                  # an explicit temperature 0 stays greedy (do_sample = temperature > 0)


def test_prompt_split_matches_extract_media():
    """llava/utils/media.py:93-122: one `<image>` per image part and nothing else (the "\\n" is an embedding appended by the encoder);
    media tokens typed inside a text part are removed and the part stripped."""
    from vila_amd.serving import _split_prompt
    img = object()
    assert _split_prompt(["look: ", img, "what is it?"]) == ("look: <image>what is it?", [img])
    assert _split_prompt("a literal <image> token") == ("a literal  token", [])
    text, images = _split_prompt(["<image> describe", img])
    assert text == "describe<image>" and images == [img]
    assert _split_prompt("a typed <vila/video> token") == ("a typed  token", [])     # every MEDIA_TOKENS value, not only <image>


class _BatchModel(_Model):
    """Answers a padded batch: row b replies with the b-th canned text, then EOS, then padding."""
    texts = ["a red square", "w1 w2", "you are helpful"]

    def generate(self, input_ids, media, max_new_tokens, eos_token_id, attention_mask=None, pad_token_id=None, **kw):
        self.calls.append((input_ids, media, max_new_tokens, eos_token_id, attention_mask, pad_token_id))
        rows = [self.tok(self.texts[b % len(self.texts)]).input_ids + [1] for b in range(input_ids.shape[0])]
        n = max(len(r) for r in rows) + 1
        return torch.tensor([r + [int(pad_token_id)] * (n - len(r)) for r in rows])


def test_generate_content_batch_pads_rows_and_consumes_images_in_row_order():
    tok = _Tok()
    m = _BatchModel(tok)
    img = np.zeros((56, 56, 3), np.uint8)
    out = serving.generate_content_batch(m, tok, [[img, "what is this ?"], "describe", [img, "a", img, "b c d e"]], max_new_tokens=9)
    assert out == ["a red square", "w1 w2", "you are helpful"]
    ids, media, n, eos, mask, pad = m.calls[0]
    assert ids.shape == mask.shape and ids.shape[0] == 3 and n == 9 and eos == 1
    assert [int((ids[b] == m.cfg.image_token_id).sum()) for b in range(3)] == [1, 0, 2] and len(media["image"]) == 3
    lens = mask.sum(1).tolist()
    assert lens[2] == ids.shape[1] and lens[1] < lens[0] < lens[2]                   # right padded to the longest row
    for b in range(3):
        assert bool(mask[b, : lens[b]].all()) and not bool(mask[b, lens[b]:].any())
        assert (ids[b, lens[b]:] == pad).all()
    assert serving.generate_content_batch(m, tok, []) == []


def test_request_batcher_groups_concurrent_requests_and_splits_on_settings():
    """Requests submitted together share a batch (<= max_batch); another max_new_tokens starts its own; a failure reaches every waiter."""
    import threading
    seen = []
    gate = threading.Event()

    def run(prompts, n, system):
        gate.wait(2)
        seen.append((list(prompts), n, system))
        if prompts[0] == "boom":
            raise RuntimeError("model failed")
        return [f"{p}:{n}" for p in prompts]

    b = serving.RequestBatcher(None, None, window_s=0.2, max_batch=3, run=run)
    try:
        futs = [b.submit(f"p{i}", 8) for i in range(5)] + [b.submit("q", 4), b.submit("boom", 2)]
        gate.set()
        got = [f.result(timeout=5) for f in futs[:6]]
        assert got == ["p0:8", "p1:8", "p2:8", "p3:8", "p4:8", "q:4"]
        with pytest.raises(RuntimeError):
            futs[6].result(timeout=5)
        sizes = [len(s[0]) for s in seen]
        assert sizes[0] == 3 and sum(sizes[:2]) == 5 and all(s[1] == 8 for s in seen[:2])      # 5 requests, max_batch 3 -> 3 + 2
        assert [s[1] for s in seen[2:]] == [4, 2]                                               # other settings never share a batch
        assert b.batches[:2] == [3, 2]
    finally:
        b.close()
