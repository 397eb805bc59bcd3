"""Serving shim over the REAL HIP model: `generate_content` and POST /chat/completions drive tower -> projector
-> splice -> prefill -> hipGraph decode on the GPU, with a real `tokenizers` / `PreTrainedTokenizerFast` tokenizer (built in memory:
no network), and the reply is checked against the CPU oracle's greedy ids for the same prompt under the margin-aware id rule.
Reference seams: `LlavaLlamaModel.generate_content` (llava/model/llava_arch.py:836-948), `server.py:171-290`."""
import base64
import io
import json

import numpy as np
import pytest
import torch

from oracle import vila_oracle as O
from tests.gpu_util import margin_aware_ids
from vila_amd import configs, serving, synthetic

pytestmark = pytest.mark.gpu


def _tokenizer(cfg):
    tokenizers = pytest.importorskip("tokenizers")
    transformers = pytest.importorskip("transformers")
    words = ["<unk>", "<|im_start|>", "<|im_end|>", "system", "user", "assistant", "what", "is", "this", "?", "describe", "the", "image", "a",
             "red", "square", "you", "are", "helpful"]
    words += [f"w{i}" for i in range(min(cfg.image_token_id, cfg.video_token_id, cfg.llm.eos_token_id) - len(words))]
    vocab = {w: i for i, w in enumerate(words)}                 # ids 0..996: every id the tiny model can emit except its media / eos ids
    tk = tokenizers.Tokenizer(tokenizers.models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = tokenizers.pre_tokenizers.WhitespaceSplit()
    tok = transformers.PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", eos_token="<|im_end|>")
    assert max(vocab.values()) < min(cfg.image_token_id, cfg.video_token_id, cfg.llm.eos_token_id)
    return tok


@pytest.fixture(scope="module")
def served():
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 12).items()}
    model = build_model(cfg, weights=w)
    return cfg, w, model, _tokenizer(cfg)


def _image():
    g = np.random.default_rng(3)
    return g.integers(0, 256, size=(56, 56, 3), dtype=np.uint8)       # already the tower's resolution: preprocessing is exact


def _oracle_reply(cfg, w, tok, parts, n):
    text, images = serving._split_prompt(parts)
    ids = serving.encode_with_images(tok, serving.chat_text(text), cfg.image_token_id)
    px = [serving.preprocess_image(im, cfg.vision.image_size).to(torch.bfloat16).float() for im in images]
    ids_o, lg_o = O.vlm_generate(px, ids, w, cfg, n, stop_at_eos=False)
    return ids, px, ids_o, lg_o


def test_generate_content_on_the_hip_model_follows_the_oracle(served):
    cfg, w, model, tok = served
    n = 6
    parts = [_image(), "what is this ?"]
    ids, px, ids_o, lg_o = _oracle_reply(cfg, w, tok, parts, n)
    # the GPU path teacher-forced with the oracle's ids: margin-aware bit-exact ids ...
    e, _, _ = model._embed(ids[None], {"image": [p.to(torch.bfloat16).cuda() for p in px]})
    _, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=ids_o, use_graph=False)
    # ... and the served reply (free-running hipGraph decode inside generate_content) decodes the same tokens up to the first
    # non-decisive step
    reply = serving.generate_content(model, tok, parts, max_new_tokens=n, eos_token_id=-1)
    free = torch.tensor(tok(reply, add_special_tokens=False).input_ids)
    decisive = margin_aware_ids(lg, lg_o, ids_o)
    nd = (~decisive).nonzero().flatten()
    k = int(nd[0]) if nd.numel() else n
    want = tok.decode(ids_o[:k].tolist(), skip_special_tokens=True).split()
    assert reply.split()[:len(want)] == want, (reply, want, free.tolist(), ids_o.tolist())


def test_chat_completions_endpoint_on_the_hip_model(served):
    pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient
    from PIL import Image
    cfg, w, model, tok = served
    arr = _image()
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG")
    url = "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()
    client = TestClient(serving.create_app(model, tok, model_name="NVILA-tiny"))
    # temperature 0 = greedy (do_sample = temperature > 0, server.py:185-187); a request WITHOUT the field samples at the reference's
    # default temperature 0.2 / top_p 0.9 (server.py:101-102) — checked below
    body = {"model": "NVILA-tiny", "max_tokens": 5, "temperature": 0.0,
            "messages": [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": url}}, {"type": "text", "text": "describe the image"}]}]}
    r = client.post("/chat/completions", json=body)
    assert r.status_code == 200, r.text
    text = r.json()["choices"][0]["message"]["content"][0]["text"]
    direct = serving.generate_content(model, tok, [arr, "describe the image"], max_new_tokens=5)
    assert text == direct                                   # the endpoint is generate_content + the OpenAI envelope; PNG round trip is lossless
    r = client.post("/chat/completions", json=dict(body, stream=True))
    events = [l for l in r.text.split("\n\n") if l]
    assert events[-1] == "data: [DONE]"
    assert "".join(json.loads(ev[6:])["choices"][0]["delta"]["content"] for ev in events[:-1]).strip() == text
    sampled = client.post("/chat/completions", json={k: v for k, v in body.items() if k != "temperature"})
    assert sampled.status_code == 200 and isinstance(sampled.json()["choices"][0]["message"]["content"][0]["text"], str)
    # text-only request and a second image in one request also go through the HIP path
    r = client.post("/chat/completions", json={"model": "NVILA-tiny", "max_tokens": 3, "messages": [{"role": "user", "content": "what is this ?"}]})
    assert r.status_code == 200 and isinstance(r.json()["choices"][0]["message"]["content"][0]["text"], str)


def test_batched_serving_shares_the_weight_pass(served):
    """`generate_content_batch` / `RequestBatcher` (server.py:171-290: concurrent requests): three greedy prompts of different lengths and image
    counts as ONE padded batch through the batched decode step; identical prompts give identical replies, a second call reproduces the batch, and the endpoint with a batching window answers like the plain one."""
    pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient
    cfg, w, model, tok = served
    img = _image()
    prompts = [[img, "what is this ?"], "describe the image", [img, "what is this ?"], [img, "a red square", img, "you are helpful"]]
    batch = serving.generate_content_batch(model, tok, prompts, max_new_tokens=5, eos_token_id=-1)
    assert getattr(model.llm, "_bdecode", None) is not None, "the batched decode step was not taken"
    assert len(batch) == 4 and batch[0] == batch[2] and all(isinstance(t, str) and t for t in batch)
    # (rows against their solo runs under the margin rule: tests/test_gpu_batch_decode.py — a plain string comparison here would hinge on
    # non-decisive argmaxes of the tiny model)
    again = serving.generate_content_batch(model, tok, prompts, max_new_tokens=5, eos_token_id=-1)
    assert again == batch                                            # deterministic
    # through the request batcher: submitted together -> one batch of 4
    b = serving.RequestBatcher(model, tok, window_s=0.5, max_batch=8,
                               run=lambda ps, n, system: serving.generate_content_batch(model, tok, ps, max_new_tokens=n, system=system, eos_token_id=-1))
    try:
        futs = [b.submit(p, 5) for p in prompts]
        assert [f.result(timeout=60) for f in futs] == batch
        assert b.batches == [4]
    finally:
        b.close()
    # the endpoint with a batcher: requests go through the continuous batcher (ONE worker thread owns the model); a lone greedy request is a
    # live row of the batched step, a sampled one runs solo on the same thread
    app = serving.create_app(model, tok, model_name="NVILA-tiny", batch_window_s=0.01, max_batch=4)
    cb = app.state.batcher
    assert isinstance(cb, serving.ContinuousBatcher)
    client = TestClient(app)
    ask = {"model": "NVILA-tiny", "max_tokens": 4, "temperature": 0.0, "messages": [{"role": "user", "content": "what is this ?"}]}
    r = client.post("/chat/completions", json=ask)
    assert r.status_code == 200, r.text
    text = r.json()["choices"][0]["message"]["content"][0]["text"]
    assert isinstance(text, str) and text
    assert cb.submit("what is this ?", 4, temperature=0.0).result(timeout=60) == text          # deterministic
    r = client.post("/chat/completions", json={k: v for k, v in ask.items() if k != "temperature"})        # sampled: solo, same thread
    assert r.status_code == 200 and any(ev[0] == "solo" for ev in cb.events)
    # a late request joins the running batch: submit a long reply, then a short one once steps have run
    import time as _t
    n0 = len(cb.events)
    fa = cb.submit([img, "describe the image"], 40)
    while not any(ev[0] == "run" for ev in cb.events[n0:]):
        _t.sleep(0.001)
    fb = cb.submit("a red square", 6)
    tb, ta = fb.result(timeout=60), fa.result(timeout=60)
    assert isinstance(ta, str) and isinstance(tb, str)
    admits = [ev for ev in cb.events[n0:] if ev[0] == "admit"]
    assert len(admits) == 2 and admits[1][2] > admits[0][2]                                   # B was admitted after steps of A had run
    assert len(cb.thread_ids) == 1
    cb.close()


def test_streamed_reply_equals_the_plain_one(served):
    """`llm.generate(streamer=...)`: the tokens the host learns of every 16 graph replays reach a `TextStream` — same ids as the plain call, pieces
    that add up to the reply; the endpoint's `stream=True` forwards them (server.py:241-270)."""
    pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient
    cfg, w, model, tok = served
    parts = [_image(), "describe the image"]
    n = 40
    plain = serving.generate_content(model, tok, parts, max_new_tokens=n, eos_token_id=-1)
    st = serving.TextStream(tok)
    reply = serving.generate_content(model, tok, parts, max_new_tokens=n, eos_token_id=-1, streamer=st)
    pieces = list(st)
    assert reply == plain and len(st.token_ids) == n and "".join(pieces).strip() == plain
    assert len([p for p in pieces if p]) > 4                              # word by word, not one lump
    client = TestClient(serving.create_app(model, tok, model_name="NVILA-tiny"))
    body = {"model": "NVILA-tiny", "max_tokens": 12, "temperature": 0.0, "stream": True, "messages": [{"role": "user", "content": "what is this ?"}]}
    r = client.post("/chat/completions", json=body)
    events = [e for e in r.text.split("\n\n") if e]
    text = "".join(json.loads(e[6:])["choices"][0]["delta"]["content"] for e in events[:-1])
    assert r.status_code == 200 and events[-1] == "data: [DONE]"
    assert text.strip() == serving.generate_content(model, tok, "what is this ?", max_new_tokens=12)
