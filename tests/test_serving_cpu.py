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

    def generate(self, input_ids, media, max_new_tokens, eos_token_id, media_config=None, **sampling):      # llava_arch.py:823-829
        self.calls.append((input_ids, media, max_new_tokens, eos_token_id))
        self.media_config = media_config
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


def test_prompt_split_equals_the_reference_executed_extract_media():
    """tests/golden/prompt_split_ref.json = the reference's own `extract_media` (llava/utils/media.py:93-122, ast-extracted and executed with
    the reference's MEDIA_TOKENS / make_list / Image / Video; oracle/make_golden_prompt.py) on prompts with images, a 3-frame video, typed
    media tokens and whitespace: `_split_prompt` rewrites the text identically and collects the same number of images."""
    import json
    import os
    from vila_amd.serving import MEDIA_TOKENS, Video, _split_prompt
    path = os.path.join(os.path.dirname(__file__), "golden", "prompt_split_ref.json")
    fx = json.load(open(path))
    assert fx["media_tokens"] == MEDIA_TOKENS
    assert len(fx["cases"]) >= 8
    for c in fx["cases"]:
        parts = [object() if p == "IMG" else Video([object()] * int(p[3:])) if p.startswith("VID") else p for p in c["parts"]]
        text, images = _split_prompt(parts)
        assert text == c["text"], (c["parts"], text, c["text"])
        assert len(images) == c["n_images"]


def test_video_frame_selection_equals_the_reference_executed_load_video(tmp_path):
    """tests/golden/video_sampling_ref.json = the reference's own `_load_video` (llava/utils/media.py:39-86, ast-extracted, executed over a
    stand-in capture that serves index-stamped frames, and over real PNG directories; oracle/make_golden_video_sampling.py): evenly spread
    frames, the fps rule with its clamp, containers that report more frames than they hold, a zero-fps container, repeated indices."""
    import json
    import os
    from PIL import Image
    from vila_amd.serving import load_video_frames, video_frame_indices
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "video_sampling_ref.json")))
    assert len(fx["file"]) >= 9 and len(fx["dir"]) >= 3
    for c in fx["file"]:
        got = video_frame_indices(c["grabbable"], c["num_frames"], c["fps"], c["video_fps"])
        assert got == c["indices"], (c, got)
    for c in fx["dir"]:
        d = tmp_path / f"frames_{c['n_files']}_{c['num_frames']}"
        d.mkdir()
        for i in range(c["n_files"]):
            Image.new("RGB", (2, 2), (i, 0, 0)).save(d / f"frame_{i:04d}.png")
        frames = load_video_frames(str(d), num_frames=c["num_frames"])
        assert [f.getpixel((0, 0))[0] for f in frames] == c["indices"]
    # the OpenCV branch of load_video_frames over the SAME stand-in capture the reference ran on (this image has no cv2: it is injected)
    import sys
    import numpy as np_
    from oracle.make_golden_video_sampling import fake_cv2
    for c in fx["file"]:
        sys.modules["cv2"] = fake_cv2(c["frame_count"], c["grabbable"], c["video_fps"])
        try:
            if c["grabbable"] == 0:
                continue
            frames = load_video_frames("clip.mp4", num_frames=c["num_frames"], fps=c["fps"])
        finally:
            del sys.modules["cv2"]
        assert [int(np_.asarray(f)[0, 0, 2]) + 256 * int(np_.asarray(f)[0, 0, 1]) for f in frames] == c["indices"], c
    clip = tmp_path / "clip.mp4"
    clip.write_bytes(b"not a video")
    try:
        import cv2  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="OpenCV"):
            load_video_frames(str(clip))


def test_video_url_content_becomes_one_image_token_per_frame(monkeypatch):
    """server.py:47-52, 214-221: a `video_url` part carries `frames` / `fps`; its frames reach the model as images, one `<image>` each
    (extract_media).  The decoder is stubbed (no OpenCV here); a non-base64 URL is refused, local paths included."""
    import base64
    from fastapi.testclient import TestClient
    tok = _Tok()
    m = _Model(tok)
    seen = {}

    def fake_frames(path, num_frames=8, fps=0.0):
        seen.update(path=path, num_frames=num_frames, fps=fps, body=open(path, "rb").read())
        return [np.zeros((56, 56, 3), np.uint8)] * 3
    monkeypatch.setattr(serving, "load_video_frames", fake_frames)
    client = TestClient(serving.create_app(m, tok, "NVILA-8B"))
    url = "data:video/mp4;base64," + base64.b64encode(b"fake mp4 bytes").decode()
    body = {"model": "NVILA-8B", "temperature": 0.0,
            "messages": [{"role": "user", "content": [{"type": "video_url", "video_url": {"url": url}, "frames": 3, "fps": 0}, {"type": "text", "text": "what happens ?"}]}]}
    r = client.post("/chat/completions", json=body)
    assert r.status_code == 200, r.text
    ids, media = m.calls[-1][0], m.calls[-1][1]
    assert int((ids == m.cfg.image_token_id).sum()) == 3 and len(media["image"]) == 3
    assert seen["num_frames"] == 3 and seen["fps"] == 0.0 and seen["path"].endswith(".mp4") and seen["body"] == b"fake mp4 bytes"
    import os
    assert not os.path.exists(os.path.dirname(seen["path"]))         # the request's temp directory does not outlive its decoding
    monkeypatch.setattr(serving, "load_video_frames", lambda path, num_frames=8, fps=0.0: [])
    assert client.post("/chat/completions", json=body).status_code != 200          # a request whose video yields no frame is refused
    body["messages"][0]["content"][0]["video_url"]["url"] = "/etc/passwd"
    r = client.post("/chat/completions", json=body)
    assert r.status_code == 500 and "Invalid video url" in r.json()["error"]


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


# ----------------------------------------------------------------------------------------------------------------------
# continuous batching (server.py:171-290): the scheduler over a stub engine — rows join and leave between steps
# ----------------------------------------------------------------------------------------------------------------------
class _StubEngine:
    """What `HipBatchEngine` offers, minus the GPU: a request "name:n" replies tokens 100+i (i < n) then EOS; every slot — live or idle —
    advances on every step, exactly like the batched kernel."""
    eos = {1}

    def __init__(self, n_slots=4, step_sleep=0.0):
        import threading
        self.n_slots, self.step_sleep = n_slots, step_sleep
        self.script = {b: None for b in range(n_slots)}
        self.n_out = [0] * n_slots
        self.out = [[] for _ in range(n_slots)]
        self.threads, self.max_live, self.calls = set(), 0, []
        self._threading = threading

    def _mark(self, what):
        self.threads.add(self._threading.get_ident())
        self.calls.append(what)

    def embed(self, prompt, system):
        self._mark("embed")
        from types import SimpleNamespace
        n = int(prompt.split(":")[1])
        return SimpleNamespace(shape=(7,), script=[100 + i for i in range(n)] + [1] + [55] * 64)

    def fits(self, n_prompt, max_new):
        return max_new <= 48

    def admit(self, slot, e):
        self._mark("admit")
        self.script[slot], self.n_out[slot], self.out[slot] = e.script, 0, []
        return e.script[0]

    def run(self, k):
        import time as _t
        self._mark("run")
        self.max_live = max(self.max_live, sum(s is not None for s in self.script.values()))
        for _ in range(k):
            _t.sleep(self.step_sleep)
            for b in range(self.n_slots):
                sc = self.script[b]
                self.out[b].append(sc[1 + self.n_out[b]] if sc is not None else 7)
                self.n_out[b] += 1

    def read(self):
        self._mark("read")
        top = max(max(self.n_out), 1)
        return list(self.n_out), [r[:top] + [0] * (top - len(r)) for r in self.out]

    def release(self, slots):
        self._mark("release")
        for b in slots:
            self.n_out[b], self.out[b] = 0, []
            if b not in self._live:
                self.script[b] = None

    _live = ()

    def solo(self, prompt, max_new_tokens, system, **gen):
        self._mark("solo")
        assert all(s is None for s in self.script.values()) or not any(self.n_out), "a solo request ran beside live rows"
        return f"solo {prompt} t={gen.get('temperature')}"

    def decode(self, toks):
        return " ".join(str(t) for t in toks)


def _want(n, max_new=48):
    return " ".join(str(100 + i) for i in range(min(n, max_new)))


def test_continuous_batcher_admits_a_late_request_between_steps_and_retires_rows_at_eos():
    import threading
    import time as _t
    eng = _StubEngine(n_slots=4, step_sleep=0.002)
    b = serving.ContinuousBatcher(eng, max_batch=4, chunk=4)
    # the engine must know which rows are live when told to re-wind the idle ones
    real_release = eng.release

    def release(slots):
        eng._live = ()
        real_release(slots)
    eng.release = release
    try:
        fa = b.submit("A:30", 48)
        while not any(ev[0] == "run" for ev in b.events):
            _t.sleep(0.001)
        fb = b.submit("B:5", 48)                                       # arrives while A is mid-reply
        assert fb.result(timeout=30) == _want(5) and fa.result(timeout=30) == _want(30)
        admits = [ev for ev in b.events if ev[0] == "admit"]
        assert len(admits) == 2 and admits[1][2] > 0 and admits[1][3] == 1, admits      # B joined after steps had run, beside one live row
        retire = [ev for ev in b.events if ev[0] == "retire"]
        assert retire[0][1] == admits[1][1] and retire[0][2] < retire[1][2]             # B (5 tokens) left first, A went on
        assert eng.max_live == 2
        # max_new_tokens cuts a reply; the row is retired at that length
        assert b.submit("C:40", 10).result(timeout=30) == _want(40, 10)
        # more requests than rows: slots are handed on as rows retire
        futs = [b.submit(f"R{i}:{3 + 2 * i}", 48) for i in range(9)]
        assert [f.result(timeout=60) for f in futs] == [_want(3 + 2 * i) for i in range(9)]
        assert eng.max_live <= 4 and len([ev for ev in b.events if ev[0] == "admit"]) == 12
        # a sampled request and one too long for the slots run SOLO on the same worker thread, after the live rows drained, in arrival order
        f1 = b.submit("L:20", 48)
        f2 = b.submit("S:4", 16, temperature=0.7, top_p=0.8)
        f3 = b.submit("T:4", 200)                                      # does not fit the slots' reply capacity
        f4 = b.submit("U:4", 48)
        assert f2.result(timeout=30) == "solo S:4 t=0.7" and f3.result(timeout=30).startswith("solo T:4")
        assert f1.result(timeout=30) == _want(20) and f4.result(timeout=30) == _want(4)
        assert eng.threads == b.thread_ids and threading.get_ident() not in eng.threads and len(eng.threads) == 1
    finally:
        b.close()


def test_continuous_batcher_returns_the_slot_when_admission_fails_and_survives_a_failing_decode():
    """a prefill that raises after the slot was taken used to leak the slot — after max_batch such failures the
    free list was empty, nothing was live and the worker spun while every greedy request hung; a tokenizer error in the post-chunk loop
    killed the worker thread.  Now the slot goes back, only the failing request fails, and a dead worker fails what it held."""
    eng = _StubEngine(n_slots=2, step_sleep=0.0)
    real_admit, real_decode = eng.admit, eng.decode

    def admit(slot, e):
        if e.script[0] == 100 + 0 and len(e.script) == 3 + 1 + 64:     # prompts "X:3" fail in the prefill
            raise RuntimeError("HIP out of memory (stub)")
        return real_admit(slot, e)

    def decode(toks):
        if len(toks) == 6:                                             # replies of 6 tokens fail in the tokenizer
            raise ValueError("tokenizer (stub)")
        return real_decode(toks)
    eng.admit, eng.decode = admit, decode
    b = serving.ContinuousBatcher(eng, max_batch=2, chunk=4)
    try:
        for _ in range(5):                                             # more failures than there are slots
            with pytest.raises(RuntimeError, match="out of memory"):
                b.submit("X:3", 48).result(timeout=30)
        assert len([ev for ev in b.events if ev[0] == "admit_failed"]) == 5
        assert b.submit("A:9", 48).result(timeout=30) == _want(9)     # the slots are still there
        fa, fb, fc = b.submit("B:6", 48), b.submit("C:11", 48), b.submit("D:4", 48)
        with pytest.raises(ValueError, match="tokenizer"):
            fa.result(timeout=30)                                      # only this request fails ...
        assert fb.result(timeout=30) == _want(11) and fc.result(timeout=30) == _want(4)   # ... its neighbours and successors go on
        assert b._thread.is_alive()
        # a worker that dies fails everything it held and everything submitted afterwards instead of hanging
        eng.read = lambda: (_ for _ in ()).throw(KeyboardInterrupt())  # not an Exception: escapes the per-chunk handler
        fd = b.submit("E:20", 48)
        with pytest.raises(RuntimeError, match="worker died"):
            fd.result(timeout=30)
        with pytest.raises(RuntimeError, match="worker died"):
            b.submit("F:3", 48).result(timeout=30)
        # ADVICE round 5, the race: a request that passes `submit`'s first check while the worker is alive, and whose put lands BEHIND the dying
        # worker's drain, used to sit in a queue nobody reads.  Replayed deterministically: the first check sees a live worker, the put "loses the
        # race" (the queue is already drained and `dead` set by the time it returns) — the re-check behind the put must fail the request.
        err, real_put = b.dead, b._q.put
        assert err is not None
        b.dead = None
        def late_put(item):
            real_put(item)
            b.dead = err
        b._q.put = late_put
        with pytest.raises(RuntimeError, match="worker died"):
            b.submit("G:5", 48).result(timeout=5)
    finally:
        b._q.put = real_put
        b.close()


def test_endpoint_routes_every_request_through_the_one_worker_thread():
    """with a batcher configured, sampled requests (the default, temperature 0.2) used to run inline on the event-loop thread
    beside the worker's batch.  Now every request is executed by the batcher's thread."""
    pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient
    import threading
    tok = _Tok()
    m = _Model(tok)
    m.llm = None                                                       # no batched step on the stub: the static-window batcher + its model lock
    seen = []
    orig = m.generate

    def generate(*a, **k):
        seen.append(threading.get_ident())
        return orig(*a, **k)
    m.generate = generate
    app = serving.create_app(m, tok, model_name="stub", batch_window_s=0.01)
    assert isinstance(app.state.batcher, serving.RequestBatcher)
    client = TestClient(app)
    body = {"model": "stub", "max_tokens": 4, "messages": [{"role": "user", "content": "what is this ?"}]}
    assert client.post("/chat/completions", json=body).status_code == 200                              # sampled (default temperature)
    assert app.state.batcher.model_lock.acquire(timeout=1)             # the lock the sampled request ran under is free again
    app.state.batcher.model_lock.release()
    app.state.batcher.close()


# ----------------------------------------------------------------------------------------------------------------------
# streaming (server.py:241-270): tokens leave as they are generated — TextStream == HF's TextIteratorStreamer, SSE framing, the batcher
# ----------------------------------------------------------------------------------------------------------------------
def _bpe_tokenizer():
    """A byte-level BPE tokenizer trained here on a few lines (no files, no network): sub-word pieces, spaces inside tokens, newlines,
    CJK — everything the streamer's release rule looks at."""
    tokenizers = pytest.importorskip("tokenizers")
    transformers = pytest.importorskip("transformers")
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tk = Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    corpus = ["the quick brown fox jumps over the lazy dog", "a red square sits on a blue table\nnext to a green circle",
              "streaming replies leave the server word by word", "图片里有一个红色的方块", "hello world, hello again!  two  spaces"] * 4
    tk.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=320, special_tokens=["<|im_end|>"], initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    return transformers.PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|im_end|>")


def test_text_stream_equals_hf_text_iterator_streamer_piece_by_piece():
    """The release rule (up to the last space; everything after a newline or a CJK character; the rest at the end) against the class the
    reference's server iterates (`transformers.TextIteratorStreamer`, server.py:22) — same token sequences, token by token, every piece
    (the empty ones included) equal and in order."""
    tok = _bpe_tokenizer()
    from transformers import TextIteratorStreamer
    texts = ["the quick brown fox", "a red square\nnext to a green circle\n", "图片里有一个红色的方块 and a fox", "hello  world,  two  spaces!", "x",
             "streaming replies leave the server word by word"]
    eos = tok.eos_token_id
    for text in texts:
        ids = tok(text, add_special_tokens=False).input_ids + [eos]
        assert len(ids) > len(text.split())                                  # sub-word pieces: words DO span tokens
        hf = TextIteratorStreamer(tok, skip_special_tokens=True)
        mine = serving.TextStream(tok)
        for t in ids:
            hf.put(torch.tensor([t]))
            mine.put(torch.tensor([t]))
        hf.end()
        mine.end()
        want, got = list(hf), list(mine)
        assert got == want, (text, got, want)
        assert "".join(got) == tok.decode(ids, skip_special_tokens=True) == text
        assert mine.token_ids == ids
    # lists, nested one-row lists and whole chunks are taken token by token too; two rows are refused like HF does
    a, b = serving.TextStream(tok), serving.TextStream(tok)
    ids = tok("the quick brown fox jumps", add_special_tokens=False).input_ids
    for t in ids:
        a.put([t])
    b.put([ids])
    a.end(), b.end()
    assert list(a) == list(b)
    with pytest.raises(ValueError, match="batch size 1"):
        serving.TextStream(tok).put([[1, 2], [3, 4]])
    # a failure reaches the consumer; a producer that never streamed hands over the whole reply
    s = serving.TextStream(tok)
    s.put(ids[:2])
    s.fail(RuntimeError("boom"))
    with pytest.raises(RuntimeError, match="boom"):
        list(s)
    s = serving.TextStream(tok)
    s.finish("whole reply")
    assert "".join(s) == "whole reply"


def test_sse_chunks_follow_the_reference_chunk_generator():
    """server.py:243-268: a lone space is held back and prepended, a trailing stop string is cut (and the piece stripped), empty pieces send
    nothing, `[DONE]` closes."""
    ev = list(serving.sse_chunks(["", "a ", " ", "red", "", " square <|im_end|>", ""], "M"))
    assert ev[-1] == "data: [DONE]\n\n" and all(e.startswith("data: ") and e.endswith("\n\n") for e in ev)
    body = [json.loads(e[6:]) for e in ev[:-1]]
    assert [b["choices"][0]["delta"]["content"] for b in body] == ["a ", " red", "square"]
    assert all(b["object"] == "chat.completion.chunk" and b["model"] == "M" for b in body)


def test_continuous_batcher_streams_a_row_while_it_decodes():
    """A streamed request's pieces arrive while its row is still live (first token at admission, then after every chunk of steps), stop at the
    EOS, and add up to the reply the future gets; a solo request streams through the engine's streamer or, failing that, as one piece."""
    import time as _t

    class _WordTok:
        def decode(self, ids, skip_special_tokens=True):
            return " ".join(str(i) for i in ids if not (skip_special_tokens and i == 1))
    eng = _StubEngine(n_slots=2, step_sleep=0.01)
    b = serving.ContinuousBatcher(eng, max_batch=2, chunk=4)
    try:
        st = serving.TextStream(_WordTok(), timeout=30)
        fut = b.submit("A:21", 48, stream=st)
        first = next(st)                                                  # blocks until the admission's token was put
        early = not fut.done()                                            # 21 tokens x 10 ms per step: the row is still decoding
        pieces = [first] + list(st)
        assert early, "the first piece only arrived with the finished reply"
        assert "".join(pieces) == fut.result(timeout=30) == _want(21)
        assert st.token_ids == [100 + i for i in range(21)] + [1]         # every token once, the EOS included (HF puts it too), nothing after it
        # max_new_tokens cuts the stream where it cuts the reply
        st = serving.TextStream(_WordTok(), timeout=30)
        fut = b.submit("B:30", 10, stream=st)
        assert "".join(st) == fut.result(timeout=30) == _want(30, 10) and len(st.token_ids) == 10
        # a solo request (sampled): the stub engine takes no streamer -> the whole reply as one piece
        st = serving.TextStream(_WordTok(), timeout=30)
        fut = b.submit("S:4", 16, stream=st, temperature=0.7)
        assert "".join(st) == fut.result(timeout=30) == "solo S:4 t=0.7"
        # a failing request fails its stream too
        st = serving.TextStream(_WordTok(), timeout=30)
        fut = b.submit("broken", 16, stream=st)                           # the stub's embed() cannot parse this prompt
        with pytest.raises(Exception):
            list(st)
        assert fut.exception(timeout=30) is not None
    finally:
        b.close()


def test_endpoint_streams_tokens_from_the_generate_call():
    """`stream=True` without a batcher: the generate call runs on its own thread (the reference's Thread + TextIteratorStreamer) and the
    endpoint forwards the pieces as they are put — here a model that feeds its streamer one token at a time."""
    pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient
    tok = _Tok()

    class _StreamingModel(_Model):
        def generate(self, input_ids, media, max_new_tokens, eos_token_id, media_config=None, streamer=None, **sampling):
            reply = self.tok("a red square on a table").input_ids + [1]
            self.streamed = streamer is not None
            for t in reply:
                if streamer is not None:
                    streamer.put(torch.tensor([t]))
            if streamer is not None:
                streamer.end()
            return torch.tensor([reply])
    m = _StreamingModel(tok)
    client = TestClient(serving.create_app(m, tok, model_name="stub"))
    body = {"model": "stub", "max_tokens": 8, "temperature": 0.0, "stream": True, "messages": [{"role": "user", "content": "what is this ?"}]}
    r = client.post("/chat/completions", json=body)
    assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream") and m.streamed
    events = [e for e in r.text.split("\n\n") if e]
    words = [json.loads(e[6:])["choices"][0]["delta"]["content"] for e in events[:-1]]
    assert events[-1] == "data: [DONE]" and len(words) == 6 and "".join(words) == "a red square on a table"
    # an error before any text is the reference's 500 body, not a broken stream
    r = client.post("/chat/completions", json=dict(body, model="other"))
    assert r.status_code == 500 and "configured to use the model" in r.json()["error"]

    class _Failing(_Model):
        def generate(self, *a, **k):
            raise RuntimeError("tower on fire")
    r = TestClient(serving.create_app(_Failing(tok), tok, model_name="stub")).post("/chat/completions", json=body)
    assert r.status_code == 500 and "tower on fire" in r.json()["error"]
