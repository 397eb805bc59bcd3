"""Serving shim: `generate_content` + an OpenAI-style `/chat/completions` endpoint over the HIP model.

Mirrors the parts of the reference a client sees:
  * `LlavaLlamaModel.generate_content(prompt)` (llava/model/llava_arch.py:836-948): a prompt is a string or a list of parts (strings and
    images); media are extracted in order, pre-processed, the text gets one `<image>` token per image, greedy generation, decoded text
  * `server.py:171-290`: POST /chat/completions with OpenAI-style messages (`text` / `image_url` parts, base64 data URLs), the response
    object layout, `stream=True` as server-sent `chat.completion.chunk` events
What stays in the reference: conversation templates beyond the Qwen2 chat form, video decoding, structured output (xgrammar).
The tokenizer is whatever the checkpoint ships (`transformers.AutoTokenizer`), handed in by the caller; images are pre-processed the way
`SiglipImageProcessor` does for NVILA (resize to the tower's resolution, bicubic, rescale 1/255, normalise mean = std = 0.5).
"""
from __future__ import annotations

import base64
import io
import json
import re
import time
import uuid
from typing import Any, Dict, Iterator, List, Optional, Sequence, Union

import numpy as np
import torch

IMAGE_TOKEN = "<image>"
MEDIA_TOKENS = {"image": "<image>", "video": "<vila/video>"}          # llava/constants.py:32-35
_DATA_URL = re.compile(r"^data:image/(png|jpe?g);base64,(.*)$", re.S)


def preprocess_image(img, size: int) -> torch.Tensor:
    """PIL image / HxWx3 uint8 array / 3xHxW float tensor in [0,1] -> [3, size, size] float32 in [-1, 1] (SigLIP processor semantics)."""
    if isinstance(img, torch.Tensor):
        x = img.float()
        if x.dim() == 3 and x.shape[0] != 3 and x.shape[-1] == 3:
            x = x.permute(2, 0, 1)
        if x.max() > 1.5:
            x = x / 255.0
    else:
        arr = np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)
        x = torch.from_numpy(arr.copy()).float().permute(2, 0, 1) / 255.0
    if x.shape[-2:] != (size, size):
        x = torch.nn.functional.interpolate(x[None], size=(size, size), mode="bicubic", align_corners=False, antialias=True)[0].clamp(0, 1)
    return (x - 0.5) / 0.5


def load_image(url: str):
    """`data:image/...;base64,` URLs (server.py:56-75).  Remote http(s) fetches are left to the caller (no network in this build)."""
    m = _DATA_URL.match(url)
    if not m:
        raise ValueError("only base64 data URLs (data:image/png|jpeg;base64,...) are supported by this shim")
    from PIL import Image
    return Image.open(io.BytesIO(base64.b64decode(m.group(2)))).convert("RGB")


def _split_prompt(prompt: Union[str, Sequence[Any]]):
    """llava/utils/media.py:93-122 (extract_media): text with exactly one `<image>` per image part (the "\n" after an image is NOT
    text: BasicImageEncoder appends it as an embedding, encoders/image/basic.py:22-27), images in order; media tokens typed by the
    user inside a text part are removed and the part stripped (:104-108)."""
    text, images = "", []
    for part in ([prompt] if isinstance(prompt, str) else prompt):
        if isinstance(part, str):
            for token in MEDIA_TOKENS.values():                       # every media token, the video one included (media.py:103-106)
                if token in part:
                    part = part.replace(token, "").strip()
            text += part
        else:
            images.append(part)
            text += IMAGE_TOKEN
    return text, images


def chat_text(text: str, system: Optional[str] = None) -> str:
    """Qwen2 chat form with the generation prompt appended (tokenize_conversation(add_generation_prompt=True), llava/utils/tokenizer.py)."""
    s = f"<|im_start|>system\n{system}<|im_end|>\n" if system else ""
    return s + f"<|im_start|>user\n{text}<|im_end|>\n<|im_start|>assistant\n"


def encode_with_images(tokenizer, text: str, image_token_id: int) -> torch.Tensor:
    """tokenizer_image_token (llava/mm_utils.py): tokenize the pieces between <image> tags, put the media id in between."""
    ids: List[int] = []
    for i, piece in enumerate(text.split(IMAGE_TOKEN)):
        if i:
            ids.append(image_token_id)
        if piece:
            ids.extend(tokenizer(piece, add_special_tokens=False).input_ids if callable(tokenizer) else tokenizer.encode(piece))
    return torch.tensor(ids, dtype=torch.int64)


def generate_content(model, tokenizer, prompt: Union[str, Sequence[Any]], max_new_tokens: int = 128, system: Optional[str] = None,
                     eos_token_id=None, device: Optional[str] = None, temperature: float = 0.0, top_p: float = 1.0, top_k: int = 50,
                     seed: Optional[int] = None) -> str:
    """Text + images in, decoded reply out — the contract of `LlavaLlamaModel.generate_content` for image / text prompts.
    temperature > 0 samples (server.py:185-187: do_sample = temperature > 0, with the request's top_p and HF's default top_k = 50)."""
    text, images = _split_prompt(prompt)
    cfg = model.cfg
    dev = device or str(model.device)
    ids = encode_with_images(tokenizer, chat_text(text, system), cfg.image_token_id)[None].to(dev)
    media = {"image": [preprocess_image(im, cfg.vision.image_size).to(device=dev, dtype=torch.bfloat16) for im in images]}
    eos = eos_token_id if eos_token_id is not None else getattr(tokenizer, "eos_token_id", None)
    gen = dict(max_new_tokens=max_new_tokens, eos_token_id=eos)
    if temperature and temperature > 0:
        gen.update(do_sample=True, temperature=float(temperature), top_p=float(top_p), top_k=int(top_k), seed=seed)
    out = model.generate(input_ids=ids, media=media, **gen)
    toks = out[0].tolist()
    stop = set(eos) if isinstance(eos, (list, tuple)) else {eos}
    for k, t in enumerate(toks):                       # HF returns the EOS as the last token; decode(skip_special_tokens) drops it
        if t in stop:
            toks = toks[:k]
            break
    return tokenizer.decode(toks, skip_special_tokens=True).strip()


def generate_content_batch(model, tokenizer, prompts: Sequence[Union[str, Sequence[Any]]], max_new_tokens: int = 128, system: Optional[str] = None,
                           eos_token_id=None, pad_token_id: Optional[int] = None, device: Optional[str] = None) -> List[str]:
    """Several greedy requests as ONE padded batch (server.py:171-290 serves concurrent requests; here they share every pass over the
    weights: `vila_llm_decode_step_batch`, up to 16 rows).  Each prompt is tokenised on its own, rows are right-padded, the images of all
    prompts are encoded by one tower call and consumed in row order (`_embed`, llava_arch.py:454-466).  Returns one decoded reply per prompt."""
    if not prompts:
        return []
    cfg = model.cfg
    dev = device or str(model.device)
    rows, images = [], []
    for prompt in prompts:
        text, imgs = _split_prompt(prompt)
        rows.append(encode_with_images(tokenizer, chat_text(text, system), cfg.image_token_id))
        images.extend(imgs)
    eos = eos_token_id if eos_token_id is not None else getattr(tokenizer, "eos_token_id", None)
    stop = set(eos) if isinstance(eos, (list, tuple)) else {eos}
    pad = pad_token_id if pad_token_id is not None else (getattr(tokenizer, "pad_token_id", None) or 0)
    L = max(int(r.numel()) for r in rows)
    ids = torch.full((len(rows), L), int(pad), dtype=torch.int64)
    mask = torch.zeros((len(rows), L), dtype=torch.bool)
    for b, r in enumerate(rows):
        ids[b, : r.numel()] = r
        mask[b, : r.numel()] = True
    media = {"image": [preprocess_image(im, cfg.vision.image_size).to(device=dev, dtype=torch.bfloat16) for im in images]}
    out = model.generate(input_ids=ids.to(dev), media=media, attention_mask=mask.to(dev), max_new_tokens=max_new_tokens, eos_token_id=eos,
                         pad_token_id=int(pad))
    replies = []
    for row in out.tolist():
        toks = row
        for k, t in enumerate(row):
            if t in stop:
                toks = row[:k]
                break
        replies.append(tokenizer.decode(toks, skip_special_tokens=True).strip())
    return replies


class RequestBatcher:
    """Groups greedy requests that arrive within `window_s` of each other (at most `max_batch` <= 16) into one `generate_content_batch` call.
    `submit` returns a `concurrent.futures.Future` of the reply; one worker thread owns the model.  Requests with different `max_new_tokens`
    or system prompts do not share a batch (the batch runs to the longest reply; a row that stops early is padded like HF does)."""

    def __init__(self, model, tokenizer, window_s: float = 0.005, max_batch: int = 16, run=None):
        import queue
        import threading
        self.model, self.tokenizer, self.window_s, self.max_batch = model, tokenizer, float(window_s), int(max(1, min(16, max_batch)))
        self._run = run or (lambda prompts, n, system: generate_content_batch(self.model, self.tokenizer, prompts, max_new_tokens=n, system=system))
        self._q: "queue.Queue" = queue.Queue()
        self.batches: List[int] = []                     # sizes of the batches that ran (observability / tests)
        self._stop = False
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()

    def submit(self, prompt, max_new_tokens: int = 128, system: Optional[str] = None):
        from concurrent.futures import Future
        f: Future = Future()
        self._q.put((prompt, int(max_new_tokens), system, f))
        return f

    def close(self):
        self._stop = True
        self._q.put(None)
        self._thread.join(timeout=5)

    def _loop(self):
        import queue
        held = None
        while not self._stop:
            item = held if held is not None else self._q.get()
            held = None
            if item is None:
                break
            group = [item]
            deadline = time.time() + self.window_s
            while len(group) < self.max_batch:
                try:
                    nxt = self._q.get(timeout=max(0.0, deadline - time.time()))
                except queue.Empty:
                    break
                if nxt is None:
                    self._stop = True
                    break
                if nxt[1] != item[1] or nxt[2] != item[2]:           # other generation settings: it starts the next group
                    held = nxt
                    break
                group.append(nxt)
            try:
                with torch.inference_mode():
                    replies = self._run([g[0] for g in group], item[1], item[2])
                self.batches.append(len(group))
                for g, r in zip(group, replies):
                    g[3].set_result(r)
            except Exception as e:                                    # every waiter of the group sees the failure
                for g in group:
                    g[3].set_exception(e)


_MODELS = None


def _request_models():
    """pydantic request schema (server.py:38-110), built once at module scope so FastAPI can resolve the annotations."""
    global _MODELS
    if _MODELS is None:
        import pydantic
        ChatMessage = pydantic.create_model("ChatMessage", role=(str, ...), content=(Union[str, List[Dict[str, Any]]], ...))
        ChatCompletionRequest = pydantic.create_model(
            "ChatCompletionRequest", model=(str, ...), messages=(List[ChatMessage], ...), max_tokens=(Optional[int], 512),
            temperature=(Optional[float], 0.2), top_p=(Optional[float], 0.9), stream=(Optional[bool], False))      # server.py:101-102
        _MODELS = (ChatMessage, ChatCompletionRequest)
    return _MODELS


def create_app(model, tokenizer, model_name: str = "NVILA-8B", batch_window_s: Optional[float] = None):
    """FastAPI app with the reference's POST /chat/completions (server.py:171-290).  Import-time optional: needs fastapi + pydantic.
    batch_window_s: when set, greedy requests (temperature 0) that arrive within that window share a batched decode (`RequestBatcher`)."""
    from fastapi import FastAPI
    from fastapi.responses import JSONResponse, StreamingResponse

    ChatMessage, ChatCompletionRequest = _request_models()
    app = FastAPI()
    batcher = RequestBatcher(model, tokenizer, window_s=batch_window_s) if batch_window_s is not None else None
    app.state.batcher = batcher

    def _prompt_of(messages):
        parts: List[Any] = []
        system = None
        for m in messages:
            if m.role == "system" and isinstance(m.content, str):
                system = m.content
            elif m.role == "user":
                if isinstance(m.content, str):
                    parts.append(m.content)
                else:
                    for c in m.content:
                        if c.get("type") == "text":
                            parts.append(c["text"])
                        elif c.get("type") == "image_url":
                            parts.append(load_image(c["image_url"]["url"]))
                        else:
                            raise NotImplementedError(f"Unsupported content type: {c.get('type')}")
            elif m.role == "assistant" and isinstance(m.content, str):
                parts.append(f"<|im_end|>\n<|im_start|>assistant\n{m.content}<|im_end|>\n<|im_start|>user\n")
        return parts, system

    async def chat_completions(request):
        try:
            if request.model != model_name:
                raise ValueError(f"The endpoint is configured to use the model {model_name}, but the request model is {request.model}")
            parts, system = _prompt_of(request.messages)
            temperature = request.temperature if request.temperature is not None else 0.2
            if batcher is not None and not temperature:
                import asyncio
                fut = batcher.submit(parts, request.max_tokens or 512, system)
                text = await asyncio.get_running_loop().run_in_executor(None, fut.result)
            else:
                with torch.inference_mode():
                    text = generate_content(model, tokenizer, parts, max_new_tokens=request.max_tokens or 512, system=system,
                                            temperature=temperature, top_p=request.top_p if request.top_p is not None else 0.9)
            if request.stream:
                def chunks() -> Iterator[str]:
                    for i, word in enumerate(re.findall(r"\S+\s*", text)):
                        yield "data: " + json.dumps({"id": str(i), "object": "chat.completion.chunk", "created": time.time(), "model": request.model,
                                                     "choices": [{"delta": {"content": word}}]}) + "\n\n"
                    yield "data: [DONE]\n\n"
                return StreamingResponse(chunks())
            return {"id": uuid.uuid4().hex, "object": "chat.completion", "created": time.time(), "model": request.model,
                    "choices": [{"message": {"role": "assistant", "content": [{"type": "text", "text": text}]}}]}
        except Exception as e:                                       # server.py:292-297: errors come back as a 500 JSON body
            return JSONResponse(status_code=500, content={"error": str(e)})

    chat_completions.__annotations__["request"] = ChatCompletionRequest      # a real class, not a string (postponed annotations)
    app.post("/chat/completions")(chat_completions)
    return app
