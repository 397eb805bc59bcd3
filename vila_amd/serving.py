"""Serving shim: `generate_content` + an OpenAI-style `/chat/completions` endpoint over the HIP model.

Mirrors the parts of the reference a client sees:
  * `LlavaLlamaModel.generate_content(prompt)` (llava/model/llava_arch.py:836-948): a prompt is a string or a list of parts (strings and
    images); media are extracted in order, pre-processed, the text gets one `<image>` token per image, greedy generation, decoded text
  * `server.py:171-290`: POST /chat/completions with OpenAI-style messages (`text` / `image_url` / `video_url` parts, base64 data URLs), the
    response object layout, `stream=True` as server-sent `chat.completion.chunk` events that leave WHILE the reply is generated
    (`TextStream` = `transformers.TextIteratorStreamer`'s release rule, fed by `llm.generate(streamer=...)` or by the continuous batcher)
  * `process_image` / `process_images` (llava/mm_utils.py:442-541): every `image_aspect_ratio` mode — resize, pad, dynamic (the NVILA-Lite
    tiler), dynamic_s2 (the NVILA tiler)
What stays in the reference: conversation templates beyond the Qwen2 chat form, video FILE decoding (OpenCV), structured output (xgrammar).
The tokenizer is whatever the checkpoint ships (`transformers.AutoTokenizer`), handed in by the caller; images are pre-processed the way
`SiglipImageProcessor` does for NVILA (resize to the tower's resolution, bicubic, rescale 1/255, normalise mean = std = 0.5).
"""
from __future__ import annotations

import base64
import glob
import io
import itertools
import json
import os
import re
import threading
import time
import uuid
from types import SimpleNamespace
from typing import Any, Dict, Iterator, List, Optional, Sequence, Union

import numpy as np
import torch

IMAGE_TOKEN = "<image>"
MEDIA_TOKENS = {"image": "<image>", "video": "<vila/video>"}          # llava/constants.py:32-35
_DATA_URL = re.compile(r"^data:image/(png|jpe?g);base64,(.*)$", re.S)


def _to_pil(img):
    """PIL image / HxWx3 uint8 array / 3xHxW (or HxWx3) tensor in [0,1] or [0,255] -> PIL RGB image."""
    from PIL import Image
    if hasattr(img, "convert"):
        return img.convert("RGB")                                    # mm_utils.py:452
    if isinstance(img, torch.Tensor):
        x = img.detach().float().cpu()
        if x.dim() == 3 and x.shape[0] == 3 and x.shape[-1] != 3:
            x = x.permute(1, 2, 0)
        if x.max() <= 1.5:
            x = x * 255.0
        img = x.round().clamp(0, 255).to(torch.uint8).numpy()
    return Image.fromarray(np.ascontiguousarray(np.asarray(img, dtype=np.uint8)), "RGB")


def preprocess_image(img, size: int) -> torch.Tensor:
    """-> [3, size, size] float32 in [-1, 1], the pixels `SiglipImageProcessor.preprocess` hands the tower (mm_utils.py:442-541, row a1): PIL
    bicubic resize to size x size (skipped when the image already has that size), rescale by 1/255, normalise with mean = std = 0.5.
    PIL does the resampling — the reference's input producer is kept as it is, so the pixels ARE the reference's
    (tests/test_s2_tiler_cpu.py holds this function to HF-processor-executed vectors)."""
    from PIL import Image
    pil = _to_pil(img)
    if pil.size != (size, size):
        pil = pil.resize((size, size), resample=Image.BICUBIC)
    x = torch.from_numpy(np.asarray(pil, dtype=np.uint8).copy()).permute(2, 0, 1).float()
    return (x * (1.0 / 255.0) - 0.5) / 0.5


def process_image(img, cfg, enable_dynamic_res: bool = False, enable_dynamic_s2: bool = False, max_tiles: Optional[int] = None):
    """`process_image` (mm_utils.py:442-523) for the SigLIP tower, `cfg.aspect_mode` standing for `data_args.image_aspect_ratio`:
      * "dynamic_s2" + enable_dynamic_s2 -> ([n_tiles, 3, S, S], block_size): the tiles of every scale (`dynamic_s2_preprocess`)
      * "dynamic*"   + enable_dynamic_res -> [n_tiles, 3, S, S]: the grid closest to the picture's aspect ratio + a thumbnail
        (`dynamic_preprocess`, min_tiles .. max_tiles, `max_tiles=` overrides the config's)
      * "resize" -> PIL resize to S x S; "pad" -> centred on a square of the processor's mean colour (`expand2square`), then the processor
      * anything else -> the processor's default, which for SigLIP is the resize
    every tile / picture through `preprocess_image` (= `SiglipImageProcessor.preprocess`).  Pinned by tests/golden/dynamic_tiles.npz = the
    reference's own function executed on the same pictures."""
    from .host import dynamic_preprocess, dynamic_s2_preprocess, expand2square
    pil = _to_pil(img)
    size = cfg.vision.image_size
    mode = cfg.aspect_mode
    if "dynamic_s2" in mode and enable_dynamic_s2:
        tiles, block_size = dynamic_s2_preprocess(pil, list(cfg.s2_scales), int(cfg.max_tiles), size)
        return torch.stack([preprocess_image(t, size) for t in tiles]), block_size
    if "dynamic" in mode and enable_dynamic_res:
        max_num = int(max_tiles) if max_tiles is not None else int(cfg.max_tiles)
        tiles = dynamic_preprocess(pil, min_num=int(cfg.min_tiles), max_num=max_num, image_size=size)
        return torch.stack([preprocess_image(t, size) for t in tiles])
    if mode == "resize":
        pil = pil.resize((size, size))
    if mode == "pad":
        pil = expand2square(pil, tuple(int(x * 255) for x in (0.5, 0.5, 0.5)))           # SiglipImageProcessor.image_mean
    return preprocess_image(pil, size)


def process_images(images, cfg, enable_dynamic_res: bool = False, max_tiles: Optional[int] = None) -> torch.Tensor:
    """`process_images` (mm_utils.py:526-541): every picture through `process_image`; tiled pictures ([n, 3, S, S] each) are concatenated,
    whole ones stacked; pictures whose results differ in shape (different tile counts) are refused with the reference's message."""
    new_images = [process_image(im, cfg, enable_dynamic_res=enable_dynamic_res, max_tiles=max_tiles) for im in images]
    if not all(x.shape == new_images[0].shape for x in new_images):
        raise ValueError("The shape of images in new_images is different!")
    return torch.cat(new_images, dim=0) if new_images[0].dim() == 4 else torch.stack(new_images, dim=0)


def preprocess_media(images, cfg):
    """The media half of `generate_content` (llava_arch.py:857-879): ONE image under a tiling recipe becomes its tiles — `dynamic_s2`: the
    tiles of every scale with media_config["image"]["block_sizes"] = [block_size]; `dynamic`: the aspect-ratio grid + thumbnail (the prompt
    then carries one `<image>\n` per tile, see `prepare_prompt`); otherwise every image goes through `process_images` whole.
    -> (list of [3, size, size] float tensors, media_config)."""
    mode = cfg.aspect_mode
    if len(images) == 1 and mode == "dynamic_s2":
        tiles, block_size = process_image(images[0], cfg, enable_dynamic_s2=True)
        return list(tiles), {"image": {"block_sizes": [block_size]}}
    if len(images) == 1 and mode == "dynamic":
        return list(process_image(images[0], cfg, enable_dynamic_res=True)), {}
    return ([] if not images else list(process_images(images, cfg))), {}


def load_image(url: str):
    """`data:image/...;base64,` URLs (server.py:56-75).  Remote http(s) fetches are left to the caller (no network in this build)."""
    m = _DATA_URL.match(url)
    if not m:
        raise ValueError("only base64 data URLs (data:image/png|jpeg;base64,...) are supported by this shim")
    from PIL import Image
    return Image.open(io.BytesIO(base64.b64decode(m.group(2)))).convert("RGB")


class Video:
    """A video prompt part: its already-extracted frames (images, or one [T, H, W, 3] array).  In this snapshot of the reference a `Video` part
    becomes `num_video_frames` `<image>` tokens and its frames join the IMAGE list (llava/utils/media.py:114-119); picking the frames out of a
    file (`_load_video`, cv2) stays with the caller."""

    def __init__(self, frames) -> None:
        self.frames = list(frames)


_VIDEO_URL = re.compile(r"^data:video/(mp4);base64,(.*)$")            # server.py:57


def video_frame_indices(frame_count: int, num_frames: int, fps: float = 0.0, video_fps: float = 0.0) -> List[int]:
    """Which frames of a `frame_count`-frame video the reference keeps (llava/utils/media.py:62-72): with `fps > 0` one frame every 1 / fps
    seconds of the video's duration, clamped to `num_frames`; else `num_frames` positions spread evenly over the whole video
    (`np.round(np.linspace(0, n - 1, num_frames))`).  Pinned by tests/golden/video_sampling_ref.json = the reference's own `_load_video`
    executed over a synthetic capture."""
    if fps and fps > 0:
        duration = frame_count / video_fps if video_fps > 0 else 0    # (a container without a frame rate selects nothing: the loader raises)
        stamps = np.arange(0, duration, 1.0 / fps)[:num_frames]
        return [int(t * video_fps) for t in stamps]
    return [int(i) for i in np.round(np.linspace(0, frame_count - 1, num_frames)).astype(int)]


def load_video_frames(path: str, num_frames: int = 8, fps: float = 0.0):
    """`_load_video` (llava/utils/media.py:39-86): a DIRECTORY of frame images (sorted, `num_frames` spread evenly) or a video file (needs
    OpenCV, which this image does not ship: imported lazily, a missing module is reported as such) -> list of PIL images."""
    from PIL import Image
    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "*")))
        if not files:
            raise ValueError(f"Video '{path}' has no frames.")
        return [Image.open(files[i]).convert("RGB") for i in video_frame_indices(len(files), num_frames)]
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("decoding a video FILE needs OpenCV (cv2), as in the reference (llava/utils/media.py:47); pass a directory of frames "
                           "or a serving.Video(frames) part instead") from e
    cap = cv2.VideoCapture(path)
    video_fps = cap.get(cv2.CAP_PROP_FPS)
    n = int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
    while n > 0:                                                     # the container's frame count may overshoot (:53-59)
        cap.set(cv2.CAP_PROP_POS_FRAMES, n - 1)
        if cap.grab():
            break
        n -= 1
    else:
        raise ValueError(f"Video '{path}' has no frames.")
    frames = {}
    idx = video_frame_indices(n, num_frames, fps, video_fps)
    for i in idx:
        if i in frames:
            continue
        cap.set(cv2.CAP_PROP_POS_FRAMES, i)
        ok, frame = cap.read()
        if ok:
            frames[i] = Image.fromarray(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB))
    return [frames[i] for i in idx if i in frames]


def load_video_url_frames(url: str, num_frames: int = 8, fps: float = 0.0):
    """A request's `video_url` -> frames.  The reference writes the body to a temp file that stays (server.py:60-78); here the file lives
    only while it is decoded, so a long-running server does not keep one video per request on disk."""
    import shutil
    path = load_video(url)
    try:
        frames = load_video_frames(path, num_frames=num_frames, fps=fps)
        if not frames:                       # `_load_video` itself returns [] here (media.py:62-86, pinned); a REQUEST with no frames is refused
            raise ValueError("video_url: no frame could be selected (a container that reports no frame rate under the fps rule, or "
                             "unreadable frames)")
        return frames
    finally:
        shutil.rmtree(os.path.dirname(path), ignore_errors=True)


def load_video(url: str) -> str:
    """`load_video` of server.py:60-78: a `data:video/mp4;base64,` URL is written to a temporary .mp4 whose path is returned.  Remote http(s)
    fetches are left to the caller (no network in this build); local paths are NOT accepted from a request (the Python API takes them:
    `load_video_frames(path)` / `Video(frames)`)."""
    m = _VIDEO_URL.match(url)
    if m:
        import tempfile
        path = os.path.join(tempfile.mkdtemp(), f"{uuid.uuid5(uuid.NAMESPACE_DNS, url)}.mp4")
        with open(path, "wb") as f:
            f.write(base64.b64decode(m.group(2)))
        return path
    raise ValueError(f"Invalid video url: {url[:64]}")


def _split_prompt(prompt: Union[str, Sequence[Any]]):
    """llava/utils/media.py:93-122 (extract_media): text with exactly one `<image>` per image part (the "\n" after an image is NOT
    text: BasicImageEncoder appends it as an embedding, encoders/image/basic.py:22-27), images in order; a `Video` part is one `<image>`
    per frame (:114-119); media tokens typed by the user inside a text part are removed and the part stripped (:104-108).
    Pinned by tests/golden/prompt_split_ref.json = the reference's own function executed on the same prompts."""
    text, images = "", []
    for part in ([prompt] if isinstance(prompt, str) else prompt):
        if isinstance(part, str):
            for token in MEDIA_TOKENS.values():                       # every media token, the video one included (media.py:103-106)
                if token in part:
                    part = part.replace(token, "").strip()
            text += part
        elif isinstance(part, Video):
            images.extend(part.frames)
            text += IMAGE_TOKEN * len(part.frames)
        else:
            images.append(part)
            text += IMAGE_TOKEN
    return text, images


def prepare_prompt(prompt: Union[str, Sequence[Any]], cfg):
    """Prompt parts -> (text, tiles, media_config): `extract_media` + the media branch of `generate_content` (llava_arch.py:842-879).  Under
    the `dynamic` recipe a single image's `<image>` becomes one `<image>\n` per tile (:864-866; the dataset path builds the same text,
    mm_utils.py:408-424).  The text is stripped like `tokenize_conversation` does with every message (llava/utils/tokenizer.py:77-78)."""
    text, images = _split_prompt(prompt)
    tiles, media_config = preprocess_media(images, cfg)
    if len(images) == 1 and cfg.aspect_mode == "dynamic":
        text = text.replace(IMAGE_TOKEN, f"{IMAGE_TOKEN}\n" * len(tiles))
    return text.strip(), tiles, media_config


def chat_text(text: str, system: Optional[str] = None) -> str:
    """Qwen2 chat form with the generation prompt appended (tokenize_conversation(add_generation_prompt=True), llava/utils/tokenizer.py)."""
    s = f"<|im_start|>system\n{system}<|im_end|>\n" if system else ""
    return s + f"<|im_start|>user\n{text}<|im_end|>\n<|im_start|>assistant\n"


def prompt_text(tokenizer, text: str, system: Optional[str] = None) -> str:
    """The string the model is prompted with: the tokenizer's OWN chat template with the generation prompt appended, exactly what
    `generate_content` builds (`tokenize_conversation(conversation, tokenizer, add_generation_prompt=True)`, llava_arch.py:921 ->
    llava/utils/tokenizer.py:109-113) — Qwen2's template, for one, supplies a default system turn.  A tokenizer without a template (the
    synthetic stand-ins) gets the bare Qwen2 chat form (`chat_text`)."""
    if getattr(tokenizer, "chat_template", None) and hasattr(tokenizer, "apply_chat_template"):
        turns = ([{"role": "system", "content": system}] if system else []) + [{"role": "user", "content": text}]
        return tokenizer.apply_chat_template(turns, add_generation_prompt=True, tokenize=False)
    return chat_text(text, system)


def encode_with_images(tokenizer, text: str, image_token_id: int) -> torch.Tensor:
    """tokenizer_image_token (llava/mm_utils.py): tokenize the pieces between <image> tags, put the media id in between."""
    ids: List[int] = []
    for i, piece in enumerate(text.split(IMAGE_TOKEN)):
        if i:
            ids.append(image_token_id)
        if piece:
            ids.extend(tokenizer(piece, add_special_tokens=False).input_ids if callable(tokenizer) else tokenizer.encode(piece))
    return torch.tensor(ids, dtype=torch.int64)


def _is_cjk(cp: int) -> bool:
    """transformers/generation/streamers.py `TextStreamer._is_chinese_char`: the CJK Unified Ideographs blocks."""
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F
            or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class TextStream:
    """Token ids in, text pieces out, across threads — what `transformers.TextIteratorStreamer` is to the reference's server
    (server.py:22, 241-270: `for new_text in streamer`).  `put` / `end` follow `TextStreamer.put` / `end` token by token: the cache of
    undecoded tokens is decoded whole; text up to the last space is released (a word may still change with the next token), everything
    after a newline (the cache restarts) or a CJK character.  Pieces — empty ones too, as HF queues them — wait in a queue for the
    consumer; `fail(exc)` makes the consumer raise.  Pinned by tests/test_serving_cpu.py against HF's class run on the same token sequences."""

    _END = object()

    def __init__(self, tokenizer, skip_special_tokens: bool = True, timeout: Optional[float] = None):
        import queue
        self.tokenizer, self.skip_special_tokens, self.timeout = tokenizer, skip_special_tokens, timeout
        self._cache: List[int] = []
        self._print_len = 0
        self._q: "queue.Queue" = queue.Queue()
        self._ended = False
        self.token_ids: List[int] = []                   # everything that was put (observability / tests)

    def _decode(self) -> str:
        return self.tokenizer.decode(self._cache, skip_special_tokens=self.skip_special_tokens)

    def put(self, value) -> None:
        ids = value.tolist() if hasattr(value, "tolist") else value
        if isinstance(ids, int):
            ids = [ids]
        if ids and isinstance(ids[0], (list, tuple)):
            if len(ids) > 1:
                raise ValueError("TextStreamer only supports batch size 1")
            ids = ids[0]
        for t in ids:
            self.token_ids.append(int(t))
            self._cache.append(int(t))
            text = self._decode()
            if text.endswith("\n"):
                piece = text[self._print_len:]
                self._cache, self._print_len = [], 0
            elif len(text) > 0 and _is_cjk(ord(text[-1])):
                piece = text[self._print_len:]
                self._print_len += len(piece)
            else:
                piece = text[self._print_len: text.rfind(" ") + 1]
                self._print_len += len(piece)
            self._q.put(piece)

    def end(self) -> None:
        piece = ""
        if self._cache:
            piece = self._decode()[self._print_len:]
            self._cache, self._print_len = [], 0
        self._q.put(piece)
        self._q.put(self._END)
        self._ended = True

    def finish(self, text: str) -> None:
        """For a producer whose `generate` took no streamer (it never called `put` / `end`): the whole reply as one piece, then the end."""
        if not self._ended:
            if not self.token_ids:
                self._q.put(text)
            self.end()

    def fail(self, exc: BaseException) -> None:
        self._q.put(exc)
        self._ended = True

    def __iter__(self):
        return self

    def __next__(self) -> str:
        item = self._q.get(timeout=self.timeout)
        if item is self._END:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item


def sse_chunks(pieces, model_name: str, stop_str: str = "<|im_end|>") -> Iterator[str]:
    """The `chunk_generator` of server.py:243-268: text pieces -> server-sent `chat.completion.chunk` events.  A lone " " is held back and
    prepended to the next piece; a piece that ends with the stop string loses it (and is stripped); empty pieces send nothing; `[DONE]` closes."""
    prepend_space = False
    chunk_id = 0
    for new_text in pieces:
        if new_text == " ":
            prepend_space = True
            continue
        if stop_str and new_text.endswith(stop_str):
            new_text = new_text[: -len(stop_str)].strip()
            prepend_space = False
        elif prepend_space:
            new_text = " " + new_text
            prepend_space = False
        if len(new_text):
            chunk = {"id": str(chunk_id), "object": "chat.completion.chunk", "created": time.time(), "model": model_name,
                     "choices": [{"delta": {"content": new_text}}]}
            yield f"data: {json.dumps(chunk)}\n\n"
    yield "data: [DONE]\n\n"


def _eos_of(tokenizer, eos_token_id=None):
    """The ids generation stops on: the caller's, else the tokenizer's `stop_token_ids` (what `default_generation_config` hands HF as
    `eos_token_id`, llava_arch.py:961-962: the end-of-turn tokens `infer_stop_tokens` read off the chat template), else its EOS."""
    if eos_token_id is not None:
        return eos_token_id
    stop = getattr(tokenizer, "stop_token_ids", None)
    if stop:
        return list(stop) if len(stop) > 1 else stop[0]
    return getattr(tokenizer, "eos_token_id", None)


def generate_content(model, tokenizer, prompt: Union[str, Sequence[Any]], max_new_tokens: int = 128, system: Optional[str] = None,
                     eos_token_id=None, device: Optional[str] = None, temperature: float = 0.0, top_p: float = 1.0, top_k: int = 50,
                     seed: Optional[int] = None, streamer=None) -> str:
    """Text + images in, decoded reply out — the contract of `LlavaLlamaModel.generate_content` for image / text prompts.
    temperature > 0 samples (server.py:185-187: do_sample = temperature > 0, with the request's top_p and HF's default top_k = 50).
    streamer: a `TextStream` (or any HF streamer) that receives the new tokens while they are generated (`llm.generate(streamer=...)`)."""
    cfg = model.cfg
    dev = device or str(model.device)
    text, tiles, media_config = prepare_prompt(prompt, cfg)
    ids = encode_with_images(tokenizer, prompt_text(tokenizer, text, system), cfg.image_token_id)[None].to(dev)
    media = {"image": [t.to(device=dev, dtype=torch.bfloat16) for t in tiles]}
    eos = _eos_of(tokenizer, eos_token_id)
    gen = dict(max_new_tokens=max_new_tokens, eos_token_id=eos)
    if temperature and temperature > 0:
        gen.update(do_sample=True, temperature=float(temperature), top_p=float(top_p), top_k=int(top_k), seed=seed)
    if streamer is not None:
        gen.update(streamer=streamer)
    out = model.generate(input_ids=ids, media=media, media_config=media_config, **gen)
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
    rows, tiles, blocks = [], [], []
    for prompt in prompts:
        text, t, mc = prepare_prompt(prompt, cfg)             # per request, like generate_content: one image -> its tiles
        rows.append(encode_with_images(tokenizer, prompt_text(tokenizer, text, system), cfg.image_token_id))
        tiles.extend(t)
        blocks.extend(mc.get("image", {}).get("block_sizes", [None] * len(t)))
    eos = _eos_of(tokenizer, eos_token_id)
    stop = set(eos) if isinstance(eos, (list, tuple)) else {eos}
    pad = pad_token_id if pad_token_id is not None else (getattr(tokenizer, "pad_token_id", None) or 0)
    L = max(int(r.numel()) for r in rows)
    ids = torch.full((len(rows), L), int(pad), dtype=torch.int64)
    mask = torch.zeros((len(rows), L), dtype=torch.bool)
    for b, r in enumerate(rows):
        ids[b, : r.numel()] = r
        mask[b, : r.numel()] = True
    media = {"image": [t.to(device=dev, dtype=torch.bfloat16) for t in tiles]}
    media_config = {"image": {"block_sizes": blocks}} if any(b is not None for b in blocks) else {}
    out = model.generate(input_ids=ids.to(dev), media=media, media_config=media_config, attention_mask=mask.to(dev), max_new_tokens=max_new_tokens,
                         eos_token_id=eos, pad_token_id=int(pad))
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
        self.model_lock = threading.Lock()               # whoever drives the model outside the worker (sampled requests) holds this
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
                with self.model_lock, torch.inference_mode():
                    replies = self._run([g[0] for g in group], item[1], item[2])
                self.batches.append(len(group))
                for g, r in zip(group, replies):
                    g[3].set_result(r)
            except Exception as e:                                    # every waiter of the group sees the failure
                for g in group:
                    g[3].set_exception(e)


class HipBatchEngine:
    """The model side of `ContinuousBatcher`: KV slots of ONE open batched-decode session (`HipQwen2ForCausalLM.batch_open`), driven by
    the batcher's worker thread only."""

    def __init__(self, model, tokenizer, n_slots: int = 8, max_ctx: int = 2048, max_new_tokens: int = 1024, eos_token_id=None):
        self.model, self.tokenizer = model, tokenizer
        self.n_slots, self.max_ctx, self.max_new_tokens = int(n_slots), int(max_ctx), int(max_new_tokens)
        eos = _eos_of(tokenizer, eos_token_id)
        self.eos = set(eos) if isinstance(eos, (list, tuple)) else {eos}
        self.st = None

    def _session(self):
        llm = self.model.llm
        if self.st is None or getattr(llm, "_bdecode", None) is not self.st:       # first use, or the weights moved and the session was dropped
            self.st = llm.batch_open(self.n_slots, self.max_ctx, self.max_new_tokens)
        return self.st

    def fits(self, n_prompt_tokens: int, max_new_tokens: int) -> bool:
        return max_new_tokens <= self.max_new_tokens and n_prompt_tokens + max_new_tokens <= self.max_ctx

    def embed(self, prompt, system):
        """Prompt -> spliced embeddings [S, H] on the device (tower + projector + splice run here, on the worker thread)."""
        cfg = self.model.cfg
        dev = str(self.model.device)
        text, tiles, media_config = prepare_prompt(prompt, cfg)
        ids = encode_with_images(self.tokenizer, prompt_text(self.tokenizer, text, system), cfg.image_token_id)[None].to(dev)
        media = {"image": [t.to(device=dev, dtype=torch.bfloat16) for t in tiles]}
        e, _, _ = self.model._embed(ids, media, media_config)
        return e[0]

    def admit(self, slot: int, embeds) -> int:
        return self.model.llm.batch_admit(self._session(), slot, embeds)

    def run(self, k: int) -> None:
        self.model.llm.batch_run(self._session(), k)

    def read(self):
        st = self._session()
        n = st.n_out.tolist()
        top = max(max(n), 1)
        return n, st.out_ids[:, :min(top, st.out_ids.shape[1])].tolist()

    def release(self, slots) -> None:
        self.model.llm.batch_release(self._session(), slots)

    def solo(self, prompt, max_new_tokens, system, streamer=None, **gen) -> str:
        return generate_content(self.model, self.tokenizer, prompt, max_new_tokens=max_new_tokens, system=system, streamer=streamer, **gen)

    def decode(self, toks) -> str:
        return self.tokenizer.decode(toks, skip_special_tokens=True).strip()


class ContinuousBatcher:
    """Continuous batching in front of the batched decode step (the reference's server answers requests as they arrive,
    server.py:171-290).  ONE worker thread owns the model (every request — greedy, sampled, batched or not — is executed by
    it, so no two threads ever drive the sessions, the staging buffers or the captured graphs).  Between chunks of <= `chunk` decode steps the
    worker admits waiting greedy requests into free rows (each newcomer is prefilled alone into its KV slot and joins the next step), retires
    rows at EOS / max_new_tokens and hands their slot to the next request — a request that arrives one step after a batch started waits for at
    most one chunk, not for the whole batch.  Requests the batched step cannot serve (sampling, replies or prompts beyond the slots' cache) run
    solo on the same thread once the live rows have drained; arrival order is kept (a solo request at the head blocks later admissions)."""

    def __init__(self, engine, max_batch: int = 8, chunk: int = 8):
        import queue
        import threading
        self.engine = engine
        self.max_batch = int(max(1, min(16, max_batch, getattr(engine, "n_slots", max_batch))))
        self.chunk = int(max(1, chunk))
        self._q: "queue.Queue" = queue.Queue()
        self.events: List[tuple] = []                    # ("admit" | "retire" | "solo" | "run", ...) — observability / tests
        self.thread_ids = set()
        self._stop = False
        self.dead: Optional[BaseException] = None        # set (BEFORE the queue is drained) when the worker thread has died
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()

    def submit(self, prompt, max_new_tokens: int = 128, system: Optional[str] = None, stream: Optional["TextStream"] = None, **gen):
        """-> Future of the decoded reply.  gen: temperature / top_p / top_k / seed — a request with temperature > 0 is a solo one.
        stream: a `TextStream` that receives the request's tokens while the row is still decoding — its first token at admission, then
        the new ones after every chunk of steps (server.py:241-270 streams a reply as it is generated); `end()` when the row retires,
        `fail()` with the exception the future gets."""
        from concurrent.futures import Future
        f: Future = Future()
        req = SimpleNamespace(prompt=prompt, max_new=int(max_new_tokens), system=system, gen=gen, fut=f, stream=stream)
        if self.dead is not None:
            self._fail(req, self.dead)
            return f
        self._q.put(req)
        # the worker may have died between the check above and the put — it sets `dead` BEFORE it drains the queue, so a request
        # that slipped in behind the drain sees `dead` here and fails now instead of waiting in a queue nobody reads (a request the drain did
        # reach is failed there; failing a future / stream twice is harmless: `_fail` ignores the second)
        if self.dead is not None:
            self._fail(req, self.dead)
        return f

    def close(self):
        self._stop = True
        self._q.put(None)
        self._thread.join(timeout=10)

    @staticmethod
    def _greedy(req) -> bool:
        t = req.gen.get("temperature")
        return not t or t <= 0

    @staticmethod
    def _fail(req, ex) -> None:
        if getattr(req, "_failed", False):               # (a dying worker's drain and `submit`'s re-check may both reach a request)
            return
        req._failed = True
        if not req.fut.done():
            req.fut.set_exception(ex)
        if getattr(req, "stream", None) is not None:
            req.stream.fail(ex)

    def _emit(self, row, new) -> None:
        """A row's new tokens -> its stream, like HF's `streamer.put` per token: up to AND including the first EOS, never past max_new_tokens."""
        st = row.req.stream
        if st is None or row.closed:
            return
        for t in new:
            if row.sent >= row.req.max_new:
                break
            st.put([t])
            row.sent += 1
            if t in self.engine.eos:
                row.closed = True
                break

    def _finish(self, row, toks):
        eng = self.engine
        for k, t in enumerate(toks):                       # HF stops AFTER emitting eos; decode(skip_special_tokens) drops it
            if t in eng.eos:
                toks = toks[:k]
                break
        row.req.fut.set_result(eng.decode(toks))
        if row.req.stream is not None:
            row.req.stream.end()

    def _loop(self):
        """The worker: `_loop_body` under a guard — if the worker dies, everything it held and everything still queued fails with the
        cause instead of hanging (the futures / streams of a dead worker would never resolve)."""
        import collections
        import queue
        pending = collections.deque()
        rows: Dict[int, Any] = {}
        try:
            self._loop_body(pending, rows)
        except BaseException as ex:                                    # noqa: BLE001 — the thread is going away either way
            self._stop = True
            err = RuntimeError(f"batcher worker died: {ex!r}")
            self.dead = err                                             # first: `submit` re-checks it behind its put (see there)
            for r in list(rows.values()):
                self._fail(r.req, err)
            for p in pending:
                self._fail(p, err)
            while True:
                try:
                    item = self._q.get(block=False)
                except queue.Empty:
                    break
                if item is not None:
                    self._fail(item, err)

    def _loop_body(self, pending, rows):
        import queue
        import threading
        self.thread_ids.add(threading.get_ident())
        eng = self.engine
        free = list(range(self.max_batch))
        steps = 0
        with torch.inference_mode():
            while True:
                # ---- take what has arrived (block only when there is nothing to do) ----
                while True:
                    try:
                        item = self._q.get(block=not rows and not pending)
                    except queue.Empty:
                        break
                    if item is None:
                        self._stop = True
                        break
                    pending.append(item)
                if self._stop:
                    for r in list(rows.values()) + [SimpleNamespace(req=p) for p in pending]:
                        if not r.req.fut.done():
                            self._fail(r.req, RuntimeError("batcher closed"))
                    return
                # ---- admit greedy requests at the head of the line into free rows; a solo request waits for the rows to drain ----
                while pending and free:
                    req = pending[0]
                    slot = None                                        # the KV slot this admission took, if it got that far
                    try:
                        if not self._greedy(req) or req.gen.get("_solo"):
                            break
                        e = eng.embed(req.prompt, req.system)
                        if not eng.fits(int(e.shape[0]), req.max_new) or req.max_new < 1:
                            req.gen = dict(req.gen, _solo=True)
                            break
                        pending.popleft()
                        slot = free.pop(0)
                        first = eng.admit(slot, e)
                        row = SimpleNamespace(req=req, toks=[first], read=0, sent=0, closed=False)
                        self.events.append(("admit", slot, steps, len(rows)))
                        self._emit(row, [first])
                        if first in eng.eos or req.max_new <= 1:
                            self._finish(row, row.toks)
                            eng.release([slot])
                            free.append(slot)
                            self.events.append(("retire", slot, steps))
                            slot = None
                        else:
                            rows[slot] = row
                            slot = None                                # owned by `rows` from here on
                    except Exception as ex:                            # the request fails, the batch goes on — and the slot goes back
                        if pending and pending[0] is req:
                            pending.popleft()
                        if slot is not None:
                            rows.pop(slot, None)
                            try:
                                eng.release([slot])
                            except Exception:
                                pass
                            if slot not in free:
                                free.append(slot)
                            self.events.append(("admit_failed", slot, steps))
                        self._fail(req, ex)
                if not rows:
                    if pending and (not self._greedy(pending[0]) or pending[0].gen.get("_solo")):
                        req = pending.popleft()
                        gen = {k: v for k, v in req.gen.items() if k != "_solo" and v is not None}
                        self.events.append(("solo", steps))
                        try:
                            if req.stream is not None:
                                gen["streamer"] = req.stream
                            text = eng.solo(req.prompt, req.max_new, req.system, **gen)
                            req.fut.set_result(text)
                            if req.stream is not None:
                                req.stream.finish(text)                 # (an engine whose generate took no streamer)
                        except Exception as ex:
                            self._fail(req, ex)
                    continue
                # ---- one chunk of batched steps: never past the row that is closest to its max_new_tokens ----
                k = min(self.chunk, min(r.req.max_new - len(r.toks) for r in rows.values()))
                try:
                    eng.run(k)
                    n_out, out = eng.read()
                except Exception as ex:
                    for r in rows.values():
                        self._fail(r.req, ex)
                    free.extend(rows)
                    rows.clear()
                    continue
                steps += k
                self.events.append(("run", k, len(rows)))
                done = []
                for slot, row in rows.items():
                    try:                                               # a tokenizer / stream failure fails THIS request only
                        new = out[slot][row.read:n_out[slot]]
                        row.toks.extend(new)
                        row.read = n_out[slot]
                        self._emit(row, new)
                        if any(t in eng.eos for t in row.toks) or len(row.toks) >= row.req.max_new:
                            self._finish(row, row.toks[:row.req.max_new])
                            done.append(slot)
                    except Exception as ex:
                        self._fail(row.req, ex)
                        done.append(slot)
                for slot in done:
                    del rows[slot]
                    self.events.append(("retire", slot, steps))
                idle = done + free                                     # idle rows took part in the steps too: re-wind them
                eng.release(idle)
                free.extend(done)


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


def create_app(model, tokenizer, model_name: str = "NVILA-8B", batch_window_s: Optional[float] = None, max_batch: int = 8,
               stream_timeout_s: Optional[float] = 600.0):
    """FastAPI app with the reference's POST /chat/completions (server.py:171-290).  Import-time optional: needs fastapi + pydantic.
    batch_window_s: when set (any value), requests go through a batcher whose ONE worker thread owns the model: `ContinuousBatcher` (greedy
    requests join / leave the batched decode step between steps; sampled ones run solo on the same thread) where the model has the batched
    step, else `RequestBatcher` (greedy requests that arrive within the window share a batch; the rest run under the batcher's lock)."""
    from fastapi import FastAPI
    from fastapi.responses import JSONResponse, StreamingResponse

    ChatMessage, ChatCompletionRequest = _request_models()
    app = FastAPI()
    batcher = None
    if batch_window_s is not None:
        # continuous batching when the model has the batched decode step (a bf16 head-dim-128 decoder); else the static-window batcher
        try:
            ok = hasattr(model.llm, "batch_open") and model.llm.lcfg.head_dim == 128 and getattr(model.llm, "_w4", None) is None
        except AttributeError:
            ok = False
        batcher = ContinuousBatcher(HipBatchEngine(model, tokenizer, n_slots=max_batch), max_batch=max_batch) if ok else \
            RequestBatcher(model, tokenizer, window_s=batch_window_s)
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
                        elif c.get("type") == "video_url":           # server.py:47-52, 214-221: `frames` (default 8) / `fps` (default 2) ride in the content
                            frames = c.get("frames", 8) if c.get("frames") is not None else 8
                            fps = c.get("fps", 2) if c.get("fps") is not None else 2
                            parts.append(Video(load_video_url_frames(c["video_url"]["url"], num_frames=int(frames), fps=float(fps))))
                        else:
                            raise NotImplementedError(f"Unsupported content type: {c.get('type')}")
            elif m.role == "assistant" and isinstance(m.content, str):
                parts.append(f"<|im_end|>\n<|im_start|>assistant\n{m.content}<|im_end|>\n<|im_start|>user\n")
        return parts, system

    model_lock = batcher.model_lock if isinstance(batcher, RequestBatcher) else threading.Lock()
    app.state.model_lock = model_lock

    def _solo(parts, n, system, temperature, top_p, streamer=None):
        """A request the batcher does not take: under the model lock — never beside the worker thread's batch, never beside another solo
        request."""
        with model_lock, torch.inference_mode():
            return generate_content(model, tokenizer, parts, max_new_tokens=n, system=system, temperature=temperature, top_p=top_p, streamer=streamer)

    async def chat_completions(request):
        import asyncio
        try:
            if request.model != model_name:
                raise ValueError(f"The endpoint is configured to use the model {model_name}, but the request model is {request.model}")
            parts, system = _prompt_of(request.messages)
            temperature = request.temperature if request.temperature is not None else 0.2
            top_p = request.top_p if request.top_p is not None else 0.9
            n = request.max_tokens or 512
            loop = asyncio.get_running_loop()
            if request.stream:
                # server.py:241-270: the reply leaves as it is generated.  The tokens reach a `TextStream` from whoever runs the request (the
                # continuous batcher's worker after every chunk of steps, else a thread of its own: the reference's Thread + TextIteratorStreamer)
                stream = TextStream(tokenizer, timeout=stream_timeout_s)
                if isinstance(batcher, ContinuousBatcher):
                    batcher.submit(parts, n, system, stream=stream, temperature=temperature, top_p=top_p)
                else:
                    def run():
                        try:
                            stream.finish(_solo(parts, n, system, temperature, top_p, streamer=stream))
                        except Exception as ex:                       # the consumer raises it
                            stream.fail(ex)
                    threading.Thread(target=run, daemon=True).start()
                first = await loop.run_in_executor(None, lambda: next(stream, None))      # a failure before any text is still a 500 body
                pieces = stream if first is None else itertools.chain([first], stream)
                return StreamingResponse(sse_chunks(pieces, request.model), media_type="text/event-stream")
            if isinstance(batcher, ContinuousBatcher):
                text = await loop.run_in_executor(None, batcher.submit(parts, n, system, temperature=temperature, top_p=top_p).result)
            elif batcher is not None and not temperature:
                text = await loop.run_in_executor(None, batcher.submit(parts, n, system).result)
            else:
                text = _solo(parts, n, system, temperature, top_p)
            return {"id": uuid.uuid4().hex, "object": "chat.completion", "created": time.time(), "model": request.model,
                    "choices": [{"message": {"role": "assistant", "content": [{"type": "text", "text": text}]}}]}
        except Exception as e:                                       # server.py:292-297: errors come back as a 500 JSON body
            return JSONResponse(status_code=500, content={"error": str(e)})

    chat_completions.__annotations__["request"] = ChatCompletionRequest      # a real class, not a string (postponed annotations)
    app.post("/chat/completions")(chat_completions)
    return app
