"""REFERENCE-EXECUTED fixture for `process_image` / `process_images` and the `dynamic` tiler (the input producer of every
NVILA-Lite script: `--image_aspect_ratio dynamic`) — TEST INFRASTRUCTURE.

`llava/mm_utils.py` cannot be imported here (torchvision), but `find_closest_aspect_ratio` (:283-296), `dynamic_preprocess` (:299-338),
`dynamic_s2_preprocess` (:341-405), `process_image` (:442-523), `process_images` (:526-541) and `dynamic_process_images_and_prompt` (:408-424)
are PIL / integer / string code: their definitions are taken out of the file with `ast` and EXECUTED unchanged, with HF's
`SiglipImageProcessor` as `data_args.image_processor` (the `vision_tower.image_processor` of the reference), on seeded synthetic images.

Stored per image size:
  * `dynamic`  — tile count and the CRC32 of every tile's RGB bytes (`dynamic_preprocess`, two (min, max) settings), and for `process_image(
    enable_dynamic_res=True)` the stacked tensor's shape, a 3 x 16 x 16 corner of the LAST tile (the thumbnail where there is one), sum and
    sum of squares
  * `resize` / `pad` / `""` (processor default) — corner, sum, sum of squares of the `[3, S, S]` tensor
  * the prompt the dataset path builds for the image (`dynamic_process_images_and_prompt`: `<image>\\n` per tile)
and once: `process_images` on three pictures (stacked `[3, 3, S, S]`) and its error on mixed shapes.

    python oracle/make_golden_dynamic_tiles.py      # seconds; needs /root/reference; writes tests/golden/dynamic_tiles.npz
"""
from __future__ import annotations

import ast
import os
import sys
import zlib
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden_s2_tiles import synthetic_image  # noqa: E402

REF = "/root/reference/llava/mm_utils.py"
OUT = os.path.join(ROOT, "tests", "golden", "dynamic_tiles.npz")
SIZE = 448
SETTINGS = [(1, 12), (1, 6), (4, 9)]                  # (min_tiles, max_tiles): the scripts' 1..12, a video_max_tiles-style cap, a floor above 1
CASES = [(448, 448), (1344, 1344), (600, 800), (800, 600), (1600, 900), (900, 1600), (400, 1200), (1200, 400), (2000, 1000), (37, 53),
         (64, 64), (4000, 3000), (1000, 333), (1345, 1343), (897, 449), (640, 480), (5000, 400), (449, 447)]
NAMES = ("find_closest_aspect_ratio", "dynamic_preprocess", "dynamic_s2_preprocess", "process_image", "process_images",
         "dynamic_process_images_and_prompt")


def reference_functions():
    import torch
    from PIL import Image
    mod = ast.parse(open(REF).read())
    keep = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name in NAMES]
    assert sorted(n.name for n in keep) == sorted(NAMES)
    ns = {"Image": Image, "torch": torch, "os": os, "DEFAULT_IMAGE_TOKEN": "<image>", "tv_tensors": None, "v2": None}   # llava/constants.py:38
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns


def prints(t, corner=16):
    a = t.numpy().astype(np.float64)
    return a[..., :corner, :corner].astype(np.float32), np.float64(a.sum()), np.float64((a * a).sum())


class Siglip446Processor:
    """What `process_image` sees of transformers 4.46's `SiglipImageProcessor` (the version the reference pins): `size`, `image_mean`,
    `preprocess` and NO `crop_size` attribute (transformers 5.x's processor carries `crop_size = None`, which would send mm_utils.py:459-465 down
    the CLIP branch).  The pixels come from the installed processor unchanged."""

    def __init__(self, proc):
        self._proc = proc
        self.size = {"height": int(proc.size["height"]), "width": int(proc.size["width"])}
        self.image_mean = list(proc.image_mean)

    def preprocess(self, image, return_tensors="pt"):
        return self._proc.preprocess(image, return_tensors=return_tensors)


def data_args(proc, mode, mn=1, mx=12):
    return SimpleNamespace(image_processor=proc, image_aspect_ratio=mode, min_tiles=mn, max_tiles=mx, s2_scales=[448, 896, 1344])


def main():
    from transformers import SiglipImageProcessor
    ns = reference_functions()
    proc = Siglip446Processor(SiglipImageProcessor(size={"height": SIZE, "width": SIZE}))
    out = {"cases": np.asarray(CASES, dtype=np.int64), "settings": np.asarray(SETTINGS, dtype=np.int64), "image_size": np.int64(SIZE)}
    for k, (w, h) in enumerate(CASES):
        img = synthetic_image(w, h, 100 + k)
        for s, (mn, mx) in enumerate(SETTINGS):
            tiles = ns["dynamic_preprocess"](img, min_num=mn, max_num=mx, image_size=SIZE)
            assert all(t.size == (SIZE, SIZE) for t in tiles)
            out[f"crc_{k}_{s}"] = np.asarray([zlib.crc32(t.convert("RGB").tobytes()) for t in tiles], dtype=np.int64)
        px = ns["process_image"](img, data_args(proc, "dynamic"), None, enable_dynamic_res=True)
        out[f"dyn_shape_{k}"] = np.asarray(px.shape, dtype=np.int64)
        out[f"dyn_px_{k}"], out[f"dyn_sum_{k}"], out[f"dyn_sq_{k}"] = prints(px[-1])
        px6 = ns["process_image"](img, data_args(proc, "dynamic"), None, enable_dynamic_res=True, max_tiles=6)     # the `max_tiles=` override (:474-477)
        out[f"dyn6_shape_{k}"] = np.asarray(px6.shape, dtype=np.int64)
        for mode in ("resize", "pad", ""):
            t = ns["process_image"](img, data_args(proc, mode), None)
            assert tuple(t.shape) == (3, SIZE, SIZE)
            tag = mode or "default"
            out[f"{tag}_px_{k}"], out[f"{tag}_sum_{k}"], out[f"{tag}_sq_{k}"] = prints(t)
        # `dynamic` WITHOUT enable_dynamic_res (several images in one prompt, llava_arch.py:877): the processor's default
        t = ns["process_image"](img, data_args(proc, "dynamic"), None)
        out[f"dyn_off_sum_{k}"] = np.float64(t.numpy().astype(np.float64).sum())
        _, prompt = ns["dynamic_process_images_and_prompt"]([img], "Look: <image> what is it?", data_args(proc, "dynamic"))
        out[f"prompt_{k}"] = np.asarray(prompt)
        print(f"{w}x{h}: dynamic {tuple(px.shape)}, max 6 {tuple(px6.shape)}")
    imgs = [synthetic_image(w, h, 200 + i) for i, (w, h) in enumerate([(640, 480), (300, 900), (448, 448)])]
    cfg = SimpleNamespace(image_aspect_ratio="pad", min_tiles=1, max_tiles=12, s2_scales=[448, 896, 1344])
    st = ns["process_images"](imgs, proc, cfg)
    out["stack_shape"] = np.asarray(st.shape, dtype=np.int64)
    out["stack_px"], out["stack_sum"], out["stack_sq"] = prints(st)
    cfg = SimpleNamespace(image_aspect_ratio="dynamic", min_tiles=1, max_tiles=12, s2_scales=[448, 896, 1344])
    twin = synthetic_image(640, 480, 300)
    st = ns["process_images"]([imgs[0], twin], proc, cfg, enable_dynamic_res=True, max_tiles=6)      # tiles of both pictures, concatenated (:533-534)
    out["stack_dyn_shape"] = np.asarray(st.shape, dtype=np.int64)
    out["stack_dyn_sum"] = np.float64(st.numpy().astype(np.float64).sum())
    # the dataset path's prompt for TWO pictures in one message (mm_utils.py:408-424): each `<image>` becomes that picture's tiles
    px2, prompt2 = ns["dynamic_process_images_and_prompt"]([imgs[0], imgs[2]], "A <image> B <image> C", data_args(proc, "dynamic"))
    out["prompt2"], out["prompt2_shape"] = np.asarray(prompt2), np.asarray(px2.shape, dtype=np.int64)
    try:                                                              # pictures with different tile counts do not stack (:539-540)
        ns["process_images"]([imgs[0], imgs[2]], proc, cfg, enable_dynamic_res=True, max_tiles=6)
        raise AssertionError("expected the reference to refuse")
    except ValueError as e:
        out["stack_error"] = np.asarray(str(e))
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes)")


if __name__ == "__main__":
    main()
