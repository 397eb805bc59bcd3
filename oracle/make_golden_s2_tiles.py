"""REFERENCE-EXECUTED fixture for the dynamic_s2 TILER and the image pre-processing (SURVEY §8 f1 host half, row a1) — TEST INFRASTRUCTURE.

`llava/mm_utils.py` cannot be imported here (torchvision), but `find_closest_aspect_ratio` (:283-296) and `dynamic_s2_preprocess` (:341-405)
are pure PIL / integer code: their function bodies are taken out of the file with `ast` and EXECUTED, on seeded synthetic images of >= 12
sizes (square, 3:4, 4:3, 16:9, 9:16, 1:3, 3:1, tiny, huge, the exact-tie cases of the ratio search).  Stored per case: block_size, tile count,
CRC32 of every tile's RGB bytes.  The pixel leg is HF's `SiglipImageProcessor` (the `vision_tower.image_processor` the reference calls in
`process_image`, mm_utils.py:442-541; transformers as installed, PIL backend): pixel_values of the first tile of every case and of the plain
resize-to-448^2 path, as a 3 x 24 x 24 corner + the tensor's sum and sum of squares.

    python oracle/make_golden_s2_tiles.py       # seconds; needs /root/reference; writes tests/golden/s2_tiles.npz
"""
from __future__ import annotations

import ast
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/llava/mm_utils.py"
OUT = os.path.join(ROOT, "tests", "golden", "s2_tiles.npz")
SCALES, MAX_NUM, SIZE = [448, 896, 1344], 12, 448
# (width, height): aspect ratios 1:1, 3:4, 4:3, 16:9, 9:16, 1:3, 3:1, 2:1, 1:2, tiny, huge, exact ties (e.g. 3:3 vs 2:2 at small area), odd sizes
CASES = [(448, 448), (1344, 1344), (600, 800), (800, 600), (1600, 900), (900, 1600), (400, 1200), (1200, 400), (2000, 1000), (500, 1000),
         (37, 53), (64, 64), (4000, 3000), (3000, 4000), (1000, 333), (333, 1000), (1345, 1343), (897, 449), (1792, 896), (640, 480)]


def synthetic_image(w: int, h: int, seed: int):
    """Smooth gradients + seeded noise: resampling filters and crop offsets both leave a trace."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 255.0 / max(w - 1, 1)), (y * 255.0 / max(h - 1, 1)), ((x + y) % 256).astype(np.float64)], -1)
    noise = rng.integers(-40, 41, size=(h, w, 3))
    return Image.fromarray(np.clip(base + noise, 0, 255).astype(np.uint8), "RGB")


def reference_functions():
    mod = ast.parse(open(REF).read())
    keep = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name in ("find_closest_aspect_ratio", "dynamic_s2_preprocess")]
    assert len(keep) == 2
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns["dynamic_s2_preprocess"]


def pixel_prints(t):
    a = t.numpy().astype(np.float64)
    return a[:, :24, :24].astype(np.float32), np.float64(a.sum()), np.float64((a * a).sum())


def main():
    from transformers import SiglipImageProcessor
    ref_tiler = reference_functions()
    proc = SiglipImageProcessor(size={"height": SIZE, "width": SIZE})
    out = {"cases": np.asarray(CASES, dtype=np.int64), "s2_scales": np.asarray(SCALES), "max_num": np.int64(MAX_NUM), "image_size": np.int64(SIZE)}
    for k, (w, h) in enumerate(CASES):
        img = synthetic_image(w, h, k)
        tiles, block = ref_tiler(img, s2_scales=list(SCALES), max_num=MAX_NUM, image_size=SIZE)
        out[f"block_{k}"] = np.asarray(block, dtype=np.int64)
        out[f"crc_{k}"] = np.asarray([zlib.crc32(t.convert("RGB").tobytes()) for t in tiles], dtype=np.int64)
        assert all(t.size == (SIZE, SIZE) for t in tiles)
        pv = proc.preprocess(tiles[0], return_tensors="pt")["pixel_values"][0]                    # mm_utils.py:470: every tile through the processor
        out[f"tile0_px_{k}"], out[f"tile0_sum_{k}"], out[f"tile0_sq_{k}"] = pixel_prints(pv)
        pv = proc.preprocess(img, return_tensors="pt")["pixel_values"][0]                         # mm_utils.py:523: the plain path (SigLIP default = resize)
        out[f"plain_px_{k}"], out[f"plain_sum_{k}"], out[f"plain_sq_{k}"] = pixel_prints(pv)
        print(f"{w}x{h}: block {block}, {len(tiles)} tiles")
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes)")


if __name__ == "__main__":
    main()
