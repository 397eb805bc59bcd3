"""The dynamic_s2 tiler (SURVEY §8 f1 host half) and the image pre-processing (row a1) against REFERENCE-EXECUTED vectors
(oracle/make_golden_s2_tiles.py: `find_closest_aspect_ratio` / `dynamic_s2_preprocess` taken out of llava/mm_utils.py:283-296,341-405 with ast and
executed; HF `SiglipImageProcessor` for the pixels).  Integer / byte work: bit-exact.  Pixels: <= 1e-6 (fp32 rounding order of the rescale)."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle.make_golden_s2_tiles import synthetic_image
from vila_amd import configs, serving
from vila_amd.host import dynamic_s2_preprocess, dynamic_s2_tile_plan, find_closest_aspect_ratio

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "s2_tiles.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(GOLDEN)


def test_tile_grid_and_every_tile_bit_exact_vs_reference(fx):
    scales, max_num, size = [int(x) for x in fx["s2_scales"]], int(fx["max_num"]), int(fx["image_size"])
    cases = fx["cases"].tolist()
    assert len(cases) >= 12
    seen_blocks = set()
    for k, (w, h) in enumerate(cases):
        img = synthetic_image(w, h, k)
        tiles, block = dynamic_s2_preprocess(img, scales, max_num, size)
        assert tuple(block) == tuple(int(x) for x in fx[f"block_{k}"]), f"{w}x{h}: block {block} vs reference {fx[f'block_{k}'].tolist()}"
        crc = [zlib.crc32(t.convert("RGB").tobytes()) for t in tiles]
        assert crc == fx[f"crc_{k}"].tolist(), f"{w}x{h}: tile bytes differ from the reference's"
        assert len(tiles) == 1 + 4 + block[0] * block[1]
        plan, b2 = dynamic_s2_tile_plan(w, h, scales, max_num, size)
        assert b2 == block and sum(len(bx) for _, bx in plan) == len(tiles)
        seen_blocks.add(tuple(block))
    assert len(seen_blocks) >= 6                                      # square, wide, tall, 2 x 6, 6 x 2, ... — not one grid over and over


def test_ratio_search_tie_rule():
    """mm_utils.py:292-294: on an exact tie of |aspect - cols/rows| a later (larger) grid replaces the best one only if the image has more than
    half that grid's pixels."""
    ratios = sorted({(i, j) for n in range(1, 13) for i in range(1, n + 1) for j in range(1, n + 1) if 1 <= i * j <= 12}, key=lambda x: x[0] * x[1])
    assert find_closest_aspect_ratio(1.0, ratios, 100, 100, 448) == (1, 1)              # tiny square: stays 1 x 1
    assert find_closest_aspect_ratio(1.0, ratios, 2000, 2000, 448) == (3, 3)            # big square: the largest square grid wins the ties
    assert find_closest_aspect_ratio(2.0, ratios, 2000, 1000, 448) == (4, 2)


def test_preprocess_image_equals_the_hf_processor_pixels(fx):
    size = int(fx["image_size"])
    scales, max_num = [int(x) for x in fx["s2_scales"]], int(fx["max_num"])
    for k, (w, h) in enumerate(fx["cases"].tolist()):
        img = synthetic_image(w, h, k)
        px = serving.preprocess_image(img, size)                                          # the plain path: resize to size x size
        assert px.shape == (3, size, size) and px.dtype == torch.float32
        a = px.numpy().astype(np.float64)
        assert np.abs(a[:, :24, :24] - fx[f"plain_px_{k}"]).max() <= 1e-6, f"{w}x{h}: plain-path pixels differ"
        assert abs(a.sum() - float(fx[f"plain_sum_{k}"])) <= 1e-6 * a.size and abs((a * a).sum() - float(fx[f"plain_sq_{k}"])) <= 1e-6 * a.size
        tiles, _ = dynamic_s2_preprocess(img, scales, max_num, size)
        t0 = serving.preprocess_image(tiles[0], size).numpy().astype(np.float64)
        assert np.abs(t0[:, :24, :24] - fx[f"tile0_px_{k}"]).max() <= 1e-6
        assert abs(t0.sum() - float(fx[f"tile0_sum_{k}"])) <= 1e-6 * t0.size
    # a uint8 array and a CHW float tensor of the same picture give the same pixels as the PIL image
    img = synthetic_image(640, 480, 3)
    ref = serving.preprocess_image(img, size)
    arr = np.asarray(img)
    assert torch.equal(serving.preprocess_image(arr, size), ref)
    assert torch.equal(serving.preprocess_image(torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255.0, size), ref)


def test_preprocess_media_tiles_one_image_under_dynamic_s2():
    cfg = configs.tiny_s2() if hasattr(configs, "tiny_s2") else configs.tiny("mlp_downsample")
    cfg.dynamic_s2 = True
    size = cfg.vision.image_size
    cfg.s2_scales = (size, 2 * size, 3 * size)
    img = synthetic_image(1600, 900, 5)
    tiles, mc = serving.preprocess_media([img], cfg)
    (rows, cols), = mc["image"]["block_sizes"]
    assert (rows, cols) == (3, 4) or rows * cols >= 9                                    # 16:9 -> a wide grid of 9..12 tiles
    assert len(tiles) == 1 + 4 + rows * cols and all(t.shape == (3, size, size) for t in tiles)
    # two images in one prompt: the reference tiles only a single image (llava_arch.py:860), the rest are resized whole
    tiles2, mc2 = serving.preprocess_media([img, img], cfg)
    assert len(tiles2) == 2 and mc2 == {}
    cfg.dynamic_s2 = False
    tiles3, mc3 = serving.preprocess_media([img], cfg)
    assert len(tiles3) == 1 and mc3 == {}
