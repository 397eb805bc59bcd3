"""`process_image` / `process_images` and the `dynamic` tiler (every NVILA-Lite script runs `--image_aspect_ratio dynamic`)
against REFERENCE-EXECUTED vectors: oracle/make_golden_dynamic_tiles.py takes `dynamic_preprocess`, `process_image`, `process_images` and
`dynamic_process_images_and_prompt` out of llava/mm_utils.py with ast and executes them unchanged over HF's SiglipImageProcessor.
Integer / byte work (grids, tile bytes, prompt text): bit-exact.  Pixels: <= 1e-6 (fp32 rounding order of rescale + normalise)."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle.make_golden_s2_tiles import synthetic_image
from vila_amd import configs, serving
from vila_amd.host import dynamic_preprocess, dynamic_tile_plan, expand2square

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "dynamic_tiles.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(GOLDEN)


def _cfg(mode, size, mn=1, mx=12):
    cfg = configs.nvila_lite_3b()
    assert cfg.vision.image_size == size
    cfg.image_aspect_ratio, cfg.min_tiles, cfg.max_tiles = mode, mn, mx
    return cfg


def _close(t, px, s, sq, corner=16):
    a = t.numpy().astype(np.float64)
    assert np.abs(a[..., :corner, :corner] - px).max() <= 1e-6
    assert abs(a.sum() - float(s)) <= 1e-6 * a.size and abs((a * a).sum() - float(sq)) <= 1e-6 * a.size


def test_dynamic_grid_and_every_tile_bit_exact_vs_reference(fx):
    size = int(fx["image_size"])
    counts = set()
    for k, (w, h) in enumerate(fx["cases"].tolist()):
        img = synthetic_image(w, h, 100 + k)
        for s, (mn, mx) in enumerate(fx["settings"].tolist()):
            tiles = dynamic_preprocess(img, min_num=mn, max_num=mx, image_size=size)
            crc = [zlib.crc32(t.convert("RGB").tobytes()) for t in tiles]
            assert crc == fx[f"crc_{k}_{s}"].tolist(), f"{w}x{h} ({mn}..{mx}): tile bytes differ from the reference's"
            (tw, th), boxes, thumb = dynamic_tile_plan(w, h, mn, mx, size)
            assert len(tiles) == len(boxes) + int(thumb) and tw * th == len(boxes) * size * size and mn <= len(boxes) <= mx
            assert thumb == (len(boxes) != 1)
            counts.add(len(tiles))
    assert len(counts) >= 6                                              # 1, 3, 4, 5, 7, 9, 10, 13 tiles: not one grid over and over


def test_process_image_every_aspect_mode_vs_reference(fx):
    size = int(fx["image_size"])
    for k, (w, h) in enumerate(fx["cases"].tolist()):
        img = synthetic_image(w, h, 100 + k)
        px = serving.process_image(img, _cfg("dynamic", size), enable_dynamic_res=True)
        assert list(px.shape) == fx[f"dyn_shape_{k}"].tolist() and px.dtype == torch.float32
        _close(px[-1], fx[f"dyn_px_{k}"], fx[f"dyn_sum_{k}"], fx[f"dyn_sq_{k}"])
        assert list(serving.process_image(img, _cfg("dynamic", size), enable_dynamic_res=True, max_tiles=6).shape) == fx[f"dyn6_shape_{k}"].tolist()
        for mode in ("resize", "pad", ""):
            tag = mode or "default"
            t = serving.process_image(img, _cfg(mode, size))
            assert t.shape == (3, size, size)
            _close(t, fx[f"{tag}_px_{k}"], fx[f"{tag}_sum_{k}"], fx[f"{tag}_sq_{k}"])
        t = serving.process_image(img, _cfg("dynamic", size))            # the tiling recipe WITHOUT the switch: the processor's default
        assert abs(float(t.double().sum()) - float(fx[f"dyn_off_sum_{k}"])) <= 1e-6 * t.numel()
    # `pad` really pads (a wide picture's top rows are the mean colour -> 127 / 255 -> (x - 0.5) / 0.5 just below 0), `resize` does not
    wide = synthetic_image(1200, 400, 107)
    top = serving.process_image(wide, _cfg("pad", size))[:, :100]
    assert float(top.abs().max()) < 0.01 and float(serving.process_image(wide, _cfg("resize", size))[:, :100].abs().max()) > 0.5
    assert expand2square(wide, (127, 127, 127)).size == (1200, 1200) and expand2square(synthetic_image(64, 64, 1), (0, 0, 0)).size == (64, 64)


def test_process_images_stacks_whole_pictures_and_concatenates_tiles(fx):
    size = int(fx["image_size"])
    imgs = [synthetic_image(w, h, 200 + i) for i, (w, h) in enumerate([(640, 480), (300, 900), (448, 448)])]
    st = serving.process_images(imgs, _cfg("pad", size))
    assert list(st.shape) == fx["stack_shape"].tolist()
    _close(st, fx["stack_px"], fx["stack_sum"], fx["stack_sq"])
    twin = synthetic_image(640, 480, 300)
    st = serving.process_images([imgs[0], twin], _cfg("dynamic", size), enable_dynamic_res=True, max_tiles=6)
    assert list(st.shape) == fx["stack_dyn_shape"].tolist()
    assert abs(float(st.double().sum()) - float(fx["stack_dyn_sum"])) <= 1e-6 * st.numel()
    with pytest.raises(ValueError) as e:                                 # different tile counts: refused with the reference's words
        serving.process_images([imgs[0], imgs[2]], _cfg("dynamic", size), enable_dynamic_res=True, max_tiles=6)
    assert str(e.value) == str(fx["stack_error"])


def test_prepare_prompt_under_the_dynamic_recipe(fx):
    """One image -> its tiles + one `<image>\\n` per tile in the text (llava_arch.py:862-866; the dataset path's text, executed from
    mm_utils.py:408-424, is the fixture); several images -> whole pictures, text untouched (llava_arch.py:877)."""
    size = int(fx["image_size"])
    cfg = _cfg("dynamic", size)
    for k, (w, h) in enumerate(fx["cases"].tolist()):
        img = synthetic_image(w, h, 100 + k)
        text, tiles, mc = serving.prepare_prompt(["Look: ", img, " what is it?"], cfg)
        assert text == str(fx[f"prompt_{k}"]) and mc == {}
        assert len(tiles) == int(fx[f"dyn_shape_{k}"][0]) == text.count("<image>") and all(t.shape == (3, size, size) for t in tiles)
    img = synthetic_image(1600, 900, 104)
    text, tiles, mc = serving.prepare_prompt([img, img, "compare"], cfg)
    assert text == "<image><image>compare" and len(tiles) == 2 and mc == {}
    text, tiles, mc = serving.prepare_prompt(["what? ", img], cfg)       # a trailing "\n" goes with the message's strip (tokenizer.py:77-78)
    assert text == "what? " + "<image>\n" * 8 + "<image>" and len(tiles) == 9
    # the other recipes leave the text alone
    text, tiles, mc = serving.prepare_prompt([img, "describe"], _cfg("resize", size))
    assert text == "<image>describe" and len(tiles) == 1 and mc == {}
    s2 = _cfg("dynamic_s2", size)
    s2.dynamic_s2 = True
    text, tiles, mc = serving.prepare_prompt([img, "describe"], s2)
    (rows, cols), = mc["image"]["block_sizes"]
    assert text == "<image>describe" and len(tiles) == 1 + 4 + rows * cols


def test_aspect_mode_resolution_and_checkpoint_round_trip(tmp_path):
    cfg = configs.tiny("mlp_downsample")
    assert cfg.aspect_mode == ""
    cfg.dynamic_s2 = True
    assert cfg.aspect_mode == "dynamic_s2"                               # older configs carry only the flag
    cfg.dynamic_s2 = False
    cfg.image_aspect_ratio, cfg.min_tiles, cfg.max_tiles, cfg.video_max_tiles = "dynamic", 2, 6, 4
    from vila_amd import checkpoint
    from vila_amd.vlm import HipLlavaLlamaModel
    m = HipLlavaLlamaModel(cfg, device="cpu")
    checkpoint.save_pretrained(m, str(tmp_path / "ck"))
    got = checkpoint.config_from_pretrained(str(tmp_path / "ck"))
    assert (got.image_aspect_ratio, got.min_tiles, got.max_tiles, got.video_max_tiles) == ("dynamic", 2, 6, 4) and got.aspect_mode == "dynamic"
