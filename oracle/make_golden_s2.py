"""Golden fixtures for the dynamic_s2 path produced by EXECUTING the reference's own functions.

`llava/model/llava_arch.py` cannot be imported here (it pulls in deepspeed), so the three functions are taken from its source with
`ast` and exec'd unchanged in a namespace that provides what they use (torch, F, einops.rearrange):
    LlavaMetaModel.merge_chessboard / split_chessboard / merge_features_for_dynamic_s2   (llava_arch.py:255-364)
and driven exactly like the dynamic_s2 branch of `encode_images` (llava_arch.py:369-390).

    python oracle/make_golden_s2.py      ->  tests/golden/dynamic_s2.npz
"""
from __future__ import annotations

import ast
import os
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F
from einops import rearrange

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/llava/model/llava_arch.py"


def load_reference_functions():
    src = open(REF).read()
    tree = ast.parse(src)
    wanted = {"merge_chessboard", "split_chessboard", "merge_features_for_dynamic_s2"}
    ns = {"torch": torch, "F": F, "rearrange": rearrange}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "LlavaMetaModel":
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name in wanted:
                    fn.decorator_list = []                      # drop @staticmethod: called as plain functions / with a fake self
                    code = ast.get_source_segment(src, fn)
                    code = textwrap.dedent(code)
                    code = "\n".join(l for l in code.splitlines() if not l.strip().startswith("@staticmethod"))
                    exec(compile(code, REF, "exec"), ns)
    assert wanted <= set(ns), wanted - set(ns)
    return ns


def ref_encode_s2(ns, feats, block_sizes, scales, resize_idx, projector):
    """The dynamic_s2 branch of LlavaMetaModel.encode_images (llava_arch.py:369-390) with the tower output given."""
    tower = types.SimpleNamespace(scales=list(scales), resize_output_to_scale_idx=resize_idx)
    fake_self = types.SimpleNamespace(get_vision_tower=lambda: tower, merge_chessboard=ns["merge_chessboard"])
    image_features, new_block_sizes = ns["merge_features_for_dynamic_s2"](fake_self, feats, block_sizes)
    image_features = [ns["split_chessboard"](x, bs[0], bs[1]) for x, bs in zip(image_features, new_block_sizes)]
    proj_in = torch.cat([rearrange(x, "b c h w -> b (h w) c") for x in image_features], dim=0)
    out = projector(proj_in)
    out = list(out.split([bs[0] * bs[1] for bs in new_block_sizes], dim=0))
    out = [ns["merge_chessboard"](x, bs[0], bs[1]) for x, bs in zip(out, new_block_sizes)]
    out = [rearrange(x, "1 c h w -> (h w) c") for x in out]
    return proj_in, out, new_block_sizes


def main():
    from oracle.make_golden import run_projector, run_vision
    from vila_amd import configs, synthetic
    ns = load_reference_functions()
    cfg = configs.tiny_s2()
    seed = 11
    w = synthetic.make_weights(cfg, seed)
    # image A: block size (2, 3) (non-square last scale) -> 1 + 4 + 6 tiles; image B: block_sizes None (single tile)
    block_sizes = [(2, 3), None]
    n_tiles = 1 + 4 + 6 + 1
    px = synthetic.make_pixels(cfg, n_tiles, seed)
    feats = run_vision(cfg, w, px)[cfg.vision.select_layer]
    fx = {"seed": np.int64(seed), "block_sizes": np.array([[2, 3], [0, 0]]), "tower_out": feats.numpy()}
    proj_in, outs, nbs = ref_encode_s2(ns, feats, list(block_sizes), cfg.s2_scales, cfg.s2_resize_output_to_scale_idx,
                                       lambda x: run_projector(cfg, w, x))
    fx["proj_in"] = proj_in.numpy()
    fx["new_block_sizes"] = np.array(nbs)
    for i, o in enumerate(outs):
        fx[f"tokens_{i}"] = o.numpy()
    # integer-valued chessboard / area-interpolation fixture: pins orderings and window edges exactly
    g = 4
    t = (torch.arange(15 * g * g * 2, dtype=torch.float32).reshape(15, g * g, 2) % 61) - 30
    pin, _, _ = ref_encode_s2(ns, t, [(3, 3), None], (8, 16, 24), -1, lambda x: x[:, :1])
    fx["int_tiles"] = t.numpy()
    fx["int_proj_in"] = pin.numpy()
    # round 4: s2_resize_output_to_scale_idx other than the last scale (llava_arch.py:340-358: every scale is area-interpolated to THAT scale's
    # grid and the image's blocks become s x s): the same integer tiles with the output at scale 0 (1 x 1 block) and scale 1 (2 x 2 blocks)
    for r in (0, 1):
        pin_r, _, nbs_r = ref_encode_s2(ns, t, [(3, 3), None], (8, 16, 24), r, lambda x: x[:, :1])
        fx[f"int_proj_in_r{r}"] = pin_r.numpy()
        fx[f"int_new_block_sizes_r{r}"] = np.array(nbs_r)
    # and a non-square last scale (2 x 3 tiles) pooled DOWN to the middle scale's 2 x 2 blocks
    t2 = (torch.arange(12 * g * g * 2, dtype=torch.float32).reshape(12, g * g, 2) % 53) - 26
    pin_m, _, nbs_m = ref_encode_s2(ns, t2, [(2, 3), None], (8, 16, 24), 1, lambda x: x[:, :1])
    fx["int_tiles_23"] = t2.numpy()
    fx["int_proj_in_23_r1"] = pin_m.numpy()
    fx["int_new_block_sizes_23_r1"] = np.array(nbs_m)
    path = os.path.join(ROOT, "tests", "golden", "dynamic_s2.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, {k: v.shape for k, v in fx.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
