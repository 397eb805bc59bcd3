"""Golden fixtures for the video encoders produced by EXECUTING the reference's own code — TEST INFRASTRUCTURE.

`llava/model/encoders/video/{basic,tsp}.py` cannot be imported here (`llava.model` pulls in deepspeed), so `pool`,
`BasicVideoEncoder._process_features` and `TSPVideoEncoder._process_features` are taken from their source files with `ast` and exec'd
unchanged (tsp.py:10-11,28-52; basic.py:30-41); the `super()._process_features(...)` call inside TSP is bound to the extracted Basic
function through a two-class shim with the same inheritance.

    python oracle/make_golden_video.py      ->  tests/golden/video_encoders.npz
"""
from __future__ import annotations

import ast
import os
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_DIR = "/root/reference/llava/model/encoders/video"


def _method_source(path: str, cls: str, name: str) -> str:
    src = open(path).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == name:
                    return textwrap.dedent(ast.get_source_segment(src, fn))
    raise KeyError((cls, name))


def _function_source(path: str, name: str) -> str:
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return ast.get_source_segment(src, node)
    raise KeyError(name)


def load_reference_classes():
    from typing import List, Optional, Tuple  # noqa: F401  (names used by the extracted annotations)
    ns = {"torch": torch, "Optional": Optional, "List": List, "Tuple": Tuple}
    exec(compile(_function_source(os.path.join(REF_DIR, "tsp.py"), "pool"), "tsp.py", "exec"), ns)
    basic = textwrap.indent(_method_source(os.path.join(REF_DIR, "basic.py"), "BasicVideoEncoder", "_process_features"), "    ")
    tsp = textwrap.indent(_method_source(os.path.join(REF_DIR, "tsp.py"), "TSPVideoEncoder", "_process_features"), "    ")
    shim = ("class BasicVideoEncoder:\n" + basic + "\n\nclass TSPVideoEncoder(BasicVideoEncoder):\n"
            "    def __init__(self, pool_sizes, sep_tokens=None):\n        self.pool_sizes = pool_sizes\n        self.sep_tokens = sep_tokens\n" + tsp + "\n")
    exec(compile(shim, "video_encoders_shim", "exec"), ns)
    return ns


def main():
    ns = load_reference_classes()
    g = torch.Generator().manual_seed(5)
    H = 64
    fx = {}
    # case A: NVILA-Video recipe shape in small: 16 frames, 4x4 tokens, pool [[8,1,1]], end token only
    # case B: two pool sizes incl. spatial pooling + start token (2 rows) + separator
    # case C: BasicVideoEncoder (no pooling), 3 frames
    cases = {"A": (16, 4, [[8, 1, 1]], 0, 1, 0), "B": (8, 6, [[4, 2, 3], [2, 1, 1]], 2, 1, 1), "C": (3, 4, None, 0, 2, 0)}
    for name, (nt, nl, pools, n_s, n_e, n_sep) in cases.items():
        x = torch.randn(nt, nl * nl, H, generator=g)
        start = torch.randn(n_s, H, generator=g) if n_s else None
        end = torch.randn(n_e, H, generator=g) if n_e else None
        sep = torch.randn(n_sep, H, generator=g) if n_sep else None
        if pools is None:
            out = ns["BasicVideoEncoder"]()._process_features(x, start_token_embeds=start, end_token_embeds=end)
        else:
            out = ns["TSPVideoEncoder"](pools)._process_features(x, start_token_embeds=start, end_token_embeds=end, sep_token_embeds=sep)
        fx[f"{name}_in"] = x.numpy()
        fx[f"{name}_out"] = out.numpy()
        fx[f"{name}_pools"] = np.array(pools if pools is not None else [[1, 1, 1]], dtype=np.int32)
        for k, t in (("start", start), ("end", end), ("sep", sep)):
            fx[f"{name}_{k}"] = (t if t is not None else torch.zeros(0, H)).numpy()
    # integer fixture: pins the window / ordering exactly (values are small integers, means of 8 are exact in bf16 for these)
    nt, nl = 16, 4
    xi = ((torch.arange(nt * nl * nl * 8).reshape(nt, nl * nl, 8) * 7) % 64).float() * 8
    fx["int_in"] = xi.numpy()
    fx["int_out"] = ns["TSPVideoEncoder"]([[8, 2, 2]])._process_features(xi, start_token_embeds=None, end_token_embeds=None,
                                                                         sep_token_embeds=None).numpy()
    path = os.path.join(ROOT, "tests", "golden", "video_encoders.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, {k: v.shape for k, v in fx.items()})


if __name__ == "__main__":
    main()
