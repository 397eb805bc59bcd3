"""REFERENCE-EXECUTED fixture for the prompt half of `generate_content`: `extract_media` — TEST INFRASTRUCTURE.

`llava.utils.media` imports cv2 / requests / the llava package, so `extract_media` is taken from its file with `ast` and exec'd UNCHANGED
(llava/utils/media.py:93-122), with `MEDIA_TOKENS` read out of llava/constants.py:32-35, `make_list` from llava/utils/utils.py:22-23 and the
`Image` / `Video` classes from llava/media.py (all executed from their files).  `_extract_image` is the reference's for PIL images (identity);
`_extract_video` (cv2 decoding) is replaced by "the frames the test supplies", `config.num_video_frames` of them.
Each case is a prompt (a list of parts: strings, "IMG" = one PIL image, "VID3" = a 3-frame video); stored: the text the reference rewrites the
message to and how many images it collected.  tests/test_serving_cpu.py holds `vila_amd.serving._split_prompt` to it.

    python oracle/make_golden_prompt.py       # writes tests/golden/prompt_split_ref.json; needs /root/reference
"""
from __future__ import annotations

import ast
import json
import os
import sys
import types
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/llava"
OUT = os.path.join(ROOT, "tests", "golden", "prompt_split_ref.json")

CASES = [
    ["look: ", "IMG", "what is it?"],
    ["a literal <image> token"],
    ["<image> describe", "IMG"],
    ["a typed <vila/video> token"],
    ["IMG", "IMG", "compare <image> the two", "IMG"],
    ["watch ", "VID3", " then answer"],
    ["VID3", "IMG", "  spaces stay  ", "<image>"],
    ["no media at all"],
]


def _node_source(path, pred):
    src = open(path).read()
    for node in ast.parse(src).body:
        if pred(node):
            return ast.get_source_segment(src, node)
    raise KeyError(path)


def load_reference():
    import PIL.Image
    from typing import Any, Dict, List, Optional, Union
    ns = {"defaultdict": defaultdict, "Any": Any, "Dict": Dict, "List": List, "Optional": Optional, "Union": Union, "PIL": PIL,
          "PretrainedConfig": object, "logger": types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)}
    is_assign = lambda name: (lambda n: isinstance(n, ast.Assign) and any(getattr(t, "id", None) == name for t in n.targets))
    exec(compile(_node_source(f"{REF}/constants.py", is_assign("MEDIA_TOKENS")), "constants.py", "exec"), ns)
    exec(compile(_node_source(f"{REF}/utils/utils.py", lambda n: isinstance(n, ast.FunctionDef) and n.name == "make_list"), "utils.py", "exec"), ns)
    for cls in ("Media", "File", "Image", "Video"):
        exec(compile(_node_source(f"{REF}/media.py", lambda n, c=cls: isinstance(n, ast.ClassDef) and n.name == c), "media.py", "exec"), ns)
    exec(compile(_node_source(f"{REF}/utils/media.py", lambda n: isinstance(n, ast.FunctionDef) and n.name == "_extract_image"), "utils/media.py", "exec"), ns)
    exec(compile(_node_source(f"{REF}/utils/media.py", lambda n: isinstance(n, ast.FunctionDef) and n.name == "extract_media"), "utils/media.py", "exec"), ns)
    return ns


def main():
    import PIL.Image
    ns = load_reference()
    frames = {}
    ns["_extract_video"] = lambda video, config: frames[id(video)]          # stands for cv2 decoding: the frames this test supplies
    out = []
    for parts in CASES:
        built = []
        for p in parts:
            if p == "IMG":
                built.append(PIL.Image.new("RGB", (8, 8)))
            elif p.startswith("VID"):
                v = ns["Video"]("clip.mp4")
                frames[id(v)] = [PIL.Image.new("RGB", (8, 8)) for _ in range(int(p[3:]))]
                built.append(v)
            else:
                built.append(p)
        msg = [{"from": "human", "value": built}]
        media = ns["extract_media"](msg, types.SimpleNamespace(num_video_frames=3))
        assert set(media) <= {"image"}
        out.append({"parts": parts, "text": msg[0]["value"], "n_images": len(media.get("image", []))})
        print(parts, "->", repr(msg[0]["value"]), out[-1]["n_images"])
    json.dump({"media_tokens": ns["MEDIA_TOKENS"], "cases": out}, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
