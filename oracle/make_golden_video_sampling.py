"""REFERENCE-EXECUTED fixture for the frame selection of a video prompt part: `_load_video` (llava/utils/media.py:39-86) — TEST INFRASTRUCTURE.

The function is taken from its file with `ast` and executed unchanged.  OpenCV is not installed here, so `cv2` is a STAND-IN capture object
that serves synthetic frames whose pixel value is the frame's index (and, like real containers, may report more frames than it can grab); the
directory branch runs on real PNG files.  Stored per case: the frame indices the reference returned.  tests/test_serving_cpu.py holds
`vila_amd.serving.video_frame_indices` / `load_video_frames` to it.

    python oracle/make_golden_video_sampling.py      # writes tests/golden/video_sampling_ref.json; needs /root/reference
"""
from __future__ import annotations

import ast
import glob
import json
import os
import sys
import tempfile
import types

import numpy as np
import PIL.Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "video_sampling_ref.json")
SRC = "/root/reference/llava/utils/media.py"

# (container frame count, frames that can really be grabbed, container fps, num_frames asked, fps asked)
CASES = [(100, 100, 25.0, 8, 0.0), (100, 97, 25.0, 8, 0.0), (31, 31, 30.0, 8, 2.0), (300, 300, 30.0, 8, 2.0), (300, 300, 30.0, 64, 0.0),
         (7, 7, 24.0, 8, 0.0), (48, 48, 24.0, 6, 0.5), (10, 10, 0.0, 4, 2.0), (1, 1, 10.0, 8, 0.0)]
DIR_CASES = [(20, 8), (5, 8), (9, 3)]                    # (frame files in the directory, num_frames asked)


def fake_cv2(count, grabbable, vfps):
    class Cap:
        def __init__(self, path):
            self.pos = 0

        def get(self, prop):
            return {1: vfps, 2: count}[prop]

        def set(self, prop, value):
            self.pos = int(value)

        def grab(self):
            return self.pos < grabbable

        def read(self):
            if self.pos >= grabbable:
                return False, None
            f = np.zeros((2, 2, 3), dtype=np.uint8)
            f[..., 0], f[..., 1], f[..., 2] = self.pos % 256, self.pos // 256, 7        # "BGR": the index rides in the first two channels
            return True, f
    return types.SimpleNamespace(VideoCapture=Cap, CAP_PROP_FPS=1, CAP_PROP_FRAME_COUNT=2, CAP_PROP_POS_FRAMES=3, COLOR_BGR2RGB=4,
                                 cvtColor=lambda f, code: f[..., ::-1])


def main():
    from typing import List
    src = open(SRC).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "_load_video")
    out = {"file": [], "dir": []}
    for count, grab, vfps, nf, fps in CASES:
        ns = {"os": os, "glob": glob, "np": np, "PIL": PIL, "List": List, "cv2": fake_cv2(count, grab, vfps),
              "logger": types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)}
        exec(compile(ast.get_source_segment(src, fn), "utils/media.py", "exec"), ns)
        frames = ns["_load_video"]("clip.mp4", num_frames=nf, fps=fps)
        idx = [int(np.asarray(f)[0, 0, 2]) + 256 * int(np.asarray(f)[0, 0, 1]) for f in frames]      # (after the stand-in's BGR -> RGB flip)
        out["file"].append({"frame_count": count, "grabbable": grab, "video_fps": vfps, "num_frames": nf, "fps": fps, "indices": idx})
        print("file", count, grab, vfps, nf, fps, "->", idx)
    for n_files, nf in DIR_CASES:
        with tempfile.TemporaryDirectory() as d:
            for i in range(n_files):
                PIL.Image.new("RGB", (2, 2), (i, 0, 0)).save(os.path.join(d, f"frame_{i:04d}.png"))
            ns = {"os": os, "glob": glob, "np": np, "PIL": PIL, "List": List, "cv2": None, "logger": None}
            exec(compile(ast.get_source_segment(src, fn), "utils/media.py", "exec"), ns)
            frames = ns["_load_video"](d, num_frames=nf, fps=0.0)
            idx = [int(np.asarray(f.convert("RGB"))[0, 0, 0]) for f in frames]
        out["dir"].append({"n_files": n_files, "num_frames": nf, "indices": idx})
        print("dir", n_files, nf, "->", idx)
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
