"""REFERENCE-EXECUTED fixture for every branch of `DataCollator.__call__` (llava/data/collate.py:13-159) — TEST INFRASTRUCTURE.

`oracle/make_golden_collate.py` pins the FORMAT of one batch for the GPU test; this file pins the collator's LOGIC for the HIP-side mirror
(`vila_amd/data.py: DataCollator`): the class is taken out of its file with `ast` and executed unchanged on integer-tagged instances (every
media object is a one-element tensor holding its own serial number, so the fixture can say exactly which objects survived, in which order):
  single        three single-sample instances: a tiled dynamic_s2 image, text only, a one-tile image + a video
  prebatched    an instance that is already a batch (lists of ids / labels / media per sample) next to a single one
  truncated     `model_max_length` cuts a row behind its first image token: objects and block sizes beyond it are dropped (both the plain
                count rule and the dynamic_s2 tiles rule, collate.py:84-110)
  sizes         `original_image_sizes` passed through / defaulted to None per image
  mismatch      more image objects than image tokens before any truncation: the ValueError and its text (collate.py:55-66)

    python oracle/make_golden_collate_cases.py       # writes tests/golden/collate_cases_ref.json; needs /root/reference
"""
from __future__ import annotations

import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden_collate import load_collator  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "collate_cases_ref.json")
IMG, VID, PAD = 900, 901, 999


def obj(k):
    return torch.tensor([k])


def ids(*toks):
    return torch.tensor(list(toks), dtype=torch.int64)


def lab(t, n_ignored):
    l = t.clone()
    l[:n_ignored] = -100
    return l


def cases():
    """name -> (model_max_length, instances as plain python: tensors are rebuilt by `materialise`)."""
    c = {}
    c["single"] = (64, [
        {"input_ids": [5, IMG, 7, 8, 9, 10], "ignored": 3, "image": list(range(0, 9)), "block_sizes": [[2, 2]]},
        {"input_ids": [11, 12, 13], "ignored": 1},
        {"input_ids": [14, VID, 15, IMG, 16, 17, 18, 19], "ignored": 4, "image": [20], "block_sizes": [None], "video": [30]},
    ])
    c["prebatched"] = (64, [
        {"input_ids": [[5, IMG, 6], [7, 8, IMG, IMG, 9]], "ignored": [1, 2], "image": [[40], [41, 42]], "video": [[], []]},
        {"input_ids": [21, IMG, 22, 23], "ignored": 2, "image": [43]},
        {"input_ids": [[24, 25], [26, VID]], "ignored": [0, 1], "video": [[], [50]]},
    ])
    c["truncated"] = (6, [
        {"input_ids": [5, IMG, 6, 7, 8, 9, IMG, 10], "ignored": 2, "image": [60, 61]},                      # the second <image> is cut off
        {"input_ids": [5, IMG, 6, 7, 8, 9, IMG, 10], "ignored": 2, "image": list(range(70, 70 + 10 + 7)), "block_sizes": [[2, 2], [1, 2]]},
        {"input_ids": [1, 2, 3], "ignored": 0},
    ])
    c["sizes"] = (64, [
        {"input_ids": [5, IMG, IMG], "ignored": 0, "image": [80, 81], "original_image_sizes": [[640, 480], [100, 200]]},
        {"input_ids": [6, IMG], "ignored": 0, "image": [82]},
        {"input_ids": [[7, IMG]], "ignored": [0], "image": [[83]], "original_image_sizes": [[[3, 4]]]},
    ])
    c["mismatch"] = (64, [{"input_ids": [5, IMG, 6], "ignored": 0, "image": [90, 91]}])
    c["mismatch_s2"] = (64, [{"input_ids": [5, IMG, 6], "ignored": 0, "image": list(range(18)), "block_sizes": [[2, 2], [2, 2]]}])
    return c


def materialise(inst):
    """Plain python -> what BaseDataset.__getitem__ returns (base.py:99-190): tensors for ids / labels / media objects, tuples for block sizes."""
    out = {}
    pre = isinstance(inst["input_ids"][0], list)
    if pre:
        out["input_ids"] = [ids(*r) for r in inst["input_ids"]]
        out["labels"] = [lab(t, n) for t, n in zip(out["input_ids"], inst["ignored"])]
    else:
        out["input_ids"] = ids(*inst["input_ids"])
        out["labels"] = lab(out["input_ids"], inst["ignored"])
    for name in ("image", "video"):
        if name in inst:
            out[name] = [[obj(k) for k in row] for row in inst[name]] if pre else [obj(k) for k in inst[name]]
    if "block_sizes" in inst:
        conv = lambda b: None if b is None else tuple(b)
        out["block_sizes"] = [[conv(b) for b in row] for row in inst["block_sizes"]] if pre else [conv(b) for b in inst["block_sizes"]]
    if "original_image_sizes" in inst:
        out["original_image_sizes"] = inst["original_image_sizes"]
    return out


def tokenizer(max_len):
    return types.SimpleNamespace(media_tokens={"image": "<image>", "video": "<vila/video>"}, media_token_ids={"image": IMG, "video": VID},
                                 pad_token_id=PAD, model_max_length=max_len)


def main():
    Collator = load_collator()
    fx = {"image_id": IMG, "video_id": VID, "pad_id": PAD, "cases": {}}
    for name, (max_len, insts) in cases().items():
        rec = {"model_max_length": max_len, "instances": insts}
        try:
            b = Collator(tokenizer(max_len))([materialise(i) for i in insts])
            rec.update(input_ids=b["input_ids"].tolist(), labels=b["labels"].tolist(), attention_mask=b["attention_mask"].tolist(),
                       image=[int(t) for t in b["media"]["image"]], video=[int(t) for t in b["media"]["video"]],
                       block_sizes=[None if x is None else list(x) for x in b["media_config"]["image"]["block_sizes"]],
                       original_image_sizes=b["media_config"]["image"]["original_image_sizes"], video_config=b["media_config"]["video"],
                       gt_selection_maps=b["gt_selection_maps"], keys=sorted(b))
            print(name, "ids", [len(r) for r in rec["input_ids"]], "image", rec["image"], "video", rec["video"], "blocks", rec["block_sizes"])
        except ValueError as e:
            rec["error"] = str(e)
            print(name, "ValueError:", e)
        fx["cases"][name] = rec
    json.dump(fx, open(OUT, "w"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
