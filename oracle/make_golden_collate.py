"""REFERENCE-EXECUTED fixture for the data-side boundary of the SFT step: what `DataCollator.__call__` hands `model(**batch)` — TEST INFRASTRUCTURE.

SURVEY §2 keeps the data pipeline out of scope ("we feed synthetic batches in the collator's output format", §8d); this pins that FORMAT.  The
reference's `DataCollator` (llava/data/collate.py:13-159) is taken from its file with `ast` and executed unchanged on three synthetic instances of
the dynamic_s2 recipe — a 2 x 2-block image (9 tiles, `block_sizes` [(2, 2)]), a text-only sample, a sample with a one-tile image (`block_sizes`
[None]) and a 3-frame video — with a stub tokenizer (media_tokens / media_token_ids / pad_token_id / model_max_length).  Stored: the padded ids /
labels / mask, the flattened `block_sizes`, and for every media object of the batch its index in the seeded pixel pool, so that the GPU test can
rebuild the exact batch where /root/reference does not exist and call `HipLlavaLlamaModel(**batch)` on it (tests/test_gpu_integration.py).

    python oracle/make_golden_collate.py      # writes tests/golden/collate_batch_ref.npz; needs /root/reference
"""
from __future__ import annotations

import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vila_amd import configs, synthetic      # noqa: E402

REF = "/root/reference/llava"
OUT = os.path.join(ROOT, "tests", "golden", "collate_batch_ref.npz")
SEED = 17
PAD_ID = 999
IGNORE_INDEX = -100


def load_collator():
    from dataclasses import dataclass
    from typing import Any, Dict, Sequence
    src = open(f"{REF}/data/collate.py").read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "DataCollator")
    ns = {"dataclass": dataclass, "Any": Any, "Dict": Dict, "Sequence": Sequence, "torch": torch, "PreTrainedTokenizer": object, "IGNORE_INDEX": IGNORE_INDEX,
          "logger": types.SimpleNamespace(warning=lambda *a, **k: None)}
    exec(compile(ast.get_source_segment(src, node), "data/collate.py", "exec"), ns)
    return ns["DataCollator"]


def instances(cfg):
    """-> (instances in BaseDataset.__getitem__'s form (base.py:99-190), pixel pool [15, 3, H, W])."""
    pool = synthetic.make_pixels(cfg, 9 + 1 + 3 + 2, SEED)                # 9 tiles of the 2 x 2 image | 1 tile | 3 frames | 2 spare
    g = torch.Generator().manual_seed(SEED)
    def sample(n, img_at=(), vid_at=()):
        ids = torch.randint(0, 900, (n,), generator=g)
        for p in img_at:
            ids[p] = cfg.image_token_id
        for p in vid_at:
            ids[p] = cfg.video_token_id
        lab = ids.clone()
        lab[: n // 2] = IGNORE_INDEX
        return ids, lab
    i0, l0 = sample(11, img_at=(1,))
    i1, l1 = sample(7)
    i2, l2 = sample(13, img_at=(6,), vid_at=(2,))
    inst = [
        {"input_ids": i0, "labels": l0, "image": [pool[k] for k in range(9)], "block_sizes": [(2, 2)]},
        {"input_ids": i1, "labels": l1},                                  # text only: no "image" key (collate.py:151 takes len() of it)
        {"input_ids": i2, "labels": l2, "image": [pool[9]], "block_sizes": [None], "video": [pool[10:13]]},
    ]
    return inst, pool


def main():
    cfg = configs.tiny_s2()
    tok = types.SimpleNamespace(media_tokens={"image": "<image>", "video": "<vila/video>"},
                                media_token_ids={"image": cfg.image_token_id, "video": cfg.video_token_id}, pad_token_id=PAD_ID, model_max_length=64)
    inst, pool = instances(cfg)
    batch = load_collator()(tok)(inst)
    assert set(batch) == {"input_ids", "media", "media_config", "labels", "attention_mask", "gt_selection_maps"}
    where = {}
    for k in range(pool.shape[0]):
        where[pool[k].data_ptr()] = k
    img_idx = [where[t.data_ptr()] for t in batch["media"]["image"]]
    vid = batch["media"]["video"]
    vid_first = [where[v[0].data_ptr()] for v in vid]
    fx = {"seed": np.int64(SEED), "pad_id": np.int64(PAD_ID), "input_ids": batch["input_ids"].numpy(), "labels": batch["labels"].numpy(),
          "attention_mask": batch["attention_mask"].numpy(), "image_pool_index": np.array(img_idx), "video_first_pool_index": np.array(vid_first),
          "video_frames": np.array([int(v.shape[0]) for v in vid]),
          "block_sizes": np.array([[-1, -1] if b is None else list(b) for b in batch["media_config"]["image"]["block_sizes"]]),
          "media_config_keys": np.array(sorted(batch["media_config"])), "image_config_keys": np.array(sorted(batch["media_config"]["image"])),
          "gt_selection_maps_is_none": np.bool_(batch["gt_selection_maps"] is None)}
    np.savez_compressed(OUT, **fx)
    print("input_ids", tuple(batch["input_ids"].shape), "images", img_idx, "videos", vid_first, fx["video_frames"].tolist(), "block_sizes",
          batch["media_config"]["image"]["block_sizes"])
    print("wrote", OUT)


if __name__ == "__main__":
    main()
