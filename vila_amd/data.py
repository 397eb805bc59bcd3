"""Raw samples -> the batch `HipLlavaLlamaModel.forward(**batch)` / `SFTTrainer` take: the data-side producer of the SFT step, host-only.

  * `build_instance`  — one conversation with its pictures -> `{"input_ids", "labels", "image", "block_sizes", ...}`: what
                        `BaseDataset.__getitem__` returns (llava/data/base.py:99-190) for image media under every `image_aspect_ratio`
                        recipe (tiles + block sizes for dynamic_s2, tiles + one `<image>\\n` per tile for dynamic, whole pictures otherwise)
  * `DataCollator`    — instances -> padded ids / labels / mask, the media of all samples in row order, flattened block sizes
                        (llava/data/collate.py:13-159), including its truncation rules when `model_max_length` cuts media tokens off

Pinned by tests/golden/collate_cases_ref.json (= the reference's own DataCollator, ast-extracted and executed on integer-tagged instances) and
by the fixtures of the pieces `build_instance` is made of (conversation_ref.json, dynamic_tiles.npz, s2_tiles.npz).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from .configs import IGNORE_INDEX
from .conversation import preprocess_conversation

_log = logging.getLogger(__name__)


def build_instance(conversation: Sequence[Dict[str, Any]], cfg, tokenizer, no_system_prompt: bool = False) -> Dict[str, Any]:
    """conversation: messages `{"from": "human" | "gpt", "value": str | [str | image, ...]}`.  Media are taken out of every message in order
    (`extract_media`, llava/utils/media.py:93-122 — each picture leaves one `<image>` in the text), then processed as the dataset does
    (base.py:112-127): dynamic_s2 -> the tiles of every scale + one block size per picture; dynamic -> grid tiles + thumbnail and the FIRST
    message's text re-written with one `<image>\\n` per tile (mm_utils.py:408-424); else whole pictures.  Labels: `preprocess_conversation`."""
    from . import serving
    conv, pictures = [], []
    for m in conversation:
        text, imgs = serving._split_prompt(m["value"])
        conv.append({"from": m["from"], "value": text})
        pictures.extend(imgs)
    out: Dict[str, Any] = {}
    mode = cfg.aspect_mode
    if pictures:
        if mode == "dynamic_s2":
            tiles, blocks = [], []
            for p in pictures:
                t, b = serving.process_image(p, cfg, enable_dynamic_s2=True)
                tiles.append(t)
                blocks.append(b)
            out["image"], out["block_sizes"] = torch.cat(tiles), blocks
        elif mode == "dynamic":
            tiles = [serving.process_image(p, cfg, enable_dynamic_res=True) for p in pictures]
            pieces = conv[0]["value"].split(serving.IMAGE_TOKEN)
            text = pieces[0]
            for k, t in enumerate(tiles):                               # (pictures are expected in the first message, as in the reference)
                text += f"{serving.IMAGE_TOKEN}\n" * t.shape[0] + (pieces[k + 1] if k + 1 < len(pieces) else "")
            conv[0]["value"] = text + "".join(pieces[len(tiles) + 1:])
            out["image"] = torch.cat(tiles)
        else:
            out["image"] = serving.process_images(pictures, cfg)
        out["original_image_sizes"] = [tuple(serving._to_pil(p).size) for p in pictures]
    out.update(preprocess_conversation(conv, tokenizer, no_system_prompt=no_system_prompt))
    return out


def _per_sample(instances, names):
    """Instances (single samples, or instances that already are a batch: lists per field) -> one record per sample."""
    for inst in instances:
        single = isinstance(inst["input_ids"], torch.Tensor)
        n = 1 if single else len(inst["input_ids"])
        wrap = (lambda v: [v]) if single else (lambda v: list(v))
        media = {}
        for name in names:
            objs = inst.get(name)
            media[name] = [[] for _ in range(n)] if objs is None else [list(o) for o in wrap(objs)]
        if "block_sizes" in inst:
            blocks = [list(b) for b in wrap(inst["block_sizes"])]
        else:
            blocks = [[None] * len(objs) for objs in media["image"]] if "image" in media else [[] for _ in range(n)]
        sizes = inst.get("original_image_sizes")
        n_img = [len(objs) for objs in (wrap(inst["image"]) if inst.get("image") is not None else [[] for _ in range(n)])]
        sizes = [list(s) for s in wrap(sizes)] if sizes is not None else [[None] * k for k in n_img]
        for k, (ids, lab) in enumerate(zip(wrap(inst["input_ids"]), wrap(inst["labels"]))):
            yield {"ids": ids, "labels": lab, "media": {name: media[name][k] for name in names}, "blocks": blocks[k], "sizes": sizes[k]}


class DataCollator:
    """`DataCollator(tokenizer)(instances) -> batch` with the reference's keys.  The tokenizer supplies `media_tokens`, `media_token_ids`,
    `pad_token_id` and `model_max_length` (what `build_llm_and_tokenizer` / `checkpoint.load_tokenizer` attach)."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, instances: Sequence[Dict[str, Any]]) -> Dict[str, Any]:
        tk = self.tokenizer
        names = list(tk.media_tokens)
        rows = list(_per_sample(instances, names))

        def tiled(r, name):                                              # a dynamic_s2 sample: its pictures are counted by block sizes
            return name == "image" and any(b is not None for b in r["blocks"])

        def count(ids, name) -> int:
            return int((ids == tk.media_token_ids[name]).sum().item())
        for name in names:                                               # collate.py:55-66
            for r in rows:
                have = len(r["blocks"]) if tiled(r, name) else len(r["media"][name])
                want = count(r["ids"], name)
                if have != want:
                    raise ValueError(f"Number mismatch between {name} objects and {name} tokens. There are {want} {name} tokens but {have} {name} objects.")
        pad = torch.nn.utils.rnn.pad_sequence
        input_ids = pad([r["ids"] for r in rows], batch_first=True, padding_value=tk.pad_token_id)[:, : tk.model_max_length]
        labels = pad([r["labels"] for r in rows], batch_first=True, padding_value=IGNORE_INDEX)[:, : tk.model_max_length]
        attention_mask = input_ids.ne(tk.pad_token_id)
        media: Dict[str, List[Any]] = {}
        for name in names:                                               # media whose token fell behind model_max_length go (collate.py:84-110)
            media[name] = []
            for k, r in enumerate(rows):
                objs, left = r["media"][name], count(input_ids[k], name)
                if tiled(r, name):
                    big = sum(x * y for x, y in r["blocks"])
                    small_each = (len(objs) - big) // len(r["blocks"])
                    keep = sum(x * y for x, y in r["blocks"][:left]) + small_each * left
                    r["blocks"] = r["blocks"][:left]
                else:
                    keep = left
                    if name == "image":
                        r["blocks"] = r["blocks"][:left]
                if len(objs) > keep:
                    _log.warning(f"Truncating the number of {name} objects from {len(objs)} to {keep}")
                media[name].extend(objs[:keep])
        maps = [inst.get("gt_selection_map") for inst in instances]
        assert all(m is not None for m in maps) or all(m is None for m in maps)      # grounding data and regular data do not mix (collate.py:138-140)
        return {"input_ids": input_ids, "media": media,
                "media_config": {"image": {"block_sizes": [b for r in rows for b in r["blocks"]],
                                           "original_image_sizes": [s for r in rows for s in r["sizes"]]}, "video": {}},
                "labels": labels, "attention_mask": attention_mask,
                "gt_selection_maps": torch.stack(maps, dim=0) if maps and maps[0] is not None else None}


# ----------------------------------------------------------------------------------------------------------------------
# datasets on disk -> instances (the front of `run.train`): the LLaVA-format json every NVILA SFT mixture is made of
# ----------------------------------------------------------------------------------------------------------------------
def remove_media_tokens(text: str) -> str:
    """llava/data/dataset_impl/utils.py:10-13: the json's own `<image>` / `<video>` markers go; the media are re-attached in front."""
    for token in ("<image>", "<video>"):
        text = text.replace(token + "\n", "").replace("\n" + token, "").replace(token, "")
    return text.strip()


def pad_to_global_batch(n: int, global_batch_size: Optional[int]) -> Tuple[int, int]:
    """How `LLaVADataset.__init__` (dataset_impl/llava.py:30-38) grows a dataset of n samples so that it fills whole global batches:
    -> (times the instance list is repeated first, samples then drawn at random from the repeated list and appended)."""
    if global_batch_size is None:
        return 1, 0
    times = 1
    residual = global_batch_size - n % global_batch_size
    if residual != global_batch_size:
        if global_batch_size // n >= 2:
            times = global_batch_size // n
            residual = global_batch_size - (n * times) % global_batch_size
        return times, residual
    return 1, 0


class LLaVADataset:
    """`llava.data.LLaVADataset` (dataset_impl/llava.py:16-74): a json list of `{"conversations": [...], "image" | "images" | "video": path(s)}`
    records under `media_dir`.  `process` re-attaches the pictures in front of the first (human) message; `__getitem__` is
    `BaseDataset.__getitem__` = `build_instance` (with its resample-on-failure rule, base.py:183-188).  Video records are refused: frame
    sampling from video files is outside this library's path (the serving shim's `load_video_frames` shows the rule)."""

    def __init__(self, data_path: str, media_dir: Optional[str], cfg, tokenizer, global_batch_size: Optional[int] = None,
                 no_system_prompt: bool = False, max_num_images: Optional[int] = None, resample_on_failure: bool = True, seed: Optional[int] = None):
        import json
        import random
        self.data_path, self.media_dir, self.cfg, self.tokenizer = data_path, media_dir or "", cfg, tokenizer
        self.no_system_prompt, self.max_num_images, self.resample_on_failure = no_system_prompt, max_num_images, resample_on_failure
        with open(data_path) as fh:
            self.instances = [json.loads(l) for l in fh if l.strip()] if data_path.endswith(".jsonl") else json.load(fh)
        self._rng = random.Random(seed)                                   # (the reference draws from the global `random` state)
        times, extra = pad_to_global_batch(len(self.instances), global_batch_size)
        self.instances = self.instances * times
        self.instances.extend([self.instances[i] for i in self._rng.sample(range(len(self.instances)), extra)])

    def __len__(self) -> int:
        return len(self.instances)

    def process(self, instance: Dict[str, Any]) -> List[Dict[str, Any]]:
        import copy
        import os
        from PIL import Image
        messages = copy.deepcopy(instance["conversations"])
        if "video" in instance:
            raise NotImplementedError("video records: decode the frames upstream and pass them as pictures")
        medias = []
        for key in ("image", "images"):
            if key in instance:
                paths = instance[key] if isinstance(instance[key], (list, tuple)) else [instance[key]]
                medias.extend(os.path.join(self.media_dir, p) for p in paths)
                if self.max_num_images is not None:
                    medias = medias[: min(self.max_num_images, len(medias))]
        for m in messages:
            m["value"] = remove_media_tokens(m["value"])
        if messages[0]["from"] != "human":
            raise ValueError(f"First message is not from human: {messages}")
        messages[0]["value"] = [Image.open(p).convert("RGB") for p in medias] + [messages[0]["value"]]
        return messages

    def __getitem__(self, index: int) -> Dict[str, Any]:
        try:
            return build_instance(self.process(self.instances[index]), self.cfg, self.tokenizer, no_system_prompt=self.no_system_prompt)
        except Exception as e:
            if not self.resample_on_failure:
                raise
            _log.exception("Error processing instance '%s': '%s'. Resampling.", self.instances[index], e)
            return self[self._rng.randint(0, len(self.instances) - 1)]


class RepeatedDataset:
    """`name*3` in a mixture (llava/data/builder.py:65-76)."""
    def __init__(self, dataset, times: int):
        self.dataset, self.times = dataset, times

    def __len__(self) -> int:
        return len(self.dataset) * self.times

    def __getitem__(self, index: int):
        return self.dataset[index % len(self.dataset)]


class ConcatDataset:
    """The mixture as one index space; `sample_lens` is what `VILADistributedSampler` needs to balance the datasets over the ranks."""
    def __init__(self, datasets: Sequence[Any]):
        self.datasets = list(datasets)
        self.sample_lens = [len(d) for d in self.datasets]

    def __len__(self) -> int:
        return sum(self.sample_lens)

    def __getitem__(self, index: int):
        if index < 0 or index >= len(self):
            raise IndexError(index)
        for d, n in zip(self.datasets, self.sample_lens):
            if index < n:
                return d[index]
            index -= n


def parse_mixture(mixture: str, mixtures: Optional[Dict[str, List[str]]] = None) -> List[str]:
    """`a+b+c`, names of registered mixtures expanded until none is left, sorted (llava/data/builder.py:58-62)."""
    mixtures = mixtures or {}
    names = mixture.split("+") if "+" in mixture else [mixture]
    while any(n in mixtures for n in names):
        names = [x for n in names for x in mixtures.get(n, [n])]
    return sorted(names)


def build_dataset(mixture: str, registry: Dict[str, Dict[str, Any]], cfg, tokenizer, global_batch_size: Optional[int] = None,
                  mixtures: Optional[Dict[str, List[str]]] = None, seed: Optional[int] = None) -> ConcatDataset:
    """`build_dataset` of llava/data/builder.py:85-151 for registries of LLaVA-format datasets: every name of the mixture (`name*times` repeats
    it) is instantiated from its registry entry `{"_target_": "llava.data.LLaVADataset", "data_path": ..., "media_dir": ...}` with the global
    batch size the run uses, and the datasets are concatenated in the mixture's (sorted) order."""
    out = []
    for name in parse_mixture(mixture, mixtures):
        times = 1
        if "@" in name:
            raise NotImplementedError("subset slicing by a filter index (`name@subset`)")
        if "*" in name:
            name, t = name.split("*")
            times = int(t)
        if name not in registry:
            raise ValueError(f"Dataset '{name}' is not found in the registries.")
        ent = dict(registry[name])
        target = ent.pop("_target_", "llava.data.LLaVADataset")
        if target.rsplit(".", 1)[-1] != "LLaVADataset":
            raise NotImplementedError(f"dataset class '{target}'")
        known = {k: ent[k] for k in ("data_path", "media_dir", "no_system_prompt", "max_num_images", "resample_on_failure") if k in ent}
        d = LLaVADataset(cfg=cfg, tokenizer=tokenizer, global_batch_size=global_batch_size, seed=seed, **known)
        out.append(RepeatedDataset(d, times) if times > 1 else d)
    return ConcatDataset(out)
