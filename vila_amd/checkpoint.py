"""Checkpoint I/O in the reference's three-folder layout (SURVEY.md §5 "Checkpoint / resume", §8f row 4).

`LlavaMetaModel.save_pretrained` (llava/model/llava_arch.py:158-204) splits ONE state_dict by key prefix into
    <dir>/llm/            keys `model.embed_tokens.weight`, `model.layers.N....`, `lm_head.weight`        (HF Qwen2 layout)
    <dir>/vision_tower/   keys `vision_model....`   (prefix `vision_tower.vision_tower.` stripped, :177-179)
    <dir>/mm_projector/   keys `layers.1.weight` ...
plus a top-level config.json whose `llm_cfg / vision_tower_cfg / mm_projector_cfg` point at the sub-folders
(llava/model/utils/utils.py:25-55).  Weights are HF safetensors (`model.safetensors` or shards + `model.safetensors.index.json`).
The HIP modules keep exactly these key names, so loading is a name-for-name copy straight into the (fused / flat) parameter
storage; unknown keys of real checkpoints (`vision_model.head.*`: the pooling head VILA never calls) are ignored.
"""
from __future__ import annotations

import json
import os
from typing import Optional, Dict, Iterable

import torch
from safetensors import safe_open
from safetensors.torch import save_file

from .configs import LlmConfig, VilaConfig, VisionConfig

_PARTS = (("llm", "llm", ""), ("vision_tower", "vision_tower", "vision_tower."), ("mm_projector", "mm_projector", ""))


def _shards(items: Iterable, max_bytes: int):
    cur, size = {}, 0
    for k, t in items:
        n = t.numel() * t.element_size()
        if cur and size + n > max_bytes:
            yield cur
            cur, size = {}, 0
        cur[k] = t
        size += n
    if cur:
        yield cur


def _save_folder(folder: str, sd: Dict[str, torch.Tensor], cfg_json: dict, max_shard_bytes: int) -> None:
    os.makedirs(folder, exist_ok=True)
    # tied / view tensors must be materialised separately for safetensors
    items = [(k, v.detach().to("cpu").contiguous().clone()) for k, v in sd.items()]
    shards = list(_shards(items, max_shard_bytes))
    if len(shards) == 1:
        save_file(shards[0], os.path.join(folder, "model.safetensors"), metadata={"format": "pt"})
    else:
        wmap, total = {}, 0
        for i, sh in enumerate(shards):
            name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(folder, name), metadata={"format": "pt"})
            for k, t in sh.items():
                wmap[k] = name
                total += t.numel() * t.element_size()
        with open(os.path.join(folder, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": wmap}, f, indent=1)
    with open(os.path.join(folder, "config.json"), "w") as f:
        json.dump(cfg_json, f, indent=1)


def _folder_tensors(folder: str):
    idx = os.path.join(folder, "model.safetensors.index.json")
    files = sorted(set(json.load(open(idx))["weight_map"].values())) if os.path.exists(idx) else ["model.safetensors"]
    for fn in files:
        with safe_open(os.path.join(folder, fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def save_pretrained(model, output_dir: str, max_shard_bytes: int = 5 << 30) -> None:
    """Write the model in the reference's layout (llava_arch.py:158-204)."""
    cfg: VilaConfig = model.cfg
    c, v = cfg.llm, cfg.vision
    llm_cfg = {"model_type": "qwen2", "architectures": ["Qwen2ForCausalLM"], "hidden_size": c.hidden_size, "intermediate_size": c.intermediate_size,
               "num_hidden_layers": c.num_hidden_layers, "num_attention_heads": c.num_attention_heads, "num_key_value_heads": c.num_key_value_heads,
               "vocab_size": c.vocab_size, "rms_norm_eps": c.rms_norm_eps, "rope_theta": c.rope_theta, "tie_word_embeddings": c.tie_word_embeddings,
               "eos_token_id": c.eos_token_id, "torch_dtype": "bfloat16", "hidden_act": "silu"}
    vt_cfg = {"model_type": "siglip_vision_model", "hidden_size": v.hidden_size, "intermediate_size": v.intermediate_size,
              "num_hidden_layers": v.num_hidden_layers, "num_attention_heads": v.num_attention_heads, "image_size": v.image_size,
              "patch_size": v.patch_size, "num_channels": v.num_channels, "layer_norm_eps": v.layer_norm_eps, "hidden_act": "gelu_pytorch_tanh",
              # VILA never uses SigLIP's pooling head; saying so makes the reference's SiglipVisionModel (modeling_siglip.py:1166) build none, so
              # its from_pretrained finds every parameter it has in this folder (no "newly initialized" head)
              "vision_use_head": False}
    pj_cfg = {"model_type": "v2l_projector", "mm_projector_type": cfg.mm_projector_type}
    for (attr, folder, strip), sub_cfg in zip(_PARTS, (llm_cfg, vt_cfg, pj_cfg)):
        sd = getattr(model, attr).state_dict()
        sd = {(k[len(strip):] if strip and k.startswith(strip) else k): t for k, t in sd.items()}
        if attr == "llm" and c.tie_word_embeddings:
            sd.pop("lm_head.weight", None)
        _save_folder(os.path.join(output_dir, folder), sd, sub_cfg, max_shard_bytes)
    top = {"model_type": "llava_llama", "architectures": ["LlavaLlamaModel"], "llm_cfg": llm_cfg, "vision_tower_cfg": vt_cfg,
           "mm_projector_cfg": pj_cfg, "mm_projector_type": cfg.mm_projector_type, "mm_vision_select_layer": v.select_layer,
           "mm_vision_select_feature": "cls_patch", "dynamic_s2": cfg.dynamic_s2, "s2_scales": ",".join(str(s) for s in cfg.s2_scales),
           "s2_resize_output_to_scale_idx": cfg.s2_resize_output_to_scale_idx, "image_aspect_ratio": cfg.image_aspect_ratio or None,
           "min_tiles": cfg.min_tiles, "max_tiles": cfg.max_tiles, "video_max_tiles": cfg.video_max_tiles, "chat_template": cfg.chat_template or None,
           "hidden_size": c.hidden_size,
           "mm_hidden_size": cfg.mm_hidden_size, "image_token_id": cfg.image_token_id, "video_token_id": cfg.video_token_id, "newline_token_id": cfg.newline_token_id,
           "_name_or_path": output_dir}
    with open(os.path.join(output_dir, "config.json"), "w") as f:
        json.dump(top, f, indent=1)
    tok = getattr(model, "tokenizer", None)                     # llava_arch.py:164-165: the tokenizer lives in llm/ (a synthetic stand-in has nothing to save)
    if tok is not None and hasattr(tok, "save_pretrained"):
        tok.save_pretrained(os.path.join(output_dir, "llm"))


def resolve_projector_type(top: dict, model_dir: str = "") -> str:
    """The reference's LlavaConfig has NO top-level `mm_projector_type`: the type lives in the projector's own config
    (`MultimodalProjectorConfig.mm_projector_type`, base_projector.py:126-131), reachable as the `mm_projector_cfg` dict of the top
    config.json or as <dir>/mm_projector/config.json (llava/model/utils/utils.py:25-55).  Our own save_pretrained also writes a
    top-level copy; it is only the last fallback."""
    pj = top.get("mm_projector_cfg")
    if isinstance(pj, dict) and pj.get("mm_projector_type"):
        return pj["mm_projector_type"]
    for path in ([pj] if isinstance(pj, str) else []) + ([os.path.join(model_dir, "mm_projector")] if model_dir else []):
        f = os.path.join(path, "config.json")
        if os.path.exists(f):
            t = json.load(open(f)).get("mm_projector_type")
            if t:
                return t
    return top.get("mm_projector_type") or "mlp_downsample"


def config_from_pretrained(model_dir: str) -> VilaConfig:
    top = json.load(open(os.path.join(model_dir, "config.json")))

    def sub(key, folder):
        cfg = top.get(key)
        if isinstance(cfg, dict):
            return cfg
        path = cfg if isinstance(cfg, str) else os.path.join(model_dir, folder)
        return json.load(open(os.path.join(path, "config.json")))
    l, v = sub("llm_cfg", "llm"), sub("vision_tower_cfg", "vision_tower")

    def opt(d, key, default):
        """A key the reference's LlavaConfig writes as null (every field it was not given: configuration_llava.py:26-70) counts as absent."""
        val = d.get(key)
        return default if val is None else val
    llm = LlmConfig(hidden_size=l["hidden_size"], intermediate_size=l["intermediate_size"], num_hidden_layers=l["num_hidden_layers"],
                    num_attention_heads=l["num_attention_heads"], num_key_value_heads=l["num_key_value_heads"],
                    head_dim=opt(l, "head_dim", l["hidden_size"] // l["num_attention_heads"]), vocab_size=l["vocab_size"],
                    rms_norm_eps=opt(l, "rms_norm_eps", 1e-6), rope_theta=_rope_theta(l),
                    tie_word_embeddings=bool(opt(l, "tie_word_embeddings", False)), eos_token_id=_first_id(opt(l, "eos_token_id", 151645)))
    vis = VisionConfig(hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"], num_hidden_layers=v["num_hidden_layers"],
                       num_attention_heads=v["num_attention_heads"], image_size=v["image_size"], patch_size=v["patch_size"],
                       num_channels=opt(v, "num_channels", 3), layer_norm_eps=opt(v, "layer_norm_eps", 1e-6),
                       select_layer=opt(top, "mm_vision_select_layer", -2))
    scales = opt(top, "s2_scales", "448,896,1344")
    return VilaConfig(vision=vis, llm=llm, mm_projector_type=resolve_projector_type(top, model_dir),
                      image_token_id=opt(top, "image_token_id", 151649), video_token_id=opt(top, "video_token_id", 151650),
                      newline_token_id=opt(top, "newline_token_id", 198),
                      dynamic_s2=bool(opt(top, "dynamic_s2", False)), s2_scales=tuple(int(s) for s in str(scales).split(",")),
                      s2_resize_output_to_scale_idx=opt(top, "s2_resize_output_to_scale_idx", -1),
                      image_aspect_ratio=str(opt(top, "image_aspect_ratio", "")), min_tiles=int(opt(top, "min_tiles", 1)),
                      max_tiles=int(opt(top, "max_tiles", 12)), video_max_tiles=int(opt(top, "video_max_tiles", 1)),
                      chat_template=str(opt(top, "chat_template", "")), name=os.path.basename(model_dir.rstrip("/")))


def _rope_theta(l: dict) -> float:
    """`rope_theta` at the top of the LLM config (transformers <= 4.x, what the reference pins) or inside `rope_parameters` (5.x)."""
    if l.get("rope_theta") is not None:
        return float(l["rope_theta"])
    rp = l.get("rope_parameters") or {}
    return float(rp.get("rope_theta", 1e6))


def _first_id(v) -> int:
    return int(v[0]) if isinstance(v, (list, tuple)) else int(v)


def load_weights_into(model, model_dir: str, strict: bool = True) -> Dict[str, list]:
    """Name-for-name copy of the three folders into the (already constructed) HIP modules."""
    report = {"missing": [], "ignored": []}
    for attr, folder, strip in _PARTS:
        mod = getattr(model, attr)
        params = dict(mod.named_parameters())
        seen = set()
        with torch.no_grad():
            for k, t in _folder_tensors(os.path.join(model_dir, folder)):
                name = strip + k
                if name in params:
                    params[name].copy_(t.to(params[name].dtype))
                    seen.add(name)
                else:
                    report["ignored"].append(f"{folder}/{k}")          # e.g. vision_model.head.* (pooling head, unused by VILA)
        tied = attr == "llm" and model.cfg.llm.tie_word_embeddings
        report["missing"] += [f"{folder}/{n}" for n in params if n not in seen and not (tied and n == "lm_head.weight")]
    if strict and report["missing"]:
        raise KeyError(f"checkpoint {model_dir} lacks parameters: {report['missing'][:8]}{' ...' if len(report['missing']) > 8 else ''}")
    return report


MEDIA_TOKENS = {"image": "<image>", "video": "<vila/video>"}          # llava/constants.py:32-35


def load_tokenizer(model_dir: str, model_max_length: Optional[int] = None, chat_template: Optional[str] = None):
    """The tokenizer half of `build_llm_and_tokenizer` (language_model/builder.py:190-211): `AutoTokenizer.from_pretrained(<dir>/llm,
    padding_side="right", use_fast=True, legacy=False)`, `model_max_length`, the config's named chat template, the stop tokens read off it,
    the media tokens added as special tokens and their ids recorded in `media_token_ids` (`conversation.prepare_tokenizer`).  None when the
    folder holds no tokenizer files or transformers is absent (the model then keeps its stand-in)."""
    llm_dir = os.path.join(model_dir, "llm")
    if not any(os.path.exists(os.path.join(llm_dir, f)) for f in ("tokenizer.json", "tokenizer_config.json", "tokenizer.model", "vocab.json")):
        return None
    try:
        from transformers import AutoTokenizer
    except ImportError:
        return None
    from .conversation import prepare_tokenizer
    tok = AutoTokenizer.from_pretrained(llm_dir, padding_side="right", use_fast=True, legacy=False)
    if model_max_length is not None:
        tok.model_max_length = model_max_length
    return prepare_tokenizer(tok, chat_template or None, MEDIA_TOKENS)


def load_pretrained(model_dir: str, device="cuda", dtype=torch.bfloat16):
    """`llava.load(model_path)` for the HIP model: config from config.json, weights from the three sub-folders, the tokenizer from llm/ when one
    was saved there — the media-token and newline ids of the config then follow THAT tokenizer, as in the reference (the ids are not stored in its
    config.json)."""
    from .vlm import HipLlavaLlamaModel
    cfg = config_from_pretrained(model_dir)
    tok = load_tokenizer(model_dir, chat_template=cfg.chat_template or None)
    if tok is not None:
        cfg.image_token_id, cfg.video_token_id = int(tok.media_token_ids["image"]), int(tok.media_token_ids["video"])
        nl = tok("\n").input_ids
        if len(nl) == 1:
            cfg.newline_token_id = int(nl[0])
    model = HipLlavaLlamaModel(cfg, device, dtype, tokenizer=tok)
    load_weights_into(model, model_dir)
    return model


# ----------------------------------------------------------------------------------------------------------------------
# optimizer state (resume of an SFT run: the three model folders hold bf16 weights only)
# ----------------------------------------------------------------------------------------------------------------------
def save_optimizer(trainer, output_dir: str, max_shard_bytes: int = 4 << 30) -> None:
    """<dir>/optimizer/: fp32 master copy, exp_avg, exp_avg_sq of the flat parameter buffer (sharded safetensors) + the step count and
    the name -> (offset, numel, shape) index, so that a resume does not depend on the in-memory layout of this build."""
    f = trainer.flat
    if f.master is None:
        raise ValueError("the trainer was built without optimizer state")
    folder = os.path.join(output_dir, "optimizer")
    os.makedirs(folder, exist_ok=True)
    per = max(1, max_shard_bytes // 4)
    files = []
    for key, t in (("master", f.master), ("exp_avg", f.m), ("exp_avg_sq", f.v)):
        for i, o in enumerate(range(0, f.numel, per)):
            name = f"{key}-{i:05d}.safetensors"
            save_file({key: t[o:o + per].detach().to("cpu").contiguous()}, os.path.join(folder, name), metadata={"format": "pt"})
            files.append({"file": name, "key": key, "offset": o, "numel": int(min(per, f.numel - o))})
    meta = {"step": int(f.step_count), "numel": int(f.numel), "files": files, "bucket_steps": dict(getattr(f, "bucket_steps", {})),
            "bucket_step_floor": int(getattr(f, "bucket_step_floor", 0)),
            "index": {n: [int(o), int(k), list(shape)] for n, (o, k, shape) in f.index.items()},
            "hyper": {"lr": trainer.lr, "betas": list(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.wd}}
    with open(os.path.join(folder, "optimizer.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


def load_optimizer(trainer, model_dir: str) -> None:
    """Inverse of save_optimizer.  Tensors are matched BY NAME through the saved index (a different flat layout is fine); the bf16
    parameters are re-derived from the restored master copy, as an AdamW step would leave them."""
    f = trainer.flat
    folder = os.path.join(model_dir, "optimizer")
    meta = json.load(open(os.path.join(folder, "optimizer.json")))
    bufs = {"master": f.master, "exp_avg": f.m, "exp_avg_sq": f.v}
    # validate the whole state BEFORE the first byte is copied into the live buffers: every tensor of this model present with the shape
    # it has here, every file entry inside the saved extent (a failed load used to leave half-overwritten state behind)
    missing = [n for n in f.index if n not in meta["index"]]
    if missing:
        raise KeyError(f"optimizer state lacks tensors: {missing[:8]}")
    bad = [n for n, (o, k, shape) in f.index.items() if int(meta["index"][n][1]) != int(k) or list(meta["index"][n][2]) != list(shape)]
    if bad:
        n = bad[0]
        raise ValueError(f"optimizer state tensor '{n}' has shape {meta['index'][n][2]}, the model has {list(f.index[n][2])} "
                         f"({len(bad)} mismatching tensors)")
    for ent in meta["files"]:
        if ent["key"] not in bufs or ent["offset"] < 0 or ent["offset"] + ent["numel"] > meta["numel"]:
            raise ValueError(f"optimizer state file entry out of range: {ent}")
    same_layout = meta["numel"] == f.numel and all(n in f.index and list(f.index[n][:2]) == v[:2] for n, v in meta["index"].items())
    with torch.no_grad():
        for ent in meta["files"]:
            with safe_open(os.path.join(folder, ent["file"]), framework="pt", device="cpu") as fh:
                chunk = fh.get_tensor(ent["key"])
            o, k = ent["offset"], ent["numel"]
            if same_layout:
                bufs[ent["key"]][o:o + k].copy_(chunk)
                continue
            for n, (so, sk, _) in meta["index"].items():          # re-map tensor by tensor
                lo, hi = max(so, o), min(so + sk, o + k)
                if lo < hi and n in f.index:
                    do = f.index[n][0]
                    bufs[ent["key"]][do + (lo - so): do + (hi - so)].copy_(chunk[lo - o: hi - o])
        f.params.copy_(f.master)
    f.bucket_steps = {str(k): int(v) for k, v in meta.get("bucket_steps", {}).items()}
    f.step_count = int(meta["step"])
    # a checkpoint written before per-bucket counts existed: every bucket has seen `step` updates
    f.bucket_step_floor = int(meta.get("bucket_step_floor", 0)) if "bucket_steps" in meta else f.step_count
