"""REFERENCE-EXECUTED fixture for rows a6-a9 of SURVEY §8: `_embed` (splice) + `__truncate_sequence` + `__batchify_sequence` +
`repack_multimodal_data` — TEST INFRASTRUCTURE.

`llava.model.llava_arch` cannot be imported here (it pulls in deepspeed / hydra / the whole package), so the METHODS are taken from the file with
`ast` and exec'd UNCHANGED inside a class of the same name (`LlavaMetaForCausalLM`, so the double-underscore names mangle as in the reference):
    llava/model/llava_arch.py:   LlavaMetaForCausalLM._embed (412-490), .__embed_media_tokens (492-517), .__truncate_sequence (519-526),
                                 .__batchify_sequence (528-555), .repack_multimodal_data (557-800); LlavaMetaModel.encode_images (366-394)
    llava/model/utils/packing.py: _get_unpad_data, set_seqlens_in_batch (12-25)   — the varlen description of the packed row
    llava/model/encoders/base.py: BaseEncoder;  encoders/image/basic.py: BasicImageEncoder;  encoders/video/basic.py: BasicVideoEncoder;
    encoders/video/tsp.py: pool, TSPVideoEncoder                                                          (whole classes, unchanged)
The shim supplies only what the methods reach for: `self.llm.model.embed_tokens` (an nn.Embedding holding the synthetic table), `self.tokenizer`
(media_token_ids, padding_side, model_max_length, `tokenizer("\\n").input_ids`), `self.encoders`, `self.training`, `self.device`,
`get_vision_tower()` = the reference's `VisionTower` (its `__init__` / `feature_select` / `forward`, extracted the same way) around the reference SigLIP
(`modeling_siglip.py`) and
`get_mm_projector()` = the reference projector (both loaded by file path, as oracle/make_golden.py does); `get_pg_manager()` returns None (no
sequence parallelism) and `distributed.all_gather(x)` returns `[x]` (one rank).

Cases (tiny config, three ragged samples: image + text / text only / text + 3-frame video + text + image):
    eval_right, eval_left    inference, both padding sides
    train_trunc              training mode, model_max_length = 22: the cut goes through the middle of the video block
    train_tsp                training mode, TSPVideoEncoder [[3, 1, 1]] in place of the BasicVideoEncoder
each followed by repack_multimodal_data on its output.

    python oracle/make_golden_embed.py        # seconds; writes tests/golden/embed_splice_ref.npz; needs /root/reference
"""
from __future__ import annotations

import ast
import os
import sys
import textwrap
import types
import warnings
from collections import defaultdict, deque
from itertools import chain

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import make_golden as G                     # noqa: E402
from vila_amd import configs, synthetic                 # noqa: E402

REF = "/root/reference/llava/model"
IGNORE_INDEX = -100
SEED = 13
OUT = os.path.join(ROOT, "tests", "golden", "embed_splice_ref.npz")
ARCH_METHODS = ["_embed", "__embed_media_tokens", "__truncate_sequence", "__batchify_sequence", "repack_multimodal_data"]     # class LlavaMetaForCausalLM
MODEL_METHODS = ["encode_images"]                                                                                            # class LlavaMetaModel


def _class_source(path, cls):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            return ast.get_source_segment(src, node)
    raise KeyError((path, cls))


def _function_source(path, name):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return ast.get_source_segment(src, node)
    raise KeyError((path, name))


def _methods_source(path, cls, names):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            deco = lambda fn: "".join("@" + ast.get_source_segment(src, d) + "\n" for d in fn.decorator_list)      # (get_source_segment starts at `def`)
            got = {fn.name: deco(fn) + textwrap.dedent(ast.get_source_segment(src, fn, padded=True)) for fn in node.body if isinstance(fn, ast.FunctionDef)}
            return [textwrap.indent(got[n], "    ") for n in names]
    raise KeyError((path, cls))


def load_reference():
    """-> namespace holding the reference's LlavaMetaForCausalLM (the six methods), BaseEncoder, BasicImageEncoder, BasicVideoEncoder,
    TSPVideoEncoder, pool — their source text executed as it stands in /root/reference."""
    from functools import partial
    from typing import Any, Dict, List, Optional, Tuple
    from einops import rearrange
    distributed = types.SimpleNamespace(all_gather=lambda x: [x])
    ns = {"torch": torch, "nn": torch.nn, "Any": Any, "Dict": Dict, "List": List, "Optional": Optional, "Tuple": Tuple, "partial": partial,
          "deque": deque, "defaultdict": defaultdict, "chain": chain, "warnings": warnings, "rearrange": rearrange, "IGNORE_INDEX": IGNORE_INDEX,
          "get_pg_manager": lambda: None, "distributed": distributed, "ABC": object}
    exec(compile(_class_source(f"{REF}/encoders/base.py", "BaseEncoder"), "encoders/base.py", "exec"), ns)
    exec(compile(_class_source(f"{REF}/encoders/image/basic.py", "BasicImageEncoder"), "encoders/image/basic.py", "exec"), ns)
    exec(compile(_class_source(f"{REF}/encoders/video/basic.py", "BasicVideoEncoder"), "encoders/video/basic.py", "exec"), ns)
    exec(compile(_function_source(f"{REF}/encoders/video/tsp.py", "pool"), "encoders/video/tsp.py", "exec"), ns)
    exec(compile(_class_source(f"{REF}/encoders/video/tsp.py", "TSPVideoEncoder"), "encoders/video/tsp.py", "exec"), ns)
    # flash-attn varlen description of the packed row (llava/model/utils/packing.py:12-25), as llava_llama.py:125-130 sets it up
    ns["F"] = torch.nn.functional
    exec(compile(_function_source(f"{REF}/utils/packing.py", "_get_unpad_data"), "utils/packing.py", "exec"), ns)
    exec(compile(_function_source(f"{REF}/utils/packing.py", "set_seqlens_in_batch"), "utils/packing.py", "exec"), ns)
    # VisionTower.__init__ / feature_select / forward and its dtype / device properties (multimodal_encoder/vision_encoder.py:32-52, 133-184)
    vt = "\n\n".join(_methods_source(f"{REF}/multimodal_encoder/vision_encoder.py", "VisionTower", ["__init__", "feature_select", "forward", "dtype", "device"]))
    exec(compile("class VisionTower(torch.nn.Module):\n" + vt + "\n", "multimodal_encoder/vision_encoder.py", "exec"), ns)
    body = "\n\n".join(_methods_source(f"{REF}/llava_arch.py", "LlavaMetaModel", MODEL_METHODS) +
                       _methods_source(f"{REF}/llava_arch.py", "LlavaMetaForCausalLM", ARCH_METHODS))
    exec(compile("class LlavaMetaForCausalLM(torch.nn.Module):\n" + body + "\n", "llava_arch.py", "exec"), ns)
    return ns


class _Tokenizer:
    def __init__(self, cfg, side, max_len):
        self.media_token_ids = {"image": cfg.image_token_id, "video": cfg.video_token_id}
        self.padding_side, self.model_max_length = side, max_len
        self._nl = cfg.newline_token_id

    def __call__(self, text):
        assert text == "\n"
        return types.SimpleNamespace(input_ids=[self._nl])


def build_model(ns, cfg, w, side="right", max_len=4096, tsp=None):
    ms, bp = G.ref_siglip(), G.ref_projector()
    v = cfg.vision
    vc = ms.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                               num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                               num_channels=v.num_channels, layer_norm_eps=v.layer_norm_eps, hidden_act="gelu_pytorch_tanh")
    vc._attn_implementation = "eager"
    tower = ms.SiglipVisionModel(vc).train(False)
    tower.load_state_dict({k[len("vision_tower.vision_tower."):]: t for k, t in w.items() if k.startswith("vision_tower.")}, strict=False)
    proj = bp.MultimodalProjector(bp.MultimodalProjectorConfig(cfg.mm_projector_type),
                                  types.SimpleNamespace(mm_hidden_size=cfg.mm_hidden_size, hidden_size=cfg.llm.hidden_size)).train(False)
    proj.load_state_dict({k[len("mm_projector."):]: t for k, t in w.items() if k.startswith("mm_projector.")}, strict=True)
    m = ns["LlavaMetaForCausalLM"]()
    emb = torch.nn.Embedding.from_pretrained(w["llm.model.embed_tokens.weight"].clone(), freeze=True)
    m.llm = types.SimpleNamespace(model=types.SimpleNamespace(embed_tokens=emb))
    m.tokenizer = _Tokenizer(cfg, side, max_len)
    m.config = types.SimpleNamespace(dynamic_s2=False)
    # the reference's own VisionTower wrapper around the reference SigLIP (select_layer -2, select_feature cls_patch as every NVILA script passes)
    vt = ns["VisionTower"]("siglip", types.SimpleNamespace(mm_vision_select_layer=cfg.vision.select_layer, mm_vision_select_feature="cls_patch"))
    vt.vision_tower = tower
    m.get_vision_tower = lambda: vt
    m.get_mm_projector = lambda: proj
    object.__setattr__(m, "_dev", torch.device("cpu"))
    type(m).device = property(lambda self: self._dev)
    enc_v = ns["TSPVideoEncoder"](m, tsp) if tsp is not None else ns["BasicVideoEncoder"](m)
    m.encoders = {"image": ns["BasicImageEncoder"](m), "video": enc_v}
    return m


def case(cfg):
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, SEED).items()}
    px = synthetic.make_pixels(cfg, 5, SEED).to(torch.bfloat16).float()           # tiles 0, 1: images; 2-4: the video's frames
    g = torch.Generator().manual_seed(SEED)
    L = 12
    ids = torch.randint(0, 900, (3, L), generator=g)
    mask = torch.ones((3, L), dtype=torch.bool)
    ids[0, 0] = cfg.image_token_id
    mask[1, 7:] = False
    ids[1, 9] = cfg.image_token_id                       # a media id inside the padding: removed with the padding before the scan (:449-450)
    ids[2, 2] = cfg.video_token_id
    ids[2, 8] = cfg.image_token_id
    mask[2, 11:] = False
    labels = torch.randint(0, 900, (3, L), generator=g)
    labels[:, :3] = IGNORE_INDEX
    return w, px, ids, labels, mask


def main():
    torch.manual_seed(0)
    ns = load_reference()
    cfg = configs.tiny("mlp_downsample")
    w, px, ids, labels, mask = case(cfg)
    media = lambda: {"image": [px[0], px[1]], "video": [px[2:5]]}
    fx = {"seed": np.int64(SEED), "input_ids": ids.numpy(), "labels": labels.numpy(), "mask": mask.numpy()}
    cases = {"eval_right": dict(side="right", training=False), "eval_left": dict(side="left", training=False),
             "train_trunc": dict(side="right", training=True, max_len=22), "train_tsp": dict(side="right", training=True, tsp=[[3, 1, 1]])}
    with torch.no_grad():
        for name, c in cases.items():
            m = build_model(ns, cfg, w, side=c["side"], max_len=c.get("max_len", 4096), tsp=c.get("tsp"))
            m.train(c["training"])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")           # the reference warns when it truncates
                e, l, am = m._embed(ids.clone(), media(), {"image": {}, "video": {}}, labels.clone(), mask.clone())
            ns["set_seqlens_in_batch"](torch.sum(am, dim=1))                       # llava_llama.py:126-128
            pe, pm, pp, pl = m.repack_multimodal_data(e, am, None, l)
            idx, cu, mx = ns["_get_unpad_data"](pm)                              # what HF's flash-attention path asks for the packed row
            fx[f"{name}_unpad_indices"], fx[f"{name}_cu_seqlens"], fx[f"{name}_max_seqlen"] = idx.numpy(), cu.numpy(), np.int64(mx)
            for k, t in (("embeds", e), ("labels", l), ("mask", am), ("packed_embeds", pe), ("packed_mask", pm), ("packed_pos", pp), ("packed_labels", pl)):
                fx[f"{name}_{k}"] = t.numpy()
            print(f"{name}: embeds {tuple(e.shape)}, seqlens {am.sum(1).tolist()}, packed {tuple(pe.shape)}")
    fx["train_trunc_max_len"] = np.int64(22)
    fx["train_tsp_pools"] = np.array([[3, 1, 1]], dtype=np.int32)
    np.savez_compressed(OUT, **fx)
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes)")


if __name__ == "__main__":
    main()
