"""Reference-side glue: install the HIP modules at the three builder seams of a live VILA model.

`LlavaMetaModel.init_vlm` (llava/model/llava_arch.py:73-75) builds `self.llm`, `self.vision_tower`, `self.mm_projector` through
`build_llm_and_tokenizer` / `build_vision_tower` / `build_mm_projector`.  `swap_in_hip_modules(vlm)` replaces the three modules in
place: the HIP modules keep the reference's parameter names, so the weights move with one
`load_state_dict` per module and everything that calls `vlm.llm(...)`, `vlm.get_vision_tower()(images)`,
`vlm.get_mm_projector()(feats)` or `vlm.llm.generate(inputs_embeds=...)` keeps working unchanged.
"""
from __future__ import annotations

from typing import Optional

import torch

from .configs import LlmConfig, VilaConfig, VisionConfig


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def projector_type_of(top_cfg=None, projector=None) -> str:
    """mm_projector_type the way the reference stores it: on the live projector's own config (`vlm.mm_projector.config`,
    base_projector.py:126-137) or inside LlavaConfig.mm_projector_cfg; a top-level field exists only in configs this repo wrote."""
    t = _get(getattr(projector, "config", None), "mm_projector_type") if projector is not None else None
    if not t and top_cfg is not None:
        t = _get(_get(top_cfg, "mm_projector_cfg"), "mm_projector_type") if _get(top_cfg, "mm_projector_cfg") is not None else None
        t = t or _get(top_cfg, "mm_projector_type")
    return t or "mlp_downsample"


def vila_config_from_hf(llm_cfg, vision_cfg, top_cfg=None, projector=None) -> VilaConfig:
    """VilaConfig from the HF sub-configs a VILA checkpoint carries (Qwen2Config, SiglipVisionConfig, LlavaConfig)."""
    rope_theta = _get(llm_cfg, "rope_theta")
    if rope_theta is None:                                  # newer transformers keep it inside rope_parameters
        rope_theta = (_get(llm_cfg, "rope_parameters") or {}).get("rope_theta", 1e6)
    n_heads = _get(llm_cfg, "num_attention_heads")
    llm = LlmConfig(hidden_size=_get(llm_cfg, "hidden_size"), intermediate_size=_get(llm_cfg, "intermediate_size"),
                    num_hidden_layers=_get(llm_cfg, "num_hidden_layers"), num_attention_heads=n_heads,
                    num_key_value_heads=_get(llm_cfg, "num_key_value_heads", n_heads),
                    head_dim=_get(llm_cfg, "head_dim") or _get(llm_cfg, "hidden_size") // n_heads,
                    vocab_size=_get(llm_cfg, "vocab_size"), rms_norm_eps=_get(llm_cfg, "rms_norm_eps", 1e-6), rope_theta=float(rope_theta),
                    tie_word_embeddings=bool(_get(llm_cfg, "tie_word_embeddings", False)),
                    eos_token_id=_get(llm_cfg, "eos_token_id", 151645) if isinstance(_get(llm_cfg, "eos_token_id", 151645), int) else 151645)
    vis = VisionConfig(hidden_size=_get(vision_cfg, "hidden_size"), intermediate_size=_get(vision_cfg, "intermediate_size"),
                       num_hidden_layers=_get(vision_cfg, "num_hidden_layers"), num_attention_heads=_get(vision_cfg, "num_attention_heads"),
                       image_size=_get(vision_cfg, "image_size"), patch_size=_get(vision_cfg, "patch_size"),
                       num_channels=_get(vision_cfg, "num_channels", 3), layer_norm_eps=_get(vision_cfg, "layer_norm_eps", 1e-6),
                       select_layer=_get(top_cfg, "mm_vision_select_layer", -2) if top_cfg is not None else -2)
    scales = str(_get(top_cfg, "s2_scales", "448,896,1344")) if top_cfg is not None else "448,896,1344"
    kw = {}
    if top_cfg is not None:
        for src, dst in (("image_token_id", "image_token_id"), ("newline_token_id", "newline_token_id"), ("image_aspect_ratio", "image_aspect_ratio"),
                         ("min_tiles", "min_tiles"), ("max_tiles", "max_tiles"), ("video_max_tiles", "video_max_tiles"), ("chat_template", "chat_template")):
            if _get(top_cfg, src) is not None:
                kw[dst] = _get(top_cfg, src)
    return VilaConfig(vision=vis, llm=llm, mm_projector_type=projector_type_of(top_cfg, projector), dynamic_s2=bool(_get(top_cfg, "dynamic_s2", False)) if top_cfg is not None else False,
                      s2_scales=tuple(int(s) for s in scales.split(",")),
                      s2_resize_output_to_scale_idx=_get(top_cfg, "s2_resize_output_to_scale_idx", -1) if top_cfg is not None else -1,
                      name="from-hf", **kw)


def swap_in_hip_modules(vlm, cfg: Optional[VilaConfig] = None, device=None, strict: bool = True):
    """Replace vlm.llm / vlm.vision_tower / vlm.mm_projector by their HIP counterparts, weights included.  Returns the VilaConfig."""
    from .modules import HipMultimodalProjector, HipQwen2ForCausalLM, HipSiglipVisionTower
    if cfg is None:
        cfg = vila_config_from_hf(vlm.llm.config, vlm.vision_tower.config, getattr(vlm, "config", None), getattr(vlm, "mm_projector", None))
    for name, cls in (("llm", HipQwen2ForCausalLM), ("vision_tower", HipSiglipVisionTower), ("mm_projector", HipMultimodalProjector)):
        old = getattr(vlm, name)
        dev = device or next(old.parameters()).device
        new = cls(cfg, device=dev, dtype=torch.bfloat16)
        sd = {k: v for k, v in old.state_dict().items() if ".vision_model.head." not in k}   # pooling head: unused by VILA
        if name == "llm" and cfg.llm.tie_word_embeddings:
            sd.pop("lm_head.weight", None)                  # tied: the HIP module reads embed_tokens for the head
        missing, unexpected = new.load_state_dict(sd, strict=False)
        if cfg.llm.tie_word_embeddings:
            missing = [m for m in missing if m != "lm_head.weight"]
        if strict and (missing or unexpected):
            raise KeyError(f"{name}: state_dict mismatch, missing={missing[:4]} unexpected={unexpected[:4]}")
        if hasattr(new, "refuse"):
            new.refuse()                                    # re-establish the fused q/k/v storage after load_state_dict
        setattr(vlm, name, new)
    return cfg
