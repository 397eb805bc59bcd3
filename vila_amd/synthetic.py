"""Seeded synthetic weights and inputs at the reference's state_dict names and shapes.

There are no checkpoints on disk and no network, so every parity and perf run uses
weights drawn here.  Key names follow SURVEY.md Appendix C so the three-folder checkpoint split of
`llava/model/llava_arch.py:158-204` (prefixes `llm.`, `vision_tower.vision_tower.`, `mm_projector.`)
keeps working.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import torch

from .configs import VilaConfig

Spec = Tuple[str, Tuple[int, ...], str]  # (name, shape, kind)


def vision_specs(cfg: VilaConfig) -> List[Spec]:
    v = cfg.vision
    p = "vision_tower.vision_tower.vision_model."
    s: List[Spec] = [
        (p + "embeddings.patch_embedding.weight", (v.hidden_size, v.num_channels, v.patch_size, v.patch_size), "w"),
        (p + "embeddings.patch_embedding.bias", (v.hidden_size,), "b"),
        (p + "embeddings.position_embedding.weight", (v.num_patches, v.hidden_size), "w"),
    ]
    for i in range(v.num_hidden_layers):
        l = f"{p}encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            s += [(l + ln + ".weight", (v.hidden_size,), "g"), (l + ln + ".bias", (v.hidden_size,), "b")]
        for pr in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s += [(l + f"self_attn.{pr}.weight", (v.hidden_size, v.hidden_size), "w"),
                  (l + f"self_attn.{pr}.bias", (v.hidden_size,), "b")]
        s += [(l + "mlp.fc1.weight", (v.intermediate_size, v.hidden_size), "w"),
              (l + "mlp.fc1.bias", (v.intermediate_size,), "b"),
              (l + "mlp.fc2.weight", (v.hidden_size, v.intermediate_size), "w"),
              (l + "mlp.fc2.bias", (v.hidden_size,), "b")]
    s += [(p + "post_layernorm.weight", (v.hidden_size,), "g"), (p + "post_layernorm.bias", (v.hidden_size,), "b")]
    return s


def projector_specs(cfg: VilaConfig) -> List[Spec]:
    """Layer indices follow the nn.Sequential positions of base_projector.py:145-174."""
    c, h = cfg.mm_hidden_size, cfg.llm.hidden_size
    p = "mm_projector.layers."
    t = cfg.mm_projector_type
    if t in ("mlp_downsample", "mlp_downsample_2x2_fix"):
        return [(p + "1.weight", (4 * c,), "g"), (p + "1.bias", (4 * c,), "b"),
                (p + "2.weight", (h, 4 * c), "w"), (p + "2.bias", (h,), "b"),
                (p + "4.weight", (h, h), "w"), (p + "4.bias", (h,), "b")]
    if t == "mlp_downsample_3x3_fix":
        return [(p + "1.weight", (9 * c,), "g"), (p + "1.bias", (9 * c,), "b"),
                (p + "2.weight", (3 * c, 9 * c), "w"), (p + "2.bias", (3 * c,), "b"),
                (p + "4.weight", (3 * c,), "g"), (p + "4.bias", (3 * c,), "b"),
                (p + "5.weight", (h, 3 * c), "w"), (p + "5.bias", (h,), "b"),
                (p + "7.weight", (h, h), "w"), (p + "7.bias", (h,), "b")]
    raise ValueError(f"Unknown projector type: {t}")  # base_projector.py:215


def llm_specs(cfg: VilaConfig) -> List[Spec]:
    c = cfg.llm
    p = "llm."
    s: List[Spec] = [(p + "model.embed_tokens.weight", (c.vocab_size, c.hidden_size), "w")]
    for i in range(c.num_hidden_layers):
        l = f"{p}model.layers.{i}."
        s += [(l + "input_layernorm.weight", (c.hidden_size,), "g"),
              (l + "self_attn.q_proj.weight", (c.q_size, c.hidden_size), "w"),
              (l + "self_attn.q_proj.bias", (c.q_size,), "b"),
              (l + "self_attn.k_proj.weight", (c.kv_size, c.hidden_size), "w"),
              (l + "self_attn.k_proj.bias", (c.kv_size,), "b"),
              (l + "self_attn.v_proj.weight", (c.kv_size, c.hidden_size), "w"),
              (l + "self_attn.v_proj.bias", (c.kv_size,), "b"),
              (l + "self_attn.o_proj.weight", (c.hidden_size, c.q_size), "w"),
              (l + "post_attention_layernorm.weight", (c.hidden_size,), "g"),
              (l + "mlp.gate_proj.weight", (c.intermediate_size, c.hidden_size), "w"),
              (l + "mlp.up_proj.weight", (c.intermediate_size, c.hidden_size), "w"),
              (l + "mlp.down_proj.weight", (c.hidden_size, c.intermediate_size), "w")]
    s += [(p + "model.norm.weight", (c.hidden_size,), "g")]
    if not c.tie_word_embeddings:
        s += [(p + "lm_head.weight", (c.vocab_size, c.hidden_size), "h")]
    return s


def all_specs(cfg: VilaConfig) -> List[Spec]:
    return vision_specs(cfg) + projector_specs(cfg) + llm_specs(cfg)


def _draw(name: str, shape, kind: str, cfg: VilaConfig, seed: int, device) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    x = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
    if kind == "w":
        a = float(getattr(cfg, "lm_head_tail", 0.0))
        if a > 0 and name == "llm.model.embed_tokens.weight" and cfg.llm.tie_word_embeddings:
            # a tied head IS the embedding table: the heavy-tailed row norms that make the full-depth fixtures' argmax decisive go here
            return x * cfg.init_std * lm_head_row_scale(name, shape[0], cfg).to(device)[:, None]
        return x * cfg.init_std
    if kind == "h":
        x = x * cfg.lm_head_std
        a = float(getattr(cfg, "lm_head_tail", 0.0))
        if a > 0:
            x = x * lm_head_row_scale(name, shape[0], cfg).to(device)[:, None]
        return x
    if kind == "b":
        return x * 0.02
    if kind == "g":  # norm gains: 1 + noise so a dropped gain multiply is visible to parity tests
        return 1.0 + 0.1 * x
    raise ValueError(kind)


def lm_head_row_scale(name: str, rows: int, cfg: VilaConfig) -> torch.Tensor:
    """Pareto(a) row norms of the synthetic lm_head (configs.VilaConfig.lm_head_tail), always drawn with the CPU generator in fp64 and
    normalised so that the largest is lm_head_tail_max; keyed by (name, lm_head_tail_seed) only, so that the rows keep their direction
    when the seed of the scales changes (oracle/make_golden_full.py searches over it: logits scale row by row)."""
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32((name + "/tail").encode()) ^ (int(cfg.lm_head_tail_seed) * 0x9E3779B1)) & 0x7FFFFFFF)
    u = torch.rand(rows, generator=g, dtype=torch.float64).clamp_min(1e-12)
    s = u.pow(-1.0 / float(cfg.lm_head_tail))
    s = (s / s.max() * float(cfg.lm_head_tail_max)).float()
    unit = [int(r) for r in getattr(cfg, "lm_head_tail_unit_rows", ()) if 0 <= int(r) < rows]
    if unit:
        s[torch.tensor(unit, dtype=torch.int64)] = 1.0
    return s


def make_weights(cfg: VilaConfig, seed: int = 0, device="cpu", dtype=torch.float32,
                 specs: List[Spec] | None = None) -> Dict[str, torch.Tensor]:
    """Each tensor has its own generator keyed by (name, seed): independent of order, dtype and subset."""
    out: Dict[str, torch.Tensor] = {}
    for name, shape, kind in (specs if specs is not None else all_specs(cfg)):
        out[name] = _draw(name, shape, kind, cfg, seed, device).to(dtype)
    return out


def make_pixels(cfg: VilaConfig, n_tiles: int, seed: int = 0, device="cpu", dtype=torch.float32) -> torch.Tensor:
    """U(-1,1) pixels = the range SiglipImageProcessor emits ((x/255-0.5)/0.5, llava/mm_utils.py:442-541)."""
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    v = cfg.vision
    x = torch.rand((n_tiles, v.num_channels, v.image_size, v.image_size), generator=g) * 2 - 1
    return x.to(device=device, dtype=dtype)


def make_prompt(cfg: VilaConfig, n_text: int, n_images: int = 1, seed: int = 0) -> torch.Tensor:
    """`[<image>]*n_images + n_text` random text ids that avoid the media / eos ids."""
    g = torch.Generator(device="cpu").manual_seed(2000 + seed)
    hi = min(cfg.llm.vocab_size, cfg.image_token_id, cfg.llm.eos_token_id) - 1
    ids = torch.randint(0, hi, (n_text,), generator=g, dtype=torch.int64)
    # `_media_plan` registers the video token too, so a stray video id without a video would raise.  The committed fixtures pin the draw above
    # (their configs keep the video id at or above `hi`), so a config whose video id lies INSIDE the range has those draws moved off it instead
    # of the range changed
    vid = getattr(cfg, "video_token_id", None)
    if vid is not None and 0 <= vid < hi:
        ids = torch.where(ids == vid, torch.full_like(ids, vid - 1 if vid > 0 else 1), ids)
    img = torch.full((n_images,), cfg.image_token_id, dtype=torch.int64)
    return torch.cat([img, ids])


def param_count(cfg: VilaConfig) -> int:
    n = 0
    for _, shape, _ in all_specs(cfg):
        k = 1
        for d in shape:
            k *= d
        n += k
    return n
