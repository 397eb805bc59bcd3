"""Architecture constants for the NVILA hot path.

The reference never spells these numbers out in-tree: it names the checkpoints
(`scripts/NVILA/stage1_9tile.sh:15,18` -> Qwen2.5-7B + paligemma-siglip-so400m-patch14-448,
`scripts/NVILA-Lite/align.sh:7,22`) and lets HF configs supply them.  SURVEY.md §3.3/§3.4/§8
pins the values used here.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict

IGNORE_INDEX = -100  # llava/constants.py:26


@dataclass
class VisionConfig:
    """SigLIP vision tower (llava/model/multimodal_encoder/siglip/modeling_siglip.py)."""

    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    image_size: int = 448
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-6
    # vision_encoder.py:44-52: hidden_states[select_layer]; -2 => run (L-1) layers
    select_layer: int = -2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def num_used_layers(self) -> int:
        # hidden_states has L+1 entries (embeddings + L layers); index select_layer
        idx = self.select_layer if self.select_layer >= 0 else self.num_hidden_layers + 1 + self.select_layer
        return idx


@dataclass
class LlmConfig:
    """Qwen2 causal LM (HF transformers qwen2, pinned 4.46.0 at pyproject.toml:17)."""

    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    head_dim: int = 128
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    tie_word_embeddings: bool = False
    eos_token_id: int = 151645  # <|im_end|>

    @property
    def q_size(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_key_value_heads * self.head_dim


@dataclass
class VilaConfig:
    vision: VisionConfig = field(default_factory=VisionConfig)
    llm: LlmConfig = field(default_factory=LlmConfig)
    # base_projector.py:145-174
    mm_projector_type: str = "mlp_downsample"
    image_token_id: int = 151649     # "<image>" added token (llava/constants.py:39-48)
    video_token_id: int = 151650     # "<vila/video>" (llava/constants.py:46)
    newline_token_id: int = 198      # tokenizer("\n").input_ids for Qwen2 (encoders/image/basic.py:22-27)
    init_std: float = 0.02
    lm_head_std: float = 0.05
    # > 0: the rows of the synthetic lm_head get Pareto(a = lm_head_tail) norms, the largest scaled to lm_head_tail_max x lm_head_std
    # (a peaked next-token distribution like a trained model's: i.i.d. Gaussian rows give top-1 / top-2 margins of ~0.2 sigma over 152 k
    # candidates, below 4x the bf16 logit error, so hardly any greedy step of a parity test would be decisive)
    lm_head_tail: float = 0.0
    lm_head_tail_seed: int = 0
    lm_head_tail_max: float = 10.0
    # rows of the tailed table whose scale is pinned to 1 (a TIED head is the embedding table: with the prompt's and the teacher-forced tokens'
    # rows pinned, the hidden states do not depend on the tail, so oracle/make_golden_lite3b.py can search tails with one decoder pass)
    lm_head_tail_unit_rows: tuple = ()
    name: str = "nvila-8b"
    # dynamic_s2 multi-scale recipe (scripts/NVILA/stage1_9tile.sh:19-22); off = the README benchmark setting (README.md:87)
    dynamic_s2: bool = False
    s2_scales: tuple = (448, 896, 1344)
    s2_resize_output_to_scale_idx: int = -1
    max_tiles: int = 12              # configuration_llava.py:52 — the most tiles the last dynamic_s2 scale / the `dynamic` tiler may use (mm_utils.py:299,341)
    min_tiles: int = 1               # configuration_llava.py:51 — `dynamic_preprocess(min_num=...)` (mm_utils.py:477)
    video_max_tiles: int = 1         # configuration_llava.py:53
    # How `process_image` (mm_utils.py:442-523) turns a picture into tower inputs: "resize" (NVILA stage 4), "pad" (expand to a square on the
    # processor's mean colour), "dynamic" (InternVL-style tiles + thumbnail: every NVILA-Lite script), "dynamic_s2" (every NVILA 9-tile script).
    # "" = not stated: "dynamic_s2" when the dynamic_s2 flag is on, else the processor's default (SigLIP: resize).
    image_aspect_ratio: str = ""
    chat_template: str = ""          # `--chat_template qwen2` (language_model/builder.py:194-200): "" = the tokenizer's own template

    @property
    def mm_hidden_size(self) -> int:
        """VisionTowerDynamicS2.hidden_size = C * len(scales) (vision_encoder.py:274-276)."""
        return self.vision.hidden_size * (len(self.s2_scales) if self.dynamic_s2 else 1)

    @property
    def aspect_mode(self) -> str:
        """The `image_aspect_ratio` the pre-processing follows (see the field)."""
        return self.image_aspect_ratio or ("dynamic_s2" if self.dynamic_s2 else "")

    @property
    def downsample(self) -> int:
        return {"mlp_downsample": 2, "mlp_downsample_2x2_fix": 2, "mlp_downsample_3x3_fix": 3}[self.mm_projector_type]

    @property
    def tokens_per_tile(self) -> int:
        g = self.vision.grid
        d = self.downsample
        gd = (g + d - 1) // d
        return gd * gd

    def to_dict(self):
        return asdict(self)


def nvila_8b() -> VilaConfig:
    """BASELINE.json configs[1]: SigLIP-so400m/14-448 + mlp_downsample + Qwen2.5-7B."""
    return VilaConfig()


def nvila_lite_3b() -> VilaConfig:
    """BASELINE.json configs[0]: 3x3 projector, Qwen2.5-3B-shaped LLM (tied head)."""
    return VilaConfig(
        llm=LlmConfig(hidden_size=2048, intermediate_size=11008, num_hidden_layers=36,
                      num_attention_heads=16, num_key_value_heads=2, head_dim=128,
                      vocab_size=151944, tie_word_embeddings=True),
        mm_projector_type="mlp_downsample_3x3_fix",
        name="nvila-lite-3b",
    )


def tiny(proj: str = "mlp_downsample", layers_v: int = 3, layers_l: int = 2, tied: bool = False,
         image: int = 56) -> VilaConfig:
    """Small config with the REAL head dims (72 / 128) used by the golden fixtures and unit tests."""
    return VilaConfig(
        vision=VisionConfig(hidden_size=144, intermediate_size=272, num_hidden_layers=layers_v,
                            num_attention_heads=2, image_size=image, patch_size=14),
        llm=LlmConfig(hidden_size=512, intermediate_size=1088, num_hidden_layers=layers_l,
                      num_attention_heads=4, num_key_value_heads=2, head_dim=128, vocab_size=1000,
                      tie_word_embeddings=tied, eos_token_id=999),
        mm_projector_type=proj,
        image_token_id=998,
        video_token_id=997,
        newline_token_id=11,
        init_std=0.05,
        lm_head_std=0.08,
        name=f"tiny-{proj}",
    )


def tiny_s2(layers_v: int = 3, layers_l: int = 2) -> VilaConfig:
    """tiny config with the dynamic_s2 recipe: scales (56, 112, 168) = 1x, 2x, 3x of the 56-px tile."""
    cfg = tiny("mlp_downsample", layers_v, layers_l)
    cfg.dynamic_s2 = True
    cfg.s2_scales = (56, 112, 168)
    cfg.name = "tiny-dynamic-s2"
    return cfg


def nvila_8b_s2() -> VilaConfig:
    """The full NVILA-8B recipe: dynamic_s2, mm_hidden 3456, 14 tiles -> 2304 tokens for a square image."""
    cfg = nvila_8b()
    cfg.dynamic_s2 = True
    cfg.name = "nvila-8b-dynamic-s2"
    return cfg


def reduced_8b(layers_v: int = 2, layers_l: int = 2, vocab: int = 152064) -> VilaConfig:
    """NVILA-8B widths with few layers: full-size GEMM/attention shapes at a CPU-oracle-friendly cost."""
    cfg = nvila_8b()
    cfg.vision.num_hidden_layers = layers_v
    cfg.llm.num_hidden_layers = layers_l
    cfg.llm.vocab_size = vocab
    cfg.name = f"nvila-8b-L{layers_v}v{layers_l}"
    return cfg
