"""Host-side integration glue (vila_amd/integration.py): config mapping from the HF sub-configs a VILA checkpoint carries."""
import dataclasses

import pytest

from vila_amd import configs
from vila_amd.integration import vila_config_from_hf


def test_config_from_hf_objects_matches_nvila_8b():
    transformers = pytest.importorskip("transformers")
    ref = configs.nvila_8b()
    l, v = ref.llm, ref.vision
    hl = transformers.Qwen2Config(vocab_size=l.vocab_size, hidden_size=l.hidden_size, intermediate_size=l.intermediate_size,
                                  num_hidden_layers=l.num_hidden_layers, num_attention_heads=l.num_attention_heads,
                                  num_key_value_heads=l.num_key_value_heads, rms_norm_eps=l.rms_norm_eps, rope_theta=l.rope_theta,
                                  tie_word_embeddings=l.tie_word_embeddings, eos_token_id=l.eos_token_id)
    hv = transformers.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                                         num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                                         layer_norm_eps=v.layer_norm_eps)
    got = vila_config_from_hf(hl, hv, {"mm_projector_type": "mlp_downsample", "mm_vision_select_layer": -2})
    assert dataclasses.asdict(got.llm) == dataclasses.asdict(l)
    assert dataclasses.asdict(got.vision) == dataclasses.asdict(v)
    assert got.mm_projector_type == ref.mm_projector_type and got.tokens_per_tile == 256 and not got.dynamic_s2


def test_config_from_plain_dicts_and_dynamic_s2():
    ref = configs.nvila_8b_s2()
    l, v = dataclasses.asdict(ref.llm), dataclasses.asdict(ref.vision)
    got = vila_config_from_hf(l, v, {"mm_projector_type": "mlp_downsample", "dynamic_s2": True, "s2_scales": "448,896,1344",
                                     "s2_resize_output_to_scale_idx": -1})
    assert got.dynamic_s2 and got.s2_scales == (448, 896, 1344) and got.mm_hidden_size == 3456
    assert dataclasses.asdict(got.llm) == l
