"""Cross-boundary checkpoint test: what `vila_amd.checkpoint.save_pretrained` writes must load into the REFERENCE's own classes.

`LlavaMetaModel.load_pretrained` / `init_vlm` (llava/model/llava_arch.py:73-75,158-204) rebuild the three sub-models from the three
folders with
    llm/           -> HF `Qwen2ForCausalLM.from_pretrained(<dir>/llm)`                         (language_model/builder.py:178-180)
    vision_tower/  -> the reference's `SiglipVisionModel` (multimodal_encoder/siglip/modeling_siglip.py, loaded by file path here
                      because `import llava.model` needs deepspeed) with the folder's safetensors
    mm_projector/  -> the reference's `MultimodalProjector` (multimodal_projector/base_projector.py) built from the folder's
                      config.json (`mm_projector_type` lives THERE, base_projector.py:126-131)
and every tensor must come back bit-identical.  The other direction (reference state_dicts -> HIP modules) is
tests/test_gpu_integration.py.  Needs /root/reference and transformers: skipped where they are absent (the GPU box)."""
import json
import os

import pytest
import torch

from vila_amd import checkpoint, configs
from vila_amd.vlm import HipLlavaLlamaModel

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only mounted in the build container")


def _rand_model(cfg, seed):
    m = HipLlavaLlamaModel(cfg, device="cpu")
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g).to(p.dtype))
    return m


@pytest.mark.parametrize("proj,tied", [("mlp_downsample", False), ("mlp_downsample_3x3_fix", True)])
def test_saved_checkpoint_loads_into_the_reference_classes(tmp_path, proj, tied):
    pytest.importorskip("transformers")
    from transformers import Qwen2ForCausalLM
    from oracle.make_golden import ref_projector, ref_siglip
    cfg = configs.tiny(proj, tied=tied)
    m = _rand_model(cfg, 5)
    d = str(tmp_path / "ckpt")
    checkpoint.save_pretrained(m, d, max_shard_bytes=1 << 20)
    ours = m.state_dict()

    # ---- llm/ through HF from_pretrained (config.json + sharded safetensors + index) ----
    llm, info = Qwen2ForCausalLM.from_pretrained(os.path.join(d, "llm"), torch_dtype=torch.bfloat16, output_loading_info=True)
    assert not info["missing_keys"] and not info["unexpected_keys"] and not info["mismatched_keys"], info
    hf_sd = llm.state_dict()
    c = cfg.llm
    assert (llm.config.hidden_size, llm.config.num_hidden_layers, llm.config.num_key_value_heads, llm.config.vocab_size) == \
           (c.hidden_size, c.num_hidden_layers, c.num_key_value_heads, c.vocab_size)
    assert bool(llm.config.tie_word_embeddings) == tied
    n = 0
    for k, t in ours.items():
        if k.startswith("llm."):
            assert torch.equal(hf_sd[k[4:]], t), k
            n += 1
    assert n == len([k for k in hf_sd]) or tied                      # tied: HF lists lm_head.weight as an alias of embed_tokens
    if tied:
        assert torch.equal(hf_sd["lm_head.weight"], ours["llm.model.embed_tokens.weight"])

    # ---- vision_tower/ into the reference's SiglipVisionModel ----
    ms = ref_siglip()
    vcfg = json.load(open(os.path.join(d, "vision_tower", "config.json")))
    sig_cfg = ms.SiglipVisionConfig(**{k: v for k, v in vcfg.items() if k != "model_type"})
    tower = ms.SiglipVisionModel(sig_cfg)
    sd = dict(checkpoint._folder_tensors(os.path.join(d, "vision_tower")))
    missing, unexpected = tower.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("vision_model.head.") for k in missing), (missing, unexpected)     # pooling head: unused by VILA
    ref_sd = tower.state_dict()
    for k, t in ours.items():
        if k.startswith("vision_tower.vision_tower."):
            assert torch.equal(ref_sd[k[len("vision_tower.vision_tower."):]].to(torch.bfloat16), t), k

    # ---- mm_projector/ into the reference's MultimodalProjector; the type comes from the folder's own config ----
    bp = ref_projector()
    pcfg = json.load(open(os.path.join(d, "mm_projector", "config.json")))
    assert pcfg["mm_projector_type"] == proj
    top = json.load(open(os.path.join(d, "config.json")))
    from types import SimpleNamespace
    projector = bp.MultimodalProjector(bp.MultimodalProjectorConfig(pcfg["mm_projector_type"]),
                                       SimpleNamespace(mm_hidden_size=top["mm_hidden_size"], hidden_size=top["hidden_size"]))
    projector.load_state_dict(dict(checkpoint._folder_tensors(os.path.join(d, "mm_projector"))), strict=True)
    ref_sd = projector.state_dict()
    for k, t in ours.items():
        if k.startswith("mm_projector."):
            assert torch.equal(ref_sd[k[len("mm_projector."):]].to(torch.bfloat16), t), k
    # and back: the folders load into a fresh HIP model bit for bit (round trip through the reference's layout)
    m2 = checkpoint.load_pretrained(d, device="cpu")
    for k, t in ours.items():
        assert torch.equal(m2.state_dict()[k], t), k
