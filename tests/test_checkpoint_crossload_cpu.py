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

    # ... and through the reference class's own from_pretrained (multimodal_encoder/siglip_encoder.py loads the tower this way): nothing missing
    tower2, vinfo = ms.SiglipVisionModel.from_pretrained(os.path.join(d, "vision_tower"), torch_dtype=torch.bfloat16, output_loading_info=True)
    assert not vinfo["missing_keys"] and not vinfo["unexpected_keys"] and not vinfo["mismatched_keys"], vinfo
    sd2 = tower2.state_dict()
    for k, t in ours.items():
        if k.startswith("vision_tower.vision_tower."):
            assert torch.equal(sd2[k[len("vision_tower.vision_tower."):]], t), k

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


def _reference_save_pretrained():
    """`LlavaMetaModel.save_pretrained` / get_llm / get_vision_tower / get_mm_projector (llava/model/llava_arch.py:157-220), taken from the file
    with ast and executed unchanged inside a class of the same name (`import llava.model` itself needs deepspeed)."""
    import ast
    import os.path as osp
    import textwrap
    from collections import OrderedDict
    path = f"{REF}/llava/model/llava_arch.py"
    src = open(path).read()
    want = ["save_pretrained", "get_llm", "get_vision_tower", "get_mm_projector"]
    body = []
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == "LlavaMetaModel":
            got = {fn.name: ast.get_source_segment(src, fn) for fn in node.body if isinstance(fn, ast.FunctionDef)}
            body = [textwrap.indent(textwrap.dedent(got[n]), "    ") for n in want]
    ns = {"os": os, "osp": osp, "OrderedDict": OrderedDict, "torch": torch}
    exec(compile("class LlavaMetaModel(torch.nn.Module):\n" + "\n\n".join(body) + "\n", "llava_arch.py", "exec"), ns)
    return ns["LlavaMetaModel"]


@pytest.mark.parametrize("proj,tied", [("mlp_downsample", False), ("mlp_downsample_3x3_fix", True)])
def test_checkpoint_written_by_the_reference_loads_into_the_hip_model(tmp_path, proj, tied):
    """The other direction, with the writer being the REFERENCE'S OWN code: its `save_pretrained` (ast-extracted, executed unchanged) over HF
    `Qwen2ForCausalLM`, the reference `SiglipVisionModel` (+ an HF `SiglipImageProcessor`, as the tower wrapper holds), the reference
    `MultimodalProjector` and a `LlavaConfig` (llava/model/configuration_llava.py, loaded by file path) writes the three-folder layout;
    `vila_amd.checkpoint.load_pretrained` reads the config out of it (projector type from `mm_projector_cfg`, sizes from the embedded
    sub-configs) and every tensor arrives bit for bit."""
    pytest.importorskip("transformers")
    from types import SimpleNamespace
    from transformers import SiglipImageProcessor
    from oracle import make_golden as G
    from vila_amd import synthetic
    cfg = configs.tiny(proj, tied=tied)
    w = {k: v.to(torch.bfloat16) for k, v in synthetic.make_weights(cfg, 21).items()}
    ms, bp = G.ref_siglip(), G.ref_projector()
    lc = G.load_by_path("ref_configuration_llava", f"{REF}/llava/model/configuration_llava.py")
    v = cfg.vision
    vc = ms.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                               num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                               num_channels=v.num_channels, layer_norm_eps=v.layer_norm_eps, hidden_act="gelu_pytorch_tanh")
    tower = ms.SiglipVisionModel(vc).to(torch.bfloat16)
    tower.load_state_dict({k[len("vision_tower.vision_tower."):]: t for k, t in w.items() if k.startswith("vision_tower.")}, strict=False)
    projector = bp.MultimodalProjector(bp.MultimodalProjectorConfig(cfg.mm_projector_type),
                                       SimpleNamespace(mm_hidden_size=cfg.mm_hidden_size, hidden_size=cfg.llm.hidden_size)).to(torch.bfloat16)
    projector.load_state_dict({k[len("mm_projector."):]: t for k, t in w.items() if k.startswith("mm_projector.")}, strict=True)
    llm, _ = G.build_hf_llm(cfg, {k: t.float() for k, t in w.items()})
    llm = llm.to(torch.bfloat16)
    m = _reference_save_pretrained()()
    m.llm = llm
    m.mm_projector = projector
    wrapper = torch.nn.Module()                                    # VisionTower (multimodal_encoder/vision_encoder.py:32-52): .vision_tower, .config, .image_processor
    wrapper.vision_tower = tower
    wrapper.config = tower.config
    wrapper.image_processor = SiglipImageProcessor(size={"height": v.image_size, "width": v.image_size})
    m.vision_tower = wrapper
    m.config = lc.LlavaConfig(hidden_size=cfg.llm.hidden_size, mm_hidden_size=cfg.mm_hidden_size, mm_vision_select_layer=-2,
                              mm_vision_select_feature="cls_patch", mm_projector_type=cfg.mm_projector_type)
    d = str(tmp_path / "ref_ckpt")
    m.save_pretrained(d)                                             # the reference's own writer
    assert sorted(x for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))) == ["llm", "mm_projector", "vision_tower"]
    got_cfg = checkpoint.config_from_pretrained(d)
    assert got_cfg.mm_projector_type == proj
    assert (got_cfg.llm.hidden_size, got_cfg.llm.num_hidden_layers, got_cfg.llm.num_key_value_heads, got_cfg.llm.vocab_size, got_cfg.llm.tie_word_embeddings) == \
           (cfg.llm.hidden_size, cfg.llm.num_hidden_layers, cfg.llm.num_key_value_heads, cfg.llm.vocab_size, tied)
    assert (got_cfg.vision.hidden_size, got_cfg.vision.num_hidden_layers, got_cfg.vision.image_size, got_cfg.vision.patch_size) == \
           (v.hidden_size, v.num_hidden_layers, v.image_size, v.patch_size)
    assert got_cfg.llm.rope_theta == cfg.llm.rope_theta and got_cfg.llm.eos_token_id == cfg.llm.eos_token_id and got_cfg.llm.rms_norm_eps == cfg.llm.rms_norm_eps
    top = json.load(open(os.path.join(d, "config.json")))
    assert top["s2_scales"] is None and top["image_aspect_ratio"] is None                  # what the reference's LlavaConfig writes for fields it was not given
    assert got_cfg.vision.select_layer == -2 and got_cfg.dynamic_s2 is False and got_cfg.s2_scales == (448, 896, 1344)
    assert got_cfg.s2_resize_output_to_scale_idx == top["s2_resize_output_to_scale_idx"] == 0   # LlavaConfig's own default (the NVILA scripts pass -1)
    hip = checkpoint.load_pretrained(d, device="cpu")
    ours = hip.state_dict()
    n = 0
    for k, t in w.items():
        if k in ours:
            assert torch.equal(ours[k], t), k
            n += 1
    assert n >= 60 and all(k in ours for k in w if not (tied and k == "llm.lm_head.weight"))


def test_our_checkpoint_resolves_through_the_reference_config_loader(tmp_path):
    """How the reference finds the three sub-models of a checkpoint: `LlavaConfig.from_pretrained(dir)` (configuration_llava.py, HF
    PretrainedConfig) then `get_model_config(config)` (llava/model/utils/utils.py:25-55, ast-extracted, executed unchanged) -> [llm, vision_tower,
    mm_projector] paths.  On a checkpoint written by `vila_amd.checkpoint.save_pretrained` the config parses, carries the fields the reference's
    builders read, and the three paths exist and hold a config.json + weights each."""
    pytest.importorskip("transformers")
    import ast
    import os.path as osp
    from transformers import AutoConfig, PretrainedConfig
    from oracle import make_golden as G
    cfg = configs.tiny("mlp_downsample")
    m = _rand_model(cfg, 7)
    d = str(tmp_path / "ckpt")
    checkpoint.save_pretrained(m, d)
    lc = G.load_by_path("ref_configuration_llava", f"{REF}/llava/model/configuration_llava.py")
    config = lc.LlavaConfig.from_pretrained(d)
    config.resume_path = d                                               # llava/model/builder.py:110-113 does this right after loading the config
    assert config.hidden_size == cfg.llm.hidden_size and config.mm_hidden_size == cfg.mm_hidden_size
    assert config.mm_vision_select_layer == -2 and config.mm_vision_select_feature == "cls_patch"
    # builder.py:159-167 `prepare_config_for_eval` (extracted, executed unchanged): needs `vision_tower_cfg` and takes the dtype
    bsrc = open(f"{REF}/llava/model/builder.py").read()
    pfn = next(n for n in ast.parse(bsrc).body if isinstance(n, ast.FunctionDef) and n.name == "prepare_config_for_eval")
    bns = {"PretrainedConfig": PretrainedConfig}
    exec(compile(ast.get_source_segment(bsrc, pfn), "builder.py", "exec"), bns)
    bns["prepare_config_for_eval"](config, {"torch_dtype": torch.bfloat16})
    assert config.model_dtype == "torch.bfloat16" and isinstance(config.vision_tower_cfg, dict)
    src = open(f"{REF}/llava/model/utils/utils.py").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_model_config")
    ns = {"os": os, "osp": osp, "PretrainedConfig": PretrainedConfig, "repo_exists": lambda p: False, "HFValidationError": Exception,
          "snapshot_download": None}
    exec(compile(ast.get_source_segment(src, fn), "utils.py", "exec"), ns)
    paths = ns["get_model_config"](config)
    assert [os.path.basename(p) for p in paths] == ["llm", "vision_tower", "mm_projector"]
    for p in paths:
        assert os.path.isfile(os.path.join(p, "config.json")) and any(f.endswith(".safetensors") for f in os.listdir(p)), p
    llm_cfg = AutoConfig.from_pretrained(paths[0])                      # language_model/builder.py:66-70 does exactly this
    assert llm_cfg.model_type == "qwen2" and llm_cfg.hidden_size == cfg.llm.hidden_size
    assert json.load(open(os.path.join(paths[2], "config.json")))["mm_projector_type"] == "mlp_downsample"
