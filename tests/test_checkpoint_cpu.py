"""Checkpoint I/O in the reference's three-folder layout (llava_arch.py:158-204): CPU-only (tensor plumbing, no kernels)."""
import json
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import save_file

from vila_amd import checkpoint, configs
from vila_amd.vlm import HipLlavaLlamaModel


def _rand_model(cfg, seed):
    m = HipLlavaLlamaModel(cfg, device="cpu")
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g).to(p.dtype))
    return m


def test_save_layout_and_roundtrip(tmp_path):
    cfg = configs.tiny("mlp_downsample_3x3_fix", tied=True)
    m = _rand_model(cfg, 1)
    d = str(tmp_path / "ckpt")
    checkpoint.save_pretrained(m, d, max_shard_bytes=1 << 20)          # small shards: exercises the index.json path
    assert sorted(os.listdir(d)) == ["config.json", "llm", "mm_projector", "vision_tower"]
    assert os.path.exists(os.path.join(d, "llm", "model.safetensors.index.json"))
    pj_keys = {k for k, _ in checkpoint._folder_tensors(os.path.join(d, "mm_projector"))}
    assert pj_keys == {f"layers.{i}.{k}" for i in (1, 2, 4, 5, 7) for k in ("weight", "bias")}
    keys = {k for k, _ in checkpoint._folder_tensors(os.path.join(d, "vision_tower"))}
    assert "vision_model.embeddings.patch_embedding.weight" in keys and all(k.startswith("vision_model.") for k in keys)
    idx = json.load(open(os.path.join(d, "llm", "model.safetensors.index.json")))
    assert "model.layers.0.self_attn.q_proj.weight" in idx["weight_map"] and "lm_head.weight" not in idx["weight_map"]   # tied head
    top = json.load(open(os.path.join(d, "config.json")))
    assert top["architectures"] == ["LlavaLlamaModel"] and top["llm_cfg"]["model_type"] == "qwen2"
    m2 = checkpoint.load_pretrained(d, device="cpu")
    assert m2.cfg.mm_projector_type == cfg.mm_projector_type and m2.cfg.llm.tie_word_embeddings
    sd1, sd2 = m.state_dict(), m2.state_dict()
    assert set(sd1) == set(sd2)
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    # q/k/v stayed views of one fused buffer after loading
    a = getattr(m2.llm.model.layers, "0").self_attn
    assert a.k_proj.weight.data_ptr() == a.q_proj.weight.data_ptr() + a.q_proj.weight.numel() * 2


def test_reference_checkpoint_extras_are_ignored_and_missing_detected(tmp_path):
    cfg = configs.tiny("mlp_downsample")
    m = _rand_model(cfg, 2)
    d = str(tmp_path / "ckpt")
    checkpoint.save_pretrained(m, d)
    # a real SigLIP checkpoint also carries the pooling head: must be ignored, not fatal
    vt = os.path.join(d, "vision_tower", "model.safetensors")
    tensors = {}
    with safe_open(vt, framework="pt") as f:
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    tensors["vision_model.head.probe"] = torch.zeros(1, 1, 144)
    save_file(tensors, vt)
    m2 = HipLlavaLlamaModel(cfg, device="cpu")
    rep = checkpoint.load_weights_into(m2, d)
    assert rep["ignored"] == ["vision_tower/vision_model.head.probe"] and not rep["missing"]
    del tensors["vision_model.embeddings.patch_embedding.bias"]
    save_file(tensors, vt)
    try:
        checkpoint.load_weights_into(HipLlavaLlamaModel(cfg, device="cpu"), d)
        raise AssertionError("missing parameter not detected")
    except KeyError as e:
        assert "patch_embedding.bias" in str(e)


def test_projector_type_is_read_where_the_reference_stores_it(tmp_path):
    """The reference's LlavaConfig has no top-level `mm_projector_type` (base_projector.py:126-131: it lives in the projector's own
    config).  A reference-shaped config.json — sub-configs as dicts or as sub-folder paths — must resolve the 3x3 projector."""
    import json
    from vila_amd.checkpoint import config_from_pretrained, resolve_projector_type
    llm = {"hidden_size": 64, "intermediate_size": 128, "num_hidden_layers": 1, "num_attention_heads": 2, "num_key_value_heads": 1, "vocab_size": 100}
    vt = {"hidden_size": 32, "intermediate_size": 64, "num_hidden_layers": 2, "num_attention_heads": 2, "image_size": 28, "patch_size": 14}
    d = tmp_path / "as_dicts"
    d.mkdir()
    json.dump({"llm_cfg": llm, "vision_tower_cfg": vt, "mm_projector_cfg": {"model_type": "v2l_projector", "mm_projector_type": "mlp_downsample_3x3_fix"}},
              open(d / "config.json", "w"))
    assert config_from_pretrained(str(d)).mm_projector_type == "mlp_downsample_3x3_fix"
    p = tmp_path / "as_paths"
    for sub, cfg in (("llm", llm), ("vision_tower", vt), ("mm_projector", {"mm_projector_type": "mlp_downsample_2x2_fix"})):
        (p / sub).mkdir(parents=True)
        json.dump(cfg, open(p / sub / "config.json", "w"))
    json.dump({"llm_cfg": str(p / "llm"), "vision_tower_cfg": str(p / "vision_tower"), "mm_projector_cfg": str(p / "mm_projector")}, open(p / "config.json", "w"))
    assert config_from_pretrained(str(p)).mm_projector_type == "mlp_downsample_2x2_fix"
    assert resolve_projector_type({}) == "mlp_downsample"


def test_live_model_projector_type_comes_from_the_projector_config():
    from types import SimpleNamespace as NS
    from vila_amd.integration import projector_type_of
    proj = NS(config=NS(mm_projector_type="mlp_downsample_3x3_fix"))
    assert projector_type_of(NS(mm_projector_cfg={"mm_projector_type": "mlp_downsample"}), proj) == "mlp_downsample_3x3_fix"
    assert projector_type_of(NS(mm_projector_cfg={"mm_projector_type": "mlp_downsample_2x2_fix"}), None) == "mlp_downsample_2x2_fix"
    assert projector_type_of(None, None) == "mlp_downsample"


def test_optimizer_state_without_bucket_counts_resumes_bias_correction_at_the_saved_step(tmp_path):
    """a checkpoint written before per-bucket step counts existed has only `step`.  Loading it must not restart AdamW's bias
    correction at 1 on warm moments: every bucket resumes at the saved step (`bucket_step_floor`), and a later save / load keeps that."""
    import json
    from vila_amd import checkpoint, configs
    from vila_amd.train import SFTTrainer
    from vila_amd.vlm import HipLlavaLlamaModel
    torch.manual_seed(0)
    tr = SFTTrainer(HipLlavaLlamaModel(configs.tiny("mlp_downsample"), device="cpu"))
    tr.flat.step_count = 37
    tr.flat.bucket_steps = {"llm.model.norm.": 37}
    d = str(tmp_path / "ck")
    checkpoint.save_optimizer(tr, d)
    meta_path = os.path.join(d, "optimizer", "optimizer.json")
    meta = json.load(open(meta_path))
    meta.pop("bucket_steps"); meta.pop("bucket_step_floor")            # what a pre-round-3 writer left
    json.dump(meta, open(meta_path, "w"))
    tr2 = SFTTrainer(HipLlavaLlamaModel(configs.tiny("mlp_downsample"), device="cpu"))
    checkpoint.load_optimizer(tr2, d)
    assert tr2.flat.step_count == 37 and tr2.flat.bucket_steps == {} and tr2.flat.bucket_step_floor == 37
    seen = []
    from vila_amd import ops
    orig = ops.adamw_step
    ops.adamw_step = lambda *a, **k: seen.append(a[10])                 # the `step` argument of the bias correction
    try:
        tr2._adamw_bucket("mm_projector.", 1.0)
    finally:
        ops.adamw_step = orig
    assert seen == [38] and tr2.flat.bucket_steps["mm_projector."] == 38
    d2 = str(tmp_path / "ck2")
    checkpoint.save_optimizer(tr2, d2)
    tr3 = SFTTrainer(HipLlavaLlamaModel(configs.tiny("mlp_downsample"), device="cpu"))
    checkpoint.load_optimizer(tr3, d2)
    assert tr3.flat.bucket_step_floor == 37 and tr3.flat.bucket_steps == {"mm_projector.": 38}
    sd = tr3.flat.optimizer_state()
    tr4 = SFTTrainer(HipLlavaLlamaModel(configs.tiny("mlp_downsample"), device="cpu"))
    tr4.flat.load_optimizer_state(sd)
    assert tr4.flat.bucket_step_floor == 37 and tr4.flat.bucket_steps == {"mm_projector.": 38}


def test_tokenizer_is_saved_in_llm_and_drives_the_media_ids_on_load(tmp_path):
    """llava_arch.py:164-165 saves the tokenizer into llm/; language_model/builder.py:190-211 loads it from there, adds the media tokens and records
    their ids — which is where `<image>` / `<vila/video>` get their ids from (the reference's config.json does not carry them).  An offline-built
    fast tokenizer stands for Qwen2's: its ids differ from the config defaults, so the loaded model must follow the tokenizer."""
    pytest.importorskip("transformers")
    pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"[UNK]": 0, "\n": 1, "hello": 2, "world": 3, "what": 4, "is": 5, "this": 6}
    t = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    t.pre_tokenizer = pre_tokenizers.Split(" ", "removed")
    tok = PreTrainedTokenizerFast(tokenizer_object=t, unk_token="[UNK]")
    tok.add_tokens(["<image>", "<vila/video>"], special_tokens=True)
    ids = {"image": tok.convert_tokens_to_ids("<image>"), "video": tok.convert_tokens_to_ids("<vila/video>")}
    tok.media_token_ids = dict(ids)
    cfg = configs.tiny("mlp_downsample")
    assert ids["image"] != cfg.image_token_id                        # the test is only meaningful if the tokenizer disagrees with the defaults
    m = HipLlavaLlamaModel(cfg, device="cpu", tokenizer=tok)
    d = str(tmp_path / "ckpt")
    checkpoint.save_pretrained(m, d)
    assert os.path.exists(os.path.join(d, "llm", "tokenizer.json")) and os.path.exists(os.path.join(d, "llm", "config.json"))
    m2 = checkpoint.load_pretrained(d, device="cpu")
    assert hasattr(m2.tokenizer, "save_pretrained") and m2.tokenizer.padding_side == "right"
    assert m2.tokenizer.media_token_ids == ids and m2.tokenizer.media_tokens == {"image": "<image>", "video": "<vila/video>"}
    assert (m2.cfg.image_token_id, m2.cfg.video_token_id) == (ids["image"], ids["video"])
    nl = m2.tokenizer("\n").input_ids                               # (next to a qwen2 config.json AutoTokenizer re-tokenises this toy vocabulary its own way)
    assert m2.cfg.newline_token_id == (nl[0] if len(nl) == 1 else cfg.newline_token_id)
    got = m2.tokenizer("hello <image> world").input_ids
    assert ids["image"] in got and got.count(ids["image"]) == 1
    # a checkpoint without tokenizer files keeps the stand-in and the ids of its config.json
    d2 = str(tmp_path / "plain")
    checkpoint.save_pretrained(HipLlavaLlamaModel(cfg, device="cpu"), d2)
    m3 = checkpoint.load_pretrained(d2, device="cpu")
    assert not hasattr(m3.tokenizer, "save_pretrained") and m3.cfg.image_token_id == cfg.image_token_id
