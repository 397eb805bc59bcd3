"""Drop-in test at the reference's three builder seams with the REAL HF classes in the loop.

A stand-in for `LlavaLlamaModel` is assembled from what `build_llm_and_tokenizer` / `build_vision_tower` / `build_mm_projector`
return in the reference: `transformers.Qwen2ForCausalLM`, a SiglipVisionTower-shaped wrapper around
`transformers.SiglipVisionModel` (attribute `.vision_tower`, llava/model/multimodal_encoder/siglip_encoder.py:25-63) and a torch
restatement of `MultimodalProjector("mlp_downsample")` with the reference's `layers.N` names (base_projector.py:145-160).  The HF
modules run in bf16 on the same GPU through torch's own kernels; `swap_in_hip_modules` then replaces all three in place and the
same call sites must give the same answers.  Tolerances (two bf16 pipelines): features rel-L2 <= 2e-2, logits <= 3e-2, greedy ids
equal wherever HF's top-1 margin exceeds 4x the observed max-abs logit error.
"""
import types

import pytest
import torch
import torch.nn as nn

from tests.gpu_util import rel_l2
from vila_amd import configs

pytestmark = pytest.mark.gpu


class _DownSample(nn.Module):                     # DownSampleBlock (base_projector.py:47-69): 2x2 space-to-depth, odd grids zero-padded
    def forward(self, x):
        B, N, C = x.shape
        g = int(N ** 0.5)
        x = x.reshape(B, g, g, C)
        if g % 2:
            x = torch.cat([x, x.new_zeros(B, 1, g, C)], 1)
            x = torch.cat([x, x.new_zeros(B, g + 1, 1, C)], 2)
        gp = x.shape[1]
        x = x.reshape(B, gp // 2, 2, gp // 2, 2, C).permute(0, 1, 3, 2, 4, 5)
        return x.reshape(B, (gp // 2) ** 2, 4 * C)


class _Projector(nn.Module):
    def __init__(self, c, h):
        super().__init__()
        self.layers = nn.Sequential(_DownSample(), nn.LayerNorm(4 * c), nn.Linear(4 * c, h), nn.GELU(), nn.Linear(h, h))

    def forward(self, x):
        return self.layers(x)


class _Tower(nn.Module):
    """SiglipVisionTower-shaped: `.vision_tower` holds a module whose parameters live under `vision_model.` — the layout of the
    reference's own modeling_siglip and of every VILA checkpoint.  (Recent transformers releases flattened SiglipVisionModel; the
    shim restores the nesting so the key names are the reference's.)"""

    def __init__(self, hf_model, select_layer=-2):
        super().__init__()
        if hasattr(hf_model, "vision_model"):
            self.vision_tower = hf_model
        else:
            self.vision_tower = nn.Module()
            self.vision_tower.vision_model = hf_model
        object.__setattr__(self, "_run", hf_model)       # not a registered submodule: no duplicate state_dict keys
        self.config = hf_model.config
        self.select_layer = select_layer

    def forward(self, images):
        out = self._run(images.to(next(self.parameters()).dtype), output_hidden_states=True)
        return out.hidden_states[self.select_layer]


@pytest.fixture(scope="module")
def vlm_pair():
    transformers = pytest.importorskip("transformers")
    cfg = configs.tiny("mlp_downsample", layers_v=3, layers_l=3)
    l, v = cfg.llm, cfg.vision
    torch.manual_seed(0)
    hl = transformers.Qwen2Config(vocab_size=l.vocab_size, hidden_size=l.hidden_size, intermediate_size=l.intermediate_size,
                                  num_hidden_layers=l.num_hidden_layers, num_attention_heads=l.num_attention_heads,
                                  num_key_value_heads=l.num_key_value_heads, rms_norm_eps=l.rms_norm_eps, rope_theta=l.rope_theta,
                                  tie_word_embeddings=False, max_position_embeddings=4096, use_sliding_window=False,
                                  eos_token_id=l.eos_token_id, pad_token_id=None, bos_token_id=None)
    hl._attn_implementation = "eager"
    hv = transformers.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                                         num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                                         layer_norm_eps=v.layer_norm_eps, hidden_act="gelu_pytorch_tanh")
    hv._attn_implementation = "eager"
    llm = transformers.Qwen2ForCausalLM(hl)
    for p in llm.parameters():                    # HF's default init (std 0.02) makes near-constant logits: widen for a meaningful test
        if p.dim() == 2:
            nn.init.normal_(p, std=0.05)
    vlm = nn.Module()
    vlm.llm = llm.eval().to("cuda", torch.bfloat16)
    vlm.vision_tower = _Tower(transformers.SiglipVisionModel(hv)).eval().to("cuda", torch.bfloat16)
    vlm.mm_projector = _Projector(v.hidden_size, l.hidden_size).eval().to("cuda", torch.bfloat16)
    vlm.config = types.SimpleNamespace(mm_projector_type="mlp_downsample", mm_vision_select_layer=-2)
    g = torch.Generator().manual_seed(1)
    px = (torch.rand(2, 3, v.image_size, v.image_size, generator=g) * 2 - 1).to("cuda", torch.bfloat16)
    emb = (torch.randn(1, 40, l.hidden_size, generator=g) * 0.5).to("cuda", torch.bfloat16)
    with torch.no_grad():
        ref = {"feat": vlm.vision_tower(px)}
        ref["proj"] = vlm.mm_projector(ref["feat"])
        ref["logits"] = vlm.llm(inputs_embeds=emb).logits.float()
        ref["ids"] = vlm.llm.generate(inputs_embeds=emb, attention_mask=torch.ones(1, 40, dtype=torch.long, device="cuda"), max_new_tokens=6,
                                      do_sample=False, eos_token_id=None, pad_token_id=0)
    from vila_amd.integration import swap_in_hip_modules
    got_cfg = swap_in_hip_modules(vlm)
    return cfg, got_cfg, vlm, px, emb, ref


def test_swap_keeps_config_and_module_contracts(vlm_pair):
    import dataclasses
    from vila_amd.modules import HipMultimodalProjector, HipQwen2ForCausalLM, HipSiglipVisionTower
    cfg, got_cfg, vlm, px, emb, ref = vlm_pair
    assert dataclasses.asdict(got_cfg.llm) == dataclasses.asdict(cfg.llm)
    assert dataclasses.asdict(got_cfg.vision) == dataclasses.asdict(cfg.vision)
    assert isinstance(vlm.llm, HipQwen2ForCausalLM) and isinstance(vlm.vision_tower, HipSiglipVisionTower)
    assert isinstance(vlm.mm_projector, HipMultimodalProjector)


def test_tower_and_projector_seams_match_hf(vlm_pair):
    cfg, got_cfg, vlm, px, emb, ref = vlm_pair
    feat = vlm.vision_tower(px)
    assert feat.shape == ref["feat"].shape
    assert rel_l2(feat, ref["feat"]) < 2e-2, f"tower rel={rel_l2(feat, ref['feat']):.3e}"
    proj = vlm.mm_projector(ref["feat"])
    assert proj.shape == ref["proj"].shape
    assert rel_l2(proj, ref["proj"]) < 2e-2, f"projector rel={rel_l2(proj, ref['proj']):.3e}"


def test_llm_seam_matches_hf_forward_and_generate(vlm_pair):
    cfg, got_cfg, vlm, px, emb, ref = vlm_pair
    out = vlm.llm(inputs_embeds=emb)
    lg = out.logits.float()
    assert lg.shape == ref["logits"].shape
    assert rel_l2(lg, ref["logits"]) < 3e-2, f"logits rel={rel_l2(lg, ref['logits']):.3e}"
    ids = vlm.llm.generate(inputs_embeds=emb, attention_mask=torch.ones(1, 40, dtype=torch.long, device="cuda"), max_new_tokens=6,
                           eos_token_id=-1)
    # first token: decided by the prefill logits both sides produced above
    err = float((lg[0, -1] - ref["logits"][0, -1]).abs().max())
    top2 = ref["logits"][0, -1].topk(2).values
    if float(top2[0] - top2[1]) > 4 * err:
        assert int(ids[0, 0]) == int(ref["ids"][0, 0])
    assert ids.shape == (1, 6)


def _collated_batch(device):
    """The batch the REFERENCE'S OWN DataCollator produced (tests/golden/collate_batch_ref.npz, oracle/make_golden_collate.py), rebuilt from the
    seeded pixel pool: exactly the dict `Trainer` hands `model(**batch)`."""
    import os
    import numpy as np
    from vila_amd import configs, synthetic
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "collate_batch_ref.npz"))
    cfg = configs.tiny_s2()
    pool = synthetic.make_pixels(cfg, 15, int(fx["seed"])).to(torch.bfloat16)
    images = [pool[int(k)].to(device) for k in fx["image_pool_index"]]
    videos = [pool[int(k):int(k) + int(n)].to(device) for k, n in zip(fx["video_first_pool_index"], fx["video_frames"])]
    blocks = [None if b[0] < 0 else (int(b[0]), int(b[1])) for b in fx["block_sizes"]]
    batch = {"input_ids": torch.from_numpy(fx["input_ids"]).to(device), "media": {"image": images, "video": videos},
             "media_config": {"image": {"block_sizes": blocks, "original_image_sizes": [None] * len(blocks)}, "video": {}},
             "labels": torch.from_numpy(fx["labels"]).to(device), "attention_mask": torch.from_numpy(fx["attention_mask"]).to(device),
             "gt_selection_maps": None}
    assert sorted(batch["media_config"]) == list(fx["media_config_keys"]) and sorted(batch["media_config"]["image"]) == list(fx["image_config_keys"])
    return cfg, pool, batch, blocks


def test_the_reference_collators_batch_goes_straight_into_the_hip_model():
    """`loss = model(**batch).loss; loss.backward()` with `batch` = the output of the reference's DataCollator (dynamic_s2 recipe: a 2 x 2-block
    image of 9 tiles, a text-only sample, a one-tile image + a 3-frame video; padded with the tokenizer's pad id, `gt_selection_maps=None` riding
    along as in the reference): training mode through the autograd seam, and eval mode — both against the fp32 oracle on the same batch."""
    from oracle import vila_oracle as O
    from vila_amd import synthetic
    from vila_amd.vlm import build_model
    cfg, pool, batch, blocks = _collated_batch("cuda")
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 23).items()}
    ids, labels, mask = batch["input_ids"].cpu(), batch["labels"].cpu(), batch["attention_mask"].cpu()
    tiles = [t.float().cpu() for t in batch["media"]["image"]]
    vids = [v.float().cpu() for v in batch["media"]["video"]]
    ref = float(O.vlm_sft_loss(tiles, ids, labels, mask, w, cfg, packed=True, block_sizes=blocks, videos=vids))
    model = build_model(cfg, weights=w)
    model.enable_autograd(use_c_abi=False)
    model.train()
    out = model(**batch)
    assert out.loss.requires_grad and abs(float(out.loss) - ref) < 1e-2 * abs(ref), (float(out.loss), ref)
    out.loss.backward()
    g = dict(model.mm_projector.named_parameters())["layers.1.weight"].grad
    assert g is not None and float(g.float().norm()) > 0
    model.eval()
    with torch.no_grad():
        ev = model(**batch)
    assert abs(float(ev.loss) - ref) < 1e-2 * abs(ref), (float(ev.loss), ref)
    print(f"reference-collated batch: training loss {float(out.loss):.5f}, eval loss {float(ev.loss):.5f}, oracle {ref:.5f}")
