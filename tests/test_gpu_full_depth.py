"""BASELINE configs[1] at FULL depth (NVILA-8B: 26 ViT + 28 LLM layers, 1 x 448^2 image + 512-token prompt, S = 769) against the
fp32 CPU oracle.

The oracle needs minutes and ~20 GB at this size, so it ran once (oracle/make_golden_full.py) and its KB-sized fingerprints are
committed as tests/golden/nvila8b_full_depth.npz (an ORACLE-executed fixture: the chain to the reference is oracle <- reference-executed
tiny-depth fixtures, tests/test_oracle_golden.py): top-32 logits of the prefill's last row and of 8 teacher-forced decode steps, the
greedy ids, a few tower / projector / embedding rows.  Here the SAME weights are drawn with the CPU generator (tensor by tensor,
~1 min), the HIP path runs, and the stated rules apply: hidden rows rel-L2 <= 2e-2 (tower, projector), logits <= 3e-2 on the stored
entries, token ids bit-exact at every step whose oracle top-1/top-2 margin exceeds 4x the observed max-abs logit error
(teacher-forced), and the free-running hipGraph decode must follow the oracle up to the first non-decisive step.
"""
import os

import numpy as np
import pytest
import torch

from tests.gpu_util import rel_l2
from vila_amd import configs, synthetic

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "nvila8b_full_depth.npz")
FINGERPRINT_KEYS = ("llm.model.layers.0.mlp.gate_proj.weight", "llm.model.layers.27.self_attn.q_proj.bias", "llm.lm_head.weight",
                    "vision_tower.vision_tower.vision_model.encoder.layers.25.mlp.fc1.weight", "mm_projector.layers.2.weight")


def test_full_depth_logits_and_ids_vs_oracle_golden():
    from vila_amd.vlm import build_model
    fx = np.load(GOLDEN)
    cfg = configs.nvila_8b()
    seed = int(fx["seed"])
    # the fixture's synthetic lm_head has heavy-tailed row norms (a peaked next-token distribution, oracle/make_golden_full.py)
    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = float(fx["lm_head_tail"]), int(fx["lm_head_tail_seed"]), float(fx["lm_head_tail_max"])
    specs = {n: (shape, kind) for n, shape, kind in synthetic.all_specs(cfg)}
    for i, k in enumerate(FINGERPRINT_KEYS):       # same CPU RNG stream as the host the golden was made on?
        shape, kind = specs[k]
        got = synthetic._draw(k, shape, kind, cfg, seed, "cpu").to(torch.bfloat16).float().reshape(-1)[:16].numpy()
        assert np.array_equal(got, fx[f"fp_w{i}"]), f"CPU generator stream differs from the golden's host for {k}: cannot compare"
    px = synthetic.make_pixels(cfg, 1, seed).to(torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 512, 1, seed)
    assert np.array_equal(px.float().reshape(-1)[:16].numpy(), fx["fp_pixels"]) and np.array_equal(ids.numpy(), fx["input_ids"])

    model = build_model(cfg, seed=seed, draw_device="cpu")
    pxg = px.cuda()
    feats = model.vision_tower(pxg)
    sel = torch.from_numpy(fx["vit_rows"])
    assert rel_l2(feats[0, [0, 511, 1023], :256], sel) < 2e-2, f"tower rows rel={rel_l2(feats[0, [0, 511, 1023], :256], sel):.3e}"
    assert abs(float(feats.float().norm()) / float(fx["vit_norm"]) - 1) < 1e-2
    proj = model.mm_projector(feats)
    psel = torch.from_numpy(fx["proj_rows"])
    assert rel_l2(proj[0, [0, 127, 255], :256], psel) < 2e-2, f"projector rows rel={rel_l2(proj[0, [0, 127, 255], :256], psel):.3e}"
    e, _, _ = model._embed(ids[None], {"image": [pxg[0]]})
    assert e.shape == (1, 769, cfg.llm.hidden_size)
    esel = torch.from_numpy(fx["embed_rows"])
    assert rel_l2(e[0, [0, 255, 256, 257, 768], :256], esel) < 2e-2

    gold = torch.from_numpy(fx["greedy_ids"])
    n = len(gold)
    out, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=gold, use_graph=False)
    top_ids = torch.from_numpy(fx["top_ids"]).long()
    top_vals = torch.from_numpy(fx["top_vals"])
    got = lg.float().cpu().gather(1, top_ids)
    rel = rel_l2(got, top_vals)
    assert rel < 3e-2, f"full-depth logits (top-32 entries of 1 prefill + {n - 1} decode rows) rel={rel:.3e}"
    # SURVEY §8c id rule with the error observed AT EACH STEP (max-abs over that step's 32 fixture entries): one global maximum over all
    # 8 x 32 entries made the rule hinge on a single outlier entry and on one step of this fixture (margins 0.2 .. 1.5 against a 0.37-0.41
    # worst entry: two summation orders of the same decode attention flipped it between "1 decisive step" and "none")
    err_t = (got - top_vals).abs().max(dim=1).values
    err = float(err_t.max())
    decisive = (top_vals[:, 0] - top_vals[:, 1]) > 4 * err_t
    print(f"full depth: per-step max-abs err {[round(float(x), 3) for x in err_t]}, margins {[round(float(x), 3) for x in (top_vals[:, 0] - top_vals[:, 1])]}")
    # the fixture was built so that most steps ARE decisive: a pass on one lucky step (round 2) is not accepted
    assert int(decisive.sum()) >= 6, (f"only {int(decisive.sum())} of {n} steps decisive: per-step err {err_t.tolist()}, "
                                      f"margins {(top_vals[:, 0] - top_vals[:, 1]).tolist()}")
    am = lg.float().cpu().argmax(-1)
    assert torch.equal(am[decisive], gold[decisive]), f"ids {am.tolist()} vs oracle {gold.tolist()} (err {err:.3e}, decisive {decisive.tolist()})"
    # free-running greedy through the captured hipGraph: identical to the oracle until the first non-decisive step
    free = model.generate(input_ids=ids[None], media={"image": [pxg[0]]}, max_new_tokens=n, eos_token_id=-1)[0].cpu()
    nd = (~decisive).nonzero().flatten()
    k = int(nd[0]) if nd.numel() else n
    assert torch.equal(free[:k], gold[:k]), f"free-running {free.tolist()} vs oracle {gold.tolist()} (first {k} must match)"
    print(f"full depth: logits rel {rel:.3e}, max-abs err {err:.3e}, decisive {int(decisive.sum())}/{n}, ids {am.tolist()}")


def test_lite3b_full_depth_vs_reference_executed_golden():
    """BASELINE configs[0] at its real size: NVILA-Lite-3B-shaped widths (hidden 2048, 16/2 heads, FFN 11008, tied head, 3x3 projector), 26 ViT +
    36 decoder layers, 1 image + 32-token prompt (S = 154), against the REFERENCE-EXECUTED fixture (reference SigLIP + projector + HF Qwen2 in
    fp32, oracle/make_golden_lite3b.py; the oracle's own run of the same case is held to it on CPU).  Same rules as the 8B test."""
    from vila_amd.vlm import build_model
    path = os.path.join(os.path.dirname(__file__), "golden", "nvila_lite3b_full_depth_ref.npz")
    fx = np.load(path)
    cfg = configs.nvila_lite_3b()
    seed = int(fx["seed"])
    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = float(fx["lm_head_tail"]), int(fx["lm_head_tail_seed"]), float(fx["lm_head_tail_max"])
    keys = ("llm.model.layers.0.mlp.gate_proj.weight", "llm.model.layers.35.self_attn.q_proj.bias", "llm.model.embed_tokens.weight",
            "vision_tower.vision_tower.vision_model.encoder.layers.25.mlp.fc1.weight", "mm_projector.layers.2.weight")
    specs = {n: (shape, kind) for n, shape, kind in synthetic.all_specs(cfg)}
    for i, k in enumerate(keys):
        shape, kind = specs[k]
        got = synthetic._draw(k, shape, kind, cfg, seed, "cpu").to(torch.bfloat16).float().reshape(-1)[:16].numpy()
        assert np.array_equal(got, fx[f"fp_w{i}"]), f"CPU generator stream differs from the golden's host for {k}: cannot compare"
    px = synthetic.make_pixels(cfg, 1, seed).to(torch.bfloat16)
    ids = torch.from_numpy(fx["input_ids"])
    assert np.array_equal(px.float().reshape(-1)[:16].numpy(), fx["fp_pixels"])
    model = build_model(cfg, seed=seed, draw_device="cpu")
    pxg = px.cuda()
    feats = model.vision_tower(pxg)
    assert rel_l2(feats[0, [0, 511, 1023], :256], torch.from_numpy(fx["vit_rows"])) < 2e-2
    proj = model.mm_projector(feats)
    n_img = proj.shape[1]
    assert n_img == 121                                              # ceil(32 / 3)^2 tokens of the 3x3 projector
    assert rel_l2(proj[0, [0, n_img // 2, n_img - 1], :256], torch.from_numpy(fx["proj_rows"])) < 2e-2
    e, _, _ = model._embed(ids[None], {"image": [pxg[0]]})
    S = e.shape[1]
    assert S == n_img + 1 + 32
    assert rel_l2(e[0, [0, n_img - 1, n_img, n_img + 1, S - 1], :256], torch.from_numpy(fx["embed_rows"])) < 2e-2
    gold = torch.from_numpy(fx["greedy_ids"])
    n = len(gold)
    out, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=gold, use_graph=False)
    top_ids, top_vals = torch.from_numpy(fx["top_ids"]).long(), torch.from_numpy(fx["top_vals"])
    got = lg.float().cpu().gather(1, top_ids)
    rel = rel_l2(got, top_vals)
    assert rel < 3e-2, f"Lite-3B full-depth logits rel={rel:.3e}"
    err_t = (got - top_vals).abs().max(dim=1).values
    decisive = (top_vals[:, 0] - top_vals[:, 1]) > 4 * err_t
    assert int(decisive.sum()) >= 6, f"only {int(decisive.sum())} of {n} steps decisive (err {err_t.tolist()})"
    am = lg.float().cpu().argmax(-1)
    assert torch.equal(am[decisive], gold[decisive]), f"ids {am.tolist()} vs reference {gold.tolist()} (decisive {decisive.tolist()})"
    free = model.generate(input_ids=ids[None], media={"image": [pxg[0]]}, max_new_tokens=n, eos_token_id=-1)[0].cpu()
    nd = (~decisive).nonzero().flatten()
    k = int(nd[0]) if nd.numel() else n
    assert torch.equal(free[:k], gold[:k]), f"free-running {free.tolist()} vs reference {gold.tolist()} (first {k} must match)"
    print(f"Lite-3B full depth: logits rel {rel:.3e}, per-step err {[round(float(x), 3) for x in err_t]}, margins "
          f"{[round(float(x), 3) for x in (top_vals[:, 0] - top_vals[:, 1])]}, decisive {int(decisive.sum())}/{n}")
