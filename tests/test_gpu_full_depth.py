"""BASELINE configs[0] / [1] / [2] at FULL depth against REFERENCE-EXECUTED fixtures.

`tests/golden/nvila8b_full_depth_ref.npz` (oracle/make_golden_full_ref.py) and `nvila_lite3b_full_depth_ref.npz` (oracle/make_golden_lite3b.py)
hold what the reference's own SigLIP + projector (loaded by file path) and HF `Qwen2ForCausalLM` in fp32 produce at 26 + 28 (26 + 36) layers on
seeded synthetic weights: KB-sized fingerprints.  Here the SAME weights are drawn with the CPU generator (tensor by tensor, ~1 min), the HIP
path runs, and the stated rules apply: hidden rows rel-L2 <= 2e-2 (tower, projector, spliced embeddings), logits <= 3e-2 on the stored top-32
entries, token ids bit-exact at every step whose top-1 / top-2 margin exceeds 4x the max-abs logit error observed at that step.

Round 4 (missing #5):
  * the recorded steps are TEACHER-FORCED WITH A RANDOM ID SEQUENCE stored in the fixture — every step has its own input token, hidden state and
    argmax token; the round-3 fixture followed greedy decoding into ONE attractor token with margins 60-150x the error.  Asserted: >= 6 of 8 steps
    decisive, >= 5 DISTINCT argmax tokens among them, and margins that are not all far outside the error (several decisive steps within 20x);
  * the 8B test reads the reference-executed file directly (the oracle-executed twin is held to it on CPU, tests/test_oracle_golden.py);
  * configs[2]'s forward: the SFT micro-batch the bench times (4 x (1 image + 512 tokens), labels on the last 256 text positions) at full
    depth — loss through the inference `forward(labels=)` (padded batch) AND through the trainer's packed forward, |delta| <= 1e-2 against HF's
    own loss; top-32 logits of 8 labelled rows per sample.
"""
import os

import numpy as np
import pytest
import torch

from tests.gpu_util import rel_l2
from vila_amd import configs, synthetic

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KEYS_8B = ("llm.model.layers.0.mlp.gate_proj.weight", "llm.model.layers.27.self_attn.q_proj.bias", "llm.lm_head.weight",
           "vision_tower.vision_tower.vision_model.encoder.layers.25.mlp.fc1.weight", "mm_projector.layers.2.weight")
KEYS_3B = ("llm.model.layers.0.mlp.gate_proj.weight", "llm.model.layers.35.self_attn.q_proj.bias", "llm.model.embed_tokens.weight",
           "vision_tower.vision_tower.vision_model.encoder.layers.25.mlp.fc1.weight", "mm_projector.layers.2.weight")


def _same_host_stream(cfg, fx, keys, seed):
    """Same CPU RNG stream as the host the golden was made on?  (else: cannot compare)"""
    specs = {n: (shape, kind) for n, shape, kind in synthetic.all_specs(cfg)}
    for i, k in enumerate(keys):
        shape, kind = specs[k]
        got = synthetic._draw(k, shape, kind, cfg, seed, "cpu").to(torch.bfloat16).float().reshape(-1)[:16].numpy()
        assert np.array_equal(got, fx[f"fp_w{i}"]), f"CPU generator stream differs from the golden's host for {k}: cannot compare"


def _teacher_forced_steps(model, e, fx, what, min_distinct=5, min_decisive=6):
    """The fixture's 8 steps: prefill row + 7 decode steps fed `forced_ids`.  Returns (logits [8, V] on the host, decisive mask, per-step error)."""
    forced = torch.from_numpy(fx["forced_ids"])
    want = torch.from_numpy(fx["tf_argmax_ids"])
    n = len(want)
    out, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=forced, use_graph=False)
    lg = lg.float().cpu()
    top_ids, top_vals = torch.from_numpy(fx["top_ids"]).long(), torch.from_numpy(fx["top_vals"])
    got = lg.gather(1, top_ids)
    rel = rel_l2(got, top_vals)
    assert rel < 3e-2, f"{what}: full-depth logits (top-32 entries of 1 prefill + {n - 1} decode rows) rel={rel:.3e}"
    # SURVEY §8c id rule with the error observed AT EACH STEP (max-abs over that step's 32 fixture entries)
    err_t = (got - top_vals).abs().max(dim=1).values
    margin = top_vals[:, 0] - top_vals[:, 1]
    decisive = margin > 4 * err_t
    ratio = margin / err_t.clamp_min(1e-9)
    print(f"{what}: logits rel {rel:.3e}; per-step max-abs err {[round(float(x), 3) for x in err_t]}, margins {[round(float(x), 3) for x in margin]}, "
          f"margin / err {[round(float(x), 1) for x in ratio]}, reference argmax {want.tolist()}")
    assert int(decisive.sum()) >= min_decisive, f"{what}: only {int(decisive.sum())} of {n} steps decisive (err {err_t.tolist()}, margins {margin.tolist()})"
    am = lg.argmax(-1)
    assert torch.equal(am[decisive], want[decisive]), f"{what}: ids {am.tolist()} vs reference {want.tolist()} (decisive {decisive.tolist()})"
    # the id evidence is real: different tokens win the decisive steps, by margins the path could have lost
    assert len(set(want[decisive].tolist())) >= min_distinct, f"{what}: the decisive steps carry only {len(set(want[decisive].tolist()))} distinct tokens"
    assert int((decisive & (ratio <= 20)).sum()) >= 3, f"{what}: margins far above the error on all but {int((decisive & (ratio <= 20)).sum())} steps: {ratio.tolist()}"
    return lg, decisive, err_t


def _free_running(model, ids, pxg, fx, err, what):
    """Greedy through the captured hipGraph: identical to the reference's greedy ids up to the first step whose (reference) margin is not
    decisive against the error observed on the teacher-forced steps."""
    gold, gm = torch.from_numpy(fx["greedy_ids"]), torch.from_numpy(fx["greedy_margins"])
    n = len(gold)
    free = model.generate(input_ids=ids[None], media={"image": [pxg[0]]}, max_new_tokens=n, eos_token_id=-1)[0].cpu()
    nd = (gm <= 4 * err).nonzero().flatten()
    k = int(nd[0]) if nd.numel() else n
    # (k may be 0: the prefill row's own top-2 margin can be below 4x the error — then the free run has nothing it must reproduce)
    assert torch.equal(free[:k], gold[:k]), f"{what}: free-running {free.tolist()} vs reference {gold.tolist()} (first {k} must match)"
    return k


@pytest.fixture(scope="module")
def nvila8b():
    from vila_amd.vlm import build_model
    fx = np.load(os.path.join(GOLDEN, "nvila8b_full_depth_ref.npz"))
    cfg = configs.nvila_8b()
    seed = int(fx["seed"])
    # the fixture's synthetic lm_head has tailed row norms chosen by oracle/make_golden_full.py (distinct winners, margins near the error)
    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = float(fx["lm_head_tail"]), int(fx["lm_head_tail_seed"]), float(fx["lm_head_tail_max"])
    _same_host_stream(cfg, fx, KEYS_8B, seed)
    model = build_model(cfg, seed=seed, draw_device="cpu")
    return fx, cfg, seed, model


def test_full_depth_logits_and_ids_vs_reference_executed_golden(nvila8b):
    fx, cfg, seed, model = nvila8b
    px = synthetic.make_pixels(cfg, 1, seed).to(torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 512, 1, seed)
    assert np.array_equal(px.float().reshape(-1)[:16].numpy(), fx["fp_pixels"]) and np.array_equal(ids.numpy(), fx["input_ids"])
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
    assert rel_l2(e[0, [0, 255, 256, 257, 768], :256], torch.from_numpy(fx["embed_rows"])) < 2e-2
    lg, decisive, err_t = _teacher_forced_steps(model, e, fx, "NVILA-8B full depth")
    k = _free_running(model, ids, pxg, fx, float(err_t.max()), "NVILA-8B full depth")
    print(f"NVILA-8B full depth: decisive {int(decisive.sum())}/8, free-running greedy follows the reference for {k} steps")


def test_full_depth_sft_forward_loss_and_logits_rows_vs_reference(nvila8b):
    """BASELINE configs[2] at 26 + 28 layers: the micro-batch the bench times, forward only, against HF's own loss (reference-executed)."""
    from oracle.make_golden_full import sft_batch
    fx, cfg, seed, model = nvila8b
    # this pin uses the PLAIN synthetic head (`sft_head_tail` = 0): the tailed one is the id test's (it quadruples the logit scale and with it
    # the loss error, while the stated |delta| <= 1e-2 is for a loss ~ ln V) — same decoder, same hidden states, only lm_head is re-drawn
    assert float(fx["sft_head_tail"]) == 0.0
    tail = cfg.lm_head_tail
    cfg.lm_head_tail = 0.0
    specs = {n: (shape, kind) for n, shape, kind in synthetic.all_specs(cfg)}
    with torch.no_grad():
        model.llm.lm_head.weight.copy_(synthetic._draw("llm.lm_head.weight", *specs["llm.lm_head.weight"], cfg, seed, "cpu"))
    cfg.lm_head_tail = tail
    spx, sids, slabels = sft_batch(cfg, seed)
    assert np.array_equal(sids.numpy(), fx["sft_input_ids"]) and np.array_equal(slabels.numpy(), fx["sft_labels"])
    assert np.array_equal(spx.reshape(4, -1)[:, :16].numpy(), fx["sft_fp_pixels"])
    images = [spx[i].to(torch.bfloat16).cuda() for i in range(4)]
    n_items = int(fx["sft_num_items"])
    want = float(fx["sft_loss"])
    # (1) the inference forward(labels=) — llava_llama.py:94-159 outside training: a padded batch (all rows 769 long), loss = sum / num_items
    r = model(input_ids=sids, media={"image": images}, labels=slabels, num_items_in_batch=n_items)
    got = float(r.loss)
    print(f"configs[2] full-depth forward loss: HIP {got:.5f} vs reference {want:.5f}")
    assert abs(got - want) <= 1e-2, f"forward(labels) loss {got:.5f} vs HF {want:.5f}"
    # (2) logits of 8 labelled rows per sample (top-32 entries of each) through the packed prefill of the same four sequences
    e, lab, _ = model._embed(sids, {"image": images}, None, slabels)
    S = e.shape[1]
    rows = torch.from_numpy(fx["sft_rows"]).long()
    assert S == 769 and int(rows.max()) < S
    cu = torch.arange(0, 5 * S, S, dtype=torch.int32, device="cuda")
    pos = torch.arange(S, dtype=torch.int32, device="cuda").repeat(4)
    last = torch.cat([rows + b * S for b in range(4)]).to(torch.int32).cuda()
    out = model.llm.prefill_packed(e.reshape(4 * S, -1), pos, cu, S, last_rows=last)
    lgr = out.last_logits.float().cpu().view(4, len(rows), -1)
    top_ids, top_vals = torch.from_numpy(fx["sft_top_ids"]).long(), torch.from_numpy(fx["sft_top_vals"])
    gotv = lgr.gather(2, top_ids)
    assert rel_l2(gotv, top_vals) < 3e-2, f"logits rows of the packed 4 x 769 batch rel={rel_l2(gotv, top_vals):.3e}"
    # (3) the trainer's packed forward (the step the bench times: SFTTrainer.forward_backward, loss before any update)
    from vila_amd.train import SFTTrainer
    tr = SFTTrainer(model, lr=0.0, weight_decay=0.0, optimizer_state=False)
    loss = float(tr.forward_backward(sids, images, slabels, None, n_items))
    print(f"configs[2] full-depth packed training forward loss: HIP {loss:.5f} vs reference {want:.5f}")
    assert abs(loss - want) <= 1e-2, f"packed training forward loss {loss:.5f} vs HF {want:.5f}"


FULL_DEPTH_COS_MIN = 0.998      # the stated exception to GRAD_COS_MIN = 0.999 for weight gradients 54 bf16 layers deep (see the test's docstring)


def test_full_depth_backward_probe_gradients_vs_reference(nvila8b):
    """Round 6 (parity hardening ii): the BACKWARD at the full 26 + 28 layer depth.  One sample of configs[2] (1 image + 512
    tokens, S = 769, 256 targets) through `SFTTrainer.forward_backward`; the gradients of 14 probe tensors that together see the whole chain —
    the patch embedding (its gradient has crossed all 54 layers), tower layers 0 / 25, both projector tensors, decoder layers 0 / 13 / 27, the
    final norm, lm_head — against torch autograd through the REFERENCE'S OWN modules in fp32 (reference SigLIP + projector by file path, HF
    Qwen2ForCausalLM with HF's loss; oracle/make_golden_full_grads_ref.py, tests/golden/nvila8b_full_depth_grads_ref.npz: a seeded random subset of
    <= 16 384 elements per tensor + the full tensor's norm).
    Bounds.  SURVEY 8c states cosine >= 0.999; at THIS depth that is a statement about bf16, not about the kernels: a weight gradient dY^T X
    inherits the error of its X, and bf16 activations 54 layers deep differ from fp32 by a few % (the forward pins allow rel-L2 2e-2 per stage).
    Stated per-tensor exception for this test, with the measurements it was read from (profiles/r06_full_depth_backward_probes.txt): every probe
    >= 0.998 (measured 0.99847 ... 0.99981; >= 0.999 for the last layer's o_proj, the final norm and lm_head), AND — the fixture's second pass —
    every probe at least as close to the fp32 gradient as THE REFERENCE'S OWN MODULES RUN IN BF16 are (`cosb_<k>`: 0.985 ... 0.999 on the same
    tensors: the HIP step's fp32 accumulation and fp32 softmax / norm statistics put it an order of magnitude closer than a plain bf16 run).
    Norms within 3 %, the loss within 1e-2."""
    from oracle.make_golden_full import sft_batch
    from tests.gpu_util import GRAD_COS_MIN, grad_cos
    from vila_amd.train import SFTTrainer
    fx, cfg, seed, model = nvila8b
    gx = np.load(os.path.join(os.path.dirname(__file__), "golden", "nvila8b_full_depth_grads_ref.npz"))
    tail = cfg.lm_head_tail
    cfg.lm_head_tail = 0.0                                        # the plain synthetic head, as in the forward pin above
    specs = {n: (shape, kind) for n, shape, kind in synthetic.all_specs(cfg)}
    with torch.no_grad():
        model.llm.lm_head.weight.copy_(synthetic._draw("llm.lm_head.weight", *specs["llm.lm_head.weight"], cfg, seed, "cpu"))
    cfg.lm_head_tail = tail
    spx, sids, slabels = sft_batch(cfg, seed)
    assert int(gx["seed"]) == seed and np.array_equal(sids[0].numpy(), gx["input_ids"]) and np.array_equal(slabels[0].numpy(), gx["labels"])
    assert np.array_equal(spx[0].reshape(-1)[:16].numpy(), gx["fp_pixels"])
    n_items = int(gx["num_items"])
    tr = SFTTrainer(model, lr=0.0, weight_decay=0.0, optimizer_state=False)
    loss = float(tr.forward_backward(sids[0:1], [spx[0].to(torch.bfloat16).cuda()], slabels[0:1], None, n_items))
    torch.cuda.synchronize()
    want = float(gx["loss"])
    print(f"full-depth one-sample loss: HIP {loss:.5f} vs reference {want:.5f}")
    assert abs(loss - want) <= 1e-2 * max(1.0, abs(want)), (loss, want)
    grads = tr.flat.named_grads()
    report, bad = [], []
    for k, name in enumerate(gx["names"].tolist()):
        idx = torch.from_numpy(gx[f"gi_{k}"]).long()
        ref = torch.from_numpy(gx[f"gv_{k}"])
        got = grads[name].reshape(-1)[idx.cuda()].float().cpu()
        cos = grad_cos("full_depth", name, got, ref)
        cos_ref_bf16 = float(gx[f"cosb_{k}"])
        ratio = float(got.double().norm() / ref.double().norm())
        full = float(grads[name].float().norm()) / float(gx[f"gn_{k}"])
        report.append(f"{name.split('.', 2)[-1][-48:]:48s} cos {cos:.5f} (reference in bf16: {cos_ref_bf16:.5f}) |subset| ratio {ratio:.4f} |full| ratio {full:.4f}")
        tight = name in ("llm.model.layers.27.self_attn.o_proj.weight", "llm.model.norm.weight", "llm.lm_head.weight")
        if cos < (GRAD_COS_MIN if tight else FULL_DEPTH_COS_MIN) or cos < cos_ref_bf16 or not (0.97 <= ratio <= 1.03) or not (0.97 <= full <= 1.03):
            bad.append((name, round(cos, 5), round(ratio, 4), round(full, 4)))
    print("\n".join(report))
    assert not bad, bad


def test_lite3b_full_depth_vs_reference_executed_golden():
    """BASELINE configs[0] at its real size: NVILA-Lite-3B-shaped widths (hidden 2048, 16/2 heads, FFN 11008, tied head, 3x3 projector), 26 ViT +
    36 decoder layers, 1 image + 32-token prompt (S = 154), against the REFERENCE-EXECUTED fixture (reference SigLIP + projector + HF Qwen2 in
    fp32, oracle/make_golden_lite3b.py; the oracle's own run of the same case is held to it on CPU).  Same rules as the 8B test."""
    from vila_amd.vlm import build_model
    fx = np.load(os.path.join(GOLDEN, "nvila_lite3b_full_depth_ref.npz"))
    cfg = configs.nvila_lite_3b()
    seed = int(fx["seed"])
    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = float(fx["lm_head_tail"]), int(fx["lm_head_tail_seed"]), float(fx["lm_head_tail_max"])
    cfg.lm_head_tail_unit_rows = tuple(int(r) for r in fx["lm_head_tail_unit_rows"])      # rows the decoder can see: scale 1 (make_golden_lite3b.py)
    _same_host_stream(cfg, fx, KEYS_3B, seed)
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
    # (>= 3 distinct winners here, not 5: the final hidden states of this 36-layer random decoder are nearly parallel from step to step, so however
    # the tied head's tail is drawn, a few heavy rows win most steps — oracle/make_golden_lite3b.py searches 1792 tails and keeps the most diverse)
    # (>= 5 decisive steps asked, 6 chosen by the calibrated search: margins this close to the error move when ANY kernel on the path changes its
    # summation order — the round-4 tower LayerNorm fusion turned 6 decisive steps into 5 until the calibration was re-measured)
    lg, decisive, err_t = _teacher_forced_steps(model, e, fx, "Lite-3B full depth", min_distinct=3, min_decisive=5)
    k = _free_running(model, ids, pxg, fx, float(err_t.max()), "Lite-3B full depth")
    print(f"Lite-3B full depth: decisive {int(decisive.sum())}/8, free-running greedy follows the reference for {k} steps")
