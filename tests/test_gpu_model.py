"""GPU parity tests, model level: the drop-in modules (through the C-ABI) against
  (1) the committed golden fixtures produced by EXECUTING the reference's modules (tests/golden, oracle/make_golden.py)
  (2) the CPU oracle on the same seeded inputs, at tiny and at NVILA-8B widths (few layers so the fp32 oracle takes seconds)
Stated tolerances (bf16 GPU vs fp32 CPU): hidden states rel-L2 <= 2e-2, logits <= 3e-2;
greedy token ids bit-exact wherever the oracle's top-1/top-2 margin exceeds 4x the observed max-abs logit error.
"""
import os

import numpy as np
import pytest
import torch

from oracle import vila_oracle as O
from tests.gpu_util import margin_aware_ids, max_abs, rel_l2
from vila_amd import configs, synthetic

pytestmark = pytest.mark.gpu

CASES = {
    "tiny_2x2": (configs.tiny("mlp_downsample"), 0),
    "tiny_2x2fix": (configs.tiny("mlp_downsample_2x2_fix", image=70), 1),
    "tiny_3x3_tied": (configs.tiny("mlp_downsample_3x3_fix", tied=True), 2),
}


def _bf16_weights(cfg, seed):
    """The oracle sees exactly the bf16-rounded weights the GPU uses, so only arithmetic differs."""
    w = synthetic.make_weights(cfg, seed)
    return {k: v.to(torch.bfloat16).float() for k, v in w.items()}


@pytest.fixture(scope="module", params=list(CASES))
def case(request, golden_dir):
    from vila_amd.vlm import build_model
    cfg, seed = CASES[request.param]
    fx = np.load(os.path.join(golden_dir, request.param + ".npz"))
    w = _bf16_weights(cfg, seed)
    model = build_model(cfg, weights=w)
    return cfg, seed, fx, w, model


def test_vision_tower_vs_golden_and_oracle(case):
    cfg, seed, fx, w, model = case
    px = synthetic.make_pixels(cfg, 2, seed).to(torch.bfloat16)
    out = model.vision_tower(px.cuda())
    ref = O.vision_tower_forward(px.float(), w, cfg.vision)
    assert rel_l2(out, ref) < 2e-2, f"vs oracle rel={rel_l2(out, ref):.3e}"
    assert rel_l2(out, torch.from_numpy(fx["vit_selected"])) < 2e-2, f"vs golden rel={rel_l2(out, torch.from_numpy(fx['vit_selected'])):.3e}"


def test_projector_vs_golden_and_oracle(case):
    cfg, seed, fx, w, model = case
    feats = torch.from_numpy(fx["vit_selected"]).to(torch.bfloat16)
    out = model.mm_projector(feats.cuda())
    ref = O.projector_forward(feats.float(), w, cfg.mm_projector_type)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-2, f"vs oracle rel={rel_l2(out, ref):.3e}"
    assert rel_l2(out, torch.from_numpy(fx["projector_out"])) < 2e-2


def test_embed_splice_vs_golden(case):
    cfg, seed, fx, w, model = case
    px = synthetic.make_pixels(cfg, 2, seed).to(torch.bfloat16)
    ids = torch.from_numpy(fx["input_ids"])[None]
    e, labels, mask = model._embed(ids, {"image": [px[0].cuda()]})
    assert bool(mask.all()) and e.shape[1] == fx["spliced_embeds"].shape[1]
    assert rel_l2(e, torch.from_numpy(fx["spliced_embeds"])) < 2e-2
    n_img = cfg.tokens_per_tile + 1
    assert bool((labels[0, :n_img] == -100).all())
    # text rows are exact table rows
    tab = w["llm.model.embed_tokens.weight"]
    assert torch.equal(e[0, n_img:].float().cpu(), tab[ids[0, 1:]])


def test_llm_prefill_hidden_and_logits(case):
    cfg, seed, fx, w, model = case
    e = torch.from_numpy(fx["spliced_embeds"]).to(torch.bfloat16)
    S = e.shape[1]
    pos = torch.arange(S, dtype=torch.int32, device="cuda")
    r = model.llm.prefill_packed(e[0].cuda(), pos, None, S, want_all_logits=True, want_layer_hidden=True)
    logits, _, hs = O.qwen2_forward(e.float(), w, cfg.llm, return_hidden=True)
    for i in range(cfg.llm.num_hidden_layers + 1):
        assert rel_l2(r.layer_hidden[i], hs[i][0]) < 2e-2, f"layer {i} rel={rel_l2(r.layer_hidden[i], hs[i][0]):.3e}"
    assert rel_l2(r.all_logits, logits[0]) < 3e-2, f"logits rel={rel_l2(r.all_logits, logits[0]):.3e}"
    assert rel_l2(r.all_logits[-1], torch.from_numpy(fx["llm_logits_last"])) < 4e-2


def test_greedy_ids_margin_aware_bit_exact(case):
    cfg, seed, fx, w, model = case
    e = torch.from_numpy(fx["spliced_embeds"]).to(torch.bfloat16)
    gold = torch.from_numpy(fx["greedy_ids"])
    n = len(gold)
    # teacher-forced: feed the reference's ids, compare argmax at every step whose margin is decisive
    ids_o, lg_o = O.greedy_generate(e.float(), w, cfg, n, stop_at_eos=False, forced_ids=gold)
    out, lg = model.llm.generate(inputs_embeds=e.cuda(), max_new_tokens=n, return_logits=True, forced_ids=gold, use_graph=False)
    err = max_abs(lg, lg_o)
    top2 = lg_o.topk(2, -1).values
    margin = (top2[:, 0] - top2[:, 1])
    decisive = margin > 4 * err
    assert int(decisive.sum()) >= 1, f"no decisive step: margins {margin.tolist()} err {err:.3e}"
    assert err < 0.08 * float(lg_o.abs().max()), f"max-abs logit error {err:.3e} vs max |logit| {float(lg_o.abs().max()):.3e}"
    got = out[0].cpu()
    assert torch.equal(got[decisive], ids_o[decisive]), f"ids {got.tolist()} vs {ids_o.tolist()} (err {err:.3e})"
    # free-running greedy through the hipGraph path must reproduce the eager path exactly
    free_eager = model.llm.generate(inputs_embeds=e.cuda(), max_new_tokens=n, use_graph=False, eos_token_id=-1)
    free_graph = model.llm.generate(inputs_embeds=e.cuda(), max_new_tokens=n, use_graph=True, eos_token_id=-1)
    assert torch.equal(free_eager, free_graph)
    if bool(decisive.all()):
        assert torch.equal(free_graph[0].cpu(), gold), f"{free_graph[0].tolist()} vs golden {gold.tolist()}"


def test_decode_matches_prefill_logits(case):
    """Self-consistency: the M=1 decode kernels (GEMV, split-KV attention, fused RoPE/KV append) must reproduce the
    logits the prefill kernels (MFMA GEMM, flash attention) give at the same positions."""
    cfg, seed, fx, w, model = case
    e = torch.from_numpy(fx["spliced_embeds"]).to(torch.bfloat16).cuda()
    gold = torch.from_numpy(fx["greedy_ids"])
    n = len(gold)
    _, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=gold, use_graph=False)
    tail = model.llm.embed_tokens(gold[: n - 1].cuda())
    full = torch.cat([e[0], tail], 0)
    S = full.shape[0]
    r = model.llm.prefill_packed(full, torch.arange(S, dtype=torch.int32, device="cuda"), None, S, want_all_logits=True)
    ref = r.all_logits[e.shape[1] - 1:]
    assert rel_l2(lg, ref) < 1.5e-2, f"decode vs prefill rel={rel_l2(lg, ref):.3e}"


@pytest.mark.parametrize("mode", [1, 2])
def test_decode_attention_over_256_key_slices_matches_the_single_block_kernel(mode):
    """vila_decode_force_attn: per-head blocks over 256-key slices + the merge in the o_proj GEMV prologue (2 = default, 1 = fewer o_proj
    blocks) must give the logits of the one-block-per-head kernel (0; same math, another summation order) — on a context long enough for
    three active slices, eager and through the captured graph."""
    from vila_amd import _lib
    from vila_amd.vlm import build_model
    lib = _lib.load()
    cfg = configs.tiny("mlp_downsample")
    model = build_model(cfg, seed=5)
    H = cfg.llm.hidden_size
    g = torch.Generator().manual_seed(3)
    e = (torch.randn(1, 600, H, generator=g) * 0.5).to(torch.bfloat16).cuda()          # 600 prompt rows -> slices 0..2 active while decoding
    n = 6
    lib.vila_decode_force_attn(0)
    model.llm._invalidate()
    ids0, lg0 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, use_graph=False, eos_token_id=-1)
    try:
        lib.vila_decode_force_attn(mode)
        model.llm._invalidate()
        ids1, lg1 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=ids0[0].cpu(), use_graph=False, eos_token_id=-1)
        idsg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)
        idse = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=False, eos_token_id=-1)
    finally:
        lib.vila_decode_force_attn(2)
        model.llm._invalidate()
    assert rel_l2(lg1, lg0) < 5e-3, f"split-256 vs single-block decode logits rel={rel_l2(lg1, lg0):.3e}"
    assert torch.equal(idsg, idse)


def test_norm_fused_into_the_splitk_reduce_equals_the_separate_launches():
    """Round 4: where a prefill GEMM is K-sliced (LLM down_proj at 512+ rows, tower fc2 at one image) its reduce kernel also writes the NEXT
    block's normalisation (gemm256.hip splitk_reduce_norm_kernel).  LLM: same per-element arithmetic and the same summation order as
    norm_kernel<RMS> -> all-row logits BIT-equal with the fusion on and off.  Tower: the separate path uses the one-wave-per-row LayerNorm
    (another fp32 summation order) -> features equal to bf16 rounding."""
    from vila_amd import _lib
    from vila_amd.vlm import build_model
    lib = _lib.load()
    cfg = configs.reduced_8b(layers_v=3, layers_l=3, vocab=32000)
    cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
    model = build_model(cfg, seed=13)
    g = torch.Generator().manual_seed(13)
    T = 640                                                        # >= 512 rows: down_proj (K = 18944) takes the split-K path
    e = (torch.randn(T, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16).cuda()
    pos = torch.arange(T, dtype=torch.int32, device="cuda")
    px = synthetic.make_pixels(cfg, 1, 13).to(torch.bfloat16).cuda()
    out = {}
    try:
        for on in (0, 1):
            lib.vila_gemm_force_fuse_norm(on)
            r = model.llm.prefill_packed(e, pos, None, T, want_all_logits=True, want_layer_hidden=True)
            out[on] = (r.all_logits.clone(), r.layer_hidden.clone(), model.vision_tower(px).clone())
    finally:
        lib.vila_gemm_force_fuse_norm(1)
    assert torch.equal(out[0][1], out[1][1]), "decoder hidden states differ with the fused RMSNorm"
    assert torch.equal(out[0][0], out[1][0]), "logits differ with the fused RMSNorm"
    assert rel_l2(out[1][2], out[0][2]) < 2e-3, f"tower features fused vs separate LayerNorm rel={rel_l2(out[1][2], out[0][2]):.3e}"


def test_rope_and_kv_scatter_fused_into_the_qkv_reduce_equal_the_separate_launches():
    """Round 6: at >= 512 rows the prefill's q/k/v projection is K-sliced and its reduce adds the bias, rotates q and k (HF's roundings) and writes
    K / V into the cache (gemm256.hip splitk_reduce_rope_kernel) — rope_kv_kernel's work; o_proj is K-sliced with the post-attention RMSNorm in its
    reduce.  With every eligible GEMM pinned to the K-sliced kernel (force_tile 5) the fused and the separate forms run the SAME GEMM kernels, the
    same slice order and the same per-element expressions -> hidden states, logits and the KV cache must be BIT-equal.  Against the ring-GEMM path of
    rounds 1-5 (another fp32 summation order) the logits agree to bf16 rounding.  Packed: two sequences, so seq_of_tok / positions index the cache."""
    from vila_amd import _lib
    from vila_amd.vlm import build_model
    lib = _lib.load()
    cfg = configs.reduced_8b(layers_v=1, layers_l=3, vocab=32000)
    cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
    model = build_model(cfg, seed=17)
    g = torch.Generator().manual_seed(17)
    lens = [400, 369]                                              # T = 769 = 3 x 256 + 1: the benchmark's row count (extra-row fragment path)
    T = sum(lens)
    e = (torch.randn(T, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16).cuda()
    pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).cuda()
    cu = torch.tensor([0, lens[0], T], dtype=torch.int32, device="cuda")
    seq = torch.cat([torch.full((n,), i, dtype=torch.int32) for i, n in enumerate(lens)]).cuda()
    out = {}
    try:
        for name, tile, fuse in (("ring", 0, 0), ("sliced", 5, 0), ("fused", 5, 1), ("default", 0, 1)):
            lib.vila_gemm_force_tile(tile)
            lib.vila_prefill_force_fusions(fuse, fuse)
            cache = model.llm.new_cache(512, n_slots=2)
            r = model.llm.prefill_packed(e, pos, cu, max(lens), cache=cache, seq_of_tok=seq, want_all_logits=True, want_layer_hidden=True)
            out[name] = (r.all_logits.clone(), r.layer_hidden.clone(), cache.k.clone(), cache.v.clone())
    finally:
        lib.vila_gemm_force_tile(0)
        lib.vila_prefill_force_fusions(-1, -1)
    for i, what in enumerate(("logits", "hidden states", "K cache", "V cache")):
        assert torch.equal(out["sliced"][i], out["fused"][i]), f"{what} differ between the fused reduce and reduce + rope_kv / norm launches"
    assert rel_l2(out["default"][0], out["fused"][0]) < 2e-3          # (force_tile 5 also moves gate/up off the 256-wide kernel: not bit-comparable)
    assert torch.equal(out["default"][2][0], out["fused"][2][0]), "layer 0's K cache: the default dispatch is not the fused K-sliced path at 769 rows"
    assert float(out["fused"][2].float().abs().sum()) > 0 and float(out["fused"][3].float().abs().sum()) > 0
    for s, n in enumerate(lens):                                   # rows beyond a sequence's length stay untouched
        assert float(out["fused"][2][:, s, :, n:].float().abs().sum()) == 0
    # three layers of bf16 activations with another fp32 summation order in two GEMMs per layer: measured 7.7e-3 (the path's stated logits tolerance is 3e-2)
    assert rel_l2(out["fused"][0], out["ring"][0]) < 2e-2, f"K-sliced vs ring path logits rel={rel_l2(out['fused'][0], out['ring'][0]):.3e}"
    assert rel_l2(out["fused"][2], out["ring"][2]) < 2e-2


def test_chained_decode_step_equals_the_plain_step():
    """The opt-in chained decode step (vila_decode_force_chain(1): kernels alternate over two streams, stream their weights while the predecessor
    finishes and wait on device-side arrival counts; api.hip — measured slower than the plain step and OFF by default, profiles/
    r04_decode_chain_ab.log) must produce the plain step's logits BIT FOR BIT (same kernels, same summation order; only the hand-off differs) at
    NVILA-8B widths, eager and through the captured graph, and must not report a given-up wait."""
    from vila_amd import _lib
    from vila_amd.vlm import build_model
    lib = _lib.load()
    cfg = configs.reduced_8b(layers_v=2, layers_l=3, vocab=32000)
    cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
    model = build_model(cfg, seed=11)
    g = torch.Generator().manual_seed(11)
    e = (torch.randn(1, 300, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16).cuda()
    n = 12
    lib.vila_decode_force_chain(0)
    model.llm._invalidate()
    ids0, lg0 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, use_graph=False, eos_token_id=-1)
    free0 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)
    try:
        lib.vila_decode_force_chain(1)
        model.llm._invalidate()
        ids1, lg1 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, use_graph=False, eos_token_id=-1)
        free1 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)      # raises if a wait gave up
        again = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)      # replay: counters re-zeroed per token
    finally:
        lib.vila_decode_force_chain(0)
        model.llm._invalidate()
    assert torch.equal(lg1, lg0), f"chained vs plain decode logits differ: max {float((lg1 - lg0).abs().max()):.3e}"
    assert torch.equal(ids1, ids0) and torch.equal(free1, free0) and torch.equal(again, free0)


@pytest.mark.parametrize("shape", ["8b_width", "tiny", "lite3b_width"])
def test_persistent_decode_step_equals_the_launch_path(shape):
    """Round 6: the batch-1 decode token as ONE persistent launch (decode_persist.hip: 5 phases per layer behind fence-free grid barriers, the
    next phase's weights streaming across every barrier) against the per-kernel step (prologue + 5 launches per layer + lm_head).  Same
    arithmetic operation for operation, so the logits must be equal BIT FOR BIT — eager and through the captured graph, over a context that
    crosses a 256-key slice boundary, and no bounded wait may give up.  Shapes: NVILA-8B widths (28 / 4 heads, K = 3584 / 18944), the tiny
    config of the golden fixtures, Lite-3B widths (16 / 2 heads, hidden 2048: a row pair does not fill its 7-chunk batch)."""
    from vila_amd import _lib
    from vila_amd.vlm import build_model
    lib = _lib.load()
    if shape == "8b_width":
        cfg = configs.reduced_8b(layers_v=2, layers_l=3, vocab=32000)
        cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
        S = 250                                                       # 250 + 12 steps crosses key 256: a second slice appears mid-run
    elif shape == "lite3b_width":
        cfg = configs.nvila_lite_3b()
        cfg.llm.num_hidden_layers, cfg.vision.num_hidden_layers = 2, 2
        S = 120
    else:
        cfg = configs.tiny("mlp_downsample")
        S = 20
    model = build_model(cfg, seed=12)
    g = torch.Generator().manual_seed(12)
    e = (torch.randn(1, S, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16).cuda()
    n = 12
    try:
        lib.vila_decode_force_persist(0)
        model.llm._invalidate()
        ids0, lg0 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, use_graph=False, eos_token_id=-1)
        free0 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)
        lib.vila_decode_force_persist(1)
        model.llm._invalidate()
        ids1, lg1 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, use_graph=False, eos_token_id=-1)
        free1 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)      # raises if a wait gave up
        again = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)      # replay: barrier words re-zeroed per token
    finally:
        lib.vila_decode_force_persist(0)
        model.llm._invalidate()
    assert torch.isfinite(lg1).all()
    assert torch.equal(lg1, lg0), f"persistent vs per-kernel decode logits differ: max {float((lg1 - lg0).abs().max()):.3e}"
    assert torch.equal(ids1, ids0) and torch.equal(free1, free0) and torch.equal(again, free0)


def test_vlm_generate_end_to_end(case):
    cfg, seed, fx, w, model = case
    px = synthetic.make_pixels(cfg, 2, seed).to(torch.bfloat16)
    ids = torch.from_numpy(fx["input_ids"])[None]
    n = len(fx["greedy_ids"])
    out = model.generate(input_ids=ids, media={"image": [px[0].cuda()]}, max_new_tokens=n, eos_token_id=-1)
    assert out.shape == (1, n)
    ids_o, lg_o = O.vlm_generate([px[0].float()], ids[0], w, cfg, n, stop_at_eos=False)
    # the whole VLM path (tower -> projector -> splice -> prefill -> decode), teacher-forced with the oracle's ids: bit-exact ids at
    # every decisive step, and the free-running graph path must follow the oracle up to the first non-decisive step
    e, _, _ = model._embed(ids, {"image": [px[0].cuda()]})
    _, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=ids_o, use_graph=False)
    assert rel_l2(lg, lg_o) < 3e-2, f"end-to-end logits rel={rel_l2(lg, lg_o):.3e}"
    margin_aware_ids(lg, lg_o, ids_o, free_ids=out[0])


def test_forward_loss_padded_batch(case):
    cfg, seed, fx, w, model = case
    e = torch.from_numpy(fx["train_embeds"]).to(torch.bfloat16)
    m = torch.from_numpy(fx["train_mask"]).bool()
    lab = torch.from_numpy(fx["train_labels"])
    out = model.llm(inputs_embeds=e.cuda(), attention_mask=m.cuda(), labels=lab.cuda(), num_items_in_batch=int(fx["train_num_items"]))
    assert abs(float(out.loss) - float(fx["train_loss"])) < 1e-2 * abs(float(fx["train_loss"])), (float(out.loss), float(fx["train_loss"]))


# ---------------------------------------------------------------------------------------------------------------------
# NVILA-8B widths, few layers: the real GEMM / attention shapes against the fp32 CPU oracle
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def wide():
    from vila_amd.vlm import build_model
    cfg = configs.reduced_8b(layers_v=3, layers_l=2, vocab=32000)
    cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
    w = _bf16_weights(cfg, 7)
    model = build_model(cfg, weights=w)
    return cfg, w, model


def test_wide_vision_projector(wide):
    cfg, w, model = wide
    px = synthetic.make_pixels(cfg, 1, 7).to(torch.bfloat16)
    feat = model.vision_tower(px.cuda())
    ref = O.vision_tower_forward(px.float(), w, cfg.vision)
    assert feat.shape == (1, 1024, 1152)
    assert rel_l2(feat, ref) < 2e-2, f"vit rel={rel_l2(feat, ref):.3e}"
    out = model.mm_projector(feat)
    refp = O.projector_forward(ref, w, cfg.mm_projector_type)
    assert out.shape == (1, 256, 3584)
    assert rel_l2(out, refp) < 2e-2, f"proj rel={rel_l2(out, refp):.3e}"


def test_wide_llm_prefill_and_decode(wide):
    cfg, w, model = wide
    px = synthetic.make_pixels(cfg, 1, 7).to(torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 32, 1, 7)[None]
    e, _, _ = model._embed(ids, {"image": [px[0].cuda()]})
    assert e.shape == (1, 289, 3584)
    e_ref, _ = O.vlm_prefill_embeds([px[0].float()], ids[0], w, cfg)
    assert rel_l2(e, e_ref) < 2e-2
    S = e.shape[1]
    r = model.llm.prefill_packed(e[0], torch.arange(S, dtype=torch.int32, device="cuda"), None, S, want_all_logits=True, want_layer_hidden=True)
    logits, _, hs = O.qwen2_forward(e.float().cpu(), w, cfg.llm, return_hidden=True)
    for i in range(cfg.llm.num_hidden_layers + 1):
        assert rel_l2(r.layer_hidden[i], hs[i][0]) < 2e-2, f"layer {i} rel={rel_l2(r.layer_hidden[i], hs[i][0]):.3e}"
    assert rel_l2(r.all_logits, logits[0]) < 3e-2, f"logits rel={rel_l2(r.all_logits, logits[0]):.3e}"
    # decode 6 tokens teacher-forced on the oracle's ids
    n = 6
    ids_o, lg_o = O.greedy_generate(e.float().cpu(), w, cfg, n, stop_at_eos=False)
    out, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=ids_o, use_graph=False)
    err = max_abs(lg, lg_o)
    top2 = lg_o.topk(2, -1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 4 * err
    assert rel_l2(lg, lg_o) < 3e-2, f"decode logits rel={rel_l2(lg, lg_o):.3e}"
    assert torch.equal(out[0].cpu()[decisive], ids_o[decisive])


@pytest.mark.parametrize("S", [257, 260, 272])
def test_wide_llm_prefill_tail_rows_via_gemv(wide, S):
    """S = k * 256 + (1..16): the leftover rows of the MLP GEMMs ride in the last 256-row tile as an extra fragment (round 3; rounds 1 / 2:
    through the decode GEMV kernels, still the path under VILA_GEMM_EX=0) — every row must match the oracle, the leftover rows included."""
    cfg, w, model = wide
    g = torch.Generator().manual_seed(S)
    e = (torch.randn(1, S, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16)
    r = model.llm.prefill_packed(e[0].cuda(), torch.arange(S, dtype=torch.int32, device="cuda"), None, S, want_all_logits=True)
    logits, _ = O.qwen2_forward(e.float(), w, cfg.llm)
    assert rel_l2(r.all_logits, logits[0]) < 3e-2, f"all rows rel={rel_l2(r.all_logits, logits[0]):.3e}"
    assert rel_l2(r.all_logits[256:], logits[0, 256:]) < 3e-2, f"tail rows rel={rel_l2(r.all_logits[256:], logits[0, 256:]):.3e}"
