"""GPU parity at the shapes of the BASELINE configs that had no test at their real widths:

  configs[0]  NVILA-Lite-3B widths (hidden 2048, 16 / 2 heads -> GQA ratio 8, FFN 11008, tied head, 3x3 projector with its 10 368-wide
              LayerNorm), few layers: tower -> projector -> prefill hidden states / logits -> 6 teacher-forced decode steps vs the oracle
  configs[3]  the 64-frame video prompt: causal hd-128 GQA 28 / 4 attention at S = 16 480 against an fp32 reference on sampled query rows,
              and a 64-image tower batch against per-image calls
plus the 32-rows-per-wave variants of the forward attention kernel (taken when the grid has >= 2 rounds of 256-row blocks).
"""
import pytest
import torch

from oracle import vila_oracle as O
from tests.gpu_util import max_abs, randn_bf16, rel_l2
from vila_amd import configs, synthetic

pytestmark = pytest.mark.gpu


def _bf16_weights(cfg, seed):
    return {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, seed).items()}


# ---------------------------------------------------------------------------------------------------------------------
# configs[0]: NVILA-Lite-3B widths
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def lite():
    from vila_amd.vlm import build_model
    cfg = configs.nvila_lite_3b()
    cfg.vision.num_hidden_layers = 3                 # hidden_states[-2] of 3 layers = 2 layers run
    cfg.llm.num_hidden_layers = 2
    cfg.llm.vocab_size = 32000
    cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
    w = _bf16_weights(cfg, 11)
    model = build_model(cfg, weights=w)
    return cfg, w, model


def test_lite3b_widths_tower_projector_prefill_decode(lite):
    cfg, w, model = lite
    assert cfg.llm.hidden_size == 2048 and cfg.llm.num_attention_heads // cfg.llm.num_key_value_heads == 8 and cfg.llm.tie_word_embeddings
    px = synthetic.make_pixels(cfg, 1, 11).to(torch.bfloat16)
    feat = model.vision_tower(px.cuda())
    ref = O.vision_tower_forward(px.float(), w, cfg.vision)
    assert rel_l2(feat, ref) < 2e-2, f"vit rel={rel_l2(feat, ref):.3e}"
    out = model.mm_projector(feat)
    refp = O.projector_forward(ref, w, cfg.mm_projector_type)
    assert out.shape == (1, 121, 2048)                               # 32 -> 33 (zero pad) -> 11 x 11, K = 9 * 1152 = 10 368
    assert rel_l2(out, refp) < 2e-2, f"proj rel={rel_l2(out, refp):.3e}"

    ids = synthetic.make_prompt(cfg, 32, 1, 11)[None]                # BASELINE configs[0]: 1 image + 32-token prompt
    e, _, _ = model._embed(ids, {"image": [px[0].cuda()]})
    assert e.shape == (1, 122 + 32, 2048)
    e_ref, _ = O.vlm_prefill_embeds([px[0].float()], ids[0], w, cfg)
    assert rel_l2(e, e_ref) < 2e-2
    S = e.shape[1]
    r = model.llm.prefill_packed(e[0], torch.arange(S, dtype=torch.int32, device="cuda"), None, S, want_all_logits=True, want_layer_hidden=True)
    logits, _, hs = O.qwen2_forward(e.float().cpu(), w, cfg.llm, return_hidden=True)
    for i in range(cfg.llm.num_hidden_layers + 1):
        assert rel_l2(r.layer_hidden[i], hs[i][0]) < 2e-2, f"layer {i} rel={rel_l2(r.layer_hidden[i], hs[i][0]):.3e}"
    assert rel_l2(r.all_logits, logits[0]) < 3e-2, f"logits rel={rel_l2(r.all_logits, logits[0]):.3e}"
    n = 6
    ids_o, lg_o = O.greedy_generate(e.float().cpu(), w, cfg, n, stop_at_eos=False)
    out, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=ids_o, use_graph=False)
    err = max_abs(lg, lg_o)
    top2 = lg_o.topk(2, -1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 4 * err
    assert rel_l2(lg, lg_o) < 3e-2, f"decode logits rel={rel_l2(lg, lg_o):.3e}"
    assert bool(decisive.any()), f"no decisive step: err {err:.3e} margins {(top2[:, 0] - top2[:, 1]).tolist()}"
    assert torch.equal(out[0].cpu()[decisive], ids_o[decisive])
    # the captured graph must produce the same ids as the eager steps (GQA ratio 8 decode attention, K = 2048 / 11008 GEMVs)
    free_e = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=False, eos_token_id=-1)
    free_g = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)
    assert torch.equal(free_e, free_g)


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: long causal sequence, 64-image tower batch
# ---------------------------------------------------------------------------------------------------------------------
def _rows_ref(q, k, v, rows, causal):
    """fp32 attention output for the sampled query rows only: [len(rows), Hq, D]."""
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    G = Hq // Hkv
    kf, vf = k.float(), v.float()
    out = torch.empty((len(rows), Hq, D), device=q.device, dtype=torch.float32)
    for i, r in enumerate(rows):
        n = r + 1 if causal else T
        qr = q[r].float().view(Hkv, G, D)
        s = torch.einsum("hgd,khd->hgk", qr, kf[:n]) * D ** -0.5
        p = torch.softmax(s, -1)
        out[i] = torch.einsum("hgk,khd->hgd", p, vf[:n]).reshape(Hq, D)
    return out


def test_attention_video_prompt_16480_tokens_sampled_rows(ops_mod):
    """BASELINE configs[3] layout (i): 64 x 257 image tokens + 32 text tokens = 16 480, causal GQA 28 / 4, hd 128.  fp32 reference on
    sampled query rows (first / last rows, tile and block boundaries, random rows); tolerance as every attention test: rel-L2 <= 8e-3."""
    T, Hq, Hkv, D = 16480, 28, 4, 128
    qkv = randn_bf16(T, (Hq + 2 * Hkv) * D, seed=91)
    q = qkv[:, : Hq * D].view(T, Hq, D)
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    out, lse = ops_mod.attn_fwd(q, k, v, True, return_lse=True)
    g = torch.Generator().manual_seed(5)
    rows = sorted(set([0, 1, 15, 16, 63, 64, 127, 128, 255, 256, 257, 511, 512, 8191, 8192, 16383, 16384, T - 33, T - 2, T - 1] +
                      torch.randint(0, T, (44,), generator=g).tolist()))
    ref = _rows_ref(q, k, v, rows, True)
    got = out[rows].float()
    assert rel_l2(got, ref) < 8e-3, f"rel={rel_l2(got, ref):.3e} max={max_abs(got, ref):.3e}"
    worst = max(rel_l2(got[i], ref[i]) for i in range(len(rows)))
    assert worst < 2e-2, f"worst sampled row rel={worst:.3e}"
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()


@pytest.mark.parametrize("T,Hq,Hkv,D,causal,nseq", [
    (8192, 16, 16, 72, False, 8),        # 8 images x 1024 tokens: 4 x 16 x 8 = 512 blocks of 256 rows -> 32 rows per wave
    (4700, 28, 4, 128, True, 1),         # 19 x 28 = 532 blocks, causal, ragged last block
    (2400, 48, 8, 64, True, 4),          # hd 64, 4 sequences of 600
])
def test_attention_forward_32_rows_per_wave(ops_mod, T, Hq, Hkv, D, causal, nseq):
    q, k, v = randn_bf16(T, Hq, D, seed=31), randn_bf16(T, Hkv, D, seed=32), randn_bf16(T, Hkv, D, seed=33)
    n = T // nseq
    cu = torch.arange(0, T + 1, n, dtype=torch.int32, device="cuda")
    out = ops_mod.attn_fwd(q, k, v, causal, n_seq=nseq) if not causal else ops_mod.attn_fwd(q, k, v, causal, cu_seqlens=cu, max_seqlen=n)
    g = torch.Generator().manual_seed(6)
    for s in range(0, nseq, max(1, nseq // 3)):
        a = s * n
        rows = sorted(set([0, 1, n // 2, n - 1] + torch.randint(0, n, (12,), generator=g).tolist()))
        ref = _rows_ref(q[a:a + n], k[a:a + n], v[a:a + n], rows, causal)
        got = out[a:a + n][rows].float()
        assert rel_l2(got, ref) < 8e-3, f"seq {s} rel={rel_l2(got, ref):.3e}"


@pytest.fixture(scope="module")
def ops_mod():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from vila_amd import _lib, ops as _ops
    _lib.load()
    return _ops


def test_tower_batch_of_64_frames_equals_per_image_calls():
    """The 64-frame batch of BASELINE configs[3] through the tower at its real widths (4 layers): the batched launch (one patch-embed
    GEMM, 65 536-row GEMMs, 32-rows-per-wave attention) must agree with one-image calls, which take other tile shapes and kernels."""
    from vila_amd.vlm import build_model
    cfg = configs.reduced_8b(layers_v=5, layers_l=1, vocab=1024)
    cfg.image_token_id, cfg.llm.eos_token_id = 1023, 1022
    model = build_model(cfg, seed=3)
    px = synthetic.make_pixels(cfg, 64, 3).to(torch.bfloat16).cuda()
    batch = model.vision_tower(px)
    assert batch.shape == (64, 1024, 1152) and torch.isfinite(batch.float()).all()
    for i in (0, 17, 63):
        one = model.vision_tower(px[i:i + 1])
        assert rel_l2(batch[i], one[0]) < 6e-3, f"image {i}: rel={rel_l2(batch[i], one[0]):.3e}"
    proj = model.mm_projector(batch)
    one = model.mm_projector(batch[5:6])
    assert rel_l2(proj[5], one[0]) < 6e-3
