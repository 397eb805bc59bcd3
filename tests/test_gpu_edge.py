"""GPU edge cases of the hot path: text-only prompts, batched ragged prefill with per-slot KV cache, long context across
many decode-attention splits, multi-tile (video-style) image batches, error behaviour (reference messages)."""
import pytest
import torch

from oracle import vila_oracle as O
from tests.gpu_util import margin_aware_ids, rel_l2
from vila_amd import configs, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 5).items()}
    return cfg, w, build_model(cfg, weights=w)


def test_text_only_prompt_generates(tiny):
    cfg, w, model = tiny
    ids = synthetic.make_prompt(cfg, 10, 0, 5)[None]
    e, labels, mask = model._embed(ids, {})
    assert e.shape == (1, 10, cfg.llm.hidden_size) and bool(mask.all())
    assert torch.equal(e[0].float().cpu(), w["llm.model.embed_tokens.weight"][ids[0]])
    out = model.generate(input_ids=ids, media={}, max_new_tokens=4, eos_token_id=-1)
    ids_o, lg_o = O.vlm_generate([], ids[0], w, cfg, 4, stop_at_eos=False)
    assert out.shape == (1, 4)
    _, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=4, return_logits=True, forced_ids=ids_o, use_graph=False)
    margin_aware_ids(lg, lg_o, ids_o, free_ids=out[0])


def test_splice_errors_use_reference_messages(tiny):
    cfg, w, model = tiny
    px = synthetic.make_pixels(cfg, 2, 5).to(torch.bfloat16).cuda()
    ids = synthetic.make_prompt(cfg, 6, 1, 5)[None]
    with pytest.raises(ValueError, match="Not all image embeddings are consumed!"):
        model._embed(ids, {"image": [px[0], px[1]]})
    with pytest.raises(IndexError):
        model._embed(torch.cat([ids, ids], 1), {"image": [px[0]]})
    bad = torch.zeros(1, 3, 28, 28, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="doesn't match model"):
        model.vision_tower(bad)


def test_ragged_batch_prefill_with_slots_matches_per_sample(tiny):
    """3 sequences of different lengths packed into one stream (cu_seqlens) write K/V into their own cache slots; logits equal
    the single-sequence runs and the oracle."""
    cfg, w, model = tiny
    llm = model.llm
    g = torch.Generator().manual_seed(6)
    lens = [17, 5, 40]
    embs = [(torch.randn(n, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16) for n in lens]
    packed = torch.cat(embs, 0).cuda()
    cu = torch.tensor([0, 17, 22, 62], dtype=torch.int32, device="cuda")
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32).cuda()
    seq = torch.cat([torch.full((n,), i) for i, n in enumerate(lens)]).to(torch.int32).cuda()
    cache = llm.new_cache(64, n_slots=3)
    last = torch.tensor([16, 21, 61], dtype=torch.int32, device="cuda")
    r = llm.prefill_packed(packed, pos, cu, 40, cache=cache, seq_of_tok=seq, last_rows=last, want_all_logits=True)
    for i, e in enumerate(embs):
        lg, past = O.qwen2_forward(e.float()[None], w, cfg.llm)
        a, b = int(cu[i]), int(cu[i + 1])
        assert rel_l2(r.all_logits[a:b], lg[0]) < 3e-2
        assert rel_l2(r.last_logits[i], lg[0, -1]) < 3e-2
        # the cache slot holds this sequence's rotated keys of layer 0
        k_ref = past[0][0][0]                                       # [kv_heads, S, hd]
        k_got = cache.k[0, i, :, : lens[i]].float().cpu()
        assert rel_l2(k_got, k_ref) < 2e-2


def test_long_context_decode_crosses_many_splits(tiny):
    """600-token prompt: the decode attention runs 10+ KV splits; teacher-forced logits must match the prefill kernels."""
    cfg, w, model = tiny
    llm = model.llm
    g = torch.Generator().manual_seed(7)
    S, n = 600, 70                                                   # crosses the 64-key split boundary at 640
    e = (torch.randn(1, S, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16).cuda()
    forced = torch.randint(0, 900, (n,), generator=g)
    out, lg = llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=forced, use_graph=False)
    full = torch.cat([e[0], llm.embed_tokens(forced[: n - 1].cuda())], 0)
    T = full.shape[0]
    r = llm.prefill_packed(full, torch.arange(T, dtype=torch.int32, device="cuda"), None, T, want_all_logits=True)
    assert rel_l2(lg, r.all_logits[S - 1:]) < 1.5e-2, f"rel={rel_l2(lg, r.all_logits[S - 1:]):.3e}"
    # and the graph-replayed free-running decode equals the eager one over the boundary
    a = llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=False, eos_token_id=-1)
    b = llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)
    assert torch.equal(a, b)


def test_split_kv_decode_path_for_large_caches(tiny):
    """Caches larger than 2048 positions use the split-KV + merge kernels instead of the single-launch per-head kernel:
    both must reproduce the prefill logits."""
    cfg, w, model = tiny
    llm = model.llm
    g = torch.Generator().manual_seed(8)
    S, n = 150, 6
    e = (torch.randn(1, S, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16).cuda()
    forced = torch.randint(0, 900, (n,), generator=g)
    full = torch.cat([e[0], llm.embed_tokens(forced[: n - 1].cuda())], 0)
    T = full.shape[0]
    ref = llm.prefill_packed(full, torch.arange(T, dtype=torch.int32, device="cuda"), None, T, want_all_logits=True).all_logits[S - 1:]
    for max_ctx in (256, 2304):
        cache = llm.new_cache(max_ctx)
        _, lg = llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=forced, use_graph=False, cache=cache)
        assert rel_l2(lg, ref) < 1.5e-2, f"max_ctx={max_ctx} rel={rel_l2(lg, ref):.3e}"


def test_generate_stops_after_eos(tiny):
    cfg, w, model = tiny
    ids = synthetic.make_prompt(cfg, 8, 0, 9)[None]
    free = model.generate(input_ids=ids, media={}, max_new_tokens=12, eos_token_id=-1)
    eos = int(free[0, 3])
    out = model.generate(input_ids=ids, media={}, max_new_tokens=12, eos_token_id=eos)
    first = free[0].tolist().index(eos)
    assert out[0].tolist() == free[0, : first + 1].tolist()        # HF: the eos token itself is emitted, nothing after it


def test_kv_cache_too_small_is_an_error(tiny):
    cfg, w, model = tiny
    e = torch.zeros(1, 20, cfg.llm.hidden_size, device="cuda", dtype=torch.bfloat16)
    cache = model.llm.new_cache(16)
    with pytest.raises(ValueError, match="KV cache too small"):
        model.llm.generate(inputs_embeds=e, max_new_tokens=4, cache=cache)


def test_multi_tile_batch_like_video_frames():
    """8 tiles through the tower + projector in one call (the per-frame path of a video prompt, utils/media.py:114-119)."""
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample_2x2_fix", image=70)
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 8).items()}
    model = build_model(cfg, weights=w)
    px = synthetic.make_pixels(cfg, 8, 8).to(torch.bfloat16)
    feats = model.encode_images(px.cuda())
    ref = O.encode_images(px.float(), w, cfg)
    assert feats.shape == ref.shape == (8, 9, cfg.llm.hidden_size)
    assert rel_l2(feats, ref) < 2e-2
    ids = torch.cat([torch.full((8,), cfg.image_token_id), synthetic.make_prompt(cfg, 5, 0, 8)])[None]
    e, _, _ = model._embed(ids, {"image": [p.cuda() for p in px]})
    e_ref, _ = O.vlm_prefill_embeds([p.float() for p in px], ids[0], w, cfg)
    assert e.shape == e_ref.shape and rel_l2(e, e_ref) < 2e-2
