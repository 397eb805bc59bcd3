"""Video encoders: BasicVideoEncoder (llava/model/encoders/video/basic.py:13-53) and TSPVideoEncoder
(video/tsp.py:10-64).  tests/golden/video_encoders.npz holds outputs of the REFERENCE's own `pool` / `_process_features` code
(oracle/make_golden_video.py execs it from source).  CPU: the oracle restatement equals them bit for bit; GPU: the one-launch HIP
assembly (`vila_video_pool_bf16`) is bit-exact on the integer fixture (ordering, windows) and within 2e-2 on values, and the whole
video path (frames -> tower -> projector -> pool -> "\\n" per pooled frame -> splice at <vila/video>) matches the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import vila_oracle as O
from vila_amd import configs, synthetic
from vila_amd.host import splice_plan

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "video_encoders.npz")


def _case(fx, name):
    t = lambda k: torch.from_numpy(fx[f"{name}_{k}"])
    opt = lambda k: (t(k) if t(k).shape[0] else None)
    return t("in"), [tuple(int(v) for v in p) for p in fx[f"{name}_pools"]], opt("start"), opt("end"), opt("sep"), t("out")


@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_oracle_video_features_equal_reference_output(name):
    fx = np.load(GOLDEN)
    x, pools, start, end, sep, ref = _case(fx, name)
    got = O.tsp_process_features(x, pools, start, end, sep)
    assert torch.equal(got, ref)
    if name == "C":
        assert torch.equal(O.video_process_features(x, start, end), ref)


def test_oracle_integer_fixture_and_ragged_split():
    fx = np.load(GOLDEN)
    x = torch.from_numpy(fx["int_in"])
    assert torch.equal(O.tsp_process_features(x, [(8, 2, 2)], None, None), torch.from_numpy(fx["int_out"]))
    with pytest.raises(RuntimeError):                   # the reference's view() rejects 16 frames pooled by 3
        O.tsp_process_features(x, [(3, 1, 1)], None, None)


def test_splice_plan_with_image_and_video_tokens_matches_oracle():
    """Two media names in one batch: each name's blocks are consumed in the order ITS token appears (llava_arch.py:454-466)."""
    cfg = configs.tiny()
    g = torch.Generator().manual_seed(3)
    H = 16
    w = {"llm.model.embed_tokens.weight": torch.randn(cfg.llm.vocab_size, H, generator=g)}
    ids = torch.randint(0, 900, (2, 9), generator=g)
    ids[0, 1] = cfg.video_token_id; ids[0, 5] = cfg.image_token_id
    ids[1, 0] = cfg.image_token_id; ids[1, 3] = cfg.video_token_id
    mask = torch.ones(2, 9, dtype=torch.bool); mask[1, 7:] = False
    labels = torch.randint(0, 900, (2, 9), generator=g)
    media = {"image": [torch.randn(3, H, generator=g), torch.randn(2, H, generator=g)],
             "video": [torch.randn(5, H, generator=g), torch.randn(4, H, generator=g)]}
    e_ref, l_ref, m_ref = O.embed_splice(ids, media, w, cfg, labels=labels, attention_mask=mask)
    tok = {"image": cfg.image_token_id, "video": cfg.video_token_id}
    plan = splice_plan(ids, mask, labels, {n: [int(t.shape[0]) for t in media[n]] for n in tok}, tok)
    flat = torch.cat([t for n in tok for t in media[n]], 0)
    out = torch.zeros(plan.B * plan.S, H)
    out[plan.txt_dst.long()] = w["llm.model.embed_tokens.weight"][plan.txt_src.long()]
    out[plan.img_dst.long()] = flat[plan.img_src.long()]
    assert not plan.img_src_identity
    assert torch.equal(out.view(plan.B, plan.S, H), e_ref) and torch.equal(plan.labels, l_ref) and torch.equal(plan.mask, m_ref)
    with pytest.raises(ValueError, match="Not all video embeddings are consumed!"):
        splice_plan(ids, mask, labels, {"image": [3, 2], "video": [5, 4, 1]}, tok)
    with pytest.raises(IndexError, match="pop from an empty deque"):
        splice_plan(ids, mask, labels, {"image": [3, 2], "video": [5]}, tok)


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_video_pool_kernel_vs_reference_output(name):
    from tests.gpu_util import rel_l2
    from vila_amd import ops
    fx = np.load(GOLDEN)
    x, pools, start, end, sep, ref = _case(fx, name)
    bf = lambda t: None if t is None else t.to(torch.bfloat16).cuda()
    outs = []
    for p in pools:
        outs.append(ops.video_pool(bf(x), p, bf(start), bf(end)))
        if sep is not None:
            outs.append(bf(sep))
    got = torch.cat(outs, 0)
    ref_bf = O.tsp_process_features(x.to(torch.bfloat16).float(), pools, None if start is None else start.to(torch.bfloat16).float(),
                                    None if end is None else end.to(torch.bfloat16).float(), None if sep is None else sep.to(torch.bfloat16).float())
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < 2e-2 and rel_l2(got, ref_bf) < 4e-3, (rel_l2(got, ref), rel_l2(got, ref_bf))


@pytest.mark.gpu
def test_video_pool_kernel_integer_fixture_bit_exact_and_errors():
    from vila_amd import ops
    fx = np.load(GOLDEN)
    x = torch.from_numpy(fx["int_in"]).to(torch.bfloat16).cuda()
    got = ops.video_pool(x, (8, 2, 2))
    assert torch.equal(got.float().cpu(), torch.from_numpy(fx["int_out"]))
    with pytest.raises(ValueError, match="invalid for pooling"):
        ops.video_pool(x, (3, 1, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("enc", ["basic", "tsp"])
def test_video_path_end_to_end_vs_oracle(enc):
    """frames -> tower -> projector -> (pool) -> "\\n" per (pooled) frame -> spliced at <vila/video>, next to an <image> of the same batch."""
    from tests.gpu_util import rel_l2
    from vila_amd.vlm import TSPVideoEncoder, build_model
    cfg = configs.tiny("mlp_downsample_2x2_fix", image=56)
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 8).items()}
    model = build_model(cfg, weights=w)
    px = synthetic.make_pixels(cfg, 9, 8).to(torch.bfloat16)
    video, image = px[:8], px[8]
    pools = [(4, 1, 1)]
    if enc == "tsp":
        model.encoders["video"] = TSPVideoEncoder(model, pools)
        ref_v = O.tsp_video_encoder([video.float()], w, cfg, pools)
    else:
        ref_v = O.basic_video_encoder([video.float()], w, cfg)
    got_v = model.encoders["video"]([video.cuda()], {})
    assert len(got_v) == 1 and got_v[0].shape == ref_v[0].shape
    assert got_v[0].shape[0] == (2 if enc == "tsp" else 8) * (cfg.tokens_per_tile + 1)
    assert rel_l2(got_v[0], ref_v[0]) < 2e-2, f"video tokens rel={rel_l2(got_v[0], ref_v[0]):.3e}"
    ids = synthetic.make_prompt(cfg, 6, 2, 8)
    ids[0] = cfg.image_token_id; ids[1] = cfg.video_token_id
    ref_i = O.basic_image_encoder([image.float()], w, cfg)
    e_ref, _, m_ref = O.embed_splice(ids[None], {"image": ref_i, "video": ref_v}, w, cfg)
    e, _, m = model._embed(ids[None], {"video": [video.cuda()], "image": [image.cuda()]})
    assert e.shape == e_ref.shape and torch.equal(m.cpu(), m_ref)
    assert rel_l2(e, e_ref) < 2e-2, f"spliced embeds rel={rel_l2(e, e_ref):.3e}"


@pytest.mark.parametrize("pool_sizes,start,end,sep", [
    ([[1, 1, 1]], [], [11], []),
    ([[2, 1, 1]], [], [11], []),
    ([[2, 2, 2], [1, 1, 1]], [21, 22], [23], [24, 25]),
    ([[4, 1, 2], [2, 2, 1]], [], [], [24]),
])
def test_sft_media_block_table_reproduces_the_encoder_blocks(pool_sizes, start, end, sep):
    """The SFT step's integer plan of the media blocks (`SFTTrainer._media_blocks`: rows of the media feature buffer / token rows) expands to
    exactly the blocks the oracle's encoders build (basic.py:30-41, tsp.py:28-52) — host logic, no GPU."""
    from types import SimpleNamespace
    from vila_amd.train import SFTTrainer
    cfg = configs.tiny("mlp_downsample")
    Tm, H = cfg.tokens_per_tile, 6
    nl = int(Tm ** 0.5)
    g = torch.Generator().manual_seed(3)
    frames = [4, 8]
    n_img = 2
    n_tiles = n_img + sum(frames)
    proj = torch.randn(n_tiles, Tm, H, generator=g)
    table = torch.randn(64, H, generator=g)
    fake = SimpleNamespace(cfg=cfg, _video_tokens=lambda: (tuple(tuple(p) for p in pool_sizes), start, end, sep))
    rows = [torch.arange(i * Tm, (i + 1) * Tm) for i in range(n_tiles)]
    img_blocks, vid_blocks, pools, n_buf = SFTTrainer._media_blocks(fake, rows, frames, n_tiles * Tm)
    # the buffer the driver builds: projector rows, then the pooled rows of every (video, pool size) in `pools` order
    buf = [proj.reshape(-1, H)]
    for t0, nf, pool, off, cnt in pools:
        assert off == sum(b.shape[0] for b in buf)
        f = proj[t0:t0 + nf].view(nf, nl, nl, H)
        for dim, p in enumerate(pool):
            f = O.pool(f, p, dim)
        buf.append(f.reshape(-1, H))
        assert buf[-1].shape[0] == cnt
    buf = torch.cat(buf, 0)
    assert buf.shape[0] == n_buf
    expand = lambda blk: torch.stack([buf[v] if v >= 0 else table[-1 - v] for v in blk.tolist()], 0) if blk.numel() else torch.empty(0, H)
    emb = lambda ids: table[torch.tensor(ids)] if ids else None
    for i in range(n_img):
        assert torch.equal(expand(img_blocks[i]), torch.cat([proj[i], table[cfg.newline_token_id][None]], 0))
    t0 = n_img
    for v, nf in enumerate(frames):
        want = O.tsp_process_features(proj[t0:t0 + nf], pool_sizes, emb(start), emb(end), emb(sep))
        got = expand(vid_blocks[v])
        assert got.shape == want.shape and torch.allclose(got, want, atol=1e-6), (v, got.shape, want.shape)
        t0 += nf
    with pytest.raises(ValueError, match="invalid for pooling"):
        bad = SimpleNamespace(cfg=cfg, _video_tokens=lambda: (((3, 1, 1),), [], [11], []))
        SFTTrainer._media_blocks(bad, rows, frames, n_tiles * Tm)
