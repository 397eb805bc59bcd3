"""dynamic_s2 multi-scale path.  Fixture tests/golden/dynamic_s2.npz comes from EXECUTING the reference's
merge_chessboard / split_chessboard / merge_features_for_dynamic_s2 (oracle/make_golden_s2.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import vila_oracle as O
from tests.gpu_util import rel_l2
from vila_amd import configs, host, synthetic

BLOCKS = [(2, 3), None]


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "dynamic_s2.npz"))


def test_oracle_s2_matches_reference_bit_exact(fx):
    cfg = configs.tiny_s2()
    x, nbs = O.s2_merge_to_projector_input(torch.from_numpy(fx["tower_out"]), BLOCKS, cfg.s2_scales, -1)
    assert torch.equal(x, torch.from_numpy(fx["proj_in"])) and [list(b) for b in nbs] == fx["new_block_sizes"].tolist()
    xi, _ = O.s2_merge_to_projector_input(torch.from_numpy(fx["int_tiles"]), [(3, 3), None], (8, 16, 24), -1)
    assert torch.equal(xi, torch.from_numpy(fx["int_proj_in"]))
    # round 4: s2_resize_output_to_scale_idx = 0 / 1 (llava_arch.py:340-358), executed from the reference: output grids of 1 x 1 and 2 x 2 blocks
    for r in (0, 1):
        xr, nbs = O.s2_merge_to_projector_input(torch.from_numpy(fx["int_tiles"]), [(3, 3), None], (8, 16, 24), r)
        assert torch.equal(xr, torch.from_numpy(fx[f"int_proj_in_r{r}"])) and [list(b) for b in nbs] == fx[f"int_new_block_sizes_r{r}"].tolist()
    xm, nbs = O.s2_merge_to_projector_input(torch.from_numpy(fx["int_tiles_23"]), [(2, 3), None], (8, 16, 24), 1)
    assert torch.equal(xm, torch.from_numpy(fx["int_proj_in_23_r1"])) and [list(b) for b in nbs] == fx["int_new_block_sizes_23_r1"].tolist()


def test_host_plan_for_every_resize_index():
    """`s2_plan(..., resize_idx)`: the image's OUTPUT blocks are the last scale's bh x bw for -1 / last, s x s for an earlier scale (the
    reference's new_block_sizes); descriptor words 1 / 2 carry both grids; the final chessboard merge (`perms`) covers the output grid."""
    for r, want in ((-1, [(2, 3), (1, 1)]), (2, [(2, 3), (1, 1)]), (1, [(2, 2), (1, 1)]), (0, [(1, 1), (1, 1)])):
        plan = host.s2_plan([(2, 3), None], [8, 16, 24], 4, 2, r)
        assert plan.block_sizes_out == want and plan.n_tiles == 1 + 4 + 6 + 1
        assert plan.n_blocks == sum(a * b for a, b in want) and plan.desc.shape == (plan.n_blocks, 6) and plan.tile_desc.shape == (12, 8)
        d = plan.desc[0].tolist()
        assert (d[1] & 0xffff, d[2] & 0xffff) == (2, 3)
        assert ((d[1] >> 16) or (d[1] & 0xffff), (d[2] >> 16) or (d[2] & 0xffff)) == want[0]
        gd = 2
        assert [int(p.numel()) for p in plan.perms] == [gd * gd * a * b for a, b in want]
        assert sorted(torch.cat(plan.perms).tolist()) == list(range(plan.n_blocks * gd * gd))       # a bijection onto the projector's rows


def test_host_plan_reproduces_reference_token_order(fx):
    cfg = configs.tiny_s2()
    plan = host.s2_plan(BLOCKS, list(cfg.s2_scales), cfg.vision.grid, cfg.downsample)
    assert plan.n_tiles == 12 and plan.n_blocks == 7 and plan.splits == [1, 2]
    w = synthetic.make_weights(cfg, int(fx["seed"]))
    y = O.projector_forward(torch.from_numpy(fx["proj_in"]), w, cfg.mm_projector_type)
    flat = y.reshape(-1, y.shape[-1])
    for i, perm in enumerate(plan.perms):
        ref = torch.from_numpy(fx[f"tokens_{i}"])
        assert rel_l2(flat[perm.long()], ref) < 1e-5


@pytest.mark.gpu
def test_s2_merge_kernel_bit_exact_on_integers(fx):
    from vila_amd import ops
    t = torch.from_numpy(fx["int_tiles"])                       # [15, 16, 2] -> pad channels to 8 for the 16-B kernel path
    t8 = torch.cat([t, torch.zeros(15, 16, 6)], -1).to(torch.bfloat16).cuda()
    plan = host.s2_plan([(3, 3), None], [8, 16, 24], 4, 2)
    out = ops.s2_merge(t8, plan.desc.cuda(), 3, plan.splits)    # [10, 16, 24]
    ref = torch.from_numpy(fx["int_proj_in"])                   # [10, 16, 6] = 3 scales x 2 channels
    got = torch.cat([out[..., 0:2], out[..., 8:10], out[..., 16:18]], -1).float().cpu()
    # integers and halves/quarters of small integers are exact in bf16
    assert torch.equal(got, ref.to(torch.bfloat16).float())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["r0", "r1", "23_r1"])
def test_s2_merge_kernel_bit_exact_for_other_resize_indices(fx, case):
    """The reference-executed integer fixtures with the output at scale 0 / 1 (down-sampling the larger scales, a 2 x 3 last scale pooled to
    2 x 2 blocks): the kernel's adaptive-pool windows and block order must reproduce them exactly."""
    from vila_amd import ops
    tiles, blocks, r = (("int_tiles_23", [(2, 3), None], 1) if case == "23_r1" else ("int_tiles", [(3, 3), None], int(case[1])))
    t = torch.from_numpy(fx[tiles])
    t8 = torch.cat([t, torch.zeros(t.shape[0], 16, 6)], -1).to(torch.bfloat16).cuda()
    plan = host.s2_plan(blocks, [8, 16, 24], 4, 2, r)
    out = ops.s2_merge(t8, plan.desc.cuda(), 3, plan.splits)
    ref = torch.from_numpy(fx[f"int_proj_in_{case}"])
    got = torch.cat([out[..., 0:2], out[..., 8:10], out[..., 16:18]], -1).float().cpu()
    assert got.shape == ref.shape
    # window means of small integers over 1, 2, 3, 4, 6 or 9 cells: the fp32 mean rounded once to bf16 on both sides
    assert torch.equal(got, ref.to(torch.bfloat16).float())


@pytest.mark.gpu
@pytest.mark.parametrize("r", [0, 1])
def test_s2_encode_images_with_another_resize_index_vs_oracle(r):
    """encode_images end to end with cfg.s2_resize_output_to_scale_idx = 0 / 1 (the reference handles every index, llava_arch.py:340-358; the
    shipped recipes use -1): per-image token counts follow the new block sizes, values follow the oracle."""
    from vila_amd.vlm import build_model
    cfg = configs.tiny_s2()
    cfg.s2_resize_output_to_scale_idx = r
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 21).items()}
    model = build_model(cfg, weights=w)
    px = synthetic.make_pixels(cfg, 12, 21).to(torch.bfloat16)
    outs = model.encode_images(px.cuda(), block_sizes=list(BLOCKS))
    refs = O.encode_images_dynamic_s2(px.float(), BLOCKS, w, cfg)
    assert len(outs) == len(refs) == 2
    for o, ref in zip(outs, refs):
        assert o.shape == ref.shape and rel_l2(o, ref) < 2e-2, (tuple(o.shape), tuple(ref.shape))


@pytest.mark.gpu
def test_s2_encode_images_vs_golden_and_oracle(fx):
    from vila_amd.vlm import build_model
    cfg = configs.tiny_s2()
    seed = int(fx["seed"])
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, seed).items()}
    model = build_model(cfg, weights=w)
    from vila_amd import ops
    feats = torch.from_numpy(fx["tower_out"]).to(torch.bfloat16)
    plan = host.s2_plan(BLOCKS, list(cfg.s2_scales), cfg.vision.grid, cfg.downsample)
    x = ops.s2_merge(feats.cuda(), plan.desc.cuda(), 3, plan.splits)
    xr, _ = O.s2_merge_to_projector_input(feats.float(), BLOCKS, cfg.s2_scales, -1)
    assert rel_l2(x, xr) < 2e-3                                  # same bf16 inputs, fp32 window means, one bf16 rounding
    px = synthetic.make_pixels(cfg, 12, seed).to(torch.bfloat16)
    outs = model.encode_images(px.cuda(), block_sizes=list(BLOCKS))
    refs = O.encode_images_dynamic_s2(px.float(), BLOCKS, w, cfg)
    assert len(outs) == 2
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert o.shape == r.shape == fx[f"tokens_{i}"].shape
        assert rel_l2(o, r) < 2e-2, f"image {i} vs oracle {rel_l2(o, r):.3e}"
        assert rel_l2(o, torch.from_numpy(fx[f"tokens_{i}"])) < 3e-2
    with pytest.raises(AssertionError, match="does not match length of image_features"):
        model.encode_images(px[:11].cuda(), block_sizes=list(BLOCKS))


@pytest.mark.gpu
def test_s2_generate_end_to_end():
    from vila_amd.vlm import build_model
    cfg = configs.tiny_s2()
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 12).items()}
    model = build_model(cfg, weights=w)
    px = synthetic.make_pixels(cfg, 14, 12).to(torch.bfloat16)     # square image: 1 + 4 + 9 tiles
    ids = synthetic.make_prompt(cfg, 6, 1, 12)[None]
    media_cfg = {"image": {"block_sizes": [(3, 3)]}}
    # the image encoder stacks `images`; for dynamic_s2 the 14 tiles of ONE image arrive as one [14,3,H,W] entry list
    feats = model.encode_images(px.cuda(), block_sizes=[(3, 3)])
    ref = O.encode_images_dynamic_s2(px.float(), [(3, 3)], w, cfg)[0]
    assert feats.shape == (1, 36, cfg.llm.hidden_size) and rel_l2(feats[0], ref) < 2e-2


# ---------------------------------------------------------------------------------------------------------------------
# backward of the merge (SFT step of the dynamic_s2 recipe)
# ---------------------------------------------------------------------------------------------------------------------
def _merge_bwd_formula(dy, tile_desc, g, C, n_scales, splits):
    """Python restatement of s2_merge_bwd_kernel's index arithmetic (TEST CODE: checks the gather formula against autograd on the CPU,
    the kernel itself is checked against autograd on the GPU below)."""
    n_tiles = tile_desc.shape[0]
    dx = torch.zeros((n_tiles, g * g, C), dtype=torch.float64)
    for t in range(n_tiles):
        blk0, w1, w2, k, ti, tj, single, _ = tile_desc[t].tolist()
        bh, bw = w1 & 0xffff, w2 & 0xffff
        obh, obw = (w1 >> 16) or bh, (w2 >> 16) or bw
        for tok in range(g * g):
            if single:
                dx[t, tok] = dy[blk0, tok].view(n_scales, C).sum(0)
                continue
            sh, sw = (splits[k], splits[k]) if k < n_scales - 1 else (bh, bw)
            Hout, Wout, Hk, Wk = g * obh, g * obw, g * sh, g * sw
            yy, xx = ti * g + tok // g, tj * g + tok % g
            for Y in range((yy * Hout) // Hk, ((yy + 1) * Hout + Hk - 1) // Hk):
                ys, ye = (Y * Hk) // Hout, ((Y + 1) * Hk + Hout - 1) // Hout
                for X in range((xx * Wout) // Wk, ((xx + 1) * Wout + Wk - 1) // Wk):
                    xs, xe = (X * Wk) // Wout, ((X + 1) * Wk + Wout - 1) // Wout
                    b, pos = blk0 + (Y // g) * obw + (X // g), (Y % g) * g + (X % g)
                    dx[t, tok] += dy[b, pos, k * C:(k + 1) * C] / ((ye - ys) * (xe - xs))
    return dx


@pytest.mark.parametrize("blocks,scales,g,r", [([(2, 3), None], [56, 112, 168], 4, -1), ([(3, 3)], [8, 16, 24], 4, -1), ([(1, 2), (1, 1), None, (3, 2)], [8, 16, 24], 2, -1),
                                               ([(1, 1)], [8, 16], 3, -1), ([(2, 3), None], [8, 16, 24], 4, 1), ([(3, 3), (1, 2)], [8, 16, 24], 2, 0),
                                               ([(3, 2), None, (1, 1)], [8, 16, 24], 3, 1)])
def test_s2_merge_backward_formula_is_the_autograd_adjoint(blocks, scales, g, r):
    """Up- and down-sampling scales (a 1x2 image's 2x2 middle scale is LARGER than its output grid), `None` blocks, several images, and
    (round 4) output grids taken from an earlier scale."""
    plan = host.s2_plan(blocks, scales, g, 2, r)
    C = 3
    gen = torch.Generator().manual_seed(1)
    feats = torch.randn(plan.n_tiles, g * g, C, generator=gen, dtype=torch.float64, requires_grad=True)
    x, _ = O.s2_merge_to_projector_input(feats, blocks, scales, r)
    dy = torch.randn(x.shape, generator=gen, dtype=torch.float64)
    (x * dy).sum().backward()
    got = _merge_bwd_formula(dy, plan.tile_desc, g, C, len(scales), plan.splits)
    assert plan.tile_desc.shape == (plan.n_tiles, 8)
    assert float((got - feats.grad).abs().max()) < 5e-6            # the reference interpolates in fp32 (llava_arch.py:341)


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,scales,g,r", [([(2, 3), None], [56, 112, 168], 4, -1), ([(1, 2), (1, 1), None, (3, 2)], [8, 16, 24], 2, -1), ([(3, 3)], [448, 896, 1344], 32, -1),
                                               ([(2, 3), None, (3, 3)], [8, 16, 24], 4, 1), ([(3, 2)], [8, 16, 24], 4, 0)])
def test_s2_merge_backward_kernel_vs_autograd(blocks, scales, g, r):
    from vila_amd import ops
    plan = host.s2_plan(blocks, scales, g, 2, r)
    C = 16 if g < 32 else 1152
    gen = torch.Generator().manual_seed(2)
    feats = torch.randn(plan.n_tiles, g * g, C, generator=gen).to(torch.bfloat16).float().requires_grad_(True)
    x, _ = O.s2_merge_to_projector_input(feats, blocks, scales, r)
    dy = torch.randn(x.shape, generator=gen).to(torch.bfloat16)
    (x * dy.float()).sum().backward()
    got = ops.s2_merge_bwd(dy.cuda(), plan.tile_desc.cuda(), len(scales), plan.splits)
    assert got.shape == feats.shape
    assert rel_l2(got, feats.grad) < 3e-3, f"rel={rel_l2(got, feats.grad):.3e}"        # fp32 sums, one bf16 rounding
