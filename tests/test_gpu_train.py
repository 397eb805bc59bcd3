"""GPU parity tests of the SFT step: backward operators against PyTorch fp32 autograd of the same
op, and the whole forward+backward (ViT + projector + packed LLM + loss) against autograd through the CPU oracle.
Tolerances: operator gradients rel-L2 <= 1.5e-2 (bf16 in/out); model gradients cosine >= 0.999 (SURVEY 8c's stated bound,
`gpu_util.GRAD_COS_MIN`; per-tensor exceptions are listed in GRAD_COS_EXCEPTIONS with the measured value and the reason) and
rel-L2 <= 6e-2 per tensor (bf16 GPU vs fp32 CPU through ~10 layers), loss |delta| <= 1e-2 relative."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import GRAD_COS_MIN, grad_cos, max_abs, randn_bf16, rel_l2

# Per-tensor exceptions to the stated cosine bound: (test tag, tensor-name suffix) -> (bound, measured, why).  Everything else: >= 0.999.
GRAD_COS_EXCEPTIONS = {}


def _cos_bound(test: str, name: str) -> float:
    for (t, suffix), (bound, _measured, _why) in GRAD_COS_EXCEPTIONS.items():
        if t == test and name.endswith(suffix):
            return bound
    return GRAD_COS_MIN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from vila_amd import _lib, ops as _ops
    _lib.load()
    return _ops


@pytest.mark.parametrize("R,C", [(769, 3584), (64, 64), (3076, 512), (13, 24), (1025, 4304)])
def test_transpose_pads_rows(ops, R, C):
    x = randn_bf16(R, C, seed=1)
    t = ops.transpose(x)
    Rp = (R + 63) // 64 * 64
    assert t.shape == (C, Rp)
    assert torch.equal(t[:, :R], x.t())
    assert float(t[:, R:].float().abs().sum()) == 0.0


def test_linear_backward_via_transposed_gemm(ops):
    from vila_amd.train import linear_bwd
    M, N, K = 771, 264, 136
    x, w, dy = randn_bf16(M, K, seed=2), randn_bf16(N, K, seed=3, scale=K ** -0.5), randn_bf16(M, N, seed=4)
    gw = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    gb = torch.empty(N, device="cuda", dtype=torch.bfloat16)
    dx = linear_bwd(x, w, dy, gw, gb)
    assert rel_l2(dx, dy.float() @ w.float()) < 5e-3
    assert rel_l2(gw, dy.float().t() @ x.float()) < 5e-3
    assert rel_l2(gb, dy.float().sum(0)) < 5e-3


def _attn_ref_grads(q, k, v, do, causal, cu):
    qf, kf, vf = [t.float().detach().requires_grad_(True) for t in (q, k, v)]
    T, Hq, D = q.shape
    G = Hq // k.shape[1]
    outs = []
    bounds = cu.tolist()
    for a, b in zip(bounds[:-1], bounds[1:]):
        s = torch.einsum("qhd,khd->hqk", qf[a:b], kf[a:b].repeat_interleave(G, 1)) * D ** -0.5
        if causal:
            n = b - a
            s = s.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool, device=q.device), 1), float("-inf"))
        outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vf[a:b].repeat_interleave(G, 1)))
    o = torch.cat(outs, 0)
    o.backward(do.float())
    return o.detach(), qf.grad, kf.grad, vf.grad


@pytest.mark.parametrize("bounds,Hq,Hkv,D,causal", [
    ([0, 1024], 16, 16, 72, False), ([0, 196, 392], 2, 2, 72, False), ([0, 50], 2, 2, 72, False),
    ([0, 769], 28, 4, 128, True), ([0, 100, 357, 400, 401], 4, 2, 128, True), ([0, 130], 4, 4, 64, True), ([0, 97], 2, 1, 128, False),
    # grids above one block per CU take the 4-wave dK/dV kernel (the cases above the 8-wave two-group one)
    ([0, 1024, 2048], 16, 16, 72, False), ([0, 1300, 2600], 8, 8, 128, True),
])
def test_attention_backward(ops, bounds, Hq, Hkv, D, causal):
    T = bounds[-1]
    uniform = len(set(b - a for a, b in zip(bounds[:-1], bounds[1:]))) == 1
    cu = torch.tensor(bounds, dtype=torch.int32, device="cuda")
    q, k, v, do = randn_bf16(T, Hq, D, seed=5), randn_bf16(T, Hkv, D, seed=6), randn_bf16(T, Hkv, D, seed=7), randn_bf16(T, Hq, D, seed=8)
    o_ref, dq_ref, dk_ref, dv_ref = _attn_ref_grads(q, k, v, do, causal, cu)
    mx = max(b - a for a, b in zip(bounds[:-1], bounds[1:]))
    kw = dict(n_seq=len(bounds) - 1) if (uniform and not causal) else dict(cu_seqlens=cu, max_seqlen=mx)
    o, lse = ops.attn_fwd(q, k, v, causal, return_lse=True, **kw)
    assert rel_l2(o, o_ref) < 8e-3
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attn_bwd(q, k, v, o, do, lse, causal, dq, dk, dv, **kw)
    for name, got, ref in (("dq", dq, dq_ref), ("dk", dk, dk_ref), ("dv", dv, dv_ref)):
        assert rel_l2(got, ref) < 1.5e-2, f"{name} rel={rel_l2(got, ref):.3e}"


@pytest.mark.parametrize("rms", [False, True])
@pytest.mark.parametrize("rows,cols", [(300, 1152), (64, 3584), (10, 144), (3076, 3584), (37, 4096), (5, 4608), (13, 100)])
def test_norm_backward(ops, rms, rows, cols):
    x = randn_bf16(rows, cols, seed=9, scale=1.5) + 0.3
    w = randn_bf16(cols, seed=10, scale=0.1) + 1
    b = randn_bf16(cols, seed=11, scale=0.1)
    dy = randn_bf16(rows, cols, seed=12)
    xf, wf, bf = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    if rms:
        y = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    else:
        y = F.layer_norm(xf, (cols,), wf, bf, 1e-6)
    y.backward(dy.float())
    dw = torch.zeros(cols, device="cuda", dtype=torch.bfloat16)
    db = torch.zeros(cols, device="cuda", dtype=torch.bfloat16)
    dx = ops.norm_bwd(x, w, dy, dw, None if rms else db, 1e-6, rms)
    assert rel_l2(dx, xf.grad) < 1e-2, f"dx rel={rel_l2(dx, xf.grad):.3e}"
    assert rel_l2(dw, wf.grad) < 1e-2, f"dw rel={rel_l2(dw, wf.grad):.3e}"
    if not rms:
        assert rel_l2(db, bf.grad) < 1e-2


def test_activation_and_swiglu_backward(ops):
    z, dy = randn_bf16(500, 272, seed=13, scale=2), randn_bf16(500, 272, seed=14)
    for act, fn in ((1, lambda t: F.gelu(t, approximate="tanh")), (2, F.gelu)):
        zf = z.float().requires_grad_(True)
        y = fn(zf)
        y.backward(dy.float())
        assert rel_l2(ops.act_fwd(z, act), y) < 4e-3
        assert rel_l2(ops.act_bwd(z, dy, act), zf.grad) < 6e-3
    g, u = randn_bf16(300, 1088, seed=15, scale=2), randn_bf16(300, 1088, seed=16)
    da = randn_bf16(300, 1088, seed=17)
    gf, uf = g.float().requires_grad_(True), u.float().requires_grad_(True)
    a = F.silu(gf) * uf
    a.backward(da.float())
    assert rel_l2(ops.silu_mul(g, u), a) < 6e-3
    dg, du = ops.silu_mul_bwd(g, u, da)
    assert rel_l2(dg, gf.grad) < 6e-3 and rel_l2(du, uf.grad) < 6e-3


def test_cross_entropy_sum_over_items(ops):
    n, V = 37, 1000
    g = torch.Generator().manual_seed(18)
    logits = (torch.randn(n, V, generator=g) * 3).cuda()
    labels = torch.randint(0, V, (n,), generator=g).cuda()
    lf = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(lf, labels, reduction="sum") / 29.0
    ref.backward()
    loss = torch.zeros(1, device="cuda")
    d = ops.ce_loss(logits, labels, loss, 1.0 / 29.0)
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
    assert rel_l2(d, lf.grad) < 6e-3


def test_scatter_add_rows_with_duplicates(ops):
    src = randn_bf16(50, 512, seed=19)
    rows = torch.tensor([3, 7, 3, 3, 999] * 10, dtype=torch.int32, device="cuda")
    dst = torch.zeros(1000, 512, device="cuda", dtype=torch.bfloat16)
    ops.scatter_add_rows(src, dst, rows)
    ref = torch.zeros(1000, 512, device="cuda").index_add_(0, rows.long(), src.float())
    assert rel_l2(dst, ref) < 1e-2


def test_depth_to_space_is_adjoint_of_space_to_depth(ops):
    for g_, k in ((5, 2), (32, 2), (7, 3), (32, 3)):
        x = randn_bf16(2, g_ * g_, 16, seed=20)
        y = ops.space_to_depth(x, k)
        dy = randn_bf16(*y.shape, seed=21)
        dx = ops.depth_to_space(dy, g_, k)
        # <S x, dy> == <x, S^T dy>
        lhs = float((y.float() * dy.float()).sum())
        rhs = float((x.float() * dx.float()).sum())
        assert abs(lhs - rhs) < 2e-2 * (abs(lhs) + 1)


@pytest.mark.parametrize("n", [4096 + 24, 4096 + 27, 3])
def test_adamw_matches_torch(ops, n):
    p0 = torch.randn(n, device="cuda")
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    master, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    param = p0.to(torch.bfloat16)
    for step in range(1, 4):
        g = torch.randn(n, device="cuda").to(torch.bfloat16)
        ref_p.grad = g.float()
        opt.step()
        ops.adamw_step(master, m, v, g, param, 1e-2, 0.9, 0.999, 1e-8, 0.1, step)
    assert max_abs(master, ref_p.detach()) < 1e-5
    assert torch.equal(param, master.to(torch.bfloat16))
    # the lean (<= 32 VGPR, buffer-addressed) kernel computes the same update bit for bit
    master2, m2, v2, param2 = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), p0.to(torch.bfloat16)
    master1, m1, v1, param1 = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), p0.to(torch.bfloat16)
    for step in range(1, 4):
        g = torch.randn(n, device="cuda").to(torch.bfloat16)
        ops.adamw_step(master1, m1, v1, g, param1, 1e-2, 0.9, 0.999, 1e-8, 0.1, step)
        ops.adamw_step(master2, m2, v2, g, param2, 1e-2, 0.9, 0.999, 1e-8, 0.1, step, lean=True)
    for a, b in ((master1, master2), (m1, m2), (v1, v2), (param1, param2)):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------
# whole step vs autograd through the CPU oracle
# ---------------------------------------------------------------------------------------------------------------------
def _sft_vs_oracle(cfg, seed, ids, labels, mask, n_images, tag="sft", rel_max=6e-2, c_abi=(False,), block_sizes=None, video_frames=(), tsp=None):
    """One forward+backward of the HIP trainer vs fp32 autograd through the restated reference forward (packed branch of
    llava_llama.py:125-134): loss <= 1e-2 relative, every gradient tensor cosine >= 0.999 (`_cos_bound`) and rel-L2 <= rel_max.
    c_abi: which drivers to check against the ONE oracle run — False = the Python-orchestrated operator calls, True = the whole
    forward + backward as one `vila_sft_fwd_bwd` call with the grad-ready callback.
    video_frames: frame counts of the videos (one per <vila/video> token), their frames are drawn behind the `n_images` image tiles;
    tsp: pool_sizes of a TSPVideoEncoder (None = the BasicVideoEncoder)."""
    from oracle import vila_oracle as O
    from vila_amd import synthetic
    from vila_amd.train import SFTTrainer, count_targets
    from vila_amd.vlm import build_model, TSPVideoEncoder
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, seed).items()}
    px_all = synthetic.make_pixels(cfg, n_images + sum(video_frames), seed).to(torch.bfloat16)
    px, vids, at = px_all[:n_images], [], n_images
    for nf in video_frames:
        vids.append(px_all[at:at + nf]); at += nf
    n_items = count_targets(ids, labels, mask, (cfg.image_token_id, cfg.video_token_id))
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    ref = O.vlm_sft_loss([p.float() for p in px], ids, labels, mask, wr, cfg, num_items_in_batch=n_items, packed=True, block_sizes=block_sizes,
                         videos=[v.float() for v in vids], video_encoder=None if tsp is None else {"pool_sizes": tsp})
    ref.backward()
    out = []
    for use_c in c_abi:
        model = build_model(cfg, weights=w)
        if tsp is not None:
            model.encoders["video"] = TSPVideoEncoder(model, tsp)
        tr = SFTTrainer(model, optimizer_state=False)
        fb = tr.forward_backward_c if use_c else tr.forward_backward
        loss = fb(ids, [p.cuda() for p in px], labels, mask, n_items, block_sizes, **({"videos": [v.cuda() for v in vids]} if vids else {}))
        torch.cuda.synchronize()
        assert abs(float(loss) - float(ref)) < 1e-2 * abs(float(ref)), (use_c, float(loss), float(ref))
        grads = tr.flat.named_grads()
        bad, worst = [], (1.0, 0.0)
        for name, gref in ((k, v.grad) for k, v in wr.items()):
            if name not in grads or gref is None:
                continue
            got = grads[name].float().cpu()
            if float(gref.norm()) < 1e-6:
                assert float(got.norm()) < 1e-3, name
                continue
            cos = grad_cos(f"{tag}{'/c' if use_c else ''}", name, got, gref)
            rel = rel_l2(got, gref)
            worst = (min(worst[0], cos), max(worst[1], rel))
            if cos < _cos_bound(tag, name) or rel > rel_max:
                bad.append((name, round(cos, 4), round(rel, 4)))
        assert not bad, (use_c, bad)
        out.append((tr, float(loss), float(ref), worst))
        del model
    return out


@pytest.mark.parametrize("proj", ["mlp_downsample", "mlp_downsample_3x3_fix"])
def test_sft_forward_backward_matches_oracle_autograd(proj):
    """Both drivers: the Python-orchestrated operator calls and ONE `vila_sft_fwd_bwd` call with the grad-ready callback."""
    from vila_amd import configs
    cfg = configs.tiny(proj, tied=(proj != "mlp_downsample"))
    g = torch.Generator().manual_seed(22)
    L = 14
    ids = torch.randint(0, 900, (2, L), generator=g)
    ids[0, 0] = cfg.image_token_id
    ids[1, 0] = cfg.image_token_id; ids[1, 5] = cfg.image_token_id
    mask = torch.ones(2, L, dtype=torch.bool); mask[0, 11:] = False
    labels = torch.randint(0, 900, (2, L), generator=g); labels[:, :6] = -100
    res = _sft_vs_oracle(cfg, 3, ids, labels, mask, 3, tag=f"tiny/{proj}", c_abi=(False, True))
    # the gradient buckets were announced in backward order and cover the exchange — identically by both drivers
    orders = [[p for p, _, _ in tr.reducer.log] for tr, _, _, _ in res]
    assert orders[0] == orders[1]
    assert orders[0][0] in ("llm.lm_head.", "llm.model.norm.") and orders[0][-1].endswith("embeddings.")


def test_sft_forward_backward_at_8b_widths_matches_oracle_autograd():
    """BASELINE configs[2] shapes per sample (1 x 448^2 image + 512 text tokens, S = 769) at NVILA-8B WIDTHS with 3 ViT + 2 LLM layers,
    b = 2 packed (T = 1538): the wgrad K tail (K = 1538 -> padded), the 28-head / 4-KV-head dK/dV two-group path, hd-72 non-causal
    backward at 1024 tokens, the 4608-wide projector and a 32000-row head all run at their real sizes against fp32 autograd."""
    from vila_amd import configs, synthetic
    cfg = configs.reduced_8b(layers_v=4, layers_l=2, vocab=32000)      # select_layer = -2 -> 3 ViT layers run
    cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
    b, T = 2, 512
    ids = torch.stack([synthetic.make_prompt(cfg, T, 1, 40 + i) for i in range(b)], 0)
    labels = ids.clone()
    labels[:, : 1 + T - 256] = -100
    mask = torch.ones_like(ids, dtype=torch.bool)
    for use_c, (tr, loss, ref, worst) in zip((False, True), _sft_vs_oracle(cfg, 17, ids, labels, mask, b, tag="8b_width", c_abi=(False, True))):
        print(f"8B-width SFT fwd+bwd ({'one vila_sft_fwd_bwd call' if use_c else 'python-orchestrated'}): loss {loss:.5f} vs oracle {ref:.5f}; "
              f"worst grad cosine {worst[0]:.4f}, worst rel-L2 {worst[1]:.4f}")


def test_sft_dynamic_s2_forward_backward_matches_oracle_autograd():
    """The NVILA-8B training recipe (scripts/NVILA/stage1_9tile.sh:19-22: dynamic_s2, scales 1x / 2x / 3x): tower on every tile of every
    scale -> merge kernel -> 3C-wide projector -> chessboard re-merge folded into the splice, and backward through all of it (merge
    adjoint kernel) vs fp32 autograd through the restated reference (llava_arch.py:298-390), both drivers.  Two samples: a 2 x 3 image
    (1 + 4 + 6 tiles, up- AND down-sampled scales) and a `block_sizes = None` image."""
    from vila_amd import configs
    cfg = configs.tiny_s2()
    blocks = [(2, 3), None]
    g = torch.Generator().manual_seed(23)
    L = 12
    ids = torch.randint(0, 900, (2, L), generator=g)
    ids[0, 1] = cfg.image_token_id
    ids[1, 0] = cfg.image_token_id
    mask = torch.ones(2, L, dtype=torch.bool); mask[1, 9:] = False
    labels = torch.randint(0, 900, (2, L), generator=g); labels[:, :5] = -100
    res = _sft_vs_oracle(cfg, 5, ids, labels, mask, 12, tag="tiny_s2", c_abi=(False, True), block_sizes=blocks)
    orders = [[p for p, _, _ in tr.reducer.log] for tr, _, _, _ in res]
    assert orders[0] == orders[1] and "mm_projector." in orders[0]


def test_sft_dynamic_s2_with_videos_matches_oracle_autograd():
    """Videos inside the dynamic_s2 recipe (the NVILA-8B SFT stages mix both): the video encoders call encode_images WITHOUT block sizes
    (video/basic.py:48, tsp.py:59), so every frame is a one-tile image whose features are repeated over the scales (llava_arch.py:309-314,
    367-368).  One 2 x 2 image + a 2-frame video in sample 0, a 4-frame video in sample 1: BasicVideoEncoder through both drivers, then the
    pooling encoder (its pooled rows sit behind projector BLOCKS whose index is no longer the tile index)."""
    from vila_amd import configs
    cfg = configs.tiny_s2()
    g = torch.Generator().manual_seed(29)
    L = 14
    ids = torch.randint(0, 900, (2, L), generator=g)
    ids[0, 1] = cfg.image_token_id; ids[0, 6] = cfg.video_token_id
    ids[1, 2] = cfg.video_token_id
    mask = torch.ones(2, L, dtype=torch.bool); mask[1, 11:] = False
    labels = torch.randint(0, 900, (2, L), generator=g); labels[:, :7] = -100
    n_tiles = 1 + 4 + 4                                                   # the image: 1x and 2x scales in full, its own 2 x 2 blocks at the last scale
    res = _sft_vs_oracle(cfg, 7, ids, labels, mask, n_tiles, tag="tiny_s2_video", c_abi=(False, True), block_sizes=[(2, 2)], video_frames=(2, 4))
    assert abs(res[0][1] - res[1][1]) < 1e-2 * abs(res[0][1])
    _sft_vs_oracle(cfg, 7, ids, labels, mask, n_tiles, tag="tiny_s2_video_tsp", c_abi=(False, True), block_sizes=[(2, 2)], video_frames=(2, 4), tsp=[[2, 2, 1], [1, 1, 1]])


def test_sft_dynamic_s2_at_8b_widths_matches_oracle_autograd():
    """One square image of the NVILA-8B recipe at its real widths: 1 + 4 + 9 = 14 tiles of 448^2 -> 3 x 1152 = 3456 channels -> the
    13 824-wide projector LayerNorm / fc1 (dgrad + wgrad at K = 13 824) -> 2304 image tokens + 64 text tokens; 2 ViT + 2 LLM layers."""
    from vila_amd import configs, synthetic
    cfg = configs.reduced_8b(layers_v=3, layers_l=2, vocab=32000)      # select_layer = -2 -> 2 ViT layers run
    cfg.dynamic_s2 = True
    cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
    ids = synthetic.make_prompt(cfg, 64, 1, 50)[None]
    labels = ids.clone(); labels[:, :33] = -100
    mask = torch.ones_like(ids, dtype=torch.bool)
    for use_c, (tr, loss, ref, worst) in zip((False, True), _sft_vs_oracle(cfg, 19, ids, labels, mask, 14, tag="8b_width_s2", c_abi=(False, True), block_sizes=[(3, 3)])):
        print(f"8B-width dynamic_s2 SFT fwd+bwd ({'one vila_sft_fwd_bwd call' if use_c else 'python-orchestrated'}): loss {loss:.5f} vs oracle "
              f"{ref:.5f}; worst grad cosine {worst[0]:.4f}, worst rel-L2 {worst[1]:.4f}")


def test_two_identical_steps_produce_identical_bits():
    """Round 6 (parity hardening iii): no reduction of the step is order-dependent any more — the column sums (bias gradients),
    the norm backward's dw / db, the CE row sum, the embedding-row scatter with REPEATED token ids and the clipping norm are fixed-order two-pass
    sums (train.hip).  Two fresh trainers on the same batch: gradients, loss and the post-step master weights are equal BIT FOR BIT, with the
    weight-gradient GEMMs on their side stream and the per-bucket optimizer on its own."""
    from vila_amd import configs, synthetic
    from vila_amd.train import SFTTrainer
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    px = synthetic.make_pixels(cfg, 2, 6).to(torch.bfloat16)
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 40, (2, 24), generator=g)                    # a 40-id alphabet over 48 positions: every id repeats
    ids[:, 0] = cfg.image_token_id
    labels = ids.clone(); labels[:, :6] = -100
    outs = []
    for _ in range(2):
        tr = SFTTrainer(build_model(cfg, seed=6), lr=1e-3, max_grad_norm=1.0)
        loss = float(tr.forward_backward(ids, [p.cuda() for p in px], labels))
        torch.cuda.synchronize()
        grads = tr.flat.grads.clone()
        tr2 = SFTTrainer(build_model(cfg, seed=6), lr=1e-3, max_grad_norm=1.0)
        l2 = [float(tr2.step(ids, [p.cuda() for p in px], labels)) for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((loss, grads, l2, tr2.flat.master.clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and torch.equal(outs[0][3], outs[1][3])
    assert float(outs[0][1].float().abs().sum()) > 0


@pytest.mark.parametrize("algo", ["all_reduce", "direct"])
def test_gradient_exchange_over_nccl_in_a_world_of_one_equals_the_step_without_exchange(algo):
    """Round 6 (VERDICT round 5, item 9 i): the data-parallel exchange on the backend the driver's --gpus N uses.  backend "nccl" (= RCCL) is
    brought up in a world of ONE and `GradReducer(force=True)` pushes EVERY gradient bucket of a real forward + backward through it — the
    per-bucket all-reduce and the all-pairs form (all-to-all of shards + rank-ordered fp32 sum + all-gather) — on the optimizer stream, with
    the per-bucket AdamW behind each.  A sum over one rank is the identity, and the step's reductions are deterministic, so gradients and
    post-step masters must equal the no-exchange step's BIT FOR BIT; every bucket must have been handed to the group."""
    import torch.distributed as dist
    from vila_amd import configs, synthetic
    from vila_amd.train import GradReducer, SFTTrainer
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    px = synthetic.make_pixels(cfg, 2, 9).to(torch.bfloat16)
    ids = torch.stack([synthetic.make_prompt(cfg, 20, 1, 90 + i) for i in range(2)], 0)
    labels = ids.clone(); labels[:, :9] = -100

    def run(tr):
        losses = [float(tr.step(ids, [p.cuda() for p in px], labels)) for _ in range(2)]
        torch.cuda.synchronize()
        return losses, tr.flat.grads.clone(), tr.flat.master.clone()
    plain = run(SFTTrainer(build_model(cfg, seed=9), lr=1e-3))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        tr = SFTTrainer(build_model(cfg, seed=9), lr=1e-3)
        tr.reducer = GradReducer(tr.flat, None, force=True, algo=algo)
        assert tr.reducer.active() and dist.get_backend() == "nccl"
        got = run(tr)
        n_buckets = len({p for p, _, _ in tr.reducer.log})
        assert n_buckets >= cfg.llm.num_hidden_layers + cfg.vision.num_used_layers + 3
        spans = {pre: (a, b) for pre, a, b in tr.reducer.log}
        assert tr.reducer.exchanged_bytes == 2 * sum(b - a for a, b in spans.values()) * 2   # two steps x every announced bucket's bf16 slice
        assert tr.reducer.exchanged_bytes >= 2 * tr.flat.numel * 2 * 0.9                     # (all but the parameters no backward reaches)
    finally:
        dist.destroy_process_group()
    assert got[0] == plain[0], (got[0], plain[0])
    assert torch.equal(got[1], plain[1]) and torch.equal(got[2], plain[2])


def test_sft_step_updates_parameters_and_lowers_loss():
    from vila_amd import configs, synthetic
    from vila_amd.train import SFTTrainer, count_targets
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    model = build_model(cfg, seed=4)
    tr = SFTTrainer(model, lr=2e-3)
    px = synthetic.make_pixels(cfg, 1, 4).to(torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 16, 1, 4)[None]
    labels = ids.clone(); labels[:, :8] = -100
    losses = [float(tr.step(ids, [px[0].cuda()], labels)) for _ in range(6)]
    assert losses[-1] < losses[0] - 0.5, losses
    # inference through the C-ABI model path sees the updated (flat) parameters
    out = model.generate(input_ids=ids, media={"image": [px[0].cuda()]}, max_new_tokens=2, eos_token_id=-1)
    assert out.shape == (1, 2)


def test_captured_decode_graph_follows_the_weights_into_and_through_training():
    """ADVICE (round 1): the decode hipGraph bakes weight pointers in.  generate -> SFTTrainer(model) (parameters move into the flat buffer)
    -> generate -> training steps -> generate must never replay a graph that reads the old storage: after the move the captured path
    reproduces the same ids, after training it agrees with the eager launches on the NEW weights (and the logits did change)."""
    from vila_amd import configs, synthetic
    from vila_amd.train import SFTTrainer
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    model = build_model(cfg, seed=6)
    px = synthetic.make_pixels(cfg, 1, 6).to(torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 16, 1, 6)[None]
    media = {"image": [px[0].cuda()]}
    kw = dict(max_new_tokens=6, eos_token_id=-1)
    before = model.generate(input_ids=ids, media=media, **kw)                      # captures the graph on the original storage
    e, _, _ = model._embed(ids, media)
    _, lg0 = model.llm.generate(inputs_embeds=e, return_logits=True, use_graph=False, **kw)
    tr = SFTTrainer(model, lr=5e-3)                                                # parameters become views of the flat buffer
    moved = model.generate(input_ids=ids, media=media, **kw)
    assert torch.equal(moved, before)
    labels = ids.clone(); labels[:, :8] = -100
    for _ in range(4):
        tr.step(ids, [px[0].cuda()], labels)
    e, _, _ = model._embed(ids, media)
    graph = model.llm.generate(inputs_embeds=e, use_graph=True, **kw)
    eager, lg1 = model.llm.generate(inputs_embeds=e, return_logits=True, use_graph=False, **kw)
    assert torch.equal(graph, eager)
    assert float((lg1[0] - lg0[0]).abs().max()) > 1e-2                            # the weights really moved


def test_per_bucket_adamw_equals_one_flat_step():
    """The optimizer stream applies AdamW bucket by bucket as soon as a layer's gradients are final (overlapped with the backward of
    the layers below).  Same kernels, same gradients: after two steps the parameters, master copy and moments must equal those of the
    path that waits for the whole backward and updates the flat buffer in one launch (taken when a global clipping norm is set; 1e30
    never clips, its scale is exactly 1.0) up to the run-to-run jitter of the fp32 atomics in the norm-weight gradients."""
    from vila_amd import configs, synthetic
    from vila_amd.train import SFTTrainer
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    px = synthetic.make_pixels(cfg, 2, 6).to(torch.bfloat16)
    ids = torch.stack([synthetic.make_prompt(cfg, 20, 1, 6), synthetic.make_prompt(cfg, 20, 1, 7)], 0)
    labels = ids.clone(); labels[:, :9] = -100
    outs = []
    for clip in (None, 1e30):
        model = build_model(cfg, seed=6)
        tr = SFTTrainer(model, lr=1e-3, weight_decay=0.01, max_grad_norm=clip)
        losses = [float(tr.step(ids, [p.cuda() for p in px], labels)) for _ in range(2)]
        torch.cuda.synchronize()
        outs.append((losses, tr.flat.params.clone(), tr.flat.master.clone(), tr.flat.m.clone(), tr.flat.v.clone(), tr))
    (la, pa, ma, m1a, va, tra), (lb, pb, mb, m1b, vb, trb) = outs
    # (not bit-for-bit across two runs: the loss scalar and the norm-weight gradients are accumulated with fp32 atomics)
    assert all(abs(x - y) < 1e-5 * abs(y) for x, y in zip(la, lb)), (la, lb)
    # tensors no bucket covers (27th ViT layer, post_layernorm: never reached by hidden_states[-2]) have zero gradients; the per-bucket
    # path leaves them untouched (HF skips parameters without a gradient), the flat launch decays them: compare the covered slices
    covered = torch.zeros(tra.flat.numel, dtype=torch.bool, device="cuda")
    for _, a, b in tra.reducer.log:
        covered[a:b] = True
    assert bool(covered.any()) and not bool(covered.all())
    for name, x, y in (("params", pa, pb), ("master", ma, mb), ("exp_avg", m1a, m1b), ("exp_avg_sq", va, vb)):
        d = float((x[covered].float() - y[covered].float()).abs().max())
        assert d <= 2e-3 * float(y[covered].float().abs().max()), (name, d)       # <= 1 bf16 ulp of the largest entry
        assert rel_l2(x[covered], y[covered]) < 1e-4, (name, rel_l2(x[covered], y[covered]))
    # what no bucket covers is left alone by BOTH paths (torch.optim.AdamW semantics for grad = None; ADVICE round 2: the flat launch
    # used to apply weight decay there, so the trained weights depended on a performance switch)
    assert torch.equal(ma[~covered], mb[~covered]) and torch.equal(pa[~covered], pb[~covered])
    assert float(m1a[~covered].abs().max()) == 0 and float(m1b[~covered].abs().max()) == 0


@pytest.mark.parametrize("use_c", [False, True])
def test_autograd_seam_reference_training_call_site(use_c):
    """SURVEY §8b / VERDICT round 2: after the swap the reference's own call site (transformer_normalize_monkey_patch.py:183-249) —
    `loss = model(**inputs).loss; loss.backward(); optimizer.step(); model.zero_grad()` — must produce gradients.  `.grad` of every
    parameter vs fp32 autograd through the oracle (cosine >= 0.999), gradient accumulation over two micro-batches, a torch optimizer step
    on the parameters, zero_grad(set_to_none=True) and a fresh backward."""
    from oracle import vila_oracle as O
    from vila_amd import configs, synthetic
    from vila_amd.train import count_targets
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 8).items()}
    px = synthetic.make_pixels(cfg, 2, 8).to(torch.bfloat16)
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(0, 900, (2, 12), generator=g); ids[:, 0] = cfg.image_token_id
    labels = torch.randint(0, 900, (2, 12), generator=g); labels[:, :5] = -100
    mask = torch.ones(2, 12, dtype=torch.bool); mask[1, 10:] = False
    n_items = count_targets(ids, labels, mask, cfg.image_token_id)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    ref = O.vlm_sft_loss([p.float() for p in px], ids, labels, mask, wr, cfg, num_items_in_batch=n_items, packed=True)
    ref.backward()

    model = build_model(cfg, weights=w)
    model.enable_autograd(use_c_abi=use_c)
    model.train()
    inputs = dict(input_ids=ids, media={"image": [p.cuda() for p in px]}, labels=labels, attention_mask=mask, num_items_in_batch=n_items)
    params = {}
    for prefix, mod in (("llm.", model.llm), ("vision_tower.", model.vision_tower), ("mm_projector.", model.mm_projector)):
        params.update({prefix + n: p for n, p in mod.named_parameters()})
    assert all(p.requires_grad and p.grad is None for p in params.values())
    loss = model(**inputs).loss                                   # the reference's compute_loss
    assert loss.requires_grad and abs(float(loss) - float(ref)) < 1e-2 * abs(float(ref))
    loss.backward()                                               # accelerator.backward(loss)
    worst = 1.0
    for name, gref in ((k, v.grad) for k, v in wr.items()):
        got = params[name].grad
        assert got is not None and got.shape == params[name].shape, name
        if gref is None or float(gref.norm()) < 1e-6:            # never reached by hidden_states[-2] (27th ViT layer, post_layernorm): zero here
            assert float(got.float().norm()) < 1e-3, name
            continue
        cos = grad_cos("seam", name, got.float().cpu(), gref)
        worst = min(worst, cos)
        assert cos >= _cos_bound("seam", name), (name, cos)
    # gradient accumulation: a second micro-batch adds to .grad (same batch -> doubled, up to bf16 rounding of the sum)
    g1 = {n: p.grad.float().clone() for n, p in params.items()}
    (model(**inputs).loss * 0.5).backward()                       # upstream scaling (loss / gradient_accumulation_steps) reaches the grads
    name = "llm.model.layers.0.mlp.down_proj.weight"
    assert rel_l2(params[name].grad, 1.5 * g1[name]) < 1e-2
    # a torch optimizer works on the parameters; the HIP path sees the update (parameters are views of the flat buffer)
    opt = torch.optim.SGD(list(params.values()), lr=1e-2)
    before = float(model(**inputs).loss)
    opt.step()
    model.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in params.values())
    after = model(**inputs).loss
    assert float(after) < before
    after.backward()
    assert rel_l2(params[name].grad, g1[name]) > 1e-3             # fresh gradients of the UPDATED weights, not stale ones
    model.eval()
    with torch.no_grad():
        out = model(**inputs)                                     # eval mode: the inference forward (no autograd)
    assert out.loss is not None and not out.loss.requires_grad
    print(f"autograd seam ({'one C-ABI call' if use_c else 'python-orchestrated'}): loss {float(loss):.5f} vs oracle {float(ref):.5f}, worst cosine {worst:.4f}")


def test_autograd_seam_honours_frozen_components():
    """the reference's stages freeze parts of the model (llava/train/args.py tune_language_model / tune_vision_tower /
    tune_mm_projector; stage 1 trains the projector alone).  A frozen component's parameters keep requires_grad = False and never get a
    `.grad`; the trainable ones get the same gradients as the all-trainable seam (the tower's backward is skipped, nothing else changes)."""
    from vila_amd import configs, synthetic
    from vila_amd.train import count_targets
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 9).items()}
    px = synthetic.make_pixels(cfg, 2, 9).to(torch.bfloat16)
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, 900, (2, 12), generator=g); ids[:, 0] = cfg.image_token_id
    labels = torch.randint(0, 900, (2, 12), generator=g); labels[:, :5] = -100
    n_items = count_targets(ids, labels, None, cfg.image_token_id)
    inputs = dict(input_ids=ids, media={"image": [p.cuda() for p in px]}, labels=labels, num_items_in_batch=n_items)

    def grads(**tune):
        model = build_model(cfg, weights=w)
        model.enable_autograd(use_c_abi=False, **tune)
        model.train()
        loss = model(**inputs).loss
        loss.backward()
        out = {}
        for prefix, mod in (("llm.", model.llm), ("vision_tower.", model.vision_tower), ("mm_projector.", model.mm_projector)):
            out.update({prefix + n: p for n, p in mod.named_parameters()})
        return float(loss), out
    loss_all, p_all = grads()
    loss_s1, p_s1 = grads(tune_language_model=False, tune_vision_tower=False, tune_mm_projector=True)       # stage 1: align the projector
    assert abs(loss_all - loss_s1) < 1e-6
    for n, p in p_s1.items():
        if n.startswith("mm_projector."):
            assert p.requires_grad and p.grad is not None and rel_l2(p.grad, p_all[n].grad) < 1e-6, n
        else:
            assert not p.requires_grad and p.grad is None, n
    loss_s2, p_s2 = grads(tune_vision_tower=False)                                                        # tower frozen, LLM + projector train
    for n, p in p_s2.items():
        if n.startswith("vision_tower."):
            assert not p.requires_grad and p.grad is None, n
        else:
            assert p.grad is not None and rel_l2(p.grad, p_all[n].grad) < 1e-6, n
    opt = torch.optim.SGD([p for p in p_s1.values() if p.requires_grad], lr=1e-2)                           # what an HF Trainer would build
    opt.step()


@pytest.mark.parametrize("use_c", [False, True])
def test_sft_step_with_video_media_matches_oracle_autograd(use_c):
    """Round 4 (VERDICT "missing #4": video media through the SFT step / autograd seam used to raise).  `<vila/video>` tokens under the
    BasicVideoEncoder (encoders/video/basic.py:13-53: every frame = its tokens + "\\n", frames concatenated): a batch with one image sample and
    one 3-frame video sample — loss and every gradient against fp32 autograd through the oracle's restatement (embed_splice with both media
    deques, llava_arch.py:454-466), both drivers; then the same batch through the autograd seam (the reference's `model(**inputs).loss`)."""
    from oracle import vila_oracle as O
    from vila_amd import configs, synthetic
    from vila_amd.train import SFTTrainer, count_targets
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    seed = 31
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, seed).items()}
    px = synthetic.make_pixels(cfg, 4, seed).to(torch.bfloat16)
    images, video = [px[0]], px[1:4]                                    # sample 0: one image; sample 1: one 3-frame video
    g = torch.Generator().manual_seed(seed)
    L = 14
    ids = torch.randint(0, 900, (2, L), generator=g)
    ids[0, 0] = cfg.image_token_id
    ids[1, 2] = cfg.video_token_id
    labels = torch.randint(0, 900, (2, L), generator=g)
    labels[:, :6] = -100
    mask = torch.ones(2, L, dtype=torch.bool); mask[0, 11:] = False
    n_items = count_targets(ids, labels, mask, (cfg.image_token_id, cfg.video_token_id))
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    ref = O.vlm_sft_loss([p.float() for p in images], ids, labels, mask, wr, cfg, num_items_in_batch=n_items, packed=True, videos=[video.float()])
    ref.backward()
    model = build_model(cfg, weights=w)
    tr = SFTTrainer(model, optimizer_state=False)
    fb = tr.forward_backward_c if use_c else tr.forward_backward
    loss = fb(ids, [p.cuda() for p in images], labels, mask, n_items, None, videos=[video.cuda()])
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 1e-2 * abs(float(ref)), (float(loss), float(ref))
    grads = tr.flat.named_grads()
    worst = 1.0
    for name, gref in ((k, v.grad) for k, v in wr.items()):
        if name not in grads or gref is None:
            continue
        got = grads[name].float().cpu()
        if float(gref.norm()) < 1e-6:
            assert float(got.norm()) < 1e-3, name
            continue
        cos = grad_cos("video", name, got, gref)
        worst = min(worst, cos)
        assert cos >= _cos_bound("video", name), (name, cos)
    if not use_c:
        # the reference's call site: model(**inputs).loss with media = {"image": [...], "video": [...]} through the autograd seam
        m2 = build_model(cfg, weights=w)
        m2.enable_autograd(use_c_abi=False)
        m2.train()
        out = m2(input_ids=ids, media={"image": [p.cuda() for p in images], "video": [video.cuda()]}, labels=labels, attention_mask=mask,
                 num_items_in_batch=n_items)
        assert abs(float(out.loss) - float(ref)) < 1e-2 * abs(float(ref))
        out.loss.backward()
        p = dict(m2.mm_projector.named_parameters())["layers.1.weight"]
        assert p.grad is not None and grad_cos("video/seam", "mm_projector.layers.1.weight", p.grad.float().cpu(), wr["mm_projector.layers.1.weight"].grad) >= GRAD_COS_MIN
    print(f"SFT with video media ({'one C-ABI call' if use_c else 'python-orchestrated'}): loss {float(loss):.5f} vs oracle {float(ref):.5f}, worst cosine {worst:.4f}")


class _MapTokenizer:
    """The synthetic tokenizer plus a fixed string -> ids map for the video encoder's start / end / separator strings."""

    def __init__(self, inner, table):
        self._inner, self._table = inner, dict(table)

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def __call__(self, text):
        from types import SimpleNamespace
        if text in self._table:
            return SimpleNamespace(input_ids=list(self._table[text]))
        return self._inner(text)


@pytest.mark.parametrize("pool_sizes,start,end,sep", [
    ([[2, 1, 1]], None, "\n", None),                       # temporal pooling only (NVILA-Video's shape of recipe: scripts/NVILA/stage4.sh:50 uses [[8,1,1]])
    ([[2, 2, 2], [1, 1, 1]], "<s>", "<e>", "<sep>"),      # two pool sizes over the same frames (pooled + unpooled), start / end / separator tokens
    ([[4, 2, 1], [1, 1, 2]], None, None, "<sep>"),        # no end token, anisotropic windows
])
def test_sft_step_with_tsp_video_encoder_matches_oracle_autograd(pool_sizes, start, end, sep):
    """Row a13 over row a7's TSPVideoEncoder (encoders/video/tsp.py:14-64, an nn.Module inside the graph the reference trains): the pooled
    rows go through `vila_video_pool_bf16`, their gradients come back through its adjoint `vila_video_pool_bwd_bf16` (accumulating across
    pool sizes), the start / end / separator tokens' gradients land in the embedding table.  Loss and EVERY parameter's gradient against fp32
    autograd through the oracle's restatement of the encoder, plus the autograd seam on the same batch."""
    from oracle import vila_oracle as O
    from vila_amd import configs, synthetic
    from vila_amd.train import SFTTrainer, count_targets
    from vila_amd.vlm import build_model, TSPVideoEncoder
    cfg = configs.tiny("mlp_downsample")
    seed = 37
    table = {"<s>": [21, 22], "<e>": [23], "<sep>": [24, 25, 26]}
    tok = lambda t: None if t is None else ([cfg.newline_token_id] if t == "\n" else table[t])
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, seed).items()}
    px = synthetic.make_pixels(cfg, 9, seed).to(torch.bfloat16)
    images, vid_a, vid_b = [px[0]], px[1:5], px[5:9]                     # sample 0: image + 4-frame video; sample 1: 4-frame video
    g = torch.Generator().manual_seed(seed)
    L = 16
    ids = torch.randint(0, 900, (2, L), generator=g)
    ids[0, 0] = cfg.image_token_id
    ids[0, 5] = cfg.video_token_id
    ids[1, 3] = cfg.video_token_id
    labels = torch.randint(0, 900, (2, L), generator=g)
    labels[:, :7] = -100
    mask = torch.ones(2, L, dtype=torch.bool); mask[1, 13:] = False
    n_items = count_targets(ids, labels, mask, (cfg.image_token_id, cfg.video_token_id))
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    enc = {"pool_sizes": pool_sizes, "start_ids": tok(start), "end_ids": tok(end) or [], "sep_ids": tok(sep)}
    ref = O.vlm_sft_loss([p.float() for p in images], ids, labels, mask, wr, cfg, num_items_in_batch=n_items, packed=True,
                         videos=[vid_a.float(), vid_b.float()], video_encoder=enc)
    ref.backward()

    def make():
        m = build_model(cfg, weights=w)
        m.tokenizer = _MapTokenizer(m.tokenizer, table)
        m.encoders["video"] = TSPVideoEncoder(m, pool_sizes, start_tokens=start, end_tokens=end, sep_tokens=sep)
        return m
    model = make()
    tr = SFTTrainer(model, optimizer_state=False)
    loss = tr.forward_backward(ids, [p.cuda() for p in images], labels, mask, n_items, None, videos=[vid_a.cuda(), vid_b.cuda()])
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 1e-2 * abs(float(ref)), (float(loss), float(ref))
    grads = tr.flat.named_grads()
    worst, n_checked = 1.0, 0
    for name, gref in ((k, v.grad) for k, v in wr.items()):
        if name not in grads or gref is None:
            continue
        got = grads[name].float().cpu()
        if float(gref.norm()) < 1e-6:
            assert float(got.norm()) < 1e-3, name
            continue
        cos = grad_cos("tsp", name, got, gref)
        worst = min(worst, cos)
        n_checked += 1
        assert cos >= _cos_bound("tsp", name), (name, cos)
    assert n_checked > 20
    # the tower's gradient exists only through the pooled rows for the videos: it must be there
    assert float(grads["vision_tower.vision_tower.vision_model.embeddings.patch_embedding.weight"].float().norm()) > 0
    # the encoder's own tokens were trained: their embedding rows carry gradient, as in the oracle
    ge, ge_ref = grads["llm.model.embed_tokens.weight"].float().cpu(), wr["llm.model.embed_tokens.weight"].grad
    for t in (tok(start) or []) + (tok(end) or []) + (tok(sep) or []):
        assert float(ge_ref[t].norm()) > 0 and grad_cos("tsp", f"embed_tokens.row{t}", ge[t], ge_ref[t]) >= _cos_bound("tsp", f"embed_tokens.row{t}"), t
    # the one-call driver (`vila_sft_fwd_bwd` with the batch's `pools` array): same loss, same gradients
    tr_c = SFTTrainer(make(), optimizer_state=False)
    loss_c = tr_c.forward_backward_c(ids, [p.cuda() for p in images], labels, mask, n_items, None, videos=[vid_a.cuda(), vid_b.cuda()])
    torch.cuda.synchronize()
    assert abs(float(loss_c) - float(ref)) < 1e-2 * abs(float(ref)), (float(loss_c), float(ref))
    grads_c = tr_c.flat.named_grads()
    for name, got in grads.items():
        if float(got.float().norm()) > 1e-6:
            assert grad_cos("tsp/c_vs_py", name, grads_c[name].float(), got.float()) >= GRAD_COS_MIN, name
    # the autograd seam on the same batch (the reference's call site)
    m2 = make()
    m2.enable_autograd(use_c_abi=False)
    m2.train()
    out = m2(input_ids=ids, media={"image": [p.cuda() for p in images], "video": [vid_a.cuda(), vid_b.cuda()]}, labels=labels, attention_mask=mask,
             num_items_in_batch=n_items)
    assert abs(float(out.loss) - float(ref)) < 1e-2 * abs(float(ref))
    out.loss.backward()
    p = dict(m2.mm_projector.named_parameters())["layers.1.weight"]
    assert p.grad is not None and grad_cos("tsp/seam", "mm_projector.layers.1.weight", p.grad.float().cpu(), wr["mm_projector.layers.1.weight"].grad) >= GRAD_COS_MIN
    print(f"SFT with TSPVideoEncoder {pool_sizes} start={start!r} end={end!r} sep={sep!r}: loss {float(loss):.5f} vs oracle {float(ref):.5f}, "
          f"worst cosine {worst:.4f} over {n_checked} tensors")


def test_video_pool_adjoint_is_the_transpose_of_the_pooling():
    """<pool(x), y> == <x, pool_bwd(y)> on integer-valued inputs (exact in bf16 / fp32), and accumulate adds."""
    from vila_amd import ops
    g = torch.Generator().manual_seed(5)
    for nt, nl, Cc, pool in [(4, 4, 16, (2, 2, 2)), (6, 6, 8, (3, 2, 1)), (2, 3, 24, (1, 3, 3)), (8, 2, 8, (8, 1, 1))]:
        x = torch.randint(-4, 5, (nt, nl * nl, Cc), generator=g).to(torch.bfloat16).cuda()
        n_out = (nt // pool[0]) * (nl // pool[1]) * (nl // pool[2])
        k = pool[0] * pool[1] * pool[2]
        y = (torch.randint(-4, 5, (n_out, Cc), generator=g) * k).to(torch.bfloat16).cuda()       # multiples of the window: y / k exact
        px = ops.video_pool(x * k, pool)                                                          # x * k: the mean is exact
        by = ops.video_pool_bwd(y, nt, nl, pool)
        lhs = float((px.double() * y.double()).sum())                                            # sum_w (sum_{i in w} x_i k / k) y_w
        rhs = float((x.double() * by.double()).sum()) * k                                         # sum_i x_i (y_w / k), times k
        assert lhs == rhs, (pool, lhs, rhs)
        twice = ops.video_pool_bwd(y, nt, nl, pool, out=by.clone(), accumulate=True)
        assert torch.equal(twice.float(), 2 * by.float())
    with pytest.raises(ValueError, match="invalid for pooling"):
        ops.video_pool(torch.zeros(3, 4, 8, dtype=torch.bfloat16).cuda(), (2, 1, 1))
