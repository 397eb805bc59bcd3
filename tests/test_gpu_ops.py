"""GPU parity tests, operator level: every HIP kernel (called through the C-ABI) against a plain PyTorch fp32
reference of the same op on the same seeded bf16 inputs.  Tolerances are stated per test."""

import pytest
import torch

from tests.gpu_util import max_abs, randn_bf16, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from vila_amd import _lib, ops as _ops
    _lib.load()
    return _ops


GEMM_SHAPES = [
    (128, 128, 64), (16, 16, 8), (1, 24, 40), (200, 264, 136), (130, 132, 72),
    (769, 3584, 3584), (289, 512, 3584), (1024, 4304, 1152), (1024, 1152, 4304), (256, 3584, 4608),
    (1024, 1152, 592), (7, 1000, 512),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain_bias_residual(ops, M, N, K):
    """bf16 in / fp32 accumulate / bf16 out.  Tolerance: rel-L2 <= 4e-3 (one bf16 rounding of the output ~ 2^-9)."""
    a = randn_bf16(M, K, seed=1)
    w = randn_bf16(N, K, seed=2, scale=K ** -0.5)
    bias = randn_bf16(N, seed=3)
    res = randn_bf16(M, N, seed=4)
    ref = a.float() @ w.float().t()
    out = ops.gemm(a, w)
    assert rel_l2(out, ref) < 4e-3, f"plain rel={rel_l2(out, ref):.3e}"
    out = ops.gemm(a, w, bias=bias, residual=res)
    ref2 = ref + bias.float() + res.float()
    assert rel_l2(out, ref2) < 4e-3, f"bias+res rel={rel_l2(out, ref2):.3e}"
    out32 = ops.gemm(a, w, out_f32=True)
    assert out32.dtype == torch.float32
    assert rel_l2(out32, ref) < 2e-5, f"fp32-out rel={rel_l2(out32, ref):.3e}"


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 6, 7, 8])
@pytest.mark.parametrize("M,N,K", [(300, 264, 136), (769, 3584, 512), (1, 24, 40), (3076, 1088, 512), (513, 260, 72), (700, 520, 128)])
def test_gemm_every_tile_shape(ops, tile, M, N, K):
    """The three tile shapes (128x128, 128x64, 256x128) must agree with the fp32 reference on every epilogue."""
    from vila_amd import _lib
    lib = _lib.load()
    a = randn_bf16(M, K, seed=41)
    w = randn_bf16(N, K, seed=42, scale=K ** -0.5)
    w2 = randn_bf16(N, K, seed=43, scale=K ** -0.5)
    bias, res = randn_bf16(N, seed=44), randn_bf16(M, N, seed=45)
    ref = a.float() @ w.float().t()
    lib.vila_gemm_force_tile(tile)
    try:
        assert rel_l2(ops.gemm(a, w, bias=bias, residual=res), ref + bias.float() + res.float()) < 4e-3
        assert rel_l2(ops.gemm(a, w, out_f32=True), ref) < 2e-5
        assert rel_l2(ops.gemm(a, w, bias=bias, epi=1), torch.nn.functional.gelu(ref + bias.float(), approximate="tanh")) < 5e-3
        gu = torch.nn.functional.silu(ref) * (a.float() @ w2.float().t())
        assert rel_l2(ops.gemm(a, w, w2=w2, epi=3), gu) < 5e-3
    finally:
        lib.vila_gemm_force_tile(0)


@pytest.mark.parametrize("M,N,K", [(769, 3584, 18944), (600, 520, 1024), (769, 4608, 3584), (1024, 1152, 4304), (515, 300, 1096)])
def test_gemm_splitk_with_workspace(ops, M, N, K):
    """Split-K over grid.y of the 256x256 LDS-DMA kernel (fp32 slabs + reduce with bias/residual)."""
    from vila_amd import _lib
    lib = _lib.load()
    a = randn_bf16(M, K, seed=51)
    w = randn_bf16(N, K, seed=52, scale=K ** -0.5)
    bias, res = randn_bf16(N, seed=53), randn_bf16(M, N, seed=54)
    ws = torch.empty(8 * M * N, device="cuda", dtype=torch.float32)
    ref = a.float() @ w.float().t() + bias.float() + res.float()
    lib.vila_gemm_force_tile(5)
    try:
        out = ops.gemm(a, w, bias=bias, residual=res, ws=ws)
    finally:
        lib.vila_gemm_force_tile(0)
    assert rel_l2(out, ref) < 4e-3, f"rel={rel_l2(out, ref):.3e}"
    x = res.clone()
    ops.gemm(a, w, residual=x, out=x, ws=ws)      # in place on the residual stream, automatic choice
    assert rel_l2(x, a.float() @ w.float().t() + res.float()) < 4e-3


@pytest.mark.parametrize("M,N,K", [(769, 18944, 1024), (769, 18944, 3584), (289, 18944, 3584), (600, 16640, 1096)])
def test_gemm_gateup_tail_round_split(ops, M, N, K):
    """Gate/up GEMM whose last round of 256x256 tiles is under-filled: full rounds fused + K-sliced tail + silu(g)*u reduce.
    Must equal the single-launch result (same tolerance as every gate/up epilogue: rel-L2 <= 5e-3 vs fp32)."""
    a = randn_bf16(M, K, seed=61)
    w = randn_bf16(N, K, seed=62, scale=K ** -0.5)
    w2 = randn_bf16(N, K, seed=63, scale=K ** -0.5)
    ws = torch.empty(6 * M * 3584, device="cuda", dtype=torch.float32)
    ref = torch.nn.functional.silu(a.float() @ w.float().t()) * (a.float() @ w2.float().t())
    out_plain = ops.gemm(a, w, w2=w2, epi=3)
    out_split = ops.gemm(a, w, w2=w2, epi=3, ws=ws)
    assert rel_l2(out_plain, ref) < 5e-3
    assert rel_l2(out_split, ref) < 5e-3, f"rel={rel_l2(out_split, ref):.3e}"
    assert rel_l2(out_split, out_plain) < 3e-3


def test_gemm_detects_transpose_and_identity(ops):
    """A = I with an asymmetric W must give exactly W^T rows (guide rule: symmetric inputs hide a swapped C layout)."""
    n = 256
    a = torch.eye(n, device="cuda", dtype=torch.bfloat16)
    w = (torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n) % 251 - 125).to(torch.bfloat16)
    out = ops.gemm(a, w, out_f32=True)
    assert torch.equal(out, w.float().t().contiguous())


def test_gemm_strided_views_and_inplace_residual(ops):
    M, N, K = 300, 256, 192
    big_a = randn_bf16(M, K + 64, seed=5)
    a = big_a[:, :K]                      # lda != K
    w = randn_bf16(N, K, seed=6, scale=K ** -0.5)
    x = randn_bf16(M, N, seed=7)
    ref = x.float() + a.float() @ w.float().t()
    ops.gemm(a, w, residual=x, out=x)     # x += a w^T in place (how the residual stream is updated)
    assert rel_l2(x, ref) < 4e-3


@pytest.mark.parametrize("epi,fn", [(1, lambda t: torch.nn.functional.gelu(t, approximate="tanh")),
                                    (2, lambda t: torch.nn.functional.gelu(t))])
def test_gemm_gelu_epilogues(ops, epi, fn):
    M, N, K = 257, 272, 144
    a = randn_bf16(M, K, seed=8)
    w = randn_bf16(N, K, seed=9, scale=K ** -0.5)
    b = randn_bf16(N, seed=10)
    ref = fn(a.float() @ w.float().t() + b.float())
    out = ops.gemm(a, w, bias=b, epi=epi)
    assert rel_l2(out, ref) < 5e-3, f"rel={rel_l2(out, ref):.3e}"


@pytest.mark.parametrize("M,N,K", [(769, 18944, 3584), (100, 1088, 512), (33, 40, 64)])
def test_gemm_gateup(ops, M, N, K):
    a = randn_bf16(M, K, seed=11)
    wg = randn_bf16(N, K, seed=12, scale=K ** -0.5)
    wu = randn_bf16(N, K, seed=13, scale=K ** -0.5)
    ref = torch.nn.functional.silu(a.float() @ wg.float().t()) * (a.float() @ wu.float().t())
    out = ops.gemm(a, wg, w2=wu, epi=3)
    assert out.shape == (M, N)
    assert rel_l2(out, ref) < 5e-3, f"rel={rel_l2(out, ref):.3e}"


def test_gemm_rejects_bad_shapes(ops):
    a = randn_bf16(8, 12)
    w = randn_bf16(8, 12)
    with pytest.raises(ValueError, match="multiple of 8"):
        ops.gemm(a, w)


@pytest.mark.parametrize("rows,cols", [(1024, 1152), (256, 4608), (121, 10368), (5, 144), (3, 16384)])
def test_layernorm_rmsnorm(ops, rows, cols):
    x = randn_bf16(rows, cols, seed=14, scale=2.0) + 0.5
    w = randn_bf16(cols, seed=15, scale=0.1) + 1
    b = randn_bf16(cols, seed=16, scale=0.1)
    ref = torch.nn.functional.layer_norm(x.float(), (cols,), w.float(), b.float(), 1e-6)
    out = ops.layernorm(x, w, b, 1e-6)
    assert rel_l2(out, ref) < 3e-3, f"ln rel={rel_l2(out, ref):.3e}"
    x32 = x.float()
    y = (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16)
    ref = (w * y).float()                        # HF: weight * bf16(normalised)  (Qwen2RMSNorm)
    out = ops.rmsnorm(x, w, 1e-6)
    assert rel_l2(out, ref) < 3e-3, f"rms rel={rel_l2(out, ref):.3e}"


@pytest.mark.parametrize("g,k", [(4, 2), (5, 2), (7, 2), (32, 2), (4, 3), (5, 3), (7, 3), (32, 3)])
def test_space_to_depth_bit_exact(ops, g, k):
    from oracle import vila_oracle as O
    C = 16
    x = (torch.arange(2 * g * g * C, dtype=torch.float32).reshape(2, g * g, C) % 255) - 127
    ref = O.downsample_block(x, k)
    out = ops.space_to_depth(x.to("cuda", torch.bfloat16), k)
    assert torch.equal(out.float().cpu(), ref)


def _attn_ref(q, k, v, causal, cu=None):
    """fp32 reference: softmax(QK^T/sqrt(d) + mask) V per sequence, GQA by repeat."""
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    qf, kf, vf = q.float(), k.float().repeat_interleave(Hq // Hkv, 1), v.float().repeat_interleave(Hq // Hkv, 1)
    out = torch.zeros_like(qf)
    bounds = [0, T] if cu is None else cu.tolist()
    for a, b in zip(bounds[:-1], bounds[1:]):
        s = torch.einsum("qhd,khd->hqk", qf[a:b], kf[a:b]) * D ** -0.5
        if causal:
            n = b - a
            s = s.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool, device=q.device), 1), float("-inf"))
        out[a:b] = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vf[a:b])
    return out


@pytest.mark.parametrize("T,Hq,Hkv,D,causal", [
    (1024, 16, 16, 72, False), (16, 2, 2, 72, False), (100, 4, 4, 72, False),
    (769, 28, 4, 128, True), (5, 4, 2, 128, True), (289, 4, 2, 128, True), (200, 4, 4, 64, True), (130, 2, 2, 128, False),
])
def test_attention_forward(ops, T, Hq, Hkv, D, causal):
    """Tolerance: rel-L2 <= 8e-3 (P is rounded to bf16 before PV, output rounded to bf16)."""
    qkv = randn_bf16(T, (Hq + 2 * Hkv) * D, seed=17)
    q = qkv[:, : Hq * D].view(T, Hq, D)                       # strided views of a fused buffer, as in the model
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    ref = _attn_ref(q, k, v, causal)
    out, lse = ops.attn_fwd(q, k, v, causal, return_lse=True)
    assert rel_l2(out, ref) < 8e-3, f"rel={rel_l2(out, ref):.3e} max={max_abs(out, ref):.3e}"
    kf = k.float().repeat_interleave(Hq // Hkv, 1)
    s = torch.einsum("qhd,khd->hqk", q.float(), kf) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    assert max_abs(lse, torch.logsumexp(s, -1)) < 2e-2


def test_attention_varlen_and_batched(ops):
    Hq, Hkv, D = 4, 2, 128
    cu = torch.tensor([0, 100, 357, 400, 401], dtype=torch.int32, device="cuda")
    T = 401
    q, k, v = randn_bf16(T, Hq, D, seed=18), randn_bf16(T, Hkv, D, seed=19), randn_bf16(T, Hkv, D, seed=20)
    ref = _attn_ref(q, k, v, True, cu)
    out = ops.attn_fwd(q, k, v, True, cu_seqlens=cu, max_seqlen=257)
    assert rel_l2(out, ref) < 8e-3, f"varlen rel={rel_l2(out, ref):.3e}"
    # uniform batches without cu_seqlens (the ViT path: B images x N tokens)
    B, N = 3, 196
    q, k, v = randn_bf16(B * N, 2, 72, seed=21), randn_bf16(B * N, 2, 72, seed=22), randn_bf16(B * N, 2, 72, seed=23)
    cu2 = torch.arange(0, B * N + 1, N, dtype=torch.int32, device="cuda")
    ref = _attn_ref(q, k, v, False, cu2)
    out = ops.attn_fwd(q, k, v, False, n_seq=B)
    assert rel_l2(out, ref) < 8e-3, f"batched rel={rel_l2(out, ref):.3e}"


def test_attention_running_max_jump(ops):
    """Force the online-softmax rescale branch: a late key dominates (guide rule 26): full-tensor reference."""
    T, H, D = 300, 2, 128
    q, k, v = randn_bf16(T, H, D, seed=24), randn_bf16(T, H, D, seed=25), randn_bf16(T, H, D, seed=26)
    k[200] = (q[250].float() * 1.5).to(torch.bfloat16)       # key 200 (4th KV tile) spikes against query 250
    k[70] = (q[10].float() * -2.0).to(torch.bfloat16)
    for causal in (False, True):
        ref = _attn_ref(q, k, v, causal)
        out = ops.attn_fwd(q, k, v, causal)
        assert rel_l2(out, ref) < 8e-3, f"causal={causal} rel={rel_l2(out, ref):.3e}"
        assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("N,K", [(3584, 3584), (4608, 3584), (3584, 18944), (1000, 512), (6, 64)])
def test_gemv_plain_and_fused(ops, N, K):
    x = randn_bf16(K, seed=27)
    w = randn_bf16(N, K, seed=28, scale=K ** -0.5)
    b = randn_bf16(N, seed=29)
    r = randn_bf16(N, seed=30)
    g = randn_bf16(K, seed=31, scale=0.1) + 1
    ref = w.float() @ x.float()
    out = ops.gemv(x, w, out_f32=True)
    assert rel_l2(out, ref) < 1e-4, f"f32 rel={rel_l2(out, ref):.3e}"
    out = ops.gemv(x, w, bias=b, residual=r)
    ref2 = (ref + b.float()).to(torch.bfloat16).float() + r.float()
    assert rel_l2(out, ref2) < 4e-3, f"bias+res rel={rel_l2(out, ref2):.3e}"
    x32 = x.float()
    xn = (g * (x32 * torch.rsqrt(x32.pow(2).mean() + 1e-6)).to(torch.bfloat16)).float()
    out = ops.gemv(x, w, norm_w=g, eps=1e-6, out_f32=True)
    assert rel_l2(out, w.float() @ xn) < 2e-3, f"norm rel={rel_l2(out, w.float() @ xn):.3e}"


def test_gemv_gateup(ops):
    K, N = 3584, 18944
    x = randn_bf16(K, seed=32)
    wg, wu = randn_bf16(N, K, seed=33, scale=K ** -0.5), randn_bf16(N, K, seed=34, scale=K ** -0.5)
    ref = torch.nn.functional.silu(wg.float() @ x.float()) * (wu.float() @ x.float())
    out = ops.gemv(x, wg, w2=wu)
    assert rel_l2(out, ref) < 8e-3, f"rel={rel_l2(out, ref):.3e}"


def test_argmax_first_maximum(ops):
    g = torch.Generator().manual_seed(35)
    x = torch.randn(152064, generator=g).cuda()
    assert int(ops.argmax(x)) == int(torch.argmax(x))
    x[1234] = 50.0
    x[99999] = 50.0
    assert int(ops.argmax(x)) == 1234          # ties -> first index
    y = torch.full((1000,), -3.0, device="cuda")
    assert int(ops.argmax(y)) == 0


def test_embed_and_copy_rows(ops):
    table = randn_bf16(1000, 512, seed=36)
    ids = torch.tensor([[3, 999, 0], [7, 7, 512]])
    out = ops.embed_tokens(table, ids)
    assert torch.equal(out, table[ids.cuda()])
    dst = torch.zeros(10, 512, device="cuda", dtype=torch.bfloat16)
    src_row = torch.tensor([5, 6, 900], dtype=torch.int32, device="cuda")
    dst_row = torch.tensor([9, 0, 4], dtype=torch.int32, device="cuda")
    ops.copy_rows(table, dst, src_row, dst_row, 3)
    assert torch.equal(dst[9], table[5]) and torch.equal(dst[0], table[6]) and torch.equal(dst[4], table[900])
    assert float(dst[1].float().abs().sum()) == 0


# ---------------------------------------------------------------------------------------------------------------------
# backward GEMMs on the forward tensors as they lie (contraction-major operands, vila_gemm_bf16_t)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [8, 4096 + 8, 3_000_000 + 16])
def test_grad_accum_f32_is_the_fp32_sum_rounded_once(ops, n):
    """vila_grad_accum_f32 (gradient accumulation over micro-batches): modes 0 / 1 keep the running sum in fp32 BIT-EXACTLY (bf16 -> fp32 is
    exact and fp32 addition is the same on both sides); mode 2 = bf16(acc + g) with one rounding, acc untouched, out aliasing g."""
    gs = [randn_bf16(n, seed=40 + k) * (10.0 ** -k) for k in range(9)]              # later micro-batches 1e-8 of the first: bf16 sums would absorb them
    acc = torch.full((n,), 7.0, device="cuda", dtype=torch.float32)                # mode 0 must overwrite whatever is there
    want = torch.zeros(n, device="cuda", dtype=torch.float32)
    for k, g in enumerate(gs[:-1]):
        ops.grad_accum(acc, g, mode=0 if k == 0 else 1)
        want = g.float() if k == 0 else want + g.float()
    assert torch.equal(acc, want)
    last = gs[-1].clone()
    keep = acc.clone()
    ops.grad_accum(acc, last, out=last, mode=2)
    assert torch.equal(acc, keep)
    assert torch.equal(last, (want + gs[-1].float()).to(torch.bfloat16))


@pytest.mark.parametrize("T,N,K", [(3076, 3584, 3584), (1538, 4608, 3584), (1000, 1032, 1496), (777, 520, 1032), (128, 128, 128), (3076, 18944, 3584)])
def test_gemm_t_dgrad_and_wgrad_match_fp32(ops, T, N, K):
    """dX = dY . W (W read as [contraction N][K]) and dW = dY^T . X (both read as [contraction T][rows]) against fp32 matmuls of the
    same bf16 tensors — no transposed copies anywhere.  Covers M / N tails of the 256x256 tiles, ragged contraction lengths (T = 777,
    3076: zero-filled k-rows) and the residual epilogue.  Tolerance: rel-L2 <= 4e-3 (one bf16 rounding of the output)."""
    x = randn_bf16(T, K, seed=11)
    w = randn_bf16(N, K, seed=12, scale=K ** -0.5)
    dy = randn_bf16(T, N, seed=13)
    res = randn_bf16(T, K, seed=14)
    dx = ops.gemm_t(dy, w, b_cm=True)
    ref_dx = dy.float() @ w.float()
    assert rel_l2(dx, ref_dx) < 4e-3, f"dgrad rel={rel_l2(dx, ref_dx):.3e}"
    dx = ops.gemm_t(dy, w, b_cm=True, residual=res)
    assert rel_l2(dx, ref_dx + res.float()) < 4e-3
    dw = torch.full((N, K), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm_t(dy, x, a_cm=True, b_cm=True, out=dw)
    ref_dw = dy.float().t() @ x.float()
    assert rel_l2(dw, ref_dw) < 4e-3, f"wgrad rel={rel_l2(dw, ref_dw):.3e}"
    # transpose-detecting: a swapped layout flag cannot pass on a non-symmetric problem (shape check or values)
    if N != K:
        with pytest.raises(ValueError):
            ops.gemm_t(dy, w, b_cm=False)


def test_gemm_t_splitk_and_a_cm_only(ops):
    """Under-filled grids with a workspace are sliced over K (the lm_head dgrad: 1024 x 3584 outputs, contraction 152064 ... here 19008);
    the a_cm-only combination (A^T . W^T with W in the forward layout) is covered as well."""
    ws = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    dy = randn_bf16(520, 19008, seed=21, scale=0.1)
    w = randn_bf16(19008, 1032, seed=22)
    out = ops.gemm_t(dy, w, b_cm=True, ws=ws)
    assert rel_l2(out, dy.float() @ w.float()) < 4e-3
    a = randn_bf16(2048, 520, seed=23)            # stored [K=2048][M=520]
    b = randn_bf16(264, 2048, seed=24, scale=2048 ** -0.5)
    out = ops.gemm_t(a, b, a_cm=True)
    assert rel_l2(out, a.float().t() @ b.float().t()) < 4e-3


def test_gemm_t_rejects_what_the_kernel_cannot_read(ops):
    dy, w = randn_bf16(256, 260, seed=1), randn_bf16(260, 516, seed=2)      # contraction-major rows must be a multiple of 8 (516 is not)
    with pytest.raises(ValueError):
        ops.gemm_t(dy, w, b_cm=True)


@pytest.mark.parametrize("sched", [1, 2, 3, 5, 6, 10])
def test_gemm256_every_kept_schedule_matches_fp32(ops, sched):
    """The K-loop schedules kept in gemm256_kernel.h (vila_gemm_force_sched): lock-step 0 (= 10) / 1 / 2 / 3 and the role-split 5 / 6 whose
    two wave groups run one barrier interval apart.  Forward layout (256x256 kernel pinned) and both backward layouts, ragged M / N / K:
    rows that are not a multiple of 256, a contraction of 23 K-tiles + a tail, odd K-tile counts (buffer parity at the loop end) and the
    two-tile minimum.  Tolerance: one bf16 rounding of the output (rel-L2 <= 4e-3)."""
    from vila_amd import _lib
    lib = _lib.load()
    try:
        lib.vila_gemm_force_sched(sched)
        lib.vila_gemm_force_tile(4)
        for (M, N, K) in [(1000, 1032, 1496), (769, 520, 128), (512, 512, 192), (300, 264, 4160)]:
            a = randn_bf16(M, K, seed=31)
            w = randn_bf16(N, K, seed=32, scale=K ** -0.5)
            res = randn_bf16(M, N, seed=33)
            out = ops.gemm(a, w, residual=res)
            assert rel_l2(out, a.float() @ w.float().t() + res.float()) < 4e-3, (sched, M, N, K)
        lib.vila_gemm_force_tile(0)
        if sched in (1, 2, 5, 6):
            for (T, N, K) in [(1000, 1032, 1496), (777, 520, 1032), (192, 256, 136)]:
                x = randn_bf16(T, K, seed=34)
                w = randn_bf16(N, K, seed=35, scale=K ** -0.5)
                dy = randn_bf16(T, N, seed=36)
                assert rel_l2(ops.gemm_t(dy, w, b_cm=True), dy.float() @ w.float()) < 4e-3, (sched, "dgrad", T, N, K)
                assert rel_l2(ops.gemm_t(dy, x, a_cm=True, b_cm=True), dy.float().t() @ x.float()) < 4e-3, (sched, "wgrad", T, N, K)
    finally:
        lib.vila_gemm_force_sched(0)
        lib.vila_gemm_force_tile(0)


def test_gemm_whole_rounds_plus_sliced_tail_tiles(ops):
    """Tile quantisation path (gemm256.hip try_hybrid): 17 x 16 = 272 tiles of 256^2 = one whole round + 16 tail tiles, which are sliced
    over K into compact per-tile slabs and finished (bias-free, residual) by the tail reduce kernel.  Forward and both backward layouts,
    with the policy on (workspace lent) and off (hook) giving the same result up to the summation order."""
    from vila_amd import _lib
    lib = _lib.load()
    ws = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    M, N, K = 4300, 4092, 1088                                 # 17 x 16 tiles, ragged last row tile and last column tile
    a = randn_bf16(M, K, seed=41)
    w = randn_bf16(N, K, seed=42, scale=K ** -0.5)
    res = randn_bf16(M, N, seed=43)
    ref = a.float() @ w.float().t() + res.float()
    on = ops.gemm(a, w, residual=res, ws=ws)
    assert rel_l2(on, ref) < 4e-3
    try:
        lib.vila_gemm_force_hybrid(0)
        off = ops.gemm(a, w, residual=res, ws=ws)
    finally:
        lib.vila_gemm_force_hybrid(1)
    assert rel_l2(off, ref) < 4e-3 and rel_l2(on, off.float()) < 4e-3
    # wgrad: dW[N2, K2] = dY^T X with N2 x K2 = 4352 x 4096 outputs (272 tiles), contraction T = 1100 (ragged)
    T = 1100
    dy = randn_bf16(T, 4352, seed=44, scale=T ** -0.5)
    x = randn_bf16(T, 4096, seed=45)
    dw = ops.gemm_t(dy, x, a_cm=True, b_cm=True, ws=ws)
    assert rel_l2(dw, dy.float().t() @ x.float()) < 4e-3
    # dgrad with residual: dX[4300, 4096] = dY[4300, 1088] . W[1088, 4096]
    dyy = randn_bf16(4300, 1088, seed=46, scale=1088 ** -0.5)
    ww = randn_bf16(1088, 4096, seed=47)
    r2 = randn_bf16(4300, 4096, seed=48)
    dx = ops.gemm_t(dyy, ww, b_cm=True, residual=r2, ws=ws)
    assert rel_l2(dx, dyy.float() @ ww.float() + r2.float()) < 4e-3


@pytest.mark.parametrize("grp", [-1, 4, 3, 2, 0])
def test_gemm256_grouped_tile_order_is_a_bijection_on_ragged_grids(ops, grp):
    """Round 3: tiles are walked in 8 x 4 patches per XCD (columns grouped by `grp`, gemm256_kernel.h gemm256_tile_of) instead of 32 x 1 strips.
    Every order must still produce every output tile exactly once: ragged grids (19 row tiles x 4 column tiles with a 232-wide last column — the
    last group is narrower than grp = 3), forward and contraction-major layouts, and the whole-rounds + K-sliced-tail launch whose reduce kernel
    decodes tile ids with the same function (350 tiles = 256 + 94)."""
    from vila_amd import _lib
    lib = _lib.load()
    M, N, K = 4616, 1000, 320
    a, w = randn_bf16(M, K, seed=91), randn_bf16(N, K, seed=92, scale=K ** -0.5)
    res = randn_bf16(M, N, seed=93)
    ref = a.float() @ w.float().t() + res.float()
    lib.vila_gemm_force_group(grp)
    lib.vila_gemm_force_tile(4)
    try:
        out = ops.gemm(a, w, residual=res)
        assert rel_l2(out, ref) < 4e-3, f"forward layout: {rel_l2(out, ref):.3e}"
        w_up = randn_bf16(N, K, seed=96, scale=K ** -0.5)                 # fused gate/up over the whole grid (19 x 8 tiles of 128 columns)
        want = torch.nn.functional.silu(a.float() @ w.float().t()) * (a.float() @ w_up.float().t())
        out = ops.gemm(a, w, w2=w_up, epi=3)
        assert rel_l2(out, want) < 6e-3, f"gate/up: {rel_l2(out, want):.3e}"
        lib.vila_gemm_force_tile(0)
        at, wt = a.t().contiguous(), w.t().contiguous()                  # stored contraction-major
        out = ops.gemm_t(at, wt, a_cm=True, b_cm=True, residual=res)
        assert rel_l2(out, ref) < 4e-3, f"both operands contraction-major: {rel_l2(out, ref):.3e}"
        out = ops.gemm_t(a, wt, b_cm=True, residual=res)
        assert rel_l2(out, ref) < 4e-3, f"dgrad layout: {rel_l2(out, ref):.3e}"
        # whole rounds + K-sliced tail tiles (needs a workspace): 70 x 5 = 350 tiles
        M2, N2, K2 = 17920, 1032, 1024
        a2, w2 = randn_bf16(M2, K2, seed=94), randn_bf16(N2, K2, seed=95, scale=K2 ** -0.5)
        ws = torch.empty(8 * 96 * 65536, device="cuda", dtype=torch.float32)
        ref2 = a2.float() @ w2.float().t()
        out2 = ops.gemm_t(a2.t().contiguous(), w2.t().contiguous(), a_cm=True, b_cm=True, ws=ws)
        assert rel_l2(out2, ref2) < 4e-3, f"rounds + sliced tail, contraction-major: {rel_l2(out2, ref2):.3e}"
        assert float((out2.float() - ref2).abs().max()) < 0.25
        lib.vila_gemm_force_tile(4)
        out2 = ops.gemm(a2, w2, ws=ws)
        assert rel_l2(out2, ref2) < 4e-3, f"rounds + sliced tail, forward layout: {rel_l2(out2, ref2):.3e}"
    finally:
        lib.vila_gemm_force_tile(0)
        lib.vila_gemm_force_group(-1)


@pytest.mark.parametrize("M", [257, 260, 272, 769, 3076])
def test_gemm256_leftover_rows_ride_in_the_last_row_tile(ops, M):
    """M = 256 k + r, 1 <= r <= 16: the last 256-row tile carries the r rows as an extra 16-row fragment (gemm256_kernel.h, EX) in every
    mode that can meet such a shape on the path: plain / bias / residual / GELU / fp32 out, fused gate/up (whole rounds + K-sliced tail
    round), split-K slabs.  Every row — the leftover ones separately — against the fp32 reference; and equal to the policy-off result."""
    from vila_amd import _lib
    lib = _lib.load()
    K, N = 1024, 3584
    a = randn_bf16(M, K, seed=71)
    w, w2 = randn_bf16(N, K, seed=72, scale=K ** -0.5), randn_bf16(N, K, seed=73, scale=K ** -0.5)
    bias, res = randn_bf16(N, seed=74), randn_bf16(M, N, seed=75)
    ref = a.float() @ w.float().t()
    lo = (M // 256) * 256
    lib.vila_gemm_force_tile(4)                                    # pin the 256x256 kernel (the dispatcher would take other tiles for small M)
    lib.vila_gemm_force_ex(2)                                      # ... and its extra-fragment variant (default policy: only when it saves a round)
    try:
        for name, out, want, tol in (
                ("plain+bias+res", ops.gemm(a, w, bias=bias, residual=res), ref + bias.float() + res.float(), 4e-3),
                ("gelu", ops.gemm(a, w, bias=bias, epi=1), torch.nn.functional.gelu(ref + bias.float(), approximate="tanh"), 5e-3),
                ("fp32", ops.gemm(a, w, out_f32=True), ref, 2e-5)):
            assert rel_l2(out, want) < tol, f"{name}: rel={rel_l2(out, want):.3e}"
            assert rel_l2(out[lo:], want[lo:]) < tol, f"{name}: leftover rows rel={rel_l2(out[lo:], want[lo:]):.3e}"
            assert rel_l2(out[lo - 16:lo], want[lo - 16:lo]) < tol
    finally:
        lib.vila_gemm_force_tile(0)
        lib.vila_gemm_force_ex(-1)
    # fused gate/up (N = 18944: whole rounds + tail round split over K with a workspace) and the split-K slabs (K = 18944)
    F = 18944
    wg, wu = randn_bf16(F, K, seed=76, scale=K ** -0.5), randn_bf16(F, K, seed=77, scale=K ** -0.5)
    ws = torch.empty(8 * M * 4608, device="cuda", dtype=torch.float32)
    g = a.float() @ wg.float().t()
    want = torch.nn.functional.silu(g) * (a.float() @ wu.float().t())
    lib.vila_gemm_force_tile(4)
    try:
        for kw in (dict(), dict(ws=ws)):
            out = ops.gemm(a, wg, w2=wu, epi=3, **kw)
            assert rel_l2(out, want) < 6e-3 and rel_l2(out[lo:], want[lo:]) < 6e-3, f"gate/up {list(kw)}: {rel_l2(out, want):.3e} / {rel_l2(out[lo:], want[lo:]):.3e}"
    finally:
        lib.vila_gemm_force_tile(0)
    if M >= 512:
        a2, wd = randn_bf16(M, F, seed=78, scale=0.5), randn_bf16(N, F, seed=79, scale=F ** -0.5)
        want = a2.float() @ wd.float().t() + res.float()
        out = ops.gemm(a2, wd, residual=res, ws=ws)                # under-filled grid + workspace -> K-sliced slabs + reduce
        assert rel_l2(out, want) < 4e-3 and rel_l2(out[lo:], want[lo:]) < 4e-3, f"split-K: {rel_l2(out, want):.3e} / {rel_l2(out[lo:], want[lo:]):.3e}"
