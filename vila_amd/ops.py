"""Operator-level Python wrappers over the C-ABI (device pointers + the current HIP stream).  PyTorch only owns
the buffers.  Every op raises on CPU tensors: there is no fallback path."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check

EPI_NONE, EPI_GELU_TANH, EPI_GELU_ERF, EPI_GATEUP = 0, 1, 2, 3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype=torch.bfloat16, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise _lib.VilaHipError(f"{name} must live on the GPU: vila_amd has no CPU path (got device {t.device})")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         epi: int = EPI_NONE, w2: Optional[torch.Tensor] = None, out_f32: bool = False,
         out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias) + residual   (nn.Linear semantics; EPI_GATEUP: silu(a w^T) * (a w2^T))."""
    _need(a, name="a"); _need(w, name="w")
    # w may carry extra zero-padded columns (transpose() pads the contraction dim to 64): K is a's width
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    if a.shape[1] != w.shape[1] and w.shape[1] != (a.shape[1] + 63) // 64 * 64:
        raise ValueError(f"gemm: contraction widths differ (a: {a.shape[1]}, w: {w.shape[1]}); only the zero padding of transpose() to a "
                         "multiple of 64 is accepted")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    if residual is not None:
        _need(residual, name="residual"); assert residual.shape == (M, N) and residual.stride(1) == 1
    if ws is not None:   # fp32 workspace: allows split-K on under-filled grids
        check(_lib.load().vila_gemm_bf16_ws(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(w2), _p(bias), _p(residual),
                                            residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0),
                                            1 if out_f32 else 0, M, N, K, epi, ws.data_ptr(), ws.numel() * ws.element_size(), _stream()),
              "vila_gemm_bf16_ws")
        return out
    check(_lib.load().vila_gemm_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(w2), _p(bias), _p(residual),
                                     residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0),
                                     1 if out_f32 else 0, M, N, K, epi, _stream()), "vila_gemm_bf16")
    return out


def gemm_t(a: torch.Tensor, w: torch.Tensor, a_cm: bool = False, b_cm: bool = False, bias: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = A . B^T (+bias)(+residual) with operands read as they lie: a_cm -> `a` is stored [K, M], b_cm -> `w` is stored [K, N]
    (vila_gemm_bf16_t).  dgrad: gemm_t(dy, W, b_cm=True);  wgrad: gemm_t(dy, x, a_cm=True, b_cm=True)."""
    _need(a, name="a"); _need(w, name="w")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if a_cm else a.shape
    N, Kw = (w.shape[1], w.shape[0]) if b_cm else w.shape
    if K != Kw:
        raise ValueError(f"gemm_t: contraction widths differ ({K} vs {Kw})")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    assert out.shape == (M, N) and out.stride(1) == 1
    if residual is not None:
        _need(residual, name="residual"); assert residual.shape == (M, N) and residual.stride(1) == 1
    check(_lib.load().vila_gemm_bf16_t(a.data_ptr(), a.stride(0), int(a_cm), w.data_ptr(), w.stride(0), int(b_cm), _p(bias), _p(residual),
                                       residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), M, N, K,
                                       _p(ws), ws.numel() * ws.element_size() if ws is not None else 0, _stream()), "vila_gemm_bf16_t")
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    _need(x, name="x")
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    y = torch.empty_like(x2)
    check(_lib.load().vila_layernorm_bf16(x2.data_ptr(), w.data_ptr(), _p(b), y.data_ptr(), x2.shape[0], x2.shape[1], eps, _stream()), "layernorm")
    return y.view(x.shape)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    _need(x, name="x")
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    y = torch.empty_like(x2)
    check(_lib.load().vila_rmsnorm_bf16(x2.data_ptr(), w.data_ptr(), y.data_ptr(), x2.shape[0], x2.shape[1], eps, _stream()), "rmsnorm")
    return y.view(x.shape)


def space_to_depth(x: torch.Tensor, k: int) -> torch.Tensor:
    """[B, g*g, C] -> [B, ceil(g/k)^2, k*k*C]  (flat_square / flat_square_2x2 / flat_square_3x3)."""
    _need(x, name="x")
    B, N, Cc = x.shape
    g = int(round(N ** 0.5))
    assert g * g == N
    gd = (g + k - 1) // k
    y = torch.empty((B, gd * gd, k * k * Cc), device=x.device, dtype=x.dtype)
    check(_lib.load().vila_space_to_depth_bf16(x.contiguous().data_ptr(), y.data_ptr(), B, g, Cc, k, _stream()), "space_to_depth")
    return y


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool, scale: Optional[float] = None,
             cu_seqlens: Optional[torch.Tensor] = None, max_seqlen: Optional[int] = None, n_seq: int = 1,
             return_lse: bool = False):
    """q [T, Hq, D], k/v [T, Hkv, D] (last dim contiguous; token/head strides free) -> o [T, Hq, D] (+ lse [Hq, T])."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, name=n); assert t.dim() == 3 and t.stride(2) == 1
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    if scale is None:
        scale = D ** -0.5
    if cu_seqlens is not None:
        assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_cuda
        n_seq = cu_seqlens.numel() - 1
        assert max_seqlen is not None
    else:
        max_seqlen = T // n_seq
    o = torch.empty((T, Hq, D), device=q.device, dtype=q.dtype)
    lse = torch.empty((Hq, T), device=q.device, dtype=torch.float32) if return_lse else None
    check(_lib.load().vila_attn_fwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), q.stride(0), k.stride(0),
                                         v.stride(0), o.stride(0), q.stride(1), k.stride(1), v.stride(1), o.stride(1),
                                         _p(cu_seqlens), n_seq, T, max_seqlen, Hq, Hkv, D, 1 if causal else 0, float(scale),
                                         _p(lse), _stream()), "attn_fwd")
    return (o, lse) if return_lse else o


def gemv(x: torch.Tensor, w: torch.Tensor, norm_w: Optional[torch.Tensor] = None, eps: float = 0.0,
         w2: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         out_f32: bool = False) -> torch.Tensor:
    _need(x, name="x"); _need(w, name="w")
    N, K = w.shape
    y = torch.empty((N,), device=x.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    check(_lib.load().vila_gemv_bf16(x.data_ptr(), _p(norm_w), eps, w.data_ptr(), _p(w2), _p(bias), _p(residual),
                                     None if out_f32 else y.data_ptr(), y.data_ptr() if out_f32 else None, N, K,
                                     1 if w2 is not None else 0, _stream()), "gemv")
    return y


def gemv_w4(x: torch.Tensor, mat, norm_w: Optional[torch.Tensor] = None, eps: float = 0.0, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W4A16 GEMV on a vila_amd.quant.W4Matrix (tile-major int4, group 128); gate/up matrices give silu(Wg x) * (Wu x)."""
    _need(x, name="x"); _need(mat.q, dtype=torch.int32, name="mat.q"); _need(mat.sz, dtype=torch.int32, name="mat.sz")
    y = torch.empty((mat.N,), device=x.device, dtype=torch.bfloat16)
    check(_lib.load().vila_gemv_w4_bf16(x.data_ptr(), _p(norm_w), eps, mat.q.data_ptr(), mat.sz.data_ptr(), _p(bias), _p(residual),
                                        y.data_ptr(), mat.N, mat.K, mat.mode, _stream()), "gemv_w4")
    return y


def argmax(logits: torch.Tensor) -> torch.Tensor:
    _need(logits, dtype=torch.float32, name="logits")
    out = torch.empty((1,), device=logits.device, dtype=torch.int64)
    ws = torch.empty((4096,), device=logits.device, dtype=torch.uint8)
    check(_lib.load().vila_argmax_f32(logits.data_ptr(), logits.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "argmax")
    return out


def sample(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 50, top_p: float = 1.0, seed: int = 0,
           counter: Optional[torch.Tensor] = None, return_dist: bool = False):
    """HF sampling on the device (GenerationMixin.sample): logits / temperature -> TopK (any k >= 1; 0 = no top-k filter, as HF treats it) ->
    TopP -> softmax -> draw with splitmix64(seed, *counter).  -> token id [1] i64 (and, on request, the distribution that was sampled as
    (probabilities, token ids): 64 slots for top_k in 1..64, the dense vocabulary otherwise)."""
    _need(logits, dtype=torch.float32, name="logits")
    lib = _lib.load()
    out = torch.empty((1,), device=logits.device, dtype=torch.int64)
    ws = torch.empty((lib.vila_sample_workspace_bytes(),), device=logits.device, dtype=torch.uint8)
    small = 1 <= int(top_k) <= 64
    dist = torch.zeros((128 if small else logits.numel(),), device=logits.device, dtype=torch.float32) if return_dist else None
    sp = _lib.VilaSampling(float(temperature), int(top_k), float(top_p), int(seed) & 0xFFFFFFFFFFFFFFFF)
    import ctypes as C
    check(lib.vila_sample_f32(logits.data_ptr(), logits.numel(), C.byref(sp), _p(counter), out.data_ptr(), ws.data_ptr(), _p(dist), _stream()), "sample")
    if return_dist:
        if small:
            return out, dist[:64].clone(), dist[64:].view(torch.int32).clone()
        return out, dist, torch.arange(logits.numel(), device=logits.device, dtype=torch.int32)
    return out


def embed_tokens(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    _need(table, name="embed table")
    ids = ids.to(device=table.device, dtype=torch.int64).contiguous()
    out = torch.empty((*ids.shape, table.shape[1]), device=table.device, dtype=table.dtype)
    check(_lib.load().vila_embed_tokens(table.data_ptr(), table.shape[0], table.shape[1], ids.data_ptr(), ids.numel(),
                                        out.data_ptr(), _stream()), "embed_tokens")
    return out


def copy_rows(src: torch.Tensor, dst: torch.Tensor, src_row: Optional[torch.Tensor], dst_row: Optional[torch.Tensor], n: int) -> None:
    _need(src, name="src"); _need(dst, name="dst")
    for t in (src_row, dst_row):
        assert t is None or (t.dtype == torch.int32 and t.is_cuda)
    check(_lib.load().vila_copy_rows(src.data_ptr(), dst.data_ptr(), _p(src_row), _p(dst_row), n, src.shape[-1], _stream()), "copy_rows")


def quant_rows_i8(x: torch.Tensor):
    """bf16 [rows, cols] -> (int8 [rows, cols], fp32 scale [rows]): per-token dynamic symmetric quantisation (scale = max|row| / 127)."""
    _need(x, name="x")
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    q = torch.empty(x2.shape, device=x.device, dtype=torch.int8)
    sc = torch.empty((x2.shape[0],), device=x.device, dtype=torch.float32)
    check(_lib.load().vila_quant_rows_i8(x2.data_ptr(), q.data_ptr(), sc.data_ptr(), x2.shape[0], x2.shape[1], _stream()), "quant_rows_i8")
    return q, sc


def gemm_w8a8(aq: torch.Tensor, sx: torch.Tensor, wq: torch.Tensor, sw: torch.Tensor, bias: Optional[torch.Tensor] = None,
              residual: Optional[torch.Tensor] = None, epi: int = EPI_NONE) -> torch.Tensor:
    """out[M,N] bf16 = epi((aq[M,K] . wq[N,K]^T) * sx[m] * sw[n] + bias) (+ residual); int8 operands, int32 accumulate."""
    assert aq.dtype == torch.int8 and wq.dtype == torch.int8 and aq.is_cuda and aq.stride(1) == 1 and wq.stride(1) == 1
    M, K = aq.shape
    N = wq.shape[0]
    out = torch.empty((M, N), device=aq.device, dtype=torch.bfloat16)
    check(_lib.load().vila_gemm_w8a8(aq.data_ptr(), aq.stride(0), wq.data_ptr(), wq.stride(0), sx.data_ptr(), sw.data_ptr(), _p(bias), _p(residual),
                                     residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), M, N, K, epi, _stream()),
          "vila_gemm_w8a8")
    return out


def video_pool(feats: torch.Tensor, pool, start_rows: Optional[torch.Tensor] = None, end_rows: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """feats [nt, nl*nl, C] (one video's projected frames) -> [(nt/pt) * (n_start + (nl/ph)(nl/pw) + n_end), C]: TSPVideoEncoder's
    pool + _process_features (video/tsp.py:10-11,28-52); pool (1,1,1) = BasicVideoEncoder (video/basic.py:30-41)."""
    _need(feats, name="feats")
    nt, ns, Cc = feats.shape
    nl = int(round(ns ** 0.5))
    assert nl * nl == ns
    pt, ph, pw = (int(p) for p in pool)
    n_s = 0 if start_rows is None else start_rows.shape[0]
    n_e = 0 if end_rows is None else end_rows.shape[0]
    if pt <= 0 or ph <= 0 or pw <= 0 or nt % pt or nl % ph or nl % pw:
        raise ValueError(f"shape '[{nt}, {nl}, {nl}]' is invalid for pooling by ({pt}, {ph}, {pw}): every pooled dimension must divide evenly")
    rows = (nt // pt) * (n_s + (nl // ph) * (nl // pw) + n_e)
    if out is None:
        out = torch.empty((rows, Cc), device=feats.device, dtype=feats.dtype)
    assert out.shape == (rows, Cc) and out.is_contiguous()
    check(_lib.load().vila_video_pool_bf16(feats.contiguous().data_ptr(), out.data_ptr(), nt, nl, Cc, pt, ph, pw,
                                           _p(start_rows.contiguous() if start_rows is not None else None), n_s,
                                           _p(end_rows.contiguous() if end_rows is not None else None), n_e, _stream()), "video_pool")
    return out


def video_pool_bwd(dpooled: torch.Tensor, nt: int, nl: int, pool, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """Adjoint of video_pool's feature rows: dpooled [(nt/pt)(nl/ph)(nl/pw), C] -> dfeats [nt, nl*nl, C] (the `mean` backward of tsp.py:10-11);
    accumulate adds into `out` (the later pool sizes of one video)."""
    _need(dpooled, name="dpooled")
    pt, ph, pw = (int(p) for p in pool)
    Cc = dpooled.shape[-1]
    assert dpooled.is_contiguous() and dpooled.numel() == (nt // pt) * (nl // ph) * (nl // pw) * Cc
    if out is None:
        assert not accumulate
        out = torch.empty((nt, nl * nl, Cc), device=dpooled.device, dtype=dpooled.dtype)
    assert out.is_contiguous() and out.numel() == nt * nl * nl * Cc
    check(_lib.load().vila_video_pool_bwd_bf16(dpooled.data_ptr(), out.data_ptr(), nt, nl, Cc, pt, ph, pw, int(accumulate), _stream()), "video_pool_bwd")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# SFT-step operators (backward + optimizer)
# ----------------------------------------------------------------------------------------------------------------------
def _L():
    return _lib.load()


def transpose(x: torch.Tensor) -> torch.Tensor:
    """[R, C] -> [C, round_up(R, 64)] (zero-padded): the K-contiguous operand form gemm() needs for dgrad / wgrad
    (64 = the K-tile of the 256x256 kernel, so the padded token dimension never needs a K tail)."""
    _need(x, name="x"); assert x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    Rp = (R + 63) // 64 * 64
    out = torch.empty((Cc, Rp), device=x.device, dtype=x.dtype)
    check(_L().vila_transpose_bf16(x.data_ptr(), out.data_ptr(), R, Cc, x.stride(0), Rp, _stream()), "transpose")
    return out


def act_fwd(z: torch.Tensor, act: int) -> torch.Tensor:
    y = torch.empty_like(z)
    check(_L().vila_act_fwd_bf16(z.data_ptr(), y.data_ptr(), z.numel(), act, _stream()), "act_fwd")
    return y


def act_bwd(z: torch.Tensor, dy: torch.Tensor, act: int) -> torch.Tensor:
    dz = torch.empty_like(z)
    check(_L().vila_act_bwd_bf16(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), z.numel(), act, _stream()), "act_bwd")
    return dz


def silu_mul(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    a = torch.empty_like(g)
    check(_L().vila_silu_mul_fwd_bf16(g.data_ptr(), u.data_ptr(), a.data_ptr(), g.numel(), _stream()), "silu_mul")
    return a


def silu_mul_bwd(g, u, da):
    dg, du = torch.empty_like(g), torch.empty_like(u)
    check(_L().vila_silu_mul_bwd_bf16(g.data_ptr(), u.data_ptr(), da.data_ptr(), dg.data_ptr(), du.data_ptr(), g.numel(), _stream()), "silu_mul_bwd")
    return dg, du


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(a) if out is None else out
    check(_L().vila_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "add")
    return out


def grad_accum(acc: torch.Tensor, g: torch.Tensor, out: Optional[torch.Tensor] = None, mode: int = 1) -> None:
    """Gradient accumulation in fp32 (vila_grad_accum_f32): mode 0 acc = g, 1 acc += g, 2 out = bf16(acc + g)."""
    assert acc.dtype == torch.float32 and g.dtype == torch.bfloat16 and acc.numel() == g.numel() and acc.is_contiguous() and g.is_contiguous()
    assert int(mode) in (0, 1, 2) and g.numel() % 8 == 0, (mode, g.numel())
    if int(mode) == 2:       # a mode-2 call without (or with a mis-shaped) output used to be caught only inside the library
        assert out is not None and out.dtype == torch.bfloat16 and out.numel() == g.numel() and out.is_contiguous() and out.device == g.device
    check(_L().vila_grad_accum_f32(acc.data_ptr(), g.data_ptr(), _p(out), g.numel(), int(mode), _stream()), "grad_accum")


def colsum(x: torch.Tensor, out: torch.Tensor, accumulate: bool = False, period: int = 0) -> None:
    assert x.dim() == 2 and x.stride(1) == 1 and out.is_contiguous()
    scratch = torch.empty((int(_L().vila_colsum_scratch_floats(x.shape[0], x.shape[1])),), device=x.device, dtype=torch.float32) if period == 0 else None
    check(_L().vila_colsum_bf16(x.data_ptr(), out.data_ptr(), _p(scratch), x.shape[0], x.shape[1], x.stride(0), int(accumulate), period, _stream()), "colsum")


def norm_bwd(x, w, dy, dw_out, db_out, eps: float, rms: bool, accumulate: bool = False) -> torch.Tensor:
    x2, dy2 = x.reshape(-1, x.shape[-1]), dy.reshape(-1, x.shape[-1])
    dx = torch.empty_like(x2)
    scratch = torch.empty((int(_L().vila_norm_bwd_scratch_floats(x2.shape[0], x2.shape[1])),), device=x.device, dtype=torch.float32)
    check(_L().vila_norm_bwd_bf16(x2.data_ptr(), w.data_ptr(), dy2.data_ptr(), dx.data_ptr(), dw_out.data_ptr(), _p(db_out),
                                  scratch.data_ptr(), x2.shape[0], x2.shape[1], eps, int(rms), int(accumulate), _stream()), "norm_bwd")
    return dx.view(x.shape)


def ce_loss(logits: torch.Tensor, labels: torch.Tensor, loss_acc: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """logits [n, V] fp32, labels [n] i64 -> dlogits [n, V] bf16 (into `out` when given); loss_acc (fp32 scalar on device) += sum CE * scale."""
    _need(logits, dtype=torch.float32, name="logits")
    n, V = logits.shape
    d = torch.empty((n, V), device=logits.device, dtype=torch.bfloat16) if out is None else out
    assert d.shape == (n, V) and d.dtype == torch.bfloat16 and d.is_contiguous() and labels.is_contiguous()
    row_loss = torch.empty((max(n, 1),), device=logits.device, dtype=torch.float32)
    check(_L().vila_ce_loss_f32(logits.data_ptr(), labels.data_ptr(), d.data_ptr(), loss_acc.data_ptr(), row_loss.data_ptr(), n, V, logits.stride(0), scale,
                                _stream()), "ce_loss")
    return d


def scatter_add_rows(src: torch.Tensor, dst: torch.Tensor, rows: torch.Tensor) -> None:
    assert rows.dtype == torch.int32 and src.is_contiguous() and dst.is_contiguous()
    check(_L().vila_scatter_add_rows_bf16(src.data_ptr(), dst.data_ptr(), rows.data_ptr(), rows.numel(), src.shape[-1], _stream()), "scatter_add_rows")


def depth_to_space(dy: torch.Tensor, g: int, k: int) -> torch.Tensor:
    B, _, CC = dy.shape
    Cc = CC // (k * k)
    dx = torch.empty((B, g * g, Cc), device=dy.device, dtype=dy.dtype)
    check(_L().vila_depth_to_space_bf16(dy.contiguous().data_ptr(), dx.data_ptr(), B, g, Cc, k, _stream()), "depth_to_space")
    return dx


def im2col(px: torch.Tensor, patch: int, kp: int) -> torch.Tensor:
    B, Cc, H, W = px.shape
    out = torch.empty((B * (H // patch) * (W // patch), kp), device=px.device, dtype=px.dtype)
    check(_L().vila_im2col_bf16(px.contiguous().data_ptr(), out.data_ptr(), B, Cc, H, W, patch, kp, _stream()), "im2col")
    return out


def rope_table(positions: torch.Tensor, head_dim: int, theta: float):
    S = positions.numel()
    cs = torch.empty((S, head_dim // 2), device=positions.device, dtype=torch.float32)
    sn = torch.empty_like(cs)
    check(_L().vila_rope_table_f32(positions.data_ptr(), cs.data_ptr(), sn.data_ptr(), S, head_dim, theta, _stream()), "rope_table")
    return cs, sn


def rope_fwd_(qkv: torch.Tensor, cs, sn, positions, nq: int, nkv: int, hd: int) -> None:
    check(_L().vila_rope_fwd_bf16(qkv.data_ptr(), cs.data_ptr(), sn.data_ptr(), positions.data_ptr(), qkv.shape[0], nq, nkv, hd, _stream()), "rope_fwd")


def rope_bwd_(dqkv: torch.Tensor, cs, sn, nq: int, nkv: int, hd: int) -> None:
    check(_L().vila_rope_bwd_bf16(dqkv.data_ptr(), cs.data_ptr(), sn.data_ptr(), dqkv.shape[0], nq, nkv, hd, _stream()), "rope_bwd")


def attn_bwd(q, k, v, o, do, lse, causal: bool, dq, dk, dv, scale: Optional[float] = None, cu_seqlens=None, max_seqlen=None, n_seq: int = 1,
             parts: int = 7, delta: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All of q,k,v,o,do,dq,dk,dv are [T, H, D] views (last dim contiguous).  lse [Hq, T] fp32 from attn_fwd(return_lse=True).
    parts (bit mask): 1 = delta = rowsum(dO o O), 2 = dQ, 4 = dK / dV; the default does all three.  Returns delta [Hq, T] fp32, to be
    passed back in when the parts are launched separately (dQ and dK / dV share no output and may run on different streams)."""
    import ctypes as C
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    ts = (C.c_int64 * 8)(*[t.stride(0) for t in (q, k, v, o, do, dq, dk, dv)])
    hs = (C.c_int32 * 8)(*[t.stride(1) for t in (q, k, v, o, do, dq, dk, dv)])
    if cu_seqlens is not None:
        n_seq = cu_seqlens.numel() - 1
    else:
        max_seqlen = T // n_seq
    if delta is None:
        if not parts & 1:
            raise ValueError("attn_bwd: parts without 1 (delta) need the delta tensor of an earlier call")
        delta = torch.empty((Hq, T), device=q.device, dtype=torch.float32)
    check(_L().vila_attn_bwd_bf16_parts(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                        dv.data_ptr(), ts, hs, _p(cu_seqlens), n_seq, T, int(max_seqlen), Hq, Hkv, D, int(causal),
                                        float(scale if scale is not None else D ** -0.5), lse.data_ptr(), delta.data_ptr(), int(parts), _stream()),
          "attn_bwd")
    return delta


def adamw_step(master, m, v, grad, param, lr, beta1, beta2, eps, wd, step: int, grad_scale: float = 1.0, lean: bool = False) -> None:
    """torch.optim.AdamW update on flat buffers.  lean: the <= 32-VGPR kernel that fits beside a resident 256x256 GEMM block (side streams)."""
    if lean:
        check(_L().vila_adamw_step_lean(master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), param.data_ptr(), master.numel(), lr, beta1,
                                        beta2, eps, wd, step, grad_scale, _stream()), "adamw_lean")
        return
    check(_L().vila_adamw_step(master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), param.data_ptr(), master.numel(), lr, beta1, beta2,
                               eps, wd, step, grad_scale, _stream()), "adamw")


def sumsq(x: torch.Tensor) -> torch.Tensor:
    out = torch.zeros((1,), device=x.device, dtype=torch.float32)
    scratch = torch.empty((2048,), device=x.device, dtype=torch.float32)          # VILA_SUMSQ_SCRATCH_FLOATS
    check(_L().vila_sumsq_bf16(x.data_ptr(), x.numel(), out.data_ptr(), scratch.data_ptr(), _stream()), "sumsq")
    return out


def s2_merge(feats: torch.Tensor, desc: torch.Tensor, n_scales: int, splits) -> torch.Tensor:
    """dynamic_s2: tower output [n_tiles, N, C] + block descriptors [n_blocks, 6] i32 -> projector input [n_blocks, N, n_scales*C]."""
    import ctypes as C
    _need(feats, name="feats")
    assert desc.dtype == torch.int32 and desc.is_cuda and desc.dim() == 2 and desc.shape[1] == 6
    n_tiles, N, Cc = feats.shape
    g = int(round(N ** 0.5))
    out = torch.empty((desc.shape[0], N, n_scales * Cc), device=feats.device, dtype=feats.dtype)
    sp = (C.c_int32 * max(len(splits), 1))(*splits) if len(splits) else (C.c_int32 * 1)(1)
    check(_L().vila_s2_merge_bf16(feats.contiguous().data_ptr(), out.data_ptr(), desc.contiguous().data_ptr(), desc.shape[0], g, Cc, n_scales, sp,
                                  _stream()), "s2_merge")
    return out


def s2_merge_bwd(dy: torch.Tensor, tile_desc: torch.Tensor, n_scales: int, splits) -> torch.Tensor:
    """Adjoint of s2_merge: dy [n_blocks, N, n_scales*C] + per-tile descriptors [n_tiles, 8] i32 (host.s2_plan) -> dfeats [n_tiles, N, C]."""
    import ctypes as C
    _need(dy, name="dy")
    assert tile_desc.dtype == torch.int32 and tile_desc.is_cuda and tile_desc.dim() == 2 and tile_desc.shape[1] == 8
    _, N, CC = dy.shape
    Cc = CC // n_scales
    g = int(round(N ** 0.5))
    dx = torch.empty((tile_desc.shape[0], N, Cc), device=dy.device, dtype=dy.dtype)
    sp = (C.c_int32 * max(len(splits), 1))(*splits) if len(splits) else (C.c_int32 * 1)(1)
    check(_L().vila_s2_merge_bwd_bf16(dy.contiguous().data_ptr(), dx.data_ptr(), tile_desc.contiguous().data_ptr(), tile_desc.shape[0], g, Cc,
                                      n_scales, sp, _stream()), "s2_merge_bwd")
    return dx
