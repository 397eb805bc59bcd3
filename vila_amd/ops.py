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
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias) + residual   (nn.Linear semantics; EPI_GATEUP: silu(a w^T) * (a w2^T))."""
    _need(a, name="a"); _need(w, name="w")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    if residual is not None:
        _need(residual, name="residual"); assert residual.shape == (M, N) and residual.stride(1) == 1
    check(_lib.load().vila_gemm_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(w2), _p(bias), _p(residual),
                                     residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0),
                                     1 if out_f32 else 0, M, N, K, epi, _stream()), "vila_gemm_bf16")
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


def argmax(logits: torch.Tensor) -> torch.Tensor:
    _need(logits, dtype=torch.float32, name="logits")
    out = torch.empty((1,), device=logits.device, dtype=torch.int64)
    ws = torch.empty((4096,), device=logits.device, dtype=torch.uint8)
    check(_lib.load().vila_argmax_f32(logits.data_ptr(), logits.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "argmax")
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
