"""W4A16 weight quantisation for the decode path (BASELINE configs[4]).

The reference's quantised numbers come from TinyChat (external `mit-han-lab/llm-awq`, README.md:87,247-251); nothing of it is
in-tree, so this module defines the converter from bf16 checkpoints to the packed format `vila_amd/csrc/gemv_w4.hip` consumes:
AWQ-style asymmetric uint4, groups of 128 along the input dimension, bf16 scale + integer zero point per group.  (AWQ's
activation-aware per-channel scaling search needs calibration data and changes accuracy, not the kernel format; it is not done.)
The five decoder-layer projections are quantised; embeddings, norms, biases and lm_head stay bf16 (as AWQ does).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib

GROUP = 128


def quantize_w4(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """w [N, K] -> (Wq [N, K/8] int32 packed nibbles, Wsz [N, K/128] int32 {bf16 scale | bf16 (128+zero) << 16})."""
    N, K = w.shape
    assert K % GROUP == 0, f"K={K} must be a multiple of {GROUP}"
    wf = w.float().view(N, K // GROUP, GROUP)
    mn, mx = wf.amin(-1, keepdim=True), wf.amax(-1, keepdim=True)
    scale = ((mx - mn) / 15.0).clamp_min(1e-8).to(torch.bfloat16).float()
    zero = torch.round(-mn / scale).clamp(0, 15)
    q = (torch.round(wf / scale) + zero).clamp(0, 15).to(torch.int32).view(N, K // 8, 8)
    # nibble j (j < 4) = element 2j, nibble j + 4 = element 2j + 1
    word = torch.zeros((N, K // 8), dtype=torch.int32, device=w.device)
    for j in range(4):
        word |= q[..., 2 * j] << (4 * j)
        word |= q[..., 2 * j + 1] << (4 * (j + 4))
    s_bits = scale.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    z_bits = (zero + 128.0).to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    sz = (s_bits | (z_bits << 16)).view(N, K // GROUP)
    return word.contiguous(), sz.contiguous()


def dequantize_w4(wq: torch.Tensor, wsz: torch.Tensor) -> torch.Tensor:
    """Inverse of quantize_w4 in fp32: (q - zero) * scale  — the weights the CPU oracle uses for W4 parity."""
    N, K8 = wq.shape
    q = torch.empty((N, K8, 8), dtype=torch.float32, device=wq.device)
    for j in range(4):
        q[..., 2 * j] = ((wq >> (4 * j)) & 0xF).float()
        q[..., 2 * j + 1] = ((wq >> (4 * (j + 4))) & 0xF).float()
    scale = (wsz & 0xFFFF).to(torch.int16).view(torch.bfloat16).float()
    zero = ((wsz >> 16) & 0xFFFF).to(torch.int16).view(torch.bfloat16).float() - 128.0
    q = q.view(N, K8 * 8 // GROUP, GROUP)
    return ((q - zero[..., None]) * scale[..., None]).view(N, K8 * 8)


class W4Weights:
    """Packed int4 copies of the decoder-layer projections of a HipQwen2ForCausalLM + the ctypes layer table."""

    def __init__(self, llm):
        c = llm.lcfg
        self.tensors = []
        layers = (_lib.VilaLlmLayerW4 * c.num_hidden_layers)()
        for i in range(c.num_hidden_layers):
            l = getattr(llm.model.layers, str(i))
            a = l.self_attn
            wqkv = torch.cat([a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data], 0)
            L = layers[i]
            for name, w in (("qkv", wqkv), ("o", a.o_proj.weight.data), ("gate", l.mlp.gate_proj.weight.data),
                            ("up", l.mlp.up_proj.weight.data), ("down", l.mlp.down_proj.weight.data)):
                q, sz = quantize_w4(w)
                self.tensors += [q, sz]
                setattr(L, name + "_q", q.data_ptr())
                setattr(L, name + "_sz", sz.data_ptr())
        self.layers = layers
        self.ptr = C.cast(layers, C.POINTER(_lib.VilaLlmLayerW4))

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors)

    def dequantized_state(self, llm) -> Dict[str, torch.Tensor]:
        """fp32 weights equal to what the kernels compute with, keyed by the reference's names (for the CPU oracle)."""
        c = llm.lcfg
        out = {}
        it = iter(self.tensors)
        for i in range(c.num_hidden_layers):
            p = f"llm.model.layers.{i}."
            qkv = dequantize_w4(next(it), next(it)).cpu()
            qw, kw, vw = qkv.split([c.q_size, c.kv_size, c.kv_size], 0)
            out[p + "self_attn.q_proj.weight"], out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = qw, kw, vw
            out[p + "self_attn.o_proj.weight"] = dequantize_w4(next(it), next(it)).cpu()
            out[p + "mlp.gate_proj.weight"] = dequantize_w4(next(it), next(it)).cpu()
            out[p + "mlp.up_proj.weight"] = dequantize_w4(next(it), next(it)).cpu()
            out[p + "mlp.down_proj.weight"] = dequantize_w4(next(it), next(it)).cpu()
        return out
