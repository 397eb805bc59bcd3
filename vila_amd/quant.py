"""W4A16 weight quantisation for the decode path (BASELINE configs[4]).

The reference's quantised numbers come from TinyChat (external `mit-han-lab/llm-awq`, README.md:87,247-251); nothing of it is
in-tree, so this module defines the converter from bf16 checkpoints to the packed format `vila_amd/csrc/gemv_w4.hip` consumes:
AWQ-style asymmetric uint4, groups of 128 along the input dimension, bf16 scale + integer zero point per group.  (AWQ's
activation-aware per-channel scaling search needs calibration data and changes accuracy, not the kernel format; it is not done.)
The five decoder-layer projections are quantised; embeddings, norms, biases and lm_head stay bf16 (as AWQ does).

Two representations:
  logical   Wq [N, K/8] int32 (nibble p<4 = element 2p, nibble p+4 = element 2p+1 of each 8-run), Wsz [N, K/128] int32
            {bf16 scale | bf16 (128 + zero) << 16}: what `dequantize_w4` (the parity reference) reads
  tiled     the HBM layout of the kernels: 16-row tiles, [N/16][K/128][64 lanes][4] words + [N/16][K/128][16] scale/zero
            (`tile_w4`), rows permuted per use (`W4Matrix.pack`): gate/up interleaved, q/k heads with RoPE partners adjacent
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib

GROUP = 128
TILE = 16


def quantize_w4(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """w [N, K] -> logical (Wq [N, K/8] int32, Wsz [N, K/128] int32)."""
    N, K = w.shape
    assert K % GROUP == 0, f"K={K} must be a multiple of {GROUP}"
    wf = w.float().view(N, K // GROUP, GROUP)
    mn, mx = wf.amin(-1, keepdim=True).clamp_max(0.0), wf.amax(-1, keepdim=True).clamp_min(0.0)
    scale = ((mx - mn) / 15.0).clamp_min(1e-8).to(torch.bfloat16).float()
    zero = torch.round(-mn / scale).clamp(0, 15)
    q = (torch.round(wf / scale) + zero).clamp(0, 15).to(torch.int32).view(N, K // 8, 8)
    word = torch.zeros((N, K // 8), dtype=torch.int32, device=w.device)
    for p in range(4):
        word |= q[..., 2 * p] << (4 * p)
        word |= q[..., 2 * p + 1] << (4 * (p + 4))
    s_bits = scale.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    z_bits = (zero + 128.0).to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    sz = (s_bits | (z_bits << 16)).view(N, K // GROUP)
    return word.contiguous(), sz.contiguous()


def dequantize_w4(wq: torch.Tensor, wsz: torch.Tensor) -> torch.Tensor:
    """Inverse of quantize_w4 in fp32: (q - zero) * scale  — the weights the CPU oracle uses for W4 parity."""
    N, K8 = wq.shape
    q = torch.empty((N, K8, 8), dtype=torch.float32, device=wq.device)
    for p in range(4):
        q[..., 2 * p] = ((wq >> (4 * p)) & 0xF).float()
        q[..., 2 * p + 1] = ((wq >> (4 * (p + 4))) & 0xF).float()
    scale = (wsz & 0xFFFF).to(torch.int16).view(torch.bfloat16).float()
    zero = ((wsz >> 16) & 0xFFFF).to(torch.int16).view(torch.bfloat16).float() - 128.0
    q = q.view(N, K8 * 8 // GROUP, GROUP)
    return ((q - zero[..., None]) * scale[..., None]).view(N, K8 * 8)


def tile_w4(wq: torch.Tensor, wsz: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """logical -> tile-major HBM layout; rows are zero-padded to a multiple of 16 (padded outputs are never written)."""
    N, K8 = wq.shape
    G = K8 * 8 // GROUP
    Np = (N + TILE - 1) // TILE * TILE
    if Np != N:
        wq = torch.cat([wq, wq.new_zeros((Np - N, K8))], 0)
        wsz = torch.cat([wsz, wsz.new_zeros((Np - N, G))], 0)
    # [tile, n, group, g, w] -> [tile, group, g, n, w]      (lane = 16 * g + n)
    q_t = wq.view(Np // TILE, TILE, G, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
    sz_t = wsz.view(Np // TILE, TILE, G).permute(0, 2, 1).contiguous()
    return q_t.view(-1), sz_t.view(-1)


def rope_interleave_rows(n_rope_heads: int, n_heads: int, head_dim: int, device=None) -> torch.Tensor:
    """Row permutation of a fused qkv matrix: packed row h*hd + 2i + b  <-  row h*hd + i + b*hd/2 for the first n_rope_heads heads."""
    half = head_dim // 2
    i = torch.arange(half, device=device)
    one = torch.stack([i, i + half], 1).reshape(-1)
    idx = [one + h * head_dim if h < n_rope_heads else torch.arange(head_dim, device=device) + h * head_dim for h in range(n_heads)]
    return torch.cat(idx)


@dataclass
class W4Matrix:
    """One quantised projection: tiled device buffers + (optionally) the logical arrays they came from."""
    q: torch.Tensor
    sz: torch.Tensor
    N: int                       # outputs (for gate/up: rows of ONE of the two matrices)
    K: int
    mode: int                    # 0 plain, 1 gate/up interleaved
    logical: Optional[Tuple[torch.Tensor, torch.Tensor]] = None

    @staticmethod
    def pack(w: torch.Tensor, w_up: Optional[torch.Tensor] = None, row_perm: Optional[torch.Tensor] = None, keep_logical: bool = True):
        N, K = w.shape
        rows = w
        if w_up is not None:
            rows = torch.stack([w, w_up], 1).reshape(2 * N, K)
        wq, wsz = quantize_w4(rows)
        tq, tsz = wq, wsz
        if row_perm is not None:
            tq, tsz = wq[row_perm], wsz[row_perm]
        q_t, sz_t = tile_w4(tq, tsz)
        return W4Matrix(q_t, sz_t, N, K, 1 if w_up is not None else 0, (wq, wsz) if keep_logical else None)

    def dequantized(self):
        """fp32 weights as the kernel reconstructs them; gate/up mode returns (gate, up)."""
        d = dequantize_w4(*self.logical)
        if self.mode == 1:
            d = d.view(self.N, 2, self.K)
            return d[:, 0].contiguous(), d[:, 1].contiguous()
        return d

    def nbytes(self) -> int:
        return self.q.numel() * 4 + self.sz.numel() * 4


class W4Weights:
    """Packed int4 copies of the decoder-layer projections of a HipQwen2ForCausalLM + the ctypes layer table."""

    def __init__(self, llm, keep_logical: bool = True):
        c = llm.lcfg
        assert c.head_dim % TILE == 0
        self.mats = []
        layers = (_lib.VilaLlmLayerW4 * c.num_hidden_layers)()
        perm = rope_interleave_rows(c.num_attention_heads + c.num_key_value_heads, c.num_attention_heads + 2 * c.num_key_value_heads,
                                    c.head_dim, device=llm.device)
        for i in range(c.num_hidden_layers):
            l = getattr(llm.model.layers, str(i))
            a = l.self_attn
            wqkv = torch.cat([a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data], 0)
            m = {"qkv": W4Matrix.pack(wqkv, row_perm=perm, keep_logical=keep_logical),
                 "o": W4Matrix.pack(a.o_proj.weight.data, keep_logical=keep_logical),
                 "gateup": W4Matrix.pack(l.mlp.gate_proj.weight.data, l.mlp.up_proj.weight.data, keep_logical=keep_logical),
                 "down": W4Matrix.pack(l.mlp.down_proj.weight.data, keep_logical=keep_logical)}
            for name, mat in m.items():
                setattr(layers[i], name + "_q", mat.q.data_ptr())
                setattr(layers[i], name + "_sz", mat.sz.data_ptr())
            self.mats.append(m)
        self.layers = layers
        self.ptr = C.cast(layers, C.POINTER(_lib.VilaLlmLayerW4))

    def nbytes(self) -> int:
        return sum(mat.nbytes() for m in self.mats for mat in m.values())

    def dequantized_state(self, llm) -> Dict[str, torch.Tensor]:
        """fp32 weights equal to what the kernels compute with, keyed by the reference's names (for the CPU oracle)."""
        c = llm.lcfg
        out = {}
        for i, m in enumerate(self.mats):
            p = f"llm.model.layers.{i}."
            qw, kw, vw = m["qkv"].dequantized().cpu().split([c.q_size, c.kv_size, c.kv_size], 0)
            out[p + "self_attn.q_proj.weight"], out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = qw, kw, vw
            out[p + "self_attn.o_proj.weight"] = m["o"].dequantized().cpu()
            g, u = m["gateup"].dequantized()
            out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"] = g.cpu(), u.cpu()
            out[p + "mlp.down_proj.weight"] = m["down"].dequantized().cpu()
        return out


# ----------------------------------------------------------------------------------------------------------------------
# W8A8 vision tower (BASELINE configs[4]: "W8A8 vision tower"; TinyChat is external, README.md:87)
# ----------------------------------------------------------------------------------------------------------------------
def quantize_w8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """w [N, K] -> (int8 [N, K], fp32 scale [N]): symmetric per-OUTPUT-CHANNEL, scale = max|w_row| / 127, q = round(w / scale)."""
    wf = w.float()
    scale = (wf.abs().amax(-1) / 127.0).clamp_min(1e-12)
    q = torch.round(wf / scale[:, None]).clamp(-127, 127).to(torch.int8)
    return q.contiguous(), scale.contiguous()


def dequantize_w8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.float() * scale[:, None].float()


class W8VitWeights:
    """int8 copies (+ per-row scales) of the fused q/k/v, out_proj, fc1, fc2 weights of every encoder layer the tower runs, and the
    host array of VilaVitLayerW8 the C-ABI takes.  Biases, LayerNorms, patch / position embeddings stay bf16 in the tower."""

    def __init__(self, tower):
        v = tower.vcfg
        vm = tower.vision_tower.vision_model
        n_run = v.num_used_layers
        self.tensors = []
        self.layers_c = (_lib.VilaVitLayerW8 * max(n_run, 1))()
        for i in range(n_run):
            l = getattr(vm.encoder.layers, str(i))
            a = l.self_attn
            wqkv = torch.cat([a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data], 0)
            t = {}
            for name, w in (("wqkv", wqkv), ("wo", a.out_proj.weight.data), ("fc1", l.mlp.fc1.weight.data), ("fc2", l.mlp.fc2.weight.data)):
                t[name + "_q"], t[name + "_s"] = quantize_w8(w)
            self.tensors.append(t)
            L = self.layers_c[i]
            for k, val in t.items():
                setattr(L, k, val.data_ptr())
        self.ptr = C.cast(self.layers_c, C.POINTER(_lib.VilaVitLayerW8))

    def dequantized_state(self, prefix: str = "vision_tower.vision_tower.vision_model.") -> Dict[str, torch.Tensor]:
        """fp32 weights equal to what the int8 GEMMs multiply by (for the CPU oracle): keys as in the reference's state_dict."""
        out = {}
        for i, t in enumerate(self.tensors):
            l = f"{prefix}encoder.layers.{i}."
            wqkv = dequantize_w8(t["wqkv_q"], t["wqkv_s"]).cpu()
            d = wqkv.shape[0] // 3
            out[l + "self_attn.q_proj.weight"], out[l + "self_attn.k_proj.weight"], out[l + "self_attn.v_proj.weight"] = wqkv[:d], wqkv[d:2 * d], wqkv[2 * d:]
            out[l + "self_attn.out_proj.weight"] = dequantize_w8(t["wo_q"], t["wo_s"]).cpu()
            out[l + "mlp.fc1.weight"] = dequantize_w8(t["fc1_q"], t["fc1_s"]).cpu()
            out[l + "mlp.fc2.weight"] = dequantize_w8(t["fc2_q"], t["fc2_s"]).cpu()
        return out
