"""CPU oracle for the NVILA hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  Nothing under `vila_amd/` imports it; the product path raises if the HIP library is missing.

What this is: a plain fp32 torch-CPU restatement of the reference's algorithm for SURVEY.md §8 rows
a2-a12 (+ the loss of a11), written from the reference files cited on every function.  It does not
import `/root/reference` or `transformers`, so it travels to the GPU box.

How it is pinned: the reference ships no tests, golden vectors or fixtures for this path,
so `oracle/make_golden.py` EXECUTES the reference's own `modeling_siglip.py` and `base_projector.py`
(loaded by file path) and HF `Qwen2ForCausalLM` (the third-party dependency the reference pins at
`pyproject.toml:17`, transformers==4.46.0; 5.15.0 is what is installed here) on seeded tiny configs and
commits their outputs under `tests/golden/`.  `tests/test_oracle_golden.py` checks this restatement
against those fixtures.  The HF-LLM leg is therefore pinned against transformers 5.15.0, not the 4.46.0
the reference names — same arithmetic (cf. the in-tree mirror
`llava/model/language_model/fp8activationqwen2.py:992-1055`), but say so: the LLM boundary is
"pinned against a newer release of the un-vendored dependency".
Round 3: the same three pieces of reference code were also run ONCE AT FULL SIZE (26 + 28 layers at NVILA-8B widths, S = 769, fp32;
`oracle/make_golden_full_ref.py` -> `tests/golden/nvila8b_full_depth_ref.npz`), and the oracle's own full-depth fixture equals that
run to 2.4e-6 of the largest logit with identical greedy ids (`tests/test_oracle_golden.py`).

Round 4: every remaining restated piece is held to the reference's OWN code, taken from its files with `ast` and executed unchanged
(`llava.model` itself cannot be imported here: deepspeed / hydra):
  * `embed_splice`, `repack`, `get_unpad_data`, `encode_images`, the image / video encoders  <- `LlavaMetaForCausalLM._embed`, `__truncate_sequence`,
    `__batchify_sequence`, `repack_multimodal_data`, `LlavaMetaModel.encode_images`, `VisionTower`, `BaseEncoder` / `BasicImageEncoder` /
    `BasicVideoEncoder` / `TSPVideoEncoder`, packing.py `_get_unpad_data`           (oracle/make_golden_embed.py -> embed_splice_ref.npz)
  * autograd through `vlm_sft_loss`, also through the pooling video encoder  <- reference SigLIP + projector + encoders + HF Qwen2 under autograd
                                                                    (make_golden_grads.py, make_golden_grads_video.py -> tiny_sft_grads*.npz)
  * the dynamic_s2 merge / tiler / image pre-processing, the video encoders' forward, the sampling chain: make_golden_s2*.py, make_golden_video.py,
    make_golden_sampling.py; full depth at NVILA-8B / Lite-3B size: make_golden_full_ref.py, make_golden_lite3b.py.

Round 4, late: the input producers on the data side of the path are NOT restated here (they are product host code, `vila_amd/{serving,host,
conversation,data}.py`) but pinned the same way — `process_image(s)` / `dynamic_preprocess`, `tokenize_conversation` / `preprocess_conversation` /
`infer_stop_tokens`, `DataCollator`: make_golden_dynamic_tiles.py, make_golden_conversation.py, make_golden_collate_cases.py.

All tensors fp32 unless noted.  `w` is a flat dict keyed by the reference's state_dict names
(SURVEY.md Appendix C).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100  # llava/constants.py:26

VT = "vision_tower.vision_tower.vision_model."
PJ = "mm_projector.layers."
LM = "llm."


# ----------------------------------------------------------------------------------------------
# a2: SiglipVisionEmbeddings.forward  (llava/model/multimodal_encoder/siglip/modeling_siglip.py:320-329)
# ----------------------------------------------------------------------------------------------
def siglip_embeddings(pixels: torch.Tensor, w: Dict[str, torch.Tensor], vcfg) -> torch.Tensor:
    """Conv2d(k=s=patch, valid) -> flatten(2).transpose(1,2) -> + position_embedding[0:N]."""
    B, C, H, W = pixels.shape
    P = vcfg.patch_size
    gh, gw = H // P, W // P
    # patchify: token (gy,gx) row-major (:323); K index = c*P*P + ky*P + kx (= weight.view(out, C*P*P))
    x = pixels.reshape(B, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * P * P)
    wt = w[VT + "embeddings.patch_embedding.weight"].reshape(vcfg.hidden_size, -1)
    y = x @ wt.t() + w[VT + "embeddings.patch_embedding.bias"]
    return y + w[VT + "embeddings.position_embedding.weight"][None, : gh * gw]


# ----------------------------------------------------------------------------------------------
# a3: SiglipEncoderLayer.forward (:728-764), SiglipAttention eager (:389-439), SiglipMLP (:711-715)
# ----------------------------------------------------------------------------------------------
def siglip_attention(x: torch.Tensor, w, prefix: str, num_heads: int) -> torch.Tensor:
    B, N, D = x.shape
    hd = D // num_heads
    q = F.linear(x, w[prefix + "q_proj.weight"], w[prefix + "q_proj.bias"])
    k = F.linear(x, w[prefix + "k_proj.weight"], w[prefix + "k_proj.bias"])
    v = F.linear(x, w[prefix + "v_proj.weight"], w[prefix + "v_proj.bias"])
    q = q.view(B, N, num_heads, hd).transpose(1, 2)
    k = k.view(B, N, num_heads, hd).transpose(1, 2)
    v = v.view(B, N, num_heads, hd).transpose(1, 2)
    s = (q @ k.transpose(2, 3)) * (hd ** -0.5)            # :408 scale = head_dim**-0.5, non-causal
    p = torch.softmax(s.float(), dim=-1)                  # :424 softmax in fp32
    o = (p @ v).transpose(1, 2).reshape(B, N, D)
    return F.linear(o, w[prefix + "out_proj.weight"], w[prefix + "out_proj.bias"])


def siglip_mlp(x, w, prefix):
    h = F.linear(x, w[prefix + "fc1.weight"], w[prefix + "fc1.bias"])
    h = F.gelu(h, approximate="tanh")                     # hidden_act = gelu_pytorch_tanh (so400m config)
    return F.linear(h, w[prefix + "fc2.weight"], w[prefix + "fc2.bias"])


def siglip_encoder_layer(x, w, i: int, vcfg) -> torch.Tensor:
    l = f"{VT}encoder.layers.{i}."
    D = x.shape[-1]
    r = x
    h = F.layer_norm(x, (D,), w[l + "layer_norm1.weight"], w[l + "layer_norm1.bias"], vcfg.layer_norm_eps)
    x = r + siglip_attention(h, w, l + "self_attn.", vcfg.num_attention_heads)
    r = x
    h = F.layer_norm(x, (D,), w[l + "layer_norm2.weight"], w[l + "layer_norm2.bias"], vcfg.layer_norm_eps)
    return r + siglip_mlp(h, w, l + "mlp.")


# ----------------------------------------------------------------------------------------------
# a4: VisionTower.forward + feature_select (llava/model/multimodal_encoder/vision_encoder.py:44-52,133-177)
# ----------------------------------------------------------------------------------------------
def vision_tower_forward(pixels, w, vcfg, return_all: bool = False):
    """hidden_states[select_layer] with select_feature="cls_patch" (keeps every token; SigLIP has no CLS).

    The reference runs all L layers (+post_layernorm) and indexes hidden_states; layers past the selected
    index cannot affect it (modeling_siglip.py:994-1011), so the oracle stops there.
    """
    x = siglip_embeddings(pixels, w, vcfg)
    hs = [x]
    n_used = vcfg.select_layer if vcfg.select_layer >= 0 else vcfg.num_hidden_layers + 1 + vcfg.select_layer
    for i in range(n_used):
        x = siglip_encoder_layer(x, w, i, vcfg)
        hs.append(x)
    return hs if return_all else x


# ----------------------------------------------------------------------------------------------
# SURVEY §8f row 3 (BASELINE configs[4]): W8A8 vision tower.  The reference's quantised numbers come from the EXTERNAL TinyChat backend
# (README.md:87; nothing in-tree): parity unpinned against the reference.  This is the dequantise-then-fp32 oracle of the SAME scheme the
# HIP path implements: per-output-channel symmetric int8 weights (given already dequantised in `w`), per-token dynamic symmetric int8
# activations in front of each of the four linears of a layer.
# ----------------------------------------------------------------------------------------------
def fake_quant_rows(x: torch.Tensor) -> torch.Tensor:
    """x -> dequant(quant(x)) with scale = max|row| / 127, round-half-even, clamp to [-127, 127]."""
    s = (x.abs().amax(-1, keepdim=True) / 127.0)
    s = torch.where(s > 0, s, torch.ones_like(s))
    return torch.round(x / s).clamp(-127, 127) * s


def vision_tower_forward_w8a8(pixels, w, vcfg):
    """vision_tower_forward with the inputs of q/k/v, out_proj, fc1 and fc2 fake-quantised per token (weights in `w` are the dequantised
    int8 weights).  Activations between kernels are rounded to bf16 where the HIP path stores bf16, so the integer grids line up."""
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(siglip_embeddings(pixels, w, vcfg))
    H = vcfg.num_attention_heads
    n_used = vcfg.select_layer if vcfg.select_layer >= 0 else vcfg.num_hidden_layers + 1 + vcfg.select_layer
    for i in range(n_used):
        l = f"{VT}encoder.layers.{i}."
        h = bf(F.layer_norm(x, (x.shape[-1],), w[l + "layer_norm1.weight"], w[l + "layer_norm1.bias"], vcfg.layer_norm_eps))
        hq = fake_quant_rows(h)
        B, N, D = hq.shape
        hd = D // H
        q = bf(F.linear(hq, w[l + "self_attn.q_proj.weight"], w[l + "self_attn.q_proj.bias"])).view(B, N, H, hd).transpose(1, 2)
        k = bf(F.linear(hq, w[l + "self_attn.k_proj.weight"], w[l + "self_attn.k_proj.bias"])).view(B, N, H, hd).transpose(1, 2)
        v = bf(F.linear(hq, w[l + "self_attn.v_proj.weight"], w[l + "self_attn.v_proj.bias"])).view(B, N, H, hd).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(2, 3)) * hd ** -0.5, -1) @ v
        a = bf(a.transpose(1, 2).reshape(B, N, D))
        x = bf(x + F.linear(fake_quant_rows(a), w[l + "self_attn.out_proj.weight"], w[l + "self_attn.out_proj.bias"]))
        h = bf(F.layer_norm(x, (D,), w[l + "layer_norm2.weight"], w[l + "layer_norm2.bias"], vcfg.layer_norm_eps))
        f = bf(F.gelu(F.linear(fake_quant_rows(h), w[l + "mlp.fc1.weight"], w[l + "mlp.fc1.bias"]), approximate="tanh"))
        x = bf(x + F.linear(fake_quant_rows(f), w[l + "mlp.fc2.weight"], w[l + "mlp.fc2.bias"]))
    return x


# ----------------------------------------------------------------------------------------------
# a5: DownSampleBlock.flat_square / flat_square_2x2 / flat_square_3x3 + MultimodalProjector.forward
#     (llava/model/multimodal_projector/base_projector.py:58-71, 84-97, 110-123, 145-174, 248-252)
# ----------------------------------------------------------------------------------------------
def flat_square(x: torch.Tensor, k: int) -> torch.Tensor:
    """[n, w, h, c] -> [n, ceil(w/k), ceil(h/k), k*k*c]; zero pad first (2x2: odd only, 3x3: to a multiple)."""
    n, w_, h_, c = x.shape
    if w_ % k != 0:
        x = torch.cat([x, torch.zeros((n, k - w_ % k, h_, c), dtype=x.dtype)], dim=1)
        n, w_, h_, c = x.shape
    if h_ % k != 0:
        x = torch.cat([x, torch.zeros((n, w_, k - h_ % k, c), dtype=x.dtype)], dim=2)
        n, w_, h_, c = x.shape
    x = x.contiguous().view(n, w_, h_ // k, c * k)
    x = x.permute(0, 2, 1, 3).contiguous()
    x = x.view(n, h_ // k, w_ // k, c * k * k)
    x = x.permute(0, 2, 1, 3).contiguous()
    return x


def downsample_block(x: torch.Tensor, k: int) -> torch.Tensor:
    B, N, C = x.shape
    g = int(N ** 0.5)
    y = flat_square(x.reshape(B, g, g, C), k)
    return y.reshape(B, -1, y.shape[-1])


def projector_forward(x: torch.Tensor, w, ptype: str) -> torch.Tensor:
    def lin(t, i):
        return F.linear(t, w[f"{PJ}{i}.weight"], w[f"{PJ}{i}.bias"])

    def ln(t, i):
        return F.layer_norm(t, (t.shape[-1],), w[f"{PJ}{i}.weight"], w[f"{PJ}{i}.bias"], 1e-5)

    if ptype in ("mlp_downsample", "mlp_downsample_2x2_fix"):
        x = downsample_block(x, 2)
        return lin(F.gelu(lin(ln(x, 1), 2)), 4)           # nn.GELU() = erf form (:150)
    if ptype == "mlp_downsample_3x3_fix":
        x = downsample_block(x, 3)
        x = F.gelu(lin(ln(x, 1), 2))
        x = F.gelu(lin(ln(x, 4), 5))
        return lin(x, 7)
    raise ValueError(f"Unknown projector type: {ptype}")


# ----------------------------------------------------------------------------------------------
# a6/a7: encode_images plain branch (llava_arch.py:366-394), BasicImageEncoder.forward (encoders/image/basic.py:41-79)
# ----------------------------------------------------------------------------------------------
def encode_images(pixels, w, cfg, block_sizes=None):
    """LlavaMetaForCausalLM.encode_images (llava_arch.py:366-394).  dynamic_s2: block_sizes None = one None per input (:367-368 — how the
    video encoders reach it: every frame is a one-tile image whose features are repeated over the scales, :309-314); the per-image token
    lists are stacked when they all have the same length (:389-390)."""
    if getattr(cfg, "dynamic_s2", False):
        outs = encode_images_dynamic_s2(pixels, [None] * len(pixels) if block_sizes is None else block_sizes, w, cfg)
        return torch.stack(outs, 0) if all(o.shape[0] == outs[0].shape[0] for o in outs) else outs
    return projector_forward(vision_tower_forward(pixels, w, cfg.vision), w, cfg.mm_projector_type)


def embed_tokens(ids: torch.Tensor, w) -> torch.Tensor:
    return w[LM + "model.embed_tokens.weight"][ids]


def basic_image_encoder(images: Sequence[torch.Tensor], w, cfg) -> List[torch.Tensor]:
    """stack -> encode_images -> append embed(tokenizer("\\n")) to each image's tokens."""
    feats = encode_images(torch.stack(list(images), 0), w, cfg)
    end = embed_tokens(torch.tensor([cfg.newline_token_id]), w)
    return [torch.cat([f, end], 0) for f in feats]


# ----------------------------------------------------------------------------------------------
# a7 (video): BasicVideoEncoder (llava/model/encoders/video/basic.py:13-53), TSPVideoEncoder (video/tsp.py:10-64)
# ----------------------------------------------------------------------------------------------
def pool(x: torch.Tensor, size: int, dim: int) -> torch.Tensor:
    """tsp.py:10-11."""
    return x.view(x.shape[:dim] + (-1, size) + x.shape[dim + 1:]).mean(dim + 1)


def video_process_features(features: torch.Tensor, start: Optional[torch.Tensor], end: Optional[torch.Tensor]) -> torch.Tensor:
    """BasicVideoEncoder._process_features (basic.py:30-41): features [n_frames, n_tok, H]; start / end [k, H] token embeddings
    are put before / after EVERY frame's tokens; frames are then flattened."""
    if start is not None:
        features = torch.cat([torch.stack([start] * features.shape[0], 0), features], 1)
    if end is not None:
        features = torch.cat([features, torch.stack([end] * features.shape[0], 0)], 1)
    return features.flatten(0, 1)


def tsp_process_features(inputs: torch.Tensor, pool_sizes, start, end, sep=None) -> torch.Tensor:
    """TSPVideoEncoder._process_features (tsp.py:28-52): for every pool size, mean-pool the [nt, nl, nl, H] grid over dims 0,1,2 in
    turn, flatten (h, w), add start/end tokens per pooled frame, append the separator; outputs of all pool sizes concatenated."""
    nt, ns = inputs.shape[:2]
    nl = int(ns ** 0.5)
    outputs = []
    for pool_size in pool_sizes:
        f = inputs.view(nt, nl, nl, -1)
        for dim, p in enumerate(pool_size):
            f = pool(f, p, dim=dim)
        f = video_process_features(f.flatten(1, 2), start, end)
        if sep is not None:
            f = torch.cat([f, sep], 0)
        outputs.append(f)
    return torch.cat(outputs, 0)


def basic_video_encoder(videos: Sequence[torch.Tensor], w, cfg) -> List[torch.Tensor]:
    """video/basic.py:43-53: all frames of all videos through encode_images, split per video, tokens assembled per frame."""
    feats = encode_images(torch.cat(list(videos), 0), w, cfg)
    end = embed_tokens(torch.tensor([cfg.newline_token_id]), w)
    return [video_process_features(f, None, end) for f in torch.split(feats, [int(v.shape[0]) for v in videos])]


def tsp_video_encoder(videos: Sequence[torch.Tensor], w, cfg, pool_sizes, sep_ids: Optional[Sequence[int]] = None,
                      start_ids: Optional[Sequence[int]] = None, end_ids: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
    """video/tsp.py:54-64.  end_ids None = the "\n" end token (the encoder's default end_tokens, tsp.py:18-19)."""
    feats = encode_images(torch.cat(list(videos), 0), w, cfg)
    end_ids = [cfg.newline_token_id] if end_ids is None else list(end_ids)
    end = embed_tokens(torch.tensor(end_ids), w) if end_ids else None
    start = embed_tokens(torch.tensor(list(start_ids)), w) if start_ids else None
    sep = embed_tokens(torch.tensor(list(sep_ids)), w) if sep_ids else None
    return [tsp_process_features(f, pool_sizes, start, end, sep) for f in torch.split(feats, [int(v.shape[0]) for v in videos])]


# ----------------------------------------------------------------------------------------------
# a8: LlavaMetaForCausalLM._embed + __batchify_sequence (llava_arch.py:412-490, 528-555)
# ----------------------------------------------------------------------------------------------
def embed_splice(input_ids: torch.Tensor, media_embeds, w, cfg,
                 labels: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 padding_side: str = "right", max_length: Optional[int] = None):
    """Returns (inputs_embeds [B,S,H], labels [B,S] i64, attention_mask [B,S] bool).
    media_embeds: list of image embedding blocks, or {name: blocks} with names "image" / "video" (one deque per name, each popped
    when ITS media token is met: llava_arch.py:454-466).
    max_length = `tokenizer.model_max_length` in training mode: `__truncate_sequence` (llava_arch.py:519-526) cuts every sample
    AFTER media expansion, only if some sample is longer."""
    labels = labels if labels is not None else torch.full_like(input_ids, IGNORE_INDEX)
    attention_mask = attention_mask if attention_mask is not None else torch.ones_like(input_ids, dtype=torch.bool)
    text = embed_tokens(input_ids, w)
    B = input_ids.shape[0]
    queues = {n: list(v) for n, v in media_embeds.items()} if isinstance(media_embeds, dict) else {"image": list(media_embeds)}
    media_tokens = {cfg.image_token_id: "image", getattr(cfg, "video_token_id", -1): "video"}       # :454-457
    ins, labs = [], []
    for k in range(B):
        ids_k = input_ids[k][attention_mask[k]]      # :449-450 remove padding
        te_k = text[k][attention_mask[k]]
        lb_k = labels[k][attention_mask[k]]
        pi, pl = [], []
        for pos in range(len(ids_k)):                # :459-477 (run-length form of the while loop)
            if int(ids_k[pos]) in media_tokens:
                m = queues.get(media_tokens[int(ids_k[pos])], []).pop(0)
                pi.append(m)
                pl.append(torch.full((m.shape[0],), IGNORE_INDEX, dtype=lb_k.dtype))
            else:
                pi.append(te_k[pos:pos + 1])
                pl.append(lb_k[pos:pos + 1])
        ins.append(torch.cat(pi, 0))
        labs.append(torch.cat(pl, 0))
    for name, q in queues.items():
        if q:
            raise ValueError(f"Not all {name} embeddings are consumed!")   # :481-484
    if max_length is not None and any(x.shape[0] > max_length for x in ins):      # :519-526
        ins = [x[:max_length] for x in ins]
        labs = [x[:max_length] for x in labs]
    S = max(x.shape[0] for x in ins)
    H = ins[0].shape[1]
    out_e = torch.zeros((B, S, H), dtype=ins[0].dtype)
    out_l = torch.full((B, S), IGNORE_INDEX, dtype=labels.dtype)
    out_m = torch.zeros((B, S), dtype=torch.bool)
    for k in range(B):
        n = ins[k].shape[0]
        sl = slice(0, n) if padding_side == "right" else slice(S - n, S)
        out_e[k, sl] = ins[k]
        out_l[k, sl] = labs[k]
        out_m[k, sl] = True
    return out_e, out_l, out_m


# ----------------------------------------------------------------------------------------------
# a9: repack_multimodal_data non-SP branch (llava_arch.py:744-800) + packing._get_unpad_data (utils/packing.py:12-21)
# ----------------------------------------------------------------------------------------------
def repack(inputs_embeds, attention_mask, labels):
    """-> (embeds [1,ΣS+1,H], mask [1,ΣS+1] i32, position_ids [1,ΣS+1] i32, labels [1,ΣS+1], seqlens [B])."""
    B = inputs_embeds.shape[0]
    seqlens = [int(attention_mask[k].sum()) for k in range(B)]
    e = [inputs_embeds[k][attention_mask[k]] for k in range(B)]
    m = [torch.ones(n, dtype=torch.int32) for n in seqlens]
    p = [torch.arange(n, dtype=torch.int32) for n in seqlens]
    l = [labels[k][attention_mask[k]].clone() for k in range(B)]
    e.append(torch.zeros(1, inputs_embeds.shape[-1], dtype=inputs_embeds.dtype))   # :754-758 dummy token
    m.append(torch.tensor([0], dtype=torch.int32))
    p.append(torch.tensor([0], dtype=torch.int32))
    l.append(torch.tensor([IGNORE_INDEX], dtype=labels.dtype))
    for x in l:
        x[0] = IGNORE_INDEX                                                          # :760-762
    return (torch.cat(e, 0)[None], torch.cat(m, 0)[None], torch.cat(p, 0)[None], torch.cat(l, 0)[None],
            torch.tensor(seqlens, dtype=torch.int32))


def get_unpad_data(attention_mask: torch.Tensor, seqlens_in_batch: Optional[torch.Tensor] = None):
    if seqlens_in_batch is None:
        seqlens_in_batch = attention_mask.sum(dim=1)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    cu = F.pad(torch.cumsum(seqlens_in_batch, 0, dtype=torch.int32), (1, 0))
    return indices, cu, int(seqlens_in_batch.max())


# ----------------------------------------------------------------------------------------------
# a10/a11: HF Qwen2 (transformers/models/qwen2/modeling_qwen2.py — third-party, pinned ==4.46.0 at
# pyproject.toml:17, NOT under /root/reference; call sites language_model/builder.py:178-180,
# llava_llama.py:134-141, llava_arch.py:833; in-tree mirror of the attention math
# llava/model/language_model/fp8activationqwen2.py:992-1055).  Arithmetic per SURVEY.md Appendix B.
# ----------------------------------------------------------------------------------------------
def rms_norm(x, weight, eps):
    x32 = x.float()
    y = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)
    return weight * y.to(x.dtype)


def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = position_ids.float()[..., None] * inv          # [..., hd/2]
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def qwen2_attention(x, w, l: str, lcfg, cos, sin, attn_bias, past: Optional[Tuple[torch.Tensor, torch.Tensor]]):
    B, S, _ = x.shape
    nq, nk, hd = lcfg.num_attention_heads, lcfg.num_key_value_heads, lcfg.head_dim
    q = F.linear(x, w[l + "q_proj.weight"], w[l + "q_proj.bias"]).view(B, S, nq, hd).transpose(1, 2)
    k = F.linear(x, w[l + "k_proj.weight"], w[l + "k_proj.bias"]).view(B, S, nk, hd).transpose(1, 2)
    v = F.linear(x, w[l + "v_proj.weight"], w[l + "v_proj.bias"]).view(B, S, nk, hd).transpose(1, 2)
    c, s_ = cos[:, None], sin[:, None]
    q = q * c + rotate_half(q) * s_
    k = k * c + rotate_half(k) * s_
    if past is not None:
        k = torch.cat([past[0], k], 2)
        v = torch.cat([past[1], v], 2)
    new_past = (k, v)
    rep = nq // nk
    kk = k.repeat_interleave(rep, 1)                    # repeat_kv
    vv = v.repeat_interleave(rep, 1)
    s = (q @ kk.transpose(2, 3)) * (hd ** -0.5) + attn_bias
    p = torch.softmax(s.float(), -1)
    o = (p @ vv).transpose(1, 2).reshape(B, S, nq * hd)
    return F.linear(o, w[l + "o_proj.weight"]), new_past


def qwen2_mlp(x, w, l: str):
    return F.linear(F.silu(F.linear(x, w[l + "gate_proj.weight"])) * F.linear(x, w[l + "up_proj.weight"]),
                    w[l + "down_proj.weight"])


def causal_bias(S_q: int, S_k: int, key_mask: Optional[torch.Tensor] = None,
                segment_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B or 1, 1, S_q, S_k] additive mask: causal (queries are the last S_q keys), optional key padding,
    optional block-diagonal segments (what flash-attn varlen does for the packed row, packing.py:12-21)."""
    i = torch.arange(S_q)[:, None] + (S_k - S_q)
    j = torch.arange(S_k)[None, :]
    ok = (j <= i)[None, None]
    if key_mask is not None:
        ok = ok & key_mask[:, None, None, :].bool()
    if segment_ids is not None:
        ok = ok & (segment_ids[:, None, :, None] == segment_ids[:, None, None, :])
    bias = torch.zeros(ok.shape, dtype=torch.float32)
    return bias.masked_fill(~ok, float("-inf"))


def qwen2_forward(inputs_embeds, w, lcfg, position_ids: Optional[torch.Tensor] = None,
                  attention_mask: Optional[torch.Tensor] = None, past=None,
                  segment_ids: Optional[torch.Tensor] = None, return_hidden: bool = False):
    """-> (logits [B,S,V] fp32, new_past, hidden_states list if asked)."""
    B, S, _ = inputs_embeds.shape
    n_past = 0 if past is None else past[0][0].shape[2]
    if position_ids is None:
        position_ids = torch.arange(n_past, n_past + S)[None].expand(B, S)
    cos, sin = rope_cos_sin(position_ids, lcfg.head_dim, lcfg.rope_theta)
    bias = causal_bias(S, n_past + S, attention_mask, segment_ids)
    if attention_mask is not None:
        # fully-masked (padding) query rows: keep softmax finite, output is ignored downstream
        dead = torch.isinf(bias).all(-1, keepdim=True)
        bias = torch.where(dead, torch.zeros_like(bias), bias)
    x = inputs_embeds
    hs = [x]
    new_past = []
    for i in range(lcfg.num_hidden_layers):
        l = f"{LM}model.layers.{i}."
        h = rms_norm(x, w[l + "input_layernorm.weight"], lcfg.rms_norm_eps)
        a, kv = qwen2_attention(h, w, l + "self_attn.", lcfg, cos, sin, bias, None if past is None else past[i])
        new_past.append(kv)
        x = x + a
        h = rms_norm(x, w[l + "post_attention_layernorm.weight"], lcfg.rms_norm_eps)
        x = x + qwen2_mlp(h, w, l + "mlp.")
        hs.append(x)
    x = rms_norm(x, w[LM + "model.norm.weight"], lcfg.rms_norm_eps)
    head = w[LM + "lm_head.weight"] if (LM + "lm_head.weight") in w else w[LM + "model.embed_tokens.weight"]
    logits = F.linear(x, head).float()
    return (logits, new_past, hs) if return_hidden else (logits, new_past)


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor, num_items_in_batch: Optional[int] = None):
    """HF ForCausalLMLoss: shift by one, fp32 CE, sum / num_items_in_batch when supplied else mean
    (patched compute_loss supplies the GLOBAL count: llava/train/transformer_normalize_monkey_patch.py:261-268)."""
    lg = logits[..., :-1, :].float().reshape(-1, logits.shape[-1])
    lb = labels[..., 1:].reshape(-1)
    if num_items_in_batch is None:
        return F.cross_entropy(lg, lb, ignore_index=IGNORE_INDEX, reduction="mean")
    return F.cross_entropy(lg, lb, ignore_index=IGNORE_INDEX, reduction="sum") / num_items_in_batch


# ----------------------------------------------------------------------------------------------
# a12: LlavaMetaForCausalLM.generate (llava_arch.py:823-833) -> HF GenerationMixin greedy (do_sample=False)
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def greedy_generate(inputs_embeds, w, cfg, max_new_tokens: int, stop_at_eos: bool = True,
                    forced_ids: Optional[Sequence[int]] = None):
    """Batch-1 greedy decode with KV cache.  Returns (ids [n], per-step fp32 logits [n,V]).
    forced_ids = teacher forcing (feed these ids, still record argmax-able logits) for margin-aware parity."""
    lcfg = cfg.llm
    logits, past = qwen2_forward(inputs_embeds, w, lcfg)
    ids, step_logits = [], []
    last = logits[:, -1]
    for t in range(max_new_tokens):
        step_logits.append(last[0].clone())
        nxt = int(last[0].argmax())
        ids.append(nxt)
        if stop_at_eos and forced_ids is None and nxt == lcfg.eos_token_id:
            break
        feed = nxt if forced_ids is None else int(forced_ids[t])
        e = embed_tokens(torch.tensor([[feed]]), w)
        logits, past = qwen2_forward(e, w, lcfg, past=past)
        last = logits[:, -1]
    return torch.tensor(ids, dtype=torch.int64), torch.stack(step_logits)


@torch.no_grad()
def vlm_prefill_embeds(pixels_list: Sequence[torch.Tensor], input_ids: torch.Tensor, w, cfg):
    """generate()'s `_embed` leg for a batch-1 prompt: images -> tokens(+\\n) -> spliced embeds [1,S,H]."""
    media = basic_image_encoder(pixels_list, w, cfg) if len(pixels_list) else []
    e, _, m = embed_splice(input_ids[None] if input_ids.dim() == 1 else input_ids, media, w, cfg)
    return e, m


@torch.no_grad()
def vlm_generate(pixels_list, input_ids, w, cfg, max_new_tokens: int, **kw):
    e, _ = vlm_prefill_embeds(pixels_list, input_ids, w, cfg)
    return greedy_generate(e, w, cfg, max_new_tokens, **kw)


def vlm_sft_loss(pixels_list, input_ids, labels, attention_mask, w, cfg, num_items_in_batch=None, packed=True, block_sizes=None, videos=None,
                 video_encoder=None):
    """Training forward of llava_llama.py:94-159 (packing branch): _embed -> repack -> llm(..., labels).
    dynamic_s2 (cfg.dynamic_s2, llava_arch.py:369-390): pixels_list = the tiles of every scale of every image, block_sizes = one entry per
    image (media_config["image"]["block_sizes"]); BasicImageEncoder appends the "\n" embedding to each IMAGE's merged tokens."""
    if len(pixels_list) and getattr(cfg, "dynamic_s2", False):
        feats = encode_images_dynamic_s2(torch.stack(list(pixels_list), 0), block_sizes, w, cfg)
        end = embed_tokens(torch.tensor([cfg.newline_token_id]), w)
        media = [torch.cat([f, end], 0) for f in feats]
    else:
        media = basic_image_encoder(pixels_list, w, cfg) if len(pixels_list) else []
    if videos:                                                # one block per <vila/video> token
        if video_encoder is None:                             # BasicVideoEncoder (encoders/video/basic.py:43-53)
            vid = basic_video_encoder(videos, w, cfg)
        else:                                                 # TSPVideoEncoder (encoders/video/tsp.py:14-64): dict(pool_sizes, start_ids, end_ids, sep_ids)
            vid = tsp_video_encoder(videos, w, cfg, video_encoder["pool_sizes"], video_encoder.get("sep_ids"), video_encoder.get("start_ids"),
                                    video_encoder.get("end_ids"))
        media = {"image": media, "video": vid}
    e, l, m = embed_splice(input_ids, media, w, cfg, labels=labels, attention_mask=attention_mask)
    if packed:
        pe, pm, pp, pl, seqlens = repack(e, m, l)
        seg = torch.repeat_interleave(torch.arange(len(seqlens) + 1),
                                      torch.cat([seqlens.long(), torch.tensor([1])]))[None]
        logits, _ = qwen2_forward(pe, w, cfg.llm, position_ids=pp.long(), segment_ids=seg)
        return causal_lm_loss(logits, pl, num_items_in_batch)
    logits, _ = qwen2_forward(e, w, cfg.llm, attention_mask=m)
    return causal_lm_loss(logits, l, num_items_in_batch)


# ----------------------------------------------------------------------------------------------
# SURVEY §8(f) row 1 — dynamic_s2 multi-scale path (the full NVILA-8B recipe, scripts/NVILA/stage*_9tile.sh:19-22:
# --dynamic_s2 True --s2_scales 448,896,1344 --s2_resize_output_to_scale_idx -1)
#   merge_chessboard / split_chessboard            llava/model/llava_arch.py:255-296
#   merge_features_for_dynamic_s2                  llava/model/llava_arch.py:298-364
#   encode_images, dynamic_s2 branch               llava/model/llava_arch.py:369-390
#   VisionTowerDynamicS2 (hidden = C * n_scales)   llava/model/multimodal_encoder/vision_encoder.py:251-276
# ----------------------------------------------------------------------------------------------
def merge_chessboard(x: torch.Tensor, num_split_h: int, num_split_w: int) -> torch.Tensor:
    """x: [B, N, C] (or [B, C, h, w]) holding num_split_h*num_split_w sub-squares along the batch dim -> [b, C, H, W]."""
    B = x.shape[0]
    if x.dim() == 3:
        N = x.shape[1]
        g = int(N ** 0.5)
        x = x.reshape(B, g, g, x.shape[2]).permute(0, 3, 1, 2)
    assert B % (num_split_h * num_split_w) == 0
    b = B // (num_split_h * num_split_w)
    rows = [torch.cat([x[(i * num_split_w + j) * b:(i * num_split_w + j + 1) * b] for j in range(num_split_w)], dim=-1)
            for i in range(num_split_h)]
    return torch.cat(rows, dim=-2)


def split_chessboard(x: torch.Tensor, num_split_h: int, num_split_w: int) -> torch.Tensor:
    B, C, H, W = x.shape
    assert H % num_split_h == 0 and W % num_split_w == 0
    h, w = H // num_split_h, W // num_split_w
    return torch.cat([x[:, :, i * h:(i + 1) * h, j * w:(j + 1) * w] for i in range(num_split_h) for j in range(num_split_w)], dim=0)


def merge_features_for_dynamic_s2(image_features: torch.Tensor, block_sizes, scales, resize_output_to_scale_idx: int = -1):
    feats_each, new_block_sizes = [], []
    cnt = 0
    for bs in block_sizes:
        if bs is None:
            cur = image_features[cnt:cnt + 1]
            g = int(cur.shape[1] ** 0.5)
            cur = cur.reshape(1, g, g, -1).permute(0, 3, 1, 2).repeat(1, len(scales), 1, 1)
            feats_each.append(cur)
            new_block_sizes.append((1, 1))
            cnt += 1
            continue
        per_scale = []
        for scale in scales[:-1]:
            n = (scale // scales[0]) ** 2
            per_scale.append(merge_chessboard(image_features[cnt:cnt + n], scale // scales[0], scale // scales[0]))
            cnt += n
        n_last = bs[0] * bs[1]
        per_scale.append(merge_chessboard(image_features[cnt:cnt + n_last], bs[0], bs[1]))
        cnt += n_last
        out_size = per_scale[resize_output_to_scale_idx].shape[-2:]
        cur = torch.cat([F.interpolate(f.to(torch.float32), size=out_size, mode="area").to(f.dtype) for f in per_scale], dim=1)
        feats_each.append(cur)
        if resize_output_to_scale_idx in (len(scales) - 1, -1):
            new_block_sizes.append(tuple(bs))
        else:
            s = scales[resize_output_to_scale_idx] // scales[0]
            new_block_sizes.append((s, s))
    assert cnt == len(image_features), f"The number of blocks ({cnt}) does not match length of image_features ({len(image_features)})!"
    return feats_each, new_block_sizes


def s2_merge_to_projector_input(image_features, block_sizes, scales, resize_idx: int = -1):
    """-> (proj_in [sum blocks, N, C*n_scales], new_block_sizes): everything between the tower and the projector."""
    feats_each, nbs = merge_features_for_dynamic_s2(image_features, block_sizes, scales, resize_idx)
    blocks = [split_chessboard(x, b[0], b[1]) for x, b in zip(feats_each, nbs)]
    x = torch.cat([t.flatten(2).transpose(1, 2) for t in blocks], dim=0)        # "b c h w -> b (h w) c"
    return x, nbs


def encode_images_dynamic_s2(pixels_tiles, block_sizes, w, cfg) -> List[torch.Tensor]:
    """dynamic_s2 branch of encode_images: list of per-image [N_tokens, hidden] tensors."""
    feats = vision_tower_forward(pixels_tiles, w, cfg.vision)
    x, nbs = s2_merge_to_projector_input(feats, block_sizes, cfg.s2_scales, cfg.s2_resize_output_to_scale_idx)
    y = projector_forward(x, w, cfg.mm_projector_type)
    outs = []
    for part, b in zip(y.split([b[0] * b[1] for b in nbs], dim=0), nbs):
        m = merge_chessboard(part, b[0], b[1])                                   # [1, C, H, W]
        outs.append(m[0].flatten(1).transpose(0, 1))                             # "1 c h w -> (h w) c"
    return outs


# ----------------------------------------------------------------------------------------------------------------------
# sampling: the distribution generate(do_sample=True) draws from
# ----------------------------------------------------------------------------------------------------------------------
def sample_distribution(logits: torch.Tensor, temperature: float, top_k: int, top_p: float) -> torch.Tensor:
    """HF's processor chain behind `llm.generate(..., do_sample=True)` (llava/model/llava_arch.py:833 -> GenerationMixin.sample; order of
    `_get_logits_processor`): TemperatureLogitsWarper (logits / T) -> TopKLogitsWarper (keep the k largest; ties with the k-th stay) ->
    TopPLogitsWarper (ascending cumulative softmax, drop while cum <= 1 - top_p, keep at least one) -> softmax.  fp64, one row.
    Pinned by tests/golden/sampling_hf.npz, which oracle/make_golden_sampling.py produced by executing transformers' own classes."""
    z = logits.double() / temperature
    if top_k:                                                 # top_k = 0: HF adds no TopKLogitsWarper (_get_logits_processor)
        kth = torch.topk(z, min(top_k, z.numel())).values[-1]
        z = z.masked_fill(z < kth, float("-inf"))
    if top_p < 1.0:
        srt, idx = torch.sort(z, descending=False)
        cum = srt.softmax(-1).cumsum(-1)
        remove = cum <= (1 - top_p)
        remove[-1] = False                                   # min_tokens_to_keep = 1
        z = z.masked_fill(torch.zeros_like(remove).scatter(0, idx, remove), float("-inf"))
    return z.softmax(-1)

