"""Drop-in nn.Modules for the three builder seams of the reference:

    build_vision_tower  (llava/model/multimodal_encoder/builder.py:30)      -> HipSiglipVisionTower
    build_mm_projector  (llava/model/multimodal_projector/builder.py:27)    -> HipMultimodalProjector
    build_llm_and_tokenizer (llava/model/language_model/builder.py:64)      -> HipQwen2ForCausalLM

Each keeps the reference's parameter names/shapes so the three-folder checkpoints of
`llava_arch.py:158-204` load with `load_state_dict`, and keeps the reference's call contract
(`vision_tower(images) -> [B,N,C]`, `mm_projector(x) -> [B,N',H]`, `llm(inputs_embeds=..., labels=...)`,
`llm.generate(inputs_embeds=..., attention_mask=...)`).  All math runs in libvila_hip.so.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops, synthetic
from ._lib import check
from .configs import IGNORE_INDEX, LlmConfig, VilaConfig, VisionConfig


# --------------------------------------------------------------------------------------------------------------
# parameter tree with the reference's names; q/k/v projections are views of one fused buffer
# --------------------------------------------------------------------------------------------------------------
def _register(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, nn.Module())
        m = getattr(m, p)
    m.register_parameter(parts[-1], param)


def _get(root: nn.Module, dotted: str) -> torch.Tensor:
    m = root
    for p in dotted.split("."):
        m = getattr(m, p)
    return m


def _fuse_groups(names: List[str]) -> Dict[str, Tuple[str, ...]]:
    """q_proj -> (q_proj, k_proj, v_proj) groups (weights and biases) that must be contiguous for the fused kernels."""
    groups = {}
    for n in names:
        if ".self_attn.q_proj." in n:
            groups[n] = (n, n.replace("q_proj", "k_proj"), n.replace("q_proj", "v_proj"))
    return groups


class _HipModule(nn.Module):
    """Builds parameters from (name, shape, kind) specs, caches the ctypes weight structs, owns a grow-only workspace."""

    def _build_params(self, specs, prefix: str, device, dtype, requires_grad=False):
        names = [n for n, _, _ in specs]
        shapes = {n: s for n, s, _ in specs}
        groups = _fuse_groups(names)
        grouped = {m for g in groups.values() for m in g}
        self._fused_groups = [tuple(m[len(prefix):] for m in g) for g in groups.values()]
        for n, shape, _ in specs:
            if n in grouped:
                continue
            _register(self, n[len(prefix):], nn.Parameter(torch.empty(shape, device=device, dtype=dtype), requires_grad=requires_grad))
        for g in groups.values():
            rows = [shapes[m][0] for m in g]
            tail = shapes[g[0]][1:]
            buf = torch.empty((sum(rows), *tail), device=device, dtype=dtype)
            o = 0
            for m, r in zip(g, rows):
                _register(self, m[len(prefix):], nn.Parameter(buf[o:o + r], requires_grad=requires_grad))
                o += r
        self._cstruct = None
        self._ws = None

    def _drop_decode_session(self) -> None:
        st = getattr(self, "_decode", None)
        if st is not None:
            if getattr(st, "graph", None) is not None:
                _lib.load().vila_graph_destroy(st.graph)
                st.graph = None
            self._decode = None

    def _drop_batch_session(self) -> None:
        bst = getattr(self, "_bdecode", None)
        if bst is not None:
            if getattr(bst, "graph", None) is not None:
                _lib.load().vila_graph_destroy(bst.graph)
                bst.graph = None
            self._bdecode = None

    def _invalidate(self) -> None:
        """Parameter storage moved (.to(), refuse(), FlatParams re-pointing, W4 quantisation): every cached object that baked in
        a weight pointer is dropped — the ctypes weight struct and, for the LLM, both decode sessions with their captured hipGraphs.
        (A batch-1 session whose cache / length / sampling key changed drops only ITSELF, `_decode_session`: the n-slot batch session
        with its KV cache and graph survives a server that alternates batched and solo requests — ADVICE round 3.)"""
        self._cstruct = None
        self._drop_decode_session()
        self._drop_batch_session()

    def refuse(self) -> None:
        """Re-establish the fused q/k/v storage after an op that re-allocated parameters (.to(), .half(), ...)."""
        for g in self._fused_groups:
            ts = [_get(self, m) for m in g]
            buf = torch.cat([t.data for t in ts], 0)
            o = 0
            for t in ts:
                t.data = buf[o:o + t.shape[0]]
                o += t.shape[0]
        self._invalidate()

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if getattr(self, "_fused_groups", None):
            self.refuse()
        self._invalidate()
        self._ws = None
        return r

    def load_weights(self, w: Dict[str, torch.Tensor], prefix: str) -> None:
        with torch.no_grad():
            for n, p in self.named_parameters():
                p.copy_(w[prefix + n])

    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty((int(nbytes),), device=device, dtype=torch.uint8)
        return self._ws

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype


# --------------------------------------------------------------------------------------------------------------
# Vision tower
# --------------------------------------------------------------------------------------------------------------
class HipSiglipVisionTower(_HipModule):
    """`VisionTower.forward` + `feature_select` of llava/model/multimodal_encoder/vision_encoder.py:44-52,133-177 for
    SiglipVisionTower (siglip_encoder.py:25-63): images [B,3,H,W] -> hidden_states[select_layer] [B,N,C], "cls_patch"."""

    def __init__(self, cfg: VilaConfig, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        self.cfg = cfg
        self.vcfg: VisionConfig = cfg.vision
        self.select_layer = cfg.vision.select_layer
        self.select_feature = "cls_patch"
        self.is_loaded = True
        self._build_params(synthetic.vision_specs(cfg), "vision_tower.", device, dtype)
        self.config = SimpleNamespace(hidden_size=cfg.vision.hidden_size, image_size=cfg.vision.image_size,
                                      patch_size=cfg.vision.patch_size)

    @property
    def hidden_size(self):
        return self.vcfg.hidden_size

    @property
    def num_patches(self):
        return self.vcfg.num_patches

    def _struct(self):
        if self._cstruct is None:
            v = self.vcfg
            vm = self.vision_tower.vision_model
            n_run = v.num_used_layers
            layers = (_lib.VilaVitLayer * max(n_run, 1))()
            for i in range(n_run):
                l = getattr(vm.encoder.layers, str(i))
                L = layers[i]
                L.ln1_w, L.ln1_b = l.layer_norm1.weight.data_ptr(), l.layer_norm1.bias.data_ptr()
                L.wq, L.bq = l.self_attn.q_proj.weight.data_ptr(), l.self_attn.q_proj.bias.data_ptr()
                L.wk, L.bk = l.self_attn.k_proj.weight.data_ptr(), l.self_attn.k_proj.bias.data_ptr()
                L.wv, L.bv = l.self_attn.v_proj.weight.data_ptr(), l.self_attn.v_proj.bias.data_ptr()
                L.wo, L.bo = l.self_attn.out_proj.weight.data_ptr(), l.self_attn.out_proj.bias.data_ptr()
                L.ln2_w, L.ln2_b = l.layer_norm2.weight.data_ptr(), l.layer_norm2.bias.data_ptr()
                L.fc1_w, L.fc1_b = l.mlp.fc1.weight.data_ptr(), l.mlp.fc1.bias.data_ptr()
                L.fc2_w, L.fc2_b = l.mlp.fc2.weight.data_ptr(), l.mlp.fc2.bias.data_ptr()
            w = _lib.VilaVitWeights()
            w.shape = _lib.VilaVitShape(v.hidden_size, v.intermediate_size, v.num_attention_heads, v.image_size, v.patch_size,
                                        v.num_channels, n_run, v.layer_norm_eps)
            w.patch_w = vm.embeddings.patch_embedding.weight.data_ptr()
            w.patch_b = vm.embeddings.patch_embedding.bias.data_ptr()
            w.pos_emb = vm.embeddings.position_embedding.weight.data_ptr()
            w.layers = C.cast(layers, C.POINTER(_lib.VilaVitLayer))
            self._cstruct = (w, layers)
        return self._cstruct[0]

    def quantize_w8(self):
        """W8A8 (BASELINE configs[4]): build int8 per-channel copies of the encoder linears; forward() then runs
        `vila_vit_forward_w8a8` (int8 x int8 matrix-core GEMMs with per-token dynamic activation scales)."""
        from .quant import W8VitWeights
        self._w8 = W8VitWeights(self)
        return self._w8

    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        if isinstance(images, list):
            images = torch.stack(images, 0)
        ops._need(images, dtype=None, name="images")
        v = self.vcfg
        if images.shape[1:] != (v.num_channels, v.image_size, v.image_size):
            raise ValueError(f"Input image size ({images.shape[2]}*{images.shape[3]}) doesn't match model ({v.image_size}*{v.image_size}).")
        x = images.to(self.dtype).contiguous()
        Bn = x.shape[0]
        lib = _lib.load()
        w = self._struct()
        out = torch.empty((Bn, v.num_patches, v.hidden_size), device=x.device, dtype=self.dtype)
        w8 = getattr(self, "_w8", None)
        if w8 is not None:
            ws = self._workspace(lib.vila_vit_w8a8_workspace_bytes(C.byref(w.shape), Bn), x.device)
            check(lib.vila_vit_forward_w8a8(C.byref(w), w8.ptr, x.data_ptr(), Bn, out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
                  "vila_vit_forward_w8a8")
            return out.to(images.dtype) if images.dtype in (torch.float16, torch.bfloat16) else out
        ws = self._workspace(lib.vila_vit_workspace_bytes(C.byref(w.shape), Bn), x.device)
        check(lib.vila_vit_forward(C.byref(w), x.data_ptr(), Bn, out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "vila_vit_forward")
        return out.to(images.dtype) if images.dtype in (torch.float16, torch.bfloat16) else out


# --------------------------------------------------------------------------------------------------------------
# Projector
# --------------------------------------------------------------------------------------------------------------
_PROJ_KIND = {"mlp_downsample": 0, "mlp_downsample_2x2_fix": 1, "mlp_downsample_3x3_fix": 2}


class HipMultimodalProjector(_HipModule):
    """`MultimodalProjector.forward` (llava/model/multimodal_projector/base_projector.py:248-252)."""

    def __init__(self, cfg: VilaConfig, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        if cfg.mm_projector_type not in _PROJ_KIND:
            raise ValueError(f"Unknown projector type: {cfg.mm_projector_type}")
        self.cfg = cfg
        self.downsample_rate = cfg.downsample
        self._build_params(synthetic.projector_specs(cfg), "mm_projector.", device, dtype)

    def _struct(self):
        if self._cstruct is None:
            L = self.layers
            w = _lib.VilaProjWeights()
            w.kind = _PROJ_KIND[self.cfg.mm_projector_type]
            w.in_dim, w.out_dim = self.cfg.mm_hidden_size, self.cfg.llm.hidden_size   # dynamic_s2: C * n_scales
            g = lambda i, n: getattr(getattr(L, str(i)), n).data_ptr()
            w.ln1_w, w.ln1_b, w.fc1_w, w.fc1_b = g(1, "weight"), g(1, "bias"), g(2, "weight"), g(2, "bias")
            if w.kind == 2:
                w.ln2_w, w.ln2_b, w.fc2_w, w.fc2_b, w.fc3_w, w.fc3_b = (g(4, "weight"), g(4, "bias"), g(5, "weight"), g(5, "bias"),
                                                                        g(7, "weight"), g(7, "bias"))
            else:
                w.fc2_w, w.fc2_b = g(4, "weight"), g(4, "bias")
            self._cstruct = (w,)
        return self._cstruct[0]

    @torch.no_grad()
    def forward(self, x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        ops._need(x, dtype=None, name="features")
        xin = x.to(self.dtype).contiguous()
        Bn, N, _ = xin.shape
        lib = _lib.load()
        w = self._struct()
        ws = self._workspace(lib.vila_proj_workspace_bytes(C.byref(w), Bn, N), xin.device)
        No = lib.vila_proj_out_tokens(w.kind, N)
        out = torch.empty((Bn, No, self.cfg.llm.hidden_size), device=xin.device, dtype=self.dtype)
        check(lib.vila_proj_forward(C.byref(w), xin.data_ptr(), Bn, N, out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "vila_proj_forward")
        return out


# --------------------------------------------------------------------------------------------------------------
# LLM
# --------------------------------------------------------------------------------------------------------------
def _as_i64(u: int) -> int:
    """uint64 bit pattern -> the int64 holding the same bits (torch has no uint64 fill)."""
    u = int(u) & 0xFFFFFFFFFFFFFFFF
    return u - (1 << 64) if u >= (1 << 63) else u


class CausalLMOutput(SimpleNamespace):
    """Fields of HF CausalLMOutputWithPast that llava_llama.py:134-159 consumes."""


class HipQwen2ForCausalLM(_HipModule):
    """HF `Qwen2ForCausalLM` as used at llava_llama.py:134-141 (`forward`) and llava_arch.py:833 (`generate`)."""

    def __init__(self, cfg: VilaConfig, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        self.cfg = cfg
        self.lcfg: LlmConfig = cfg.llm
        self._build_params(synthetic.llm_specs(cfg), "llm.", device, dtype)
        self.config = SimpleNamespace(hidden_size=cfg.llm.hidden_size, vocab_size=cfg.llm.vocab_size,
                                      eos_token_id=cfg.llm.eos_token_id, tie_word_embeddings=cfg.llm.tie_word_embeddings)
        self._decode = None
        # `llm.model.embed_tokens(ids)` is called directly by the reference (llava_arch.py:429, encoders/image/basic.py:27)
        emb = self.model.embed_tokens
        emb.forward = lambda ids, _w=emb: ops.embed_tokens(_w.weight, ids)

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return ops.embed_tokens(self.model.embed_tokens.weight, ids)

    def _struct(self):
        if self._cstruct is None:
            c = self.lcfg
            layers = (_lib.VilaLlmLayer * c.num_hidden_layers)()
            for i in range(c.num_hidden_layers):
                l = getattr(self.model.layers, str(i))
                L = layers[i]
                L.ln1_w = l.input_layernorm.weight.data_ptr()
                L.wq, L.bq = l.self_attn.q_proj.weight.data_ptr(), l.self_attn.q_proj.bias.data_ptr()
                L.wk, L.bk = l.self_attn.k_proj.weight.data_ptr(), l.self_attn.k_proj.bias.data_ptr()
                L.wv, L.bv = l.self_attn.v_proj.weight.data_ptr(), l.self_attn.v_proj.bias.data_ptr()
                L.wo = l.self_attn.o_proj.weight.data_ptr()
                L.ln2_w = l.post_attention_layernorm.weight.data_ptr()
                L.w_gate, L.w_up, L.w_down = l.mlp.gate_proj.weight.data_ptr(), l.mlp.up_proj.weight.data_ptr(), l.mlp.down_proj.weight.data_ptr()
            w = _lib.VilaLlmWeights()
            w.shape = _lib.VilaLlmShape(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                                        c.num_key_value_heads, c.head_dim, c.vocab_size, c.rms_norm_eps, c.rope_theta)
            w.embed = self.model.embed_tokens.weight.data_ptr()
            w.layers = C.cast(layers, C.POINTER(_lib.VilaLlmLayer))
            w.norm_w = self.model.norm.weight.data_ptr()
            w.lm_head = (self.model.embed_tokens.weight if c.tie_word_embeddings else self.lm_head.weight).data_ptr()
            self._cstruct = (w, layers)
        return self._cstruct[0]

    # ---- KV cache ---------------------------------------------------------------------------------------
    def new_cache(self, max_ctx: int, n_slots: int = 1):
        c = self.lcfg
        shape = (c.num_hidden_layers, n_slots, c.num_key_value_heads, max_ctx, c.head_dim)
        with torch.inference_mode(False):        # caches are reused across calls (generate() keeps `_own_cache`)
            k = torch.zeros(shape, device=self.device, dtype=self.dtype)
            v = torch.zeros(shape, device=self.device, dtype=self.dtype)
        cs = _lib.VilaKvCache(k.data_ptr(), v.data_ptr(), max_ctx, n_slots)
        return SimpleNamespace(k=k, v=v, c=cs, max_ctx=max_ctx, n_slots=n_slots)

    # ---- packed prefill -------------------------------------------------------------------------------------
    @torch.no_grad()
    def prefill_packed(self, embeds: torch.Tensor, positions: torch.Tensor, cu_seqlens: Optional[torch.Tensor], max_seqlen: int,
                       cache=None, seq_of_tok: Optional[torch.Tensor] = None, last_rows: Optional[torch.Tensor] = None,
                       want_all_logits: bool = False, want_final_hidden: bool = False, want_layer_hidden: bool = False):
        """embeds [T,H] bf16 packed stream.  Returns a namespace with last_logits / all_logits / final_hidden / layer_hidden."""
        ops._need(embeds, name="inputs_embeds")
        c = self.lcfg
        T = embeds.shape[0]
        dev = embeds.device
        lib = _lib.load()
        w = self._struct()
        ws = self._workspace(lib.vila_llm_prefill_workspace_bytes(C.byref(w.shape), T), dev)
        n_seq = 1 if cu_seqlens is None else cu_seqlens.numel() - 1
        n_last = 0 if last_rows is None else last_rows.numel()
        last_logits = torch.empty((n_last, c.vocab_size), device=dev, dtype=torch.float32) if n_last else None
        all_logits = torch.empty((T, c.vocab_size), device=dev, dtype=torch.float32) if want_all_logits else None
        final_hidden = torch.empty((T, c.hidden_size), device=dev, dtype=self.dtype) if want_final_hidden else None
        layer_hidden = torch.empty((c.num_hidden_layers + 1, T, c.hidden_size), device=dev, dtype=self.dtype) if want_layer_hidden else None
        check(lib.vila_llm_prefill(C.byref(w), embeds.contiguous().data_ptr(), positions.data_ptr(), ops._p(cu_seqlens), n_seq, T,
                                   int(max_seqlen), ops._p(seq_of_tok), C.byref(cache.c) if cache is not None else None,
                                   ops._p(last_rows), n_last, ops._p(last_logits), ops._p(all_logits), ops._p(final_hidden),
                                   ops._p(layer_hidden), ws.data_ptr(), ws.numel(), ops._stream()), "vila_llm_prefill")
        return SimpleNamespace(last_logits=last_logits, all_logits=all_logits, final_hidden=final_hidden, layer_hidden=layer_hidden)

    # ---- HF-style forward (inference logits / loss; no autograd here: training goes through vila_amd.train) ----
    @torch.no_grad()
    def forward(self, input_ids=None, inputs_embeds: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, past_key_values=None, labels: Optional[torch.Tensor] = None,
                seqlens_in_batch: Optional[torch.Tensor] = None, num_items_in_batch: Optional[int] = None, **kw):
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        Bn, S, H = inputs_embeds.shape
        dev = inputs_embeds.device
        if seqlens_in_batch is not None:
            # packed row from repack_multimodal_data (llava_arch.py:744-800); packing.py:12-21 semantics
            assert Bn == 1
            seqlens = seqlens_in_batch.to(device=dev, dtype=torch.int32)
            keep = attention_mask[0].bool() if attention_mask is not None else torch.ones(S, dtype=torch.bool, device=dev)
        else:
            mask = attention_mask.bool() if attention_mask is not None else torch.ones((Bn, S), dtype=torch.bool, device=dev)
            seqlens = mask.sum(1).to(torch.int32)
            keep = mask.reshape(-1)
        idx = torch.nonzero(keep, as_tuple=False).flatten()
        flat = inputs_embeds.reshape(Bn * S, H)
        packed = flat.index_select(0, idx).to(self.dtype)
        cu = torch.zeros(seqlens.numel() + 1, device=dev, dtype=torch.int32)
        cu[1:] = torch.cumsum(seqlens, 0)
        if position_ids is not None:
            pos = position_ids.reshape(-1).index_select(0, idx).to(torch.int32)
        else:
            seq_id = torch.repeat_interleave(torch.arange(seqlens.numel(), device=dev), seqlens.long())
            pos = (torch.arange(idx.numel(), device=dev, dtype=torch.int32) - cu[:-1][seq_id]).to(torch.int32)
        want_logits = labels is None or bool(kw.get("return_logits", False))
        r = self.prefill_packed(packed, pos, cu, int(seqlens.max()), want_all_logits=want_logits, want_final_hidden=labels is not None)
        logits = None
        if want_logits:
            logits = torch.zeros((Bn * S, self.lcfg.vocab_size), device=dev, dtype=torch.float32)
            logits.index_copy_(0, idx, r.all_logits)
            logits = logits.view(Bn, S, -1)
        loss = None
        if labels is not None:
            # HF ForCausalLMLoss: position t predicts labels[t+1] inside each row; only rows WITH a target go through the head:
            # gather -> [n_valid, V] fp32 logits -> ce_kernel (no [B*S, V] tensor, no torch loss kernels on this path)
            lb = labels.to(dev)
            tgt = torch.full((Bn, S), IGNORE_INDEX, dtype=torch.int64, device=dev)
            tgt[:, :-1] = lb[:, 1:]
            inv = torch.full((Bn * S,), -1, dtype=torch.int64, device=dev)
            inv[idx] = torch.arange(idx.numel(), device=dev)
            tgt = tgt.reshape(-1)
            ok = (tgt != IGNORE_INDEX) & (inv >= 0)
            rows = inv[ok].to(torch.int32)
            n_valid = int(rows.numel())
            acc = torch.zeros((1,), device=dev, dtype=torch.float32)
            if n_valid:
                hv = torch.empty((n_valid, H), device=dev, dtype=self.dtype)
                ops.copy_rows(r.final_hidden, hv, rows, None, n_valid)
                head = self.model.embed_tokens.weight if self.lcfg.tie_word_embeddings else self.lm_head.weight
                lg = ops.gemm(hv, head, out_f32=True)
                denom = n_valid if num_items_in_batch is None else int(num_items_in_batch)
                ops.ce_loss(lg, tgt[ok].contiguous(), acc, 1.0 / max(denom, 1))
                loss = acc[0]
            else:
                loss = acc[0] * float("nan") if num_items_in_batch is None else acc[0]     # F.cross_entropy(mean) of no targets is nan
        return CausalLMOutput(loss=loss, logits=logits, past_key_values=None)

    # ---- greedy generate ---------------------------------------------------------------------------------------
    def _decode_session(self, cache, max_new_tokens: int, sampling=None):
        """Device-resident decode state + workspace (+ captured hipGraph) reused across generate() calls.
        sampling: None (greedy) or (temperature, top_k, top_p, seed) — baked into the captured graph, hence part of the key."""
        # the seed is NOT part of the key: it lives in a device scalar the sampler reads (VilaSampling.seed_dev), so every sampled request
        # with the same (temperature, top_k, top_p) replays the same captured graph (a fresh seed per call used to drop
        # the graph, the workspace and the weight struct on every sampled request)
        key = (cache.k.data_ptr(), max_new_tokens, self.model.embed_tokens.weight.data_ptr(),
               _get(self, "model.layers.0.mlp.down_proj.weight").data_ptr(), None if sampling is None else tuple(sampling[:3]))
        if self._decode is not None and self._decode.key == key:
            if sampling is not None:
                self._decode.seed.fill_(_as_i64(sampling[3]))
            return self._decode
        old = self._decode
        if old is not None and old.key[2:4] != key[2:4]:
            self._invalidate()      # the weight storage moved: nothing that baked in a pointer survives
        else:
            self._drop_decode_session()          # another cache / length / sampling setting: only this session and its graph
        dev = self.device
        lib = _lib.load()
        w = self._struct()
        st = SimpleNamespace(key=key, cache=cache)
        st.sampling = None
        # the session outlives the call that creates it: built outside inference mode even when the first generate() runs under
        # torch.inference_mode() (llava_arch.py:823), or a later no_grad caller could not update `pos` / `token` in place
        with torch.inference_mode(False):
            st.pos = torch.zeros(1, device=dev, dtype=torch.int32)
            st.token = torch.zeros(1, device=dev, dtype=torch.int64)
            st.out_ids = torch.zeros(max(max_new_tokens, 1), device=dev, dtype=torch.int64)
            st.n_out = torch.zeros(1, device=dev, dtype=torch.int32)
            st.logits = torch.zeros(self.lcfg.vocab_size, device=dev, dtype=torch.float32)
            st.ws = torch.empty((lib.vila_llm_decode_workspace_bytes(C.byref(w.shape), cache.max_ctx),), device=dev, dtype=torch.uint8)
            st.ws[:256].zero_()                       # word 0: the chained step's error flag (vila_llm_decode_chain_error)
            st.seed = torch.zeros(1, device=dev, dtype=torch.int64)          # the sampler's seed (bit pattern of a uint64)
        if sampling is not None:
            st.seed.fill_(_as_i64(sampling[3]))
            st.sampling = _lib.VilaSampling(float(sampling[0]), int(sampling[1]), float(sampling[2]), int(sampling[3]) & 0xFFFFFFFFFFFFFFFF,
                                            st.seed.data_ptr())
        st.c = _lib.VilaDecodeState(st.pos.data_ptr(), st.token.data_ptr(), st.out_ids.data_ptr(), st.n_out.data_ptr(),
                                    max(max_new_tokens, 1), st.logits.data_ptr())
        st.graph = None
        st.stream = torch.cuda.Stream(device=dev)
        self._decode = st
        return st

    def quantize_w4(self, keep_logical: bool = True):
        """Build int4 (group-128) copies of the five decoder projections: decode then runs the W4A16 GEMVs
        (vila_llm_decode_step_w4); prefill keeps using the bf16 weights."""
        from .quant import W4Weights
        self._w4 = W4Weights(self, keep_logical)
        self._invalidate()
        return self._w4

    def decode_step(self, cache, st) -> None:
        lib = _lib.load()
        w4 = getattr(self, "_w4", None)
        sp = getattr(st, "sampling", None)
        if w4 is not None:
            if sp is not None:
                check(lib.vila_llm_decode_step_w4_sample(C.byref(self._struct()), w4.ptr, C.byref(cache.c), C.byref(st.c), st.ws.data_ptr(),
                                                         st.ws.numel(), C.byref(sp), ops._stream()), "vila_llm_decode_step_w4_sample")
                return
            check(lib.vila_llm_decode_step_w4(C.byref(self._struct()), w4.ptr, C.byref(cache.c), C.byref(st.c), st.ws.data_ptr(),
                                              st.ws.numel(), ops._stream()), "vila_llm_decode_step_w4")
            return
        if sp is not None:
            check(lib.vila_llm_decode_step_sample(C.byref(self._struct()), C.byref(cache.c), C.byref(st.c), st.ws.data_ptr(), st.ws.numel(),
                                                  C.byref(sp), ops._stream()), "vila_llm_decode_step_sample")
            return
        check(lib.vila_llm_decode_step(C.byref(self._struct()), C.byref(cache.c), C.byref(st.c), st.ws.data_ptr(), st.ws.numel(),
                                       ops._stream()), "vila_llm_decode_step")

    # ---- batched greedy generate: one weight pass per step for up to 16 sequences (vila_llm_decode_step_batch) -----------------------
    def _can_batch_decode(self, inputs_embeds, attention_mask, max_new_tokens, do_sample, forced_ids, return_logits, cache) -> bool:
        c = self.lcfg
        Bn, S = inputs_embeds.shape[0], inputs_embeds.shape[1]
        return (not do_sample and forced_ids is None and not return_logits and cache is None and getattr(self, "_w4", None) is None and
                max_new_tokens >= 1 and self._qkv_fused() and 2 <= Bn <= 16 and c.head_dim == 128 and c.hidden_size % 64 == 0 and c.intermediate_size % 64 == 0 and
                ((S + max_new_tokens + 255) // 256) * 256 <= 2048)

    def _qkv_fused(self) -> bool:
        """q/k/v of every layer are views of one buffer (the batched step reads them as ONE [q + 2kv, hidden] matrix); `refuse()` re-establishes
        it after an op that re-allocated parameters — a model somebody un-fused falls back to the per-row loop instead of raising."""
        c = self.lcfg
        for i in range(c.num_hidden_layers):
            a = _get(self, f"model.layers.{i}.self_attn")
            q, k, v = a.q_proj.weight, a.k_proj.weight, a.v_proj.weight
            if k.data_ptr() != q.data_ptr() + q.numel() * q.element_size() or v.data_ptr() != k.data_ptr() + k.numel() * k.element_size():
                return False
        return True

    def _batch_session(self, n: int, max_ctx: int, max_new_tokens: int):
        key = (n, max_ctx, max_new_tokens, self.model.embed_tokens.weight.data_ptr(), _get(self, "model.layers.0.mlp.down_proj.weight").data_ptr())
        st = getattr(self, "_bdecode", None)
        if st is not None and st.key == key:
            return st
        if st is not None and st.graph is not None:
            _lib.load().vila_graph_destroy(st.graph)
        dev, lib, w = self.device, _lib.load(), self._struct()
        st = SimpleNamespace(key=key, graph=None)
        with torch.inference_mode(False):
            st.cache = self.new_cache(max_ctx, n_slots=n)
            st.pos = torch.zeros(n, device=dev, dtype=torch.int32)
            st.token = torch.zeros(n, device=dev, dtype=torch.int64)
            st.out_ids = torch.zeros((n, max(max_new_tokens, 1)), device=dev, dtype=torch.int64)
            st.n_out = torch.zeros(n, device=dev, dtype=torch.int32)
            st.logits = torch.zeros((n, self.lcfg.vocab_size), device=dev, dtype=torch.float32)
            st.ws = torch.empty((lib.vila_llm_decode_batch_workspace_bytes(C.byref(w.shape), n),), device=dev, dtype=torch.uint8)
        st.c = _lib.VilaDecodeBatch(n, st.pos.data_ptr(), st.token.data_ptr(), st.out_ids.data_ptr(), st.n_out.data_ptr(), max(max_new_tokens, 1),
                                    st.logits.data_ptr())
        st.stream = torch.cuda.Stream(device=dev)
        self._bdecode = st
        return st

    def _batch_step(self, st) -> None:
        check(_lib.load().vila_llm_decode_step_batch(C.byref(self._struct()), C.byref(st.cache.c), C.byref(st.c), st.ws.data_ptr(), st.ws.numel(),
                                                     ops._stream()), "vila_llm_decode_step_batch")

    # ---- continuous batching (server.py:171-290 serves requests as they arrive): rows join and leave BETWEEN steps ---------
    # `vila_llm_decode_step_batch` takes per-row positions / output counters, so admission is host work: prefill the newcomer alone into its
    # KV slot, set the row's position and first token, replay the same captured graph.  A free row idles at positions 0..15 of its own slot
    # (re-wound after every chunk), which costs nothing extra: the step streams the weights once whatever the number of live rows.
    def batch_open(self, n_slots: int, max_ctx: int = 2048, max_new_tokens: int = 1024):
        c = self.lcfg
        if not (1 <= n_slots <= 16 and c.head_dim == 128 and max_ctx <= 2048 and getattr(self, "_w4", None) is None and self._qkv_fused()):
            raise ValueError("batch_open: the batched decode step serves 1..16 rows of a bf16 head-dim-128 model with caches <= 2048 positions")
        st = self._batch_session(max(n_slots, 2), max_ctx, max_new_tokens)
        st.pos.zero_(); st.n_out.zero_(); st.token.zero_()
        if st.graph is None:
            lib = _lib.load()
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(st.stream):
                self._batch_step(st)                                   # warm-up outside capture (kernel attributes)
                st.stream.synchronize()
                st.pos.zero_(); st.n_out.zero_(); st.token.zero_()
                check(lib.vila_graph_begin(st.stream.cuda_stream), "graph_begin")
                self._batch_step(st)
                g = C.c_void_p()
                check(lib.vila_graph_end(st.stream.cuda_stream, C.byref(g)), "graph_end")
                st.graph = g
                st.stream.synchronize()
                st.pos.zero_(); st.n_out.zero_(); st.token.zero_()
        return st

    def batch_admit(self, st, slot: int, embeds: torch.Tensor) -> int:
        """Prefill ONE sequence [S, H] into KV slot `slot` of the open batch and make the row live: position S, first token = argmax of the
        prefill's last row (returned), output counter 0."""
        S = int(embeds.shape[0])
        if S + 1 > st.cache.max_ctx:
            raise ValueError(f"KV cache too small: {st.cache.max_ctx} < {S} + 1")
        dev = embeds.device
        st.stream.synchronize()                                        # no step is reading the cache while the newcomer's rows are written
        pos = torch.arange(S, device=dev, dtype=torch.int32)
        seq = torch.full((S,), int(slot), device=dev, dtype=torch.int32)
        cu = torch.tensor([0, S], device=dev, dtype=torch.int32)
        last = torch.full((1,), S - 1, device=dev, dtype=torch.int32)
        r = self.prefill_packed(embeds.to(self.dtype), pos, cu, S, cache=st.cache, seq_of_tok=seq, last_rows=last)
        first = ops.argmax(r.last_logits[0])
        st.pos[slot:slot + 1].fill_(S)
        st.n_out[slot:slot + 1].zero_()
        st.token[slot:slot + 1].copy_(first)
        tok = int(first.item())                                        # (also orders the prefill before the next replay on st.stream)
        return tok

    def batch_run(self, st, k: int) -> None:
        lib = _lib.load()
        torch.cuda.current_stream().synchronize()
        with torch.cuda.stream(st.stream):
            for _ in range(int(k)):
                check(lib.vila_graph_launch(st.graph, st.stream.cuda_stream), "graph_launch")
        st.stream.synchronize()

    def batch_release(self, st, slots) -> None:
        """Rows that finished (or never started): re-wound to position 0 of their own slot so an idle row never walks off its cache."""
        slots = list(slots)
        if slots:
            idx = torch.tensor(slots, device=st.pos.device, dtype=torch.int64)
            st.pos.index_fill_(0, idx, 0)
            st.n_out.index_fill_(0, idx, 0)

    def _generate_batch(self, inputs_embeds, attention_mask, max_new_tokens, eos_token_id, pad_token_id, use_graph: bool = True,
                        forced_ids: Optional[torch.Tensor] = None, return_logits: bool = False):
        """The padded batch as ONE packed prefill (every row into its own KV-cache slot) + batched decode steps: the weights are streamed
        once per step for all rows.  Returns [B, n_new] right-padded with pad_token_id behind each row's EOS, like HF.
        forced_ids [B, n_new] / return_logits: teacher forcing for the parity tests (eager launches; the ids fed after step t are
        forced_ids[:, t]); returns (argmax ids [B, n_new], logits [n_new, B, V] fp32)."""
        ops._need(inputs_embeds, dtype=None, name="inputs_embeds")
        Bn, S, H = inputs_embeds.shape
        dev = inputs_embeds.device
        mask = attention_mask.bool() if attention_mask is not None else torch.ones((Bn, S), dtype=torch.bool, device=dev)
        lens = [int(v) for v in mask.sum(1).tolist()]
        packed = inputs_embeds.reshape(Bn * S, H)[mask.reshape(-1)].to(self.dtype).contiguous()
        cu_h = [0]
        for n in lens:
            cu_h.append(cu_h[-1] + n)
        cu = torch.tensor(cu_h, device=dev, dtype=torch.int32)
        pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).to(dev)
        seq = torch.cat([torch.full((n,), b, dtype=torch.int32) for b, n in enumerate(lens)]).to(dev)
        last = torch.tensor([c - 1 for c in cu_h[1:]], device=dev, dtype=torch.int32)
        max_ctx = ((max(lens) + max_new_tokens + 255) // 256) * 256
        st = self._batch_session(Bn, max_ctx, max_new_tokens)
        r = self.prefill_packed(packed, pos, cu, max(lens), cache=st.cache, seq_of_tok=seq, last_rows=last)
        first = torch.cat([ops.argmax(r.last_logits[b]) for b in range(Bn)])
        st.pos.copy_(torch.tensor(lens, dtype=torch.int32))
        st.n_out.zero_()
        st.token.copy_(first)
        eos = self.lcfg.eos_token_id if eos_token_id is None else eos_token_id
        eos_set = set(eos) if isinstance(eos, (list, tuple)) else {eos}
        n_steps = max_new_tokens - 1
        lib = _lib.load()
        if forced_ids is not None or return_logits:
            step_logits, ids = [r.last_logits[:Bn].float().clone()], [first]
            if forced_ids is not None:
                st.token.copy_(forced_ids[:, 0].to(dev))
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(st.stream):
                for t in range(n_steps):
                    self._batch_step(st)
                    step_logits.append(st.logits[:Bn].float().clone())
                    ids.append(torch.cat([ops.argmax(st.logits[b]) for b in range(Bn)]))
                    if forced_ids is not None and t + 1 < forced_ids.shape[1]:
                        st.token.copy_(forced_ids[:, t + 1].to(dev))
            st.stream.synchronize()
            return torch.stack(ids, 1), torch.stack(step_logits, 0)
        if use_graph and st.graph is None and n_steps > 0:
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(st.stream):
                self._batch_step(st)                                   # warm-up outside capture (kernel attributes), then restore the state
                st.stream.synchronize()
                st.pos.copy_(torch.tensor(lens, dtype=torch.int32)); st.n_out.zero_(); st.token.copy_(first)
                check(lib.vila_graph_begin(st.stream.cuda_stream), "graph_begin")
                self._batch_step(st)
                g = C.c_void_p()
                check(lib.vila_graph_end(st.stream.cuda_stream, C.byref(g)), "graph_end")
                st.graph = g
                st.stream.synchronize()
        first_h = first.tolist()
        done_rows = [t in eos_set for t in first_h]
        done = 0
        torch.cuda.current_stream().synchronize()
        with torch.cuda.stream(st.stream):
            while done < n_steps and not all(done_rows):
                chunk = min(16, n_steps - done)
                for _ in range(chunk):
                    if use_graph:
                        check(lib.vila_graph_launch(st.graph, st.stream.cuda_stream), "graph_launch")
                    else:
                        self._batch_step(st)
                done += chunk
                got = st.out_ids[:, :done].tolist()                 # one sync per 16 steps
                done_rows = [first_h[b] in eos_set or any(t in eos_set for t in got[b]) for b in range(Bn)]
        st.stream.synchronize()
        toks = torch.cat([first[:, None], st.out_ids[:, :done]], 1)
        eos1 = eos[0] if isinstance(eos, (list, tuple)) else eos
        pad = pad_token_id if pad_token_id is not None else (eos1 if eos1 is not None else self.lcfg.eos_token_id)
        rows = []
        for b, row in enumerate(toks.tolist()):
            cut = next((i + 1 for i, t in enumerate(row) if t in eos_set), len(row))      # HF stops a row AFTER emitting eos
            rows.append(row[:cut])
        n = max(len(r) for r in rows)
        out = torch.full((Bn, n), int(pad), dtype=torch.int64, device=dev)
        for b, r_ in enumerate(rows):
            out[b, :len(r_)] = torch.tensor(r_, dtype=torch.int64, device=dev)
        return out

    @torch.no_grad()
    def generate(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, max_new_tokens: Optional[int] = None,
                 eos_token_id=None, do_sample: Optional[bool] = None, temperature: Optional[float] = None, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, seed: Optional[int] = None, pad_token_id: Optional[int] = None, generation_config=None,
                 use_graph: bool = True, return_logits: bool = False, forced_ids: Optional[torch.Tensor] = None, cache=None,
                 max_length: Optional[int] = None, streamer=None, **kw):
        """`llm.generate(inputs_embeds=, attention_mask=, **generation_kwargs)` as called at llava_arch.py:833 (HF semantics: returns ONLY the
        new tokens, [B, n_new]).  Greedy search or sampling (do_sample: temperature / top_k / top_p, HF order, on the device); explicit
        keyword arguments override `generation_config` (HF GenerationConfig-like: do_sample, temperature, top_k, top_p, max_new_tokens,
        eos_token_id, pad_token_id).  The whole step (28 layers + lm_head + token choice + position advance) is one hipGraph replay; the host
        polls for EOS every 16 tokens.  A padded batch (B > 1) is served one sequence at a time through the same cache and graph (each row
        costs a batch-1 decode: weights are streamed once per row and token) and right-padded with pad_token_id like HF.
        forced_ids = teacher forcing for margin-aware parity tests.
        streamer = HF's `generate(streamer=...)` contract (what server.py:243 streams from): `put(LongTensor[1])` once per new token — the EOS
        included, nothing for a prompt given as embeddings — as the host learns of them (every 16 graph replays), then `end()`; batch size 1."""
        if streamer is not None and inputs_embeds.shape[0] > 1:
            raise ValueError("TextStreamer only supports batch size 1")          # transformers/generation/streamers.py, TextStreamer.put
        gc = generation_config
        pick = lambda v, name, default: v if v is not None else (getattr(gc, name, None) if gc is not None and getattr(gc, name, None) is not None else default)
        do_sample = bool(pick(do_sample, "do_sample", False))
        temperature, top_k, top_p = float(pick(temperature, "temperature", 1.0)), int(pick(top_k, "top_k", 50)), float(pick(top_p, "top_p", 1.0))
        max_new_tokens = int(pick(max_new_tokens, "max_new_tokens", 32))
        eos_token_id = pick(eos_token_id, "eos_token_id", None)
        pad_token_id = pick(pad_token_id, "pad_token_id", None)
        if do_sample and temperature <= 0:
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float, otherwise your next token scores will be invalid.")
        if do_sample and top_k < 0:
            raise ValueError(f"`top_k` has to be a non-negative integer, but is {top_k}")       # (0 = no top-k filter, like HF)
        if inputs_embeds.shape[0] > 1 and self._can_batch_decode(inputs_embeds, attention_mask, max_new_tokens, do_sample, forced_ids, return_logits, cache):
            return self._generate_batch(inputs_embeds, attention_mask, max_new_tokens, eos_token_id, pad_token_id, use_graph)
        if inputs_embeds.shape[0] > 1:
            rows = []
            for b in range(inputs_embeds.shape[0]):
                m = None if attention_mask is None else attention_mask[b:b + 1]
                rows.append(self.generate(inputs_embeds[b:b + 1], m, max_new_tokens, eos_token_id, do_sample, temperature, top_k, top_p,
                                          None if seed is None else seed + b, pad_token_id, None, use_graph, False, None, cache, max_length)[0])
            eos1 = eos_token_id[0] if isinstance(eos_token_id, (list, tuple)) else eos_token_id
            pad = pad_token_id if pad_token_id is not None else (eos1 if eos1 is not None else self.lcfg.eos_token_id)
            n = max(int(r.numel()) for r in rows)
            out = torch.full((len(rows), n), int(pad), dtype=torch.int64, device=rows[0].device)
            for b, r in enumerate(rows):
                out[b, : r.numel()] = r
            return out
        sampling = None
        if do_sample:
            if seed is None:
                self._sample_calls = getattr(self, "_sample_calls", 0) + 1
                seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * self._sample_calls) & 0xFFFFFFFFFFFFFFFF
            sampling = (temperature, top_k, top_p, int(seed))
        ops._need(inputs_embeds, dtype=None, name="inputs_embeds")
        x = inputs_embeds[0]
        if attention_mask is not None:
            x = x[attention_mask[0].bool()]
        S = x.shape[0]
        dev = x.device
        eos = self.lcfg.eos_token_id if eos_token_id is None else eos_token_id
        eos_set = set(eos) if isinstance(eos, (list, tuple)) else {eos}
        if cache is None:
            cache = getattr(self, "_own_cache", None)
            if cache is None or cache.max_ctx < S + max_new_tokens:
                cache = self._own_cache = self.new_cache(((S + max_new_tokens + 255) // 256) * 256)
        if cache.max_ctx < S + max_new_tokens:
            raise ValueError(f"KV cache too small: {cache.max_ctx} < {S} + {max_new_tokens}")
        pos = torch.arange(S, device=dev, dtype=torch.int32)
        last = torch.full((1,), S - 1, device=dev, dtype=torch.int32)          # (a fill kernel, not a host copy)
        r = self.prefill_packed(x.to(self.dtype), pos, None, S, cache=cache, last_rows=last)
        st = self._decode_session(cache, max_new_tokens, sampling)
        # the sampler mixes a device counter into its random number: the decode steps use the position of the token they consume
        # (S, S+1, ...); the first token, drawn from the prefill's logits, uses S - 1
        first = (ops.argmax(r.last_logits[0]) if sampling is None else
                 ops.sample(r.last_logits[0], temperature, top_k, top_p, sampling[3], counter=last))
        step_logits = [r.last_logits[0].clone()] if return_logits else None
        st.pos.fill_(S)
        st.n_out.zero_()
        st.token.copy_(first if forced_ids is None else forced_ids[:1].to(dev))
        ids = [first]
        lib = _lib.load()
        eager = (not use_graph) or return_logits or (forced_ids is not None)
        if not eager and st.graph is None:
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(st.stream):
                # warm-up launch outside capture (sets kernel attributes), then restore the state it advanced
                self.decode_step(cache, st)
                st.stream.synchronize()
                st.pos.fill_(S); st.n_out.zero_(); st.token.copy_(first)
                check(lib.vila_graph_begin(st.stream.cuda_stream), "graph_begin")
                self.decode_step(cache, st)
                g = C.c_void_p()
                check(lib.vila_graph_end(st.stream.cuda_stream, C.byref(g)), "graph_end")
                st.graph = g
                st.stream.synchronize()
        n_steps = max_new_tokens - 1
        if eager:
            for t in range(n_steps):
                self.decode_step(cache, st)
                if return_logits:
                    step_logits.append(st.logits.clone())
                    ids.append(ops.argmax(st.logits) if sampling is None else st.token.clone())
                if forced_ids is not None and t + 1 < forced_ids.numel():
                    st.token.copy_(forced_ids[t + 1:t + 2].to(dev))
            if not return_logits:
                out = torch.cat([first, st.out_ids[:n_steps]])
            else:
                out = torch.cat(ids)
        else:
            torch.cuda.current_stream().synchronize()
            done = 0
            first_id = first.item()
            stop = first_id in eos_set
            if streamer is not None:
                streamer.put(torch.tensor([first_id], dtype=torch.int64))
            with torch.cuda.stream(st.stream):
                while done < n_steps and not stop:
                    chunk = min(16, n_steps - done)
                    for _ in range(chunk):
                        check(lib.vila_graph_launch(st.graph, st.stream.cuda_stream), "graph_launch")
                    got = st.out_ids[:done + chunk].tolist()       # one sync per 16 tokens
                    if streamer is not None:
                        for t in got[done:]:
                            streamer.put(torch.tensor([t], dtype=torch.int64))
                            if t in eos_set:
                                break
                    done += chunk
                    stop = any(t in eos_set for t in got)
            st.stream.synchronize()
            out = torch.cat([first, st.out_ids[:done]])
        toks = out.tolist()
        if int(st.ws[:4].view(torch.int32).item()) != 0:    # (the stream is drained: out.tolist() above)
            st.ws[:4].zero_()
            raise RuntimeError("decode step: a bounded device-side wait gave up (the persistent token kernel's grid barrier, or the chained step's "
                               "wait for its predecessor), the generated tokens are invalid (vila_llm_decode_chain_error; VILA_DECODE_PERSIST=0 / "
                               "VILA_DECODE_CHAIN=0 select the plain per-kernel step)")
        if forced_ids is None:
            for i, t in enumerate(toks):                    # HF stops AFTER emitting eos
                if t in eos_set:
                    out = out[: i + 1]
                    break
        if streamer is not None:
            if eager:                                       # the step-by-step paths learn of their tokens at the end
                for t in out.tolist():
                    streamer.put(torch.tensor([t], dtype=torch.int64))
            streamer.end()
        out = out[None]
        return (out, torch.stack(step_logits)) if return_logits else out
