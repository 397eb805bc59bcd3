"""The VLM glue of the reference re-stated over the HIP modules: `LlavaLlamaModel` / `LlavaMetaForCausalLM`
(llava/model/language_model/llava_llama.py:41-159, llava/model/llava_arch.py:51-95,366-394,412-555,744-833) and
`BasicImageEncoder` (llava/model/encoders/image/basic.py:11-79).

Differences that are deliberate: the splice of `_embed` is index arithmetic + two row-gather kernels
instead of a python loop with `.item()` per token; greedy generation never leaves the device between tokens.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .configs import IGNORE_INDEX, VilaConfig
from .host import s2_plan, splice_plan
from .modules import HipMultimodalProjector, HipQwen2ForCausalLM, HipSiglipVisionTower


class BasicImageEncoder(nn.Module):
    """encoders/image/basic.py:11-79: stack -> parent.encode_images -> append embed(tokenizer(end_tokens))."""

    def __init__(self, parent: "HipLlavaLlamaModel", start_tokens: Optional[str] = None, end_tokens: Optional[str] = "\n"):
        super().__init__()
        object.__setattr__(self, "_parent", parent)
        self.start_tokens, self.end_tokens = start_tokens, end_tokens

    @property
    def parent(self):
        return self._parent

    def embed_tokens(self, tokens: Optional[str]) -> Optional[torch.Tensor]:
        if tokens is None:
            return None
        # the token ids live on the device after the first call: `torch.tensor(ids, device=...)` is a pageable host copy, i.e. the host
        # blocks until the stream has drained — behind the tower that was the largest gap of the TTFT timeline.  (The embedding itself is
        # looked up every time: the table may have been trained in between.)
        cache = self.__dict__.setdefault("_tok_ids_dev", {})
        key = (tokens, str(self.parent.device))                          # (a model moved with .to() must not look up with ids on the old device)
        ids = cache.get(key)
        if ids is None:
            with torch.inference_mode(False):
                ids = cache[key] = torch.tensor(self.parent.tokenizer(tokens).input_ids, device=self.parent.device)
        return self.parent.llm.embed_tokens(ids)

    def forward(self, images: List[torch.Tensor], config: Dict[str, Any]) -> List[torch.Tensor]:
        images = torch.stack(list(images), dim=0)
        start, end = self.embed_tokens(self.start_tokens), self.embed_tokens(self.end_tokens)      # enqueued ahead of the tower: off the critical path
        feats = self.parent.encode_images(images, block_sizes=config.get("block_sizes") if config else None)
        out = []
        for f in feats:
            parts = ([start] if start is not None else []) + [f] + ([end] if end is not None else [])
            out.append(torch.cat(parts, 0) if len(parts) > 1 else f)
        return out


class BasicVideoEncoder(nn.Module):
    """encoders/video/basic.py:13-53: every video [n_frames, 3, H, W] -> encode_images on all frames of all videos at once -> per frame
    [start tokens | frame tokens | end tokens], flattened.  The assembly is one `vila_video_pool_bf16` launch per video (pool 1,1,1)."""

    pool_sizes = ((1, 1, 1),)
    sep_tokens: Optional[str] = None

    def __init__(self, parent: "HipLlavaLlamaModel", start_tokens: Optional[str] = None, end_tokens: Optional[str] = "\n"):
        super().__init__()
        object.__setattr__(self, "_parent", parent)
        self.start_tokens, self.end_tokens = start_tokens, end_tokens

    @property
    def parent(self):
        return self._parent

    def embed_tokens(self, tokens: Optional[str]) -> Optional[torch.Tensor]:
        if tokens is None:
            return None
        # the token ids live on the device after the first call: `torch.tensor(ids, device=...)` is a pageable host copy, i.e. the host
        # blocks until the stream has drained — behind the tower that was the largest gap of the TTFT timeline.  (The embedding itself is
        # looked up every time: the table may have been trained in between.)
        cache = self.__dict__.setdefault("_tok_ids_dev", {})
        key = (tokens, str(self.parent.device))                          # (a model moved with .to() must not look up with ids on the old device)
        ids = cache.get(key)
        if ids is None:
            with torch.inference_mode(False):
                ids = cache[key] = torch.tensor(self.parent.tokenizer(tokens).input_ids, device=self.parent.device)
        return self.parent.llm.embed_tokens(ids)

    def _process_features(self, feats: torch.Tensor, start, end, sep) -> torch.Tensor:
        """feats [nt, ns, H] of one video -> token block (tsp.py:28-52 / basic.py:30-41)."""
        feats = feats.to(self.parent.dtype)
        outs = []
        for pool in self.pool_sizes:
            outs.append(ops.video_pool(feats, pool, start, end))
            if sep is not None:
                outs.append(sep)
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def forward(self, videos: List[torch.Tensor], config: Dict[str, Any]) -> List[torch.Tensor]:
        num_frames = [int(v.shape[0]) for v in videos]
        images = torch.cat(list(videos), dim=0)
        start, end, sep = self.embed_tokens(self.start_tokens), self.embed_tokens(self.end_tokens), self.embed_tokens(self.sep_tokens)
        feats = self.parent.encode_images(images)
        return [self._process_features(f, start, end, sep) for f in torch.split(feats, num_frames)]


class TSPVideoEncoder(BasicVideoEncoder):
    """encoders/video/tsp.py:14-64: mean-pool the projected frames over (t, h, w) windows for every entry of `pool_sizes`
    (NVILA-Video: [[8, 1, 1]], scripts/NVILA/stage4.sh:50), then the start/end tokens per pooled frame and an optional separator."""

    def __init__(self, parent: "HipLlavaLlamaModel", pool_sizes, start_tokens: Optional[str] = None, end_tokens: Optional[str] = "\n",
                 sep_tokens: Optional[str] = None):
        super().__init__(parent, start_tokens=start_tokens, end_tokens=end_tokens)
        self.pool_sizes = tuple(tuple(int(x) for x in p) for p in pool_sizes)
        self.sep_tokens = sep_tokens


class _SyntheticTokenizer:
    """Stands in for the HF tokenizer the reference attaches (`self.tokenizer`): only what the hot path touches —
    `media_token_ids`, `tokenizer("\\n").input_ids`, `padding_side`, `model_max_length`."""

    def __init__(self, cfg: VilaConfig, model_max_length: int = 8192):
        self.media_token_ids = {"image": cfg.image_token_id, "video": cfg.video_token_id}
        self.padding_side = "right"
        self.model_max_length = model_max_length
        self.eos_token_id = cfg.llm.eos_token_id
        self._nl = cfg.newline_token_id

    def __call__(self, text: str):
        if text != "\n":
            raise NotImplementedError("synthetic tokenizer only knows the image end token \"\\n\"")
        return SimpleNamespace(input_ids=[self._nl])


class HipLlavaLlamaModel(nn.Module):
    """Drop-in for `LlavaLlamaModel`: same attribute names (llm, vision_tower, mm_projector, encoders, tokenizer) and the
    same `encode_images` / `_embed` / `forward` / `generate` contracts."""

    def __init__(self, cfg: VilaConfig, device="cuda", dtype=torch.bfloat16, tokenizer=None):
        super().__init__()
        self.cfg = cfg
        self.llm = HipQwen2ForCausalLM(cfg, device, dtype)
        self.vision_tower = HipSiglipVisionTower(cfg, device, dtype)
        self.mm_projector = HipMultimodalProjector(cfg, device, dtype)
        self.tokenizer = tokenizer if tokenizer is not None else _SyntheticTokenizer(cfg)
        # configuration_llava.py:67-68: image_encoder / video_encoder hydra targets; the defaults are the two Basic encoders
        self.encoders = {"image": BasicImageEncoder(self), "video": BasicVideoEncoder(self)}
        self.training = False

    @property
    def device(self):
        return self.llm.device

    @property
    def dtype(self):
        return self.llm.dtype

    def get_vision_tower(self):
        return self.vision_tower

    def get_mm_projector(self):
        return self.mm_projector

    def load_weights(self, w: Dict[str, torch.Tensor]) -> None:
        self.llm.load_weights(w, "llm.")
        self.vision_tower.load_weights(w, "vision_tower.")
        self.mm_projector.load_weights(w, "mm_projector.")

    # llava_arch.py:366-394 (plain branch; dynamic_s2 is SURVEY §8f "next")
    def encode_images(self, images: torch.Tensor, block_sizes=None):
        if not getattr(self.cfg, "dynamic_s2", False):
            return self.get_mm_projector()(self.get_vision_tower()(images))
        # dynamic_s2 (llava_arch.py:369-390): tower on every tile of every scale, one merge kernel, projector on the
        # C*n_scales-wide blocks, chessboard re-merge of the projected blocks as a row gather
        cfg = self.cfg
        if block_sizes is None:
            block_sizes = [None] * len(images)
        plan = s2_plan(block_sizes, list(cfg.s2_scales), cfg.vision.grid, cfg.downsample, cfg.s2_resize_output_to_scale_idx)
        feats = self.get_vision_tower()(images)
        if plan.n_tiles != feats.shape[0]:
            raise AssertionError(f"The number of blocks ({plan.n_tiles}) does not match length of image_features ({feats.shape[0]})!")
        x = ops.s2_merge(feats.to(self.dtype), plan.desc.to(feats.device), len(cfg.s2_scales), plan.splits)
        y = self.get_mm_projector()(x)                                  # [n_blocks, g'^2, H]
        flat = y.reshape(-1, y.shape[-1])
        outs = []
        for perm in plan.perms:
            o = torch.empty((perm.numel(), flat.shape[1]), device=flat.device, dtype=flat.dtype)
            ops.copy_rows(flat, o, perm.to(flat.device), None, int(perm.numel()))
            outs.append(o)
        if all(o.shape[0] == outs[0].shape[0] for o in outs):
            return torch.stack(outs, 0)
        return outs

    # llava_arch.py:412-490 + 528-555
    def _embed(self, input_ids: torch.Tensor, media: Dict[str, List[torch.Tensor]], media_config: Optional[Dict[str, Dict[str, Any]]] = None,
               labels: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None, max_length: Optional[int] = None):
        dev = self.device
        media_config = media_config or {}
        # The integer work of the splice runs on the HOST while the tower runs on the GPU: ids / mask / labels are brought over
        # first (the GPU is idle, the copy costs one short sync), then the encoders are enqueued, then the plan is computed on the
        # CPU under them and only its small index tensors go back.  (Planning on device tensors forced 3-4 stream syncs — boolean
        # indexing, .sum() — each waiting for the whole tower.)
        ids_h = input_ids.cpu()
        labels_h = labels.cpu() if labels is not None else torch.full_like(ids_h, IGNORE_INDEX)
        mask_h = attention_mask.cpu().bool() if attention_mask is not None else torch.ones_like(ids_h, dtype=torch.bool)
        # __embed_media_tokens (llava_arch.py:492-517): one deque of embedding blocks per media name, in the caller's dict order
        embeds: Dict[str, List[torch.Tensor]] = {}
        for name in (media or {}):
            items = list(media[name])
            if items:
                embeds[name] = list(self.encoders[name](items, media_config.get(name, {})))
        tok_ids = dict(self.tokenizer.media_token_ids)
        B, L = ids_h.shape
        H = self.cfg.llm.hidden_size

        # ---- integer work on the ids; no per-token sync: vila_amd.host.splice_plan ----
        plan = splice_plan(ids_h, mask_h, labels_h, {n: [int(m.shape[0]) for m in embeds.get(n, [])] for n in tok_ids}, tok_ids,
                           getattr(self.tokenizer, "padding_side", "right"),
                           max_length=max_length)                                      # __truncate_sequence, llava_arch.py:519-526
        S = plan.S
        out = torch.zeros((B * S, H), device=dev, dtype=self.dtype)
        table = self.llm.model.embed_tokens.weight
        # the plan's index tensors go back in ONE pinned, non-blocking copy: the host never waits for the tower here, so the splice and the
        # prefill are enqueued behind it without a gap (six pageable .to(dev) calls each blocked until the stream reached them: ~0.5 ms of idle
        # GPU in the round-2 TTFT timeline)
        mask_all = bool(plan.mask.all())
        txt_src, txt_dst, img_src, img_dst, labels_d, mask_d = self._plan_to_device(
            [plan.txt_src, plan.txt_dst, plan.img_src, plan.img_dst, plan.labels, plan.mask], dev)
        ops.copy_rows(table, out, txt_src, txt_dst, int(plan.txt_src.numel()))
        blocks = [m for n in tok_ids for m in embeds.get(n, [])]
        if blocks:
            flat = (blocks[0] if len(blocks) == 1 else torch.cat(blocks, 0)).to(self.dtype)
            ops.copy_rows(flat, out, None if plan.img_src_identity else img_src, img_dst, int(plan.img_dst.numel()))
        mask_d._vila_all_true = mask_all               # host knowledge for generate(): no device round trip to learn that nothing is padded
        return out.view(B, S, H), labels_d, mask_d

    def _plan_to_device(self, tensors, dev):
        """Host tensors -> device tensors of the same dtype / shape through one cached pinned staging buffer and ONE non-blocking copy."""
        offs, n = [], 0
        for t in tensors:
            n = (n + 15) & ~15
            offs.append(n)
            n += t.numel() * t.element_size()
        n = max((n + 15) & ~15, 16)
        pin = getattr(self, "_plan_pin", None)
        if pin is None or pin.numel() < n:
            with torch.inference_mode(False):          # a staging buffer born inside generate()'s inference mode could not be refilled outside it
                pin = self._plan_pin = torch.empty((max(n, 1 << 16),), dtype=torch.uint8, pin_memory=True)
            self._plan_ev = None
        if getattr(self, "_plan_ev", None) is not None:
            self._plan_ev.synchronize()                # the previous call's copy has left the staging buffer
        for t, o in zip(tensors, offs):
            nb = t.numel() * t.element_size()
            if nb:
                pin[o:o + nb].copy_(t.contiguous().reshape(-1).view(torch.uint8))
        d = pin[:n].to(dev, non_blocking=True)
        self._plan_ev = torch.cuda.Event()
        self._plan_ev.record()
        return [d[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for t, o in zip(tensors, offs)]

    def enable_autograd(self, use_c_abi: Optional[bool] = None, group=None, tune_language_model: bool = True, tune_vision_tower: bool = True,
                        tune_mm_projector: bool = True):
        """Make the reference's own training call site work after the swap (llava/train/transformer_normalize_monkey_patch.py
        :183-249: `loss = model(**inputs).loss` ... `accelerator.backward(loss)`): every parameter gets requires_grad, and in training
        mode `forward(labels=...)` returns a loss that is attached to the autograd graph — its backward deposits the gradients the HIP
        step computed into `.grad` (accumulating), so `loss.backward()`, gradient accumulation, `optimizer.step()` of any torch optimizer
        and `zero_grad()` behave as with the reference's modules.  Parameters move into one flat buffer (vila_amd.train.FlatParams).
        tune_*: the reference's stage switches (llava/train/args.py tune_language_model / tune_vision_tower / tune_mm_projector; stage 1 trains the
        projector alone): a frozen component's parameters keep requires_grad = False and never receive a `.grad`; the tower's backward is not
        run at all when the tower is frozen, nor the projector's when projector and tower both are."""
        from .train import AutogradSeam
        self._seam = AutogradSeam(self, use_c_abi=use_c_abi, group=group,
                                  tune={"llm.": tune_language_model, "vision_tower.": tune_vision_tower, "mm_projector.": tune_mm_projector})
        return self._seam

    # llava_llama.py:94-159.  Inference / eval: loss without autograd.  Training mode with enable_autograd(): the SFT step behind autograd.
    def forward(self, input_ids=None, media=None, media_config=None, attention_mask=None, labels=None, packing: bool = True,
                inputs_embeds=None, num_items_in_batch=None, **kw):
        seam = getattr(self, "_seam", None)
        if seam is not None and self.training and labels is not None and inputs_embeds is None and torch.is_grad_enabled():
            if any(k not in ("image", "video") and len(v) for k, v in (media or {}).items()):
                raise NotImplementedError("training through the autograd seam takes image and video media")
            images = list((media or {}).get("image", []))
            videos = list((media or {}).get("video", []))            # BasicVideoEncoder: frames train like image tiles (train.py _with_videos)
            blocks = ((media_config or {}).get("image", {}) or {}).get("block_sizes")
            from .modules import CausalLMOutput
            return CausalLMOutput(loss=seam.loss(input_ids, images, labels, attention_mask, num_items_in_batch, blocks, videos), logits=None,
                                  past_key_values=None)
        with torch.no_grad():
            return self._forward_eval(input_ids, media, media_config, attention_mask, labels, inputs_embeds, num_items_in_batch)

    def _forward_eval(self, input_ids, media, media_config, attention_mask, labels, inputs_embeds, num_items_in_batch):
        if inputs_embeds is None:
            # the reference truncates to tokenizer.model_max_length only in training mode (llava_arch.py:522)
            cut = getattr(self.tokenizer, "model_max_length", None) if (self.training and labels is not None) else None
            inputs_embeds, labels, attention_mask = self._embed(input_ids, media, media_config, labels, attention_mask, max_length=cut)
        return self.llm(inputs_embeds=inputs_embeds, attention_mask=attention_mask, labels=labels, num_items_in_batch=num_items_in_batch)

    # llava_arch.py:823-833
    @torch.inference_mode()
    def generate(self, input_ids: Optional[torch.Tensor] = None, media: Optional[Dict[str, List[torch.Tensor]]] = None,
                 media_config: Optional[Dict[str, Dict[str, Any]]] = None, attention_mask: Optional[torch.Tensor] = None,
                 **generation_kwargs):
        inputs_embeds, _, attention_mask = self._embed(input_ids, media, media_config, None, attention_mask)
        if getattr(attention_mask, "_vila_all_true", False):
            attention_mask = None                      # nothing is padded: the LLM need not compact rows (a device round trip) before the prefill
        return self.llm.generate(inputs_embeds=inputs_embeds, attention_mask=attention_mask, **generation_kwargs)


def build_model(cfg: VilaConfig, weights: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0, device="cuda",
                dtype=torch.bfloat16, draw_device=None) -> HipLlavaLlamaModel:
    """Construct the VLM and fill it with `weights` (reference state_dict names) or seeded synthetic weights drawn
    directly on `device` (no checkpoints exist offline).  draw_device="cpu" draws every tensor with the CPU generator (one tensor
    at a time, so 8 B parameters never sit in host RAM at once): the values a CPU-side oracle run can reproduce."""
    from . import synthetic
    m = HipLlavaLlamaModel(cfg, device, dtype)
    if weights is not None:
        m.load_weights(weights)
    else:
        with torch.no_grad():
            for mod, prefix, specs in ((m.llm, "llm.", synthetic.llm_specs(cfg)), (m.vision_tower, "vision_tower.", synthetic.vision_specs(cfg)),
                                       (m.mm_projector, "mm_projector.", synthetic.projector_specs(cfg))):
                params = dict(mod.named_parameters())
                for name, shape, kind in specs:
                    params[name[len(prefix):]].copy_(synthetic._draw(name, shape, kind, cfg, seed, draw_device or device))
    return m
