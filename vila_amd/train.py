"""One SFT training step of NVILA on the HIP kernels.

Replaces, for the step itself, what `Trainer.training_step` (patched at
llava/train/transformer_normalize_monkey_patch.py:183-249) + autograd + DeepSpeed ZeRO-3 (`scripts/zero3.json`) do in the
reference:  `LlavaLlamaModel.forward` (llava_llama.py:94-159: `_embed` -> `repack_multimodal_data` -> `llm(..., labels)`) ->
loss = sum CE / GLOBAL num_items (`:261-268`) -> backward through LLM, mm_projector and ViT (all three trainable,
scripts/NVILA-Lite/sft.sh:25-27) -> gradient exchange -> AdamW (adamw_torch, lr 2e-5, wd 0: sft.sh:41-42).

MI355X-first choices:
  * no activation checkpointing: at b=4 x 769 tokens the saved activations are ~16 GB of the 288 GB HBM, so the re-forward
    (25 % extra FLOPs in the reference recipe, sft.sh:47) is simply not done;
  * no ZeRO: every rank keeps the full bf16 params + fp32 master/m/v (8.06 B params -> 16 + 97 GB) and the only exchange is a
    SUM all-reduce of the flat bf16 gradient buffer, bucketed per decoder/encoder layer and launched (async, RCCL's own stream)
    the moment that layer's backward has been enqueued, so it overlaps the backward of the layers below;
  * all parameters live in ONE flat bf16 buffer (module parameters are views, q/k/v fused), gradients in a second flat buffer
    of the same layout: one AdamW launch for the whole model, all-reduce slices are contiguous.
Backward is explicit (no autograd graph): every op is a C-ABI call on the current stream.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .configs import IGNORE_INDEX
from .host import repack, splice_plan
from .modules import _get


# ----------------------------------------------------------------------------------------------------------------------
# flat parameter / gradient storage
# ----------------------------------------------------------------------------------------------------------------------
class FlatParams:
    """Re-points every parameter of (llm, vision_tower, mm_projector) into one flat bf16 buffer; same layout for grads."""

    def __init__(self, model, with_optimizer_state: bool = True):
        self.model = model
        entries: List[Tuple[str, torch.nn.Parameter]] = []
        for prefix, mod in (("llm.", model.llm), ("vision_tower.", model.vision_tower), ("mm_projector.", model.mm_projector)):
            done = set()
            groups = {g[0]: g for g in getattr(mod, "_fused_groups", [])}
            member = {m for g in groups.values() for m in g}
            for n, p in mod.named_parameters():
                if n in done:
                    continue
                if n in groups:                      # keep q,k,v adjacent (fused kernels need one [q+2kv, hidden] buffer)
                    for m in groups[n]:
                        entries.append((prefix + m, _get(mod, m)))
                        done.add(m)
                elif n in member:
                    continue
                else:
                    entries.append((prefix + n, p))
                    done.add(n)
        dev = model.device
        off = 0
        self.index: Dict[str, Tuple[int, int, torch.Size]] = {}
        for n, p in entries:
            self.index[n] = (off, p.numel(), p.shape)
            off += (p.numel() + 7) // 8 * 8          # keep every tensor 16-B aligned
        self.numel = off
        self.params = torch.zeros(off, device=dev, dtype=torch.bfloat16)
        self.grads = torch.zeros(off, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            for n, p in entries:
                o, k, shape = self.index[n]
                view = self.params[o:o + k].view(shape)
                view.copy_(p.data)
                p.data = view
        for mod in (model.llm, model.vision_tower, model.mm_projector):
            mod._invalidate()       # weight structs AND the LLM's captured decode graph point at the old storage
        self.master = self.m = self.v = None
        if with_optimizer_state:
            self.master = self.params.float()
            self.m = torch.zeros_like(self.master)
            self.v = torch.zeros_like(self.master)
        self.step_count = 0
        self.bucket_steps: Dict[str, int] = {}     # AdamW updates seen per gradient bucket (bias correction; see SFTTrainer._adamw_bucket)

    def sync_master_from_params(self) -> None:
        """Call after the bf16 parameters were overwritten behind the trainer's back (checkpoint.load_weights_into / model.load_weights
        on a live trainer): the fp32 master copy is re-seeded from them, otherwise the next AdamW step would write values derived from
        the stale master over the loaded weights."""
        if self.master is not None:
            self.master.copy_(self.params)

    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        names = sorted(self.bucket_steps)
        return {"master": self.master, "exp_avg": self.m, "exp_avg_sq": self.v,
                "step": torch.tensor([self.step_count], dtype=torch.int64), "numel": torch.tensor([self.numel], dtype=torch.int64),
                "bucket_steps": torch.tensor([self.bucket_steps[n] for n in names], dtype=torch.int64),
                "bucket_names": torch.tensor(list("\n".join(names).encode()), dtype=torch.uint8),
                "bucket_step_floor": torch.tensor([int(getattr(self, "bucket_step_floor", 0))], dtype=torch.int64)}

    def load_optimizer_state(self, sd: Dict[str, torch.Tensor]) -> None:
        if int(sd["numel"][0]) != self.numel:
            raise ValueError(f"optimizer state holds {int(sd['numel'][0])} elements, the model has {self.numel}")
        for k in ("master", "exp_avg", "exp_avg_sq"):     # validate everything BEFORE the first copy into the live buffers
            if k not in sd:
                raise KeyError(f"optimizer state has no '{k}'")
            if tuple(sd[k].shape) != (self.numel,):
                raise ValueError(f"optimizer state '{k}' has shape {tuple(sd[k].shape)}, expected ({self.numel},)")
        with torch.no_grad():
            self.master.copy_(sd["master"]); self.m.copy_(sd["exp_avg"]); self.v.copy_(sd["exp_avg_sq"])
            self.params.copy_(self.master)               # bf16 params are the rounding of the master copy
        self.step_count = int(sd["step"][0])
        self.bucket_steps = {}
        self.bucket_step_floor = int(sd["bucket_step_floor"][0]) if "bucket_step_floor" in sd else 0
        if "bucket_names" in sd and sd["bucket_names"].numel():
            names = bytes(sd["bucket_names"].to(torch.uint8).tolist()).decode().split("\n")
            self.bucket_steps = {n: int(c) for n, c in zip(names, sd["bucket_steps"].tolist())}
        else:
            # a state written before per-bucket counts existed: every bucket has seen `step` updates (bias correction must not restart at 1 on
            # warm moments — the first updates after the resume would come out ~0.3x too small)
            self.bucket_step_floor = self.step_count

    def grad(self, name: str) -> torch.Tensor:
        o, k, shape = self.index[name]
        return self.grads[o:o + k].view(shape)

    def param(self, name: str) -> torch.Tensor:
        o, k, shape = self.index[name]
        return self.params[o:o + k].view(shape)

    def span(self, prefix: str) -> Tuple[int, int]:
        """[start, end) of the contiguous slice holding every tensor whose name starts with `prefix`."""
        offs = [(o, o + (k + 7) // 8 * 8) for n, (o, k, _) in self.index.items() if n.startswith(prefix)]
        return min(a for a, _ in offs), max(b for _, b in offs)

    def named_grads(self) -> Dict[str, torch.Tensor]:
        return {n: self.grad(n) for n in self.index}

    def decays(self, name: str) -> bool:
        """Whether AdamW's weight decay applies to this tensor under the reference's parameter groups (`LLaVATrainer.create_optimizer`,
        llava/train/llava_trainer.py:494-495, 537-553: `get_parameter_names(model, ALL_LAYERNORM_LAYERS)` minus every name containing
        "bias").  `ALL_LAYERNORM_LAYERS` is `[nn.LayerNorm]` for this model family — Qwen2's RMSNorm is not in it, so the decoder's norm
        weights DO decay; the LayerNorms are the tower's and the projector's, whose weights are the only 1-D non-bias tensors outside the LLM."""
        if "bias" in name:
            return False
        return not (len(self.index[name][2]) == 1 and not name.startswith("llm."))

    def decay_runs(self, prefix: str) -> List[Tuple[int, int, bool]]:
        """The slice of one gradient bucket cut into maximal runs of adjacent tensors with the same decay flag: [(start, end, decays)]."""
        ent = sorted((o, o + (k + 7) // 8 * 8, self.decays(n)) for n, (o, k, _) in self.index.items() if n.startswith(prefix))
        runs: List[Tuple[int, int, bool]] = []
        for a, b, d in ent:
            if runs and runs[-1][2] == d and runs[-1][1] == a:
                runs[-1] = (runs[-1][0], b, d)
            else:
                runs.append((a, b, d))
        return runs


LOGIT_CHUNK = 256          # rows of fp32 logits alive at a time in the SFT step's lm_head + CE


class _Done:
    """Handle of an exchange that was issued synchronously with respect to the calling stream."""

    def wait(self) -> None:
        return None


class GradReducer:
    """Bucketed SUM exchange of slices of the flat gradient buffer, one bucket per layer, issued as soon as the layer's backward has been
    enqueued (DDP-style overlap; RCCL runs on its own stream, ordered after the compute stream at call time).  With no process group (single
    GPU) it only records the order — which the CPU/gloo test checks.

    Two algorithms (`algo`, environment VILA_GRAD_EXCHANGE), same result up to the summation order of the W contributions:
      "all_reduce" (default): one async `all_reduce(SUM)` per bucket — the algorithm is RCCL's choice (a ring is single-link bound on xGMI:
                   SURVEY 8e computes 184 ms for 16 GB);
      "direct"    : the all-pairs form SURVEY 8e asks for — `all_to_all_single` of the bucket cut into W shards (every GPU sends shard j
                   straight to GPU j: all 7 xGMI links of a GPU carry 1/8 of the bucket each), the owner adds the W pieces in RANK order in
                   fp32 (`vila_grad_accum_f32`: one rounding, and the same bits whatever the arrival order), `all_gather_into_tensor` of the
                   summed shards.  2 x (W-1)/W of the bucket per GPU, like a ring, but over W-1 links at once."""

    def __init__(self, flat: FlatParams, group=None, force: Optional[bool] = None, algo: Optional[str] = None):
        self.flat = flat
        self.group = group
        self.handles = []
        self.log: List[Tuple[str, int, int]] = []
        import os
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        # force: issue the collectives even in a world of one (exercises the RCCL call path, its stream ordering and wait() on a
        # 1-GPU box: bench.py sets it under VILA_BENCH_FORCE_DIST)
        self.force = bool(os.environ.get("VILA_BENCH_FORCE_DIST")) if force is None else force
        self.algo = algo or os.environ.get("VILA_GRAD_EXCHANGE", "all_reduce")
        if self.algo not in ("all_reduce", "direct"):
            raise ValueError(f"VILA_GRAD_EXCHANGE: unknown algorithm {self.algo!r} (all_reduce | direct)")
        self._send = self._recv = self._acc = None          # direct: grow-only staging of one bucket (bf16 send / recv, fp32 sum of a shard)
        self.exchanged_bytes = 0

    def active(self) -> bool:
        return self.dist is not None and (self.dist.get_world_size(self.group) > 1 or self.force)

    def ready(self, prefix: str):
        """Announce that every gradient under `prefix` is final; returns a work handle (None without an exchange).  The
        collective is ordered after the CURRENT stream, so call it inside the stream context that produced / waited for the grads."""
        a, b = self.flat.span(prefix)
        self.log.append((prefix, a, b))
        if not self.active():
            return None
        self.exchanged_bytes += (b - a) * 2
        if self.algo == "direct":
            h = self._direct(self.flat.grads[a:b])
        else:
            h = self.dist.all_reduce(self.flat.grads[a:b], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append(h)
        return h

    def _direct(self, g: torch.Tensor):
        """all-to-all of W shards -> owner sums in rank order (fp32) -> all-gather; in place on `g` (a slice of the flat bf16 buffer)."""
        dist = self.dist
        W = dist.get_world_size(self.group)
        n = g.numel()
        shard = (-(-n // W) + 7) // 8 * 8                     # 16-byte multiples per shard
        P = shard * W
        if self._send is None or self._send.numel() < P:
            self._send = torch.empty((P,), device=g.device, dtype=g.dtype)
            self._recv = torch.empty((P,), device=g.device, dtype=g.dtype)
            self._acc = torch.empty((shard,), device=g.device, dtype=torch.float32)
        recv, acc = self._recv[:P], self._acc[:shard]
        whole = P == n                                        # every decoder / tower layer bucket: the slice itself is the send AND the gather buffer
        if whole:
            send = g
        else:
            send = self._send[:P]
            send[:n].copy_(g)
            send[n:].zero_()
        dist.all_to_all_single(recv, send, group=self.group)                     # recv[k * shard : (k + 1) * shard] = rank k's copy of MY shard
        mine = self._send[:shard]                                                # the summed shard (the staging buffer's head is free in both cases)
        if W == 1:
            mine.copy_(recv[:shard])
        else:
            for k in range(W):                                                   # acc = p0; acc += p1 ...; out = bf16(acc + p_{W-1}): rank order, fp32
                piece = recv[k * shard:(k + 1) * shard]
                ops.grad_accum(acc, piece, mine if k == W - 1 else None, mode=0 if k == 0 else (2 if k == W - 1 else 1))
        if whole:
            dist.all_gather_into_tensor(g, mine, group=self.group)
        else:
            dist.all_gather_into_tensor(recv, mine, group=self.group)
            g.copy_(recv[:n])
        return _Done()

    def wait(self) -> None:
        for h in self.handles:
            h.wait()
        self.handles = []

    def describe(self) -> str:
        w = self.dist.get_world_size(self.group) if self.dist is not None else 1
        be = self.dist.get_backend(self.group) if self.dist is not None else "none"
        nb = len(self.log)
        how = "SUM all-reduce" if self.algo == "all_reduce" else "all-to-all shards + rank-ordered fp32 sum + all-gather (all-pairs)"
        return (f"{nb} buckets (one per layer, reverse order), {how} of flat bf16 grad slices, backend={be}, world={w}"
                + ("" if w > 1 else (" (collectives issued in a world of one)" if self.active() else " (single rank: no exchange issued)")))


# ----------------------------------------------------------------------------------------------------------------------
# linear layer helpers: y = x W^T (+b);  dx = dy W ; dW = dy^T x ; db = colsum(dy)
# ----------------------------------------------------------------------------------------------------------------------
def _cm_ok(T: int, N: int, K: int) -> bool:
    """Shapes the 256x256 contraction-major kernels take (gemm256_supported): rows % 8 == 0, contraction >= 128."""
    return T >= 128 and N >= 128 and N % 8 == 0 and K % 8 == 0


def linear_bwd(x2: torch.Tensor, w: torch.Tensor, dy2: torch.Tensor, gw: torch.Tensor, gb: Optional[torch.Tensor] = None,
               need_dx: bool = True, dx_residual: Optional[torch.Tensor] = None, dy_t: Optional[torch.Tensor] = None,
               cm: bool = False, side: Optional["torch.cuda.Stream"] = None, ws: Optional[torch.Tensor] = None,
               ws_side: Optional[torch.Tensor] = None):
    """x2 [M,K], w [N,K], dy2 [M,N]  ->  writes gw [N,K] (and gb [N]); returns dx [M,K] (+ dx_residual).
    cm: read W, dY and X as they lie (vila_gemm_bf16_t: contraction-major operands through the LDS transpose reads) instead of making
    transposed copies.  side: stream for the weight-gradient GEMM — dgrad and wgrad only share inputs, so the two run concurrently
    and the partial last round of one kernel's 256x256 tiles is filled by the other's."""
    w2 = w.view(w.shape[0], -1)
    gw2 = gw.view(w.shape[0], -1) if gw.dim() != 2 else gw
    M, K = x2.shape
    N = w2.shape[0]
    if cm and _cm_ok(M, N, K):
        def wgrad():
            ops.gemm_t(dy2, x2, a_cm=True, b_cm=True, out=gw2, ws=ws_side if (side is not None and ws_side is not None) else ws)   # dW = dY^T X
            if gb is not None:
                ops.colsum(dy2, gb)
        if side is not None:
            side.wait_event(torch.cuda.current_stream().record_event())             # dY and X are final on the compute stream
            with torch.cuda.stream(side):
                wgrad()
            dy2.record_stream(side); x2.record_stream(side)
        else:
            wgrad()
        if not need_dx:
            return None
        # split-K slabs are per stream: with a side stream the dgrad may slice K only when the wgrad has a workspace of its own
        return ops.gemm_t(dy2, w2, b_cm=True, residual=dx_residual, ws=ws if (side is None or ws_side is not None) else None)   # dX = dY W (+ residual)
    dyt = dy_t if dy_t is not None else ops.transpose(dy2)           # [N, Mp]
    xt = ops.transpose(x2)                                           # [K, Mp]
    ops.gemm(dyt, xt, out=gw2)                                       # dW = dY^T X
    if gb is not None:
        ops.colsum(dy2, gb)
    if not need_dx:
        return None
    wt = ops.transpose(w2)                                           # [K, N]
    return ops.gemm(dy2, wt, residual=dx_residual)                   # dX = dY W (+ residual)


# ----------------------------------------------------------------------------------------------------------------------
# the trainer
# ----------------------------------------------------------------------------------------------------------------------
class SFTTrainer:
    def __init__(self, model, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: Optional[float] = None, optimizer_state: bool = True, group=None, decay_groups: bool = True):
        self.model = model
        self.cfg = model.cfg
        self.flat = FlatParams(model, with_optimizer_state=optimizer_state)
        self.reducer = GradReducer(self.flat, group)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.group = group
        self.decay_groups = decay_groups   # weight decay skips biases and LayerNorm weights like the reference's optimizer groups (False: decays all)
        self._decay_runs: Dict[str, List[Tuple[int, int, bool]]] = {}
        dev = model.device
        on_gpu = torch.device(dev).type == "cuda"
        # side: weight-gradient GEMMs of the decoder layers (concurrent with the dgrad chain on the compute stream)
        # opt : per-bucket gradient exchange + AdamW, as soon as a layer's gradients are final — the optimizer streams 28 B per
        #       parameter at HBM rate and the exchange runs on xGMI while the matrix cores work on the layers below
        import os
        flag = lambda name, default: os.environ.get(name, default) not in ("0", "", "false")
        self.side = torch.cuda.Stream(device=dev) if (on_gpu and flag("VILA_SFT_SIDE", "1")) else None
        self.opt = torch.cuda.Stream(device=dev) if (on_gpu and flag("VILA_SFT_OPT_STREAM", "1")) else None
        # dK / dV on its own stream beside dQ: measured 211.3 / 209.7 -> 210.3 / 209.7 ms, i.e. nothing (the chip already runs ~2 kernels at a
        # time and the step is throughput-bound): off by default, kept as a switch
        self.attn_s = torch.cuda.Stream(device=dev) if (on_gpu and flag("VILA_SFT_ATTN_STREAM", "0")) else None
        self.use_c_abi = flag("VILA_SFT_C_ABI", "0")            # step(): forward+backward as ONE vila_sft_fwd_bwd call (else Python-orchestrated ops)
        self.lean_adamw = flag("VILA_SFT_LEAN_ADAMW", "1")      # <= 32-VGPR optimizer kernel: co-resident with the GEMM blocks
        self.cm = flag("VILA_SFT_CM", "1")  # dgrad / wgrad on the tensors as they lie (no transposed copies) where the shapes allow
        # fp32 slabs for the K-sliced launches (lm_head dgrad, tower shapes, tail tiles behind whole rounds)
        self.ws = torch.empty(128 << 20, device=dev, dtype=torch.uint8) if on_gpu else None
        # the tower / projector GEMMs (M = 1024 per image, N = 1152): contraction-major too, K-sliced where their 256^2 tiles under-fill
        self.cm_vit = self.cm and flag("VILA_SFT_CM_VIT", "1")
        self.ws_side = torch.empty(128 << 20, device=dev, dtype=torch.uint8) if (on_gpu and self.side is not None) else None   # the side stream's own slabs
        self._bucket_step = False          # set per step: apply AdamW bucket by bucket (no global clipping)
        self._touched: List[str] = []      # bucket prefixes whose gradients this step produced, in the order they became final
        # gradient accumulation (`step_accumulated`): the sum of the earlier micro-batches' gradients and what `_ready` does with a bucket —
        # None: a plain step; "hold": not the last micro-batch (no exchange, no update); "add": the last one (bucket += held sum, then as usual)
        self._acc: Optional[torch.Tensor] = None
        self._acc_mode: Optional[str] = None

    def _attn_bwd(self, *a, **kw) -> None:
        """Flash-attention backward.  dQ and dK / dV share no output and both under-fill the chip (208 blocks at 4 x 769 tokens, 4-13 % of the
        MFMA rate), so once delta is done dK / dV runs on its own stream beside dQ and the two meet again in front of the QKV dgrad."""
        if self.attn_s is None:
            ops.attn_bwd(*a, **kw)
            return
        main = torch.cuda.current_stream()
        delta = ops.attn_bwd(*a, parts=1, **kw)
        self.attn_s.wait_event(main.record_event())                     # delta, dO and the saved q / k / v are final
        with torch.cuda.stream(self.attn_s):
            ops.attn_bwd(*a, parts=4, delta=delta, **kw)
            done = self.attn_s.record_event()
        ops.attn_bwd(*a, parts=2, delta=delta, **kw)
        main.wait_event(done)                                            # (every later use or free of these tensors is ordered behind this)

    def _ready(self, prefix: str) -> None:
        """Gradients under `prefix` are final once the compute stream and the wgrad stream reach this point: hand the bucket to the
        optimizer stream (exchange, then AdamW on that slice) and carry on with the layers below."""
        if self._acc_mode == "hold":                 # an earlier micro-batch of an accumulated update: its gradients only join the held sum
            self._touched.append(prefix)
            return
        if self.opt is None:
            # no optimizer stream: the exchange is ordered behind the compute stream, so the wgrad stream (the weight-gradient GEMMs and
            # bias column sums of this bucket are still in flight there) has to be joined first, or the all-reduce would read / overwrite
            # gradient slices that are being written
            if self.side is not None:
                torch.cuda.current_stream().wait_event(self.side.record_event())
            self._touched.append(prefix)
            self._add_held(prefix)
            self.reducer.ready(prefix)
            return
        main = torch.cuda.current_stream()
        self.opt.wait_event(main.record_event())
        if self.side is not None:
            self.opt.wait_event(self.side.record_event())
        self._touched.append(prefix)
        with torch.cuda.stream(self.opt):
            self._add_held(prefix)
            h = self.reducer.ready(prefix)
            if self._bucket_step:
                if h is not None:
                    h.wait()                                        # the optimizer stream waits for this bucket's all-reduce only
                self._adamw_bucket(prefix, 1.0)

    def _add_held(self, prefix: str) -> None:
        """Last micro-batch of an accumulated update: this bucket's gradient becomes (sum of the earlier micro-batches) + (this one), on the
        stream the exchange and the update of the bucket are ordered on."""
        if self._acc_mode == "add":
            a, b = self.flat.span(prefix)
            ops.grad_accum(self._acc[a:b], self.flat.grads[a:b], out=self.flat.grads[a:b], mode=2)    # bf16(fp32 sum + this one): ONE rounding

    def _adamw_bucket(self, prefix: str, grad_scale: float) -> None:
        """AdamW on the slice of one gradient bucket.  Semantics = torch.optim.AdamW with grad = None for what a step did not touch:
        a bucket that received no gradient this step (text-only batch: tower + projector; the unused 27th ViT layer and post_layernorm:
        never) is NOT updated — no weight decay, no moment decay — and its bias correction uses the number of updates IT has seen."""
        f = self.flat
        a, b = f.span(prefix)
        step = f.bucket_steps.get(prefix, getattr(f, "bucket_step_floor", 0)) + 1
        f.bucket_steps[prefix] = step
        if self.wd == 0.0 or not self.decay_groups:          # every NVILA script trains with --weight_decay 0.: one launch per bucket
            ops.adamw_step(f.master[a:b], f.m[a:b], f.v[a:b], f.grads[a:b], f.params[a:b], self.lr, self.betas[0], self.betas[1],
                           self.eps, self.wd, step, grad_scale, lean=self.lean_adamw)
            return
        # weight decay > 0: the reference's two parameter groups (biases and LayerNorm weights do not decay) = one launch per run of
        # adjacent tensors with the same flag
        if prefix not in self._decay_runs:
            self._decay_runs[prefix] = f.decay_runs(prefix)
        for ra, rb, dec in self._decay_runs[prefix]:
            ops.adamw_step(f.master[ra:rb], f.m[ra:rb], f.v[ra:rb], f.grads[ra:rb], f.params[ra:rb], self.lr, self.betas[0], self.betas[1],
                           self.eps, self.wd if dec else 0.0, step, grad_scale, lean=self.lean_adamw)

    # ------------------------------------------------------------------ ViT ------------------------------------------------
    def _vit_fwd(self, pixels: torch.Tensor):
        v = self.cfg.vision
        P = self.flat.param
        pre = "vision_tower.vision_tower.vision_model."
        B = pixels.shape[0]
        N, D, hd = v.num_patches, v.hidden_size, v.head_dim
        Kc = v.num_channels * v.patch_size * v.patch_size
        Kp = (Kc + 7) // 8 * 8
        patches = ops.im2col(pixels, v.patch_size, Kp)                                   # [B*N, Kp]
        wpad = torch.zeros((D, Kp), device=pixels.device, dtype=torch.bfloat16)
        wpad[:, :Kc] = P(pre + "embeddings.patch_embedding.weight").view(D, Kc)
        pos = P(pre + "embeddings.position_embedding.weight")
        x = torch.empty((B * N, D), device=pixels.device, dtype=torch.bfloat16)
        for b in range(B):
            ops.gemm(patches[b * N:(b + 1) * N], wpad, bias=P(pre + "embeddings.patch_embedding.bias"), residual=pos, out=x[b * N:(b + 1) * N])
        saved = SimpleNamespace(patches=patches, layers=[], B=B)
        for i in range(v.num_used_layers):
            l = f"{pre}encoder.layers.{i}."
            s = SimpleNamespace(x_in=x)
            s.h1 = ops.layernorm(x, P(l + "layer_norm1.weight"), P(l + "layer_norm1.bias"), v.layer_norm_eps)
            wqkv = self._fused(l + "self_attn.", "weight")
            bqkv = self._fused(l + "self_attn.", "bias")
            s.qkv = ops.gemm(s.h1, wqkv, bias=bqkv)
            q3 = s.qkv.view(B * N, 3 * v.num_attention_heads, hd)
            H = v.num_attention_heads
            s.a, s.lse = ops.attn_fwd(q3[:, :H], q3[:, H:2 * H], q3[:, 2 * H:], False, n_seq=B, return_lse=True)
            s.x_mid = ops.gemm(s.a.view(B * N, D), P(l + "self_attn.out_proj.weight"), bias=P(l + "self_attn.out_proj.bias"), residual=x)
            s.h2 = ops.layernorm(s.x_mid, P(l + "layer_norm2.weight"), P(l + "layer_norm2.bias"), v.layer_norm_eps)
            s.z1 = ops.gemm(s.h2, P(l + "mlp.fc1.weight"), bias=P(l + "mlp.fc1.bias"))
            s.f = ops.act_fwd(s.z1, 1)
            x = ops.gemm(s.f, P(l + "mlp.fc2.weight"), bias=P(l + "mlp.fc2.bias"), residual=s.x_mid)
            saved.layers.append(s)
        return x.view(B, N, D), saved

    def _fused(self, prefix: str, kind: str) -> torch.Tensor:
        """the fused [q+2kv, ...] view of q_proj/k_proj/v_proj (adjacent in the flat buffer)."""
        o, k, shape = self.flat.index[prefix + "q_proj." + kind]
        o2, k2, shape2 = self.flat.index[prefix + "v_proj." + kind]
        rows = (o2 + k2 - o) // (shape[1] if len(shape) == 2 else 1)
        t = self.flat.params[o:o2 + k2]
        return t.view(rows, shape[1]) if len(shape) == 2 else t

    def _fused_grad(self, prefix: str, kind: str) -> torch.Tensor:
        o, k, shape = self.flat.index[prefix + "q_proj." + kind]
        o2, k2, _ = self.flat.index[prefix + "v_proj." + kind]
        t = self.flat.grads[o:o2 + k2]
        return t.view(-1, shape[1]) if len(shape) == 2 else t

    def _vit_bwd(self, dx: torch.Tensor, saved) -> None:
        v = self.cfg.vision
        P, G = self.flat.param, self.flat.grad
        pre = "vision_tower.vision_tower.vision_model."
        B, N, D, hd, H = saved.B, v.num_patches, v.hidden_size, v.head_dim, v.num_attention_heads
        kw = dict(cm=self.cm_vit, side=self.side, ws=self.ws, ws_side=self.ws_side) if self.cm_vit else {}
        for i in reversed(range(v.num_used_layers)):
            l = f"{pre}encoder.layers.{i}."
            s = saved.layers[i]
            df = linear_bwd(s.f, P(l + "mlp.fc2.weight"), dx, G(l + "mlp.fc2.weight"), G(l + "mlp.fc2.bias"), **kw)
            dz1 = ops.act_bwd(s.z1, df, 1)
            dh2 = linear_bwd(s.h2, P(l + "mlp.fc1.weight"), dz1, G(l + "mlp.fc1.weight"), G(l + "mlp.fc1.bias"), **kw)
            dxm = ops.norm_bwd(s.x_mid, P(l + "layer_norm2.weight"), dh2, G(l + "layer_norm2.weight"), G(l + "layer_norm2.bias"), v.layer_norm_eps, False)
            dx_mid = ops.add(dx, dxm)
            da = linear_bwd(s.a.view(B * N, D), P(l + "self_attn.out_proj.weight"), dx_mid, G(l + "self_attn.out_proj.weight"), G(l + "self_attn.out_proj.bias"), **kw)
            dqkv = torch.empty_like(s.qkv)
            q3, d3 = s.qkv.view(B * N, 3 * H, hd), dqkv.view(B * N, 3 * H, hd)
            self._attn_bwd(q3[:, :H], q3[:, H:2 * H], q3[:, 2 * H:], s.a, da.view(B * N, H, hd), s.lse, False,
                           d3[:, :H], d3[:, H:2 * H], d3[:, 2 * H:], n_seq=B)
            dh1 = linear_bwd(s.h1, self._fused(l + "self_attn.", "weight"), dqkv, self._fused_grad(l + "self_attn.", "weight"), self._fused_grad(l + "self_attn.", "bias"), **kw)
            dxi = ops.norm_bwd(s.x_in, P(l + "layer_norm1.weight"), dh1, G(l + "layer_norm1.weight"), G(l + "layer_norm1.bias"), v.layer_norm_eps, False)
            dx = ops.add(dx_mid, dxi)
            self._ready(l)
        # patch embedding: weight (unpadded), bias, position embedding (sum over images)
        Kc = v.num_channels * v.patch_size * v.patch_size
        gw = torch.empty((D, saved.patches.shape[1]), device=dx.device, dtype=torch.bfloat16)
        ops.gemm(ops.transpose(dx), ops.transpose(saved.patches), out=gw)
        G(pre + "embeddings.patch_embedding.weight").view(D, Kc).copy_(gw[:, :Kc])
        ops.colsum(dx, G(pre + "embeddings.patch_embedding.bias"))
        ops.colsum(dx, G(pre + "embeddings.position_embedding.weight"), period=N)
        # unused parameters (27th layer, post_layernorm) keep zero gradients (hidden_states[-2]: vision_encoder.py:44-52)
        self._ready(pre + "embeddings.")

    # ------------------------------------------------------------------ projector ----------------------------------------
    def _proj_fwd(self, feats: torch.Tensor):
        P = self.flat.param
        t = self.cfg.mm_projector_type
        k = self.cfg.downsample
        pre = "mm_projector.layers."
        s = SimpleNamespace(g=int(round(feats.shape[1] ** 0.5)), k=k)
        s.y = ops.space_to_depth(feats, k)
        B, T, C1 = s.y.shape
        y2 = s.y.view(B * T, C1)
        s.yn = ops.layernorm(y2, P(pre + "1.weight"), P(pre + "1.bias"), 1e-5)
        s.z1 = ops.gemm(s.yn, P(pre + "2.weight"), bias=P(pre + "2.bias"))
        s.h1 = ops.act_fwd(s.z1, 2)
        if t == "mlp_downsample_3x3_fix":
            s.h1n = ops.layernorm(s.h1, P(pre + "4.weight"), P(pre + "4.bias"), 1e-5)
            s.z2 = ops.gemm(s.h1n, P(pre + "5.weight"), bias=P(pre + "5.bias"))
            s.h2 = ops.act_fwd(s.z2, 2)
            out = ops.gemm(s.h2, P(pre + "7.weight"), bias=P(pre + "7.bias"))
        else:
            out = ops.gemm(s.h1, P(pre + "4.weight"), bias=P(pre + "4.bias"))
        return out.view(B, T, -1), s

    def _proj_bwd(self, dout: torch.Tensor, s) -> torch.Tensor:
        P, G = self.flat.param, self.flat.grad
        pre = "mm_projector.layers."
        B, T, _ = s.y.shape
        d = dout.reshape(B * T, -1)
        kw = dict(cm=self.cm_vit, side=self.side, ws=self.ws, ws_side=self.ws_side) if self.cm_vit else {}
        if self.cfg.mm_projector_type == "mlp_downsample_3x3_fix":
            dh2 = linear_bwd(s.h2, P(pre + "7.weight"), d, G(pre + "7.weight"), G(pre + "7.bias"), **kw)
            dz2 = ops.act_bwd(s.z2, dh2, 2)
            dh1n = linear_bwd(s.h1n, P(pre + "5.weight"), dz2, G(pre + "5.weight"), G(pre + "5.bias"), **kw)
            dh1 = ops.norm_bwd(s.h1, P(pre + "4.weight"), dh1n, G(pre + "4.weight"), G(pre + "4.bias"), 1e-5, False)
        else:
            dh1 = linear_bwd(s.h1, P(pre + "4.weight"), d, G(pre + "4.weight"), G(pre + "4.bias"), **kw)
        dz1 = ops.act_bwd(s.z1, dh1, 2)
        dyn = linear_bwd(s.yn, P(pre + "2.weight"), dz1, G(pre + "2.weight"), G(pre + "2.bias"), **kw)
        dy = ops.norm_bwd(s.y.view(B * T, -1), P(pre + "1.weight"), dyn, G(pre + "1.weight"), G(pre + "1.bias"), 1e-5, False)
        self._ready("mm_projector.")
        return ops.depth_to_space(dy.view(B, T, -1), s.g, s.k)

    # ------------------------------------------------------------------ LLM ----------------------------------------------
    def _llm_fwd(self, x: torch.Tensor, pos: torch.Tensor, cu: torch.Tensor, max_seqlen: int):
        c = self.cfg.llm
        P = self.flat.param
        T = x.shape[0]
        nq, nkv, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        cs, sn = ops.rope_table(pos, hd, c.rope_theta)
        saved = SimpleNamespace(layers=[], cs=cs, sn=sn, cu=cu, max_seqlen=max_seqlen)
        for i in range(c.num_hidden_layers):
            l = f"llm.model.layers.{i}."
            s = SimpleNamespace(x_in=x)
            s.h1 = ops.rmsnorm(x, P(l + "input_layernorm.weight"), c.rms_norm_eps)
            s.qkv = ops.gemm(s.h1, self._fused(l + "self_attn.", "weight"), bias=self._fused(l + "self_attn.", "bias"))
            ops.rope_fwd_(s.qkv, cs, sn, pos, nq, nkv, hd)
            q3 = s.qkv.view(T, nq + 2 * nkv, hd)
            s.a, s.lse = ops.attn_fwd(q3[:, :nq], q3[:, nq:nq + nkv], q3[:, nq + nkv:], True, cu_seqlens=cu, max_seqlen=max_seqlen, return_lse=True)
            s.x_mid = ops.gemm(s.a.view(T, nq * hd), P(l + "self_attn.o_proj.weight"), residual=x)
            s.h2 = ops.rmsnorm(s.x_mid, P(l + "post_attention_layernorm.weight"), c.rms_norm_eps)
            s.g = ops.gemm(s.h2, P(l + "mlp.gate_proj.weight"))
            s.u = ops.gemm(s.h2, P(l + "mlp.up_proj.weight"))
            s.act = ops.silu_mul(s.g, s.u)
            x = ops.gemm(s.act, P(l + "mlp.down_proj.weight"), residual=s.x_mid, ws=self.ws)
            saved.layers.append(s)
        saved.x_out = x
        saved.hn = ops.rmsnorm(x, P("llm.model.norm.weight"), c.rms_norm_eps)
        return saved

    def _llm_bwd(self, dx: torch.Tensor, saved) -> torch.Tensor:
        c = self.cfg.llm
        P, G = self.flat.param, self.flat.grad
        T = dx.shape[0]
        nq, nkv, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        for i in reversed(range(c.num_hidden_layers)):
            l = f"llm.model.layers.{i}."
            s = saved.layers[i]
            kw = dict(cm=self.cm, side=self.side, ws=self.ws, ws_side=self.ws_side)
            dact = linear_bwd(s.act, P(l + "mlp.down_proj.weight"), dx, G(l + "mlp.down_proj.weight"), **kw)
            dg, du = ops.silu_mul_bwd(s.g, s.u, dact)
            dh2 = linear_bwd(s.h2, P(l + "mlp.gate_proj.weight"), dg, G(l + "mlp.gate_proj.weight"), **kw)
            dh2 = linear_bwd(s.h2, P(l + "mlp.up_proj.weight"), du, G(l + "mlp.up_proj.weight"), dx_residual=dh2, **kw)
            dxm = ops.norm_bwd(s.x_mid, P(l + "post_attention_layernorm.weight"), dh2, G(l + "post_attention_layernorm.weight"), None, c.rms_norm_eps, True)
            dx_mid = ops.add(dx, dxm)
            da = linear_bwd(s.a.view(T, nq * hd), P(l + "self_attn.o_proj.weight"), dx_mid, G(l + "self_attn.o_proj.weight"), **kw)
            dqkv = torch.empty_like(s.qkv)
            q3, d3 = s.qkv.view(T, nq + 2 * nkv, hd), dqkv.view(T, nq + 2 * nkv, hd)
            self._attn_bwd(q3[:, :nq], q3[:, nq:nq + nkv], q3[:, nq + nkv:], s.a, da.view(T, nq, hd), s.lse, True,
                           d3[:, :nq], d3[:, nq:nq + nkv], d3[:, nq + nkv:], cu_seqlens=saved.cu, max_seqlen=saved.max_seqlen)
            ops.rope_bwd_(dqkv, saved.cs, saved.sn, nq, nkv, hd)
            dh1 = linear_bwd(s.h1, self._fused(l + "self_attn.", "weight"), dqkv, self._fused_grad(l + "self_attn.", "weight"), self._fused_grad(l + "self_attn.", "bias"), **kw)
            dxi = ops.norm_bwd(s.x_in, P(l + "input_layernorm.weight"), dh1, G(l + "input_layernorm.weight"), None, c.rms_norm_eps, True)
            dx = ops.add(dx_mid, dxi)
            self._ready(l)
        return dx

    # ------------------------------------------------------------------ media plan ----------------------------------------
    def _media_plan(self, n_px: int, block_sizes):
        """Integer plan of the image side of one batch.  Plain recipe: every image is one tile, its tokens are the projector's rows
        [i*Tm, (i+1)*Tm).  dynamic_s2 (llava_arch.py:369-390): `n_px` tiles of all scales of all images, one entry of `block_sizes` per
        IMAGE; the projector runs on the merged blocks and image i's tokens are the chessboard re-merge of its blocks = `perms[i]`.
        -> (s2 plan or None, rows: per image the projector-output row of each of its tokens, n_pin = projector inputs)"""
        cfg = self.cfg
        Tm = cfg.tokens_per_tile
        if not getattr(cfg, "dynamic_s2", False):
            return None, [torch.arange(i * Tm, (i + 1) * Tm, dtype=torch.int64) for i in range(n_px)], n_px
        from .host import s2_plan
        if block_sizes is None:
            raise ValueError("dynamic_s2 training needs media_config['image']['block_sizes'] (one (h, w) or None per image)")
        plan = s2_plan(list(block_sizes), list(cfg.s2_scales), cfg.vision.grid, cfg.downsample, cfg.s2_resize_output_to_scale_idx)
        if plan.n_tiles != n_px:
            raise AssertionError(f"The number of blocks ({plan.n_tiles}) does not match length of image_features ({n_px})!")
        return plan, [p.long() for p in plan.perms], plan.n_blocks

    def _with_videos(self, images, videos):
        """`videos` (list of [n_frames, 3, H, W], each standing for one <vila/video> token): the frames join the tower batch BEHIND the images
        (video/basic.py:43-53, tsp.py:54-64 run encode_images on all frames at once).  -> (tiles in tower order, frames per video)."""
        videos = list(videos or [])
        if not videos:
            return list(images), []
        return list(images) + [f for v in videos for f in v], [int(v.shape[0]) for v in videos]

    def _block_sizes_with_frames(self, block_sizes, frames):
        """dynamic_s2: the video encoders call encode_images WITHOUT block sizes (video/basic.py:48, tsp.py:59), i.e. every frame is a one-tile
        image with block size None (llava_arch.py:367-368, 309-314: its features are repeated over the scales)."""
        if not frames or not getattr(self.cfg, "dynamic_s2", False):
            return block_sizes
        return list(block_sizes or []) + [None] * sum(frames)

    def _video_tokens(self):
        """(pool_sizes, start ids, end ids, separator ids) of the model's video encoder (video/basic.py:13-28, tsp.py:14-26); a model without
        one frames every video frame like an image: no start tokens, the "\n" end token."""
        enc = getattr(self.model, "encoders", {}).get("video")
        if enc is None:
            return ((1, 1, 1),), [], [self.cfg.newline_token_id], []
        tok = lambda t: [] if t is None else [int(x) for x in self.model.tokenizer(t).input_ids]
        pools = tuple(tuple(int(x) for x in p) for p in getattr(enc, "pool_sizes", ((1, 1, 1),)))
        return pools, tok(enc.start_tokens), tok(enc.end_tokens), tok(getattr(enc, "sep_tokens", None))

    def _media_blocks(self, rows: List[torch.Tensor], frames: List[int], n_prow: int):
        """The token block of every media placeholder as a row table over the MEDIA FEATURE BUFFER = [projector rows | pooled rows of every
        (video, pool size)]: a value >= 0 is a row of that buffer, a value -1 - id is the embedding of token `id` (the "\n" end token of an
        image, the start / end / separator tokens of the video encoder).
          image i            rows[i] + ["\n"]                                             (encoders/image/basic.py:40-53)
          video, pool 1,1,1  per frame  [start | rows[frame] | end], then the separator    (video/basic.py:30-41)
          video, pooled      per pooled frame [start | its pooled rows | end], separator   (video/tsp.py:28-52), all pool sizes back to back
        -> (image blocks, video blocks, pools = [(first projector block, n_frames, pool, buffer row offset, n pooled rows)], buffer rows)"""
        cfg = self.cfg
        tokrow = lambda ids: torch.tensor([-1 - int(t) for t in ids], dtype=torch.int64)
        n_img = len(rows) - sum(frames)
        nl_row = tokrow([cfg.newline_token_id])
        img_blocks = [torch.cat([r, nl_row]) for r in rows[:n_img]]
        vid_blocks, pools, n_buf = [], [], n_prow
        if frames:
            pool_sizes, start, end, sep = self._video_tokens()
            start, end, sep = tokrow(start), tokrow(end), tokrow(sep)
            Tm = cfg.tokens_per_tile
            nl = int(round(Tm ** 0.5))
            t0 = n_img
            for nf in frames:
                parts = []
                for pool in pool_sizes:
                    if pool == (1, 1, 1):
                        for t in range(t0, t0 + nf):
                            parts += [start, rows[t], end]
                    else:
                        pt, ph, pw = pool
                        if nl * nl != Tm or pt <= 0 or ph <= 0 or pw <= 0 or nf % pt or nl % ph or nl % pw:   # the reference's view() raises
                            raise ValueError(f"shape '[{nf}, {nl}, {nl}]' is invalid for pooling by ({pt}, {ph}, {pw}): every pooled dimension must divide evenly")
                        n_feat = (nl // ph) * (nl // pw)
                        b0 = int(rows[t0][0]) // Tm                 # the frames' projector blocks (dynamic_s2: images before them own several)
                        for i in range(nf):
                            assert torch.equal(rows[t0 + i], torch.arange((b0 + i) * Tm, (b0 + i + 1) * Tm, dtype=torch.int64))
                        for f in range(nf // pt):
                            parts += [start, n_buf + f * n_feat + torch.arange(n_feat, dtype=torch.int64), end]
                        pools.append((b0, nf, pool, n_buf, (nf // pt) * n_feat))
                        n_buf += (nf // pt) * n_feat
                    parts.append(sep)
                vid_blocks.append(torch.cat(parts) if parts else torch.empty((0,), dtype=torch.int64))
                t0 += nf
        return img_blocks, vid_blocks, pools, n_buf

    def _splice(self, input_ids, attention_mask, labels, img_blocks, vid_blocks):
        """splice_plan for the step: one block per media placeholder.  The flat media space (images, then videos) is the row space of
        `_media_rows`."""
        cfg, model = self.cfg, self.model
        # both media kinds ALWAYS take part: a `<vila/video>` token with no video supplied must raise like the reference's empty deque
        # (llava_arch.py:462-466), not be embedded as a text token
        lens = {"image": [int(b.numel()) for b in img_blocks], "video": [int(b.numel()) for b in (vid_blocks or [])]}
        toks = {"image": cfg.image_token_id, "video": cfg.video_token_id}
        return splice_plan(input_ids, attention_mask, labels, lens, toks, "right",
                           max_length=getattr(getattr(model, "tokenizer", None), "model_max_length", None))

    @staticmethod
    def _media_rows(plan_img_src: torch.Tensor, blocks: List[torch.Tensor]):
        """Map the splice plan's media-row indices (into the concatenation of the blocks of `_media_blocks`) to that table's values:
        -> (media feature buffer row of every spliced media row, or -1 - token id for a token row)."""
        if not blocks:
            return torch.empty((0,), dtype=torch.int64)
        return torch.cat(list(blocks)).to(plan_img_src.device)[plan_img_src.long()]

    # ------------------------------------------------------------------ one C-ABI call ------------------------------------
    def _c_structs(self, ptr):
        """(VilaVitWeights, VilaProjWeights, VilaLlmWeights) whose pointers come from `ptr(name)`: flat.param for the weights, flat.grad for
        the mirror structs vila_sft_fwd_bwd writes the gradients through (same names, same shapes: SURVEY Appendix C)."""
        import ctypes as C
        from . import _lib
        cfg, v, c = self.cfg, self.cfg.vision, self.cfg.llm
        keep = []
        pre = "vision_tower.vision_tower.vision_model."
        n_run = v.num_used_layers
        vl = (_lib.VilaVitLayer * max(n_run, 1))()
        for i in range(n_run):
            l, L = f"{pre}encoder.layers.{i}.", vl[i]
            L.ln1_w, L.ln1_b = ptr(l + "layer_norm1.weight"), ptr(l + "layer_norm1.bias")
            for k in ("q", "k", "v"):
                setattr(L, "w" + k, ptr(l + f"self_attn.{k}_proj.weight")); setattr(L, "b" + k, ptr(l + f"self_attn.{k}_proj.bias"))
            L.wo, L.bo = ptr(l + "self_attn.out_proj.weight"), ptr(l + "self_attn.out_proj.bias")
            L.ln2_w, L.ln2_b = ptr(l + "layer_norm2.weight"), ptr(l + "layer_norm2.bias")
            L.fc1_w, L.fc1_b, L.fc2_w, L.fc2_b = ptr(l + "mlp.fc1.weight"), ptr(l + "mlp.fc1.bias"), ptr(l + "mlp.fc2.weight"), ptr(l + "mlp.fc2.bias")
        vw = _lib.VilaVitWeights()
        vw.shape = _lib.VilaVitShape(v.hidden_size, v.intermediate_size, v.num_attention_heads, v.image_size, v.patch_size, v.num_channels, n_run, v.layer_norm_eps)
        vw.patch_w, vw.patch_b, vw.pos_emb = ptr(pre + "embeddings.patch_embedding.weight"), ptr(pre + "embeddings.patch_embedding.bias"), ptr(pre + "embeddings.position_embedding.weight")
        vw.layers = C.cast(vl, C.POINTER(_lib.VilaVitLayer))
        pw = _lib.VilaProjWeights()
        pw.kind = {"mlp_downsample": 0, "mlp_downsample_2x2_fix": 1, "mlp_downsample_3x3_fix": 2}[cfg.mm_projector_type]
        pw.in_dim, pw.out_dim = cfg.mm_hidden_size, c.hidden_size
        g = lambda i, n: ptr(f"mm_projector.layers.{i}.{n}")
        pw.ln1_w, pw.ln1_b, pw.fc1_w, pw.fc1_b = g(1, "weight"), g(1, "bias"), g(2, "weight"), g(2, "bias")
        if pw.kind == 2:
            pw.ln2_w, pw.ln2_b, pw.fc2_w, pw.fc2_b, pw.fc3_w, pw.fc3_b = g(4, "weight"), g(4, "bias"), g(5, "weight"), g(5, "bias"), g(7, "weight"), g(7, "bias")
        else:
            pw.fc2_w, pw.fc2_b = g(4, "weight"), g(4, "bias")
        ll = (_lib.VilaLlmLayer * c.num_hidden_layers)()
        for i in range(c.num_hidden_layers):
            l, L = f"llm.model.layers.{i}.", ll[i]
            L.ln1_w, L.ln2_w = ptr(l + "input_layernorm.weight"), ptr(l + "post_attention_layernorm.weight")
            for k in ("q", "k", "v"):
                setattr(L, "w" + k, ptr(l + f"self_attn.{k}_proj.weight")); setattr(L, "b" + k, ptr(l + f"self_attn.{k}_proj.bias"))
            L.wo = ptr(l + "self_attn.o_proj.weight")
            L.w_gate, L.w_up, L.w_down = ptr(l + "mlp.gate_proj.weight"), ptr(l + "mlp.up_proj.weight"), ptr(l + "mlp.down_proj.weight")
        lw = _lib.VilaLlmWeights()
        lw.shape = _lib.VilaLlmShape(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, c.head_dim,
                                     c.vocab_size, c.rms_norm_eps, c.rope_theta)
        lw.embed, lw.norm_w = ptr("llm.model.embed_tokens.weight"), ptr("llm.model.norm.weight")
        lw.lm_head = lw.embed if c.tie_word_embeddings else ptr("llm.lm_head.weight")
        lw.layers = C.cast(ll, C.POINTER(_lib.VilaLlmLayer))
        keep += [vl, ll]
        return vw, pw, lw, keep

    def forward_backward_c(self, input_ids: torch.Tensor, images: List[torch.Tensor], labels: torch.Tensor,
                           attention_mask: Optional[torch.Tensor] = None, num_items_in_batch: Optional[int] = None,
                           block_sizes=None, videos=None) -> torch.Tensor:
        """forward_backward through ONE C-ABI call (`vila_sft_fwd_bwd`): the host plans the splice / pack (integers only), the library
        runs every forward and backward kernel and calls back per gradient bucket, where this class starts the exchange + AdamW on its
        optimizer stream exactly as the Python-orchestrated path does."""
        import ctypes as C
        from . import _lib
        from ._lib import check
        model, cfg, flat = self.model, self.cfg, self.flat
        dev = model.device
        for st in (self.opt, self.side):
            if st is not None:
                torch.cuda.current_stream().wait_stream(st)
        flat.grads.zero_()
        self.reducer.log.clear()
        self._touched = []
        c = cfg.llm
        images, frames = self._with_videos(images, videos)
        n_img = len(images)                             # tiles (dynamic_s2: the tiles of every scale of every image; videos: their frames)
        pixels = torch.stack(list(images), 0).to(device=dev, dtype=torch.bfloat16).contiguous() if n_img else None
        s2, rows, n_pin = self._media_plan(n_img, self._block_sizes_with_frames(block_sizes, frames))
        img_blocks, vid_blocks, pools, n_buf = self._media_blocks(rows, frames, n_pin * cfg.tokens_per_tile)
        plan = self._splice(input_ids, attention_mask, labels, img_blocks, vid_blocks)
        rp = repack(plan.mask, plan.labels)
        T = int(rp.rows.numel())
        inv = torch.full((plan.B * plan.S,), -1, dtype=torch.int64)
        inv[rp.rows] = torch.arange(T)
        i32 = lambda t: t.to(torch.int32).contiguous().to(dev)
        txt_src, txt_dst = i32(plan.txt_src), i32(inv[plan.txt_dst.long()])
        mrow = self._media_rows(plan.img_src, img_blocks + vid_blocks)
        dst_p = inv[plan.img_dst.long()]
        is_nl = mrow < 0                                  # token rows: "\n" and the video encoder's start / end / separator tokens
        feat_src, feat_dst, nl_dst = i32(mrow[~is_nl]), i32(dst_p[~is_nl]), i32(dst_p[is_nl])
        nl_src = i32(-1 - mrow[is_nl])
        tgt = torch.full((T,), IGNORE_INDEX, dtype=torch.int64)
        tgt[:-1] = rp.labels[1:]
        valid = torch.nonzero(tgt != IGNORE_INDEX, as_tuple=False).flatten()
        n_valid = int(valid.numel())
        n_items = n_valid if num_items_in_batch is None else int(num_items_in_batch)
        rows32, tg = i32(valid), tgt[valid].contiguous().to(dev)
        pos, cu = i32(rp.position_ids), i32(rp.cu_seqlens)
        b = _lib.VilaSftBatch()
        P = lambda t: t.data_ptr() if t is not None and t.numel() else None
        b.pixels, b.n_images, b.total_tokens = P(pixels), n_img, T
        b.txt_src, b.txt_dst, b.n_txt = P(txt_src), P(txt_dst), int(txt_src.numel())
        b.feat_src, b.feat_dst, b.n_feat = P(feat_src), P(feat_dst), int(feat_dst.numel())
        b.nl_src, b.nl_dst, b.n_nl = P(nl_src), P(nl_dst), int(nl_dst.numel())
        b.positions, b.cu_seqlens, b.n_seq, b.max_seqlen = P(pos), P(cu), int(cu.numel()) - 1, int(rp.max_seqlen)
        b.target_rows, b.targets, b.n_targets, b.loss_scale = P(rows32), P(tg), n_valid, 1.0 / max(n_items, 1)
        if s2 is not None:
            s2_desc, s2_tdesc = s2.desc.contiguous().to(dev), s2.tile_desc.contiguous().to(dev)
            b.s2_desc, b.s2_tile_desc, b.s2_n_blocks, b.s2_n_scales = P(s2_desc), P(s2_tdesc), n_pin, len(cfg.s2_scales)
            for k, v in enumerate(s2.splits[:4]):
                b.s2_splits[k] = int(v)
        if pools:                                         # TSPVideoEncoder: the pooled rows behind the projector's (host array, read during the call)
            pool_arr = (C.c_int32 * (7 * len(pools)))(*[int(x) for b0, nf, pool, off, cnt in pools for x in (b0, nf, *pool, off, cnt)])
            b.pools, b.n_pools, b.n_media_rows = C.cast(pool_arr, C.c_void_p), len(pools), n_buf
        if getattr(self, "_cw", None) is None:
            self._cw = self._c_structs(lambda n: flat.param(n).data_ptr())
            self._cg = self._c_structs(lambda n: flat.grad(n).data_ptr())
        (vw, pw, lw, _), (vg, pg, lg, _) = self._cw, self._cg
        lib = _lib.load()
        need = int(lib.vila_sft_workspace_bytes(C.byref(vw), C.byref(pw), C.byref(lw), C.byref(b)))
        if need == 0:
            check(-1, "vila_sft_workspace_bytes")
        if getattr(self, "_c_ws", None) is None or self._c_ws.numel() < need:
            self._c_ws = None
            self._c_ws = torch.empty((need,), device=dev, dtype=torch.uint8)
        loss = torch.zeros((1,), device=dev, dtype=torch.float32)
        pre_v = "vision_tower.vision_tower.vision_model."
        names = {_lib.BUCKET_LM_HEAD: lambda i: "llm.lm_head.", _lib.BUCKET_FINAL_NORM: lambda i: "llm.model.norm.",
                 _lib.BUCKET_LLM_LAYER: lambda i: f"llm.model.layers.{i}.", _lib.BUCKET_EMBED: lambda i: "llm.model.embed_tokens.",
                 _lib.BUCKET_PROJECTOR: lambda i: "mm_projector.", _lib.BUCKET_VIT_LAYER: lambda i: f"{pre_v}encoder.layers.{i}.",
                 _lib.BUCKET_VIT_EMBED: lambda i: pre_v + "embeddings."}
        err = []

        def on_ready(_arg, bucket, index):
            try:
                self._ready(names[bucket](index))
            except BaseException as e:                      # an exception must not unwind through the C frame
                err.append(e)
        cb = _lib.GRAD_READY_CB(on_ready)
        check(lib.vila_sft_fwd_bwd(C.byref(vw), C.byref(vg), C.byref(pw), C.byref(pg), C.byref(lw), C.byref(lg), C.byref(b), loss.data_ptr(),
                                   self._c_ws.data_ptr(), self._c_ws.numel(), cb, None, ops._stream()), "vila_sft_fwd_bwd")
        if err:
            raise err[0]
        self._announce_absent_media(n_img)
        self._finish_backward()
        return loss[0]

    # ------------------------------------------------------------------ the step -------------------------------------------
    def forward_backward(self, input_ids: torch.Tensor, images: List[torch.Tensor], labels: torch.Tensor,
                         attention_mask: Optional[torch.Tensor] = None, num_items_in_batch: Optional[int] = None,
                         block_sizes=None, videos=None) -> torch.Tensor:
        """Forward + backward of the packed batch; gradients land in self.flat.grads (already all-reduced when DP > 1).
        Returns the (local) loss = sum CE / num_items_in_batch as a device scalar.
        dynamic_s2 (the NVILA-8B recipe, scripts/NVILA/stage1_9tile.sh:19-22): `images` = the tiles of every scale of every image in
        tower order, `block_sizes` = media_config["image"]["block_sizes"] (one (h, w) or None per image); between tower and projector
        run the merge kernel forward and its adjoint backward (llava_arch.py:298-390)."""
        model, cfg, flat = self.model, self.cfg, self.flat
        dev = model.device
        P, G = flat.param, flat.grad
        for st in (self.opt, self.side):             # the previous step's optimizer / exchange / wgrad work reads grads, writes params
            if st is not None:
                torch.cuda.current_stream().wait_stream(st)
        flat.grads.zero_()
        self.reducer.log.clear()
        self._touched = []
        c = cfg.llm
        H = c.hidden_size
        # ---- vision + projector (+ "\n" end token) ----
        images, frames = self._with_videos(images, videos)
        pixels = torch.stack(list(images), 0).to(device=dev, dtype=torch.bfloat16) if len(images) else None
        n_img = 0 if pixels is None else pixels.shape[0]          # tiles (dynamic_s2: of every scale of every image)
        s2, rows, n_pin = self._media_plan(n_img, self._block_sizes_with_frames(block_sizes, frames))
        if n_img:
            feats, vit_saved = self._vit_fwd(pixels)
            if s2 is not None:
                s2_desc, s2_tdesc = s2.desc.to(dev), s2.tile_desc.to(dev)
                feats = ops.s2_merge(feats, s2_desc, len(cfg.s2_scales), s2.splits)   # [n_blocks, N, n_scales*C]
            proj, proj_saved = self._proj_fwd(feats)                                   # [n_pin, T', H]
        table = P("llm.model.embed_tokens.weight")
        # ---- splice + pack (llava_arch.py:412-490, 744-800) ----
        # training truncates every sample to tokenizer.model_max_length AFTER media expansion (llava_arch.py:519-526)
        n_prow = proj.shape[0] * proj.shape[1] if n_img else 0                         # projector output rows
        img_blocks, vid_blocks, pools, n_buf = self._media_blocks(rows, frames, n_prow)
        plan = self._splice(input_ids, attention_mask, labels, img_blocks, vid_blocks)
        rp = repack(plan.mask, plan.labels)
        T = int(rp.rows.numel())
        # packed row index of every padded-grid position
        inv = torch.full((plan.B * plan.S,), -1, dtype=torch.int64, device=rp.rows.device)
        inv[rp.rows] = torch.arange(T, device=rp.rows.device)
        txt_dst = inv[plan.txt_dst.long()].to(torch.int32).to(dev)
        x0 = torch.empty((T, H), device=dev, dtype=torch.bfloat16)
        ops.copy_rows(table, x0, plan.txt_src.to(dev), txt_dst, int(txt_dst.numel()))
        if n_img:
            # every spliced media row is a row of the media feature buffer or a token's embedding (`_media_blocks`); rows cut off by the
            # truncation are absent from plan.img_src / img_dst
            mrow = self._media_rows(plan.img_src, img_blocks + vid_blocks).to(inv.device)
            dst_p = inv[plan.img_dst.long()]
            is_nl = mrow < 0
            feat_src = mrow[~is_nl].to(torch.int32).to(dev)
            feat_dst = dst_p[~is_nl].to(torch.int32).to(dev)
            nl_dst = dst_p[is_nl].to(torch.int32).to(dev)
            nl_src = (-1 - mrow[is_nl]).to(torch.int32).to(dev)
            n_feat, n_nl = int(feat_dst.numel()), int(nl_dst.numel())
            media = proj.reshape(n_prow, H)
            if pools:                                                                  # TSPVideoEncoder: mean over (t, h, w) windows (tsp.py:28-52)
                Tm, grid_l = cfg.tokens_per_tile, int(round(cfg.tokens_per_tile ** 0.5))
                media = torch.cat([media] + [ops.video_pool(proj[t0:t0 + nf], pool) for t0, nf, pool, _, _ in pools], 0)
                assert media.shape[0] == n_buf
            ops.copy_rows(media, x0, feat_src, feat_dst, n_feat)
            if n_nl:
                ops.copy_rows(table, x0, nl_src, nl_dst, n_nl)
        pos = rp.position_ids.to(dev)
        cu = rp.cu_seqlens.to(dev)
        # rows that have a target (HF ForCausalLMLoss: shift by one inside each packed row): integer work on the plan's device — the
        # host for host-side ids — so the step has no nonzero() launch + device sync in its middle
        lab_h = rp.labels
        tgt_h = torch.full((T,), IGNORE_INDEX, dtype=torch.int64, device=lab_h.device)
        tgt_h[:-1] = lab_h[1:]
        valid_h = torch.nonzero(tgt_h != IGNORE_INDEX, as_tuple=False).flatten()
        n_valid = int(valid_h.numel())
        valid, tgt_valid = valid_h.to(dev), tgt_h[valid_h].contiguous().to(dev)
        # ---- LLM ----
        saved = self._llm_fwd(x0, pos, cu, rp.max_seqlen)
        # ---- loss on the rows that have a target ----
        n_items = n_valid if num_items_in_batch is None else int(num_items_in_batch)
        loss = torch.zeros((1,), device=dev, dtype=torch.float32)
        dhn = torch.zeros((T, H), device=dev, dtype=torch.bfloat16)
        head_name = "llm.model.embed_tokens.weight" if c.tie_word_embeddings else "llm.lm_head.weight"
        head = P(head_name)
        if n_valid:
            rows32 = valid.to(torch.int32)
            hv = torch.empty((n_valid, H), device=dev, dtype=torch.bfloat16)
            ops.copy_rows(saved.hn, hv, rows32, None, n_valid)
            # lm_head + CE in row chunks (llava_llama.py:134-149 materialises [sum S, V] fp32 logits; SURVEY §7 step 7: never do that): the
            # rows WITH a target only (the others' gradient is exactly zero), LOGIT_CHUNK of them at a time — the fp32 logits buffer is
            # bounded (256 x 152 064 x 4 B = 156 MB, resident in the 256-MB infinity cache for the CE kernel's three passes) whatever the
            # batch; the bf16 dlogits of all rows are kept for the ONE wgrad / dgrad pair below
            dlog = torch.empty((n_valid, head.shape[0]), device=dev, dtype=torch.bfloat16)
            for r0 in range(0, n_valid, LOGIT_CHUNK):
                r1 = min(n_valid, r0 + LOGIT_CHUNK)
                logits = ops.gemm(hv[r0:r1], head, out_f32=True)                         # [<= LOGIT_CHUNK, V] fp32
                ops.ce_loss(logits, tgt_valid[r0:r1], loss, 1.0 / max(n_items, 1), out=dlog[r0:r1])
            del logits
            dhv = linear_bwd(hv, head, dlog, G(head_name), cm=self.cm, ws=self.ws)
            ops.copy_rows(dhv, dhn, None, rows32, n_valid)
        if not c.tie_word_embeddings:
            self._ready("llm.lm_head.")
        dx = ops.norm_bwd(saved.x_out, P("llm.model.norm.weight"), dhn, G("llm.model.norm.weight"), None, c.rms_norm_eps, True)
        self._ready("llm.model.norm.")
        dx0 = self._llm_bwd(dx, saved)
        # ---- embedding rows (text + "\n") and media rows ----
        ge = G("llm.model.embed_tokens.weight")
        dtxt = torch.empty((int(txt_dst.numel()), H), device=dev, dtype=torch.bfloat16)
        ops.copy_rows(dx0, dtxt, txt_dst, None, int(txt_dst.numel()))
        ops.scatter_add_rows(dtxt, ge, plan.txt_src.to(dev))
        if n_img and n_nl:
            dnl = torch.empty((n_nl, H), device=dev, dtype=torch.bfloat16)
            ops.copy_rows(dx0, dnl, nl_dst, None, n_nl)
            ops.scatter_add_rows(dnl, ge, nl_src)
        self._ready("llm.model.embed_tokens.")
        if n_img and not getattr(self, "skip_proj_bwd", False):
            full = n_feat == n_buf
            dmedia = (torch.empty if full else torch.zeros)((n_buf, H), device=dev, dtype=torch.bfloat16)   # truncated / pooled-only rows: zero
            ops.copy_rows(dx0, dmedia, feat_dst, feat_src, n_feat)
            dproj = dmedia[:n_prow]
            for t0, nf, pool, off, cnt in pools:                                       # the pooled rows' gradients back onto their frames' rows
                ops.video_pool_bwd(dmedia[off:off + cnt], nf, grid_l, pool, out=dproj[t0 * Tm:(t0 + nf) * Tm], accumulate=True)
            dfeats = self._proj_bwd(dproj.view(proj.shape[0], proj.shape[1], H), proj_saved)
            if s2 is not None:                                                         # adjoint of the merge: back onto the tower's tiles
                dfeats = ops.s2_merge_bwd(dfeats, s2_tdesc, len(cfg.s2_scales), s2.splits)
            if not getattr(self, "skip_vit_bwd", False):
                self._vit_bwd(dfeats.reshape(n_img * cfg.vision.num_patches, cfg.vision.hidden_size), vit_saved)
        self._announce_absent_media(n_img)
        self._finish_backward()
        return loss[0]

    def media_bucket_order(self) -> List[str]:
        """The buckets the projector / tower backward announces, in its order (`_proj_bwd`, `_vit_bwd`; BUCKET_PROJECTOR / BUCKET_VIT_* of the
        C-ABI step)."""
        pre = "vision_tower.vision_tower.vision_model."
        if getattr(self, "skip_proj_bwd", False):            # frozen projector + tower (autograd seam, tune_* flags): nothing is announced
            return []
        if getattr(self, "skip_vit_bwd", False):
            return ["mm_projector."]
        return (["mm_projector."] + [f"{pre}encoder.layers.{i}." for i in reversed(range(self.cfg.vision.num_used_layers))] + [pre + "embeddings."])

    def _announce_absent_media(self, n_img: int) -> None:
        """Data-parallel ranks must issue the SAME sequence of collectives.  A rank whose micro-batch is text-only while another rank's has
        images (`self._media_elsewhere`, agreed in `step()` / `agree_on_media`) announces the projector and tower buckets anyway — with the
        zero gradients its backward left there — so every rank takes part in every bucket's all-reduce and applies the same update with the
        same per-bucket step count.  The reference gets the same effect by pushing a dummy image through the encoders on ranks without media
        (llava_arch.py:508-514); here no tower pass is needed, only the rank's seat in the exchange."""
        if n_img or not getattr(self, "_media_elsewhere", False):
            return
        for prefix in self.media_bucket_order():
            self._ready(prefix)

    def _finish_backward(self) -> None:
        for st in (self.side, self.opt):
            if st is not None:
                torch.cuda.current_stream().wait_stream(st)
        if not self._bucket_step:
            self.reducer.wait()
        else:
            self.reducer.handles = []               # every handle was waited for on the optimizer stream

    def optimizer_step(self) -> None:
        f = self.flat
        f.step_count += 1
        if self._bucket_step:                        # AdamW already ran bucket by bucket inside forward_backward
            return
        scale = 1.0
        if self.max_grad_norm is not None:
            norm = float(ops.sumsq(f.grads).sqrt())                  # untouched buckets hold zeros
            scale = min(1.0, self.max_grad_norm / (norm + 1e-6))
        # the same buckets, the same per-bucket step counts as the per-bucket path: results do not depend on VILA_SFT_OPT_STREAM
        for prefix in self._touched:
            self._adamw_bucket(prefix, scale)

    def global_num_items(self, labels_packed_valid: int) -> int:
        """Token count summed over ranks (transformer_normalize_monkey_patch.py:261-263)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            t = torch.tensor([labels_packed_valid], device=self.model.device, dtype=torch.int64)
            dist.all_reduce(t, group=self.group)
            return int(t.item())
        return labels_packed_valid

    def agree_on_media(self, has_media: bool) -> bool:
        """True when SOME rank of the group has media this step (one MAX all-reduce of a flag; a no-op in a world of one).  Sets
        `_media_elsewhere` for `_announce_absent_media`.  `step()` calls it; a caller that drives `forward_backward` itself under data
        parallelism (the autograd seam) calls it through `AutogradSeam.loss`."""
        import torch.distributed as dist
        any_media = bool(has_media)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            t = torch.tensor([1 if has_media else 0], device=self.model.device, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            any_media = bool(int(t.item()))
        self._media_elsewhere = any_media and not has_media
        return any_media

    def _global_counts(self, n_local: int, has_media: bool) -> int:
        """`global_num_items` and `agree_on_media` as ONE all-reduce (SUM of [targets, has-media flag]): a step pays one tiny collective and one
        host sync for both agreements."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            t = torch.tensor([n_local, 1 if has_media else 0], device=self.model.device, dtype=torch.int64)
            dist.all_reduce(t, group=self.group)
            n_global, any_media = int(t[0].item()), bool(int(t[1].item()))
        else:
            n_global, any_media = n_local, bool(has_media)
        self._media_elsewhere = any_media and not has_media
        return n_global

    def step(self, input_ids, images, labels, attention_mask=None, block_sizes=None, videos=None) -> float:
        n_local = count_targets(input_ids, labels, attention_mask, (self.cfg.image_token_id, self.cfg.video_token_id))
        n_global = self._global_counts(n_local, len(images) + len(videos or []) > 0)
        # per-bucket AdamW needs the update to be a function of the bucket alone: not with global-norm clipping
        self._bucket_step = self.opt is not None and self.max_grad_norm is None and self.flat.master is not None
        self._acc = None          # a plain step holds no accumulator: the fp32 sum of `step_accumulated` (4 B per parameter, 32 GB at 8 B
        #                           parameters) is released as soon as accumulation is not in use
        try:
            fb = self.forward_backward_c if self.use_c_abi else self.forward_backward
            loss = fb(input_ids, images, labels, attention_mask, n_global, block_sizes, **({"videos": videos} if videos else {}))
            self.optimizer_step()
        finally:
            self._bucket_step = False
        return loss


    def step_accumulated(self, micro_batches) -> float:
        """ONE optimizer update from several micro-batches (`--gradient_accumulation_steps` of the NVILA scripts; HF `Trainer` counts the
        targets of ALL micro-batches of the update, on all ranks, into `num_items_in_batch` — so every micro-batch's loss is its sum CE over
        that one global count and the gradients simply add up, transformer_normalize_monkey_patch.py:236-249).  Each element of
        `micro_batches` is the keyword dict of `step` (`input_ids`, `images`, `labels`, `attention_mask`, `block_sizes`, `videos`).  The earlier
        micro-batches' gradients are held in one flat FP32 buffer (rounded to bf16 once, with the last micro-batch's); the exchange across ranks and the update happen once, bucket by bucket,
        under the LAST micro-batch's backward (DDP's `no_sync` for the others).  Returns the update's (local) loss = sum over the micro-batches."""
        mbs = [dict(mb) for mb in micro_batches]
        if len(mbs) == 1:
            return self.step(**mbs[0])
        tok = (self.cfg.image_token_id, self.cfg.video_token_id)
        n_local = sum(count_targets(mb["input_ids"], mb["labels"], mb.get("attention_mask"), tok) for mb in mbs)
        media = [len(mb.get("images") or []) + len(mb.get("videos") or []) > 0 for mb in mbs]
        n_global = self._global_counts(n_local, any(media))
        any_media = any(media) or self._media_elsewhere                    # on some rank, in some micro-batch of this update
        if self._acc is None:                      # fp32: 8-16 micro-batches must not round to bf16 after every add
            self._acc = torch.zeros(self.flat.grads.shape, device=self.flat.grads.device, dtype=torch.float32)
        fb = self.forward_backward_c if self.use_c_abi else self.forward_backward
        losses = []
        try:
            for i, mb in enumerate(mbs):
                last = i == len(mbs) - 1
                self._acc_mode = "add" if last else "hold"
                self._bucket_step = last and self.opt is not None and self.max_grad_norm is None and self.flat.master is not None
                # the media buckets hold gradient as soon as ANY micro-batch anywhere had media: the last micro-batch announces them even when
                # it is text-only itself (held sum + zeros), so every rank exchanges and updates the same buckets
                self._media_elsewhere = last and any_media and not media[i]
                losses.append(fb(mb["input_ids"], list(mb.get("images") or []), mb["labels"], mb.get("attention_mask"), n_global,
                                 mb.get("block_sizes"), **({"videos": mb["videos"]} if mb.get("videos") else {})))
                if not last:                         # `_finish_backward` joined the wgrad stream: the micro-batch's gradients are final here
                    ops.grad_accum(self._acc, self.flat.grads, mode=0 if i == 0 else 1)
            self.optimizer_step()
        finally:
            self._bucket_step, self._acc_mode = False, None
        return float(sum(float(l) for l in losses))


def count_targets(input_ids, labels, attention_mask, image_token_id) -> int:
    """Number of label positions that survive the shift + first-label masking of the packed row (host integer work).
    image_token_id: the media token id, or a tuple of them (image, video): a media token's own label never survives the splice."""
    media_ids = tuple(image_token_id) if isinstance(image_token_id, (tuple, list)) else (int(image_token_id),)
    mask = attention_mask.bool() if attention_mask is not None else torch.ones_like(input_ids, dtype=torch.bool)
    n = 0
    for k in range(input_ids.shape[0]):
        ids_k, lab_k = input_ids[k][mask[k]], labels[k][mask[k]]
        if ids_k.numel() == 0:                # fully masked row: contributes nothing
            continue
        keep = lab_k != IGNORE_INDEX
        for t in media_ids:
            keep = keep & (ids_k != t)
        keep[0] = False                       # first token of a sample is never a target (llava_arch.py:760-762 + HF shift)
        n += int(keep.sum())
    return n


# ----------------------------------------------------------------------------------------------------------------------
# autograd seam: the reference's own training call site works unmodified
# ----------------------------------------------------------------------------------------------------------------------
class _SftLossFn(torch.autograd.Function):
    """`loss = model(**inputs).loss; accelerator.backward(loss)` (llava/train/transformer_normalize_monkey_patch.py:183-249) over the
    explicit-backward step: forward() runs the WHOLE forward + backward (`SFTTrainer.forward_backward[_c]`, gradients land in the
    trainer's flat buffer) and returns the loss scalar; backward() only hands those gradients to autograd's consumers — it deposits
    `upstream_grad * flat.grads` into every parameter's `.grad` (accumulating, like autograd does)."""

    @staticmethod
    def forward(ctx, anchor, seam, call):
        ctx.seam = seam
        with torch.no_grad():
            loss = call()
        return loss.detach().clone().reshape(())

    @staticmethod
    def backward(ctx, g):
        ctx.seam.deposit(g)
        return None, None, None


class AutogradSeam:
    """Owns the SFTTrainer behind `HipLlavaLlamaModel.forward(labels=...)` in training mode and the flat `.grad` storage.
    `.grad` of every parameter is a view of ONE flat bf16 buffer (`accum`, same layout as the parameters): a backward is one flat
    `accum (+)= g * grads` pass (16 GB read + write at NVILA-8B) instead of 700 small tensor ops."""

    def __init__(self, model, use_c_abi: Optional[bool] = None, group=None, tune: Optional[Dict[str, bool]] = None):
        self.model = model
        self.trainer = SFTTrainer(model, optimizer_state=False, group=group)       # the optimizer is the caller's (HF Trainer's AdamW)
        if use_c_abi is not None:
            self.trainer.use_c_abi = bool(use_c_abi)
        tune = {"llm.": True, "vision_tower.": True, "mm_projector.": True, **(tune or {})}
        self.tune = tune
        # frozen tower: its backward is never needed; frozen tower AND projector: nothing behind the spliced embeddings is (the explicit
        # backward of the Python-orchestrated driver stops there; the one-call C-ABI step always runs the whole backward)
        self.trainer.skip_vit_bwd = not tune["vision_tower."]
        self.trainer.skip_proj_bwd = not tune["vision_tower."] and not tune["mm_projector."]
        if self.trainer.skip_vit_bwd:
            self.trainer.use_c_abi = False
        self.accum = torch.zeros_like(self.trainer.flat.grads)
        self.params: List[Tuple[torch.nn.Parameter, torch.Tensor]] = []       # the TRAINABLE parameters and their slices of `accum`
        self.spans: List[Tuple[int, int, tuple]] = []
        f = self.trainer.flat
        mods = {"llm.": model.llm, "vision_tower.": model.vision_tower, "mm_projector.": model.mm_projector}
        for name, (o, k, shape) in f.index.items():
            prefix = next(p for p in mods if name.startswith(p))
            prm = _get(mods[prefix], name[len(prefix):])
            prm.requires_grad_(bool(tune[prefix]))
            if tune[prefix]:
                self.params.append((prm, self.accum[o:o + k].view(shape)))
                self.spans.append((o, k, shape))
        if not self.params:
            raise ValueError("enable_autograd: every component is frozen")
        self.all_trainable = all(tune.values())
        self.anchor = self.params[0][0]

    def loss(self, input_ids, images, labels, attention_mask, num_items_in_batch, block_sizes, videos=None) -> torch.Tensor:
        tr = self.trainer
        fb = tr.forward_backward_c if tr.use_c_abi else tr.forward_backward
        tr._bucket_step = False
        tr.agree_on_media(len(images) + len(videos or []) > 0)
        return _SftLossFn.apply(self.anchor, self, lambda: fb(input_ids, images, labels, attention_mask, num_items_in_batch, block_sizes, **({"videos": videos} if videos else {})))

    def deposit(self, g: torch.Tensor) -> None:
        grads = self.trainer.flat.grads
        scale = float(g) if g.numel() == 1 else 1.0
        ours = [p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in self.params]
        none = [p.grad is None for p, _ in self.params]
        with torch.no_grad():
            if all(none):                                   # first backward after zero_grad(set_to_none=True)
                torch.mul(grads, scale, out=self.accum)     # (frozen components' slices are written too but no parameter points at them)
                for p, v in self.params:
                    p.grad = v
            elif all(ours):                                 # gradient accumulation over micro-batches
                self.accum.add_(grads, alpha=scale)
            else:                                           # somebody replaced some .grad tensors: per-parameter accumulation
                for (p, v), (o, k, shape) in zip(self.params, self.spans):
                    gv = grads[o:o + k].view(shape) * scale
                    p.grad = gv.clone() if p.grad is None else p.grad + gv
