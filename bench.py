#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: decode tokens/s (+ TTFT), NVILA-8B bf16, 1 x 448^2 image + 512-token prompt.

    python bench.py --gpus N --steps K --warmup W

A "step" = one greedy decode token through all 28 decoder layers + lm_head + argmax (one hipGraph replay).  The
prefill (ViT + projector + splice + 769-token LLM prefill) runs before the timed region and is reported as TTFT.
N > 1: one process per GPU, independent replicas (inference has no exchange step: SURVEY.md §8e), value = total tokens/s
over all ranks, time = max over ranks.  Inputs and weights are resident in HBM before timing.  Prints ONE JSON line on rank 0.

Launching: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks come from the environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Started plainly as `python bench.py --gpus N` with N > 1 (no WORLD_SIZE in the
environment) it re-executes itself under torch.distributed.run on 127.0.0.1 — the reference launches its ranks the same way
(`torchrun --nproc_per_node`, scripts/NVILA-Lite/sft.sh:15-17).  `--selftest` runs only the harness (rendezvous, barrier, max-over-
ranks timing, the one JSON line) on the gloo backend without touching a GPU: tests/test_bench_launcher_cpu.py drives it with 2 ranks.

Besides the decode metric the default line carries `prefill` (TTFT as TFLOP/s against the MFMA peak) and `sft` (one warm + two timed
NVILA-8B SFT steps, BASELINE configs[2] per-GPU workload) so that the MFMA-bound halves of the north star are driver-observed too,
and `sustained` (a >= 2 s decode replay after the timed region, so that a 5-s GPU-busy sampler has something to see).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s
A100_DECODE_TOKS = 82.1   # NVILA-8B FP16 decode tok/s on A100 (README.md:65) — other hardware, fp16
SFT_TFLOP_PER_SAMPLE = 35.8   # fwd+bwd of one 769-token sample (1 image + 512 text), no recompute


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt-tokens", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sft", action="store_true", help="decode mode: skip the bounded SFT sub-measurement (1 warm + 2 timed steps)")
    ap.add_argument("--no-sustain", action="store_true", help="decode mode: skip the >= 2 s sustained replay after the timed region")
    ap.add_argument("--eager-decode", action="store_true", help="experiment: launch the kernels of a token eagerly instead of replaying a hipGraph")
    ap.add_argument("--config", default="nvila_8b", choices=["nvila_8b", "reduced", "nvila_lite_3b"],
                    help="nvila_lite_3b = BASELINE configs[0] on the GPU (3x3 projector, 2048-wide 36-layer tied-head LLM; use --prompt-tokens 32)")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--tsp", action="store_true", help="video mode: TSPVideoEncoder pool_sizes=[[8,1,1]] (scripts/NVILA/stage4.sh:50) -> 8 x 257 tokens")
    ap.add_argument("--mode", default="decode", choices=["decode", "sft", "video"],
                    help="decode = BASELINE.json metric (default); sft = one data-parallel SFT step (BASELINE configs[2])")
    ap.add_argument("--micro-batch", type=int, default=4)
    ap.add_argument("--w4", action="store_true", help="W4A16 decode (int4 group-128 decoder projections; BASELINE configs[4], SURVEY 8f row 3)")
    ap.add_argument("--w8-vit", action="store_true", help="W8A8 vision tower (int8 x int8 per-channel ViT GEMMs; BASELINE configs[4])")
    ap.add_argument("--dynamic-s2", action="store_true", help="full NVILA-8B recipe: 14 tiles (448/896/1344) -> 2304 image tokens (SURVEY 8f row 1)")
    ap.add_argument("--batch", type=int, default=1, help="decode mode: sequences per weight pass (2..16 = the batched decode step, value = aggregate tokens/s)")
    ap.add_argument("--selftest", action="store_true", help="harness only (gloo, no GPU): rendezvous + barrier + max-over-ranks + JSON line")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` spawns its own N ranks
# ----------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a, argv) -> int:
    """Re-execute this file under torch.distributed.run with one process per GPU.  Returns the child's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL over xGMI needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                  # scripts/setups/train.sh:59
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """fd 1 -> stderr for the duration, INCLUDING what C code wrote with stdio: RCCL prints its banner with printf; with stdout a pipe the text sits
    in libc's buffer and would be flushed at process exit — behind the JSON line, on the restored fd 1 (seen on hardware in round 4: five banner
    lines after the line).  So libc's buffers are flushed before fd 1 is restored."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)


def init_dist(backend: str, dev=None):
    """Process group from the torchrun environment.  RCCL prints a version banner on STDOUT when the communicator is created;
    stdout must carry the one JSON line only, so fd 1 points at stderr while the group and its first collective come up."""
    import torch.distributed as dist
    if "RANK" not in os.environ:            # VILA_BENCH_FORCE_DIST on a plain 1-process start: a world of one on the loopback
        os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    with stdout_to_stderr():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()
        if backend == "nccl":
            torch.cuda.synchronize()
    return dist


class Deadline:
    """A wall-clock guard around a multi-rank side measurement.  When it expires, the process prints `make_line()` (rank 0: the bench's ONE JSON
    line; other ranks: nothing) and exits with status 0 through os._exit — the main thread may be stuck inside a collective that will never
    complete, so nothing that needs it (atexit handlers, process-group teardown) can be waited for.  `cancel()` -> True when the guard was
    stopped in time, False when it has fired (the caller must then stay away from stdout)."""

    def __init__(self, seconds: float, make_line=None, _exit=os._exit):
        import threading
        self._lock = threading.Lock()
        self._fired = False
        self._make_line, self._exit = make_line, _exit
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._t.start()

    def _fire(self):
        with self._lock:
            self._fired = True
            try:
                if self._make_line is not None:
                    sys.stdout.write(self._make_line() + "\n")
                sys.stdout.flush()
                sys.stderr.write("bench.py: SFT side measurement hit its deadline; exiting with the decode line only\n")
                sys.stderr.flush()
            finally:
                self._exit(0)

    def cancel(self) -> bool:
        with self._lock:
            if self._fired:
                return False
            self._t.cancel()
            return True


def timed_region(dist, dev, steps: int, step_fn, sync):
    """The contract's bracket: barrier + sync, EXACTLY `steps` steps, sync + barrier, MAX over ranks."""
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], device=dev if dev is not None else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def selftest_main(a, rank, world):
    """Harness self-test on CPU/gloo: same launcher, rendezvous, barrier, max-over-ranks and JSON plumbing as the GPU modes; the
    "step" is a fixed sleep, so value ~= world / 2 ms.  NOT a performance number."""
    dist = init_dist("gloo") if world > 1 else None
    extra = {}
    step = lambda: time.sleep(0.002)
    # dry run of the SFT data-parallel plumbing (no kernels): the trainer's flat buffers over a tiny CPU model, the global token count, the
    # media agreement and every gradient bucket's exchange over the process group, in the backward order the real step announces them.
    # `--mode sft` times it; the default mode runs it once — the GPU default mode runs the real SFT side measurement on every rank.
    from vila_amd import configs
    from vila_amd.train import SFTTrainer
    from vila_amd.vlm import HipLlavaLlamaModel
    torch.manual_seed(0)
    cfg = configs.tiny("mlp_downsample")
    tr = SFTTrainer(HipLlavaLlamaModel(cfg, device="cpu"), optimizer_state=False)
    tr.flat.grads = tr.flat.grads.float()
    order = ["llm.lm_head.", "llm.model.norm."] + [f"llm.model.layers.{i}." for i in reversed(range(cfg.llm.num_hidden_layers))]
    order += ["llm.model.embed_tokens."] + tr.media_bucket_order()
    box = {}

    def sft_step():
        box["n"] = tr.global_num_items(100 + rank)
        box["media"] = tr.agree_on_media(True)
        tr.reducer.log.clear()
        tr.flat.grads.fill_(float(rank + 1))
        for pre in order:
            tr._ready(pre)
        tr._finish_backward()
    sft_step()
    covered = torch.zeros(tr.flat.numel, dtype=torch.bool)
    for _, s0, e0 in tr.reducer.log:
        covered[s0:e0] = True
    want = float(sum(range(1, world + 1)))
    dry = {"global_num_items": box["n"], "buckets": len(tr.reducer.log), "exchange_ok": bool((tr.flat.grads[covered] == want).all()),
           **exchange_summary(tr, world)}
    if a.mode == "sft":
        step = sft_step
        extra = {"mode": "sft dry run", **dry}
    for _ in range(a.warmup):
        step()
    elapsed = timed_region(dist, None, a.steps, step, lambda: None)
    group_world = dist.get_world_size() if dist is not None else 1
    if rank == 0:
        print(json.dumps({"metric": "bench harness selftest (no GPU work)", "value": round(world * a.steps / elapsed, 2), "unit": "steps/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "selftest", "parallelism": f"gloo x{world}", "group_world_size": group_world,
                                     "requested_gpus": a.gpus, **extra},
                          "sft": {"dry_run": True, **dry}}))
    if dist is not None:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------------
# algorithmic work
# ----------------------------------------------------------------------------------------------------------------------
def decode_bytes_per_token(cfg, ctx: int, w4: bool = False) -> int:
    """Algorithmic HBM bytes per decoded token: every layer + lm_head weight once (bf16) + the KV cache.
    W4A16: layer weights cost 0.5 B + 4 B per 128-group (scale|zero) = 0.53125 B each; lm_head stays bf16."""
    c = cfg.llm
    per_layer = (c.q_size + 2 * c.kv_size) * c.hidden_size + c.q_size * c.hidden_size + 3 * c.hidden_size * c.intermediate_size
    if w4:
        w = per_layer * c.num_hidden_layers * 17 // 32 + c.vocab_size * c.hidden_size * 2
    else:
        w = (per_layer * c.num_hidden_layers + c.vocab_size * c.hidden_size) * 2
    kv = 2 * c.kv_size * 2 * c.num_hidden_layers * ctx
    return w + kv


def vit_flops(cfg, n_tiles: int) -> float:
    v = cfg.vision
    N, d, f = v.num_patches, v.hidden_size, v.intermediate_size
    per_layer = 4 * 2 * N * d * d + 2 * 2 * N * d * f + 2 * 2 * N * N * d
    return n_tiles * (v.num_used_layers * per_layer + 2 * N * v.num_channels * v.patch_size ** 2 * d)


def projector_flops(cfg, n_tiles: int) -> float:
    k, c, h = cfg.downsample, cfg.mm_hidden_size, cfg.llm.hidden_size
    T = cfg.tokens_per_tile
    if cfg.mm_projector_type == "mlp_downsample_3x3_fix":
        return n_tiles * 2 * T * (9 * c * 3 * c + 3 * c * h + h * h)
    return n_tiles * 2 * T * (k * k * c * h + h * h)


def llm_prefill_flops(cfg, S: int) -> float:
    c = cfg.llm
    d, f = c.hidden_size, c.intermediate_size
    per_layer = 2 * S * d * (c.q_size + 2 * c.kv_size) + 2 * S * c.q_size * d + 6 * S * d * f + 2 * S * S * c.q_size
    return c.num_hidden_layers * per_layer + 2 * d * c.vocab_size


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle "port", bounded sample)
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, n_prompt: int, threads: int):
    """Oracle ("port") timed on the host cores on a bounded sample of the SAME workload (fp32, NVILA-8B widths, random weights):
      * decode: 2 of 28 decoder layers, S-token prefill then 4 decode tokens; per-token = layer part x 14 + measured lm_head
      * TTFT : 2 of 26 ViT layers x 13 + patch embed + projector (1 tile) + the 2-layer prefill x 14 + last-row lm_head
    The prefill sample uses an 8-row stand-in head so that it times the layers only (HF generate keeps one logits row)."""
    from oracle import vila_oracle as O
    from vila_amd import configs, synthetic
    # torch's CPU GEMMs stop scaling (and then regress) long before 256 SMT threads on these shapes: take the fastest of a few thread
    # counts on one representative matmul (the prompt x gate_proj GEMM) and report THAT count as `cores`
    a_, b_ = torch.randn(n_prompt, cfg.llm.hidden_size), torch.randn(cfg.llm.intermediate_size, cfg.llm.hidden_size)
    best = (float("inf"), threads)
    for nt in sorted({min(threads, c) for c in (16, 32, 64, 128, threads)}):
        torch.set_num_threads(nt)
        torch.nn.functional.linear(a_, b_)
        t0 = time.perf_counter()
        torch.nn.functional.linear(a_, b_)
        best = min(best, (time.perf_counter() - t0, nt))
    threads = best[1]
    del a_, b_
    torch.set_num_threads(threads)
    L, LV = 2, 2
    c = configs.reduced_8b(layers_v=LV + 1, layers_l=L, vocab=cfg.llm.vocab_size)     # LV+1 layers: select_layer=-2 runs LV of them
    g = torch.Generator().manual_seed(0)

    def draw(specs):
        return {n: torch.randn(shape, generator=g) * 0.02 if k in ("w", "h") else torch.ones(shape) for n, shape, k in specs}
    w = draw([s for s in synthetic.llm_specs(c) if "embed_tokens" not in s[0]])
    head = w["llm.lm_head.weight"]
    w["llm.model.embed_tokens.weight"] = head           # timing only: share the table to bound host RAM
    e = torch.randn(1, n_prompt, c.llm.hidden_size, generator=g)
    with torch.no_grad():
        w["llm.lm_head.weight"] = head[:8]
        t0 = time.perf_counter()
        _, past = O.qwen2_forward(e, w, c.llm)
        t_prefill = time.perf_counter() - t0
        w["llm.lm_head.weight"] = head
        x = torch.randn(1, 1, c.llm.hidden_size, generator=g)
        n_tok = 4
        t0 = time.perf_counter()
        for _ in range(n_tok):
            _, past = O.qwen2_forward(x, w, c.llm, past=past)
        t_tok = (time.perf_counter() - t0) / n_tok
        h = torch.randn(1, 1, c.llm.hidden_size)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.linear(h, head)
        t_head = (time.perf_counter() - t0) / 3
        del past
        wv = draw(synthetic.vision_specs(c) + synthetic.projector_specs(c))
        px = torch.rand(1, 3, c.vision.image_size, c.vision.image_size, generator=g) * 2 - 1
        t0 = time.perf_counter()
        feats = O.vision_tower_forward(px, wv, c.vision)
        t_vit = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.projector_forward(feats, wv, c.mm_projector_type)
        t_proj = time.perf_counter() - t0
    nl, nv = cfg.llm.num_hidden_layers, cfg.vision.num_used_layers
    t_full = (t_tok - t_head) * (nl / L) + t_head
    ttft = t_vit * (nv / LV) + t_proj + t_prefill * (nl / L) + t_head
    del w, wv, head
    hf = hf_reference_leg(cfg, n_prompt, threads)
    live = hf is not None and "decode_tokens_per_s" in hf
    # `value`: the reference's own decoder (HF Qwen2) timed in THIS run when transformers could build it ("reference"), else the oracle port
    return {"value": hf["decode_tokens_per_s"] if live else round(1.0 / t_full, 3), "unit": "tokens/s", "cores": threads, "host_cpus": os.cpu_count(),
            "kind": "reference" if live else "port", "reference_live": hf, "port_tokens_per_s": round(1.0 / t_full, 3),
            "cores_note": "cores = torch threads actually used (calibrated: all hardware threads measured slower); host_cpus = os.cpu_count()",
            "sample": f"oracle/vila_oracle.py fp32 at NVILA-8B widths: {L} of {nl} decoder layers + full lm_head, {n_prompt}-token prefill "
                      f"({t_prefill:.2f}s, layers only) then {n_tok} decode tokens ({t_tok*1e3:.0f} ms/token measured; layer part scaled x{nl // L}); "
                      f"TTFT leg: {LV} of {nv} ViT layers ({t_vit:.2f}s, scaled x{nv / LV:g}) + projector ({t_proj:.2f}s) + prefill x{nl // L} + lm_head row",
            "prefill_s_sample": round(t_prefill, 3), "vit_s_sample": round(t_vit, 3), "projector_s": round(t_proj, 3),
            "ttft_s": round(ttft, 2), "ttft_note": "CPU TTFT estimate for the same 1 image + prompt workload (scaled from the samples above)",
            "reference_full_depth": _reference_cpu_timing()}


def hf_reference_leg(cfg, n_prompt: int, threads: int, L: int = 2, n_tok: int = 4):
    """The reference's OWN decoder class timed live on this host, in this invocation ("missing" 6): HF `Qwen2ForCausalLM` — what
    llava/model/language_model/builder.py:64 instantiates; `transformers` ships with the image, so unlike /root/reference it IS on the GPU box —
    at NVILA-8B widths with L of the 28 layers, fp32 eager attention, random weights: an n_prompt-token prefill through `forward(inputs_embeds=...)`
    and n_tok decode tokens through its KV cache (llava_arch.py:833 -> GenerationMixin's per-token forward).  Per-token time of the full model =
    layer part x 28 / L + the measured lm_head.  Returns None when transformers cannot build the model here."""
    try:
        import transformers
        from transformers import Qwen2Config, Qwen2ForCausalLM
        c = cfg.llm
        hc = Qwen2Config(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size, num_hidden_layers=L,
                         num_attention_heads=c.num_attention_heads, num_key_value_heads=c.num_key_value_heads, rms_norm_eps=c.rms_norm_eps,
                         rope_theta=c.rope_theta, tie_word_embeddings=False, max_position_embeddings=4096, use_sliding_window=False,
                         attention_dropout=0.0, pad_token_id=None, bos_token_id=None)
        hc._attn_implementation = "eager"
        torch.set_num_threads(threads)
        init = torch.nn.Linear.reset_parameters, torch.nn.Embedding.reset_parameters
        torch.nn.Linear.reset_parameters = lambda self: None
        torch.nn.Embedding.reset_parameters = lambda self: None
        try:
            llm = Qwen2ForCausalLM(hc).eval().float()
        finally:
            torch.nn.Linear.reset_parameters, torch.nn.Embedding.reset_parameters = init
        g = torch.Generator().manual_seed(0)
        with torch.no_grad():
            for n_, prm in llm.named_parameters():
                if prm.dim() > 1:
                    prm.normal_(0.0, 0.02, generator=g)
                elif "norm" in n_:
                    prm.fill_(1.0)
                else:
                    prm.zero_()
            e = torch.randn(1, n_prompt, c.hidden_size, generator=g) * 0.5
            t0 = time.perf_counter()
            r = llm(inputs_embeds=e, use_cache=True, logits_to_keep=1)
            t_prefill = time.perf_counter() - t0
            past = r.past_key_values
            tok = torch.tensor([[1]])
            r = llm(input_ids=tok, past_key_values=past, use_cache=True)          # warm (allocations of the cache growth)
            past = r.past_key_values
            t0 = time.perf_counter()
            for _ in range(n_tok):
                r = llm(input_ids=tok, past_key_values=past, use_cache=True)
                past = r.past_key_values
            t_tok = (time.perf_counter() - t0) / n_tok
            h = torch.randn(1, 1, c.hidden_size)
            t0 = time.perf_counter()
            for _ in range(3):
                llm.lm_head(h)
            t_head = (time.perf_counter() - t0) / 3
        nl = c.num_hidden_layers
        t_full = max(t_tok - t_head, 1e-9) * (nl / L) + t_head
        return {"decode_tokens_per_s": round(1.0 / t_full, 3), "decode_s_per_token": round(t_full, 4), "prefill_s_scaled": round(t_prefill * nl / L, 3),
                "measured": {"layers": L, "prefill_s": round(t_prefill, 3), "decode_s_per_token": round(t_tok, 4), "lm_head_s": round(t_head, 4)},
                "threads": threads, "dtype": "fp32", "hf_version": transformers.__version__,
                "what": f"HF Qwen2ForCausalLM (the reference's decoder class) executed on THIS host in THIS run: {L} of {nl} layers at NVILA-8B widths + "
                        f"the full lm_head, {n_prompt}-token prefill then {n_tok} decode tokens through its KV cache; layer part scaled x{nl / L:g}"}
    except Exception as ex:                                   # pragma: no cover  (an HF API drift must not take the bench line down)
        return {"error": f"{type(ex).__name__}: {ex}"}


def _reference_cpu_timing():
    """One-off, committed: the REFERENCE's own code (reference SigLIP + projector, HF Qwen2ForCausalLM fp32) timed at the full 26 + 28 layer depth
    on the build container's CPU while it produced tests/golden/nvila8b_full_depth_ref.npz (oracle/make_golden_full_ref.py) — other host than the
    GPU box's, core count inside; stands beside the scaled `port` figure measured here."""
    path = os.path.join(ROOT, "profiles", "r04_cpu_reference_timing.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    return {k: d.get(k) for k in ("decode_tokens_per_s", "decode_s_per_token", "ttft_s", "prefill_s", "tower_projector_s", "threads", "dtype", "hf_version", "what")} | {
        "source": "profiles/r04_cpu_reference_timing.json"}


# ----------------------------------------------------------------------------------------------------------------------
# SFT step (BASELINE configs[2])
# ----------------------------------------------------------------------------------------------------------------------
def sft_measure(model, cfg, a, rank, dev, dist, steps: int, warmup: int):
    """Per-GPU micro-batch of b samples (1 x 448^2 image + 512 text tokens, S = 769, packed), labels on the last 256 text
    positions, all 8.06 B params trainable, AdamW lr 2e-5 (scripts/NVILA-Lite/sft.sh:41-42).  Returns (seconds for `steps`, loss, S, trainer)."""
    from vila_amd import synthetic
    from vila_amd.train import SFTTrainer
    tr = SFTTrainer(model, lr=2e-5, weight_decay=0.0)
    b = a.micro_batch
    s2 = bool(getattr(cfg, "dynamic_s2", False))
    tiles = 14 if s2 else 1                   # dynamic_s2 (the NVILA-8B recipe): a square image = 1 + 4 + 9 tiles of 448^2, block size (3, 3)
    pixels = synthetic.make_pixels(cfg, b * tiles, rank, device=dev, dtype=torch.bfloat16)
    ids = torch.stack([synthetic.make_prompt(cfg, a.prompt_tokens, 1, 10 * rank + i) for i in range(b)], 0)
    labels = ids.clone()
    labels[:, : 1 + a.prompt_tokens - 256] = -100
    S = (cfg.tokens_per_tile * 9 if s2 else cfg.tokens_per_tile) + 1 + a.prompt_tokens
    images = [pixels[i] for i in range(b * tiles)]
    blocks = [(3, 3)] * b if s2 else None
    box = {"loss": float("nan")}

    def one():
        box["loss"] = tr.step(ids, images, labels, block_sizes=blocks)
    for _ in range(warmup):
        one()
    elapsed = timed_region(dist, dev, steps, one, torch.cuda.synchronize)
    return elapsed, float(box["loss"]), S, tr


def sft_flops_per_sample(cfg, S: int, n_targets: int = 256) -> float:
    """fwd + bwd (3x the forward GEMM / attention work, no recompute) of one packed sample: tower on its tiles, projector on its blocks,
    S-token decoder, lm_head on the rows that have a target.  The plain 769-token sample gives SURVEY §8d's 35.8 TFLOP."""
    s2 = bool(getattr(cfg, "dynamic_s2", False))
    fwd = vit_flops(cfg, 14 if s2 else 1) + projector_flops(cfg, 9 if s2 else 1)
    fwd += llm_prefill_flops(cfg, S) - 2 * cfg.llm.hidden_size * cfg.llm.vocab_size + 2 * n_targets * cfg.llm.hidden_size * cfg.llm.vocab_size
    return 3.0 * fwd


def _sft_gemm_traffic():
    """L2-fill bytes per GEMM of the contraction-major (wgrad) kernel on the four decoder shapes of the step, from the committed rocprofv3 --pmc
    passes (tools/pmc_gemm_sft.sh -> profiles/r0N_pmc_gemm_sft.json, newest round first; cannot be collected inside this process), next to the algorithmic bytes."""
    name = next((n_ for n_ in ("r06_pmc_gemm_sft.json", "r05_pmc_gemm_sft.json", "r04_pmc_gemm_sft.json", "r03_pmc_gemm_sft.json") if os.path.exists(os.path.join(ROOT, "profiles", n_))), None)
    if name is None:
        return None
    tj = os.path.join(ROOT, "profiles", name)
    with open(tj) as f:
        d = json.load(f)
    return {"kernel": "gemm256_kernel<0,0,true,true,7,256> (wgrad, operands as they lie)",
            "per_gemm": {k: {"traffic_bytes": v.get("traffic_bytes"), "algorithmic_bytes": v.get("algorithmic_bytes"), "ratio": v.get("traffic_over_algorithmic")}
                         for k, v in d.items()},
            "source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per shape over tools/gemm_bench with the kernels and tile order of that round's "
                      "final build, FETCH x2 on gfx950; L2 fills, i.e. incl. what the infinity cache serves)"}


def sft_block(elapsed: float, steps: int, b: int, S: int, world: int, loss: float, tflop_per_sample: float = SFT_TFLOP_PER_SAMPLE):
    step_s = elapsed / steps
    flops = tflop_per_sample * 1e12 * b
    return {"ms_per_step": round(step_s * 1e3, 2), "tokens_per_s": round(world * b * S / step_s, 1), "steps": steps, "micro_batch": b, "loss": round(loss, 4),
            "roofline": {"bound": "mfma", "achieved": round(flops / step_s / 1e12, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(flops / step_s / (MFMA_PEAK_TF * 1e12), 4), "traffic": _sft_gemm_traffic(),
                         "note": f"whole step incl. optimizer; {tflop_per_sample:.1f} TFLOP per {S}-token sample fwd+bwd (SURVEY §8d), no activation recompute; "
                                 "max_grad_norm = None (scripts/NVILA-Lite/sft.sh sets no clipping)"}}


SFT_1GPU_FILE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "vila_bench_sft_1gpu_ms.txt")


def exchange_summary(tr, world: int) -> dict:
    """What one step hands to the process group: the distinct gradient buckets the reducer announced (bf16 slices of the flat buffer)."""
    spans = {pre: (s0, e0) for pre, s0, e0 in tr.reducer.log}
    nbytes = sum(e0 - s0 for s0, e0 in spans.values()) * tr.flat.grads.element_size()
    out = {"world": world, "grad_exchange": tr.reducer.describe(), "exchange_algo": tr.reducer.algo,
           "exchange_bytes": int(nbytes), "exchange_active": bool(tr.reducer.active())}
    out.update(rccl_ranks_seen(tr.reducer.dist if tr.reducer.active() else None, tr.flat.grads.device))
    return out


def rccl_ranks_seen(dist, dev) -> dict:
    """The ranks the process group REALLY connects, counted through the group itself (an all-gather of every rank's id and device index on the
    compute device): the driver's SCALE record can confirm that --gpus N ran N ranks over RCCL and not N copies of a world of one."""
    if dist is None:
        return {"rccl_ranks_seen": 1, "rccl_backend": "none"}
    w = dist.get_world_size()
    mine = torch.tensor([dist.get_rank(), torch.cuda.current_device() if dev.type == "cuda" else -1], device=dev, dtype=torch.int64)
    got = [torch.empty_like(mine) for _ in range(w)]
    dist.all_gather(got, mine)
    ranks = sorted({int(t[0]) for t in got})
    return {"rccl_ranks_seen": len(ranks), "rccl_rank_ids": ranks, "rccl_devices": [int(t[1]) for t in got], "rccl_backend": dist.get_backend()}


def sft_side_measurement(model, cfg, a, rank, world, dev, dist):
    """The `sft` block of the default bench line.  All ranks agree (one tiny all-reduce) that the trainer fits before any of them enters a
    step, so a rank that failed to allocate cannot leave the others waiting inside a gradient all-reduce."""
    group = dist if (dist is not None and (world > 1 or os.environ.get("VILA_BENCH_FORCE_DIST"))) else None
    try:
        if group is not None:
            free = torch.cuda.mem_get_info(dev)[0]
            ok = torch.tensor([1.0 if free > 150e9 else 0.0], device=dev)
            group.all_reduce(ok, op=group.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                return {"error": f"skipped: a rank has < 150 GB free (this rank: {free / 1e9:.0f} GB)", "world": world}
        el, loss, S_sft, tr = sft_measure(model, cfg, a, rank, dev, group, 2, 1)
        blk = sft_block(el, 2, a.micro_batch, S_sft, world, loss)
        blk.update(exchange_summary(tr, world))
        one = os.environ.get("VILA_BENCH_SFT_1GPU_MS")
        if world == 1 and group is None:
            try:
                with open(SFT_1GPU_FILE, "w") as f:
                    f.write(str(blk["ms_per_step"]))
            except OSError:
                pass
        elif one is None and os.path.exists(SFT_1GPU_FILE):
            one = open(SFT_1GPU_FILE).read().strip()
        if world > 1:
            try:
                blk["weak_scaling_eff"] = round(float(one) / blk["ms_per_step"], 4) if one else None
            except ValueError:
                blk["weak_scaling_eff"] = None
            blk["weak_scaling_note"] = ("1-GPU ms/step / N-GPU ms/step at the same per-GPU micro-batch; the 1-GPU figure comes from VILA_BENCH_SFT_1GPU_MS or "
                                        f"from the N = 1 run of this bench on the same box ({SFT_1GPU_FILE}); null when neither exists")
        return blk
    except Exception as ex:      # the decode line must survive a failure of the side measurement; say so loudly in the JSON
        return {"error": f"{type(ex).__name__}: {ex}", "world": world}


def sft_main(a, rank, world, dev, dist):
    from vila_amd import configs
    from vila_amd.vlm import build_model
    cfg = configs.nvila_8b() if a.config == "nvila_8b" else configs.reduced_8b(3, 4)
    if a.dynamic_s2:
        cfg.dynamic_s2 = True                 # the recipe NVILA-8B is actually trained with (scripts/NVILA/stage1_9tile.sh:19-22)
        cfg.name += "-dynamic-s2"
    model = build_model(cfg, seed=0, device=dev)
    elapsed, loss, S, tr = sft_measure(model, cfg, a, rank, dev, dist, a.steps, a.warmup)
    if rank == 0:
        blk = sft_block(elapsed, a.steps, a.micro_batch, S, world, loss,
                        sft_flops_per_sample(cfg, S) / 1e12 if a.dynamic_s2 else SFT_TFLOP_PER_SAMPLE)
        print(json.dumps({
            "metric": ("SFT step throughput, NVILA-8B dynamic_s2 recipe, packed (1 + 4 + 9 tiles of 448^2) image + 512-token samples" if a.dynamic_s2
                       else "SFT step throughput, NVILA-8B, packed 1x448^2 image + 512-token samples"), "value": blk["tokens_per_s"],
            "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": blk["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (fp32 master/AdamW state)",
            "data": "synthetic", "loss": blk["loss"],
            "config": {"workload": f"{cfg.name} SFT step, micro-batch {a.micro_batch} x S={S} packed, all params trainable, AdamW",
                       "parallelism": f"dp{world}", **exchange_summary(tr, world)},
            "roofline": blk["roofline"]}))
    if dist is not None:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------------
# video prefill (BASELINE configs[3])
# ----------------------------------------------------------------------------------------------------------------------
def video_main(a, rank, dev):
    """NVILA-Video-8B-style prefill of `--frames` 448^2 frames, mlp_downsample_2x2_fix projector.  Two token layouts:
    (i) per-frame <image> tokens (llava/utils/media.py:114-119: 64 x 257 = 16448 media tokens + 32 text);
    (ii) --tsp: TSPVideoEncoder pool_sizes=[[8,1,1]] (scripts/NVILA/stage4.sh:50): temporal mean-pool by 8 -> 8 x 257 = 2056 tokens."""
    from vila_amd import configs, ops, synthetic
    from vila_amd.vlm import build_model
    cfg = configs.nvila_8b()
    cfg.mm_projector_type = "mlp_downsample_2x2_fix"
    model = build_model(cfg, seed=0, device=dev)
    F_ = a.frames
    pixels = synthetic.make_pixels(cfg, F_, 0, device=dev, dtype=torch.bfloat16)
    frames = [pixels[i] for i in range(F_)]
    if a.tsp:
        from vila_amd.vlm import TSPVideoEncoder
        model.encoders["video"] = TSPVideoEncoder(model, [[8, 1, 1]])      # the hydra target of scripts/NVILA/stage4.sh:50
        n_media = F_ // 8 * (cfg.tokens_per_tile + 1)
        ids = synthetic.make_prompt(cfg, 32, 1, 0)[None].to(dev)
        ids[0, 0] = cfg.video_token_id
        media, media_cfg = {"video": [torch.stack(frames, 0)]}, {}
    else:
        n_media = F_ * (cfg.tokens_per_tile + 1)
        ids = synthetic.make_prompt(cfg, 32, F_, 0)[None].to(dev)
        media, media_cfg = {"image": frames}, {}
    S = n_media + 32
    cache = model.llm.new_cache(((S + 64 + 255) // 256) * 256)

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e, _, _ = model._embed(ids, media, media_cfg)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        assert e.shape[1] == S, (e.shape, S)
        r = model.llm.prefill_packed(e[0], torch.arange(S, device=dev, dtype=torch.int32), None, S, cache=cache,
                                     last_rows=torch.tensor([S - 1], device=dev, dtype=torch.int32))
        first = int(ops.argmax(r.last_logits[0]))
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, first
    once()
    ts = [once() for _ in range(max(a.steps if a.steps != 128 else 3, 1))]
    enc = statistics.median(t[0] for t in ts)
    pre = statistics.median(t[1] for t in ts)
    flops = vit_flops(cfg, F_) + projector_flops(cfg, F_) + llm_prefill_flops(cfg, S)
    print(json.dumps({"metric": "TTFT, NVILA-Video-8B-style prefill", "value": round((enc + pre) * 1e3, 2), "unit": "ms", "n_gpus": 1,
                      "higher_is_better": False, "dtype": "bf16", "data": "synthetic",
                      "config": {"workload": f"{F_} frames x 448^2 -> {S} tokens ({'TSPVideoEncoder pool [[8,1,1]]' if a.tsp else 'per-frame <image> tokens'}), batch 1"},
                      "encode_ms": round(enc * 1e3, 2), "llm_prefill_ms": round(pre * 1e3, 2),
                      "roofline": {"bound": "mfma", "achieved": round(flops / (enc + pre) / 1e12, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": round(flops / (enc + pre) / (MFMA_PEAK_TF * 1e12), 4), "traffic": None},
                      "reference_note": "README.md:84: 0.7190 s on A100 FP16 (TinyChat, 64 frames, pooled tokens) - other hardware"}))


# ----------------------------------------------------------------------------------------------------------------------
# decode (the BASELINE metric)
# ----------------------------------------------------------------------------------------------------------------------
def batch_decode_main(a, rank, world, dev, dist):
    """Serving throughput: `--batch B` sequences (1 image + prompt each) decoded by ONE weight pass per step (vila_llm_decode_step_batch,
    hipGraph replay); value = aggregate generated tokens / s over the timed steps."""
    import ctypes as C
    from vila_amd import _lib, configs, ops, synthetic
    from vila_amd._lib import check
    from vila_amd.vlm import build_model
    lib = _lib.load()
    cfg = configs.nvila_8b() if a.config == "nvila_8b" else configs.reduced_8b(3, 4)
    model = build_model(cfg, seed=0, device=dev)
    llm = model.llm
    Bn = a.batch
    pixels = synthetic.make_pixels(cfg, Bn, 0, device=dev, dtype=torch.bfloat16)
    ids = torch.stack([synthetic.make_prompt(cfg, a.prompt_tokens, 1, i) for i in range(Bn)], 0).to(dev)
    e, _, m = model._embed(ids, {"image": [pixels[i] for i in range(Bn)]})
    S = e.shape[1]
    max_new = a.steps + a.warmup + 2
    t0 = time.perf_counter()
    out = llm.generate(inputs_embeds=e, attention_mask=m, max_new_tokens=max_new, eos_token_id=-1)      # builds the session, captures the graph
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    st = llm._bdecode
    assert st is not None and st.graph is not None and out.shape == (Bn, max_new)
    st.pos.fill_(S); st.n_out.zero_()

    def one():
        check(lib.vila_graph_launch(st.graph, st.stream.cuda_stream), "graph_launch")
    with torch.cuda.stream(st.stream):
        for _ in range(a.warmup):
            one()
        elapsed = timed_region(dist, dev, a.steps, one, torch.cuda.synchronize)
    if rank == 0:
        step_s = elapsed / a.steps
        byt = decode_bytes_per_token(cfg, S + a.warmup + a.steps // 2) + (Bn - 1) * 2 * cfg.llm.kv_size * 2 * cfg.llm.num_hidden_layers * (S + a.warmup + a.steps // 2)
        print(json.dumps({
            "metric": f"batched decode tokens/sec, NVILA-8B, {Bn} sequences per weight pass", "value": round(world * Bn * a.steps / elapsed, 2), "unit": "tokens/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg.name} bf16, {Bn} x (1x448^2 image + {a.prompt_tokens}-token prompt, S={S}), greedy, one hipGraph replay per step for the whole batch",
                       "parallelism": f"replicas x{world}" if world > 1 else "single GPU", "batch": Bn},
            "roofline": {"bound": "hbm", "achieved": round(byt / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byt / step_s / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None, "note": "whole step: weight bytes once + every row's KV cache"},
            "first_generate_s": round(gen_s, 3)}))
    if dist is not None:
        dist.destroy_process_group()


def decode_main(a, rank, world, dev, dist):
    from vila_amd import _lib, configs, ops, synthetic
    from vila_amd.vlm import build_model
    if a.batch > 1:
        return batch_decode_main(a, rank, world, dev, dist)
    lib = _lib.load()
    cfg = configs.nvila_8b() if a.config == "nvila_8b" else configs.nvila_lite_3b() if a.config == "nvila_lite_3b" else configs.reduced_8b(3, 4)
    n_tiles, media_cfg = 1, {}
    if a.dynamic_s2:
        cfg = configs.nvila_8b_s2()
        n_tiles, media_cfg = 14, {"image": {"block_sizes": [(3, 3)]}}
    model = build_model(cfg, seed=0, device=dev)
    if a.w8_vit:
        model.vision_tower.quantize_w8()
    llm = model.llm
    pixels = synthetic.make_pixels(cfg, n_tiles, 0, device=dev, dtype=torch.bfloat16)
    ids = synthetic.make_prompt(cfg, a.prompt_tokens, 1, 0)[None].to(dev)
    S = (cfg.tokens_per_tile * 9 if a.dynamic_s2 else cfg.tokens_per_tile) + 1 + a.prompt_tokens
    max_new = a.steps + a.warmup + 2
    cache = llm.new_cache(((S + max_new + 255) // 256) * 256)

    # ---- TTFT: pixels + ids resident on the device -> first token id on the host ----
    def ttft_once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e, _, m = model._embed(ids, {"image": [pixels[i] for i in range(n_tiles)]}, media_cfg)
        pos = torch.arange(S, device=dev, dtype=torch.int32)
        r = llm.prefill_packed(e[0], pos, None, S, cache=cache, last_rows=torch.full((1,), S - 1, device=dev, dtype=torch.int32))   # (a fill kernel: no host copy)
        first = int(ops.argmax(r.last_logits[0]))
        return time.perf_counter() - t0, first, e

    ttft_once()
    tt = []
    for _ in range(5):
        t, first, e = ttft_once()
        tt.append(t)
    ttft = statistics.median(tt)
    # the encoder part alone (ViT + projector), HIP-event timed on the current stream
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bs = media_cfg.get("image", {}).get("block_sizes")
    torch.cuda.synchronize()
    ev_a.record()
    for _ in range(3):
        model.encode_images(pixels, block_sizes=bs)
    ev_b.record()
    torch.cuda.synchronize()
    encode_ms = ev_a.elapsed_time(ev_b) / 3
    prefill_flops = vit_flops(cfg, n_tiles) + projector_flops(cfg, n_tiles) + llm_prefill_flops(cfg, S)
    pre_traffic = None          # L2-fill bytes of the prefill's dominant kernel (fused gate/up GEMM) from the committed --pmc passes of THIS command
    tj = next((p_ for p_ in (os.path.join(ROOT, "profiles", n_) for n_ in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json")) if os.path.exists(p_)), None)
    if a.config == "nvila_8b" and not a.dynamic_s2 and tj is not None:
        with open(tj) as f:
            pre_traffic = json.load(f).get("prefill_gateup")
    prefill = {"ttft_ms": round(ttft * 1e3, 3), "encode_images_ms": round(encode_ms, 3), "tflop": round(prefill_flops / 1e12, 3),
               "roofline": {"bound": "mfma", "achieved": round(prefill_flops / ttft / 1e12, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                            "frac": round(prefill_flops / ttft / (MFMA_PEAK_TF * 1e12), 4), "traffic": pre_traffic,
                            "note": "algorithmic FLOPs of ViT (26 layers) + projector + LLM prefill + last-row lm_head (SURVEY §8d) / host-observed TTFT"}}

    if a.w4:
        w4 = llm.quantize_w4(keep_logical=False)     # decode now streams int4 weights; the prefill above used bf16
    # ---- decode: capture one step in a hipGraph, replay ----
    st = llm._decode_session(cache, max_new)
    stream = st.stream

    def reset_state():
        st.pos.fill_(S); st.n_out.zero_(); st.token.fill_(first)
    reset_state()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        llm.decode_step(cache, st)                    # warm (outside capture)
        stream.synchronize()
        reset_state()
        _lib.check(lib.vila_graph_begin(stream.cuda_stream), "graph_begin")
        llm.decode_step(cache, st)
        g = C.c_void_p()
        _lib.check(lib.vila_graph_end(stream.cuda_stream, C.byref(g)), "graph_end")

        def one_step():
            if a.eager_decode:
                llm.decode_step(cache, st)
            else:
                _lib.check(lib.vila_graph_launch(g, stream.cuda_stream), "graph_launch")
        for _ in range(a.warmup):
            one_step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        for _ in range(a.steps):
            one_step()
        ev1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_generated = int(st.n_out.item())
    assert n_generated == a.warmup + a.steps, (n_generated, a.warmup, a.steps)
    chain_err = int(lib.vila_llm_decode_chain_error(st.ws.data_ptr(), stream.cuda_stream))
    assert chain_err == 0, "chained decode step: a bounded wait gave up — the timed tokens are invalid"
    chained = os.environ.get("VILA_DECODE_CHAIN", "0") == "1" and not a.w4 and cache.max_ctx <= 2048
    ctx_mid = S + a.warmup + a.steps // 2
    step_bytes = decode_bytes_per_token(cfg, ctx_mid, a.w4)
    step_s = elapsed / a.steps
    launches = int(lib.vila_llm_decode_launches(C.byref(llm._struct().shape), cache.max_ctx))

    # ---- sustained replay (>= 6 s of back-to-back tokens; context rewound whenever the cache fills) ----
    sustained = None
    if not a.no_sustain and not a.eager_decode:
        room = min(cache.max_ctx - S - 2, max_new - 2)
        n_sus = max(int(6.5 / step_s), 64)           # >= 6 s: long enough for a coarse utilisation sampler around the run to see the card busy
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        with torch.cuda.stream(stream):
            while done < n_sus:
                reset_state()
                chunk = min(room, n_sus - done)
                for _ in range(chunk):
                    one_step()
                done += chunk
        torch.cuda.synchronize()
        sus_s = time.perf_counter() - t0
        sustained = {"seconds": round(sus_s, 3), "tokens": done, "tokens_per_s": round(done / sus_s, 2),
                     "note": "corroboration run after the timed region (same graph, context rewound when the cache fills); not `value`"}

    # ---- roofline of the dominant kernel: gemv_kernel<1> (fused RMSNorm + gate/up GEMV + SiLU*mul = 54% of the decode
    # bytes).  Timed live with HIP events on the launch stream, cycling over the 28 layers' weights so the 256 MiB
    # Infinity Cache cannot serve re-reads (7.6 GB working set). ----
    c = cfg.llm
    x = torch.randn(c.hidden_size, device=dev).to(torch.bfloat16)
    act = torch.empty(c.intermediate_size, device=dev, dtype=torch.bfloat16)
    layers = [getattr(llm.model.layers, str(i)) for i in range(c.num_hidden_layers)]
    reps = 4

    def gateup_all():
        if a.w4:
            for i in range(len(layers)):
                L = w4.layers[i]
                _lib.check(lib.vila_gemv_w4_bf16(x.data_ptr(), layers[i].post_attention_layernorm.weight.data_ptr(), c.rms_norm_eps,
                                                 L.gateup_q, L.gateup_sz, None, None, act.data_ptr(),
                                                 c.intermediate_size, c.hidden_size, 1, stream.cuda_stream), "gemv_w4")
            return
        for l in layers:
            _lib.check(lib.vila_gemv_bf16(x.data_ptr(), l.post_attention_layernorm.weight.data_ptr(), c.rms_norm_eps,
                                          l.mlp.gate_proj.weight.data_ptr(), l.mlp.up_proj.weight.data_ptr(), None, None,
                                          act.data_ptr(), None, c.intermediate_size, c.hidden_size, 1, stream.cuda_stream), "gemv")
    with torch.cuda.stream(stream):
        gateup_all()
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(stream)
        for _ in range(reps):
            gateup_all()
        k1.record(stream)
    torch.cuda.synchronize()
    n_launch = reps * len(layers)
    kern_s = k0.elapsed_time(k1) * 1e-3 / n_launch           # includes the ~1.5 us launch boundary between kernels
    kern_bytes = 2 * c.intermediate_size * c.hidden_size * 2 + c.hidden_size * 2 * 2 + c.intermediate_size * 2
    if a.w4:
        kern_bytes = 2 * c.intermediate_size * c.hidden_size * 17 // 32 + c.hidden_size * 2 * 2 + c.intermediate_size * 2
    achieved = kern_bytes / kern_s / 1e9
    # HBM bytes per launch from the PMC counters cannot be collected inside this process: they come from the separate
    # rocprofv3 --pmc passes of THIS command (tools/pmc.sh), corrected as MI355X_MICROARCH.md prescribes, committed under profiles/
    traffic, traffic_src = None, None
    for tj_name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tj = os.path.join(ROOT, "profiles", tj_name)
        if a.config == "nvila_8b" and not a.w4 and os.path.exists(tj):
            with open(tj) as f:
                tdata = json.load(f)
            if tdata.get("algorithmic_bytes_per_launch") == kern_bytes:
                traffic, traffic_src = tdata["traffic_bytes_per_launch"], f"profiles/{tj_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 on gfx950)"
                break
    roofline = {"bound": "hbm", "kernel": ("gemv_w4_kernel<1>" if a.w4 else "gemv_kernel<1>") + " (RMSNorm + gate/up GEMV + SiLU*mul)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": kern_bytes, "avg_launch_us": round(kern_s * 1e6, 2),
                "whole_step": {"bytes_per_token": step_bytes, "achieved": round(step_bytes / step_s / 1e9, 1),
                               "frac": round(step_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4),
                               "gpu_ms_per_step_hip_events": round(ev0.elapsed_time(ev1) / a.steps, 4)}}

    def line(sft, cpu):
        value = world * a.steps / elapsed
        return {
            "metric": "decode tokens/sec + TTFT, NVILA-Lite-3B-shaped 1-image prompt (BASELINE configs[0] on the GPU)" if a.config == "nvila_lite_3b" else
                      "decode tokens/sec + TTFT, NVILA-8B 1-image prompt",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / A100_DECODE_TOKS, 3) if a.config == "nvila_8b" else None,
            "vs_baseline_note": "value / 82.1 tok/s (NVILA-8B FP16 on ONE A100, TinyChat backend, README.md:65) — other hardware and fp16; no MI355X number is published",
            "dtype": "w4a16 (int4 group-128 weights, bf16 activations, fp32 accumulate)" if a.w4 else "bf16",
            "data": f"synthetic (seeded random weights at {cfg.name} shapes; U(-1,1) pixels; random prompt ids)",
            "ttft_ms": round(ttft * 1e3, 3),
            "ttft_note": "median of 5: pixels+ids on device -> ViT(26 layers) + mm_projector + splice + 769-token prefill + argmax -> id on host",
            "config": {"workload": f"{cfg.name} {'W4A16 decode / bf16 prefill' if a.w4 else 'bf16'}{' + W8A8 vision tower' if a.w8_vit else ''}, 1x448^2 image + {a.prompt_tokens}-token prompt (S={S}), batch 1, greedy decode, "
                                   f"context {S + a.warmup}..{S + a.warmup + a.steps}", "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                       "decode": f"hipGraph replay of {launches} launches/token" + (", kernels chained over two streams (each streams its weights under its predecessor's tail)" if chained else ""),
                       **({"parity": "unpinned against the reference (its quantised backend, TinyChat / llm-awq, is external: no reference-held vectors); "
                                     "pinned against the dequantise-then-fp32 oracle of the same quantised weights"} if (a.w4 or a.w8_vit) else {})},
            "roofline": roofline,
            "prefill": prefill,
            "sft": sft,
            "sustained": sustained,
            "cpu_baseline": cpu,
        }

    # ---- bounded SFT sub-measurement: 1 warm + 2 timed steps of the configs[2] per-GPU workload (needs ~150 GB of the 288 GB).  With
    # world > 1 (the driver's `bench.py --gpus N`) EVERY rank runs it with the process group: the per-layer gradient buckets go through RCCL's
    # SUM all-reduce under the backward (replaces scripts/zero3.json + transformer_normalize_monkey_patch.py:242-263), so the
    # scaling runs exercise the 16.1 GB exchange without any extra flag.  The decode line above is complete at this point: with world > 1 a
    # deadline guards it, so a rank that dies inside a step (the others would sit in its all-reduce until RCCL's watchdog ends the job
    # without any output) costs the `sft` block, not the measurement. ----
    sft = None
    if not a.no_sft and not a.w4 and not a.dynamic_s2 and not a.w8_vit and a.config == "nvila_8b":
        guard = None
        if world > 1:
            secs = float(os.environ.get("VILA_BENCH_SFT_DEADLINE_S", "300"))
            guard = Deadline(secs, (lambda: json.dumps(line({"error": f"side measurement exceeded its {secs:.0f}-s deadline (a rank hung or died inside a "
                                                                       "data-parallel step); the decode line stands", "world": world}, None))) if rank == 0 else None)
        sft = sft_side_measurement(model, cfg, a, rank, world, dev, dist)
        if guard is not None and not guard.cancel():
            return                                   # the deadline is already printing / exiting: leave stdout to it

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(cfg, S, os.cpu_count() or 1)
    print(json.dumps(line(sft, cpu)))
    if dist is not None:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a, argv))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; the launcher's world size is used", file=sys.stderr)
    if a.selftest:
        return selftest_main(a, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("VILA_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on a 1-GPU box
        dist = init_dist("nccl", dev)
    if a.mode == "video":
        return video_main(a, rank, dev)
    if a.mode == "sft":
        if a.steps == 128 and a.warmup == 16:
            a.steps, a.warmup = 3, 1
        return sft_main(a, rank, world, dev, dist)
    return decode_main(a, rank, world, dev, dist)


if __name__ == "__main__":
    main()
