#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: decode tokens/s (+ TTFT), NVILA-8B bf16, 1 x 448^2 image + 512-token prompt.

    python bench.py --gpus N --steps K --warmup W

A "step" = one greedy decode token through all 28 decoder layers + lm_head + argmax (one hipGraph replay).  The
prefill (ViT + projector + splice + 769-token LLM prefill) runs before the timed region and is reported as TTFT.
N > 1: one process per GPU (torchrun), independent replicas (inference has no exchange step: SURVEY.md §8e),
value = total tokens/s over all ranks, time = max over ranks.  Inputs and weights are resident in HBM before timing.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
A100_DECODE_TOKS = 82.1   # NVILA-8B FP16 decode tok/s on A100 (README.md:65) — other hardware, fp16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt-tokens", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-decode", action="store_true", help="experiment: launch the 171 kernels per token eagerly instead of replaying a hipGraph")
    ap.add_argument("--config", default="nvila_8b", choices=["nvila_8b", "reduced"])
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--mode", default="decode", choices=["decode", "sft", "video"],
                    help="decode = BASELINE.json metric (default); sft = one data-parallel SFT step (BASELINE configs[2])")
    ap.add_argument("--micro-batch", type=int, default=4)
    ap.add_argument("--w4", action="store_true", help="W4A16 decode (int4 group-128 decoder projections; BASELINE configs[4], SURVEY 8f row 3)")
    ap.add_argument("--dynamic-s2", action="store_true", help="full NVILA-8B recipe: 14 tiles (448/896/1344) -> 2304 image tokens (SURVEY 8f row 1)")
    return ap.parse_args()


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


def cpu_baseline(cfg, n_prompt: int, threads: int):
    """Oracle ("port") timed on the host cores on a bounded sample: NVILA-8B widths with 2 of 28 decoder layers (fp32),
    S-token prefill then 4 decode tokens; per-token time is scaled to 28 layers (layer part) + the measured lm_head part."""
    from oracle import vila_oracle as O
    from vila_amd import configs, synthetic
    torch.set_num_threads(threads)
    L = 2
    c = configs.reduced_8b(layers_v=1, layers_l=L, vocab=cfg.llm.vocab_size)
    specs = [s for s in synthetic.llm_specs(c) if "embed_tokens" not in s[0]]
    g = torch.Generator().manual_seed(0)
    w = {n: torch.randn(shape, generator=g) * 0.02 if k in ("w", "h") else torch.ones(shape) for n, shape, k in specs}
    w["llm.model.embed_tokens.weight"] = w["llm.lm_head.weight"]     # timing only: share the table to bound host RAM
    e = torch.randn(1, n_prompt, c.llm.hidden_size, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        logits, past = O.qwen2_forward(e, w, c.llm)
        t_prefill = time.perf_counter() - t0
        x = torch.randn(1, 1, c.llm.hidden_size, generator=g)
        n_tok = 4
        t0 = time.perf_counter()
        for _ in range(n_tok):
            logits, past = O.qwen2_forward(x, w, c.llm, past=past)
        t_tok = (time.perf_counter() - t0) / n_tok
        h = torch.randn(1, 1, c.llm.hidden_size)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.linear(h, w["llm.lm_head.weight"])
        t_head = (time.perf_counter() - t0) / 3
    t_full = (t_tok - t_head) * (cfg.llm.num_hidden_layers / L) + t_head
    return {"value": round(1.0 / t_full, 3), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"oracle/vila_oracle.py qwen2_forward fp32, NVILA-8B widths, {L} of {cfg.llm.num_hidden_layers} decoder layers "
                      f"+ full lm_head, {n_prompt}-token prefill ({t_prefill:.2f}s) then {n_tok} decode tokens "
                      f"({t_tok*1e3:.0f} ms/token measured; layer part scaled x{cfg.llm.num_hidden_layers // L})",
            "prefill_s_sample": round(t_prefill, 3)}


def sft_main(a, rank, local, world, dev, dist):
    """BASELINE configs[2]: NVILA-8B SFT step, per-GPU micro-batch of b samples (1 x 448^2 image + 512 text tokens, S = 769,
    packed), labels on the last 256 text positions, all 8.06 B params trainable, AdamW lr 2e-5; weak scaling over ranks."""
    from vila_amd import configs, synthetic
    from vila_amd.train import SFTTrainer
    from vila_amd.vlm import build_model
    cfg = configs.nvila_8b() if a.config == "nvila_8b" else configs.reduced_8b(3, 4)
    model = build_model(cfg, seed=0, device=dev)
    tr = SFTTrainer(model, lr=2e-5, weight_decay=0.0)
    b = a.micro_batch
    pixels = synthetic.make_pixels(cfg, b, rank, device=dev, dtype=torch.bfloat16)
    ids = torch.stack([synthetic.make_prompt(cfg, a.prompt_tokens, 1, 10 * rank + i) for i in range(b)], 0)
    labels = ids.clone()
    labels[:, : 1 + a.prompt_tokens - 256] = -100
    S = cfg.tokens_per_tile + 1 + a.prompt_tokens
    images = [pixels[i] for i in range(b)]
    for _ in range(a.warmup):
        loss = tr.step(ids, images, labels)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(ids, images, labels)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        step_s = elapsed / a.steps
        flops = 35.8e12 * b              # fwd+bwd per 769-token sample, no recompute
        print(json.dumps({
            "metric": "SFT step throughput, NVILA-8B, packed 1x448^2 image + 512-token samples", "value": round(world * b * S / step_s, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(step_s * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (fp32 master/AdamW state)",
            "data": "synthetic", "loss": round(float(loss), 4),
            "config": {"workload": f"{cfg.name} SFT step, micro-batch {b} x S={S} packed, all params trainable, AdamW", "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(flops / step_s / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flops / step_s / 2.5e15, 4), "traffic": None,
                         "note": "whole step incl. optimizer and transposes; 35.8 TFLOP per sample (SURVEY §8d)"}}))
    if dist is not None:
        dist.destroy_process_group()


def video_main(a, rank, dev):
    """BASELINE configs[3]: NVILA-Video-8B-style prefill, `--frames` 448^2 frames as per-frame <image> tokens
    (llava/utils/media.py:114-119: 64 x 257 = 16448 media tokens + 32 text), mlp_downsample_2x2_fix projector."""
    from vila_amd import configs, ops, synthetic
    from vila_amd.vlm import build_model
    cfg = configs.nvila_8b()
    cfg.mm_projector_type = "mlp_downsample_2x2_fix"
    model = build_model(cfg, seed=0, device=dev)
    F_ = a.frames
    pixels = synthetic.make_pixels(cfg, F_, 0, device=dev, dtype=torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 32, F_, 0)[None].to(dev)
    S = F_ * (cfg.tokens_per_tile + 1) + 32
    cache = model.llm.new_cache(((S + 64 + 255) // 256) * 256)
    frames = [pixels[i] for i in range(F_)]

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e, _, _ = model._embed(ids, {"image": frames})
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r = model.llm.prefill_packed(e[0], torch.arange(S, device=dev, dtype=torch.int32), None, S, cache=cache,
                                     last_rows=torch.tensor([S - 1], device=dev, dtype=torch.int32))
        first = int(ops.argmax(r.last_logits[0]))
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, first
    once()
    ts = [once() for _ in range(max(a.steps if a.steps != 128 else 3, 1))]
    enc = statistics.median(t[0] for t in ts)
    pre = statistics.median(t[1] for t in ts)
    flops = F_ * 0.953e12 + 28 * (4 * S * 3584 ** 2 + 4 * S * 3584 * 512 + 6 * S * 3584 * 18944 + 2 * S * S * 3584) + 2 * 3584 * 152064
    print(json.dumps({"metric": "TTFT, NVILA-Video-8B-style prefill", "value": round((enc + pre) * 1e3, 2), "unit": "ms", "n_gpus": 1,
                      "higher_is_better": False, "dtype": "bf16", "data": "synthetic",
                      "config": {"workload": f"{F_} frames x 448^2 -> {S} tokens (per-frame <image> tokens), batch 1"},
                      "encode_ms": round(enc * 1e3, 2), "llm_prefill_ms": round(pre * 1e3, 2),
                      "roofline": {"bound": "mfma", "achieved": round(flops / (enc + pre) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": round(flops / (enc + pre) / 2.5e15, 4), "traffic": None},
                      "reference_note": "README.md:84: 0.7190 s on A100 FP16 (TinyChat, 64 frames, pooled tokens) - other hardware"}))


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("VILA_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist_
        dist = dist_
        # RCCL prints a version banner on STDOUT when the communicator is created; stdout must carry the one JSON line only,
        # so fd 1 points at stderr while the process group and its first collective come up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    if a.mode == "video":
        return video_main(a, rank, dev)
    if a.mode == "sft":
        if a.steps == 128 and a.warmup == 16:
            a.steps, a.warmup = 3, 1
        return sft_main(a, rank, local, world, dev, dist)
    from vila_amd import _lib, configs, ops, synthetic
    from vila_amd.vlm import build_model
    lib = _lib.load()
    cfg = configs.nvila_8b() if a.config == "nvila_8b" else configs.reduced_8b(3, 4)
    n_tiles, media_cfg = 1, {}
    if a.dynamic_s2:
        cfg = configs.nvila_8b_s2()
        n_tiles, media_cfg = 14, {"image": {"block_sizes": [(3, 3)]}}
    model = build_model(cfg, seed=0, device=dev)
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
        r = llm.prefill_packed(e[0], pos, None, S, cache=cache, last_rows=torch.tensor([S - 1], device=dev, dtype=torch.int32))
        first = int(ops.argmax(r.last_logits[0]))
        return time.perf_counter() - t0, first, e

    ttft_once()
    tt = []
    for _ in range(5):
        t, first, e = ttft_once()
        tt.append(t)
    ttft = statistics.median(tt)

    if a.w4:
        w4 = llm.quantize_w4(keep_logical=False)     # decode now streams int4 weights; the prefill above used bf16
    # ---- decode: capture one step in a hipGraph, replay ----
    st = llm._decode_session(cache, max_new)
    stream = st.stream
    st.pos.fill_(S); st.n_out.zero_(); st.token.fill_(first)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        llm.decode_step(cache, st)                    # warm (outside capture)
        stream.synchronize()
        st.pos.fill_(S); st.n_out.zero_(); st.token.fill_(first)
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
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
    ctx_mid = S + a.warmup + a.steps // 2
    step_bytes = decode_bytes_per_token(cfg, ctx_mid, a.w4)
    step_s = elapsed / a.steps

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
    tj = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if a.config == "nvila_8b" and not a.w4 and os.path.exists(tj):
        with open(tj) as f:
            tdata = json.load(f)
        if tdata.get("algorithmic_bytes_per_launch") == kern_bytes:
            traffic, traffic_src = tdata["traffic_bytes_per_launch"], "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 on gfx950)"
    roofline = {"bound": "hbm", "kernel": ("gemv_w4_kernel<1>" if a.w4 else "gemv_kernel<1>") + " (RMSNorm + gate/up GEMV + SiLU*mul)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": kern_bytes, "avg_launch_us": round(kern_s * 1e6, 2),
                "whole_step": {"bytes_per_token": step_bytes, "achieved": round(step_bytes / step_s / 1e9, 1),
                               "frac": round(step_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4),
                               "gpu_ms_per_step_hip_events": round(ev0.elapsed_time(ev1) / a.steps, 4)}}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(cfg, S, os.cpu_count() or 1)
    value = world * a.steps / elapsed
    out = {
        "metric": "decode tokens/sec + TTFT, NVILA-8B 1-image prompt",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": round(value / A100_DECODE_TOKS, 3) if a.config == "nvila_8b" else None,
        "vs_baseline_note": "value / 82.1 tok/s (NVILA-8B FP16 on ONE A100, TinyChat backend, README.md:65) — other hardware and fp16; no MI355X number is published",
        "dtype": "w4a16 (int4 group-128 weights, bf16 activations, fp32 accumulate)" if a.w4 else "bf16", "data": "synthetic (seeded random weights at NVILA-8B shapes; U(-1,1) pixels; random prompt ids)",
        "ttft_ms": round(ttft * 1e3, 3),
        "ttft_note": "median of 5: pixels+ids on device -> ViT(26 layers) + mm_projector + splice + 769-token prefill + argmax -> id on host",
        "config": {"workload": f"{cfg.name} {'W4A16 decode / bf16 prefill' if a.w4 else 'bf16'}, 1x448^2 image + {a.prompt_tokens}-token prompt (S={S}), batch 1, greedy decode, "
                               f"context {S + a.warmup}..{S + a.warmup + a.steps}", "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                   "decode": f"hipGraph replay of {c.num_hidden_layers * (5 if cache.c.max_ctx <= 2048 else 6) + 5} launches/token "
                             f"({5 if cache.c.max_ctx <= 2048 else 6} per layer + embed/lm_head/argmax x2/advance)"},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
