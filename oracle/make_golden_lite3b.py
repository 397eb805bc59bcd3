"""Full-depth fixtures for BASELINE configs[0] — NVILA-Lite-3B-shaped widths (SigLIP 26 layers -> 3x3 projector -> 36-layer Qwen2.5-3B-shaped
decoder, tied head), 1 x 448^2 image + 32-token prompt (S = 154), 8 recorded steps — TEST INFRASTRUCTURE.

Writes TWO files from the same seeded synthetic weights:
  * tests/golden/nvila_lite3b_full_depth_ref.npz — REFERENCE-EXECUTED: the reference's modeling_siglip.py and base_projector.py (loaded by file
    path, eager attention, fp32) and HF Qwen2ForCausalLM (fp32, eager, KV cache) — see make_golden_full_ref.py
  * tests/golden/nvila_lite3b_full_depth.npz — ORACLE-EXECUTED (oracle/vila_oracle.py)
tests/test_oracle_golden.py compares the two (CPU); tests/test_gpu_full_depth.py holds the HIP path to the reference-executed one.

Round 4: the recorded steps are TEACHER-FORCED WITH A RANDOM ID SEQUENCE (`forced_ids`) instead of following greedy
decoding into a fixed point, and the tail of the tied head's row norms is searched for distinct argmax tokens with margins inside 4..20x the
expected bf16 logit error.  The tied head IS the embedding table, so the rows of the prompt's, the forced and the "\n" token are pinned to scale 1
(`cfg.lm_head_tail_unit_rows`, stored in the fixture): the hidden states then do not depend on the tail and the search costs one [8, H] x [H, V]
product per candidate, like the 8B one (a first version without the pinning needed a decoder pass per candidate and found one attractor token
on 6-8 of 8 steps for every tail it could afford to try).  The free-running greedy ids of the chosen weights are stored too.

    python oracle/make_golden_lite3b.py        # ~8 min on 8 cores, ~25 GB RSS; needs /root/reference
    VILA_TAIL=2,10,0 python oracle/make_golden_lite3b.py     # skip the search: tail exponent, cap, seed
"""
from __future__ import annotations

import copy
import gc
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import make_golden as G                     # noqa: E402
from oracle import vila_oracle as O                     # noqa: E402
from oracle.make_golden_full import LazyBf16Weights, N_NEW, TOPK, forced_sequence, load_calibration, score_logits, score_measured      # noqa: E402
from oracle.make_golden_full_ref import build_hf_llm_streaming        # noqa: E402
from vila_amd import configs, synthetic                 # noqa: E402

SEED = 13
N_TEXT = 32
REL_ERR = 0.05           # fallback error model (no calibration file): about twice the 8B model's (36 layers, hidden 2048)
TAILS = [(a, 4.0) for a in (2.0, 2.5, 3.0, 4.0, 5.0, 6.0, 8.0)]
SEEDS = range(256)
KEYS = ("llm.model.layers.0.mlp.gate_proj.weight", "llm.model.layers.35.self_attn.q_proj.bias", "llm.model.embed_tokens.weight",
        "vision_tower.vision_tower.vision_model.encoder.layers.25.mlp.fc1.weight", "mm_projector.layers.2.weight")
OUT = os.path.join(ROOT, "tests", "golden", "nvila_lite3b_full_depth{}.npz")
EMB = "llm.model.embed_tokens.weight"


def pack(w, cfg, px, ids, forced, feats, proj, e, gen, lg, extra):
    gen, gmargin = gen
    top = lg.topk(TOPK, -1)
    n_img = proj.shape[1]
    out = {f"fp_w{i}": w[k].reshape(-1)[:16].numpy().copy() for i, k in enumerate(KEYS)}
    out.update({"fp_pixels": px.reshape(-1)[:16].numpy().copy(), "seed": np.int64(SEED), "input_ids": ids.numpy(), "forced_ids": forced.numpy(),
                "lm_head_tail": np.float32(cfg.lm_head_tail), "lm_head_tail_seed": np.int64(cfg.lm_head_tail_seed), "lm_head_tail_max": np.float32(cfg.lm_head_tail_max),
                "lm_head_tail_unit_rows": np.asarray(cfg.lm_head_tail_unit_rows, dtype=np.int64),
                "tf_argmax_ids": lg.argmax(-1).numpy().astype(np.int64), "greedy_ids": np.asarray(gen, dtype=np.int64), "greedy_margins": np.asarray(gmargin, dtype=np.float32),
                "top_ids": top.indices.numpy().astype(np.int32), "top_vals": top.values.numpy().astype(np.float32),
                "logit_absmax": lg.abs().amax(-1).numpy().astype(np.float32), "logit_norm": lg.norm(dim=-1).numpy().astype(np.float32),
                "vit_rows": feats[0, [0, 511, 1023], :256].numpy().astype(np.float32), "vit_norm": np.float32(feats.norm()),
                "proj_rows": proj[0, [0, n_img // 2, n_img - 1], :256].numpy().astype(np.float32), "proj_norm": np.float32(proj.norm()),
                "embed_rows": e[0, [0, n_img - 1, n_img, n_img + 1, e.shape[1] - 1], :256].numpy().astype(np.float32), "embed_norm": np.float32(e.norm())})
    out.update(extra)
    return out


def oracle_hidden(cfg, w, proj, ids, forced):
    """Final-norm hidden state [N_NEW, H] of the prefill's last row and of the teacher-forced steps (what the head multiplies)."""
    lc = cfg.llm
    end = O.embed_tokens(torch.tensor([cfg.newline_token_id]), w)
    e, _, _ = O.embed_splice(ids[None], [torch.cat([proj[0], end], 0)], w, cfg)
    norm_w = w["llm.model.norm.weight"]
    _, past, hs = O.qwen2_forward(e, w, lc, return_hidden=True)
    xn = [O.rms_norm(hs[-1][0, -1], norm_w, lc.rms_norm_eps)]
    for t in range(N_NEW - 1):
        _, past, hs = O.qwen2_forward(O.embed_tokens(forced[t].view(1, 1), w), w, lc, past=past, return_hidden=True)
        xn.append(O.rms_norm(hs[-1][0, -1], norm_w, lc.rms_norm_eps))
    return torch.stack(xn)


def oracle_steps(cfg, w, proj, ids, forced, greedy: bool):
    """Spliced embeds, teacher-forced logits [N_NEW, V] (and the free-running greedy ids) for the CURRENT cfg tail."""
    lc = cfg.llm
    end = O.embed_tokens(torch.tensor([cfg.newline_token_id]), w)
    e, _, _ = O.embed_splice(ids[None], [torch.cat([proj[0], end], 0)], w, cfg)
    logits, past0 = O.qwen2_forward(e, w, lc)
    last, past, steps = logits[0, -1], past0, []
    for t in range(N_NEW):
        steps.append(last.clone())
        if t + 1 == N_NEW:
            break
        logits, past = O.qwen2_forward(O.embed_tokens(forced[t].view(1, 1), w), w, lc, past=past)
        last = logits[0, -1]
    lg = torch.stack(steps).float()
    gen, gmargin = [], []
    if greedy:
        past, last = past0, lg[0]
        for t in range(N_NEW):
            nxt = int(last.argmax())
            gen.append(nxt)
            t2v = last.topk(2).values
            gmargin.append(float(t2v[0] - t2v[1]))
            if t + 1 == N_NEW:
                break
            logits, past = O.qwen2_forward(O.embed_tokens(torch.tensor([[nxt]]), w), w, lc, past=past)
            last = logits[0, -1]
    return e, lg, (gen, gmargin)


def main():
    torch.manual_seed(0)
    cfg = configs.nvila_lite_3b()
    w = LazyBf16Weights(cfg, SEED)
    px = synthetic.make_pixels(cfg, 1, SEED).to(torch.bfloat16).float()
    ids = synthetic.make_prompt(cfg, N_TEXT, 1, SEED)
    forced = forced_sequence(cfg, N_NEW, SEED)
    t0 = time.time()
    with torch.no_grad():
        # ---- oracle ----
        feats_o = O.vision_tower_forward(px, w, cfg.vision)
        proj_o = O.projector_forward(feats_o, w, cfg.mm_projector_type)
        print(f"oracle tower + projector {time.time() - t0:.0f}s", flush=True)
        # the rows the hidden states can see are pinned: prompt text ids, forced ids, the image block's "\n"
        used = sorted({int(t) for t in ids.tolist() if t != cfg.image_token_id} | {int(t) for t in forced.tolist()} | {int(cfg.newline_token_id)})
        cfg.lm_head_tail_unit_rows = tuple(used)
        fixed = os.environ.get("VILA_TAIL")
        if fixed:
            a, m, s = (float(x) for x in fixed.split(","))
            best = (None, a, m, int(s))
        else:
            cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = 3.0, 0, 3.0            # any tail: the pinned rows make XN independent of it
            w.store.pop(EMB, None)
            t1 = time.time()
            XN = oracle_hidden(cfg, w, proj_o, ids, forced)
            print(f"teacher-forced hidden states {time.time() - t1:.0f}s; |xn| {[round(float(v), 1) for v in XN.norm(dim=-1)]}", flush=True)
            shape, kind = w.specs[EMB]
            keep_tail = cfg.lm_head_tail
            cfg.lm_head_tail = 0.0
            base = synthetic._draw(EMB, shape, kind, cfg, w.seed, "cpu")           # row directions x init_std
            cfg.lm_head_tail = keep_tail
            L0 = XN @ base.t()
            XN_gpu = load_calibration("nvila_lite3b", ids, forced)
            L0g = XN_gpu @ base.t() if XN_gpu is not None else None
            print("search scored against " + ("the HIP path's measured hidden states (oracle/calib)" if XN_gpu is not None else f"the error model REL_ERR = {REL_ERR}"), flush=True)
            del base
            xn_norm = XN.norm(dim=-1)
            best = None
            for a, m in TAILS:
                for s in SEEDS:
                    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = float(a), int(s), float(m)
                    scale = synthetic.lm_head_row_scale(EMB, cfg.llm.vocab_size, cfg)
                    if L0g is not None:
                        sc, am, ratio, margin, err = score_measured(L0 * scale[None], L0g * scale[None])
                    else:
                        sc, am, ratio, margin, err = score_logits(cfg, L0 * scale[None], xn_norm, scale, row_std=cfg.init_std, rel_err=REL_ERR)
                        sc = (sc[1], sc[0], sc[2])                   # distinct in-band winners first (a deep random decoder's hidden states are alike)
                    if best is None or sc > best[0]:
                        best = (sc, a, m, s)
                        print(f"  tail a={a} max={m} seed={s}: score {sc[:3]}, ids {am.tolist()}, "
                              f"margin/err {[round(float(r), 1) for r in ratio]}", flush=True)
        _, a, m, s = best
        cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = float(a), int(s), float(m)
        w.store.pop(EMB, None)
        e_o, lg_o, gen_o = oracle_steps(cfg, w, proj_o, ids, forced, greedy=True)
        print(f"chosen tail a={a} max={m} seed={s}: S = {e_o.shape[1]}, teacher-forced argmax {lg_o.argmax(-1).tolist()}, greedy {gen_o[0]} ({time.time() - t0:.0f}s)", flush=True)
        np.savez_compressed(OUT.format(""), **pack(w, cfg, px, ids, forced, feats_o, proj_o, e_o, gen_o, lg_o, {}))
        # ---- reference ----
        t1 = time.time()
        vis_w = {k: w[k] for k in w.specs if k.startswith("vision_tower.")}
        hs = G.run_vision(cfg, vis_w, px)
        feats = hs[cfg.vision.select_layer]
        del hs, vis_w
        proj = G.run_projector(cfg, {k: w[k] for k in w.specs if k.startswith("mm_projector.")}, feats)
        gc.collect()
        llm, ver = build_hf_llm_streaming(cfg, w)
        emb = llm.model.embed_tokens
        img = torch.cat([proj[0], emb(torch.tensor([cfg.newline_token_id]))], 0)
        e = torch.cat([img if t == cfg.image_token_id else emb(torch.tensor([t])) for t in ids.tolist()], 0)[None]
        r = llm(inputs_embeds=e, use_cache=True, logits_to_keep=1)
        past, last = r.past_key_values, r.logits[0, -1].float()
        past0 = copy.deepcopy(past)
        steps = []
        for t in range(N_NEW):
            steps.append(last.clone())
            if t + 1 == N_NEW:
                break
            r = llm(input_ids=forced[t].view(1, 1), past_key_values=past, use_cache=True)
            past, last = r.past_key_values, r.logits[0, -1].float()
        lg = torch.stack(steps)
        gen, gmargin, past, last = [], [], past0, lg[0]
        for t in range(N_NEW):
            nxt = int(last.argmax())
            gen.append(nxt)
            t2v = last.topk(2).values
            gmargin.append(float(t2v[0] - t2v[1]))
            if t + 1 == N_NEW:
                break
            r = llm(input_ids=torch.tensor([[nxt]]), past_key_values=past, use_cache=True)
            past, last = r.past_key_values, r.logits[0, -1].float()
        print(f"reference (HF {ver}): teacher-forced argmax {lg.argmax(-1).tolist()}, greedy {gen} ({time.time() - t1:.0f}s)", flush=True)
        # the embedding table was popped from the lazy store while filling HF: fingerprints re-draw it
        np.savez_compressed(OUT.format("_ref"), **pack(w, cfg, px, ids, forced, feats, proj, e, (gen, gmargin), lg, {"hf_version": np.array(ver)}))
    margin = lg.topk(2, -1).values
    print(f"margins {[round(float(a - b), 3) for a, b in margin]}; wrote both fixtures in {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
