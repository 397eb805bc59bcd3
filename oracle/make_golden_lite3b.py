"""Full-depth fixtures for BASELINE configs[0] — NVILA-Lite-3B-shaped widths (SigLIP 26 layers -> 3x3 projector -> 36-layer Qwen2.5-3B-shaped
decoder, tied head), 1 x 448^2 image + 32-token prompt (S = 154), 8 greedy steps — TEST INFRASTRUCTURE.

Writes TWO files from the same seeded synthetic weights:
  * tests/golden/nvila_lite3b_full_depth_ref.npz — REFERENCE-EXECUTED: the reference's modeling_siglip.py and base_projector.py (loaded by file
    path, eager attention, fp32) and HF Qwen2ForCausalLM (fp32, eager, KV cache) — see make_golden_full_ref.py
  * tests/golden/nvila_lite3b_full_depth.npz — ORACLE-EXECUTED (oracle/vila_oracle.py)
tests/test_oracle_golden.py compares the two (CPU); tests/test_gpu_full_depth.py holds the HIP path to the reference-executed one.

    python oracle/make_golden_lite3b.py        # ~6 min on 8 cores, ~25 GB RSS; needs /root/reference
"""
from __future__ import annotations

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
from oracle.make_golden_full import LazyBf16Weights, N_NEW, TOPK      # noqa: E402
from oracle.make_golden_full_ref import build_hf_llm_streaming        # noqa: E402
from vila_amd import configs, synthetic                 # noqa: E402

SEED = 13
N_TEXT = 32
TAIL = (2.0, int(os.environ.get("VILA_TAIL_SEED", "0")), 10.0)
KEYS = ("llm.model.layers.0.mlp.gate_proj.weight", "llm.model.layers.35.self_attn.q_proj.bias", "llm.model.embed_tokens.weight",
        "vision_tower.vision_tower.vision_model.encoder.layers.25.mlp.fc1.weight", "mm_projector.layers.2.weight")
OUT = os.path.join(ROOT, "tests", "golden", "nvila_lite3b_full_depth{}.npz")


def pack(w, px, ids, feats, proj, e, gen, lg, extra):
    top = lg.topk(TOPK, -1)
    n_img = proj.shape[1]
    out = {f"fp_w{i}": w[k].reshape(-1)[:16].numpy().copy() for i, k in enumerate(KEYS)}
    out.update({"fp_pixels": px.reshape(-1)[:16].numpy().copy(), "seed": np.int64(SEED), "input_ids": ids.numpy(),
                "lm_head_tail": np.float32(TAIL[0]), "lm_head_tail_seed": np.int64(TAIL[1]), "lm_head_tail_max": np.float32(TAIL[2]),
                "greedy_ids": np.asarray(gen, dtype=np.int64), "top_ids": top.indices.numpy().astype(np.int32), "top_vals": top.values.numpy().astype(np.float32),
                "logit_absmax": lg.abs().amax(-1).numpy().astype(np.float32), "logit_norm": lg.norm(dim=-1).numpy().astype(np.float32),
                "vit_rows": feats[0, [0, 511, 1023], :256].numpy().astype(np.float32), "vit_norm": np.float32(feats.norm()),
                "proj_rows": proj[0, [0, n_img // 2, n_img - 1], :256].numpy().astype(np.float32), "proj_norm": np.float32(proj.norm()),
                "embed_rows": e[0, [0, n_img - 1, n_img, n_img + 1, e.shape[1] - 1], :256].numpy().astype(np.float32), "embed_norm": np.float32(e.norm())})
    out.update(extra)
    return out


def main():
    torch.manual_seed(0)
    cfg = configs.nvila_lite_3b()
    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = TAIL   # heavy-tailed row norms of the (tied) head: a peaked next-token distribution
    w = LazyBf16Weights(cfg, SEED)
    px = synthetic.make_pixels(cfg, 1, SEED).to(torch.bfloat16).float()
    ids = synthetic.make_prompt(cfg, N_TEXT, 1, SEED)
    t0 = time.time()
    with torch.no_grad():
        # ---- oracle ----
        feats_o = O.vision_tower_forward(px, w, cfg.vision)
        proj_o = O.projector_forward(feats_o, w, cfg.mm_projector_type)
        e_o, _ = O.vlm_prefill_embeds([px[0]], ids, w, cfg)
        gen_o, lg_o = O.greedy_generate(e_o, w, cfg, N_NEW, stop_at_eos=False)
        print(f"oracle: S = {e_o.shape[1]}, ids {gen_o.tolist()} ({time.time() - t0:.0f}s)", flush=True)
        np.savez_compressed(OUT.format(""), **pack(w, px, ids, feats_o, proj_o, e_o, gen_o.tolist(), lg_o.float(), {}))
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
        gen, steps = [], []
        for t in range(N_NEW):
            steps.append(last.clone())
            nxt = int(last.argmax())
            gen.append(nxt)
            if t + 1 == N_NEW:
                break
            r = llm(input_ids=torch.tensor([[nxt]]), past_key_values=past, use_cache=True)
            past, last = r.past_key_values, r.logits[0, -1].float()
        lg = torch.stack(steps)
        print(f"reference (HF {ver}): ids {gen} ({time.time() - t1:.0f}s)", flush=True)
        # the embedding table was popped from the lazy store while filling HF: fingerprints re-draw it
        np.savez_compressed(OUT.format("_ref"), **pack(w, px, ids, feats, proj, e, gen, lg, {"hf_version": np.array(ver)}))
    margin = lg.topk(2, -1).values
    print(f"margins {[round(float(a - b), 3) for a, b in margin]}; wrote both fixtures in {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
