"""Full-depth golden for BASELINE configs[1] (NVILA-8B, 1 x 448^2 image + 512-token prompt, S = 769) — TEST INFRASTRUCTURE.

Runs the fp32 CPU oracle (oracle/vila_oracle.py, itself pinned against the reference-executed fixtures of make_golden.py) ONCE at
the full 26 + 28 layer depth on CPU-drawn seeded weights and stores KB-sized fingerprints:
  * the top-32 (ids, values) of the prefill's last-row logits and of 8 teacher-forced decode steps, the greedy ids
  * a few rows of the tower / projector output and of the spliced embeddings
  * fingerprints of the drawn weights / inputs (so a host whose CPU RNG stream differs is detected instead of mis-compared)
tests/test_gpu_full_depth.py draws the same weights on the host of the GPU box, runs the HIP path and compares (logits 3e-2 on the
stored top-32 entries, margin-aware bit-exact ids).

    python oracle/make_golden_full.py            # ~5 min on 8 cores, ~20 GB RSS; writes tests/golden/nvila8b_full_depth.npz

Weights are held as bf16 (the values the GPU model holds) and upcast per use, so 8.06 B parameters fit in 16 GB of host RAM.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import vila_oracle as O          # noqa: E402
from vila_amd import configs, synthetic      # noqa: E402

SEED = 11
N_NEW = 8
TOPK = 32
OUT = os.path.join(ROOT, "tests", "golden", "nvila8b_full_depth.npz")
FINGERPRINT_KEYS = ("llm.model.layers.0.mlp.gate_proj.weight", "llm.model.layers.27.self_attn.q_proj.bias", "llm.lm_head.weight",
                    "vision_tower.vision_tower.vision_model.encoder.layers.25.mlp.fc1.weight", "mm_projector.layers.2.weight")


class LazyBf16Weights(dict):
    """name -> fp32 tensor, drawn with synthetic._draw on first use, kept as bf16, upcast on every access."""

    def __init__(self, cfg, seed):
        super().__init__()
        self.cfg, self.seed = cfg, seed
        self.specs = {n: (shape, kind) for n, shape, kind in synthetic.all_specs(cfg)}
        self.store = {}

    def __contains__(self, k):
        return k in self.specs

    def __getitem__(self, k):
        if k not in self.store:
            shape, kind = self.specs[k]
            self.store[k] = synthetic._draw(k, shape, kind, self.cfg, self.seed, "cpu").to(torch.bfloat16)
        return self.store[k].float()


def fingerprints(w, px, ids):
    fp = {f"fp_w{i}": w[k].reshape(-1)[:16].numpy().copy() for i, k in enumerate(FINGERPRINT_KEYS)}
    fp["fp_pixels"] = px.reshape(-1)[:16].numpy().copy()
    fp["fp_ids"] = ids[:16].numpy().copy()
    return fp


def main():
    torch.manual_seed(0)
    cfg = configs.nvila_8b()
    w = LazyBf16Weights(cfg, SEED)
    px = synthetic.make_pixels(cfg, 1, SEED).to(torch.bfloat16).float()
    ids = synthetic.make_prompt(cfg, 512, 1, SEED)
    out = fingerprints(w, px, ids)
    t0 = time.time()
    with torch.no_grad():
        feats = O.vision_tower_forward(px, w, cfg.vision)                   # [1,1024,1152] = hidden_states[-2]
        proj = O.projector_forward(feats, w, cfg.mm_projector_type)         # [1,256,3584]
        print(f"tower+projector {time.time() - t0:.1f}s", flush=True)
        e, _ = O.vlm_prefill_embeds([px[0]], ids, w, cfg)                   # [1,769,3584]
        assert e.shape == (1, 769, cfg.llm.hidden_size)
        t1 = time.time()
        ids_free, lg_free = O.greedy_generate(e, w, cfg, N_NEW, stop_at_eos=False)
        print(f"prefill + {N_NEW} decode steps {time.time() - t1:.1f}s; ids {ids_free.tolist()}", flush=True)
    top = lg_free.topk(TOPK, -1)
    out.update({
        "seed": np.int64(SEED), "input_ids": ids.numpy(), "greedy_ids": ids_free.numpy(),
        "top_ids": top.indices.numpy().astype(np.int32), "top_vals": top.values.numpy().astype(np.float32),
        "logit_absmax": lg_free.abs().amax(-1).numpy().astype(np.float32), "logit_norm": lg_free.norm(dim=-1).numpy().astype(np.float32),
        "vit_rows": feats[0, [0, 511, 1023], :256].numpy().astype(np.float32), "vit_norm": np.float32(feats.norm()),
        "proj_rows": proj[0, [0, 127, 255], :256].numpy().astype(np.float32), "proj_norm": np.float32(proj.norm()),
        "embed_rows": e[0, [0, 255, 256, 257, 768], :256].numpy().astype(np.float32), "embed_norm": np.float32(e.norm()),
    })
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes) in {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
