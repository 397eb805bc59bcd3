"""Full-depth golden for BASELINE configs[1] (NVILA-8B, 1 x 448^2 image + 512-token prompt, S = 769) — TEST INFRASTRUCTURE.

ORACLE-EXECUTED (its reference-executed twin at the same depth and weights is make_golden_full_ref.py -> nvila8b_full_depth_ref.npz, and
tests/test_oracle_golden.py holds this file to that one): runs the fp32 CPU oracle (oracle/vila_oracle.py, itself pinned against the
reference-executed fixtures of make_golden.py at tiny depth) ONCE at the full 26 + 28 layer depth on CPU-drawn seeded weights and
stores KB-sized fingerprints.  The synthetic lm_head has heavy-tailed (Pareto) row norms and the seed of those norms is searched so
that at least 7 of the 8 greedy steps have a top-1 / top-2 margin well above 4x the logit error a bf16 path shows at this depth
(the round-2 fixture's i.i.d. Gaussian head left ONE decisive step of eight):
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


TAIL_A = 2.0            # Pareto exponent of the synthetic lm_head's row norms (configs.VilaConfig.lm_head_tail)
TAIL_MAX = 10.0
REL_ERR = 0.04          # bf16 GPU path vs this oracle at full depth: logit error std ~ 4 % of the row's logit sigma (round-2 measurement)
MAX_CANDIDATES = 24


def decode_candidate(cfg, w, past, xn_last, tail_seed):
    """Greedy ids / logits of N_NEW steps for the lm_head whose row norms are drawn with `tail_seed` (the rest of the model, hence the
    prefill's KV cache and final hidden state, does not depend on it)."""
    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = TAIL_A, int(tail_seed), TAIL_MAX
    name = "llm.lm_head.weight"
    shape, kind = w.specs[name]
    w.store[name] = synthetic._draw(name, shape, kind, cfg, w.seed, "cpu").to(torch.bfloat16)      # the values the GPU model holds
    head = w[name]
    last = torch.nn.functional.linear(xn_last, head).float()
    ids, step_logits = [], []
    for t in range(N_NEW):
        step_logits.append(last.clone())
        nxt = int(last.argmax())
        ids.append(nxt)
        if t + 1 == N_NEW:
            break
        e = O.embed_tokens(torch.tensor([[nxt]]), w)
        logits, past = O.qwen2_forward(e, w, cfg.llm, past=past)
        last = logits[0, -1]
    return torch.tensor(ids, dtype=torch.int64), torch.stack(step_logits)


def predicted_decisive(cfg, lg):
    """How many steps have an oracle top-1 / top-2 margin above 4x the logit error the bf16 path is expected to show on that step's
    top-32 entries (x1.5 safety): error std of row i ~ REL_ERR * sigma_i, sigma_i = lm_head_std * sqrt(H) * row norm scale."""
    scale = synthetic.lm_head_row_scale("llm.lm_head.weight", cfg.llm.vocab_size, cfg)
    top = lg.topk(TOPK, -1)
    sig = cfg.lm_head_std * (cfg.llm.hidden_size ** 0.5) * scale[top.indices]           # [n, 32]
    err = 2.2 * REL_ERR * sig.max(-1).values                                              # max |N(0, s)| over 32 entries ~ 2.2 s
    margin = top.values[:, 0] - top.values[:, 1]
    return int((margin > 1.5 * 4 * err).sum()), margin, err


def main():
    torch.manual_seed(0)
    cfg = configs.nvila_8b()
    w = LazyBf16Weights(cfg, SEED)
    px = synthetic.make_pixels(cfg, 1, SEED).to(torch.bfloat16).float()
    ids = synthetic.make_prompt(cfg, 512, 1, SEED)
    t0 = time.time()
    with torch.no_grad():
        feats = O.vision_tower_forward(px, w, cfg.vision)                   # [1,1024,1152] = hidden_states[-2]
        proj = O.projector_forward(feats, w, cfg.mm_projector_type)         # [1,256,3584]
        print(f"tower+projector {time.time() - t0:.1f}s", flush=True)
        e, _ = O.vlm_prefill_embeds([px[0]], ids, w, cfg)                   # [1,769,3584]
        assert e.shape == (1, 769, cfg.llm.hidden_size)
        t1 = time.time()
        # prefill ONCE with a one-row stand-in head (the KV cache and the final hidden state do not depend on lm_head)
        w.store["llm.lm_head.weight"] = torch.zeros((8, cfg.llm.hidden_size), dtype=torch.bfloat16)
        _, past, hs = O.qwen2_forward(e, w, cfg.llm, return_hidden=True)
        xn_last = O.rms_norm(hs[-1][0, -1], w["llm.model.norm.weight"], cfg.llm.rms_norm_eps)
        del hs
        print(f"prefill {time.time() - t1:.1f}s", flush=True)
        best = None
        for cand in range(MAX_CANDIDATES):
            t2 = time.time()
            ids_c, lg_c = decode_candidate(cfg, w, past, xn_last, cand)
            n_dec, margin, err = predicted_decisive(cfg, lg_c)
            print(f"tail seed {cand}: ids {ids_c.tolist()} predicted decisive {n_dec}/{N_NEW} margins {[round(float(m), 2) for m in margin]} "
                  f"4x err {[round(float(4 * x), 2) for x in err]} ({time.time() - t2:.0f}s)", flush=True)
            score = (n_dec, len(set(ids_c.tolist())))
            if best is None or score > best[0]:
                best = (score, cand, ids_c, lg_c)
            if n_dec >= N_NEW - 1 and len(set(ids_c.tolist())) >= 3:
                break
        _, tail_seed, ids_free, lg_free = best
        cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = TAIL_A, int(tail_seed), TAIL_MAX
        w.store.pop("llm.lm_head.weight")
        print(f"chosen tail seed {tail_seed}: ids {ids_free.tolist()}", flush=True)
    out = fingerprints(w, px, ids)                                              # (draws the chosen lm_head for its fingerprint)
    top = lg_free.topk(TOPK, -1)
    out.update({
        "seed": np.int64(SEED), "input_ids": ids.numpy(), "greedy_ids": ids_free.numpy(),
        "lm_head_tail": np.float32(TAIL_A), "lm_head_tail_seed": np.int64(tail_seed), "lm_head_tail_max": np.float32(TAIL_MAX),
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
