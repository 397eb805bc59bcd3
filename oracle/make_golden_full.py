"""Full-depth golden for BASELINE configs[1] and configs[2] (NVILA-8B: 26 ViT + 28 LLM layers) — TEST INFRASTRUCTURE.

ORACLE-EXECUTED (its reference-executed twin at the same depth and weights is make_golden_full_ref.py -> nvila8b_full_depth_ref.npz, which is
the file the GPU test reads; tests/test_oracle_golden.py holds this file to that one): runs the fp32 CPU oracle (oracle/vila_oracle.py, itself
pinned against the reference-executed fixtures of make_golden.py at tiny depth) ONCE at the full depth on CPU-drawn seeded weights and stores
KB-sized fingerprints.

Round 4: the round-3 fixture followed GREEDY decoding into a fixed point — one heavy `lm_head` row won 7 of 8 steps by
margins 60-150x the logit error, so "ids equal" said almost nothing.  Now
  * configs[1] (1 x 448^2 image + 512-token prompt, S = 769): the 8 recorded steps are TEACHER-FORCED WITH A RANDOM ID SEQUENCE stored in the
    fixture (`forced_ids`), so every step sees a different input token and hidden state; the synthetic lm_head's row-norm tail (exponent, cap,
    seed) is searched so that the argmax of the 8 steps lands on DISTINCT tokens whose top-1 / top-2 margins sit inside 4..20x the logit error
    a bf16 path is expected to show (a margin the path could actually lose).  The hidden states do not depend on the head, so the search costs
    one [8, H] x [H, V] product per candidate.  The free-running greedy ids of the chosen head are stored too (`greedy_ids`).
  * configs[2] forward (missing #5): the SFT micro-batch the bench times — 4 samples of 1 image + 512 tokens, labels on the last
    256 text positions — through the full-depth forward: per-sample CE sums, loss = sum / num_items (llava/train/
    transformer_normalize_monkey_patch.py:261-268), and the top-32 logits of 8 labelled rows per sample.
  * a few rows of the tower / projector output and of the spliced embeddings; fingerprints of the drawn weights / inputs (so a host whose CPU
    RNG stream differs is detected instead of mis-compared).

    python oracle/make_golden_full.py            # ~12 min on 8 cores, ~25 GB RSS; writes tests/golden/nvila8b_full_depth.npz

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
SFT_B, SFT_T, SFT_LABELLED, SFT_ROWS = 4, 512, 256, 8
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


def fingerprints(w, px, ids, keys=FINGERPRINT_KEYS):
    fp = {f"fp_w{i}": w[k].reshape(-1)[:16].numpy().copy() for i, k in enumerate(keys)}
    fp["fp_pixels"] = px.reshape(-1)[:16].numpy().copy()
    fp["fp_ids"] = ids[:16].numpy().copy()
    return fp


def forced_sequence(cfg, n: int, seed: int) -> torch.Tensor:
    """The random ids the recorded steps are teacher-forced with (never a media / eos id)."""
    g = torch.Generator(device="cpu").manual_seed(3000 + seed)
    hi = min(cfg.llm.vocab_size, cfg.image_token_id, cfg.llm.eos_token_id) - 1
    return torch.randint(0, hi, (n,), generator=g, dtype=torch.int64)


def sft_batch(cfg, seed: int):
    """configs[2]'s micro-batch (bench.py sft_measure): SFT_B samples of 1 image + SFT_T text tokens, labels = ids on the last
    SFT_LABELLED text positions else -100."""
    px = synthetic.make_pixels(cfg, SFT_B, seed + 1).to(torch.bfloat16).float()
    ids = torch.stack([synthetic.make_prompt(cfg, SFT_T, 1, (seed + 1) * 10 + i) for i in range(SFT_B)], 0)
    labels = ids.clone()
    labels[:, : 1 + SFT_T - SFT_LABELLED] = -100
    return px, ids, labels


def untailed_head(cfg, w):
    """The synthetic lm_head WITHOUT the row-norm tail (bf16-rounded like the GPU model's): what configs[2]'s forward pin uses."""
    keep = cfg.lm_head_tail
    cfg.lm_head_tail = 0.0
    try:
        shape, kind = w.specs["llm.lm_head.weight"]
        return synthetic._draw("llm.lm_head.weight", shape, kind, cfg, w.seed, "cpu").to(torch.bfloat16).float()
    finally:
        cfg.lm_head_tail = keep


def sft_rows(S: int) -> torch.Tensor:
    """Row positions (in the spliced sequence) whose logits are fingerprinted: SFT_ROWS rows spread over the labelled tail."""
    return torch.linspace(S - SFT_LABELLED, S - 2, SFT_ROWS).round().long()


# ---- the lm_head search --------------------------------------------------------------------------------------------------------------
REL_ERR = 0.03          # fallback error model (no calibration file): logit error std ~ 3 % of the row's logit sigma — the first round-4 GPU run measured
#                         0.09-0.27 max-abs where a 0.8 % model (round 3's attractor-token steps) predicted 0.04-0.11
BAND = (4.0, 20.0)      # wanted: margin / expected max-abs error
TAILS = [(a, 4.0) for a in (2.0, 2.5, 3.0, 4.0, 5.0, 6.0, 8.0)]       # (the cap only rescales every logit: margins / errors do not see it)
SEEDS = range(256)


def expected_err(row_std, xn_norm, scale, top_ids, rel_err=REL_ERR):
    """Expected max-abs logit error over a step's top-32 entries: 2.2 x (max |N(0,1)| over 32) x REL_ERR x the largest row sigma among them,
    row sigma = row_std * scale_i * |xn| (random directions; row_std = lm_head_std, or init_std for a tied head)."""
    sig = row_std * scale[top_ids] * xn_norm[:, None]
    return 2.2 * rel_err * sig.max(-1).values


def score_logits(cfg, lg, xn_norm, scale, row_std=None, rel_err=REL_ERR):
    top = lg.topk(TOPK, -1)
    err = expected_err(cfg.lm_head_std if row_std is None else row_std, xn_norm, scale, top.indices, rel_err)
    margin = top.values[:, 0] - top.values[:, 1]
    ratio = margin / err
    mid = (BAND[0] * BAND[1]) ** 0.5
    inband = (ratio > 1.5 * BAND[0]) & (ratio < BAND[1] / 1.5)
    ids = top.indices[:, 0]
    n_distinct = len(set(ids[inband].tolist()))
    badness = float((ratio / mid).log().abs().sum())
    return (int(inband.sum()), n_distinct, -badness), ids, ratio, margin, err


def load_calibration(name: str, input_ids, forced):
    """oracle/calib/<name>_xn_gpu.npz (tools/dump_full_depth_hidden.py, run on an MI355X): the final-norm hidden states the HIP path feeds its
    lm_head on the same 8 steps.  They do not depend on the head, so logits_gpu(candidate) = XN_gpu @ head(candidate)^T — the search scores every
    candidate against the path's MEASURED logit error instead of a model of it.  None when the file is absent (then REL_ERR is used)."""
    path = os.path.join(ROOT, "oracle", "calib", f"{name}_xn_gpu.npz")
    if not os.path.exists(path):
        return None
    c = np.load(path)
    assert np.array_equal(c["input_ids"], input_ids.numpy()) and np.array_equal(c["forced_ids"], forced.numpy()), "calibration is for other inputs"
    return torch.from_numpy(c["xn"]).float()


def score_measured(lg, lg_gpu):
    """Like score_logits with the error MEASURED: err_t = max |logits_gpu - logits_ref| over the step's top-32 reference entries."""
    top = lg.topk(TOPK, -1)
    err = (lg_gpu.gather(1, top.indices) - top.values).abs().max(-1).values
    margin = top.values[:, 0] - top.values[:, 1]
    ratio = margin / err.clamp_min(1e-9)
    decisive = ratio > 1.25 * BAND[0]                    # the test's rule is > 4x; 25 % headroom
    inband = decisive & (ratio < BAND[1])
    ids = top.indices[:, 0]
    n_dec, n_dist = int(decisive.sum()), len(set(ids[decisive].tolist()))
    badness = float((ratio / (BAND[0] * BAND[1]) ** 0.5).log().abs().sum())
    return (min(n_dec, 6) + min(n_dist, 6), int(inband.sum()), n_dec, -badness), ids, ratio, margin, err


def search_head(cfg, w, XN, XN_gpu=None):
    """XN [n, H] = final-norm hidden state of every recorded step (XN_gpu: the HIP path's, see load_calibration).  Returns (tail_a, tail_max, tail_seed)."""
    name = "llm.lm_head.weight"
    shape, kind = w.specs[name]
    cfg.lm_head_tail = 0.0
    base = synthetic._draw(name, shape, kind, cfg, w.seed, "cpu")                 # row directions x lm_head_std; the tail multiplies rows
    L0 = XN @ base.t()
    L0g = XN_gpu @ base.t() if XN_gpu is not None else None
    del base
    xn_norm = XN.norm(dim=-1)
    best = None
    for a, m in TAILS:
        for s in SEEDS:
            cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = a, int(s), m
            scale = synthetic.lm_head_row_scale(name, shape[0], cfg)
            if L0g is not None:
                sc, ids, ratio, margin, err = score_measured(L0 * scale[None], L0g * scale[None])
            else:
                sc, ids, ratio, margin, err = score_logits(cfg, L0 * scale[None], xn_norm, scale)
            if best is None or sc > best[0]:
                best = (sc, a, m, s, ids.tolist(), [round(float(r), 1) for r in ratio])
                print(f"  tail a={a} max={m} seed={s}: score {sc[:3]}, ids {ids.tolist()}, margin/err {best[5]}", flush=True)
    return best[1], best[2], best[3]


def main():
    torch.manual_seed(0)
    cfg = configs.nvila_8b()
    w = LazyBf16Weights(cfg, SEED)
    px = synthetic.make_pixels(cfg, 1, SEED).to(torch.bfloat16).float()
    ids = synthetic.make_prompt(cfg, 512, 1, SEED)
    forced = forced_sequence(cfg, N_NEW, SEED)
    lc = cfg.llm
    t0 = time.time()
    with torch.no_grad():
        feats = O.vision_tower_forward(px, w, cfg.vision)                   # [1,1024,1152] = hidden_states[-2]
        proj = O.projector_forward(feats, w, cfg.mm_projector_type)         # [1,256,3584]
        print(f"tower+projector {time.time() - t0:.1f}s", flush=True)
        e, _ = O.vlm_prefill_embeds([px[0]], ids, w, cfg)                   # [1,769,3584]
        assert e.shape == (1, 769, lc.hidden_size)
        t1 = time.time()
        # every LLM pass below runs with a one-row stand-in head: KV cache and final hidden states do not depend on lm_head
        w.store["llm.lm_head.weight"] = torch.zeros((8, lc.hidden_size), dtype=torch.bfloat16)
        norm_w = w["llm.model.norm.weight"]
        _, past0, hs = O.qwen2_forward(e, w, lc, return_hidden=True)
        xn = [O.rms_norm(hs[-1][0, -1], norm_w, lc.rms_norm_eps)]
        del hs
        print(f"prefill {time.time() - t1:.1f}s", flush=True)
        past = past0
        for t in range(N_NEW - 1):                                           # teacher-forced with the random sequence
            _, past, hs = O.qwen2_forward(O.embed_tokens(forced[t].view(1, 1), w), w, lc, past=past, return_hidden=True)
            xn.append(O.rms_norm(hs[-1][0, -1], norm_w, lc.rms_norm_eps))
        XN = torch.stack(xn)                                                 # [8, H]
        print(f"teacher-forced steps done {time.time() - t1:.1f}s; |xn| {[round(float(v), 1) for v in XN.norm(dim=-1)]}", flush=True)
        # ---- configs[2]: hidden states of the labelled rows of the 4-sample batch (head-independent as well) ----
        spx, sids, slabels = sft_batch(cfg, SEED)
        sft_h, sft_lab = [], []
        for i in range(SFT_B):
            t2 = time.time()
            media = O.basic_image_encoder([spx[i]], w, cfg)
            ei, li, _ = O.embed_splice(sids[i][None], media, w, cfg, labels=slabels[i][None])
            _, _, hs = O.qwen2_forward(ei, w, lc, return_hidden=True)
            sft_h.append(O.rms_norm(hs[-1][0], norm_w, lc.rms_norm_eps))     # [S, H]
            sft_lab.append(li[0])
            del hs
            print(f"sft sample {i}: S = {ei.shape[1]} ({time.time() - t2:.0f}s)", flush=True)
        w.store.pop("llm.lm_head.weight")
        XN_gpu = load_calibration("nvila8b", ids, forced)
        print("search scored against " + ("the HIP path's measured hidden states (oracle/calib)" if XN_gpu is not None else f"the error model REL_ERR = {REL_ERR}"), flush=True)
        a, m, s = search_head(cfg, w, XN, XN_gpu)
        cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = a, int(s), m
        head = w["llm.lm_head.weight"]                                       # the chosen head, bf16-rounded like the GPU model's
        lg = (XN @ head.t()).float()
        scale = synthetic.lm_head_row_scale("llm.lm_head.weight", lc.vocab_size, cfg)
        sc, tf_ids, ratio, margin, err = score_logits(cfg, lg, XN.norm(dim=-1), scale)
        print(f"chosen tail a={a} max={m} seed={s}: teacher-forced argmax {tf_ids.tolist()}, margins {[round(float(x), 3) for x in margin]}, "
              f"expected err {[round(float(x), 3) for x in err]}, ratio {[round(float(x), 1) for x in ratio]}", flush=True)
        # free-running greedy with the chosen head, from the prefill's cache
        gen, gmargin, past, last = [], [], past0, lg[0]
        for t in range(N_NEW):
            nxt = int(last.argmax())
            gen.append(nxt)
            t2v = last.topk(2).values
            gmargin.append(float(t2v[0] - t2v[1]))
            if t + 1 == N_NEW:
                break
            logits, past = O.qwen2_forward(O.embed_tokens(torch.tensor([[nxt]]), w), w, lc, past=past)
            last = logits[0, -1]
        print(f"free-running greedy ids {gen}", flush=True)
        # configs[2] loss and logits rows — with the PLAIN synthetic head (no tail: `sft_head_tail` = 0 in the file).  The tailed head above is
        # chosen for the id test; it quadruples the logit scale, and with it the bf16 path's loss error (0.011 on a loss of 17.3 in the first
        # round-4 GPU run), while SURVEY §8c's |delta loss| <= 1e-2 is stated for the plain head (loss ~ ln V).  The hidden states are the same.
        head = untailed_head(cfg, w)
        S = sft_h[0].shape[0]
        rows = sft_rows(S)
        ce_sum, n_items, r_ids, r_vals = [], 0, [], []
        for h, lab in zip(sft_h, sft_lab):
            tgt = lab[1:]                                                    # HF ForCausalLMLoss: position p predicts label p + 1
            keep = (tgt != -100).nonzero().flatten()
            lgi = (h[keep] @ head.t()).float()
            ce_sum.append(float(torch.nn.functional.cross_entropy(lgi.double(), tgt[keep], reduction="sum")))
            n_items += int(keep.numel())
            tr = (h[rows] @ head.t()).float().topk(TOPK, -1)
            r_ids.append(tr.indices)
            r_vals.append(tr.values)
        loss = sum(ce_sum) / n_items
        print(f"configs[2] forward: loss {loss:.6f} over {n_items} targets, per-sample sums {[round(x, 3) for x in ce_sum]}", flush=True)
    out = fingerprints(w, px, ids)
    top = lg.topk(TOPK, -1)
    out.update({
        "seed": np.int64(SEED), "input_ids": ids.numpy(), "forced_ids": forced.numpy(), "tf_argmax_ids": tf_ids.numpy().astype(np.int64),
        "greedy_ids": np.asarray(gen, dtype=np.int64), "greedy_margins": np.asarray(gmargin, dtype=np.float32),
        "lm_head_tail": np.float32(a), "lm_head_tail_seed": np.int64(s), "lm_head_tail_max": np.float32(m),
        "top_ids": top.indices.numpy().astype(np.int32), "top_vals": top.values.numpy().astype(np.float32),
        "logit_absmax": lg.abs().amax(-1).numpy().astype(np.float32), "logit_norm": lg.norm(dim=-1).numpy().astype(np.float32),
        "vit_rows": feats[0, [0, 511, 1023], :256].numpy().astype(np.float32), "vit_norm": np.float32(feats.norm()),
        "proj_rows": proj[0, [0, 127, 255], :256].numpy().astype(np.float32), "proj_norm": np.float32(proj.norm()),
        "embed_rows": e[0, [0, 255, 256, 257, 768], :256].numpy().astype(np.float32), "embed_norm": np.float32(e.norm()),
        "sft_input_ids": sids.numpy(), "sft_labels": slabels.numpy(), "sft_fp_pixels": spx.reshape(SFT_B, -1)[:, :16].numpy().copy(),
        "sft_head_tail": np.float32(0.0), "sft_loss": np.float64(loss), "sft_ce_sums": np.asarray(ce_sum, dtype=np.float64), "sft_num_items": np.int64(n_items),
        "sft_rows": rows.numpy(), "sft_top_ids": torch.stack(r_ids).numpy().astype(np.int32), "sft_top_vals": torch.stack(r_vals).numpy().astype(np.float32),
    })
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes) in {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
