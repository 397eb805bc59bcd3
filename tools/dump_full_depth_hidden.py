"""TEST-INFRASTRUCTURE tool (runs on the GPU box): the final-norm hidden states the HIP path feeds its lm_head on the 8 teacher-forced steps of the
full-depth fixtures — what oracle/make_golden_full.py / make_golden_lite3b.py use to CALIBRATE their search for a synthetic lm_head whose argmax
margins sit inside 4..20x the bf16 path's real logit error (a margin the path could lose) instead of guessing that error from a model.

The hidden states do not depend on the head that is searched (8B: untied head; Lite-3B: the rows the decoder can see are pinned), so one run
serves every candidate: logits_gpu(candidate) = XN_gpu @ head(candidate)^T.  XN_gpu is recovered from the step's fp32 logits by least squares
against the head the model holds (V = 152 k equations for H unknowns per step; torch on the GPU — a tool, not the product path).

    gpurun -- python tools/dump_full_depth_hidden.py      # writes gpurun_out/calib/{nvila8b,nvila_lite3b}_xn_gpu.npz  (copy to oracle/calib/)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vila_amd import configs, synthetic      # noqa: E402
from vila_amd.vlm import build_model         # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "calib")


def run(name, cfg, fixture, n_text):
    fx = np.load(os.path.join(ROOT, "tests", "golden", fixture))
    seed = int(fx["seed"])
    cfg.lm_head_tail, cfg.lm_head_tail_seed, cfg.lm_head_tail_max = float(fx["lm_head_tail"]), int(fx["lm_head_tail_seed"]), float(fx["lm_head_tail_max"])
    if "lm_head_tail_unit_rows" in fx.files:
        cfg.lm_head_tail_unit_rows = tuple(int(r) for r in fx["lm_head_tail_unit_rows"])
    model = build_model(cfg, seed=seed, draw_device="cpu")
    px = synthetic.make_pixels(cfg, 1, seed).to(torch.bfloat16).cuda()
    ids = torch.from_numpy(fx["input_ids"])
    forced = torch.from_numpy(fx["forced_ids"])
    e, _, _ = model._embed(ids[None], {"image": [px[0]]})
    _, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=len(forced), return_logits=True, forced_ids=forced, use_graph=False)
    lg = lg.float()                                                     # [8, V]
    head = (model.llm.model.embed_tokens.weight if cfg.llm.tie_word_embeddings else model.llm.lm_head.weight).float()      # [V, H]
    # normal equations in fp64 on the device: (W^T W) X = W^T L^T
    A = (head.double().t() @ head.double())
    B = head.double().t() @ lg.double().t()
    X = torch.linalg.solve(A, B).t().float()                            # [8, H]
    resid = float((X @ head.t() - lg).abs().max())
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"{name}_xn_gpu.npz")
    np.savez_compressed(path, xn=X.cpu().numpy(), forced_ids=forced.numpy(), input_ids=ids.numpy(), seed=np.int64(seed), lstsq_resid=np.float32(resid),
                        logit_absmax=lg.abs().amax(-1).cpu().numpy())
    print(f"{name}: XN_gpu {tuple(X.shape)}, |xn| {[round(float(v), 1) for v in X.norm(dim=-1)]}, least-squares residual {resid:.2e} -> {path}", flush=True)
    del model
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["nvila8b", "nvila_lite3b"]
    if "nvila8b" in which:
        run("nvila8b", configs.nvila_8b(), "nvila8b_full_depth_ref.npz", 512)
    if "nvila_lite3b" in which:
        run("nvila_lite3b", configs.nvila_lite_3b(), "nvila_lite3b_full_depth_ref.npz", 32)
