import json
import os

import torch

# "grads cosine >= 0.999" — the stated bound of every model-gradient comparison (bf16 HIP step vs fp32 autograd).
# A tensor that cannot meet it carries a per-tensor exception at its call site, with the measured value and the reason.
GRAD_COS_MIN = 0.999


def grad_cos(test: str, name: str, got: torch.Tensor, ref: torch.Tensor) -> float:
    """Cosine of a gradient tensor against its fp32 reference.  With VILA_DUMP_COS=<file> every measured value is appended as a JSON line
    (what the per-tensor exceptions in the tests were read from)."""
    cos = float(torch.nn.functional.cosine_similarity(got.detach().double().cpu().flatten(), ref.detach().double().cpu().flatten(), dim=0))
    path = os.environ.get("VILA_DUMP_COS")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, "tensor": name, "cos": cos, "ref_norm": float(ref.norm()), "numel": ref.numel()}) + "\n")
    return cos


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def randn_bf16(*shape, seed=0, scale=1.0, device="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device=device, dtype=torch.bfloat16)


def margin_aware_ids(lg, lg_o, ids_o, free_ids=None):
    """SURVEY §8c id rule.  `lg` [n,V]: teacher-forced GPU logits (the oracle's ids were fed), `lg_o` / `ids_o`: the oracle's.  At every
    step whose oracle top-1/top-2 margin exceeds 4x the observed max-abs logit error the argmax must be bit-exact; `free_ids` (a
    free-running greedy run of the same model) must equal the oracle's ids up to the first non-decisive step — until then it was fed
    exactly the teacher-forced inputs.  Returns the decisive mask."""
    import torch
    lg, lg_o = lg.detach().float().cpu(), lg_o.detach().float().cpu()
    err = float((lg - lg_o).abs().max())
    top2 = lg_o.topk(2, -1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 4 * err
    assert bool(decisive.any()), f"no decisive step (err {err:.3e}, margins {(top2[:, 0] - top2[:, 1]).tolist()})"
    got = lg.argmax(-1)
    assert torch.equal(got[decisive], ids_o[decisive]), f"ids {got.tolist()} vs oracle {ids_o.tolist()} (err {err:.3e}, decisive {decisive.tolist()})"
    if free_ids is not None:
        free_ids = free_ids.detach().cpu().reshape(-1)
        nd = (~decisive).nonzero().flatten()
        k = int(nd[0]) if nd.numel() else len(ids_o)
        assert torch.equal(free_ids[:k], ids_o[:k]), f"free-running ids {free_ids.tolist()} vs oracle {ids_o.tolist()} (first {k} must match)"
    return decisive
