"""REFERENCE-EXECUTED gradients of one SFT forward + backward — TEST INFRASTRUCTURE.

The reference trains with torch autograd over its own modules (llava/train/transformer_normalize_monkey_patch.py:183-268: loss = sum CE /
num_items_in_batch, `accelerator.backward(loss)`).  This script runs exactly that on a seeded tiny config with the REAL code: the reference's
SigLIP (`modeling_siglip.py`, hidden_states[-2]) and projector (`base_projector.py`) loaded by file path, HF `Qwen2ForCausalLM`, the splice of
llava_arch.py:412-490 restated only to connect them (image token -> projector rows + "\\n", labels of media rows IGNORE, right padding with
zero embeddings), one padded batch of two samples of different lengths — and stores, for EVERY parameter, the gradient's norm and its first 64
values.  tests/test_oracle_golden.py holds autograd through the oracle (`vlm_sft_loss`, packed branch) to it; the HIP step is held to the oracle
(tests/test_gpu_train.py).

    python oracle/make_golden_grads.py         # seconds; writes tests/golden/tiny_sft_grads.npz; needs /root/reference
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import make_golden as G                     # noqa: E402
from vila_amd import configs, synthetic                 # noqa: E402

IGNORE = -100
SEED = 5
OUT = os.path.join(ROOT, "tests", "golden", "tiny_sft_grads.npz")


def case():
    """cfg, weights (fp32 values of bf16-representable numbers), pixels [2,3,H,W], ids / labels / mask [2, L] (right padded)."""
    cfg = configs.tiny("mlp_downsample")
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, SEED).items()}
    px = synthetic.make_pixels(cfg, 2, SEED).to(torch.bfloat16).float()
    a, b = synthetic.make_prompt(cfg, 14, 1, SEED), synthetic.make_prompt(cfg, 9, 1, SEED + 1)
    L = max(a.numel(), b.numel())
    ids = torch.zeros((2, L), dtype=torch.int64)
    mask = torch.zeros((2, L), dtype=torch.bool)
    ids[0, : a.numel()], ids[1, : b.numel()] = a, b
    mask[0, : a.numel()], mask[1, : b.numel()] = True, True
    labels = ids.clone()
    labels[:, :5] = IGNORE                                   # the media token and the first text tokens carry no loss
    labels[~mask] = IGNORE
    return cfg, w, px, ids, labels, mask


def main():
    torch.manual_seed(0)
    cfg, w, px, ids, labels, mask = case()
    ms, bp = G.ref_siglip(), G.ref_projector()
    import types
    v = cfg.vision
    vc = ms.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                               num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                               num_channels=v.num_channels, layer_norm_eps=v.layer_norm_eps, hidden_act="gelu_pytorch_tanh")
    vc._attn_implementation = "eager"
    tower = ms.SiglipVisionModel(vc).train(False)
    tower.load_state_dict({k[len("vision_tower.vision_tower."):]: t for k, t in w.items() if k.startswith("vision_tower.")}, strict=False)
    proj = bp.MultimodalProjector(bp.MultimodalProjectorConfig(cfg.mm_projector_type),
                                  types.SimpleNamespace(mm_hidden_size=cfg.mm_hidden_size, hidden_size=cfg.llm.hidden_size)).train(False)
    proj.load_state_dict({k[len("mm_projector."):]: t for k, t in w.items() if k.startswith("mm_projector.")}, strict=True)
    llm, ver = G.build_hf_llm(cfg, w)
    for m in (tower, proj, llm):
        for p in m.parameters():
            p.requires_grad_(True)
    feats = tower(px, output_hidden_states=True).hidden_states[cfg.vision.select_layer]
    tokens = proj(feats)                                                        # [2, T, H]
    emb = llm.model.embed_tokens
    nl = emb(torch.tensor([cfg.newline_token_id]))
    rows_e, rows_l = [], []
    for b in range(2):
        es, ls = [], []
        for t, lab, ok in zip(ids[b].tolist(), labels[b].tolist(), mask[b].tolist()):
            if not ok:
                continue
            if t == cfg.image_token_id:
                es.append(torch.cat([tokens[b], nl], 0)); ls += [IGNORE] * (tokens.shape[1] + 1)
            else:
                es.append(emb(torch.tensor([t]))); ls.append(lab)
        rows_e.append(torch.cat(es, 0)); rows_l.append(torch.tensor(ls))
    S = max(r.shape[0] for r in rows_e)
    e = torch.zeros((2, S, cfg.llm.hidden_size))
    lab = torch.full((2, S), IGNORE, dtype=torch.int64)
    am = torch.zeros((2, S), dtype=torch.long)
    e = torch.stack([torch.cat([r, torch.zeros((S - r.shape[0], r.shape[1]))], 0) for r in rows_e], 0)      # keeps the graph
    for b in range(2):
        lab[b, : rows_l[b].numel()] = rows_l[b]
        am[b, : rows_l[b].numel()] = 1
    n_items = int((lab[:, 1:] != IGNORE).sum())
    out = llm(inputs_embeds=e, attention_mask=am, labels=lab, num_items_in_batch=n_items)
    out.loss.backward()
    fx = {"seed": np.int64(SEED), "hf_version": np.array(ver), "loss": np.float64(out.loss.item()), "num_items": np.int64(n_items),
          "input_ids": ids.numpy(), "labels": labels.numpy(), "mask": mask.numpy()}
    names = []
    for prefix, mod in (("vision_tower.vision_tower.", tower), ("mm_projector.", proj), ("llm.", llm)):
        for n, p in mod.named_parameters():
            name = prefix + n
            if name not in w:
                continue                                                         # the tower's pooling head: not part of VILA
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            fx[f"gn_{len(names)}"] = np.float64(g.double().norm())
            fx[f"gv_{len(names)}"] = g.reshape(-1)[:64].detach().numpy().astype(np.float32)
            names.append(name)
    fx["names"] = np.array(names)
    np.savez_compressed(OUT, **fx)
    print(f"wrote {OUT}: loss {out.loss.item():.6f}, {n_items} targets, {len(names)} gradient tensors ({os.path.getsize(OUT)} bytes)")


if __name__ == "__main__":
    main()
