"""REFERENCE-EXECUTED full-depth BACKWARD fixture (round 6, VERDICT round 5 "parity hardening (ii)") — TEST INFRASTRUCTURE.

Row a13 was pinned at the gradient level only at tiny size and at 8B width with a reduced layer count.  This script runs ONE sample of BASELINE
configs[2] (1 x 448^2 image + 512 text tokens, S = 769, labels on the last 256 text positions) through the REFERENCE'S OWN CODE at the full
26 + 28 layer depth under torch autograd, in fp32:
  * vision tower : /root/reference/llava/model/multimodal_encoder/siglip/modeling_siglip.py (by file path, eager attention), hidden_states[-2]
  * projector    : /root/reference/llava/model/multimodal_projector/base_projector.py (`mlp_downsample`)
  * decoder      : HF Qwen2ForCausalLM (the class the reference instantiates, language_model/builder.py:64), eager attention,
                   loss = HF's own ForCausalLMLoss(labels, num_items_in_batch) exactly as llava_llama.py:134-149 calls it
and records the gradient of PROBE tensors that together see the whole chain: the patch embedding (through all 54 layers), tower layers 0 and
25, the projector, decoder layers 0 / 13 / 27, the final norm and lm_head.  Only the probes require grad (the other 8 B parameters' gradients
are never materialised), activations are kept by autograd as usual: ~50 GB RSS, ~10 min on 8 cores.

Per probe the fixture keeps a seeded random subset of <= 16 384 elements (`gi_<k>` flat indices, `gv_<k>` values) plus the full tensor's norm
`gn_<k>`: the GPU test compares cosine and norm ratio on the subset.  Weights / inputs are those of make_golden_full.py (same seed, the PLAIN
synthetic head as in the configs[2] forward pin), so the GPU side rebuilds them with build_model(draw_device="cpu").

    python oracle/make_golden_full_grads_ref.py        # needs /root/reference
    python oracle/make_golden_full_grads_ref.py --bf16 # second pass: the SAME reference modules run in bf16 (weights, activations, autograd), the
                                                       # precision the reference trains in (`--bf16 True`, llava/train/args.py:229).  Adds `gvb_<k>`
                                                       # (values at the same indices) and `cosb_<k>` = cosine(bf16 run, fp32 run): the noise floor a
                                                       # bf16 path has at this depth, which the GPU test uses as the per-tensor reference point.
"""
from __future__ import annotations

import gc
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import make_golden as G                                            # noqa: E402
from oracle.make_golden_full import LazyBf16Weights, SEED, sft_batch, untailed_head      # noqa: E402
from oracle.make_golden_full_ref import build_hf_llm_streaming                 # noqa: E402
from vila_amd import configs                                                   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "nvila8b_full_depth_grads_ref.npz")
IGNORE = -100
SUBSET = 65536          # indices drawn
KEEP = 16384            # ... of which the first KEEP are stored (1.3 MB fixture)
VT = "vision_tower.vision_tower.vision_model."
PROBES = [
    VT + "embeddings.patch_embedding.weight",
    VT + "embeddings.position_embedding.weight",
    VT + "encoder.layers.0.self_attn.q_proj.weight",
    VT + "encoder.layers.25.mlp.fc2.weight",
    "mm_projector.layers.2.weight",
    "mm_projector.layers.1.weight",
    "llm.model.layers.0.self_attn.q_proj.weight",
    "llm.model.layers.0.self_attn.k_proj.bias",
    "llm.model.layers.0.input_layernorm.weight",
    "llm.model.layers.13.mlp.down_proj.weight",
    "llm.model.layers.27.self_attn.o_proj.weight",
    "llm.model.layers.27.mlp.gate_proj.weight",
    "llm.model.norm.weight",
    "llm.lm_head.weight",
]


def subset_indices(numel: int, k: int) -> torch.Tensor:
    if numel <= SUBSET:
        return torch.arange(numel, dtype=torch.int64)[:KEEP]
    g = torch.Generator().manual_seed(7000 + k)
    return torch.randint(0, numel, (SUBSET,), generator=g, dtype=torch.int64)[:KEEP]


def main():
    torch.manual_seed(0)
    t0 = time.time()
    bf16 = "--bf16" in sys.argv
    dt = torch.bfloat16 if bf16 else torch.float32
    cfg = configs.nvila_8b()
    w = LazyBf16Weights(cfg, SEED)
    spx, sids, slabels = sft_batch(cfg, SEED)
    px, ids, labels = spx[0:1], sids[0], slabels[0]                            # sample 0 of the configs[2] micro-batch
    ms, bp = G.ref_siglip(), G.ref_projector()
    v = cfg.vision
    vc = ms.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                               num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                               num_channels=v.num_channels, layer_norm_eps=v.layer_norm_eps, hidden_act="gelu_pytorch_tanh")
    vc._attn_implementation = "eager"
    tower = ms.SiglipVisionModel(vc).train(False).to(dt)
    tower.load_state_dict({k[len("vision_tower.vision_tower."):]: w[k] for k in w.specs if k.startswith("vision_tower.")}, strict=False)
    proj = bp.MultimodalProjector(bp.MultimodalProjectorConfig(cfg.mm_projector_type),
                                  types.SimpleNamespace(mm_hidden_size=cfg.mm_hidden_size, hidden_size=cfg.llm.hidden_size)).train(False).to(dt)
    proj.load_state_dict({k[len("mm_projector."):]: w[k] for k in w.specs if k.startswith("mm_projector.")}, strict=True)
    head = untailed_head(cfg, w)
    for k in list(w.store):
        if not k.startswith("llm."):
            w.store.pop(k)
    gc.collect()
    llm, ver = build_hf_llm_streaming(cfg, w, dtype=dt)
    with torch.no_grad():
        llm.lm_head.weight.copy_(head)
    del head
    gc.collect()
    print(f"reference modules built and filled ({ver}) {time.time() - t0:.0f}s", flush=True)
    params = {}
    for prefix, mod in (("vision_tower.vision_tower.", tower), ("mm_projector.", proj), ("llm.", llm)):
        for n, p in mod.named_parameters():
            params[prefix + n] = p
            p.requires_grad_(False)
    for name in PROBES:
        params[name].requires_grad_(True)
    t1 = time.time()
    feats = tower(px.to(dt), output_hidden_states=True).hidden_states[cfg.vision.select_layer]
    tokens = proj(feats)[0]                                                     # [256, H]
    emb = llm.model.embed_tokens
    img = torch.cat([tokens, emb(torch.tensor([cfg.newline_token_id]))], 0)
    parts, labs = [], []
    for tok, lab in zip(ids.tolist(), labels.tolist()):                          # llava_arch.py:454-476
        if tok == cfg.image_token_id:
            parts.append(img)
            labs += [IGNORE] * img.shape[0]
        else:
            parts.append(emb(torch.tensor([tok])))
            labs.append(lab)
    e, li = torch.cat(parts, 0)[None], torch.tensor(labs, dtype=torch.int64)[None]
    assert e.shape[1] == 769
    n_items = int((li[0, 1:] != IGNORE).sum())
    out = llm(inputs_embeds=e, labels=li, num_items_in_batch=n_items, use_cache=False)
    loss = float(out.loss)
    print(f"forward {time.time() - t1:.0f}s: loss {loss:.6f}, {n_items} targets", flush=True)
    t2 = time.time()
    out.loss.backward()
    print(f"backward {time.time() - t2:.0f}s", flush=True)
    if bf16:                                                                     # second pass: add the bf16 run's values to the fp32 fixture
        fx = dict(np.load(OUT))
        assert fx["names"].tolist() == PROBES and int(fx["num_items"]) == n_items
        fx["loss_bf16"] = np.float64(loss)
        for k, name in enumerate(PROBES):
            flat = params[name].grad.reshape(-1).float()
            idx = torch.from_numpy(fx[f"gi_{k}"])
            got, ref = flat[idx].double(), torch.from_numpy(fx[f"gv_{k}"]).double()
            fx[f"gvb_{k}"] = flat[idx].numpy().astype(np.float32)
            fx[f"cosb_{k}"] = np.float64(torch.nn.functional.cosine_similarity(got, ref, dim=0))
            fx[f"gnb_{k}"] = np.float64(flat.double().norm())
            print(f"  {name}: bf16 reference vs fp32 reference cosine {float(fx[f'cosb_{k}']):.5f}, |g| ratio {float(fx[f'gnb_{k}'] / fx[f'gn_{k}']):.4f}", flush=True)
        np.savez_compressed(OUT, **fx)
        print(f"updated {OUT} ({os.path.getsize(OUT)} bytes): bf16 loss {loss:.5f} vs fp32 {float(fx['loss']):.5f}, {time.time() - t0:.0f}s", flush=True)
        return
    fx = {"seed": np.int64(SEED), "hf_version": np.array(ver), "loss": np.float64(loss), "num_items": np.int64(n_items),
          "input_ids": ids.numpy(), "labels": labels.numpy(), "fp_pixels": px.reshape(-1)[:16].numpy().copy(), "names": np.array(PROBES)}
    for k, name in enumerate(PROBES):
        g = params[name].grad
        assert g is not None, name
        flat = g.reshape(-1)
        idx = subset_indices(flat.numel(), k)
        fx[f"gi_{k}"] = idx.numpy()
        fx[f"gv_{k}"] = flat[idx].detach().numpy().astype(np.float32)
        fx[f"gn_{k}"] = np.float64(flat.double().norm())
        print(f"  {name}: |g| = {float(fx[f'gn_{k}']):.6e} ({flat.numel()} elements, {idx.numel()} kept)", flush=True)
    np.savez_compressed(OUT, **fx)
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes) in {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
