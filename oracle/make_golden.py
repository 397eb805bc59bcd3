"""Generate tests/golden/*.npz by EXECUTING the reference's own code in this container.

Run from the repo root (needs /root/reference and `transformers`; neither exists / is needed on the GPU box):

    python oracle/make_golden.py

What is executed:
  * `/root/reference/llava/model/multimodal_encoder/siglip/modeling_siglip.py` loaded by file path
    (SiglipVisionModel, eager attention) -> hidden_states[-2]  (vision_encoder.py:44-52)
  * `/root/reference/llava/model/multimodal_projector/base_projector.py` loaded by file path with a stub
    `timm.models.layers.Mlp` (only the PS3 head uses it, :229)
  * HF `transformers` Qwen2ForCausalLM (third-party; reference pins 4.46.0, installed here: see `hf_version`
    stored in every fixture) -> logits, loss, greedy generate(inputs_embeds=...)
Weights are NOT stored: they are re-drawn from `vila_amd.synthetic.make_weights(cfg, seed)` (per-tensor seeded
torch CPU generators); a checksum of the weights is stored so generator drift is detected.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from vila_amd import configs, synthetic  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_siglip():
    return load_by_path("ref_modeling_siglip", f"{REF}/llava/model/multimodal_encoder/siglip/modeling_siglip.py")


def ref_projector():
    import transformers  # noqa: F401  (must be imported BEFORE the timm stub: its availability probe trips on a spec-less module)
    from transformers import AutoModel  # noqa: F401  force the lazy module to resolve
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        tm = types.ModuleType("timm.models")
        tl = types.ModuleType("timm.models.layers")

        class Mlp(torch.nn.Module):  # stub; PS3-only (base_projector.py:229)
            pass

        tl.Mlp = Mlp
        timm.models = tm
        tm.layers = tl
        sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    return load_by_path("ref_base_projector", f"{REF}/llava/model/multimodal_projector/base_projector.py")


def weight_checksum(w):
    acc = 0.0
    for k in sorted(w):
        acc += float(w[k].double().abs().sum())
    return acc


def run_vision(cfg, w, pixels, timing=None):
    """timing (optional dict): receives `forward_s`, the seconds of the reference module's forward alone (not its construction / weight load)."""
    ms = ref_siglip()
    v = cfg.vision
    hf_cfg = ms.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                                   num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                                   image_size=v.image_size, patch_size=v.patch_size, num_channels=v.num_channels,
                                   layer_norm_eps=v.layer_norm_eps, hidden_act="gelu_pytorch_tanh")
    hf_cfg._attn_implementation = "eager"
    model = ms.SiglipVisionModel(hf_cfg).eval()
    pre = "vision_tower.vision_tower."
    sd = {k[len(pre):]: t for k, t in w.items() if k.startswith(pre)}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith("vision_model.head.") for m in missing), missing  # pooling head unused by VILA
    import time
    t0 = time.time()
    with torch.no_grad():
        out = model(pixels, output_hidden_states=True)
    if timing is not None:
        timing["forward_s"] = time.time() - t0
    return [h.clone() for h in out.hidden_states]


def run_projector(cfg, w, feats):
    bp = ref_projector()
    pc = bp.MultimodalProjectorConfig(cfg.mm_projector_type)
    ns = types.SimpleNamespace(mm_hidden_size=cfg.mm_hidden_size, hidden_size=cfg.llm.hidden_size)
    model = bp.MultimodalProjector(pc, ns).eval()
    pre = "mm_projector."
    sd = {k[len(pre):]: t for k, t in w.items() if k.startswith(pre)}
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        return model(feats)


def build_hf_llm(cfg, w):
    import transformers
    from transformers import Qwen2Config, Qwen2ForCausalLM

    c = cfg.llm
    hc = Qwen2Config(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                     num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                     num_key_value_heads=c.num_key_value_heads, rms_norm_eps=c.rms_norm_eps,
                     rope_theta=c.rope_theta, rope_parameters={"rope_type": "default", "rope_theta": c.rope_theta},
                     tie_word_embeddings=c.tie_word_embeddings, max_position_embeddings=4096,
                     use_sliding_window=False, attention_dropout=0.0, eos_token_id=c.eos_token_id,
                     pad_token_id=None, bos_token_id=None)
    hc._attn_implementation = "eager"
    model = Qwen2ForCausalLM(hc).eval().float()
    assert model.config.head_dim == c.head_dim if hasattr(model.config, "head_dim") else True
    pre = "llm."
    sd = {k[len(pre):]: t for k, t in w.items() if k.startswith(pre)}
    if c.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    model.load_state_dict(sd, strict=True)
    return model, transformers.__version__


def golden_case(name, cfg, seed, n_text, n_new, out_dir):
    torch.manual_seed(0)
    w = synthetic.make_weights(cfg, seed)
    pixels = synthetic.make_pixels(cfg, 2, seed)
    fx = {"seed": np.int64(seed), "weight_checksum": np.float64(weight_checksum(w))}

    hs = run_vision(cfg, w, pixels)
    sel = hs[cfg.vision.select_layer]
    fx["vit_embeddings"] = hs[0].numpy()
    fx["vit_layer1"] = hs[1].numpy()
    fx["vit_selected"] = sel.numpy()
    proj = run_projector(cfg, w, sel)
    fx["projector_out"] = proj.numpy()

    llm, ver = build_hf_llm(cfg, w)
    fx["hf_version"] = np.array(ver)
    # splice (restated here from llava_arch.py:412-490 only to feed the HF model; the oracle's own splice is
    # checked against this one in tests): prompt = <image> + text, one image
    ids = synthetic.make_prompt(cfg, n_text, 1, seed)
    fx["input_ids"] = ids.numpy()
    emb = llm.model.embed_tokens
    with torch.no_grad():
        img = torch.cat([proj[0], emb(torch.tensor([cfg.newline_token_id]))], 0)
        parts = []
        for t in ids.tolist():
            parts.append(img if t == cfg.image_token_id else emb(torch.tensor([t])))
        e = torch.cat(parts, 0)[None]
        fx["spliced_embeds"] = e.numpy()
        out = llm(inputs_embeds=e, output_hidden_states=True)
        fx["llm_hidden1"] = out.hidden_states[1].numpy()
        fx["llm_logits_last"] = out.logits[0, -1].numpy()
        fx["llm_logits_all_sha"] = np.float64(out.logits.double().abs().sum())
        gen = llm.generate(inputs_embeds=e, attention_mask=torch.ones(e.shape[:2], dtype=torch.long),
                           do_sample=False, max_new_tokens=n_new, min_new_tokens=n_new,
                           pad_token_id=0)
        fx["greedy_ids"] = gen[0].numpy().astype(np.int64)

        # training loss on a 2-sample padded (un-packed) batch with a key-padding mask, and per-sample losses.
        S = e.shape[1]
        n2 = S - 5
        e2 = torch.zeros(2, S, e.shape[2])
        e2[0] = e[0]
        e2[1, :n2] = e[0, 5:]
        am = torch.ones(2, S, dtype=torch.long)
        am[1, n2:] = 0
        g = torch.Generator().manual_seed(77 + seed)
        lab = torch.randint(0, cfg.llm.vocab_size - 2, (2, S), generator=g)
        lab[:, : S // 2] = -100
        lab[1, n2:] = -100
        n_items = int((lab[:, 1:] != -100).sum())
        out2 = llm(inputs_embeds=e2, attention_mask=am, labels=lab, num_items_in_batch=n_items)
        fx["train_embeds"] = e2.numpy()
        fx["train_mask"] = am.numpy()
        fx["train_labels"] = lab.numpy()
        fx["train_num_items"] = np.int64(n_items)
        fx["train_loss"] = np.float64(out2.loss.item())
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **fx)
    print(f"wrote {path}: greedy_ids={fx['greedy_ids'].tolist()} loss={fx['train_loss']:.6f} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)")


def golden_flat_square(out_dir):
    """Pin the space-to-depth ordering (incl. odd grids and zero padding) on integer-valued inputs."""
    bp = ref_projector()
    fx = {}
    for g in (4, 5, 7, 32):
        x = torch.arange(g * g * 3, dtype=torch.float32).reshape(1, g * g, 3) + 1
        fx[f"ds2_{g}"] = bp.DownSampleBlock()(x).numpy()
        fx[f"ds2fix_{g}"] = bp.DownSample2x2BlockFix()(x).numpy()
        fx[f"ds3fix_{g}"] = bp.DownSample3x3BlockFix()(x).numpy()
    path = os.path.join(out_dir, "flat_square.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(8)
    golden_flat_square(out_dir)
    golden_case("tiny_2x2", configs.tiny("mlp_downsample"), seed=0, n_text=12, n_new=8, out_dir=out_dir)
    golden_case("tiny_2x2fix", configs.tiny("mlp_downsample_2x2_fix", image=70), seed=1, n_text=9, n_new=6,
                out_dir=out_dir)
    golden_case("tiny_3x3_tied", configs.tiny("mlp_downsample_3x3_fix", tied=True), seed=2, n_text=7, n_new=6,
                out_dir=out_dir)


if __name__ == "__main__":
    main()
