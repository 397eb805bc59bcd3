"""REFERENCE-EXECUTED gradients of one SFT forward + backward with VIDEO media under the pooling encoder — TEST INFRASTRUCTURE.

Row a13 over row a7: the reference trains through `TSPVideoEncoder` (llava/model/encoders/video/tsp.py:14-64) with torch autograd.
This script runs that with the REAL code on a seeded tiny config: the reference's SigLIP and projector loaded by file path, HF
`Qwen2ForCausalLM`, and `pool` / `BasicVideoEncoder._process_features` / `TSPVideoEncoder._process_features` taken from their source files
unchanged (the ast extraction of oracle/make_golden_video.py — `llava.model` itself cannot be imported here: it pulls in deepspeed).  Only the
splice of llava_arch.py:412-490 is restated, to connect them: image token -> projector rows + "\\n", video token -> the encoder's block, media
labels IGNORE, right padding.  One batch: sample 0 = an image and a 4-frame video, sample 1 = a 4-frame video; two pool sizes (temporal +
spatial windows, and the unpooled frames), start / end / separator tokens.  Stored: the loss and, for EVERY parameter, the gradient's norm
and first 64 values.  tests/test_oracle_golden.py holds autograd through the oracle (`vlm_sft_loss(..., videos=, video_encoder=)`) to it;
the HIP step is held to the oracle (tests/test_gpu_train.py).

    python oracle/make_golden_grads_video.py       # seconds; writes tests/golden/tiny_sft_grads_video.npz; needs /root/reference
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import make_golden as G                     # noqa: E402
from oracle import make_golden_video as GV              # noqa: E402
from vila_amd import configs, synthetic                 # noqa: E402

IGNORE = -100
SEED = 9
POOLS = [[2, 2, 1], [1, 1, 1]]
START_IDS, END_IDS, SEP_IDS = [21, 22], [23], [24, 25]
OUT = os.path.join(ROOT, "tests", "golden", "tiny_sft_grads_video.npz")


def case():
    cfg = configs.tiny("mlp_downsample")
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, SEED).items()}
    px = synthetic.make_pixels(cfg, 9, SEED).to(torch.bfloat16).float()          # tile 0: the image; 1-4 and 5-8: the two videos' frames
    g = torch.Generator().manual_seed(SEED)
    L = 15
    ids = torch.randint(0, 900, (2, L), generator=g)
    ids[0, 0] = cfg.image_token_id
    ids[0, 4] = cfg.video_token_id
    ids[1, 2] = cfg.video_token_id
    mask = torch.ones((2, L), dtype=torch.bool)
    mask[1, 12:] = False
    labels = torch.randint(0, 900, (2, L), generator=g)
    labels[:, :6] = IGNORE
    labels[~mask] = IGNORE
    return cfg, w, px, ids, labels, mask


def main():
    torch.manual_seed(0)
    cfg, w, px, ids, labels, mask = case()
    ms, bp = G.ref_siglip(), G.ref_projector()
    ref = GV.load_reference_classes()
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
    emb = llm.model.embed_tokens
    # encode_images on the image, and on all frames of all videos at once (basic.py:43-53 / tsp.py:54-64)
    encode = lambda x: proj(tower(x, output_hidden_states=True).hidden_states[cfg.vision.select_layer])
    img_tokens = encode(px[:1])                                                 # [1, T, H]
    nl = emb(torch.tensor([cfg.newline_token_id]))
    frames = encode(px[1:])                                                     # [8, T, H]
    enc = ref["TSPVideoEncoder"](POOLS)
    start, end, sep = emb(torch.tensor(START_IDS)), emb(torch.tensor(END_IDS)), emb(torch.tensor(SEP_IDS))
    videos = [enc._process_features(f, start_token_embeds=start, end_token_embeds=end, sep_token_embeds=sep) for f in torch.split(frames, [4, 4])]
    media = {cfg.image_token_id: [torch.cat([img_tokens[0], nl], 0)], cfg.video_token_id: list(videos)}
    rows_e, rows_l = [], []
    for b in range(2):
        es, ls = [], []
        for t, lab, ok in zip(ids[b].tolist(), labels[b].tolist(), mask[b].tolist()):
            if not ok:
                continue
            if t in media:
                blk = media[t].pop(0)
                es.append(blk); ls += [IGNORE] * blk.shape[0]
            else:
                es.append(emb(torch.tensor([t]))); ls.append(lab)
        rows_e.append(torch.cat(es, 0)); rows_l.append(torch.tensor(ls))
    assert not media[cfg.image_token_id] and not media[cfg.video_token_id]
    S = max(r.shape[0] for r in rows_e)
    e = torch.stack([torch.cat([r, torch.zeros((S - r.shape[0], r.shape[1]))], 0) for r in rows_e], 0)
    lab = torch.full((2, S), IGNORE, dtype=torch.int64)
    am = torch.zeros((2, S), dtype=torch.long)
    for b in range(2):
        lab[b, : rows_l[b].numel()] = rows_l[b]
        am[b, : rows_l[b].numel()] = 1
    n_items = int((lab[:, 1:] != IGNORE).sum())
    out = llm(inputs_embeds=e, attention_mask=am, labels=lab, num_items_in_batch=n_items)
    out.loss.backward()
    fx = {"seed": np.int64(SEED), "hf_version": np.array(ver), "loss": np.float64(out.loss.item()), "num_items": np.int64(n_items),
          "input_ids": ids.numpy(), "labels": labels.numpy(), "mask": mask.numpy(), "pool_sizes": np.array(POOLS, dtype=np.int32),
          "start_ids": np.array(START_IDS), "end_ids": np.array(END_IDS), "sep_ids": np.array(SEP_IDS),
          "video_block_rows": np.array([int(v.shape[0]) for v in videos])}
    names = []
    for prefix, mod in (("vision_tower.vision_tower.", tower), ("mm_projector.", proj), ("llm.", llm)):
        for n, p in mod.named_parameters():
            name = prefix + n
            if name not in w:
                continue
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            fx[f"gn_{len(names)}"] = np.float64(g.double().norm())
            fx[f"gv_{len(names)}"] = g.reshape(-1)[:64].detach().numpy().astype(np.float32)
            names.append(name)
    # the encoder's own tokens: whole embedding-gradient rows (the first-64 slice of the table never reaches them)
    ge = emb.weight.grad
    fx["token_rows"] = np.array(START_IDS + END_IDS + SEP_IDS)
    fx["token_row_grads"] = ge[torch.tensor(START_IDS + END_IDS + SEP_IDS)].detach().numpy().astype(np.float32)
    fx["names"] = np.array(names)
    np.savez_compressed(OUT, **fx)
    print(f"wrote {OUT}: loss {out.loss.item():.6f}, {n_items} targets, {len(names)} gradient tensors, video blocks {fx['video_block_rows'].tolist()} rows "
          f"({os.path.getsize(OUT)} bytes)")


if __name__ == "__main__":
    main()
