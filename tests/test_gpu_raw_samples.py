"""Raw samples -> loss on the HIP model: conversation + pictures -> `data.build_instance` (aspect recipe, chat template, SFT labels) ->
`data.DataCollator` -> `HipLlavaLlamaModel(**batch)`, in eval mode and through the autograd seam, against the fp32 oracle on the same batch.
The host pieces are pinned bit-exactly to the reference's own functions on CPU (tests/test_{dynamic_tiler,conversation,data}_cpu.py); this
is the end-to-end leg under the two tiling recipes the NVILA scripts use (`dynamic`: NVILA-Lite, `dynamic_s2`: NVILA)."""
import json
import os

import pytest
import torch

from oracle import vila_oracle as O
from vila_amd import configs, data, synthetic

pytestmark = pytest.mark.gpu


def _tokenizer(cfg):
    """The stand-in BPE tokenizer of the conversation fixture (Qwen2 chat template, media tokens added); the config follows ITS ids, as
    `checkpoint.load_tokenizer` makes a loaded model do."""
    pytest.importorskip("tokenizers")
    from oracle.make_golden_conversation import build_tokenizer
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "conversation_ref.json")))
    from vila_amd.conversation import prepare_tokenizer
    tok = prepare_tokenizer(build_tokenizer(fx["tokenizer"]), fx["chat_template_name"])
    tok.model_max_length = 512
    nl = tok("\n").input_ids
    assert len(nl) == 1 and max(tok.media_token_ids.values()) + 2 < cfg.llm.vocab_size
    cfg.image_token_id, cfg.video_token_id, cfg.newline_token_id = tok.media_token_ids["image"], tok.media_token_ids["video"], nl[0]
    cfg.llm.eos_token_id = tok.eos_token_id
    return tok


@pytest.mark.parametrize("recipe", ["dynamic", "dynamic_s2"])
def test_raw_samples_to_loss_on_the_hip_model(recipe):
    from oracle.make_golden_s2_tiles import synthetic_image
    from vila_amd.vlm import build_model
    cfg = configs.tiny_s2() if recipe == "dynamic_s2" else configs.tiny("mlp_downsample")
    cfg.image_aspect_ratio, cfg.min_tiles, cfg.max_tiles = recipe, 1, (12 if recipe == "dynamic_s2" else 4)
    tok = _tokenizer(cfg)
    size = cfg.vision.image_size
    wide, square = synthetic_image(4 * size, 2 * size, 1), synthetic_image(size + 9, size + 9, 2)
    convs = [[{"from": "human", "value": [wide, "what is in this picture ?"]}, {"from": "gpt", "value": "a red square on a blue table"}],
             [{"from": "human", "value": "hello"}, {"from": "gpt", "value": "hello again!"}],
             [{"from": "human", "value": ["describe the image ", square]}, {"from": "gpt", "value": "there are two cats"},
              {"from": "human", "value": "and ?"}, {"from": "gpt", "value": "one dog"}]]
    inst = [data.build_instance(c, cfg, tok) for c in convs]
    if recipe == "dynamic":
        assert inst[0]["image"].shape[0] == 3 and inst[2]["image"].shape[0] == 1            # 2 x 1 grid + thumbnail; one tile
    else:
        assert inst[0]["block_sizes"] == [(2, 5)] and inst[0]["image"].shape[0] == 1 + 4 + 10      # the 2:1 picture: 5 x 2 tiles at the last scale
    batch = data.DataCollator(tok)(inst)
    blocks = batch["media_config"]["image"]["block_sizes"]
    ids, labels, mask = batch["input_ids"], batch["labels"], batch["attention_mask"]
    assert int((labels != -100).sum()) > 10 and int((ids == cfg.image_token_id).sum()) == (4 if recipe == "dynamic" else 2)
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 31).items()}
    tiles = [t.to(torch.bfloat16).float() for t in batch["media"]["image"]]
    ref = float(O.vlm_sft_loss(tiles, ids, labels, mask, w, cfg, packed=True, block_sizes=blocks if recipe == "dynamic_s2" else None))
    model = build_model(cfg, weights=w)
    model.tokenizer = tok
    dev = {"input_ids": ids.cuda(), "labels": labels.cuda(), "attention_mask": mask.cuda(),
           "media": {"image": [t.to(torch.bfloat16).cuda() for t in batch["media"]["image"]], "video": []},
           "media_config": batch["media_config"]}
    with torch.no_grad():
        ev = model(**dev)
    assert abs(float(ev.loss) - ref) < 1e-2 * abs(ref), (float(ev.loss), ref)
    model.enable_autograd(use_c_abi=False)
    model.train()
    out = model(**dev)
    assert out.loss.requires_grad and abs(float(out.loss.detach()) - ref) < 1e-2 * abs(ref), (float(out.loss.detach()), ref)
    out.loss.backward()
    g = dict(model.llm.named_parameters())["model.embed_tokens.weight"].grad
    assert g is not None and float(g.float().norm()) > 0
    print(f"{recipe}: eval loss {float(ev.loss):.5f}, training loss {float(out.loss.detach()):.5f}, oracle {ref:.5f}")
