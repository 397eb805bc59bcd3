"""The data-side producer of the SFT step (vila_amd/data.py): `DataCollator` against the REFERENCE'S OWN collator executed on integer-tagged
instances (oracle/make_golden_collate_cases.py -> tests/golden/collate_cases_ref.json), `build_instance` against the reference-executed pieces
it is made of.  Integer work: bit-exact."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle.make_golden_collate_cases import materialise, tokenizer
from vila_amd import configs, data

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fx():
    return json.load(open(os.path.join(GOLDEN, "collate_cases_ref.json")))


def test_collator_every_branch_equals_the_reference_collator(fx):
    assert {"single", "prebatched", "truncated", "sizes", "mismatch", "mismatch_s2"} <= set(fx["cases"])
    for name, rec in fx["cases"].items():
        coll = data.DataCollator(tokenizer(rec["model_max_length"]))
        insts = [materialise(i) for i in rec["instances"]]
        if "error" in rec:
            with pytest.raises(ValueError) as e:
                coll(insts)
            assert str(e.value) == rec["error"], name
            continue
        b = coll(insts)
        assert sorted(b) == rec["keys"], name
        assert b["input_ids"].tolist() == rec["input_ids"] and b["labels"].tolist() == rec["labels"], name
        assert b["attention_mask"].dtype == torch.bool and b["attention_mask"].tolist() == rec["attention_mask"], name
        assert [int(t) for t in b["media"]["image"]] == rec["image"] and [int(t) for t in b["media"]["video"]] == rec["video"], name
        blocks = [None if x is None else list(x) for x in b["media_config"]["image"]["block_sizes"]]
        assert blocks == rec["block_sizes"], name
        got_sizes = json.loads(json.dumps(b["media_config"]["image"]["original_image_sizes"]))
        assert got_sizes == rec["original_image_sizes"], name
        assert b["media_config"]["video"] == rec["video_config"] == {} and b["gt_selection_maps"] is None and rec["gt_selection_maps"] is None


def test_collator_reproduces_the_gpu_tests_reference_batch():
    """The batch tests/test_gpu_integration.py feeds the HIP model (collate_batch_ref.npz, produced by the reference's collator) comes out of
    OUR collator too, object for object."""
    from oracle.make_golden_collate import PAD_ID, instances
    ref = np.load(os.path.join(GOLDEN, "collate_batch_ref.npz"))
    cfg = configs.tiny_s2()
    tok = SimpleNamespace(media_tokens={"image": "<image>", "video": "<vila/video>"},
                          media_token_ids={"image": cfg.image_token_id, "video": cfg.video_token_id}, pad_token_id=PAD_ID, model_max_length=64)
    inst, pool = instances(cfg)
    b = data.DataCollator(tok)(inst)
    assert np.array_equal(b["input_ids"].numpy(), ref["input_ids"]) and np.array_equal(b["labels"].numpy(), ref["labels"])
    assert np.array_equal(b["attention_mask"].numpy(), ref["attention_mask"])
    where = {pool[k].data_ptr(): k for k in range(pool.shape[0])}
    assert [where[t.data_ptr()] for t in b["media"]["image"]] == ref["image_pool_index"].tolist()
    assert [where[v[0].data_ptr()] for v in b["media"]["video"]] == ref["video_first_pool_index"].tolist()
    assert [[-1, -1] if x is None else list(x) for x in b["media_config"]["image"]["block_sizes"]] == ref["block_sizes"].tolist()


def test_collator_grounding_maps_stack_or_refuse():
    tok = tokenizer(16)
    a = {"input_ids": torch.tensor([1, 2]), "labels": torch.tensor([1, 2]), "gt_selection_map": torch.ones(2, 2)}
    b = {"input_ids": torch.tensor([3]), "labels": torch.tensor([3]), "gt_selection_map": torch.zeros(2, 2)}
    assert data.DataCollator(tok)([a, b])["gt_selection_maps"].shape == (2, 2, 2)
    with pytest.raises(AssertionError):
        data.DataCollator(tok)([a, {"input_ids": torch.tensor([3]), "labels": torch.tensor([3])}])


def test_build_instance_under_every_recipe_and_through_the_collator():
    pytest.importorskip("tokenizers")
    from oracle.make_golden_conversation import build_tokenizer
    from oracle.make_golden_s2_tiles import synthetic_image
    from vila_amd import conversation as C
    from vila_amd import serving
    cfx = json.load(open(os.path.join(GOLDEN, "conversation_ref.json")))
    dfx = np.load(os.path.join(GOLDEN, "dynamic_tiles.npz"))
    tok = C.prepare_tokenizer(build_tokenizer(cfx["tokenizer"]), cfx["chat_template_name"])
    tok.model_max_length = 4096
    size = 448
    imgs = [synthetic_image(w, h, 200 + i) for i, (w, h) in enumerate([(640, 480), (300, 900), (448, 448)])]
    conv = [{"from": "human", "value": ["A ", imgs[0], " B ", imgs[2], " C"]}, {"from": "gpt", "value": "a red square"}]
    # dynamic (NVILA-Lite): the text is the reference-executed one for the same two pictures, one tile tensor per `<image>`
    cfg = configs.nvila_lite_3b()
    cfg.image_aspect_ratio = "dynamic"
    inst = data.build_instance(conv, cfg, tok)
    want_text = str(dfx["prompt2"])
    assert list(inst["image"].shape) == dfx["prompt2_shape"].tolist()
    want = C.preprocess_conversation([{"from": "human", "value": want_text}, {"from": "gpt", "value": "a red square"}], tok)
    assert torch.equal(inst["input_ids"], want["input_ids"]) and torch.equal(inst["labels"], want["labels"])
    n_img = int((inst["input_ids"] == tok.media_token_ids["image"]).sum())
    assert n_img == inst["image"].shape[0] == 13 + 1 and inst["original_image_sizes"] == [(640, 480), (448, 448)]
    assert "block_sizes" not in inst and conv[0]["value"][0] == "A "                       # the caller's conversation is left alone
    # dynamic_s2 (NVILA): tiles of every scale + one block size per picture, one `<image>` per PICTURE
    s2 = configs.nvila_8b_s2()
    inst2 = data.build_instance(conv, s2, tok)
    (r0, c0), (r1, c1) = inst2["block_sizes"]
    assert inst2["image"].shape[0] == (1 + 4 + r0 * c0) + (1 + 4 + r1 * c1) and int((inst2["input_ids"] == tok.media_token_ids["image"]).sum()) == 2
    t0, b0 = serving.process_image(imgs[0], s2, enable_dynamic_s2=True)
    assert b0 == (r0, c0) and torch.equal(inst2["image"][: t0.shape[0]], t0)
    # plain: whole pictures; text only: no media keys at all
    inst3 = data.build_instance(conv, configs.nvila_8b(), tok)
    assert inst3["image"].shape == (2, 3, size, size) and torch.equal(inst3["image"][1], serving.preprocess_image(imgs[2], size))
    inst4 = data.build_instance([{"from": "human", "value": "hello"}, {"from": "gpt", "value": "hello again!"}], cfg, tok)
    assert set(inst4) == {"input_ids", "labels"}
    # ... and the instances collate into one batch: media in row order, block sizes flattened, masks from the pad id
    tok.pad_token_id = tok.convert_tokens_to_ids("<|endoftext|>")
    batch = data.DataCollator(tok)([inst2, inst4, {k: v for k, v in inst3.items()}])
    assert batch["input_ids"].shape[0] == 3 and len(batch["media"]["image"]) == inst2["image"].shape[0] + 2
    assert batch["media_config"]["image"]["block_sizes"] == [(r0, c0), (r1, c1), None, None]
    assert batch["media_config"]["image"]["original_image_sizes"] == [(640, 480), (448, 448), (640, 480), (448, 448)]
    assert torch.equal(batch["attention_mask"], batch["input_ids"] != tok.pad_token_id)
