"""Host logic of an SFT run (vila_amd/run.py) against tests/golden/run_ref.json = the reference's own `VILADistributedSampler` and
`get_checkpoint_path` (ast-extracted, executed) and transformers' own scheduler (oracle/make_golden_run.py).  Integer work: bit-exact; learning
rates: to 1e-12 relative.  The loop itself runs over a recording stub of `SFTTrainer` (no kernels on CPU)."""
import json
import os

import pytest
import torch

from vila_amd import run

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fx():
    return json.load(open(os.path.join(GOLDEN, "run_ref.json")))


def test_sampler_order_equals_the_reference_sampler_on_every_rank_and_epoch(fx):
    assert len(fx["samplers"]) >= 8
    for c in fx["samplers"]:
        seen = []
        for rank, rec in enumerate(c["ranks"]):
            s = run.VILADistributedSampler(sum(c["lens"]), c["world"], rank, seed=c["seed"], batch_size=c["batch_size"], sample_len_list=c["lens"],
                                           gradient_accumulation_steps=c["accumulation"])
            assert len(s) == rec["len"], c
            for epoch, want in enumerate(rec["order"]):
                s.set_epoch(epoch)
                assert list(s) == want, (c["lens"], c["world"], rank, epoch)
            seen += rec["order"][0]
        assert len(set(seen)) == len(seen)                               # the ranks' shares are disjoint
        assert len(seen) % (c["world"] * c["batch_size"] * c["accumulation"]) == 0


def test_sampler_refusals():
    with pytest.raises(ValueError, match="Invalid rank 2"):
        run.VILADistributedSampler(10, 2, 2)
    with pytest.raises(NotImplementedError):
        run.VILADistributedSampler(10, 2, 0, sp_degree=2)


def test_learning_rate_of_every_update_equals_the_transformers_scheduler(fx):
    kinds = set()
    for c in fx["schedules"]:
        n_warm = run.warmup_steps(c["total"], c["warmup_ratio"], c["warmup_steps"])
        assert n_warm == c["n_warmup"]
        for k, want in enumerate(c["lr"]):
            got = fx["base_lr"] * run.lr_factor(c["kind"], k, n_warm, c["total"])
            assert abs(got - want) <= 1e-12 * fx["base_lr"], (c["kind"], c["total"], k, got, want)
        kinds.add(c["kind"])
    assert kinds == {"cosine", "linear", "constant", "constant_with_warmup"}


def test_get_checkpoint_path_equals_the_reference_rule(fx, tmp_path):
    from oracle.make_golden_run import checkpoint_layouts
    for name, (dirs, files) in checkpoint_layouts().items():
        r = tmp_path / name
        r.mkdir()
        for d in dirs:
            (r / d).mkdir()
        for f in files:
            (r / f).write_text("")
        path, cont = run.get_checkpoint_path(str(r))
        want = fx["checkpoints"][name]
        assert (None if path is None else os.path.relpath(path, str(r))) == want["path"] and cont == want["continue"], name
    assert run.get_checkpoint_path(str(tmp_path / "nope")) == (None, True)


# ------------------------------------------------------------------------------------------------------------ the loop over a stub trainer
class _Stub:
    """Records what a run feeds the step; its 'weights' are a running checksum so a resumed run can be compared with an uninterrupted one."""
    def __init__(self):
        self.lr, self.calls, self.w = None, [], 0.0

    def step(self, input_ids, images, labels, attention_mask, block_sizes=None, videos=None):
        self.calls.append((self.lr, input_ids.flatten().tolist(), len(images), block_sizes, videos))
        self.w = self.w * 0.5 + self.lr * float(input_ids.sum())
        return float(len(self.calls))


def _dataset(n):
    return [{"input_ids": torch.tensor([i, i + 1000]), "labels": torch.tensor([-100, i]), "image": [torch.tensor([i])] if i % 3 == 0 else []} for i in range(n)]


def _collate(insts):
    ids = torch.stack([x["input_ids"] for x in insts])
    return {"input_ids": ids, "labels": torch.stack([x["labels"] for x in insts]), "attention_mask": torch.ones_like(ids, dtype=torch.bool),
            "media": {"image": [t for x in insts for t in x["image"]], "video": []}, "media_config": {"image": {"block_sizes": None}}}


def _io():
    def save(tr, folder):
        json.dump({"w": tr.w}, open(os.path.join(folder, "stub.json"), "w"))
    def load(tr, folder):
        tr.w = json.load(open(os.path.join(folder, "stub.json")))["w"]
    return save, load


def _final(tr, folder):
    json.dump({"w": tr.w}, open(os.path.join(folder, "config.json"), "w"))


def test_run_feeds_the_step_sampler_batches_and_schedule_and_rotates_checkpoints(tmp_path):
    ds = _dataset(45)
    args = run.TrainArgs(output_dir=str(tmp_path / "r"), per_device_train_batch_size=2, num_train_epochs=2, save_steps=4, save_total_limit=2, seed=3,
                         sample_lens=[30, 15])
    save, load = _io()
    tr = _Stub()
    logs = []
    st = run.train(tr, ds, _collate, args, rank=1, world_size=2, save_fn=save, load_fn=load, final_save_fn=_final, log=logs.append)
    s = run.VILADistributedSampler(45, 2, 1, seed=3, batch_size=2, sample_len_list=[30, 15])
    per_epoch, epochs, total = run.plan(len(s), args)
    assert (per_epoch, epochs, total) == (len(s) // 2, 2, 2 * (len(s) // 2)) and st.global_step == total == len(tr.calls)
    want = []
    for e in range(2):
        s.set_epoch(e)
        o = list(s)
        want += [o[i:i + 2] for i in range(0, len(o), 2)]
    n_warm = run.warmup_steps(total, 0.03)
    for k, (lr, ids, n_img, blocks, vids) in enumerate(tr.calls):
        assert ids == [want[k][0], want[k][0] + 1000, want[k][1], want[k][1] + 1000]
        assert n_img == sum(1 for i in want[k] if i % 3 == 0) and blocks is None and vids is None
        assert lr == args.learning_rate * run.lr_factor("cosine", k, n_warm, total)
    assert tr.calls[0][0] == 0.0 and logs == []                        # HF: the first update of a warmed-up run has rate 0; rank 1 does not log
    assert [r["step"] for r in st.log_history] == list(range(1, total + 1)) and st.log_history[0]["learning_rate"] > 0
    assert not os.path.exists(args.output_dir)                         # rank 1 writes nothing

    tr0 = _Stub()
    run.train(tr0, ds, _collate, args, rank=0, world_size=2, save_fn=save, load_fn=load, resume=False, final_save_fn=None)
    kept = sorted(os.listdir(args.output_dir), key=lambda d: int(d.split("-")[1]))
    assert kept == [f"checkpoint-{k}" for k in range(4, total + 1, 4)][-2:]               # save_total_limit = 2, no staging folder left behind
    assert json.load(open(os.path.join(args.output_dir, kept[-1], "trainer_state.json")))["global_step"] == int(kept[-1].split("-")[1])


def test_resumed_run_replays_neither_a_batch_nor_a_learning_rate(tmp_path):
    ds = _dataset(40)
    save, load = _io()
    mk = lambda d, **kw: run.TrainArgs(output_dir=str(tmp_path / d), per_device_train_batch_size=4, num_train_epochs=3, save_steps=7, save_total_limit=None,
                                       warmup_ratio=0.1, **kw)
    whole = _Stub()
    st = run.train(whole, ds, _collate, mk("a"), save_fn=save, load_fn=load, final_save_fn=_final)
    assert st.global_step == 30 and json.load(open(tmp_path / "a" / "config.json"))["w"] == whole.w
    # the same run killed after 17 updates (its last checkpoint is 14) ...
    first = _Stub()
    run.train(first, ds, _collate, mk("b", max_steps=17), save_fn=save, load_fn=load, final_save_fn=None)
    assert sorted(os.listdir(tmp_path / "b")) == ["checkpoint-14", "checkpoint-7"]
    # ... continues from update 14 in the middle of epoch 1 with the schedule of the 30-update run
    second = _Stub()
    st2 = run.train(second, ds, _collate, mk("b"), save_fn=save, load_fn=load, final_save_fn=_final)
    assert st2.global_step == 30 and len(second.calls) == 16
    assert [c[1] for c in second.calls] == [c[1] for c in whole.calls[14:]]
    # (the killed run was planned for 17 updates, so ITS rates differ; the resumed run's are the uninterrupted run's)
    assert [c[0] for c in second.calls] == [c[0] for c in whole.calls[14:]]
    assert [r["step"] for r in st2.log_history] == list(range(1, 31))
    # a finished run (the final model's config.json lies in the run folder) is not trained again (train.py:503-507)
    third, said = _Stub(), []
    st3 = run.train(third, ds, _collate, mk("b"), save_fn=save, load_fn=load, final_save_fn=_final, log=said.append)
    assert third.calls == [] and st3.global_step == 30 and "Skipp training" in said[0]["message"]


def test_a_mixture_below_one_global_batch_is_refused():
    with pytest.raises(ValueError, match="does not fill one global batch"):
        run.train(_Stub(), _dataset(7), _collate, run.TrainArgs(per_device_train_batch_size=4), world_size=2, final_save_fn=None)


def test_accumulation_groups_consecutive_batches_into_one_update(tmp_path):
    class Acc(_Stub):
        max_grad_norm = None
        def step_accumulated(self, micro):
            self.calls.append((self.lr, [m["input_ids"].flatten().tolist() for m in micro], self.max_grad_norm))
            return 1.0
    ds = _dataset(50)
    args = run.TrainArgs(output_dir=str(tmp_path / "r"), per_device_train_batch_size=2, gradient_accumulation_steps=3, num_train_epochs=1, save_steps=0,
                         max_grad_norm=5.0)
    tr = Acc()
    st = run.train(tr, ds, _collate, args, rank=0, world_size=2, final_save_fn=None)
    s = run.VILADistributedSampler(50, 2, 0, seed=42, batch_size=2, gradient_accumulation_steps=3)
    o = list(s)
    assert len(o) == 24 and st.global_step == 4 == len(tr.calls)                      # 50 // (2 ranks x 2 x 3) = 4 updates of 3 micro-batches each
    flat = [i for _, micro, _ in tr.calls for m in micro for i in m[::2]]
    assert flat == o and all(len(micro) == 3 and clip == 5.0 for _, micro, clip in tr.calls)
    assert tr.calls[0][0] == 0.0 and tr.calls[1][0] == 2e-5                           # ceil(0.03 x 4) = 1 warm-up update


def test_default_checkpoint_functions_round_trip_a_real_trainer_on_cpu(tmp_path):
    """The default save / load of a run (weights in the reference's three folders + optimizer state) over a real `SFTTrainer` (CPU tensors, no
    kernels): a trainer resumed from `checkpoint-<k>` holds the saved masters, moments, step counts and bf16 parameters."""
    from vila_amd import configs
    from vila_amd.train import SFTTrainer
    from vila_amd.vlm import HipLlavaLlamaModel
    torch.manual_seed(0)
    tr = SFTTrainer(HipLlavaLlamaModel(configs.tiny("mlp_downsample"), device="cpu"))
    with torch.no_grad():
        tr.flat.master.normal_(0, 0.02); tr.flat.params.copy_(tr.flat.master); tr.flat.m.normal_(0, 1e-3); tr.flat.v.uniform_(0, 1e-4)
    tr.flat.step_count, tr.flat.bucket_steps = 5, {"mm_projector.": 5}
    args = run.TrainArgs(output_dir=str(tmp_path / "r"))
    st = run.TrainerState(global_step=5, epoch=0.5, max_steps=10)
    run._checkpoint(tr, args, st, 0, run._save_checkpoint_default, None)
    path, cont = run.get_checkpoint_path(args.output_dir)
    assert path.endswith("checkpoint-5") and cont and sorted(os.listdir(path)) == ["config.json", "llm", "mm_projector", "optimizer", "trainer_state.json", "vision_tower"]
    tr2 = SFTTrainer(HipLlavaLlamaModel(configs.tiny("mlp_downsample"), device="cpu"))
    run._load_checkpoint_default(tr2, path)
    for a, b in ((tr.flat.master, tr2.flat.master), (tr.flat.m, tr2.flat.m), (tr.flat.v, tr2.flat.v), (tr.flat.params, tr2.flat.params)):
        assert torch.equal(a, b)
    assert tr2.flat.step_count == 5 and tr2.flat.bucket_steps == {"mm_projector.": 5}
    assert run.TrainerState.load(os.path.join(path, "trainer_state.json")).global_step == 5


# ------------------------------------------------------------------------------------------------------------ datasets on disk -> instances
def test_media_token_stripping_mixture_parsing_and_global_batch_padding_equal_the_reference(fx):
    from vila_amd import data
    d = fx["datasets"]
    assert len(d["strip"]) >= 8 and all(data.remove_media_tokens(t) == want for t, want in d["strip"])
    assert all(data.parse_mixture(m, d["mixtures"]) == want for m, want in d["parse"])
    for c in d["pad"]:
        times, extra = data.pad_to_global_batch(c["n"], c["global_batch_size"])
        assert c["n"] * times + extra == c["len"], c
        assert ([[c["n"] * times, extra]] if extra else []) == c["drawn"], c             # what the reference asked `random.sample` for


def _write_media(root, records):
    from PIL import Image
    sizes = {}
    for r in records:
        for key in ("image", "images"):
            for p in ([r[key]] if isinstance(r.get(key), str) else r.get(key, [])):
                if p not in sizes:
                    sizes[p] = (8 + 2 * len(sizes), 6 + len(sizes))                       # every file has its own size: the picture names its path
                    os.makedirs(os.path.dirname(os.path.join(root, p)) or root, exist_ok=True)
                    Image.new("RGB", sizes[p], (10 * len(sizes), 0, 0)).save(os.path.join(root, p))
    return sizes


def test_llava_dataset_process_equals_the_reference_class_on_every_record(fx, tmp_path):
    from vila_amd import data
    d = fx["datasets"]
    root = str(tmp_path / "media")
    os.makedirs(root)
    sizes = _write_media(root, d["records"])
    by_size = {v: k for k, v in sizes.items()}
    jpath = str(tmp_path / "set.json")
    json.dump(d["records"], open(jpath, "w"))
    for k, want in enumerate(d["process"]):
        ds = data.LLaVADataset(jpath, root, cfg=None, tokenizer=None, max_num_images=want["max_num_images"], resample_on_failure=False)
        rec = d["records"][k % len(d["records"])]
        if "error" in want:
            with pytest.raises(ValueError) as e:
                ds.process(rec)
            assert str(e.value) == want["error"]
            continue
        got = ds.process(rec)
        assert rec == d["records"][k % len(d["records"])]                                # the record itself is left alone (deepcopy)
        flat = [{"from": m["from"], "value": [("media/root/" + by_size[v.size]) if not isinstance(v, str) else v for v in m["value"]]
                 if isinstance(m["value"], list) else m["value"]} for m in got]
        assert flat == want["messages"], (k, flat, want["messages"])


def test_mixture_on_disk_to_batches_through_the_run(tmp_path):
    """json files + pictures -> build_dataset -> sampler (balanced by `sample_lens`) -> DataCollator -> the step's keyword arguments: the whole
    input side of `run.train` over the stand-in tokenizer of the conversation fixture."""
    pytest.importorskip("tokenizers")
    from oracle.make_golden_conversation import build_tokenizer
    from vila_amd import configs, data
    from vila_amd.conversation import prepare_tokenizer
    cfx = json.load(open(os.path.join(GOLDEN, "conversation_ref.json")))
    tok = prepare_tokenizer(build_tokenizer(cfx["tokenizer"]), cfx["chat_template_name"])
    tok.model_max_length = 256
    cfg = configs.tiny("mlp_downsample")
    cfg.image_token_id, cfg.video_token_id = tok.media_token_ids["image"], tok.media_token_ids["video"]
    cfg.image_aspect_ratio = "resize"
    root = str(tmp_path / "m")
    os.makedirs(root)
    recs_a = [{"image": f"a{i}.png", "conversations": [{"from": "human", "value": "<image>\nwhat is this ?"}, {"from": "gpt", "value": f"a red square {i}"}]} for i in range(5)]
    recs_b = [{"conversations": [{"from": "human", "value": "hello"}, {"from": "gpt", "value": "hello again!"}]} for _ in range(3)]
    _write_media(root, recs_a)
    json.dump(recs_a, open(tmp_path / "a.json", "w")); json.dump(recs_b, open(tmp_path / "b.json", "w"))
    registry = {"pics": {"_target_": "llava.data.LLaVADataset", "data_path": str(tmp_path / "a.json"), "media_dir": root},
                "chat": {"_target_": "llava.data.LLaVADataset", "data_path": str(tmp_path / "b.json"), "media_dir": None}}
    ds = data.build_dataset("pics+chat*2", registry, cfg, tok, global_batch_size=4, seed=0)
    assert ds.sample_lens == [8, 8] and len(ds) == 16                    # sorted names: chat*2 (3 -> 4 padded, twice), pics (5 -> 8 padded)
    with pytest.raises(ValueError, match="'nope' is not found"):
        data.build_dataset("nope", registry, cfg, tok)
    seen = []

    class Rec:
        lr = None
        def step(self, **kw):
            seen.append(kw)
            return 0.5
    args = run.TrainArgs(output_dir=str(tmp_path / "out"), per_device_train_batch_size=2, num_train_epochs=1, save_steps=0, sample_lens=ds.sample_lens)
    st = run.train(Rec(), ds, data.DataCollator(tok), args, rank=0, world_size=2, final_save_fn=None)
    assert st.global_step == 4 and len(seen) == 4
    n_img = sum(len(kw["images"]) for kw in seen)
    assert n_img == 4 and all(kw["input_ids"].shape[0] == 2 and kw["attention_mask"].dtype == torch.bool for kw in seen)     # rank 0's half of each dataset
    for kw in seen:
        assert int((kw["input_ids"] == cfg.image_token_id).sum()) == len(kw["images"]) and all(t.shape == (3, 56, 56) for t in kw["images"])
        assert int((kw["labels"] != -100).sum()) > 0 and kw["videos"] is None


# ------------------------------------------------------------------------------------------------------------ the optimizer's weight-decay groups
@pytest.mark.parametrize("kind", ["mlp_downsample", "mlp_downsample_3x3_fix"])
def test_weight_decay_groups_equal_the_reference_create_optimizer_rule(fx, kind):
    """`FlatParams.decays` against the names the reference's own two statements (llava_trainer.py:494-495, executed over the reference SigLIP /
    projector + HF Qwen2) put into the decaying group: biases and LayerNorm weights out, Qwen2's RMSNorm weights IN (they are not nn.LayerNorm)."""
    from vila_amd import configs
    from vila_amd.train import FlatParams
    from vila_amd.vlm import HipLlavaLlamaModel
    ref = fx["decay"][kind]
    assert ref["layernorm_layers"] == ["LayerNorm"]
    flat = FlatParams(HipLlavaLlamaModel(configs.tiny(kind), device="cpu"), with_optimizer_state=False)
    ours = set(flat.index)
    theirs = {n for n in ref["all"] if ".vision_model.head." not in n}            # the pooling head VILA never builds (vision_use_head = false)
    assert ours - {"llm.lm_head.weight"} == theirs - {"llm.lm_head.weight"}        # (a tied head appears once in named_parameters)
    decay = set(ref["decay"])
    for n in ours & theirs:
        assert flat.decays(n) == (n in decay), n
    assert sum(flat.decays(n) for n in ours) >= 40 and any(not flat.decays(n) and "bias" not in n for n in ours)
    # the runs a bucket is cut into cover it exactly, in order, with alternating flags
    for pre in ("llm.model.layers.0.", "mm_projector.", "vision_tower.vision_tower.vision_model.encoder.layers.1."):
        a, b = flat.span(pre)
        runs = flat.decay_runs(pre)
        assert runs[0][0] == a and runs[-1][1] == b and all(r[1] == s[0] and r[2] != s[2] for r, s in zip(runs, runs[1:]))
        for n, (o, k, _) in flat.index.items():
            if n.startswith(pre):
                assert any(ra <= o and o + k <= rb and d == flat.decays(n) for ra, rb, d in runs), n


def test_adamw_with_weight_decay_equals_torch_adamw_with_the_reference_groups():
    """One update of every bucket through `SFTTrainer._adamw_bucket` (kernel stubbed by its torch restatement) == torch.optim.AdamW over two
    parameter groups built by the reference rule; with `decay_groups=False` everything decays (the pre-round-4 behaviour, kept as a switch)."""
    from tests import test_train_cpu as TC
    from vila_amd import ops
    from vila_amd.train import SFTTrainer
    orig = ops.adamw_step
    ops.adamw_step = TC._adamw_reference
    try:
        torch.manual_seed(0)
        tr = SFTTrainer(TC._tiny_model(), lr=1e-2, weight_decay=0.1)
        tr.flat.grads = torch.randn(tr.flat.numel, generator=torch.Generator().manual_seed(3))
        names = [n for n in tr.flat.index]
        ps = {n: torch.nn.Parameter(tr.flat.master[o:o + k].clone()) for n, (o, k, _) in tr.flat.index.items()}
        for n, (o, k, _) in tr.flat.index.items():
            ps[n].grad = tr.flat.grads[o:o + k].clone()
        opt = torch.optim.AdamW([{"params": [ps[n] for n in names if tr.flat.decays(n)], "weight_decay": 0.1},
                                 {"params": [ps[n] for n in names if not tr.flat.decays(n)], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
        opt.step()
        order = TC._bucket_order(tr.cfg)
        for pre in (order[1:] if tr.cfg.llm.tie_word_embeddings else order):
            tr._adamw_bucket(pre, 1.0)
        touched = [n for n in names if any(n.startswith(p) for p in order)]
        assert len(touched) > 60
        for n in touched:
            o, k, _ = tr.flat.index[n]
            assert torch.allclose(tr.flat.master[o:o + k], ps[n].detach(), atol=1e-7, rtol=1e-6), n
        n_bias = next(n for n in touched if n.endswith("q_proj.bias"))
        o, k, _ = tr.flat.index[n_bias]
        before = torch.nn.Parameter(tr.flat.master[o:o + k].clone())
        tr2 = SFTTrainer(TC._tiny_model_seeded(), lr=1e-2, weight_decay=0.1, decay_groups=False)
        tr2.flat.grads = tr.flat.grads.clone()
        tr2._adamw_bucket("llm.model.layers.0.", 1.0)
        o2, k2, _ = tr2.flat.index["llm.model.layers.0.self_attn.q_proj.bias"]
        tr3 = SFTTrainer(TC._tiny_model_seeded(), lr=1e-2, weight_decay=0.1)
        tr3.flat.grads = tr.flat.grads.clone()
        tr3._adamw_bucket("llm.model.layers.0.", 1.0)
        assert not torch.equal(tr2.flat.master[o2:o2 + k2], tr3.flat.master[o2:o2 + k2])       # the bias decays only without the groups
        ow, kw, _ = tr2.flat.index["llm.model.layers.0.mlp.down_proj.weight"]
        assert torch.equal(tr2.flat.master[ow:ow + kw], tr3.flat.master[ow:ow + kw])
    finally:
        ops.adamw_step = orig
