"""REFERENCE-EXECUTED fixture for the host logic of an SFT run (vila_amd/run.py) — TEST INFRASTRUCTURE.

  sampler      `VILADistributedSampler` (llava/train/llava_trainer.py:131-279) taken out of its file with `ast` and executed unchanged: the index
               order every rank sees, for mixtures of 1-4 datasets, 1-8 ranks, two epochs, several batch sizes / accumulation counts
  schedule     transformers' own `get_scheduler` (the object `Trainer.create_scheduler` builds for `--lr_scheduler_type ... --warmup_ratio ...`,
               scripts/NVILA-Lite/sft.sh:41-44) stepped over a dummy optimizer: the learning rate of every update
  checkpoints  `get_checkpoint_path` (llava/train/utils.py:59-79) executed over five run-folder layouts

  datasets     the front of the run: `_remove_media_tokens` (llava/data/dataset_impl/utils.py:10-13), `parse_mixture` (llava/data/builder.py:58-62)
               and `LLaVADataset` (dataset_impl/llava.py:16-74: the global-batch padding arithmetic of `__init__`, `process` on six records) —
               ast-extracted and executed over stand-in `Image` / `BaseDataset` objects that only record what they are given

  decay        the two statements of `LLaVATrainer.create_optimizer` that choose the weight-decay group (llava/train/llava_trainer.py:494-495),
               taken out of the method with `ast` and executed over the reference's own modules (reference SigLIP + projector, HF Qwen2) held
               under the names the VLM gives them: the names that decay

    python oracle/make_golden_run.py        # writes tests/golden/run_ref.json; needs /root/reference
"""
from __future__ import annotations

import ast
import json
import math
import os
import pathlib
import random
import re
import sys
import tempfile
import types
from typing import Optional

import torch
import torch.distributed as dist
from torch.utils.data import DistributedSampler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/llava"
OUT = os.path.join(ROOT, "tests", "golden", "run_ref.json")

SAMPLER_CASES = [  # (dataset lengths, world, per-device batch, accumulation, seed)
    ([37], 1, 1, 1, 0), ([64], 2, 4, 1, 42), ([100, 31], 2, 2, 1, 42), ([257, 64, 19], 4, 2, 1, 7), ([1000, 333, 90, 12], 8, 2, 2, 42),
    ([50, 50], 3, 4, 1, 1), ([9, 200], 2, 3, 2, 5), ([5, 120], 4, 2, 1, 3),
]
SCHEDULE_CASES = [  # (kind, total updates, warmup ratio, explicit warmup steps)
    ("cosine", 100, 0.03, 0), ("cosine", 17, 0.03, 0), ("cosine", 1, 0.03, 0), ("cosine", 250, 0.0, 0), ("cosine", 64, 0.1, 5),
    ("linear", 40, 0.05, 0), ("constant_with_warmup", 30, 0.1, 0), ("constant", 10, 0.5, 0),
]


def _extract(path, name, kind):
    src = open(path).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, kind) and n.name == name)
    return ast.get_source_segment(src, node)


def load_sampler():
    ns = {"DistributedSampler": DistributedSampler, "Optional": Optional, "dist": dist, "random": random, "torch": torch,
          "get_pg_manager": lambda: None}
    exec(compile(_extract(f"{REF}/train/llava_trainer.py", "VILADistributedSampler", ast.ClassDef), "llava_trainer.py", "exec"), ns)
    return ns["VILADistributedSampler"]


def load_get_checkpoint_path():
    ns = {"os": os, "pathlib": pathlib, "re": re}
    code = "from __future__ import annotations\n" + _extract(f"{REF}/train/utils.py", "get_checkpoint_path", ast.FunctionDef)
    exec(compile(code, "utils.py", "exec"), ns)
    return ns["get_checkpoint_path"]


def checkpoint_layouts():
    """name -> (sub-directories, files) of a run folder."""
    return {
        "empty": ([], []),
        "two": (["checkpoint-100", "checkpoint-20"], []),
        "staging_ignored": (["checkpoint-7", "tmp-checkpoint-9"], []),
        "file_not_dir": (["checkpoint-3"], ["checkpoint-50"]),
        "finished": (["checkpoint-100"], ["config.json"]),
    }


MEDIA_TEXTS = ["<image>\nwhat is this?", "what is <image> this?", "look\n<image>", "<video>\n<image>\ndescribe", "no media here ", "a<image>\nb\n<image>c",
               "  <image>  ", "<image><image>\n\n<image>x"]
MIXTURES = {"sft_all": ["llava_a", "mix_b*2"], "mix_b": ["docvqa", "ai2d"], "stage3": ["sft_all", "extra"]}
MIXTURE_CASES = ["llava_a", "b+a", "sft_all", "stage3+zeta*3", "mix_b+llava_a"]
PAD_CASES = [(10, 4), (12, 4), (3, 8), (3, 7), (1, 16), (5, 16), (100, 64), (7, None), (64, 64), (9, 2)]
RECORDS = [
    {"image": "a.png", "conversations": [{"from": "human", "value": "<image>\nwhat is in the picture?"}, {"from": "gpt", "value": "a cat"}]},
    {"image": ["a.png", "sub/b.png"], "conversations": [{"from": "human", "value": "compare <image> and <image>"}, {"from": "gpt", "value": "same"}]},
    {"images": ["c.png"], "image": "a.png", "conversations": [{"from": "human", "value": "two keys\n<image>"}, {"from": "gpt", "value": "ok <image>"}]},
    {"conversations": [{"from": "human", "value": "text only"}, {"from": "gpt", "value": "yes"}, {"from": "human", "value": "more"}, {"from": "gpt", "value": "no"}]},
    {"image": ["1.png", "2.png", "3.png"], "conversations": [{"from": "human", "value": "<image><image><image>many"}, {"from": "gpt", "value": "three"}]},
    {"image": "a.png", "conversations": [{"from": "gpt", "value": "wrong first speaker"}]},
]


def dataset_section():
    import copy
    import types
    from itertools import chain
    ns = {}
    exec(compile(_extract(f"{REF}/data/dataset_impl/utils.py", "_remove_media_tokens", ast.FunctionDef), "utils.py", "exec"), ns)
    strip = ns["_remove_media_tokens"]
    ns2 = {"List": list, "chain": chain, "MIXTURES": MIXTURES}
    exec(compile("from typing import List\n" + _extract(f"{REF}/data/builder.py", "parse_mixture", ast.FunctionDef), "builder.py", "exec"), ns2)

    class Media:                                                          # stands in for llava.media.Image / Video: remembers its path
        def __init__(self, path):
            self.path = path

    class Base:                                                           # stands in for BaseDataset.__init__ (base.py:75-94)
        def __init__(self, data_args=None, **kw):
            self.data_args = data_args
    drawn = []
    fake_random = types.SimpleNamespace(sample=lambda pop, k: (drawn.append((len(pop), k)), list(pop)[:k])[1])
    holder = {}
    ns3 = {"BaseDataset": Base, "Optional": Optional, "Dict": dict, "Any": object, "List": list, "copy": copy, "os": os, "random": fake_random,
           "Image": Media, "Video": Media, "make_list": lambda x: x if isinstance(x, (list, tuple)) else [x], "_remove_media_tokens": strip,
           "local_load_or_hf_load": lambda path: list(holder["instances"])}
    exec(compile("from typing import Any, Dict, List, Optional\n" + _extract(f"{REF}/data/dataset_impl/llava.py", "LLaVADataset", ast.ClassDef), "llava.py", "exec"), ns3)
    D = ns3["LLaVADataset"]
    pads = []
    for n, gbs in PAD_CASES:
        holder["instances"] = [{"k": i} for i in range(n)]
        drawn.clear()
        d = D("x.json", "m", data_args=types.SimpleNamespace(image_aspect_ratio="dynamic", max_num_images=None), global_batch_size=gbs)
        pads.append({"n": n, "global_batch_size": gbs, "len": len(d.instances), "drawn": list(drawn)})
    procs = []
    for max_images in (None, 2):
        holder["instances"] = RECORDS
        d = D("x.json", "media/root", data_args=types.SimpleNamespace(image_aspect_ratio="resize", max_num_images=max_images))
        for r in RECORDS:
            try:
                msgs = d.process(copy.deepcopy(r))
                rec = {"messages": [{"from": m["from"], "value": [v.path if isinstance(v, Media) else v for v in m["value"]] if isinstance(m["value"], list) else m["value"]}
                                    for m in msgs]}
            except ValueError as e:
                rec = {"error": str(e)}
            procs.append({"max_num_images": max_images, **rec})
    return {"strip": [[t, strip(t)] for t in MEDIA_TEXTS], "mixtures": MIXTURES, "parse": [[m, ns2["parse_mixture"](m)] for m in MIXTURE_CASES],
            "pad": pads, "records": RECORDS, "process": procs}


def decay_section():
    """-> {"decay": [...names...], "all": [...names...]} for the tiny `mlp_downsample` and `mlp_downsample_3x3_fix` configurations."""
    # llava_trainer.py:32-33 imports both names from transformers.trainer (4.46); the installed 5.x keeps them in their home modules only
    from transformers.pytorch_utils import ALL_LAYERNORM_LAYERS
    from transformers.trainer_pt_utils import get_parameter_names
    sys.path.insert(0, ROOT)
    from oracle import make_golden as G
    from vila_amd import configs, synthetic
    src = open(f"{REF}/train/llava_trainer.py").read()
    fn = next(n for c in ast.parse(src).body if isinstance(c, ast.ClassDef) and c.name == "LLaVATrainer" for n in c.body
              if isinstance(n, ast.FunctionDef) and n.name == "create_optimizer")
    stmts = [ast.get_source_segment(src, n) for n in ast.walk(fn) if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "decay_parameters"]
    assert len(stmts) == 2, stmts
    out = {}
    for kind in ("mlp_downsample", "mlp_downsample_3x3_fix"):
        cfg = configs.tiny(kind)
        w = synthetic.make_weights(cfg, 0)
        ms, bp = G.ref_siglip(), G.ref_projector()
        v = cfg.vision
        vc = ms.SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                                   num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size, num_channels=v.num_channels)
        vc._attn_implementation = "eager"
        llm, _ = G.build_hf_llm(cfg, w)

        class Tower(torch.nn.Module):                                     # VisionTower holds the HF model as `.vision_tower` (vision_encoder.py)
            def __init__(self):
                super().__init__()
                self.vision_tower = ms.SiglipVisionModel(vc)

        class VLM(torch.nn.Module):                                       # LlavaMetaModel's three attributes (llava_arch.py:73-75)
            def __init__(self):
                super().__init__()
                self.llm, self.vision_tower = llm, Tower()
                self.mm_projector = bp.MultimodalProjector(bp.MultimodalProjectorConfig(cfg.mm_projector_type),
                                                           types.SimpleNamespace(mm_hidden_size=cfg.mm_hidden_size, hidden_size=cfg.llm.hidden_size))
        opt_model = VLM()
        ns = {"get_parameter_names": get_parameter_names, "ALL_LAYERNORM_LAYERS": ALL_LAYERNORM_LAYERS, "opt_model": opt_model}
        for st in stmts:
            exec(st, ns)
        out[kind] = {"decay": sorted(ns["decay_parameters"]), "all": sorted(n for n, _ in opt_model.named_parameters()),
                     "layernorm_layers": [c.__name__ for c in ALL_LAYERNORM_LAYERS]}
    return out


def main():
    S = load_sampler()
    samplers = []
    for lens, world, bs, acc, seed in SAMPLER_CASES:
        rec = {"lens": lens, "world": world, "batch_size": bs, "accumulation": acc, "seed": seed, "ranks": []}
        for rank in range(world):
            s = S(list(range(sum(lens))), num_replicas=world, rank=rank, seed=seed, batch_size=bs, sample_len_list=lens,
                  gradient_accumulation_steps=acc)
            per_epoch = []
            for epoch in (0, 1):
                s.set_epoch(epoch)
                per_epoch.append(list(iter(s)))
            rec["ranks"].append({"len": len(s), "order": per_epoch})
        samplers.append(rec)

    from transformers import get_scheduler
    schedules = []
    for kind, total, ratio, wsteps in SCHEDULE_CASES:
        n_warm = wsteps if wsteps > 0 else math.ceil(total * ratio)          # TrainingArguments.get_warmup_steps
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=2e-5)
        sch = get_scheduler(kind, optimizer=opt, num_warmup_steps=n_warm, num_training_steps=total)
        lrs = []
        for _ in range(total + 2):                                            # two updates past the end as well
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        schedules.append({"kind": kind, "total": total, "warmup_ratio": ratio, "warmup_steps": wsteps, "n_warmup": n_warm, "lr": lrs})

    g = load_get_checkpoint_path()
    ckpt = {}
    for name, (dirs, files) in checkpoint_layouts().items():
        with tempfile.TemporaryDirectory() as t:
            run = os.path.join(t, "run")
            os.makedirs(run)
            for d in dirs:
                os.makedirs(os.path.join(run, d))
            for f in files:
                open(os.path.join(run, f), "w").close()
            path, cont = g(run)
            ckpt[name] = {"path": None if path is None else os.path.relpath(path, run), "continue": bool(cont)}
        path, cont = g(os.path.join(t, "missing"))
    ckpt["missing_folder"] = {"path": None if path is None else path, "continue": bool(cont)}

    import transformers
    json.dump({"transformers": transformers.__version__, "base_lr": 2e-5, "samplers": samplers, "schedules": schedules, "checkpoints": ckpt,
               "datasets": dataset_section(), "decay": decay_section()},
              open(OUT, "w"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(samplers), "sampler cases,", len(schedules), "schedules,", ckpt)


if __name__ == "__main__":
    main()
