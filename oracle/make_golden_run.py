"""REFERENCE-EXECUTED fixture for the host logic of an SFT run (vila_amd/run.py) — TEST INFRASTRUCTURE.

  sampler      `VILADistributedSampler` (llava/train/llava_trainer.py:131-279) taken out of its file with `ast` and executed unchanged: the index
               order every rank sees, for mixtures of 1-4 datasets, 1-8 ranks, two epochs, several batch sizes / accumulation counts
  schedule     transformers' own `get_scheduler` (the object `Trainer.create_scheduler` builds for `--lr_scheduler_type ... --warmup_ratio ...`,
               scripts/NVILA-Lite/sft.sh:41-44) stepped over a dummy optimizer: the learning rate of every update
  checkpoints  `get_checkpoint_path` (llava/train/utils.py:59-79) executed over five run-folder layouts

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
    json.dump({"transformers": transformers.__version__, "base_lr": 2e-5, "samplers": samplers, "schedules": schedules, "checkpoints": ckpt},
              open(OUT, "w"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(samplers), "sampler cases,", len(schedules), "schedules,", ckpt)


if __name__ == "__main__":
    main()
