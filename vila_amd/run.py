"""An SFT *run* around the native SFT step (SURVEY §8e partitioning, §8 f4 "an SFT run, not just a step"): host-only integer / float work.

  * `VILADistributedSampler` — which samples a rank sees, in which order (llava/train/llava_trainer.py:131-279, the sampler
                               `LLaVATrainer._get_train_sampler` always returns, :595-624): every dataset of the mixture is cut down to a whole number
                               of global batches, each rank takes its contiguous share of every dataset, shuffles the shares under `seed + epoch`
                               and spreads the smaller datasets evenly between the samples of the larger ones.  Sequence parallelism
                               (`sp_degree > 1`, LongVILA) is out of scope and refused.
  * `lr_factor` / `warmup_steps` — the `--lr_scheduler_type cosine --warmup_ratio 0.03` schedule of every NVILA script
                               (scripts/NVILA-Lite/sft.sh:41-44 -> transformers.get_scheduler): the multiplier applied to the base learning rate
                               at optimizer update k.
  * `get_checkpoint_path`    — the auto-resume rule of llava/train/utils.py:59-79 (finished run -> the run folder itself, else the highest
                               `checkpoint-<step>` folder).
  * `train`                  — sampler -> collator -> `SFTTrainer.step` -> schedule -> log -> `checkpoint-<step>` folders (weights in the reference's
                               three-folder layout + optimizer state + `trainer_state.json`, staged under `tmp-checkpoint-<step>` and renamed like
                               transformer_normalize_monkey_patch.py:100-160) with `save_total_limit` rotation and resume that replays neither a
                               batch nor a learning rate.

Pinned by tests/golden/run_ref.json: the reference's own sampler class and `get_checkpoint_path` (ast-extracted, executed) and
transformers' own scheduler, see oracle/make_golden_run.py.
"""
from __future__ import annotations

import json
import math
import os
import random
import re
import shutil
from dataclasses import asdict, dataclass, field
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Tuple


# ----------------------------------------------------------------------------------------------------------------------
# which samples a rank trains on
# ----------------------------------------------------------------------------------------------------------------------
class VILADistributedSampler:
    """`sample_len_list[d]` = number of samples of dataset d inside the concatenated mixture (their indices are contiguous, in that order).
    `batch_size` is the per-device batch.  Always drops the remainder (llava_trainer.py:163)."""

    def __init__(self, dataset, num_replicas: int, rank: int, seed: int = 0, batch_size: int = 1, sample_len_list: Optional[Sequence[int]] = None,
                 sp_degree: int = 1, gradient_accumulation_steps: int = 1, shuffle: bool = True):
        if rank >= num_replicas or rank < 0:
            raise ValueError("Invalid rank {}, rank should be in the interval [0, {}]".format(rank, num_replicas - 1))
        if max(1, sp_degree) > 1:
            raise NotImplementedError("sequence parallelism (sp_degree > 1) is outside this library's path")
        n = dataset if isinstance(dataset, int) else len(dataset)
        lens = [n] if sample_len_list is None else [int(x) for x in sample_len_list]
        assert sum(lens) == n
        self.num_replicas, self.rank, self.seed, self.epoch, self.shuffle = num_replicas, rank, seed, 0, shuffle
        self.batch_size = batch_size
        self.global_batch_size = batch_size * num_replicas
        self.dataset_lens = lens
        quantum = batch_size * gradient_accumulation_steps                # a rank's share of a dataset is a whole number of its updates
        self.per_replica_samples = [x // (num_replicas * quantum) * quantum for x in lens]
        self.num_samples = sum(self.per_replica_samples)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        shares, start = [], 0
        for n_all, n_mine in zip(self.dataset_lens, self.per_replica_samples):
            first = start + self.rank * n_mine                             # the kept head of the dataset, cut into one run per rank
            shares.append(list(range(first, first + n_mine)))
            start += n_all
        rng = random.Random(self.seed + self.epoch)                        # == random.seed(...) + random.shuffle(...) of the reference
        for s in shares:
            rng.shuffle(s)
        # the largest share first (stable); every later share is spread evenly over the slots that are still free
        free = list(range(self.num_samples))
        order = [-1] * self.num_samples
        for s in sorted(shares, key=lambda x: -len(x)):
            if not s:
                continue
            picks = [k * len(free) // len(s) for k in range(len(s))]
            for k, p in enumerate(picks):
                order[free[p]] = s[k]
            taken = set(picks)
            free = [slot for i, slot in enumerate(free) if i not in taken]
        assert -1 not in order
        return iter(order)


# ----------------------------------------------------------------------------------------------------------------------
# learning-rate schedule
# ----------------------------------------------------------------------------------------------------------------------
def warmup_steps(max_steps: int, warmup_ratio: float = 0.0, warmup_steps_arg: int = 0) -> int:
    """TrainingArguments.get_warmup_steps: an explicit step count wins, else ceil(ratio x updates)."""
    return warmup_steps_arg if warmup_steps_arg > 0 else math.ceil(max_steps * warmup_ratio)


def lr_factor(kind: str, k: int, n_warmup: int, n_total: int) -> float:
    """Multiplier of the base learning rate at optimizer update k = 0, 1, ... (the value `LambdaLR` holds while update k runs):
    `cosine` / `linear` / `constant` / `constant_with_warmup` of transformers.optimization (get_*_schedule_with_warmup)."""
    if kind == "constant":
        return 1.0
    if k < n_warmup:
        return float(k) / float(max(1, n_warmup))
    if kind == "constant_with_warmup":
        return 1.0
    if kind == "linear":
        return max(0.0, float(n_total - k) / float(max(1, n_total - n_warmup)))
    if kind == "cosine":
        progress = float(k - n_warmup) / float(max(1, n_total - n_warmup))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    raise ValueError(f"unknown lr_scheduler_type '{kind}'")


# ----------------------------------------------------------------------------------------------------------------------
# checkpoints of a run
# ----------------------------------------------------------------------------------------------------------------------
def get_checkpoint_path(output_dir: str, checkpoint_prefix: str = "checkpoint") -> Tuple[Optional[str], bool]:
    """-> (path to resume from or None, continue_training).  A `config.json` in the run folder means the run has finished."""
    output_dir = os.path.abspath(output_dir)
    if os.path.isfile(os.path.join(output_dir, "config.json")):
        return output_dir, False
    best = None
    if os.path.isdir(output_dir):
        for name in os.listdir(output_dir):
            m = re.match(f".*{checkpoint_prefix}-([0-9]+)", os.path.join(output_dir, name))
            if name.startswith(checkpoint_prefix + "-") and m and os.path.isdir(os.path.join(output_dir, name)):
                key = (int(m.group(1)), os.path.join(output_dir, name))
                best = key if best is None or key > best else best
    return (best[1] if best else None), True


@dataclass
class TrainArgs:
    """The `TrainingArguments` fields the NVILA scripts set (scripts/NVILA-Lite/sft.sh:34-50) with HF's defaults."""
    output_dir: str = "runs/sft"
    per_device_train_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    num_train_epochs: float = 1.0
    max_steps: int = -1
    learning_rate: float = 2e-5
    max_grad_norm: Optional[float] = 1.0        # HF's default; the NVILA stage-2 scripts pass 5.0
    warmup_ratio: float = 0.03
    warmup_steps: int = 0
    lr_scheduler_type: str = "cosine"
    logging_steps: int = 1
    save_steps: int = 100
    save_total_limit: Optional[int] = 1
    seed: int = 42
    data_seed: Optional[int] = None
    sample_lens: Optional[List[int]] = None


@dataclass
class TrainerState:
    global_step: int = 0
    epoch: float = 0.0
    max_steps: int = 0
    log_history: List[Dict[str, Any]] = field(default_factory=list)

    def save(self, path: str) -> None:
        with open(path, "w") as fh:
            json.dump(asdict(self), fh, indent=1)

    @classmethod
    def load(cls, path: str) -> "TrainerState":
        d = json.load(open(path))
        return cls(**{k: d[k] for k in ("global_step", "epoch", "max_steps", "log_history") if k in d})


def _save_checkpoint_default(trainer, folder: str) -> None:
    from . import checkpoint
    checkpoint.save_pretrained(trainer.model, folder)
    checkpoint.save_optimizer(trainer, folder)


def _load_checkpoint_default(trainer, folder: str) -> None:
    from . import checkpoint
    checkpoint.load_weights_into(trainer.model, folder)
    checkpoint.load_optimizer(trainer, folder)            # also re-derives the bf16 parameters from the restored masters


def _save_final_default(trainer, folder: str) -> None:
    from . import checkpoint
    checkpoint.save_pretrained(trainer.model, folder)     # writes <run>/config.json: the mark of a finished run


def _rotate(output_dir: str, limit: Optional[int]) -> None:
    if limit is None or limit <= 0:
        return
    found = sorted((int(m.group(1)), d) for d in os.listdir(output_dir) for m in [re.fullmatch(r"checkpoint-([0-9]+)", d)] if m)
    for _, d in found[:max(0, len(found) - limit)]:
        shutil.rmtree(os.path.join(output_dir, d), ignore_errors=True)


def plan(n_samples_per_rank: int, args: TrainArgs) -> Tuple[int, int, int]:
    """-> (optimizer updates per epoch, epochs to run, total updates), as `Trainer._inner_training_loop` sets them up."""
    batches = n_samples_per_rank // args.per_device_train_batch_size + (1 if n_samples_per_rank % args.per_device_train_batch_size else 0)
    per_epoch = max(batches // args.gradient_accumulation_steps, 1)
    if args.max_steps > 0:
        total = args.max_steps
        epochs = args.max_steps // per_epoch + int(args.max_steps % per_epoch > 0)
    else:
        total = math.ceil(args.num_train_epochs * per_epoch)
        epochs = math.ceil(args.num_train_epochs)
    return per_epoch, epochs, total


def train(trainer, dataset, collator: Callable[[Sequence[Any]], Dict[str, Any]], args: TrainArgs, rank: int = 0, world_size: int = 1,
          resume: bool = True, save_fn: Callable = _save_checkpoint_default, load_fn: Callable = _load_checkpoint_default,
          final_save_fn: Optional[Callable] = _save_final_default, log: Optional[Callable[[Dict[str, Any]], None]] = None, barrier: Optional[Callable[[], None]] = None) -> TrainerState:
    """Run `args.num_train_epochs` of SFT over `dataset` on this rank.  `trainer` is an `SFTTrainer` (anything with `.lr` and
    `.step(input_ids, images, labels, attention_mask, block_sizes=, videos=) -> loss`, and `.step_accumulated(list of those keyword dicts)` when
    `gradient_accumulation_steps > 1`); the gradient exchange across ranks happens inside its step.
    Only rank 0 writes checkpoints (every rank holds the same weights and optimizer state after a step); `barrier` keeps the others from running
    ahead of the rename.  A run whose folder already holds the final model is not trained again (llava/train/train.py:503-507); at the end the model is
    written into the run folder itself (train.py's `trainer.save_model(output_dir)`), which is what marks it finished."""
    acc = max(1, int(args.gradient_accumulation_steps))
    if hasattr(trainer, "max_grad_norm"):                              # HF clips the global gradient norm by default (max_grad_norm = 1.0)
        trainer.max_grad_norm = args.max_grad_norm if (args.max_grad_norm is not None and args.max_grad_norm > 0) else None
    seed = args.data_seed if args.data_seed is not None else args.seed
    sampler = VILADistributedSampler(dataset, world_size, rank, seed=seed, batch_size=args.per_device_train_batch_size,
                                     sample_len_list=args.sample_lens, gradient_accumulation_steps=args.gradient_accumulation_steps)
    bs = args.per_device_train_batch_size
    if len(sampler) < bs * acc:
        raise ValueError(f"the mixture ({len(dataset)} samples) does not fill one global batch of {bs} x {acc} x {world_size} samples: "
                         "pad the datasets to the global batch size (data.build_dataset(global_batch_size=...))")
    per_epoch, epochs, total = plan(len(sampler), args)
    n_warm = warmup_steps(total, args.warmup_ratio, args.warmup_steps)
    state = TrainerState(max_steps=total)
    if resume:
        path, cont = get_checkpoint_path(args.output_dir)
        if not cont:
            if log is not None and rank == 0:
                log({"message": f"Models has been ready under {args.output_dir}. Skipp training"})
            done = os.path.join(path, "trainer_state.json")
            return TrainerState.load(done) if os.path.isfile(done) else TrainerState(global_step=total, epoch=float(epochs), max_steps=total)
        if path is not None:
            load_fn(trainer, path)
            state = TrainerState.load(os.path.join(path, "trainer_state.json"))
            state.max_steps = total
    first_epoch, skip = divmod(state.global_step, per_epoch)           # whole epochs done, updates done inside the current one
    for epoch in range(first_epoch, epochs):
        sampler.set_epoch(epoch)
        order = list(sampler)
        for b in range(skip if epoch == first_epoch else 0, per_epoch):
            if state.global_step >= total:
                break
            trainer.lr = args.learning_rate * lr_factor(args.lr_scheduler_type, state.global_step, n_warm, total)
            micro = [_step_kwargs(collator([dataset[i] for i in order[k * bs:(k + 1) * bs]])) for k in range(b * acc, (b + 1) * acc)]
            loss = trainer.step(**micro[0]) if acc == 1 else trainer.step_accumulated(micro)
            state.global_step += 1
            state.epoch = epoch + (b + 1) / per_epoch
            if args.logging_steps and state.global_step % args.logging_steps == 0:
                # HF logs the rate the scheduler holds AFTER its step: the one the next update will use
                rec = {"loss": float(loss), "learning_rate": args.learning_rate * lr_factor(args.lr_scheduler_type, state.global_step, n_warm, total),
                       "epoch": round(state.epoch, 4), "step": state.global_step}
                state.log_history.append(rec)
                if log is not None and rank == 0:
                    log(rec)
            if args.save_steps and state.global_step % args.save_steps == 0:
                _checkpoint(trainer, args, state, rank, save_fn, barrier)
    if final_save_fn is not None:
        if rank == 0:
            os.makedirs(args.output_dir, exist_ok=True)
            final_save_fn(trainer, args.output_dir)
            state.save(os.path.join(args.output_dir, "trainer_state.json"))
        if barrier is not None:
            barrier()
    return state


def _step_kwargs(batch: Dict[str, Any]) -> Dict[str, Any]:
    """The collator's batch (llava/data/collate.py:139-159) -> the keyword arguments of `SFTTrainer.step`."""
    media, mcfg = batch.get("media") or {}, batch.get("media_config") or {}
    return {"input_ids": batch["input_ids"], "images": list(media.get("image", [])), "labels": batch["labels"],
            "attention_mask": batch.get("attention_mask"), "block_sizes": (mcfg.get("image") or {}).get("block_sizes"),
            "videos": list(media.get("video", [])) or None}


def _checkpoint(trainer, args: TrainArgs, state: TrainerState, rank: int, save_fn: Callable, barrier) -> None:
    final = os.path.join(args.output_dir, f"checkpoint-{state.global_step}")
    if rank == 0:
        staging = os.path.join(args.output_dir, f"tmp-checkpoint-{state.global_step}")
        shutil.rmtree(staging, ignore_errors=True)
        os.makedirs(staging)
        save_fn(trainer, staging)
        state.save(os.path.join(staging, "trainer_state.json"))
        shutil.rmtree(final, ignore_errors=True)
        os.rename(staging, final)
        _rotate(args.output_dir, args.save_total_limit)
    if barrier is not None:
        barrier()
