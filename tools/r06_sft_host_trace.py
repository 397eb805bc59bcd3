"""Host-side timeline of SFT steps: when does the host enter / leave each phase of SFTTrainer.step, relative to the GPU finishing the step?
usage (GPU box): python tools/r06_sft_host_trace.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vila_amd import configs, synthetic, train  # noqa: E402
from vila_amd.vlm import build_model  # noqa: E402

cfg = configs.nvila_8b()
model = build_model(cfg, seed=0, device="cuda:0")
tr = train.SFTTrainer(model, lr=2e-5, weight_decay=0.0)
b = 4
pixels = synthetic.make_pixels(cfg, b, 0, device="cuda:0", dtype=torch.bfloat16)
ids = torch.stack([synthetic.make_prompt(cfg, 512, 1, i) for i in range(b)], 0)
labels = ids.clone()
labels[:, :257] = -100
images = [pixels[i] for i in range(b)]
marks = []
T0 = [0.0]


def wrap(obj, name):
    fn = getattr(obj, name)

    def w(*a, **k):
        marks.append((name + " >", time.perf_counter() - T0[0]))
        r = fn(*a, **k)
        marks.append((name + " <", time.perf_counter() - T0[0]))
        return r
    setattr(obj, name, w)


for n in ("_vit_fwd", "_proj_fwd", "_splice", "_llm_fwd", "_llm_bwd", "_vit_bwd", "_proj_bwd", "_finish_backward", "forward_backward", "optimizer_step"):
    if hasattr(tr, n):
        wrap(tr, n)
for _ in range(2):
    tr.step(ids, images, labels)
torch.cuda.synchronize()
for s in range(3):
    marks.clear()
    torch.cuda.synchronize()
    T0[0] = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tr.step(ids, images, labels)
    t_host = time.perf_counter() - T0[0]
    tr.step(ids, images, labels)
    e1.record()
    t_host2 = time.perf_counter() - T0[0]
    torch.cuda.synchronize()
    print(f"--- pair {s}: host returned from step 1 at {t_host * 1e3:.1f} ms, from step 2 at {t_host2 * 1e3:.1f} ms; GPU time of both {e0.elapsed_time(e1):.1f} ms")
    for name, t in marks:
        print(f"   {t * 1e3:8.2f} ms  {name}")
