#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python - > gpurun_out/r2g_dbg.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from vila_amd import configs, synthetic
from vila_amd.train import SFTTrainer, count_targets
from vila_amd.vlm import build_model
cfg = configs.reduced_8b(layers_v=4, layers_l=2, vocab=32000)
cfg.image_token_id, cfg.llm.eos_token_id = 31999, 31998
b, T = 2, 512
ids = torch.stack([synthetic.make_prompt(cfg, T, 1, 40 + i) for i in range(b)], 0)
labels = ids.clone(); labels[:, : 1 + T - 256] = -100
mask = torch.ones_like(ids, dtype=torch.bool)
model = build_model(cfg, seed=17)
tr = SFTTrainer(model, optimizer_state=False)
px = synthetic.make_pixels(cfg, b, 17).to(torch.bfloat16)
n = count_targets(ids, labels, mask, cfg.image_token_id)
loss = tr.forward_backward_c(ids, [p.cuda() for p in px], labels, mask, n)
torch.cuda.synchronize()
print("loss", float(loss))
PY
echo "rc=$?"; tail -12 gpurun_out/r2g_dbg.log
