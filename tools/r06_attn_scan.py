"""Attention forward at a ladder of sequence lengths; run under rocprofv3 --kernel-trace and read the per-launch durations (tools/r06_attn_scan.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vila_amd import ops  # noqa: E402

torch.manual_seed(0)
for D, Hq, Hkv, causal in [(128, 28, 4, True), (72, 16, 16, False)]:
    for T in (16, 64, 128, 256, 512, 769, 1024):
        q = (torch.randn(T, Hq, D, device="cuda")).to(torch.bfloat16)
        k = (torch.randn(T, Hkv, D, device="cuda")).to(torch.bfloat16)
        v = (torch.randn(T, Hkv, D, device="cuda")).to(torch.bfloat16)
        for _ in range(6):
            o = ops.attn_fwd(q, k, v, causal)
        torch.cuda.synchronize()
