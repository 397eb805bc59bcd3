"""Golden vectors for the sampling path: the post-filter distribution that HF `GenerationMixin.sample` draws from, produced by EXECUTING
HF's own processors (transformers.generation.logits_process: TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper, the order of
`_get_logits_processor`) on seeded logits.  The reference calls exactly this through `llm.generate(..., do_sample=True)`
(llava/model/llava_arch.py:833; server defaults temperature 0.2 / top_p 0.9: server.py:101-102,185-187).
Run here (transformers is installed in this container; the GPU box never needs it):  python oracle/make_golden_sampling.py
-> tests/golden/sampling_hf.npz   (transformers version recorded inside; the reference pins 4.46.0, this container has 5.15.0)"""
import os

import numpy as np
import torch
import transformers
from transformers import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

CASES = [(1000, 1.0, 50, 1.0, 3.0), (4096, 0.2, 50, 0.9, 3.0), (4096, 0.7, 64, 0.5, 2.0), (5000, 1.5, 1, 0.9, 3.0), (300, 0.9, 40, 0.3, 3.0),
         (2048, 0.8, 20, 0.95, 2.0),
         # round 4 (any top_k): k > 64, k = 0 (HF adds no TopKLogitsWarper for 0: GenerationMixin._get_logits_processor), k >= V
         (8192, 1.0, 100, 0.95, 2.5), (20000, 0.7, 1000, 0.9, 3.0), (6000, 1.0, 0, 0.9, 3.0), (3000, 0.5, 5000, 0.999, 2.0)]


def main():
    out = {"transformers_version": np.array(transformers.__version__)}
    for n, (V, temperature, top_k, top_p, scale) in enumerate(CASES):
        g = torch.Generator().manual_seed(1000 + n)
        logits = (torch.randn(1, V, generator=g) * scale).float()
        ids = torch.zeros((1, 1), dtype=torch.long)
        z = TemperatureLogitsWarper(temperature)(ids, logits.clone())
        if top_k != 0:                                       # _get_logits_processor: `top_k is not None and top_k != 0`
            z = TopKLogitsWarper(top_k)(ids, z)
        if top_p < 1.0:
            z = TopPLogitsWarper(top_p)(ids, z)
        out[f"c{n}_params"] = np.array([V, temperature, top_k, top_p], dtype=np.float64)
        out[f"c{n}_logits"] = logits[0].numpy()
        out[f"c{n}_probs"] = torch.softmax(z[0].double(), -1).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "sampling_hf.npz")
    np.savez_compressed(path, **out)
    print("wrote", os.path.normpath(path), {k: v.shape for k, v in out.items() if k.endswith("_probs")})


if __name__ == "__main__":
    main()
