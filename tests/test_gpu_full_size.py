"""Full-size (BASELINE configs[1]: NVILA-8B, 1 x 448^2 image + 512-token prompt, S = 769) property tests.

The fp32 CPU oracle needs minutes per token at this size, so parity here goes through size-independent properties of the path:
  * decode == prefill: the M = 1 kernels (GEMV, fused RoPE / KV append, per-head decode attention) must reproduce the logits the
    prefill kernels (MFMA GEMMs, flash attention) give at the same positions when fed the same tokens
  * hipGraph replay == eager launches, token ids bit-exact
  * batch independence of the tower: encode_images of 2 tiles == the two tiles encoded one by one (rel-L2 <= 2e-3: the GEMM
    launcher may pick another tile shape for M = 2048 than for M = 1024; any cross-talk between images would show as O(1))
  * the W4A16 decode step agrees with the bf16 one on weights that are exactly representable in the int4 format
  * KV cache: the layer-0 K/V rows the decode steps append (fused GEMV + bias + RoPE) == the rows the prefill writes for the same
    tokens (MFMA GEMM + RoPE kernel); layer 0 sees identical inputs on both paths, so this is tight (rel-L2 <= 1e-2)
Tolerances: 28 random-weight layers amplify the bf16 rounding differences of two pipelines with different accumulation orders:
logits rel-L2 <= 5e-2 (the 2-layer cases of test_gpu_model.py hold 1.5e-2), greedy ids bit-exact wherever the top-1 margin
exceeds 4x the observed max-abs error, graph vs eager ids bit-exact unconditionally.
"""
import pytest
import torch

from tests.gpu_util import rel_l2
from vila_amd import configs, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from vila_amd.vlm import build_model
    cfg = configs.nvila_8b()
    model = build_model(cfg, seed=3)
    px = synthetic.make_pixels(cfg, 2, 3, device="cuda", dtype=torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 512, 1, 3)[None].cuda()
    return cfg, model, px, ids


def test_full_size_decode_matches_prefill_and_graph_matches_eager(full):
    cfg, model, px, ids = full
    e, _, _ = model._embed(ids, {"image": [px[0]]})
    assert e.shape == (1, 769, 3584)
    n = 12
    free_g = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=True, eos_token_id=-1)
    free_e = model.llm.generate(inputs_embeds=e, max_new_tokens=n, use_graph=False, eos_token_id=-1)
    assert torch.equal(free_g, free_e)
    forced = free_g[0].cpu()
    c_dec, c_pre = model.llm.new_cache(1024), model.llm.new_cache(1024)
    _, lg = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=forced, use_graph=False, cache=c_dec)
    tail = model.llm.embed_tokens(forced[: n - 1].cuda())
    full_seq = torch.cat([e[0], tail], 0)
    S = full_seq.shape[0]
    r = model.llm.prefill_packed(full_seq, torch.arange(S, dtype=torch.int32, device="cuda"), None, S, want_all_logits=True, cache=c_pre)
    ref = r.all_logits[e.shape[1] - 1:]
    assert rel_l2(lg, ref) < 5e-2, f"decode vs prefill rel={rel_l2(lg, ref):.3e}"
    S0 = e.shape[1]
    for name, a, b in (("K", c_dec.k, c_pre.k), ("V", c_dec.v, c_pre.v)):
        dec_rows, pre_rows = a[0, 0, :, S0:S0 + n - 1], b[0, 0, :, S0:S0 + n - 1]
        assert float(pre_rows.float().abs().max()) > 0
        assert rel_l2(dec_rows, pre_rows) < 1e-2, f"layer-0 {name} rows appended by decode vs prefill rel={rel_l2(dec_rows, pre_rows):.3e}"
        # the prompt part: written by a 769-row prefill there and a 780-row prefill here (other GEMM tilings / K-slices, and row 768's
        # MLP through the GEMV path only in the 769-row run), so equal up to bf16 rounding through 28 layers, not bit for bit
        assert rel_l2(a[:, 0, :, :S0], b[:, 0, :, :S0]) < 2e-2, f"prompt {name} rel={rel_l2(a[:, 0, :, :S0], b[:, 0, :, :S0]):.3e}"
        assert torch.equal(a[0, 0, :, :S0], b[0, 0, :, :S0]) or rel_l2(a[0, 0, :, :S0], b[0, 0, :, :S0]) < 4e-3     # layer 0: one GEMM deep
    # where the prefill's top-1 margin exceeds 4x the observed error the greedy ids must coincide
    err = float((lg.float() - ref.float()).abs().max())
    top2 = ref.float().topk(2, -1).values
    decisive = ((top2[:, 0] - top2[:, 1]) > 4 * err).cpu()
    assert decisive.any()
    assert torch.equal(lg.argmax(-1).cpu()[decisive], ref.argmax(-1).cpu()[decisive])


def test_full_size_tower_is_batch_independent(full):
    cfg, model, px, ids = full
    """Two images in one call vs one call each.  With the kernel choice pinned (every GEMM on the same tile kernel, same K order) the rows of an
    image do not depend on what else is in the batch: <= 2e-3 (bit-identical in practice).  With the automatic choice the K-slicing of fc2 depends
    on M (8 slices at 1024 rows, 6 at 2048: round 3, cold-weight policy), i.e. another fp32 summation order in 26 layers of bf16 rounding: the two
    results differ like two bf16 runs of the same math do (<= 2e-2, the suite's hidden-state tolerance), and both stay within it of each other's
    pinned result."""
    from vila_amd import _lib
    lib = _lib.load()
    both = model.encode_images(px)
    one = torch.cat([model.encode_images(px[:1]), model.encode_images(px[1:])], 0)
    assert both.shape == (2, 256, 3584)
    assert rel_l2(both, one) < 2e-2, f"automatic kernel choice: rel={rel_l2(both, one):.3e}"
    lib.vila_gemm_force_tile(7)                                  # the 128x64 ring kernel for every tower / projector GEMM, no K-slicing
    try:
        both_p = model.encode_images(px)
        one_p = torch.cat([model.encode_images(px[:1]), model.encode_images(px[1:])], 0)
    finally:
        lib.vila_gemm_force_tile(0)
    assert rel_l2(both_p, one_p) < 2e-3, f"pinned kernels: rel={rel_l2(both_p, one_p):.3e}"
    assert rel_l2(both, both_p) < 2e-2 and rel_l2(one, one_p) < 2e-2, f"{rel_l2(both, both_p):.3e} {rel_l2(one, one_p):.3e}"
    print(f"tower batch independence: auto {rel_l2(both, one):.2e}, pinned {rel_l2(both_p, one_p):.2e}, auto vs pinned {rel_l2(both, both_p):.2e} / {rel_l2(one, one_p):.2e}")


def test_full_size_w4_decode_tracks_bf16_decode():
    """Same token stream through the bf16 and the W4A16 decode steps of a 4-layer NVILA-8B-width model whose projections are
    exactly representable in int4 (so both paths use the SAME weights): logits must agree to the bf16 decode tolerance."""
    import zlib
    from tests.test_gpu_w4 import _exact_w4
    from vila_amd.vlm import build_model
    cfg = configs.reduced_8b(layers_v=1, layers_l=4, vocab=152064)
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 9).items()}
    for k in list(w):
        if k.startswith("llm.model.layers.") and k.endswith("_proj.weight"):
            w[k] = _exact_w4(tuple(w[k].shape), zlib.crc32(k.encode()) % 10007, (-9, -8, -7))
    model = build_model(cfg, weights=w)
    px = synthetic.make_pixels(cfg, 1, 9, device="cuda", dtype=torch.bfloat16)
    ids = synthetic.make_prompt(cfg, 512, 1, 9)[None].cuda()
    e, _, _ = model._embed(ids, {"image": [px[0]]})
    n = 8
    ids_bf, lg_bf = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, use_graph=False, eos_token_id=-1)
    model.llm.quantize_w4(keep_logical=False)
    _, lg_w4 = model.llm.generate(inputs_embeds=e, max_new_tokens=n, return_logits=True, forced_ids=ids_bf[0].cpu(), use_graph=False)
    assert rel_l2(lg_w4, lg_bf) < 1.5e-2, f"W4 vs bf16 decode rel={rel_l2(lg_w4, lg_bf):.3e}"
