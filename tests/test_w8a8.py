"""W8A8 vision tower (BASELINE configs[4]).  The reference's quantised numbers come from the external TinyChat
backend (README.md:87): no reference code or outputs exist in-tree — PARITY UNPINNED AGAINST THE REFERENCE.  What is pinned: the HIP int8
path against a CPU oracle that dequantises THE SAME int8 tensors and applies the same per-token activation quantisation in fp32."""
import pytest
import torch

from oracle import vila_oracle as O
from vila_amd import configs, synthetic
from vila_amd.quant import dequantize_w8, quantize_w8


def test_quantize_w8_roundtrip_and_ranges():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(37, 144, generator=g) * torch.rand(37, 1, generator=g) * 3
    q, s = quantize_w8(w)
    assert q.dtype == torch.int8 and int(q.abs().max()) == 127 and s.shape == (37,)
    assert float((dequantize_w8(q, s) - w).abs().max()) <= float(s.max()) * 0.5 + 1e-6        # half a quantisation step
    # oracle activation fake-quant: integer grid, half-even, zero rows stay zero
    x = torch.tensor([[0.0, 0.0, 0.0], [1.0, -127.0, 63.5], [0.5, 1.5, 2.5]])
    fq = O.fake_quant_rows(x)
    assert torch.equal(fq[0], x[0]) and torch.equal(fq[1], torch.tensor([1.0, -127.0, 64.0]))
    assert torch.allclose(fq[2], torch.round(x[2] / (2.5 / 127)) * (2.5 / 127))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,epi", [(1024, 3456, 1152, 0), (1024, 1152, 4304, 0), (1024, 4304, 1152, 1), (300, 264, 144, 0), (4096, 1152, 1152, 0), (129, 68, 272, 1)])
def test_gemm_w8a8_matches_integer_reference(M, N, K, epi):
    """int8 x int8 -> int32 is exact: against the same integer matmul done in fp64 on the host the only error is the bf16 rounding of
    the output (rel-L2 <= 4e-3); covers the K tail (4304 = 33.6 tiles of 128), M / N tails, bias, residual, tanh-GELU."""
    from tests.gpu_util import randn_bf16, rel_l2
    from vila_amd import ops
    x = randn_bf16(M, K, seed=3)
    w = randn_bf16(N, K, seed=4, scale=K ** -0.5)
    bias, res = randn_bf16(N, seed=5), randn_bf16(M, N, seed=6)
    xq, sx = ops.quant_rows_i8(x)
    wq, sw = quantize_w8(w)
    # the quantiser itself: same grid as the host rule
    xs = (x.float().abs().amax(-1) / 127).clamp_min(1e-30)
    assert torch.allclose(sx, xs, rtol=1e-6)
    want = torch.round(x.float() / xs[:, None]).clamp(-127, 127)
    diff = (xq.float() - want).abs()
    # same grid; a quotient that lands within an ulp of k + 0.5 may round to the neighbour (two division routines): at most 1 step, rarely
    assert float(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3, (float(diff.max()), float((diff > 0).float().mean()))
    acc = (xq.double().cpu() @ wq.double().cpu().t())
    ref = acc * sx.double().cpu()[:, None] * sw.double().cpu()[None, :] + bias.double().cpu()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    out = ops.gemm_w8a8(xq, sx, wq, sw, bias=bias, epi=epi)
    assert rel_l2(out, ref) < 4e-3, f"rel={rel_l2(out, ref):.3e}"
    out = ops.gemm_w8a8(xq, sx, wq, sw, bias=bias, residual=res, epi=epi)
    assert rel_l2(out, ref + res.double().cpu()) < 4e-3


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["tiny", "wide"])
def test_w8a8_tower_vs_dequant_oracle(which):
    """Whole tower: HIP int8 path vs the fp32 oracle on the dequantised int8 weights with the same per-token activation quantisation.
    Tolerance rel-L2 <= 3e-2 (an activation that sits on a rounding boundary may land on the neighbouring integer in bf16 vs fp32);
    and the int8 tower stays within 6e-2 of the bf16 tower (what the quantisation itself costs on these weights)."""
    from tests.gpu_util import rel_l2
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample") if which == "tiny" else configs.reduced_8b(layers_v=4, layers_l=1, vocab=1024)
    if which == "wide":
        cfg.image_token_id, cfg.video_token_id, cfg.llm.eos_token_id = 1023, 1022, 1021
    w = {k: v.to(torch.bfloat16).float() for k, v in synthetic.make_weights(cfg, 21).items()}
    model = build_model(cfg, weights=w)
    px = synthetic.make_pixels(cfg, 2, 21).to(torch.bfloat16)
    ref_bf16 = model.vision_tower(px.cuda())
    w8 = model.vision_tower.quantize_w8()
    out = model.vision_tower(px.cuda())
    wd = dict(w)
    wd.update(w8.dequantized_state())
    ref = O.vision_tower_forward_w8a8(px.float(), wd, cfg.vision)
    assert rel_l2(out, ref) < 3e-2, f"W8A8 tower vs dequant oracle rel={rel_l2(out, ref):.3e}"
    assert rel_l2(out, ref_bf16) < 6e-2, f"W8A8 vs bf16 tower rel={rel_l2(out, ref_bf16):.3e}"
    assert rel_l2(out, ref_bf16) > 1e-4             # the int8 path really ran
    # encode_images end to end (projector on the int8 tower's output)
    feats = model.encode_images(px.cuda())
    assert feats.shape == (2, cfg.tokens_per_tile, cfg.llm.hidden_size) and bool(torch.isfinite(feats.float()).all())
