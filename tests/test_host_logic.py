"""CPU tests: host-side integer logic (splice plan, repack) against the oracle restatement of llava_arch.py, and
that the C-ABI library loads and exports every symbol include/vila_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest
import torch

from oracle import vila_oracle as O
from vila_amd import configs, host, synthetic


def _emulate_splice(plan, table, media_flat, H):
    out = torch.zeros((plan.B * plan.S, H), dtype=table.dtype)
    out[plan.txt_dst.long()] = table[plan.txt_src.long()]
    if plan.img_dst.numel():
        out[plan.img_dst.long()] = media_flat
    return out.view(plan.B, plan.S, H)


@pytest.mark.parametrize("side", ["right", "left"])
def test_splice_plan_matches_oracle(side):
    cfg = configs.tiny()
    g = torch.Generator().manual_seed(3)
    H, V = 16, cfg.llm.vocab_size
    table = torch.randn(V, H, generator=g)
    w = {"llm.model.embed_tokens.weight": table}
    img = cfg.image_token_id
    # 3 samples, ragged, with 0 / 1 / 2 images, padded input ids
    L = 9
    ids = torch.randint(0, 900, (3, L), generator=g)
    mask = torch.ones(3, L, dtype=torch.bool)
    ids[0, 2] = img
    ids[1, 0] = img; ids[1, 5] = img
    mask[1, 7:] = False
    mask[2, 4:] = False
    ids[2, 6] = img            # a media id inside the PADDING must be ignored (attention_mask removes it first, :449-450)
    labels = torch.randint(0, 900, (3, L), generator=g)
    media = [torch.randn(5, H, generator=g), torch.randn(5, H, generator=g), torch.randn(3, H, generator=g)]
    e_ref, l_ref, m_ref = O.embed_splice(ids, [m.clone() for m in media], w, cfg, labels=labels, attention_mask=mask, padding_side=side)
    plan = host.splice_plan(ids, mask, labels, [m.shape[0] for m in media], img, side)
    e = _emulate_splice(plan, table, torch.cat(media, 0), H)
    assert torch.equal(e, e_ref)
    assert torch.equal(plan.labels, l_ref)
    assert torch.equal(plan.mask, m_ref)
    assert plan.seqlens.tolist() == [9 - 1 + 5, 7 - 2 + 5 + 3, 4]


@pytest.mark.parametrize("side", ["right", "left"])
@pytest.mark.parametrize("max_len", [6, 9, 11, 40])
def test_splice_plan_truncation_matches_oracle(side, max_len):
    """`__truncate_sequence` (llava_arch.py:519-526): cut AFTER media expansion, through the middle of an image if need be,
    and only when some sample exceeds model_max_length."""
    cfg = configs.tiny()
    g = torch.Generator().manual_seed(4)
    H, V = 8, cfg.llm.vocab_size
    table = torch.randn(V, H, generator=g)
    w = {"llm.model.embed_tokens.weight": table}
    img = cfg.image_token_id
    ids = torch.randint(0, 900, (3, 7), generator=g)
    mask = torch.ones(3, 7, dtype=torch.bool)
    ids[0, 1] = img
    ids[1, 0] = img; ids[1, 4] = img
    mask[2, 3:] = False
    labels = torch.randint(0, 900, (3, 7), generator=g)
    media = [torch.randn(5, H, generator=g), torch.randn(4, H, generator=g), torch.randn(6, H, generator=g)]
    e_ref, l_ref, m_ref = O.embed_splice(ids, [m.clone() for m in media], w, cfg, labels=labels, attention_mask=mask, padding_side=side,
                                         max_length=max_len)
    plan = host.splice_plan(ids, mask, labels, [m.shape[0] for m in media], img, side, max_length=max_len)
    flat = torch.cat(media, 0)
    e = torch.zeros(plan.B * plan.S, H)
    e[plan.txt_dst.long()] = table[plan.txt_src.long()]
    e[plan.img_dst.long()] = flat[plan.img_src.long()]
    assert plan.truncated == (max_len < 15)                  # longest expanded sample: 7 - 2 + 4 + 6 = 15
    assert torch.equal(e.view(plan.B, plan.S, H), e_ref)
    assert torch.equal(plan.labels, l_ref)
    assert torch.equal(plan.mask, m_ref)


def test_splice_plan_errors_match_reference():
    cfg = configs.tiny()
    ids = torch.tensor([[1, cfg.image_token_id, 2]])
    with pytest.raises(ValueError, match="Not all image embeddings are consumed!"):
        host.splice_plan(ids, None, None, [4, 4], cfg.image_token_id)
    with pytest.raises(IndexError):
        host.splice_plan(ids, None, None, [], cfg.image_token_id)


def test_splice_plan_text_only_and_empty_rows():
    cfg = configs.tiny()
    ids = torch.tensor([[5, 6, 7], [8, 9, 10]])
    mask = torch.tensor([[True, True, True], [True, False, False]])
    plan = host.splice_plan(ids, mask, None, [], cfg.image_token_id)
    assert plan.S == 3 and plan.seqlens.tolist() == [3, 1]
    assert plan.txt_dst.tolist() == [0, 1, 2, 3] and plan.txt_src.tolist() == [5, 6, 7, 8]
    assert plan.mask.tolist() == [[True, True, True], [True, False, False]]


def test_repack_matches_oracle():
    g = torch.Generator().manual_seed(5)
    B, S, H = 3, 7, 4
    e = torch.randn(B, S, H, generator=g)
    mask = torch.ones(B, S, dtype=torch.bool)
    mask[0, 5:] = False
    mask[2, 2:] = False
    labels = torch.randint(0, 50, (B, S), generator=g)
    pe, pm, pp, pl, seqlens = O.repack(e, mask, labels)
    rp = host.repack(mask, labels)
    n = int(mask.sum())
    assert torch.equal(e.reshape(-1, H)[rp.rows], pe[0, :n])                # oracle has one extra dummy row at the end
    assert torch.equal(rp.position_ids, pp[0, :n])
    assert torch.equal(rp.labels, pl[0, :n])
    idx, cu, mx = O.get_unpad_data(pm, seqlens)
    assert torch.equal(rp.cu_seqlens, cu) and rp.max_seqlen == mx
    assert torch.equal(idx, torch.arange(n))                                 # the dummy token is dropped by the unpad indices
    assert rp.seq_of_tok.tolist() == [0] * 5 + [1] * 7 + [2] * 2


def test_library_exports_every_declared_symbol():
    from vila_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for name in ("vila_hip.h", "vila_hip_tuning.h"):                       # the boundary + the tuning / test switches (not part of it)
        hdr = open(os.path.join(root, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        found = set(re.findall(r"\b(vila_[a-z0-9_]+)\s*\(", hdr))
        assert found, f"no declarations parsed from {name}"
        if name == "vila_hip.h":
            assert not any("force" in f for f in found), "process-global switches do not belong in the boundary header"
        declared |= found
    declared.discard("vila_stream_t")
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    if not os.path.exists(_lib.LIB_PATH):
        from vila_amd import build
        build.build()
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vila_abi_version() == 1


def test_ops_refuse_cpu_tensors():
    from vila_amd import _lib, ops
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(_lib.VilaHipError, match="no CPU path"):
        ops.gemm(a, a)


def test_param_names_match_reference_state_dict():
    """prefixes llm. / vision_tower.vision_tower. / mm_projector. and the HF leaf names."""
    from vila_amd.modules import HipMultimodalProjector, HipQwen2ForCausalLM, HipSiglipVisionTower
    cfg = configs.tiny("mlp_downsample_3x3_fix")
    for cls, prefix, specs in ((HipQwen2ForCausalLM, "llm.", synthetic.llm_specs(cfg)),
                               (HipSiglipVisionTower, "vision_tower.", synthetic.vision_specs(cfg)),
                               (HipMultimodalProjector, "mm_projector.", synthetic.projector_specs(cfg))):
        m = cls(cfg, device="cpu")
        sd = m.state_dict()
        assert set(sd) == {n[len(prefix):] for n, _, _ in specs}
        for n, shape, _ in specs:
            assert tuple(sd[n[len(prefix):]].shape) == tuple(shape)
    llm = HipQwen2ForCausalLM(cfg, device="cpu")
    a = llm.model.layers.__getattr__("0").self_attn
    # q/k/v are views of ONE fused buffer (what the fused QKV kernels need) and survive a dtype round trip
    assert a.k_proj.weight.data_ptr() == a.q_proj.weight.data_ptr() + a.q_proj.weight.numel() * 2
    llm = llm.float().bfloat16()
    a = llm.model.layers.__getattr__("0").self_attn
    assert a.v_proj.weight.data_ptr() == a.k_proj.weight.data_ptr() + a.k_proj.weight.numel() * 2
    assert a.v_proj.bias.data_ptr() == a.k_proj.bias.data_ptr() + a.k_proj.bias.numel() * 2
