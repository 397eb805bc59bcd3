"""generate(do_sample=True) (llava_arch.py:833 -> HF GenerationMixin.sample; the reference's server samples by default: server.py:101-102,
185-187).  The on-device sampler must reproduce HF's processor chain — logits / temperature -> TopK -> TopP -> softmax — exactly as a
DISTRIBUTION (checked against the same chain written with torch ops) and draw from it correctly (empirical frequencies); the random
stream itself cannot match torch.multinomial, so token-level parity is not defined for this op."""
import pytest
import torch

from vila_amd import configs, synthetic

pytestmark = pytest.mark.gpu


def _hf_distribution(logits, temperature, top_k, top_p):
    """The oracle's restatement of HF's processor chain (pinned on CPU against transformers' own classes: tests/test_oracle_golden.py)."""
    from oracle import vila_oracle as O
    return O.sample_distribution(logits, temperature, top_k, top_p)


def test_sampler_matches_the_hf_executed_fixture():
    """tests/golden/sampling_hf.npz holds distributions produced by transformers' TemperatureLogitsWarper / TopKLogitsWarper /
    TopPLogitsWarper themselves; the on-device sampler's post-filter probabilities must equal them (fp32 softmax on the device: 2e-5)."""
    import os
    import numpy as np
    from vila_amd import ops
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampling_hf.npz"))
    n = 0
    while f"c{n}_params" in fx:
        V, temperature, top_k, top_p = fx[f"c{n}_params"]
        logits = torch.from_numpy(fx[f"c{n}_logits"]).cuda()
        ref = torch.from_numpy(fx[f"c{n}_probs"])
        tok, dist, ids = ops.sample(logits, float(temperature), int(top_k), float(top_p), seed=n, return_dist=True)
        got = torch.zeros(int(V), dtype=torch.float64)
        valid = ids >= 0
        got[ids[valid].long().cpu()] = dist[valid].double().cpu()
        assert float((got - ref).abs().max()) < 2e-5, (n, float((got - ref).abs().max()))
        assert float(ref[int(tok)]) > 0
        n += 1
    assert n >= 6


@pytest.mark.parametrize("V,temperature,top_k,top_p", [(1000, 1.0, 50, 1.0), (152064, 0.2, 50, 0.9), (152064, 0.7, 64, 0.5), (5000, 1.5, 1, 0.9), (300, 0.9, 40, 0.3),
                                                       # any k (round 4: exact radix selection, no 64-candidate cut): k > 64, k = 0 (HF: no top-k filter), k >= V,
                                                       # flat and peaked distributions, nucleus thresholds that keep thousands of tokens or one
                                                       (152064, 1.0, 100, 0.95), (152064, 0.7, 1000, 0.9), (152064, 1.0, 0, 0.9), (4097, 1.3, 65, 1.0),
                                                       (3000, 0.5, 5000, 0.999), (152064, 0.05, 500, 0.5), (70000, 2.0, 0, 1.0)])
def test_sampled_distribution_is_hf_processor_chain(V, temperature, top_k, top_p):
    from vila_amd import ops
    g = torch.Generator().manual_seed(V + top_k)
    logits = (torch.randn(V, generator=g) * 3).cuda()
    ref = _hf_distribution(logits.cpu(), temperature, top_k, top_p)
    tok, dist, ids = ops.sample(logits, temperature, top_k, top_p, seed=1, return_dist=True)
    got = torch.zeros(V, dtype=torch.float64)
    valid = ids >= 0
    got[ids[valid].long().cpu()] = dist[valid].double().cpu()
    assert abs(float(got.sum()) - 1.0) < 1e-5
    assert float((got - ref).abs().max()) < 2e-5, f"max |p - p_hf| = {float((got - ref).abs().max()):.3e}"
    assert float(ref[int(tok)]) > 0                       # the drawn token lies in the nucleus


def test_large_k_draws_follow_the_distribution():
    """The any-k path draws by prefix sums in index order: empirical frequencies over the 120-token support against the HF chain."""
    from vila_amd import ops
    V = 8192
    logits = (torch.randn(V, generator=torch.Generator().manual_seed(9)) * 2.5).cuda()
    ref = _hf_distribution(logits.cpu(), 0.9, 120, 0.97)
    n = 6000
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    draws = torch.cat([ops.sample(logits, 0.9, 120, 0.97, seed=4321, counter=ctr.fill_(i)) for i in range(n)]).cpu()
    counts = torch.zeros(V, dtype=torch.float64)
    counts.index_add_(0, draws, torch.ones(n, dtype=torch.float64))
    support = ref > 0
    assert float(counts[~support].sum()) == 0
    big = support & (ref * n >= 5)                        # chi-square over the cells with an expected count >= 5, the rest pooled
    exp = torch.cat([ref[big] * n, (ref[support & ~big].sum() * n).reshape(1)])
    obs = torch.cat([counts[big], counts[support & ~big].sum().reshape(1)])
    chi2 = float(((obs - exp) ** 2 / exp.clamp_min(1e-9)).sum())
    assert chi2 < 2.0 * len(exp) + 40, (chi2, len(exp))
    ctr.fill_(11)
    assert int(ops.sample(logits, 0.9, 120, 0.97, seed=4321, counter=ctr)) == int(draws[11])      # deterministic in (seed, counter)


def test_draws_follow_the_distribution_and_the_counter_drives_the_stream():
    from vila_amd import ops
    V = 4096
    logits = (torch.randn(V, generator=torch.Generator().manual_seed(5)) * 2).cuda()
    ref = _hf_distribution(logits.cpu(), 0.8, 20, 0.95)
    n = 4000
    counts = torch.zeros(V, dtype=torch.float64)
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    draws = []
    for i in range(n):
        ctr.fill_(i)
        draws.append(ops.sample(logits, 0.8, 20, 0.95, seed=1234, counter=ctr))
    draws = torch.cat(draws).cpu()
    counts.index_add_(0, draws, torch.ones(n, dtype=torch.float64))
    support = ref > 0
    assert float(counts[~support].sum()) == 0             # never outside top-k / nucleus
    # chi-square against the expected counts over the support (<= 20 cells): well below the 99.9 % quantile for 19 dof (43.8)
    exp = ref[support] * n
    chi2 = float(((counts[support] - exp) ** 2 / exp).sum())
    assert chi2 < 60, chi2
    # same seed + same counter -> same token; another counter or seed -> another stream
    ctr.fill_(7)
    a = int(ops.sample(logits, 0.8, 20, 0.95, seed=1234, counter=ctr)); b = int(ops.sample(logits, 0.8, 20, 0.95, seed=1234, counter=ctr))
    assert a == b == int(draws[7])
    other = torch.cat([ops.sample(logits, 0.8, 20, 0.95, seed=99, counter=ctr.fill_(i)) for i in range(64)]).cpu()
    assert not torch.equal(other, draws[:64])


def test_sampler_rejects_what_it_cannot_do():
    from vila_amd import ops
    logits = torch.randn(1000).cuda()
    for kw in (dict(temperature=0.0), dict(top_k=-1), dict(top_p=0.0), dict(top_p=1.5)):
        args = dict(temperature=1.0, top_k=50, top_p=1.0)
        args.update(kw)
        with pytest.raises(ValueError):
            ops.sample(logits, **args)


def test_generate_do_sample_graph_equals_eager_and_batch_equals_rows():
    """Sampling inside the captured hipGraph (the counter is the device-resident position) == eager launches for the same seed; a padded
    batch of 2 == the two rows generated alone (HF call contract of llava_arch.py:833 for B > 1), right-padded with pad_token_id."""
    from vila_amd.vlm import build_model
    cfg = configs.tiny("mlp_downsample")
    model = build_model(cfg, seed=2)
    px = synthetic.make_pixels(cfg, 2, 2).to(torch.bfloat16).cuda()
    ids = torch.stack([synthetic.make_prompt(cfg, 9, 1, 3), synthetic.make_prompt(cfg, 9, 1, 4)], 0)
    mask = torch.ones_like(ids, dtype=torch.bool); mask[1, 7:] = False            # second row is shorter (right padded)
    e, _, m = model._embed(ids, {"image": [px[0], px[1]]}, None, None, mask)
    kw = dict(max_new_tokens=10, do_sample=True, temperature=0.9, top_k=30, top_p=0.95, seed=77, eos_token_id=-1)
    g = model.llm.generate(inputs_embeds=e[:1], attention_mask=m[:1], use_graph=True, **kw)
    eg = model.llm.generate(inputs_embeds=e[:1], attention_mask=m[:1], use_graph=False, **kw)
    assert g.shape == (1, 10) and torch.equal(g, eg)
    greedy = model.llm.generate(inputs_embeds=e[:1], attention_mask=m[:1], max_new_tokens=10, eos_token_id=-1)
    # a new seed must NOT drop the captured graph (the seed is a device scalar the sampler reads): same session, same graph object
    g = model.llm.generate(inputs_embeds=e[:1], attention_mask=m[:1], use_graph=True, **kw)
    sess, graph = model.llm._decode, model.llm._decode.graph
    assert graph is not None
    other = model.llm.generate(inputs_embeds=e[:1], attention_mask=m[:1], use_graph=True, **dict(kw, seed=78))
    assert model.llm._decode is sess and model.llm._decode.graph is graph, "a different seed re-captured the decode graph"
    again = model.llm.generate(inputs_embeds=e[:1], attention_mask=m[:1], use_graph=True, **kw)
    assert torch.equal(again, g) and not torch.equal(other, g)                    # the replayed graph follows the seed scalar
    assert not (torch.equal(g, greedy) and torch.equal(other, greedy))            # it really samples
    both = model.llm.generate(inputs_embeds=e, attention_mask=m, **kw)
    r0 = model.llm.generate(inputs_embeds=e[:1], attention_mask=m[:1], **kw)
    r1 = model.llm.generate(inputs_embeds=e[1:], attention_mask=m[1:], **dict(kw, seed=78))
    assert both.shape == (2, 10) and torch.equal(both[0], r0[0]) and torch.equal(both[1], r1[0])
    # greedy batch with an EOS in one row: the finished row is padded with pad_token_id
    gb = model.llm.generate(inputs_embeds=e, attention_mask=m, max_new_tokens=6, eos_token_id=int(greedy[0, 2]), pad_token_id=0)
    assert gb.shape[0] == 2 and gb.shape[1] <= 6 and (gb[0, 3:] == 0).all()
    # the VLM-level call passes the generation kwargs through (llava_arch.py:823-833)
    out = model.generate(input_ids=ids[:1], media={"image": [px[0]]}, **kw)
    assert torch.equal(out, g)
