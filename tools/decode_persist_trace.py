"""Per-phase timing of the persistent decode token kernel (decode_persist.hip) from its in-kernel s_memrealtime stamps.
usage (GPU box): python tools/decode_persist_trace.py [--layers 28] [--ctx 769] [--steps 6] [--out gpurun_out/r06_persist_trace.txt]
Per phase kind (qkv, attn, o_proj, gate/up, down, lm_head), over all layers, steps and blocks:
  rows    wave 0 of the block: first row request -> last row done
  block   first row request -> every worker of the block done
  drain   the sync wave's write-through stores acknowledged
  barrier arrival -> generation seen
  stage   barrier open -> next phase's first row (activation staging)"""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--ctx", type=int, default=769)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from vila_amd import _lib, configs
    from vila_amd.vlm import build_model
    lib = _lib.load()
    cfg = configs.nvila_8b()
    cfg.vision.num_hidden_layers = 2
    cfg.llm.num_hidden_layers = a.layers
    model = build_model(cfg, seed=1)
    llm = model.llm
    g = torch.Generator().manual_seed(1)
    e = (torch.randn(1, a.ctx, cfg.llm.hidden_size, generator=g) * 0.5).to(torch.bfloat16).cuda()
    nb = torch.cuda.get_device_properties(0).multi_processor_count
    n_ph = a.layers * 5
    lib.vila_decode_force_persist(1)
    llm.generate(inputs_embeds=e, max_new_tokens=4, use_graph=False, eos_token_id=-1)          # warm
    buf = torch.zeros((nb, n_ph, 12), dtype=torch.int64, device="cuda")
    steps = []
    lib.vila_decode_persist_trace(buf.data_ptr(), nb)
    try:
        # eager steps: every launch overwrites the buffer, so run step by step
        for s in range(a.steps):
            llm.generate(inputs_embeds=e, max_new_tokens=1 + 1, use_graph=False, eos_token_id=-1)   # prefill + 1 decode step
            torch.cuda.synchronize()
            steps.append(buf.cpu().clone())
    finally:
        lib.vila_decode_persist_trace(None, 0)
    t = torch.stack(steps).double() * 0.01                      # [steps, nb, n_ph, 5] in us (100 MHz)
    kinds = ["qkv", "attn", "o_proj", "gate/up", "down"]
    lines = []
    tok = (t[:, :, -1, 3].max(1).values - t[:, :, 0, 0].min(1).values)
    lines.append(f"persistent layers kernel: {a.layers} layers, ctx {a.ctx}, {nb} blocks; first row request -> last layer's output drained: "
                 f"{tok.mean():.1f} us (min {tok.min():.1f}, max {tok.max():.1f}) over {a.steps} steps")
    lines.append(f"{'phase':9s} {'rows(w0)':>9s} {'block':>9s} {'drain':>7s} {'barrier':>8s} {'stage':>7s} | {'phase span (chip)':>18s}")
    tot = 0.0
    for k, name in enumerate(kinds):
        idx = [l * 5 + k for l in range(a.layers)]
        ph = t[:, :, idx, :]                                                    # [steps, nb, L, 5]
        rows = (ph[..., 1] - ph[..., 0])
        block = (ph[..., 2] - ph[..., 0])
        drain = (ph[..., 3] - ph[..., 2])
        if True:
            full = [i for i in idx if i + 1 < n_ph]                              # (the last phase of the kernel has no barrier / next phase)
            ph = t[:, :, full, :]
            rows, block, drain = rows[:, :, :len(full)], block[:, :, :len(full)], drain[:, :, :len(full)]
            bar = (ph[..., 4] - ph[..., 3])
            nxt = t[:, :, [i + 1 for i in full], 0]
            stage = nxt - ph[..., 4]
            idx = full
            # chip-level span of the phase: latest barrier-open minus latest barrier-open of the previous phase
            opens = t[:, :, :, 4].max(1).values                                 # [steps, n_ph]
            prev = torch.cat([t[:, :, 0:1, 0].min(1).values, opens[:, :-1]], 1)
            span = (opens - prev)[:, idx]
            per_wave = (ph[..., 5:12] - ph[..., 0:1]).mean((0, 1, 2))
            lines.append(f"{name:9s} {rows.mean():9.2f} {block.mean():9.2f} {drain.mean():7.2f} {bar.mean():8.2f} {stage.mean():7.2f} | "
                         f"{span.mean():8.2f} us x {len(idx)} | waves done at " + " ".join(f"{float(v):.2f}" for v in per_wave))
            tot += float(span.mean()) * len(idx)
    lines.append(f"sum of the layer phases' spans: {tot:.1f} us")
    # barrier anatomy: spread of arrivals and of openings
    arr = t[:, :, :-1, 3]
    opn = t[:, :, :-1, 4]
    lines.append(f"barrier: last arrival - first arrival {float((arr.max(1).values - arr.min(1).values).mean()):.2f} us; "
                 f"last arrival -> first open {float((opn.min(1).values - arr.max(1).values).mean()):.2f} us; "
                 f"first open -> last open {float((opn.max(1).values - opn.min(1).values).mean()):.2f} us")
    # is the arrival spread a property of the block (its CU / XCD) or random?  per-block mean of (own arrival - first arrival) in the gate/up phase
    gu = [l * 5 + 3 for l in range(a.layers)]
    rel = t[:, :, gu, 3] - t[:, :, gu, 3].min(1, keepdim=True).values        # [steps, nb, L]
    per_block = rel.mean((0, 2))
    within = rel.std((0, 2)).mean()
    lines.append(f"gate/up arrival lag per block: mean over blocks {float(per_block.mean()):.2f} us, std ACROSS blocks of the per-block mean {float(per_block.std()):.2f} us, "
                 f"std within a block over layers / steps {float(within):.2f} us")
    by_xcd = [float(per_block[x::8].mean()) for x in range(8)]
    lines.append("  per-block mean lag by block % 8: " + " ".join(f"{v:.2f}" for v in by_xcd))
    order = torch.argsort(per_block)
    lines.append("  slowest blocks: " + " ".join(f"{int(b)}:{float(per_block[b]):.1f}" for b in order[-12:]) + " | fastest: " + " ".join(f"{int(b)}:{float(per_block[b]):.1f}" for b in order[:8]))
    out = "\n".join(lines)
    print(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
