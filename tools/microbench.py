"""Kernel micro-benchmarks on one MI355X (HIP-event timed on the current stream).  python tools/microbench.py [gemm|gemv|attn|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vila_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)


def bench_gemm_tiles():
    from vila_amd import _lib
    lib = _lib.load()
    print("== GEMM tile shapes (TFLOP/s): auto | 128x128 | 128x64 | 256x128 | 256x256dma | ring4 | ring3 | ring128 ==")
    for M, N, K in [(769, 4608, 3584), (769, 3584, 3584), (769, 3584, 18944), (1024, 1152, 1152), (1024, 1152, 4304), (1024, 4304, 1152),
                    (1024, 3456, 1152), (4096, 1152, 1152), (4096, 1152, 4304), (4096, 4304, 1152), (4096, 3456, 1152), (1152, 1152, 4096),
                    (4304, 1152, 4096), (1152, 4304, 4096), (3456, 1152, 4096), (256, 3584, 4608), (3076, 3584, 3584), (3076, 3584, 18944), (3584, 3584, 3080), (18944, 3584, 3080), (4096, 4096, 4096), (8192, 8192, 8192)]:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        res = []
        for tile in (0, 1, 2, 3, 4, 6, 7, 8):
            lib.vila_gemm_force_tile(tile)
            t = timeit(lambda: ops.gemm(a, w), iters=10)
            res.append(2.0 * M * N * K / t / 1e12)
        lib.vila_gemm_force_tile(0)
        print(f"M={M:5d} N={N:6d} K={K:6d}: " + " | ".join(f"{r:7.1f}" for r in res))


def bench_prefill_shapes():
    from vila_amd import _lib
    lib = _lib.load()
    print("== S=769 prefill GEMMs (us): auto | forced 128-tile kernel | gemm256 (4) | split-K (5) ==")
    M = 769
    ws = torch.empty(8 * M * 4608, device="cuda", dtype=torch.float32)
    for N, K, epi in [(4608, 3584, 0), (3584, 3584, 0), (18944, 3584, 3), (3584, 18944, 0)]:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        w2 = rnd(N, K, scale=K ** -0.5) if epi == 3 else None
        res = []
        for tile in (0, 1, 4, 5):
            lib.vila_gemm_force_tile(tile)
            t = timeit(lambda: ops.gemm(a, w, w2=w2, epi=epi, ws=ws), iters=20)
            res.append(t * 1e6)
        lib.vila_gemm_force_tile(0)
        print(f"N={N:6d} K={K:6d} epi={epi}: " + " | ".join(f"{r:7.1f}" for r in res))


def bench_gemm():
    print("== GEMM bf16 (TFLOP/s) ==")
    for M, N, K, epi in [(769, 4608, 3584, 0), (769, 3584, 3584, 0), (769, 18944, 3584, 3), (769, 3584, 18944, 0),
                         (1024, 3456, 1152, 0), (1024, 1152, 1152, 0), (1024, 4304, 1152, 1), (1024, 1152, 4304, 0),
                         (4096, 4096, 4096, 0), (8192, 8192, 8192, 0), (3076, 18944, 3584, 3)]:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        w2 = rnd(N, K, scale=K ** -0.5) if epi == 3 else None
        t = timeit(lambda: ops.gemm(a, w, w2=w2, epi=epi), iters=10)
        fl = 2.0 * M * N * K * (2 if epi == 3 else 1)
        tt = timeit(lambda: torch.matmul(a, w.t()), iters=10)
        print(f"M={M:5d} N={N:6d} K={K:6d} epi={epi}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s   (torch/hipBLASLt {2.0*M*N*K/tt/1e12:7.1f} TF/s)")


def bench_gemv():
    print("== GEMV bf16 (GB/s of weight bytes) ==")
    for N, K, mode in [(4608, 3584, 0), (3584, 3584, 0), (18944, 3584, 1), (3584, 18944, 0), (152064, 3584, 0)]:
        x, w = rnd(K), rnd(N, K, scale=K ** -0.5)
        w2 = rnd(N, K, scale=K ** -0.5) if mode else None
        g = rnd(K)
        t = timeit(lambda: ops.gemv(x, w, w2=w2, norm_w=g if K == 3584 else None, eps=1e-6), iters=50)
        by = N * K * 2 * (2 if mode else 1)
        print(f"N={N:6d} K={K:6d} mode={mode}: {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s")


def bench_w4():
    """W4A16 GEMVs; rotates over enough weight copies (> 600 MB) that the 256 MB Infinity Cache cannot serve re-reads."""
    from vila_amd.quant import W4Matrix
    print("== GEMV W4A16 (GB/s of packed weight bytes: 0.53125 B/weight) ==")
    for N, K, mode in [(4608, 3584, 0), (3584, 3584, 0), (18944, 3584, 1), (3584, 18944, 0)]:
        by = N * K * (2 if mode else 1) * 17 // 32
        ncopy = max(2, int(6e8 // by))
        mats = []
        for i in range(ncopy):
            w = rnd(N, K, scale=K ** -0.5)
            mats.append(W4Matrix.pack(w, rnd(N, K, scale=K ** -0.5) if mode else None, keep_logical=False))
        x, g = rnd(K), rnd(K)
        state = {"i": 0}

        def run():
            state["i"] = (state["i"] + 1) % ncopy
            ops.gemv_w4(x, mats[state["i"]], norm_w=g if K == 3584 else None, eps=1e-6)
        t = timeit(run, iters=4 * ncopy, warm=ncopy)
        print(f"N={N:6d} K={K:6d} mode={mode} copies={ncopy}: {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s")
        del mats


def bench_attn():
    print("== attention fwd (TFLOP/s, 4*T*T*D*H (x0.5 causal)) ==")
    for T, Hq, Hkv, D, causal, nseq in [(1024, 16, 16, 72, False, 1), (8192, 16, 16, 72, False, 8), (769, 28, 4, 128, True, 1),
                                         (4096, 28, 4, 128, True, 1), (16384, 28, 4, 128, True, 1)]:
        q, k, v = rnd(T, Hq, D), rnd(T, Hkv, D), rnd(T, Hkv, D)
        t = timeit(lambda: ops.attn_fwd(q, k, v, causal, n_seq=nseq), iters=10)
        n = T // nseq
        fl = 4.0 * n * n * D * Hq * nseq * (0.5 if causal else 1.0)
        print(f"T={T:6d} Hq={Hq} Hkv={Hkv} D={D} causal={causal} nseq={nseq}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s")


def bench_attn_bwd():
    print("== attention bwd: delta + dQ + dK/dV (TFLOP/s, 10*T*T*D*H per sequence (x0.5 causal)) ==")
    for n, nseq, Hq, Hkv, D, causal in [(769, 4, 28, 4, 128, True), (1024, 4, 16, 16, 72, False), (1024, 64, 16, 16, 72, False), (4096, 1, 28, 4, 128, True)]:
        T = n * nseq
        q, k, v, do = rnd(T, Hq, D), rnd(T, Hkv, D), rnd(T, Hkv, D), rnd(T, Hq, D)
        cu = torch.arange(0, T + 1, n, dtype=torch.int32, device="cuda")
        kw = dict(cu_seqlens=cu, max_seqlen=n) if causal else dict(n_seq=nseq)
        o, lse = ops.attn_fwd(q, k, v, causal, return_lse=True, **kw)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        t = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, causal, dq, dk, dv, **kw), iters=10)
        tf = timeit(lambda: ops.attn_fwd(q, k, v, causal, return_lse=True, **kw), iters=10)
        fl = 10.0 * n * n * D * Hq * nseq * (0.5 if causal else 1.0)
        print(f"n={n:5d} x{nseq:3d} Hq={Hq} Hkv={Hkv} D={D} causal={causal}: bwd {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s   (fwd {tf*1e6:8.1f} us  {0.4*fl/tf/1e12:7.1f} TF/s)")


def bench_tower():
    """The SigLIP tower + projector of NVILA-8B on 1 / 8 images, weights read cold every pass (0.8 GB > the 256-MB infinity cache)."""
    from vila_amd import configs
    from vila_amd.vlm import build_model
    cfg = configs.nvila_8b()
    cfg.llm.num_hidden_layers = 1
    m = build_model(cfg, seed=0)
    for n in (1, 8):
        px = torch.randn(n, 3, 448, 448, device="cuda").to(torch.bfloat16)
        t = timeit(lambda: m.encode_images(px), iters=10, warm=2)
        print(f"tower+projector, {n} image(s): {t*1e3:8.3f} ms")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    print(torch.cuda.get_device_name(0))
    if what in ("prefill",):
        bench_prefill_shapes()
    if what in ("tiles",):
        bench_gemm_tiles()
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("gemv", "all"):
        bench_gemv()
    if what in ("w4",):
        bench_w4()
    if what in ("attn", "all"):
        bench_attn()
    if what in ("tower",):
        bench_tower()
    if what in ("attn_bwd", "all"):
        bench_attn_bwd()
