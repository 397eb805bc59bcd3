"""YARDSTICK, not product code: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the exact GEMM shapes of the prefill and the
SFT step, on the same box, random bf16 data, cold-ish weights (operands cycled so the 256-MB infinity cache cannot hold them).  The product path never
calls a library GEMM (north_star: hand-written MFMA kernels); this only tells how much headroom the hand-written kernels have left per shape.
    gpurun -- python tools/exp/gemm_yardstick.py > gpurun_out/gemm_yardstick.log
Layouts: fwd  C[M,N] = A[M,K] . W[N,K]^T      (x @ w.t())
         dgrad dX[M,K] = dY[M,N] . W[N,K]      (dy @ w)
         wgrad dW[N,K] = dY[M,N]^T . X[M,K]    (dy.t() @ x)
"""
import sys
import torch

dev = "cuda"
torch.manual_seed(0)


def bench(fn, reps=20):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps          # us


def case(name, M, N, K, kinds=("fwd", "dgrad", "wgrad")):
    nbuf = max(2, min(8, int(600e6 // max(1, (N * K * 2)))))      # cycle enough weight copies to spill the infinity cache
    W = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
    X = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(2)]
    DY = [torch.randn(M, N, device=dev).to(torch.bfloat16) for _ in range(2)]
    fl = 2.0 * M * N * K
    out = []
    if "fwd" in kinds:
        t = bench(lambda i: torch.matmul(X[i & 1], W[i % nbuf].t()))
        out.append(f"fwd {t:8.1f} us {fl / t / 1e6:7.1f} TF/s")
    if "dgrad" in kinds:
        t = bench(lambda i: torch.matmul(DY[i & 1], W[i % nbuf]))
        out.append(f"dgrad {t:8.1f} us {fl / t / 1e6:7.1f} TF/s")
    if "wgrad" in kinds:
        t = bench(lambda i: torch.matmul(DY[i & 1].t(), X[i & 1]))
        out.append(f"wgrad {t:8.1f} us {fl / t / 1e6:7.1f} TF/s")
    print(f"{name:28s} M={M:6d} N={N:6d} K={K:6d} | " + " | ".join(out), flush=True)


print(torch.__version__, torch.cuda.get_device_name(0), "preferred blas:", torch.backends.cuda.preferred_blas_library())
print("# SFT step shapes (T = 3076 tokens = 4 x 769)")
case("LLM qkv", 3076, 4608, 3584)
case("LLM o_proj", 3076, 3584, 3584)
case("LLM gate (one of gate/up)", 3076, 18944, 3584)
case("LLM down", 3076, 3584, 18944)
case("ViT qkv (4 images)", 4096, 3456, 1152)
case("ViT fc1", 4096, 4304, 1152)
case("ViT fc2", 4096, 1152, 4304)
print("# prefill shapes (S = 769, one image)")
case("LLM qkv S=769", 769, 4608, 3584, ("fwd",))
case("LLM o_proj S=769", 769, 3584, 3584, ("fwd",))
case("LLM gate+up S=769 (N = 2F)", 769, 37888, 3584, ("fwd",))
case("LLM down S=769", 769, 3584, 18944, ("fwd",))
case("ViT qkv M=1024", 1024, 3456, 1152, ("fwd",))
case("ViT out M=1024", 1024, 1152, 1152, ("fwd",))
case("ViT fc1 M=1024", 1024, 4304, 1152, ("fwd",))
case("ViT fc2 M=1024", 1024, 1152, 4304, ("fwd",))
print("# square")
case("4096^3", 4096, 4096, 4096, ("fwd",))
case("8192^3", 8192, 8192, 8192, ("fwd",))
