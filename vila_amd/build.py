"""Build libvila_hip.so (gfx950 only) in-tree with hipcc.  `python -m vila_amd.build [--force] [--verbose]`.

The shared library has no PyTorch / Python dependency: it is the C-ABI drop-in boundary (include/vila_hip.h).
It is built into vila_amd/lib/ so that it travels with the repo snapshot to the GPU box (git-ignored, not
gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libvila_hip.so")
SOURCES = ["api.hip", "gemm.hip", "attn.hip", "elementwise.hip", "gemv.hip", "train.hip", "attn_bwd.hip", "gemm256.hip", "s2.hip", "gemv_w4.hip", "gemm_ring.hip", "video.hip", "gemm256_cm.hip", "gemm_i8.hip", "sample.hip", "sft.hip", "attn_bwd_dma.hip", "decode_batch.hip", "gemm_ring_splitk.hip", "decode_persist.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]  # fp-contract off: keep the HF rounding order (bf16(q*cos)+bf16(rot*sin)) explicit


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libvila_hip.so cannot be built (ROCm toolchain required)")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "vila_hip.h"))
    return hdrs


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    files = [os.path.join(CSRC, s) for s in _sources()] + _deps()
    return any(os.path.getmtime(f) > t for f in files)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    hdr_t = max(os.path.getmtime(f) for f in _deps())

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return obj
        cmd = [cc, *FLAGS, "-c", path, "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
        if verbose or r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, _sources()))
    tmp = LIB + ".tmp"
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    os.replace(tmp, LIB)
    return LIB


def build_tools(force: bool = False) -> str:
    """tools/gemm_bench: the torch-free GEMM micro-benchmark the wgrad PMC passes (tools/pmc_gemm_sft.sh) run under rocprofv3.  Built next
    to the library so that it travels to the GPU box with it (git-ignored, not gpurun-ignored)."""
    root = os.path.dirname(HERE)
    src, exe = os.path.join(root, "tools", "gemm_bench.cpp"), os.path.join(root, "tools", "gemm_bench")
    if not os.path.exists(src):
        return ""
    if not force and os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return exe
    r = subprocess.run([hipcc(), "-O2", "-std=c++17", src, "-o", exe, "-L" + LIBDIR, "-lvila_hip", "-Wl,-rpath,$ORIGIN/../vila_amd/lib"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on tools/gemm_bench.cpp:\n{r.stderr[-4000:]}")
    return exe


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(p)
    print(build_tools(force="--force" in sys.argv))
