"""The 256-wide GEMM's tile order (vila_amd/csrc/gemm256_kernel.h: gemm256_tile_of, round 3) is shared by the kernel, the whole-rounds + K-sliced-tail
launch policy and the tail's reduce kernel, so it must be a bijection from tile ids onto the tile grid for EVERY grid and group width — checked
here on the host with the real header (hipcc compiles the host side without a GPU), together with the property the order exists for: 32
consecutive ids (what one XCD works on at a time) touch few distinct operand tiles."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "gemm256_kernel.h"
#include <cstdio>
#include <set>
#include <vector>
int g_gemm256_group = -1;
int g_gemm256_ex = -1;
int main() {
    int bad = 0, cases = 0;
    for (int tiles_m = 1; tiles_m <= 80; ++tiles_m)
        for (int tiles_n = 1; tiles_n <= 20; ++tiles_n)
            for (int grp : {0, 1, 2, 3, 4, 5, 8}) {
                std::vector<int> seen(tiles_m * tiles_n, 0);
                for (int id = 0; id < tiles_m * tiles_n; ++id) {
                    int tm, tn;
                    gemm256_tile_of(id, tiles_m, tiles_n, grp, tm, tn);
                    if (tm < 0 || tm >= tiles_m || tn < 0 || tn >= tiles_n) { ++bad; continue; }
                    ++seen[tn * tiles_m + tm];
                }
                for (int v : seen) bad += (v != 1);
                ++cases;
            }
    // locality: grouped by 4 on a 74 x 14 grid (wgrad gate), any 32 consecutive ids inside a full group touch <= 9 + 4 operand tiles
    int worst = 0, worst_strip = 0;
    for (int start = 0; start + 32 <= 4 * 74 * 3; start += 7) {
        if (start % (4 * 74) + 32 > 4 * 74) continue;             // windows that straddle two column groups see both

        std::set<int> a, b, a0, b0;
        for (int id = start; id < start + 32; ++id) {
            int tm, tn;
            gemm256_tile_of(id, 74, 14, 4, tm, tn); a.insert(tm); b.insert(tn);
            gemm256_tile_of(id, 74, 14, 0, tm, tn); a0.insert(tm); b0.insert(tn);
        }
        if ((int)(a.size() + b.size()) > worst) worst = (int)(a.size() + b.size());
        if ((int)(a0.size() + b0.size()) > worst_strip) worst_strip = (int)(a0.size() + b0.size());
    }
    printf("%d grids, %d violations; operand tiles per 32 ids: grouped %d, strips %d; auto rule: %d %d %d\n", cases, bad, worst, worst_strip,
           gemm256_group(74, 14, false), gemm256_group(13, 74, false), gemm256_group(74, 14, true));
    return (bad != 0) || worst > 13 || worst_strip < 33 || gemm256_group(74, 14, false) != 4 || gemm256_group(13, 74, false) != 0 || gemm256_group(74, 14, true) != 0;
}
'''


def test_tile_order_is_a_bijection_and_local(tmp_path):
    cc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        pytest.skip("hipcc not available")
    src = tmp_path / "tile_order_check.hip"
    src.write_text(SRC)
    exe = tmp_path / "tile_order_check"
    r = subprocess.run([cc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "vila_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                        "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    print(r.stdout, file=sys.stderr)
    assert r.returncode == 0, r.stdout
