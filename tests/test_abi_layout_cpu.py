"""The C-ABI header is plain C and the ctypes mirror has the SAME layout (the boundary is `extern "C"`, plain pointers and sizes).

`include/vila_hip.h` is compiled by gcc as C99 into a program that prints sizeof / offsetof of every struct the Python side mirrors
(`vila_amd/_lib.py`); every number must equal ctypes'.  A field added on one side only (as `VilaSftBatch.pools` could have been) shifts
everything behind it — the GPU tests would read garbage; this catches it without a GPU."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _structs():
    from vila_amd import _lib
    return [(n, getattr(_lib, n)) for n in dir(_lib)
            if n.startswith("Vila") and isinstance(getattr(_lib, n), type) and issubclass(getattr(_lib, n), C.Structure)]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_plain_c_and_the_ctypes_mirror_has_its_layout(tmp_path):
    structs = _structs()
    assert len(structs) >= 14
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vila_hip.h"', '#include "vila_hip_tuning.h"', "int main(void) {"]
    for name, cls in structs:
        lines.append(f'    printf("{name} sizeof %zu\\n", sizeof({name}));')
        for field in cls._fields_:
            lines.append(f'    printf("{name} {field[0]} %zu\\n", offsetof({name}, {field[0]}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, "include/vila_hip.h must compile as C99 (a field missing on the C side shows up here by name):\n" + r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {}
    for line in out.splitlines():
        s, f, v = line.split()
        got[(s, f)] = int(v)
    for name, cls in structs:
        assert got[(name, "sizeof")] == C.sizeof(cls), f"sizeof({name}): header {got[(name, 'sizeof')]} vs ctypes {C.sizeof(cls)} (a field missing in vila_amd/_lib.py?)"
        for field in cls._fields_:
            assert got[(name, field[0])] == getattr(cls, field[0]).offset, f"{name}.{field[0]}: header offset {got[(name, field[0])]} vs ctypes {getattr(cls, field[0]).offset}"
