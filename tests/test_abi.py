"""The C-ABI library loads and exports every symbol include/garden_vis.h declares (no compute calls: there is
no GPU in the CPU test tier), and the product path fails loudly instead of falling back when there is no device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "garden_vis.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gv_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from garden_amd import lib
    assert header_functions() == sorted(lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    from garden_amd import lib
    handle = lib.load()
    for name in header_functions():
        assert hasattr(handle, name), f"libgarden_vis.so does not export {name}"
    assert handle.gv_abi_version() == 4


def test_struct_sizes_match_header():
    from garden_amd import lib
    assert ctypes.sizeof(lib.GvConfig) == 16
    assert ctypes.sizeof(lib.GvView) == 16 * 4 + 4 * 4 + 4 * 4 + 4
    assert ctypes.sizeof(lib.GvTransformLayout) == 32 and ctypes.sizeof(lib.GvMeshLayout) == 20


def test_header_is_plain_c_and_the_binding_structs_match_it(tmp_path):
    """include/garden_vis.h is what an engine binds: it compiles as strict C99 and as C++11 (warnings are errors), and
    the sizes / field offsets the C compiler gives every struct equal the ctypes mirror's (garden_amd/lib.py)."""
    import subprocess
    from garden_amd import lib
    structs = ["GvConfig", "GvView", "GvTransformLayout", "GvMeshLayout", "GvResult", "GvDeviceResult", "GvRecordLayout", "GvStats", "GvExchangeFrame", "GvExchangeItem"]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "garden_vis.h"', "int main(void) {"]
    for name in structs:
        cls = getattr(lib, name)
        lines.append(f'    printf("{name} %zu", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'    printf(" %zu", offsetof({name}, {field}));')
        lines.append('    printf("\\n");')
    lines += ["    return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines) + "\n")
    inc = os.path.join(ROOT, "include")
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, str(src), "-o", str(exe)], check=True)
    subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o",
                    str(tmp_path / "abi_cpp.o")], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for line in filter(None, out):
        name, size, *offsets = line.split()
        cls = getattr(lib, name)
        assert int(size) == ctypes.sizeof(cls), name
        assert [int(o) for o in offsets] == [getattr(cls, field).offset for field, _ in cls._fields_], name


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "garden_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "gv_oracle" not in src and "oracle_py" not in src and "from oracle" not in src, f
    assert "oracle" not in open(os.path.join(ROOT, "include", "garden_vis.h")).read().replace("oracle/", "")


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU tier")
    from garden_amd.lib import GV_E_NODEVICE, GpuVisibility, GvError
    with pytest.raises(GvError) as e:
        GpuVisibility(device=0)
    assert e.value.code == GV_E_NODEVICE and "no CPU fallback" in str(e.value)


def test_a_missing_rccl_library_is_an_error_code_not_a_crash():
    """GV_RCCL_LIBRARY names a library that does not exist: gv_exchange_unique_id returns GV_E_RCCL (the loader asked dlerror()
    twice — the second answer is NULL — and built a std::string from it: a segfault, found by bench.py's fallback test in round 4).
    In a child process: the loader's verdict is per process."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from garden_amd import lib; h = lib.load(); b = ctypes.create_string_buffer(128); "
            "rc = h.gv_exchange_unique_id(b); print('rc', rc); sys.exit(0 if rc == lib.GV_E_RCCL else 1)" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GV_RCCL_LIBRARY="/nonexistent/librccl_nowhere.so"))
    assert p.returncode == 0, (p.stdout, p.stderr[-2000:])


def test_integration_statement_table_points_at_the_shim_lines_it_names():
    """INTEGRATION.md §2a maps every statement of prepareMeshes (mesh.cpp:331-553) to a line of the shim: the rows of the gate
    (mesh.cpp:426 / :482), of the flags (:339, :488-490) and of the pool accessor (:410) must still name lines that do that."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = open(os.path.join(root, "garden_amd", "csrc", "host", "gpu_visibility_system.hpp")).read().splitlines()
    rows = {line.split("|")[1].strip(): line for line in open(os.path.join(root, "INTEGRATION.md")) if line.startswith("| ")}

    def named(row):
        last_column = rows[row].rstrip().rstrip("|").rsplit("|", 1)[1]  # (a cell of the middle column may hold an escaped pipe)
        return [shim[int(n) - 1] for n in re.findall(r"`:(\d+)`", last_column)]

    gate = named("**426**")
    assert len(gate) >= 3 and "isDrawReady(-1)" in gate[0] and "isDrawReady(shadowPasses[s].index(s))" in gate[1] and "continue" in gate[2], gate
    assert "hasAnyRefr = hasAnyOIT = hasAnyTD = false" in named("339")[0]
    assert "hasAnyRefr |=" in named("488-490")[0]
    assert "getMeshComponentPool()" in named("410-412")[0]
    assert "shadowIndex != bufferIndex" in named("252 (in `prepareSortedMeshes`)")[0]
