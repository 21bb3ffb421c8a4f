"""CPU: the C-ABI shared library builds, loads and exports every symbol include/mvldm.h declares, and
the ctypes mirror (mv_ldm_amd/_lib.py) matches the header's struct layout.  No compute calls."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "mvldm.h").read_text()


@pytest.fixture(scope="module")
def lib():
    from mv_ldm_amd import _build, _lib
    _build.build()
    return _lib.load()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(mvldm_[a-z0-9_]+)\s*\(", body)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from mv_ldm_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvldm.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.mvldm_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define MVLDM_ABI_VERSION (\d+)', HEADER).group(1))


def test_struct_layout_matches_header(tmp_path):
    """compile a tiny C program against the header and compare sizeof/offsetof with ctypes"""
    from mv_ldm_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mvldm.h"\nint main(){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mvldm_igemm_desc), sizeof(mvldm_op),'
                   'offsetof(mvldm_op,u), offsetof(mvldm_igemm_desc,c0), offsetof(mvldm_igemm_desc,out_scale),'
                   'offsetof(mvldm_igemm_desc,workspace_bytes), offsetof(mvldm_op,u.attention.scale),'
                   'sizeof(mvldm_pack_job), offsetof(mvldm_pack_job,n_out), offsetof(mvldm_pack_job,block0));return 0;}')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(_lib.IgemmDesc), C.sizeof(_lib.Op), _lib.Op.u.offset, _lib.IgemmDesc.c0.offset,
            _lib.IgemmDesc.out_scale.offset, _lib.IgemmDesc.workspace_bytes.offset,
            _lib.Op.u.offset + _lib._Attention.scale.offset,
            C.sizeof(_lib.PackJob), _lib.PackJob.n_out.offset, _lib.PackJob.block0.offset]
    assert got == want


def test_op_kind_enum_matches():
    from mv_ldm_amd import _lib
    m = re.search(r"enum\s*\{\s*MVLDM_OP_IGEMM = 1(.*?)\};", HEADER, re.S)
    names = ["MVLDM_OP_IGEMM"] + re.findall(r"(MVLDM_OP_[A-Z0-9_]+)", m.group(1))
    for i, n in enumerate(names, start=1):
        assert getattr(_lib, n.replace("MVLDM_", "")) == i


def test_product_library_reads_no_environment_knob(lib):
    """kernel-level A/B switches exist in experiment builds only (`knob_int`, csrc/common.h): no `MVLDM_*` variable name is compiled
    into the product library, so its kernels and dispatch cannot depend on the environment of the process"""
    from mv_ldm_amd import _lib
    blob = Path(_lib.LIB_PATH).read_bytes()
    assert b"MVLDM_" not in blob
