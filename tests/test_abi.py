"""The C ABI: the shared library builds, loads, and exports every symbol include/hbmrag.h
declares with the signature the ctypes layer binds (no compute without a GPU)."""
import os
import re
import subprocess

import pytest

from advanced_rag import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hbmrag.h")


def declared_symbols():
    text = open(HEADER).read()
    return re.findall(r"^HR_API\s+[\w\s\*]+?\b(hr_\w+)\s*\(", text, flags=re.M)


def test_header_and_binding_list_the_same_symbols():
    decl = declared_symbols()
    assert len(decl) >= 24 and len(set(decl)) == len(decl)
    assert set(decl) == set(nat.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = nat.load_library()
    for name in declared_symbols():
        assert getattr(lib, name) is not None
    assert lib.hr_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", nat.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (hr_\w+)", out))
    assert exported == set(declared_symbols())  # nothing else leaks (-fvisibility=hidden)


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "hbmrag.h"\nint main(void){ return HR_OK + HR_MAX_TOPK * 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                    str(tmp_path / "t.o")], check=True)


def test_code_object_targets_gfx950():
    data = open(nat.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    assert b"gfx942" not in data and b"sm_" not in data  # one target, no dual paths


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_a_gpu():
    with pytest.raises(nat.HbmRagError) as ei:
        nat.ShardHandle(8)
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "advanced-rag-milvus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
