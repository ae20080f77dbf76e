"""CPU-side checks of the boundary: the shared library loads without a GPU and exports every symbol
include/cafe_hip.h declares (no compute is called here)."""
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "cafe_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src)
    skip = {"defined", "C"}
    return sorted({n for n in names if n not in skip and not n.startswith("__")})


def test_library_exports_every_declared_symbol():
    from cafe_amd import _lib
    L = _lib.load()
    decl = _declared()
    assert len(decl) >= 25
    for n in decl:
        assert hasattr(L, n), f"{n} declared in include/cafe_hip.h but not exported"
    assert set(decl) == set(_lib.SYMBOLS), set(decl) ^ set(_lib.SYMBOLS)


def test_no_oracle_in_product_package():
    """The product package must never import or link the oracle."""
    pkg = os.path.join(ROOT, "cafe_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "cafe_oracle" not in txt and "liboracle" not in txt, f


def test_header_is_plain_c(tmp_path):
    """include/cafe_hip.h is the drop-in boundary: it must compile as C99 on its own (no C++ or torch types)."""
    import os
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "cafe_hip.h"\nint main(void) { return cafe_version() < 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(root, "include"), str(src)])
