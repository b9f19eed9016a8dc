"""CPU-side checks of the drop-in boundary: the shared library builds, loads, and exports exactly the
entry points include/sgp_hip.h declares; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "sgp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sgp_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from gaussianprocessnode_amd import _lib
    assert declared_functions() == sorted(_lib.EXPORTS)


def test_library_builds_loads_and_exports_every_symbol():
    from gaussianprocessnode_amd import _build, _lib
    path = _build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.sgp_abi_version() == 1


def test_library_is_gfx950_only():
    from gaussianprocessnode_amd import _build
    blob = open(_build.build(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import gaussianprocessnode_amd as G
    with pytest.raises(G.SGPError):
        G.SGPDevice(10, 4, 1)
    with pytest.raises(G.SGPError):
        G.kernelmatrix([[0.0]], [[1.0]], 1.0, [1.0])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gaussianprocessnode_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert not re.search(r"(import_module|__import__)\(.*oracle", src), f
