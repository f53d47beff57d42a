"""C-ABI checks that need no GPU: the shared library loads and exports every symbol that
include/handmv.h declares; argument validation that happens before any HIP call."""
import ctypes
import os
import re

from handmvnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "handmv.h")).read()
    declared = sorted(set(re.findall(r"\b(hmv_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    lib = _lib.load()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} is declared in include/handmv.h but not exported"
    assert set(declared) == set(_lib.SYMBOLS)
    assert b"gfx950" in lib.hmv_version()


def test_config_struct_matches_header_and_is_validated():
    lib = _lib.load()
    assert ctypes.sizeof(_lib.HmvConfig) == 18 * 4          # incl. the fusion kind added for cross_attn_learnable_query
    h = ctypes.c_void_p()
    c = _lib.HmvConfig()
    c.struct_size = 12                      # wrong ABI size is rejected before anything touches HIP
    assert lib.hmv_create(ctypes.byref(c), ctypes.byref(h)) == 1
    assert b"struct_size" in lib.hmv_last_error(None)
    c.struct_size = ctypes.sizeof(_lib.HmvConfig)
    c.backbone = 7
    assert lib.hmv_create(ctypes.byref(c), ctypes.byref(h)) == 1
    assert b"18, 34, 50_paper" in lib.hmv_last_error(None)
    c.backbone, c.num_views, c.fusion_layers = 2, 8, 4
    assert lib.hmv_create(ctypes.byref(c), ctypes.byref(h)) == 1
    assert b"odd" in lib.hmv_last_error(None)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under handmvnet_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "handmvnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "hmvo_" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
