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


def test_product_library_reads_one_environment_variable():
    """Development knobs (A/B switches, probes) exist only in a -DHMV_DEV_KNOBS build: a stray HMV_* variable on a user's box must
    not be able to change kernels, K orders or numerics.  Every "HMV_[A-Z0-9_]+" token in the product library's bytes is either the
    one variable it reads (HMV_GRAPHS) or an enum name quoted in an error message / an assertion text."""
    if os.environ.get("HMV_LIB"):
        return   # an A/B run against another build (possibly the -dev one): nothing to assert about that file
    data = open(_lib.LIB_PATH, "rb").read()
    tokens = set(m.decode() for m in re.findall(rb"HMV_[A-Z0-9_]+[a-z0-9]*", data))
    allowed = {"HMV_GRAPHS", "HMV_F32", "HMV_F16", "HMV_F32X3", "HMV_POS_SIN"}
    assert tokens <= allowed, sorted(tokens - allowed)
    assert "HMV_GRAPHS" in tokens
    # and the sources spell every other knob through the gated macro
    for f in os.listdir(os.path.join(ROOT, "handmvnet_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(ROOT, "handmvnet_amd", "csrc", f)).read()
            for m in re.finditer(r"(?<![A-Za-z_])getenv\(\s*\"(\w+)\"", text):
                assert m.group(1) == "HMV_GRAPHS", (f, m.group(1))
    assert b"f32x3" in _lib.load().hmv_version() and b"f16" in _lib.load().hmv_version()


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
