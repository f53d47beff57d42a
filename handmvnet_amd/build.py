"""Builds libhandmv.so (the HIP engine) in-tree with hipcc for gfx950.

    python -m handmvnet_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting handmvnet_amd/libhandmv.so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libhandmv.so")
SOURCES = ["conv_igemm.hip", "conv_stream.hip", "conv_gemm8.hip", "conv_hs.hip", "conv_ht.hip", "conv_m16.hip", "gemm_x3.hip", "conv_rds.hip", "misc_kernels.hip", "fusion_kernels.hip", "hr_fuse.hip", "engine.hip", "metrics.hip"]
HEADERS = [os.path.join(CSRC, "kernels.h"), os.path.join(os.path.dirname(HERE), "include", "handmv.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """A/B builds for one-box comparisons: compiles every source with extra -D flags into build/libhandmv_<name>.so
    (load it with HMV_LIB=build/libhandmv_<name>.so).  Development tool; the product is build()."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(os.path.dirname(HERE), "build", name)
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in SOURCES:
        op = os.path.join(out_dir, src.replace(".hip", ".o"))
        objs.append(op)
        cmd = [hipcc] + FLAGS + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, src), "-o", op]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    lib = os.path.join(os.path.dirname(HERE), "build", f"libhandmv_{name}.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m handmvnet_amd.build --variant stage HMV_TOUT=0
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
