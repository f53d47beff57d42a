"""ctypes wrapper around the CPU oracle (oracle/hmv_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under handmvnet_amd/ imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BACKBONE = {"18": 0, "34": 1, "50_paper": 2, "w40": 3, "w64": 4}


class _Cfg(ctypes.Structure):
    _fields_ = [("backbone", ctypes.c_int), ("n_levels", ctypes.c_int), ("channels", ctypes.c_int * 4),
                ("num_views", ctypes.c_int), ("image_size", ctypes.c_int), ("heatmap_size", ctypes.c_int),
                ("pos_mask", ctypes.c_int), ("fusion_layers", ctypes.c_int), ("use_gcn", ctypes.c_int),
                ("fusion", ctypes.c_int)]


def build(force: bool = False) -> None:
    """Compile both oracle variants with gcc (seconds)."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


_LIBS: Dict[str, ctypes.CDLL] = {}


def usable_cpus() -> int:
    """CPUs this process may really use: min(affinity mask, cgroup cpu quota).  An OpenMP team
    sized by the host's core count on a box that grants a 16-CPU share just thrashes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env)) if int(env) > 0 else n
    return max(1, n)


def _lib(acc: str) -> ctypes.CDLL:
    if acc not in ("f32", "f64"):
        raise ValueError(acc)
    if acc not in _LIBS:
        path = os.path.join(_HERE, f"liboracle_hmv_{acc}.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.hmvo_last_error.restype = ctypes.c_char_p
        lib.hmvo_forward.restype = ctypes.c_int
        lib.hmvo_fuse_tokens.restype = ctypes.c_int
        lib.hmvo_set_num_threads(min(usable_cpus(), 64))
        _LIBS[acc] = lib
    return _LIBS[acc]


def _fp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class Oracle:
    """One oracle instance = one set of weights loaded into one of the two libraries.
    (The C side keeps a single global tensor table per library, so use one live instance
    per accumulation type at a time.)"""

    def __init__(self, cfg, state_dict: Dict[str, np.ndarray], acc: str = "f32"):
        self.cfg = cfg
        self.acc = acc
        self.lib = _lib(acc)
        self.lib.hmvo_clear()
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            rc = self.lib.hmvo_set_tensor(k.encode(), _fp(a), shape, a.ndim)
            if rc:
                raise RuntimeError(self.lib.hmvo_last_error().decode())
        c = _Cfg()
        c.backbone = _BACKBONE[cfg.backbone_type]
        c.n_levels = len(cfg.backbone_channels)
        for i, ch in enumerate(cfg.backbone_channels):
            c.channels[i] = ch
        c.num_views, c.image_size, c.heatmap_size = cfg.num_views, cfg.image_size, cfg.heatmap_size
        c.pos_mask, c.fusion_layers, c.use_gcn = cfg.pos_mask, cfg.fusion_layers, int(cfg.use_gcn)
        c.fusion = 1 if getattr(cfg, "learnable_query", False) else 0
        self._c = c

    @property
    def num_threads(self) -> int:
        return int(self.lib.hmvo_num_threads())

    def forward(self, x: np.ndarray, bbox: np.ndarray, intrinsic: np.ndarray, stages: bool = False
                ) -> Dict[str, np.ndarray]:
        """x [B,V,3,H,W] -> the reference's output dict (+ stage dumps when stages=True)."""
        cfg = self.cfg
        x = np.ascontiguousarray(x, dtype=np.float32)
        bbox = np.ascontiguousarray(bbox, dtype=np.float32)
        intrinsic = np.ascontiguousarray(intrinsic, dtype=np.float32)
        B, V, _, H, W = x.shape
        assert V == cfg.num_views
        from handmvnet_amd.spec import heatmap_size_of, level_sizes   # shape arithmetic only (no engine code involved)
        hs, ws = heatmap_size_of(cfg, H, W)
        d = cfg.feat_dim
        out = {
            "joints_crop_img": np.zeros((B, V, 21, 2), np.float32),
            "joints_cam": np.zeros((B, 21, 3), np.float32),
            "heatmap": np.zeros((B, V, 21, hs, ws), np.float32),
        }
        st = {}
        if stages:
            fh, fw = level_sizes(cfg, H, W)[0]
            st = {"feat0": np.zeros((B * V, cfg.backbone_channels[0], fh, fw), np.float32),
                  "coords_hm": np.zeros((B * V, 21, 2), np.float32),
                  "tokens": np.zeros((B, V * 21, d), np.float32),
                  "fused": np.zeros((B, 21, d), np.float32)}
        rc = self.lib.hmvo_forward(ctypes.byref(self._c), B, H, W, _fp(x), _fp(bbox), _fp(intrinsic),
                                   _fp(out["joints_crop_img"]), _fp(out["joints_cam"]), _fp(out["heatmap"]),
                                   _fp(st.get("feat0")), _fp(st.get("coords_hm")), _fp(st.get("tokens")),
                                   _fp(st.get("fused")))
        if rc:
            raise RuntimeError(f"oracle forward failed ({rc}): " + self.lib.hmvo_last_error().decode())
        out.update(st)
        return out

    def fuse_tokens(self, tokens: np.ndarray) -> Dict[str, np.ndarray]:
        """The tail alone (handmvnet.py:225-229): joints_late_fusion + joints_decoder on a token matrix [B, V*21, d] supplied by the
        caller -- e.g. the one the implementation under test captured -- so that its tail is checked apart from the conditioning of
        everything in front of the tokens.  -> {"fused" [B,21,d], "joints_cam" [B,21,3]}."""
        cfg = self.cfg
        tokens = np.ascontiguousarray(tokens, dtype=np.float32)
        B, T, d = tokens.shape
        assert T == cfg.num_views * 21 and d == cfg.feat_dim, (tokens.shape, cfg.num_views, cfg.feat_dim)
        out = {"fused": np.zeros((B, 21, d), np.float32), "joints_cam": np.zeros((B, 21, 3), np.float32)}
        rc = self.lib.hmvo_fuse_tokens(ctypes.byref(self._c), B, _fp(tokens), _fp(out["fused"]), _fp(out["joints_cam"]))
        if rc:
            raise RuntimeError(f"oracle fuse_tokens failed ({rc}): " + self.lib.hmvo_last_error().decode())
        return out
