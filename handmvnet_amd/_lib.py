"""ctypes binding of libhandmv.so (include/handmv.h).  There is deliberately no fallback:
if the HIP extension is missing or cannot be loaded, importing the product path fails."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMV_LIB") or os.path.join(_HERE, "libhandmv.so")   # HMV_LIB: A/B-test another build

# every symbol include/handmv.h declares
SYMBOLS = ["hmv_create", "hmv_set_tensor", "hmv_finalize_weights", "hmv_workspace_bytes", "hmv_reserve", "hmv_forward",
           "hmv_last_error", "hmv_destroy", "hmv_set_capture", "hmv_read_stage", "hmv_set_profiling", "hmv_profile_count",
           "hmv_profile_get", "hmv_op_conv2d", "hmv_op_conv2d_ex", "hmv_op_conv2d_f16", "hmv_op_conv2d_sel", "hmv_op_conv2d_rd", "hmv_op_attention", "hmv_op_attention_lq", "hmv_bench_conv", "hmv_pose_metrics", "hmv_forward_frames",
           "hmv_op_prepare_frames", "hmv_set_graphs", "hmv_graph_stats", "hmv_version", "hmv_tile_rule", "hmv_profile_get_bytes", "hmv_poison_workspace", "hmv_launch_count", "hmv_set_tail_fusion", "hmv_set_chain_fusion", "hmv_set_hr_fusion", "hmv_set_x3k16_mode", "hmv_op_hr_fuse_up", "hmv_op_attention_x3"]

HMV_OK = 0


class HmvConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("backbone", ctypes.c_int32), ("n_levels", ctypes.c_int32),
                ("channels", ctypes.c_int32 * 4), ("num_views", ctypes.c_int32), ("height", ctypes.c_int32),
                ("width", ctypes.c_int32), ("image_size", ctypes.c_int32), ("heatmap_size", ctypes.c_int32),
                ("pos_enc", ctypes.c_int32), ("fusion_layers", ctypes.c_int32), ("decoder", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("device", ctypes.c_int32), ("fusion", ctypes.c_int32)]


class HandMvError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Loads libhandmv.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HandMvError(f"{LIB_PATH} is missing: build it with `python -m handmvnet_amd.build` "
                          "(hipcc --offload-arch=gfx950). handmvnet_amd has no non-HIP fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, fp = ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p
    lib.hmv_create.argtypes = [ctypes.POINTER(HmvConfig), ctypes.POINTER(vp)]
    lib.hmv_set_tensor.argtypes = [vp, ctypes.c_char_p, fp, ctypes.POINTER(ctypes.c_int64), ci]
    lib.hmv_finalize_weights.argtypes = [vp]
    lib.hmv_workspace_bytes.argtypes = [vp, ci]
    lib.hmv_workspace_bytes.restype = ctypes.c_size_t
    lib.hmv_reserve.argtypes = [vp, ci]
    lib.hmv_poison_workspace.argtypes = [vp, ci, vp]
    lib.hmv_launch_count.argtypes = [vp]
    lib.hmv_set_tail_fusion.argtypes = [vp, ci]
    lib.hmv_set_chain_fusion.argtypes = [vp, ci]
    lib.hmv_set_hr_fusion.argtypes = [vp, ci]
    lib.hmv_forward.argtypes = [vp, ci, fp, fp, fp, fp, fp, fp, vp]
    lib.hmv_last_error.argtypes = [vp]
    lib.hmv_last_error.restype = ctypes.c_char_p
    lib.hmv_destroy.argtypes = [vp]
    lib.hmv_destroy.restype = None
    lib.hmv_set_capture.argtypes = [vp, ci]
    lib.hmv_read_stage.argtypes = [vp, ctypes.c_char_p, fp, ctypes.c_size_t, vp]
    lib.hmv_set_profiling.argtypes = [vp, ci]
    lib.hmv_profile_count.argtypes = [vp]
    lib.hmv_profile_get.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                    ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
    lib.hmv_profile_get_bytes.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double)]
    lib.hmv_profile_get_bytes.restype = ctypes.c_int
    lib.hmv_op_conv2d.argtypes = [ci, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, ci, fp, ci, fp, vp]
    lib.hmv_op_conv2d_ex.argtypes = [ci, ci, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, ci, fp, ci, fp, vp]
    lib.hmv_op_conv2d_ex.restype = ctypes.c_int
    lib.hmv_op_conv2d_f16.argtypes = [ci, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, ci, fp, ci, fp, ci, ctypes.POINTER(ctypes.c_char_p), vp]
    lib.hmv_op_conv2d_f16.restype = ctypes.c_int
    lib.hmv_op_conv2d_sel.argtypes = [ci, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, ci, fp, ci, fp, ci, ctypes.POINTER(ctypes.c_char_p), vp]
    lib.hmv_op_conv2d_sel.restype = ctypes.c_int
    lib.hmv_tile_rule.argtypes = [ci, ci, ci, ci, ci]
    lib.hmv_tile_rule.restype = ctypes.c_char_p
    lib.hmv_op_conv2d_rd.argtypes = [ci, fp, ci, ci, ci, ci, fp, fp, fp, ci, fp, ci, ctypes.POINTER(ctypes.c_char_p), vp]
    lib.hmv_op_conv2d_rd.restype = ctypes.c_int
    lib.hmv_op_attention_lq.argtypes = [ci, fp, ci, ci, fp, fp, ci, ci, ci, ci, fp, vp]
    lib.hmv_op_attention_lq.restype = ctypes.c_int
    lib.hmv_op_attention.argtypes = [ci, fp, ci, ci, ci, ci, ci, fp, vp]
    lib.hmv_op_attention_x3.argtypes = [ci, fp, ci, ci, ci, ci, ci, fp, vp]
    lib.hmv_op_hr_fuse_up.argtypes = [ci, ci, fp, ci, ci, ci, ci, ci, ctypes.POINTER(vp), ctypes.POINTER(ci), ctypes.POINTER(ci),
                                      ctypes.POINTER(vp), ctypes.POINTER(vp), ci, vp, vp]
    lib.hmv_op_attention.restype = ctypes.c_int
    lib.hmv_bench_conv.argtypes = [ci] * 13 + [ctypes.POINTER(ctypes.c_float)]
    lib.hmv_bench_conv.restype = ctypes.c_int
    lib.hmv_pose_metrics.argtypes = [ci, fp, fp, ci, ci, ci, ctypes.c_float, ctypes.c_float, ci, ci, fp, fp, vp]
    lib.hmv_pose_metrics.restype = ctypes.c_int
    lib.hmv_forward_frames.argtypes = [vp, ci, fp, ci, ci, fp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                       fp, fp, fp, fp, fp, vp]
    lib.hmv_forward_frames.restype = ctypes.c_int
    lib.hmv_op_prepare_frames.argtypes = [ci, fp, ci, ci, ci, fp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                          ci, ci, fp, vp]
    lib.hmv_op_prepare_frames.restype = ctypes.c_int
    lib.hmv_set_graphs.argtypes = [vp, ci]
    lib.hmv_set_graphs.restype = ctypes.c_int
    lib.hmv_graph_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
    lib.hmv_graph_stats.restype = ctypes.c_int
    lib.hmv_version.restype = ctypes.c_char_p
    for name in ("hmv_create", "hmv_set_tensor", "hmv_finalize_weights", "hmv_reserve", "hmv_forward", "hmv_set_capture",
                 "hmv_read_stage", "hmv_set_profiling", "hmv_profile_count", "hmv_profile_get", "hmv_op_conv2d"):
        getattr(lib, name).restype = ctypes.c_int
    _lib = lib
    return lib


def check(rc: int, handle=None) -> None:
    if rc != HMV_OK:
        msg = load().hmv_last_error(handle)
        raise HandMvError(f"libhandmv error {rc}: {msg.decode() if msg else '?'}")
