"""Drop-in mirror of the reference's ``HandMvNet`` for the inference forward pass.

Same constructor, same ``forward(x, bbox, cam_params) -> dict`` and the same ``state_dict``
key layout as /root/reference/src/models/handmvnet.py:27-266, routed to the MI355X engine
(libhandmv.so) through the C ABI of include/handmv.h.  The evaluation side of ``test_step``
(handmvnet.py:352-383, 493-517: MPJPE / PA-MPJPE / PCK-AUC) runs on the device as well
(handmvnet_amd/metrics.py); training hooks, losses and the MANO mesh step are outside the
accelerated hot path.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .spec import (BACKBONE_IDS, HotPathConfig, config_from_params, executed_keys, heatmap_size_of, level_sizes,
                   remap_legacy_keys, state_dict_layout)
from .synth import synth_state_dict


def _np(v) -> np.ndarray:
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


class HandMvNet(torch.nn.Module):
    """HandMvNet(train_params, model_params, data_params) -- handmvnet.py:28."""

    def __init__(self, train_params: dict, model_params: dict, data_params: dict, init_seed: int = 0):
        super().__init__()
        self.train_params, self.model_params, self.data_params = train_params, model_params, data_params
        self.cfg: HotPathConfig = config_from_params(train_params, model_params, data_params)
        if self.cfg.backbone_type in ("18", "34") and self.cfg.early_return != 3:
            raise NotImplementedError("ResNet-18/34 are supported with backbone_early_return=3 (every release config)")
        self.debug = train_params["debug"]
        self.num_views = self.cfg.num_views
        self.batch_size = data_params["batch_size"]
        self.feat_dim = self.cfg.feat_dim
        self.pos_enc_list = list(self.cfg.pos_enc)
        self.fusion_layers = self.cfg.fusion_layers
        self.get_vertices = model_params.get("get_vertices", False)
        # handmvnet.py:117-125 (config_from_params has already rejected unknown dataset names)
        self.auc_thresh = {"dexycb": [0.0, 0.02], "ho3d": [0.0, 0.05], "mvhand": [0.0, 0.02]}[data_params.get("name", "dexycb")]
        self.example_input_array = {  # handmvnet.py:110-115 ("just for summary")
            "x": torch.zeros(2, self.num_views, 3, 256, 256), "bbox": torch.zeros(2, self.num_views, 4),
            "cam_params": {"intrinsic": torch.zeros(2, self.num_views, 4), "extrinsic": torch.zeros(2, self.num_views, 4, 4)}}
        # the reference constructor random-initialises; ours does so deterministically
        self._weights: "OrderedDict[str, np.ndarray]" = synth_state_dict(self.cfg, init_seed)
        self._engines: Dict[tuple, ctypes.c_void_p] = {}
        self._dtype = 0   # 0 = fp32 (HMV_F32), 1 = fp16 conv stack (HMV_F16, BASELINE configs[4]), 2 = split fp16 pairs (HMV_F32X3)
        self._capture = False
        self._profiling = False
        self._graphs = None   # None: the engine's default (off unless HMV_GRAPHS=1)
        self._last_key: Optional[tuple] = None

    # ------------------------------------------------------------------ Lightning-style protocol
    def freeze(self):
        return self.eval()

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):  # noqa: D401 - mirrors nn.Module.state_dict
        """nn.Module.state_dict protocol (positional destination / prefix / keep_vars as torch accepts them): the keys land
        in `destination` under `prefix`, so a parent module's state_dict() sees them like any child's."""
        if len(args) > 0:
            destination = args[0]
        if len(args) > 1:
            prefix = args[1]
        if destination is None:
            destination = OrderedDict()
        for k, v in self._weights.items():
            destination[prefix + k] = torch.from_numpy(np.array(v))   # plain tensors either way: nothing here requires grad
        return destination

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Called by a PARENT module's load_state_dict: pick this module's keys out of the prefixed dict."""
        own = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        try:
            res = self.load_state_dict(own, strict=False)
            missing_keys.extend(prefix + k for k in res.missing_keys)
            if strict:
                unexpected_keys.extend(prefix + k for k in res.unexpected_keys)
        except RuntimeError as e:
            error_msgs.append(str(e))

    def load_state_dict(self, state_dict, strict: bool = True):
        """eval.py:27-52 semantics: legacy keys are remapped, strict=True raises on any
        missing/unexpected key or shape mismatch (same message style as torch)."""
        sd = remap_legacy_keys(state_dict)
        layout = state_dict_layout(self.cfg)
        missing = [k for k in layout if k not in sd]
        unexpected = [k for k in sd if k not in layout]
        errs = []
        for k, shape in layout.items():
            if k in sd and tuple(_np(sd[k]).shape) != tuple(shape):
                errs.append(f"size mismatch for {k}: copying a param with shape {tuple(_np(sd[k]).shape)} from checkpoint, "
                            f"the shape in current model is {tuple(shape)}.")
        if strict and (missing or unexpected):
            if unexpected:
                errs.insert(0, "Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ".")
            if missing:
                errs.insert(0, "Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ".")
        if errs:
            raise RuntimeError("Error(s) in loading state_dict for HandMvNet:\n\t" + "\n\t".join(errs))
        for k in layout:
            if k in sd:
                a = _np(sd[k])
                self._weights[k] = a.astype(np.int64) if k.endswith("num_batches_tracked") else \
                    np.ascontiguousarray(a, dtype=np.float32)
        self._drop_engines()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def half(self):
        """torch-style switch to the fp16 path: conv stack in fp16 storage + fp16 MFMA (fp32 accumulate);
        heat-map logits, soft-argmax, tokens, fusion and decoder stay fp32; inputs/outputs stay fp32."""
        if self._dtype != 1:
            self._dtype = 1
            self._drop_engines()
        return self

    def float(self):
        if self._dtype != 0:
            self._dtype = 0
            self._drop_engines()
        return self

    def float32x3(self):
        """fp32-equivalent arithmetic on the fp16 matrix cores (HMV_F32X3): every value of the conv
        stack travels as a (hi, lo) fp16 pair and every product is hi*hi + lo*hi + hi*lo with fp32 accumulation."""
        if self._dtype != 2:
            self._dtype = 2
            self._drop_engines()
        return self

    # ------------------------------------------------------------------ engine management
    def _drop_engines(self):
        if self._engines:
            lib = _lib.load()
            for h in self._engines.values():
                lib.hmv_destroy(h)
        self._engines = {}

    def __del__(self):
        try:
            self._drop_engines()
        except Exception:
            pass

    def _engine(self, height: int, width: int, device_index: int):
        key = (height, width, device_index, self._dtype)
        if key in self._engines:
            return self._engines[key]
        lib = _lib.load()
        cfg = self.cfg
        c = _lib.HmvConfig()
        c.struct_size = ctypes.sizeof(_lib.HmvConfig)
        c.backbone = BACKBONE_IDS[cfg.backbone_type]
        c.n_levels = len(cfg.backbone_channels)
        for i, ch in enumerate(cfg.backbone_channels):
            c.channels[i] = ch
        c.num_views, c.height, c.width = cfg.num_views, height, width
        c.image_size, c.heatmap_size = cfg.image_size, cfg.heatmap_size
        c.pos_enc, c.fusion_layers, c.decoder = cfg.pos_mask, cfg.fusion_layers, int(cfg.use_gcn)
        c.dtype, c.device = self._dtype, device_index
        c.fusion = int(cfg.learnable_query)
        h = ctypes.c_void_p()
        _lib.check(lib.hmv_create(ctypes.byref(c), ctypes.byref(h)))
        try:
            for k in executed_keys(cfg):
                a = np.ascontiguousarray(self._weights[k], dtype=np.float32)
                shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
                _lib.check(lib.hmv_set_tensor(h, k.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim), h)
            _lib.check(lib.hmv_finalize_weights(h), h)
            lib.hmv_set_capture(h, int(self._capture))
            lib.hmv_set_profiling(h, int(self._profiling))
            if self._graphs is not None:
                lib.hmv_set_graphs(h, int(self._graphs))
        except Exception:
            lib.hmv_destroy(h)
            raise
        self._engines[key] = h
        return h

    def reserve(self, batch: int, height: int, width: int, device=None):
        """Pre-allocates the workspace (keeps hipMalloc out of a timed region)."""
        dev = torch.device(device if device is not None else "cuda")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        h = self._engine(height, width, idx)
        _lib.check(_lib.load().hmv_reserve(h, batch), h)
        return int(_lib.load().hmv_workspace_bytes(h, batch))

    # ------------------------------------------------------------------ forward (handmvnet.py:158-266)
    def forward(self, x, bbox=None, cam_params=None):
        if not isinstance(x, torch.Tensor) or x.dim() != 5:
            raise ValueError("x must be a [b, v, 3, h, w] tensor")
        if not x.is_cuda:
            raise _lib.HandMvError("handmvnet_amd runs on MI355X only: x must be a CUDA(HIP) tensor (no CPU fallback)")
        b, v, c, hh, ww = x.shape
        if c != 3:
            raise ValueError("x must have 3 channels")
        n = b * v
        if n % self.num_views:
            # the reference's .view(-1, num_views, ...) (handmvnet.py:194) raises here as well
            raise RuntimeError(f"shape '[-1, {self.num_views}, ...]' is invalid for input of {n} frames")
        batch = n // self.num_views
        dev = x.device
        x = x.contiguous().float()
        need_cam = "crop" in self.cfg.pos_enc
        bb = it = None
        if need_cam:
            if bbox is None or cam_params is None:
                raise TypeError("pos_enc contains 'crop': bbox and cam_params['intrinsic'] are required")
            bb = bbox.to(dev).reshape(-1, 4).contiguous().float()
            it = cam_params["intrinsic"].to(dev).reshape(-1, 4).contiguous().float()
            if bb.shape[0] != n or it.shape[0] != n:
                raise RuntimeError("bbox / intrinsic must hold one row per frame")
        h = self._engine(hh, ww, dev.index if dev.index is not None else torch.cuda.current_device())
        hs_h, hs_w = heatmap_size_of(self.cfg, hh, ww)   # H/8 x W/8 at the release sizes; the conv arithmetic otherwise
        out_crop = torch.empty(batch, self.num_views, 21, 2, device=dev, dtype=torch.float32)
        out_cam = torch.empty(batch, 21, 3, device=dev, dtype=torch.float32)
        out_hm = torch.empty(batch, self.num_views, 21, hs_h, hs_w, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            rc = _lib.load().hmv_forward(h, batch, x.data_ptr(), bb.data_ptr() if bb is not None else None,
                                         it.data_ptr() if it is not None else None, out_crop.data_ptr(), out_cam.data_ptr(),
                                         out_hm.data_ptr(), ctypes.c_void_p(stream))
        _lib.check(rc, h)
        self._last_key = (hh, ww, dev.index if dev.index is not None else torch.cuda.current_device(), batch, self._dtype)
        return {"joints_crop_img": out_crop, "joints_cam": out_cam, "heatmap": out_hm}

    def forward_frames(self, frames, crop_boxes, cam_params=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                       image_size=None):
        """forward() from raw camera frames: `frames` uint8 [b, v, Hf, Wf, 3] and integer crop windows `crop_boxes`
        [b, v, 4] (x1, y1, x2, y2; may leave the frame, empty = black view) replace the reference's host-side
        crop_and_pad_image -> ToTensor -> Resize(antialias=True) -> Normalize (datasets/ho3d.py:35-40, 136-149); the
        windows also serve as `bbox` for the crop-FoV columns (ho3d.py:198).  Same return dict as forward()."""
        if not isinstance(frames, torch.Tensor) or frames.dim() != 5 or frames.shape[-1] != 3 or frames.dtype != torch.uint8:
            raise ValueError("frames must be a uint8 [b, v, Hf, Wf, 3] tensor")
        if not frames.is_cuda:
            raise _lib.HandMvError("handmvnet_amd runs on MI355X only: frames must be a CUDA(HIP) tensor (no CPU fallback)")
        b, v, fh, fw, _ = frames.shape
        n = b * v
        if n % self.num_views:
            raise RuntimeError(f"shape '[-1, {self.num_views}, ...]' is invalid for input of {n} frames")
        batch = n // self.num_views
        dev = frames.device
        size = int(image_size or self.cfg.image_size)
        frames = frames.contiguous()
        boxes = crop_boxes.to(dev).reshape(-1, 4).to(torch.int32).contiguous()
        if boxes.shape[0] != n:
            raise RuntimeError("crop_boxes must hold one row per frame")
        bb = it = None
        if "crop" in self.cfg.pos_enc:
            if cam_params is None:
                raise TypeError("pos_enc contains 'crop': cam_params['intrinsic'] is required")
            bb = boxes.float()
            it = cam_params["intrinsic"].to(dev).reshape(-1, 4).contiguous().float()
            if it.shape[0] != n:
                raise RuntimeError("intrinsic must hold one row per frame")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        h = self._engine(size, size, idx)
        out_crop = torch.empty(batch, self.num_views, 21, 2, device=dev, dtype=torch.float32)
        out_cam = torch.empty(batch, 21, 3, device=dev, dtype=torch.float32)
        out_hm = torch.empty((batch, self.num_views, 21) + tuple(heatmap_size_of(self.cfg, size, size)), device=dev, dtype=torch.float32)
        m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = _lib.load().hmv_forward_frames(h, batch, frames.data_ptr(), fh, fw, boxes.data_ptr(), m3, s3,
                                                bb.data_ptr() if bb is not None else None, it.data_ptr() if it is not None else None,
                                                out_crop.data_ptr(), out_cam.data_ptr(), out_hm.data_ptr(), ctypes.c_void_p(stream))
        _lib.check(rc, h)
        self._last_key = (size, size, idx, batch, self._dtype)
        return {"joints_crop_img": out_crop, "joints_cam": out_cam, "heatmap": out_hm}

    # ------------------------------------------------------------------ evaluation (handmvnet.py:352-383, 493-517)
    def _get_metrics(self, pred_pts, target_pts):
        """handmvnet.py:352-368: (mpjpe mm, pa_mpjpe mm, auc, norm_auc, pck_values, thresholds) for [b, n, 3]
        point sets in metres -- one device launch, one device->host copy."""
        from .metrics import PoseMetrics
        scale = 1000
        mpjpe, pa_mpjpe, auc, norm_auc, pck_values, thresholds = PoseMetrics.all_metrics(
            pred_pts, target_pts, min_threshold=self.auc_thresh[0], max_threshold=self.auc_thresh[1], steps=20)
        return mpjpe * scale, pa_mpjpe * scale, auc, norm_auc, pck_values, thresholds

    def _calculate_mpjpe(self, out, inputs, mode="train"):
        """handmvnet.py:370-427 without the Lightning logging; the MANO vertex metrics need manopth (absent)."""
        from .metrics import PoseMetrics
        pred2d, gt2d = out["joints_crop_img"], inputs["joints_crop_img"].to(out["joints_crop_img"].device)
        if "joints_img_mask" in inputs:   # models/utils.py:123-131: masked joints are zeroed on both sides
            keep = (~inputs["joints_img_mask"].to(pred2d.device)).unsqueeze(-1)
            pred2d, gt2d = pred2d * keep, gt2d * keep
        gt3d = inputs["joints_cam"].to(out["joints_cam"].device)   # like the 2D ground truth: the caller's batch may sit on the host
        mpjpe, pa_mpjpe, auc_j, norm_auc_j, pck_values_j, _ = self._get_metrics(out["joints_cam"], gt3d)
        out_metrics = {f"{mode}_mpjpe2d": PoseMetrics.mpjpe(pred2d, gt2d), f"{mode}_mpjpe": mpjpe,
                       f"{mode}_pa_mpjpe": pa_mpjpe, f"{mode}_pck_j": pck_values_j, f"{mode}_auc_j": auc_j,
                       f"{mode}_norm_auc_j": norm_auc_j}
        if self.get_vertices:
            raise NotImplementedError("get_vertices needs manopth + MANO assets (joints_to_vertices.py:14-23), absent here")
        return out_metrics

    def _eval_step(self, batch, mode):
        """The body validation_step and test_step share in the reference (handmvnet.py:468-491 / 493-517): forward +
        metrics.  The training losses are not part of this build, so "loss" is None.  Like the reference, converts
        inputs["joints_cam"] / ["root_joint"] from mm to metres IN PLACE."""
        inputs = batch["data"]
        out = self.forward(inputs["rgb"], inputs["bboxes"], batch["cam_params"])
        inputs["joints_cam"] /= 1000
        if "root_joint" in inputs:
            inputs["root_joint"] /= 1000
        return {"loss": None, "metrics": self._calculate_mpjpe(out, inputs, mode=mode)}

    def validation_step(self, batch, batch_idx=0):
        """handmvnet.py:468-491: metric keys carry the "val_" prefix (val_mpjpe is what ModelCheckpoint monitors, train.py:34)."""
        return self._eval_step(batch, "val")

    def test_step(self, batch, batch_idx=0):
        """handmvnet.py:493-517: metric keys carry the "test_" prefix."""
        return self._eval_step(batch, "test")

    # ------------------------------------------------------------------ introspection (tests / bench)
    def capture_stages(self, enable: bool = True):
        self._capture = bool(enable)
        for h in self._engines.values():
            _lib.load().hmv_set_capture(h, int(enable))

    def read_stage(self, name: str) -> torch.Tensor:
        hh, ww, idx, batch, dt = self._last_key
        h = self._engines[(hh, ww, idx, dt)]
        n, d, cfg = batch * self.num_views, self.feat_dim, self.cfg
        shape = {"feat0": (n, cfg.backbone_channels[0]) + tuple(level_sizes(cfg, hh, ww)[0]), "coords_hm": (n, 21, 2),
                 "tokens": (batch, self.num_views * 21, d), "fused": (batch, 21, d)}[name]
        out = torch.empty(shape, device=f"cuda:{idx}", dtype=torch.float32)
        stream = torch.cuda.current_stream(out.device).cuda_stream
        _lib.check(_lib.load().hmv_read_stage(h, name.encode(), out.data_ptr(), out.numel(), ctypes.c_void_p(stream)), h)
        return out

    def launch_count(self) -> int:
        """Device operations (kernels, memsets, copies) enqueued by the last eager forward."""
        hh, ww, idx, _, dt = self._last_key
        return int(_lib.load().hmv_launch_count(self._engines[(hh, ww, idx, dt)]))

    def set_tail_fusion(self, enable: bool = True):
        """Fused tail kernels on (default) / off (the launch-per-op path) for the engines built so far (A/B, tests)."""
        for h in self._engines.values():
            _lib.check(_lib.load().hmv_set_tail_fusion(h, int(enable)), h)

    def set_chain_fusion(self, enable: bool = True):
        """Chained conv3 -> next conv1 launches on (default) / off (one launch per conv, same bits) for the engines built so far."""
        for h in self._engines.values():
            _lib.check(_lib.load().hmv_set_chain_fusion(h, int(enable)), h)

    def set_hr_fusion(self, enable=True):
        """HRNet: bit 0 -- the up-sampling terms of a fuse layer as one launch / one conv launch per term; bit 1 -- a four-branch module's
        last branch on a second stream beside the branch above it / one stream.  True = 3 (default: both), False = 0 (A/B, tests)."""
        mode = 3 if enable is True else (0 if enable is False else int(enable))
        for h in self._engines.values():
            _lib.check(_lib.load().hmv_set_hr_fusion(h, mode), h)

    def poison_workspace(self, value: int = 0xFF):
        """Test hook: fills the workspace of the engine the last forward ran on with `value` bytes (0xFF = NaN patterns)."""
        hh, ww, idx, _, dt = self._last_key
        h = self._engines[(hh, ww, idx, dt)]
        stream = torch.cuda.current_stream(torch.device(f"cuda:{idx}")).cuda_stream
        _lib.check(_lib.load().hmv_poison_workspace(h, int(value), ctypes.c_void_p(stream)), h)

    def use_graphs(self, enable: bool = True):
        """hipGraph replay of repeated forwards (opt-in, see include/handmv.h: hmv_set_graphs)."""
        self._graphs = bool(enable)
        for h in self._engines.values():
            _lib.load().hmv_set_graphs(h, int(enable))

    def graph_stats(self):
        """(graphs cached, replays so far) of the engine the last forward ran on."""
        hh, ww, idx, _, dt = self._last_key
        cached, replays = ctypes.c_int32(), ctypes.c_int64()
        _lib.check(_lib.load().hmv_graph_stats(self._engines[(hh, ww, idx, dt)], ctypes.byref(cached), ctypes.byref(replays)))
        return cached.value, replays.value

    def set_profiling(self, enable=True):
        """True / 1: start a fresh record list; False / 0: pause (records kept); 2: resume without clearing."""
        self._profiling = bool(enable)
        for h in self._engines.values():
            _lib.load().hmv_set_profiling(h, int(enable))

    def profile_records(self):
        """Per-launch records of the last forward (caller must have synchronised)."""
        hh, ww, idx, _, dt = self._last_key
        h = self._engines[(hh, ww, idx, dt)]
        lib = _lib.load()
        recs = []
        for i in range(lib.hmv_profile_count(h)):
            name, label = ctypes.c_char_p(), ctypes.c_char_p()
            ms, fl, by = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
            _lib.check(lib.hmv_profile_get(h, i, ctypes.byref(name), ctypes.byref(label), ctypes.byref(ms), ctypes.byref(fl)), h)
            _lib.check(lib.hmv_profile_get_bytes(h, i, ctypes.byref(by)), h)
            recs.append({"kernel": name.value.decode(), "layer": label.value.decode(), "ms": ms.value, "flops": fl.value,
                         "bytes": by.value})
        return recs
