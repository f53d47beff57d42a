"""Device-side mirror of the reference's ``PoseMetrics`` for the methods its evaluation path calls
(/root/reference/src/models/metrics.py:6-24, 64-176; call sites handmvnet.py:352-368, 381).

Every method takes device tensors and runs ``hmv_pose_metrics`` (include/handmv.h) on the tensor's
device and the current stream; return types follow the reference (0-dim tensors for mpjpe / pa_mpjpe / pck,
Python floats and lists for pck_auc).  There is no CPU path.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _run(preds: torch.Tensor, labels: torch.Tensor, thr_min: float, thr_max: float, steps: int, procrustes: bool,
         want_aligned: bool = False):
    assert preds.shape == labels.shape
    if not preds.is_cuda or not labels.is_cuda:
        raise _lib.HandMvError("handmvnet_amd metrics run on MI355X only: preds/labels must be CUDA(HIP) tensors")
    dev = preds.device
    dim = preds.shape[-1]
    n_pts = preds.shape[-2] if preds.dim() >= 2 else 1
    n_sets = preds.numel() // (dim * n_pts)
    p = preds.detach().contiguous().float()
    g = labels.detach().to(dev).contiguous().float()
    result = torch.empty(4 + 2 * steps, device=dev, dtype=torch.float32)
    aligned = torch.empty_like(p) if want_aligned else None
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = _lib.load().hmv_pose_metrics(dev.index if dev.index is not None else torch.cuda.current_device(), p.data_ptr(),
                                          g.data_ptr(), n_sets, n_pts, dim, float(thr_min), float(thr_max), int(steps),
                                          int(procrustes), aligned.data_ptr() if aligned is not None else None,
                                          result.data_ptr(), ctypes.c_void_p(stream))
    if rc != _lib.HMV_OK:
        raise _lib.HandMvError(f"hmv_pose_metrics failed with status {rc} (n_sets={n_sets}, n_pts={n_pts}, dim={dim}, "
                               f"steps={steps})")
    return result, aligned


class PoseMetrics:
    @staticmethod
    def mpjpe(preds, labels):
        """Mean Euclidean distance over every point (metrics.py:6-12)."""
        assert preds.shape == labels.shape
        return _run(preds, labels, 0.0, 0.0, 1, False)[0][0]

    @staticmethod
    def pa_mpjpe(preds, labels):
        """MPJPE after a per-pose similarity alignment; preds, labels [B, N, 3] (metrics.py:15-24)."""
        assert preds.shape == labels.shape
        return _run(preds, labels, 0.0, 0.0, 1, True)[0][1]

    @staticmethod
    def compute_similarity_transform(S1, S2):
        """S1 [B, N, 3] after the scale/rotation/translation that brings it closest to S2 (metrics.py:128-176)."""
        assert S1.shape == S2.shape and S1.dim() == 3 and S1.shape[-1] == 3
        return _run(S1, S2, 0.0, 0.0, 1, True, want_aligned=True)[1].view_as(S1)

    @staticmethod
    def pck(preds, labels, threshold, reference_len=None):
        """Fraction of points within `threshold` of their target (metrics.py:64-88)."""
        assert preds.shape == labels.shape
        if reference_len is not None:
            raise NotImplementedError("reference_len is not used by the reference's evaluation path (handmvnet.py:359-363)")
        return _run(preds, labels, float(threshold), float(threshold), 1, False)[0][4]

    @staticmethod
    def pck_auc(preds, labels, min_threshold=0, max_threshold=0.02, steps=20, reference_len=None):
        """(auc, norm_auc, pck_values, thresholds) as Python floats / lists (metrics.py:90-123)."""
        assert preds.shape == labels.shape
        if reference_len is not None:
            raise NotImplementedError("reference_len is not used by the reference's evaluation path (handmvnet.py:359-363)")
        r = _run(preds, labels, min_threshold, max_threshold, steps, False)[0].cpu()
        return r[2].item(), r[3].item(), r[4:4 + steps].tolist(), r[4 + steps:4 + 2 * steps].tolist()

    @staticmethod
    def all_metrics(preds, labels, min_threshold=0.0, max_threshold=0.02, steps=20):
        """Everything HandMvNet._get_metrics needs from ONE launch and ONE device->host copy:
        (mpjpe, pa_mpjpe, auc, norm_auc, pck_values, thresholds), unscaled."""
        r = _run(preds, labels, min_threshold, max_threshold, steps, True)[0].cpu()
        return r[0], r[1], r[2].item(), r[3].item(), r[4:4 + steps].tolist(), r[4 + steps:4 + 2 * steps].tolist()
