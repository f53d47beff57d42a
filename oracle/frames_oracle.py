"""CPU restatement (numpy) of the reference's per-view frame preparation -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/datasets/ho3d.py:35-40, 136-149 (crop_and_pad_image -> ToTensor -> Resize(antialias=True)
-> Normalize) and datasets/utils.py:40-77 (crop_and_pad_image).  torchvision is absent in this image; its tensor Resize is
torch.nn.functional.interpolate(mode="bilinear", antialias=True, align_corners=False), whose separable triangle filter
(aten UpSampleKernel, `_compute_indices_min_size_weights_aa`) is restated in aa_weights() below.  Pinned by
tests/golden/frames_cases.npz: crop_and_pad_image outputs of the real reference + torch's interpolate on them
(tests/golden/make_frames_fixture.py).  "torchvision wrapper: parity unpinned" -- only the torch operator it calls is pinned.
Only tests/ may import this file.
"""
from __future__ import annotations

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)   # ho3d.py:38
STD = np.array([0.229, 0.224, 0.225], np.float32)    # ho3d.py:39


def crop_and_pad(image: np.ndarray, box) -> np.ndarray:
    """datasets/utils.py:40-77: the (y2-y1, x2-x1) window of an HWC uint8 frame, zeros where it leaves the frame."""
    h, w = image.shape[:2]
    x1, y1, x2, y2 = (int(v) for v in box)
    out = np.zeros((y2 - y1, x2 - x1, 3), np.uint8)
    sx, sy, ex, ey = max(0, x1), max(0, y1), min(w, x2), min(h, y2)
    if ex > sx and ey > sy:
        out[sy - y1:ey - y1, sx - x1:ex - x1] = image[sy:ey, sx:ex]
    return out


def aa_weights(in_size: int, out_size: int):
    """Per output index: (first input index, normalised fp32 weights) of the antialiased bilinear filter."""
    scale = np.float32(in_size) / np.float32(out_size)
    support = np.float32(scale) if scale >= 1 else np.float32(1.0)
    invscale = np.float32(1.0) / scale if scale >= 1 else np.float32(1.0)
    res = []
    for i in range(out_size):
        center = scale * np.float32(i + 0.5)
        xmin = max(int(center - support + np.float32(0.5)), 0)
        xsize = min(int(center + support + np.float32(0.5)), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs(np.float32(j + xmin) - center + np.float32(0.5)) * invscale) for j in range(xsize)], np.float32)
        w = np.maximum(w, 0)
        res.append((xmin, (w / w.sum(dtype=np.float32)).astype(np.float32)))
    return res


def resize_aa(img_chw: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Separable: width pass then height pass, fp32."""
    c, h, w = img_chw.shape
    tmp = np.empty((c, h, out_w), np.float32)
    for ox, (x0, wt) in enumerate(aa_weights(w, out_w)):
        tmp[:, :, ox] = (img_chw[:, :, x0:x0 + len(wt)] * wt).sum(axis=2, dtype=np.float32)
    out = np.empty((c, out_h, out_w), np.float32)
    for oy, (y0, wt) in enumerate(aa_weights(h, out_h)):
        out[:, oy, :] = (tmp[:, y0:y0 + len(wt), :] * wt[None, :, None]).sum(axis=1, dtype=np.float32)
    return out


def prepare_view(frame_hwc_u8: np.ndarray, box, size: int) -> np.ndarray:
    """One view: [3, size, size] fp32 as the reference's img_transform(crop_and_pad_image(frame, box)) produces;
    an empty box stands for "no joint visible" (ho3d.py:138-140: a black image through the same transform)."""
    x1, y1, x2, y2 = (int(v) for v in box)
    if x2 <= x1 or y2 <= y1:
        crop = np.zeros((10, 10, 3), np.uint8)
    else:
        crop = crop_and_pad(frame_hwc_u8, box)
    t = crop.transpose(2, 0, 1).astype(np.float32) / np.float32(255)          # ToTensor
    t = resize_aa(t, size, size)
    return (t - MEAN[:, None, None]) / STD[:, None, None]


def prepare_batch(frames: np.ndarray, boxes: np.ndarray, size: int) -> np.ndarray:
    """frames [..., Hf, Wf, 3] uint8, boxes [..., 4] int -> [..., 3, size, size] fp32."""
    lead = frames.shape[:-3]
    f = frames.reshape((-1,) + frames.shape[-3:])
    b = boxes.reshape(-1, 4)
    out = np.stack([prepare_view(f[i], b[i], size) for i in range(f.shape[0])])
    return out.reshape(lead + (3, size, size))
