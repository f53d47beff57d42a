"""The reference's per-view frame preparation as one device op (datasets/ho3d.py:35-40, 136-149;
datasets/utils.py:40-77): uint8 HWC camera frames + integer crop windows -> normalised fp32 model input.
`HandMvNet.forward_frames` fuses the same kernel in front of the stem conv; this entry point exists for callers
that want the prepared batch itself (and for the op-level parity tests)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # ho3d.py:38
IMAGENET_STD = (0.229, 0.224, 0.225)    # ho3d.py:39


def prepare_frames(frames: torch.Tensor, crop_boxes: torch.Tensor, image_size: int, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """frames uint8 [..., Hf, Wf, 3] (device), crop_boxes int [..., 4] -> fp32 [..., 3, image_size, image_size] (NCHW view
    of the kernel's channels-last output), equal to img_transform(crop_and_pad_image(frame, box)) per view."""
    if frames.dtype != torch.uint8 or frames.shape[-1] != 3 or frames.dim() < 4:
        raise ValueError("frames must be uint8 [..., Hf, Wf, 3]")
    if not frames.is_cuda:
        raise _lib.HandMvError("handmvnet_amd runs on MI355X only: frames must be a CUDA(HIP) tensor (no CPU fallback)")
    dev = frames.device
    lead = frames.shape[:-3]
    fh, fw = frames.shape[-3], frames.shape[-2]
    f = frames.contiguous()
    boxes = crop_boxes.to(dev).reshape(-1, 4).to(torch.int32).contiguous()
    n = f.numel() // (fh * fw * 3)
    if boxes.shape[0] != n:
        raise RuntimeError("crop_boxes must hold one row per frame")
    out = torch.empty(n, image_size, image_size, 4, device=dev, dtype=torch.float32)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    with torch.cuda.device(dev):
        rc = _lib.load().hmv_op_prepare_frames(dev.index if dev.index is not None else torch.cuda.current_device(), f.data_ptr(), n,
                                               fh, fw, boxes.data_ptr(), m3, s3, image_size, image_size, out.data_ptr(),
                                               ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != _lib.HMV_OK:
        raise _lib.HandMvError(f"hmv_op_prepare_frames failed with status {rc}")
    return out[..., :3].permute(0, 3, 1, 2).reshape(tuple(lead) + (3, image_size, image_size))
