"""Checkpoint ingestion for the drop-in model: the reference's ``eval.py`` helpers, same names and behaviour
(/root/reference/src/eval.py:15-52), reading a Lightning ``.ckpt`` (``{"state_dict": ..., ...}``).

One deliberate difference: the file is opened with ``torch.load(..., weights_only=True)`` so nothing in a
checkpoint is executed.  A file the safe loader refuses raises; there is no pickle fallback.
"""
from __future__ import annotations

import torch

from .spec import remap_legacy_keys

LEGACY_KEYS = ("pose_net.conv.0.weight", "sample_net.conv.0.weight")


def is_legacy_version(state_dict) -> bool:
    """eval.py:15-24: a checkpoint from before pose_net was flattened and sample_net became a ModuleList."""
    return any(k in state_dict for k in LEGACY_KEYS)


def load_checkpoint_with_legacy_fix(checkpoint_path, model, device="cpu"):
    """eval.py:27-52.  `model` is a handmvnet_amd.HandMvNet (or anything with load_state_dict)."""
    checkpoint = torch.load(checkpoint_path, map_location=device, weights_only=True)
    state_dict = checkpoint["state_dict"]
    if is_legacy_version(state_dict):
        print("[warning] Legacy version detected. Remapping keys...")
        model.load_state_dict(remap_legacy_keys(state_dict), strict=True)
        print("[info] legacy model loaded successfully.")
    else:
        model.load_state_dict(state_dict, strict=True)
    return model
